// fsk_kernels_sparse.h — SPARSE dataflow (the reference's: gather -> sort -> run-length -> K +=, as streams).
// Included by fsk_engine_sparse.hip only.
#pragma once
#include "fsk_common.h"
#include <type_traits>

namespace fsk {

#include "fsk_sparse_kernels.inc"
#include "fsk_sparse_blocks.inc"

}  // namespace fsk

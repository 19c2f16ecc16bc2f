#!/bin/bash
# tools/build_variant.sh <name> <-D flags...>: the engine built with other compile-time constants into fastsk_amd/lib/lib<name>.so
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/fastsk_amd/lib/obj_$NAME; mkdir -p $O
for u in fsk_engine fsk_engine_dense fsk_engine_dense_small fsk_engine_sparse fsk_engine_variance fsk_multi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread "$@" -c $R/fastsk_amd/csrc/$u.hip -o $O/$u.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $R/fastsk_amd/csrc/fsk_fasta.cpp -o $O/fsk_fasta.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $O/*.o -ldl -o $R/fastsk_amd/lib/lib$NAME.so
rm -rf $O
echo built $R/fastsk_amd/lib/lib$NAME.so

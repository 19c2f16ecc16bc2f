// tests/emu/rccl_emu.h — TEST INFRASTRUCTURE ONLY.
//
// The few declarations of <rccl/rccl.h> that fastsk_amd/csrc/fsk_multi.hip uses, for the CPU test build
// (tests/emu/build_emu.py, -DFSK_EMU), and the prototypes of the stand-in that implements them over the emulated
// devices' memory (tests/emu/rccl_stub.cpp). Both are compiled into tests/emu/libfastsk_emu.so and nowhere else: the
// product library includes the real header and binds the real librccl with dlopen (tests/test_abi.py checks that it
// neither exports nor needs any symbol of this file).
#pragma once
#include <cstddef>
#include <cstdint>

#include "hip_emu.h"

typedef struct emuNcclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int* version);

// ---- what the tests steer and read (ctypes on libfastsk_emu.so) ------------------------------------------------
// One fault, armed until it fires: kind 1 = the next ncclCommInitAll fails; 2 = rank `rank`'s `nth` ncclAllReduce (counted
// per rank from 0, since the last emu_rccl_reset) returns ncclSystemError without taking part; 3 = that call never takes part
// and never returns until its communicator is aborted (a peer that does not answer); 4 = the next ncclCommInitAll takes
// `nth` milliseconds. `rendezvous_ms`: how long the other ranks of a collective wait for a rank that does not come before
// they give up with ncclSystemError (the watchdog of the real library; 0 = for ever).
void emu_rccl_set_fault(int kind, int rank, int nth, int rendezvous_ms);
// counters since the last reset: [0] ncclCommInitAll calls, [1] communicators created, [2] destroyed, [3] aborted,
// [4] ncclAllReduce calls, [5] int32 calls, [6] uint64 calls, [7] float64 calls, [8] payload bytes (per rank, summed),
// [9] calls made with another current device than the communicator's, [10] collectives whose ranks disagreed on count / type,
// [11] completed collectives, [12] ranks of the last communicator
void emu_rccl_stats(int64_t out[16]);
void emu_rccl_reset(void);
}

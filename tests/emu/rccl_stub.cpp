// tests/emu/rccl_stub.cpp — TEST INFRASTRUCTURE ONLY: a stand-in for librccl over the emulated devices of hip_emu.h.
//
// What it is for: fsk_multi.hip's RCCL collective (ncclCommInitAll on a helper thread with a deadline, one communicator
// rank per listed device, ncclAllReduce(int32 | uint64 | float64, sum) issued per row band from each engine's worker thread,
// ncclCommAbort after a failure) replaces the reference's per-thread reduce (fastsk_kernel.cpp:286-315) and can only meet
// more than one rank on a multi-GPU node. Here R worker threads meet in a real rendezvous: every rank's call deposits its
// buffer, the last to arrive adds the R buffers IN RANK ORDER (element type as declared) and writes the sums to all, the
// others wait for it — so a rank that skips a call, passes another count or type, or uses another rank's communicator is
// caught (counted in emu_rccl_stats, and the collective fails), and injected failures exercise the abort / poison paths.
// Streams of the emulator are synchronous (a launch has finished when it returns), so the collective completes inside the
// call; a real ncclAllReduce only enqueues.
#include "rccl_emu.h"

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct World {
    std::mutex m;
    std::condition_variable cv;
    int ranks = 0;
    int refs = 0;          // communicators alive
    bool aborted = false;
    // the collective in flight
    unsigned gen = 0;
    int arrived = 0;
    bool failed = false;   // this generation's collective could not be carried out (mismatch, a rank that never came)
    std::vector<void*> buf;
    std::vector<size_t> count;
    std::vector<int> type;
    std::vector<int> calls;  // per rank: ncclAllReduce calls so far
};

std::mutex g_m;  // counters and the armed fault
int64_t g_stat[16];
int g_fault_kind = 0, g_fault_rank = -1, g_fault_nth = -1, g_rendezvous_ms = 0;

void bump(int i, int64_t by = 1) {
    std::lock_guard<std::mutex> lk(g_m);
    g_stat[i] += by;
}

template <typename T>
void sum_into_all(const std::vector<void*>& buf, size_t n) {
    std::vector<T> s(n);
    memcpy(s.data(), buf[0], n * sizeof(T));
    for (size_t q = 1; q < buf.size(); ++q) {
        const T* x = static_cast<const T*>(buf[q]);
        for (size_t i = 0; i < n; ++i) s[i] += x[i];
    }
    for (void* b : buf) memcpy(b, s.data(), n * sizeof(T));
}

}  // namespace

struct emuNcclComm {
    World* world;
    int rank, device;
};

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    int kind, nth;
    {
        std::lock_guard<std::mutex> lk(g_m);
        g_stat[0] += 1;
        kind = g_fault_kind;
        nth = g_fault_nth;
        if (kind == 1 || kind == 4) g_fault_kind = 0;  // fires once
    }
    if (kind == 4) std::this_thread::sleep_for(std::chrono::milliseconds(nth));
    if (kind == 1) return ncclSystemError;
    if (!comms || ndev < 1) return ncclInvalidArgument;
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devlist && devlist[a] == devlist[b]) return ncclInvalidUsage;  // (as the real library: one rank per device)
    World* w = new World;
    w->ranks = w->refs = ndev;
    w->buf.assign((size_t)ndev, nullptr);
    w->count.assign((size_t)ndev, 0);
    w->type.assign((size_t)ndev, -1);
    w->calls.assign((size_t)ndev, 0);
    for (int r = 0; r < ndev; ++r) comms[r] = new emuNcclComm{w, r, devlist ? devlist[r] : r};
    std::lock_guard<std::mutex> lk(g_m);
    g_stat[1] += ndev;
    g_stat[12] = ndev;
    return ncclSuccess;
}

static void drop(ncclComm_t comm) {
    World* w = comm->world;
    {
        std::lock_guard<std::mutex> lk(w->m);
        --w->refs;
    }
    delete comm;  // (the World itself is never freed: a call of another rank may still be waiting inside it when the last communicator goes)
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    bump(2);
    drop(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    bump(3);
    {
        std::lock_guard<std::mutex> lk(comm->world->m);
        comm->world->aborted = true;
        comm->world->cv.notify_all();
    }
    drop(comm);
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t) {
    if (!comm || !recvbuff || sendbuff != recvbuff || op != ncclSum) return ncclInvalidArgument;  // (the engine reduces in place)
    if (datatype != ncclInt32 && datatype != ncclUint64 && datatype != ncclFloat64) return ncclInvalidArgument;
    World* w = comm->world;
    const int r = comm->rank;
    int cur = -1;
    (void)hipGetDevice(&cur);
    int kind = 0, wait_ms;
    {
        std::lock_guard<std::mutex> lk(g_m);
        g_stat[4] += 1;
        g_stat[datatype == ncclInt32 ? 5 : datatype == ncclUint64 ? 6 : 7] += 1;
        g_stat[8] += (int64_t)(count * (datatype == ncclInt32 ? 4 : 8));
        if (cur != comm->device) g_stat[9] += 1;
        wait_ms = g_rendezvous_ms;
    }
    std::unique_lock<std::mutex> lk(w->m);
    const int nth = w->calls[(size_t)r]++;
    {
        std::lock_guard<std::mutex> lg(g_m);
        if ((g_fault_kind == 2 || g_fault_kind == 3) && g_fault_rank == r && g_fault_nth == nth) {
            kind = g_fault_kind;
            g_fault_kind = 0;  // fires once
        }
    }
    if (kind == 2) return ncclSystemError;
    if (kind == 3) {  // a peer that does not answer: out of the collective until somebody aborts
        w->cv.wait(lk, [&] { return w->aborted; });
        return ncclSystemError;
    }
    if (w->aborted) return ncclSystemError;
    const unsigned gen = w->gen;
    w->buf[(size_t)r] = recvbuff;
    w->count[(size_t)r] = count;
    w->type[(size_t)r] = (int)datatype;
    if (++w->arrived == w->ranks) {
        bool same = true;
        for (int q = 1; q < w->ranks; ++q) same = same && w->count[(size_t)q] == w->count[0] && w->type[(size_t)q] == w->type[0];
        if (!same) {
            bump(10);
            w->failed = true;
        } else {
            w->failed = false;
            if (datatype == ncclInt32) sum_into_all<int32_t>(w->buf, count);
            else if (datatype == ncclUint64) sum_into_all<uint64_t>(w->buf, count);
            else sum_into_all<double>(w->buf, count);
            bump(11);
        }
        w->arrived = 0;
        ++w->gen;
        w->cv.notify_all();
        return w->failed ? ncclInvalidArgument : ncclSuccess;
    }
    auto over = [&] { return w->gen != gen || w->aborted; };
    if (wait_ms > 0) {
        if (!w->cv.wait_for(lk, std::chrono::milliseconds(wait_ms), over)) {  // a rank did not come: this collective is lost
            w->aborted = true;
            w->cv.notify_all();
            return ncclSystemError;
        }
    } else {
        w->cv.wait(lk, over);
    }
    if (w->gen == gen) return ncclSystemError;  // (aborted while waiting)
    return w->failed ? ncclInvalidArgument : ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->world->ranks;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error";
        case ncclSystemError: return "unhandled system error (emulated)";
        case ncclInvalidArgument: return "invalid argument (emulated: the ranks of a collective disagreed, or a bad call)";
        case ncclInvalidUsage: return "invalid usage (emulated)";
        default: return "emulated error";
    }
}

ncclResult_t ncclGetVersion(int* version) {
    if (version) *version = 0;
    return ncclSuccess;
}

void emu_rccl_set_fault(int kind, int rank, int nth, int rendezvous_ms) {
    std::lock_guard<std::mutex> lk(g_m);
    g_fault_kind = kind;
    g_fault_rank = rank;
    g_fault_nth = nth;
    g_rendezvous_ms = rendezvous_ms;
}

void emu_rccl_stats(int64_t out[16]) {
    std::lock_guard<std::mutex> lk(g_m);
    memcpy(out, g_stat, sizeof g_stat);
}

void emu_rccl_reset(void) {
    std::lock_guard<std::mutex> lk(g_m);
    memset(g_stat, 0, sizeof g_stat);
    g_fault_kind = 0;
    g_fault_rank = g_fault_nth = -1;
    g_rendezvous_ms = 0;
}

}  // extern "C"

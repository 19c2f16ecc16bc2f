// fsk_multi.hip — ONE engine over several GPUs of this process: fsk_create_multi and the group forms of the
// C-ABI calls (include/fastsk_amd.h).
//
// What it replaces: KernelFunction::compute_kernel fans one call out over t host threads — thread r takes
// work items r, r + T, ... (fastsk_kernel.cpp:54-93, 148, 275) — and the threads add their private triangles
// into K under range locks (fastsk_kernel.cpp:286-315). Here engine r of R lives on devices[r] with a replica
// of the packed sequences, a host thread of its own and a HIP stream pair (compute, exchange); it takes
// combos r, r + R, ... of every accumulate, and the partial triangles are summed by ONE logical all-reduce:
// RCCL's ncclAllReduce over xGMI (librccl is bound with dlopen, next to the HIP runtime the process already
// uses — no link-time dependency, and no second copy beside torch's), or the engine's own peer-mapped
// reduce-scatter + all-gather kernels (FSK_COLL_P2P). The all-reduce is issued per equal-area row band on the
// exchange stream while the compute stream accumulates the next band, ordered by events, never by the host;
// as int32 when C(g,m) * max_windows^2 < 2^31 (half the link bytes). Integer sums are order free, so the
// result is bit-identical to one engine's for every R. Variance mode deals the T Welford chains over the
// engines and sums their K_hat with one fp64 all-reduce.
#include "fsk_engine_internal.h"
#include "fsk_kernels_exchange.h"

#ifdef FSK_EMU
#include "rccl_emu.h"  // tests/emu: the nccl types and the test-only stand-in the emulated build binds in place of librccl
#else
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

using namespace fsk_detail;

namespace {

// reusable barrier of the group's worker threads (host side only: nobody waits for a GPU here). With a
// deadline a waiter that has waited `timeout_ms` BREAKS the barrier: it and every other waiter — now and
// later — return false; the group is then dead (fsk_group::poison), never left waiting for an engine that
// does not come.
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, arrived = 0;
    unsigned gen = 0;
    bool broken = false;
    // every waiter, now and later, returns false at once (the group is being poisoned: nobody may sit out a deadline —
    // or, without one, wait for ever — for an engine that has already given up)
    void release() {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
    bool wait(int timeout_ms = 0) {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned g = gen;
        if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); return true; }
        auto released = [&] { return gen != g || broken; };
        if (timeout_ms <= 0) cv.wait(lk, released);
        else if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), released)) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return gen != g;
    }
};

// one persistent host thread per engine; run(fn) executes fn(r) on thread r for every r and returns the codes
struct WorkerPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<int(int)>* fn = nullptr;
    std::vector<int> rc;
    unsigned gen = 0;
    int pending = 0;
    bool stop = false;
    void start(int n) {
        rc.assign((size_t)n, 0);
        for (int r = 0; r < n; ++r)
            th.emplace_back([this, r] {
                unsigned seen = 0;
                for (;;) {
                    const std::function<int(int)>* f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = fn;
                    }
                    const int code = (*f)(r);
                    std::lock_guard<std::mutex> lk(m);
                    rc[(size_t)r] = code;
                    if (--pending == 0) cv_done.notify_all();
                }
            });
    }
    void run(const std::function<int(int)>& f) {
        std::unique_lock<std::mutex> lk(m);
        fn = &f;
        pending = (int)th.size();
        ++gen;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
};

enum class XType { I32, U64, F64 };
size_t xsize(XType t) { return t == XType::I32 ? 4 : 8; }

// In-place sum of `count` elements over the R engines' buffers; rank r's worker thread calls it with its own
// buffer and stream, every rank the same sequence of calls. Asynchronous on `stream`.
struct Collective {
    int kind = 0, ranks = 0;
    int deadline_ms = 0;  // host-side waits inside the collective give up after this long (0: never)
    virtual ~Collective() {}
    virtual int all_reduce(int r, void* buf, size_t count, XType type, hipStream_t stream, std::string& err) = 0;
    // release whatever the collective has enqueued and can never complete (a rank missing): afterwards the
    // collective is unusable
    virtual void abort() {}
};

// ---- FSK_COLL_P2P ----------------------------------------------------------------------------------------
// rank r: record "my buffer is ready"; (host barrier: every rank's pointer and event are in place) wait for
// the others' events; sum slice r over all buffers, write it to all; record "my slice is done"; (host
// barrier) wait for the others' slices. The barrier of the NEXT call keeps a fast rank from re-recording
// `done` before a slow one has enqueued its wait; `ready` is protected by this call's second barrier.
struct P2PCollective : Collective {
    std::vector<int> dev;
    std::vector<void*> buf;
    std::vector<hipEvent_t> ready, done;
    HostBarrier bar;
    int init(const std::vector<int>& devices, std::string& err) {
        kind = FSK_COLL_P2P;
        ranks = (int)devices.size();
        dev = devices;
        buf.assign(dev.size(), nullptr);
        ready.assign(dev.size(), nullptr);
        done.assign(dev.size(), nullptr);
        bar.n = ranks;
        for (size_t r = 0; r < dev.size(); ++r) {
            DeviceScope on(dev[r]);
            if (on.err != hipSuccess || hipEventCreateWithFlags(&ready[r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&done[r], hipEventDisableTiming) != hipSuccess) {
                err = "cannot create the events of the peer-to-peer exchange";
                return FSK_EDEVICE;
            }
#ifndef FSK_EMU
            for (size_t q = 0; q < dev.size(); ++q) {
                if (dev[q] == dev[r]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev[r], dev[q]) != hipSuccess || !can) {
                    err = "device " + std::to_string(dev[r]) + " cannot map the memory of device " + std::to_string(dev[q]);
                    return FSK_EDEVICE;
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(dev[q], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                    err = std::string("hipDeviceEnablePeerAccess failed: ") + hipGetErrorString(pe);
                    return FSK_EDEVICE;
                }
                (void)hipGetLastError();
            }
#endif
        }
        return FSK_OK;
    }
    ~P2PCollective() override {
        for (size_t r = 0; r < dev.size(); ++r) {
            DeviceScope on(dev[r]);
            if (ready[r]) (void)hipEventDestroy(ready[r]);
            if (done[r]) (void)hipEventDestroy(done[r]);
        }
    }
    template <typename T, int PER>
    void launch(int r, size_t count, hipStream_t stream) {
        fsk::PeerBufs pb{};
        bool aligned = true;
        for (int q = 0; q < ranks; ++q) {
            pb.p[q] = buf[(size_t)q];
            aligned = aligned && (reinterpret_cast<uintptr_t>(buf[(size_t)q]) % 16 == 0);
        }
        const u64 R = (u64)ranks;
        if (aligned && count >= (size_t)PER * 1024) {
            const u64 pieces = count / PER;
            const u64 lo = pieces * (u64)r / R, hi = pieces * ((u64)r + 1) / R;
            if (hi > lo) {
                const uint32_t blocks = (uint32_t)std::min<u64>((hi - lo + 255) / 256, 8192);
                auto k = fsk::k_p2p_allreduce_wide<T, PER>;
                FSK_LAUNCH(k, dim3(blocks), dim3(256), 0, stream, pb, ranks, lo, hi);
            }
            if (r == ranks - 1 && pieces * PER < count)  // the tail that is not a whole 16-byte piece
                FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_p2p_allreduce<T>), dim3(1), dim3(256), 0, stream, pb, ranks, pieces * PER, (u64)count);
        } else {
            const u64 lo = (u64)count * (u64)r / R, hi = (u64)count * ((u64)r + 1) / R;
            if (hi > lo) {
                const uint32_t blocks = (uint32_t)std::min<u64>((hi - lo + 255) / 256, 8192);
                FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_p2p_allreduce<T>), dim3(blocks), dim3(256), 0, stream, pb, ranks, lo, hi);
            }
        }
    }
    void abort() override { bar.release(); }
    int all_reduce(int r, void* b, size_t count, XType type, hipStream_t stream, std::string& err) override {
        buf[(size_t)r] = b;
        hipError_t he = hipEventRecord(ready[(size_t)r], stream);
        if (!bar.wait(deadline_ms)) {
            err = "peer-to-peer all-reduce: engine " + std::to_string(r) + " waited more than " + std::to_string(deadline_ms) +
                  " ms for the other engines to reach the exchange";
            return FSK_EDEVICE;
        }
        for (int q = 0; q < ranks && he == hipSuccess; ++q)
            if (q != r) he = hipStreamWaitEvent(stream, ready[(size_t)q], 0);
        if (he == hipSuccess && count > 0) {
            if (type == XType::I32) launch<int32_t, 4>(r, count, stream);
            else if (type == XType::U64) launch<u64, 2>(r, count, stream);
            else launch<double, 2>(r, count, stream);
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipEventRecord(done[(size_t)r], stream);
        if (!bar.wait(deadline_ms)) {
            err = "peer-to-peer all-reduce: engine " + std::to_string(r) + " waited more than " + std::to_string(deadline_ms) +
                  " ms for the other engines to enqueue their slices";
            return FSK_EDEVICE;
        }
        for (int q = 0; q < ranks && he == hipSuccess; ++q)
            if (q != r) he = hipStreamWaitEvent(stream, done[(size_t)q], 0);
        if (he != hipSuccess) {
            err = std::string("peer-to-peer all-reduce: ") + hipGetErrorString(he);
            return FSK_EDEVICE;
        }
        return FSK_OK;
    }
};

// ---- FSK_COLL_RCCL ---------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    std::string path;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

#ifdef FSK_EMU
// the CPU test build has no librccl and no devices: the table is bound to tests/emu/rccl_stub.cpp, which is linked
// into libfastsk_emu.so and nowhere else — everything from here down (the init thread and its deadline, the rank
// bookkeeping, payload types, abort) is the product code
RcclApi* rccl_api(std::string&) {
    static RcclApi api;
    api.handle = &api;
    api.path = "tests/emu/rccl_stub.cpp (test build)";
    api.CommInitAll = &ncclCommInitAll;
    api.CommDestroy = &ncclCommDestroy;
    api.CommAbort = &ncclCommAbort;
    api.AllReduce = &ncclAllReduce;
    api.CommCount = &ncclCommCount;
    api.GetErrorString = &ncclGetErrorString;
    api.GetVersion = &ncclGetVersion;
    return &api;
}
#else
// librccl of the ROCm the process already runs on: the image that is mapped (torch's, when torch is
// imported), else the one beside libamdhip64, else the loader's default search.
RcclApi* rccl_api(std::string& err) {
    static std::mutex m;
    static RcclApi api;
    static bool tried = false;
    static std::string why;
    std::lock_guard<std::mutex> lk(m);
    if (!tried) {
        tried = true;
        std::vector<std::string> names;
        void* h = nullptr;
        for (const char* soname : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(soname, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (h) { api.path = std::string(soname) + " (already mapped)"; break; }
        }
        if (!h) {
            Dl_info info{};
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string dir(info.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    names.push_back(dir + "librccl.so.1");
                    names.push_back(dir + "librccl.so");
                }
            }
            names.push_back("librccl.so.1");
            names.push_back("librccl.so");
            names.push_back("/opt/rocm/lib/librccl.so.1");
            for (const std::string& n : names) {
                h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
                if (h) { api.path = n; break; }
            }
        }
        if (!h) {
            why = "librccl not found (tried the mapped image, the directory of the HIP runtime, the loader path)";
        } else {
            api.handle = h;
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
            api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
            api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(h, "ncclCommCount"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
            api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
            if (!api.CommInitAll || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
                why = "librccl (" + api.path + ") lacks ncclCommInitAll / ncclAllReduce";
                api.handle = nullptr;
            }
        }
    }
    if (!api.handle) { err = why; return nullptr; }
    return &api;
}
#endif  // !FSK_EMU

struct RcclCollective : Collective {
    RcclApi* api = nullptr;
    std::vector<ncclComm_t> comm;
    std::vector<int> dev;
    bool aborted = false;
    // ncclCommInitAll runs on a helper thread that owns everything it touches: when it has not returned within
    // the deadline (a link or a peer that never answers), fsk_create_multi reports that and leaves the thread
    // behind instead of hanging with it.
    struct InitJob {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        bool abandoned = false;  // the caller gave up waiting: the thread releases whatever it still obtains
        ncclResult_t result = ncclSuccess;
        std::vector<ncclComm_t> comm;
        std::vector<int> dev;
    };
    bool init_timed_out = false;  // (fatal for the handle whatever the collective asked for: the init may still be running on the devices)
    int init(const std::vector<int>& devices, int init_deadline_ms, std::string& err) {
        kind = FSK_COLL_RCCL;
        api = rccl_api(err);
        if (!api) return FSK_EDEVICE;
        dev = devices;
        auto job = std::make_shared<InitJob>();
        job->dev = devices;
        job->comm.assign(devices.size(), nullptr);
        RcclApi* a = api;
        std::thread th([job, a] {
            const ncclResult_t r = a->CommInitAll(job->comm.data(), (int)job->dev.size(), job->dev.data());
            bool abandoned;
            {
                std::lock_guard<std::mutex> lk(job->m);
                job->result = r;
                job->done = true;
                abandoned = job->abandoned;
                job->cv.notify_all();
            }
            if (abandoned && r == ncclSuccess)  // nobody will ever use these communicators: give them back
                for (size_t q = 0; q < job->comm.size(); ++q)
                    if (job->comm[q]) {
                        DeviceScope on(job->dev[q]);
                        (void)(a->CommAbort ? a->CommAbort(job->comm[q]) : a->CommDestroy(job->comm[q]));
                    }
        });
        {
            std::unique_lock<std::mutex> lk(job->m);
            if (init_deadline_ms > 0) {
                if (!job->cv.wait_for(lk, std::chrono::milliseconds(init_deadline_ms), [&] { return job->done; })) {
                    job->abandoned = true;
                    init_timed_out = true;
                    lk.unlock();
                    th.detach();
                    err = "ncclCommInitAll over " + std::to_string(devices.size()) + " devices did not return within " +
                          std::to_string(init_deadline_ms) + " ms (fsk_config.deadline_ms / tuning deadline_ms)";
                    return FSK_EDEVICE;
                }
            } else {
                job->cv.wait(lk, [&] { return job->done; });
            }
        }
        th.join();
        if (job->result != ncclSuccess) {
            err = std::string("ncclCommInitAll failed: ") + api->GetErrorString(job->result);
            return FSK_EDEVICE;
        }
        comm = job->comm;
        ranks = (int)dev.size();
        if (api->CommCount) {
            int n = 0;
            if (api->CommCount(comm[0], &n) == ncclSuccess) ranks = n;
        }
        return FSK_OK;
    }
    ~RcclCollective() override {
        for (size_t r = 0; r < comm.size(); ++r)
            if (comm[r]) {
                DeviceScope on(dev[r]);
                if (aborted) continue;  // (ncclCommAbort has freed it)
                (void)api->CommDestroy(comm[r]);
            }
    }
    void abort() override {
        if (aborted || !api->CommAbort) return;
        aborted = true;
        for (size_t r = 0; r < comm.size(); ++r)
            if (comm[r]) {
                DeviceScope on(dev[r]);
                (void)api->CommAbort(comm[r]);
            }
    }
    int all_reduce(int r, void* b, size_t count, XType type, hipStream_t stream, std::string& err) override {
        if (count == 0) return FSK_OK;
        if (aborted) { err = "the communicator was aborted after an earlier failure"; return FSK_EDEVICE; }
        const ncclDataType_t t = type == XType::I32 ? ncclInt32 : type == XType::U64 ? ncclUint64 : ncclFloat64;
        const ncclResult_t rc = api->AllReduce(b, b, count, t, ncclSum, comm[(size_t)r], stream);
        if (rc != ncclSuccess) {
            err = std::string("ncclAllReduce failed: ") + api->GetErrorString(rc);
            return FSK_EDEVICE;
        }
        return FSK_OK;
    }
};

constexpr int MAX_BANDS = 64;  // row bands of one accumulate's all-reduce
int64_t cell_of(int64_t row) { return row * (row + 1) / 2; }

// row boundaries (multiples of the tile edge, last = N) that cut the lower triangle into bands of about equal area
std::vector<int64_t> band_edges(int64_t N, int n_bands) {
    std::vector<int64_t> e{0};
    for (int b = 1; b < n_bands; ++b) {
        const int64_t r = (int64_t)std::llround((double)N * std::sqrt((double)b / n_bands) / fsk::TILE) * fsk::TILE;
        if (r > e.back() && r < N) e.push_back(r);
    }
    e.push_back(N);
    return e;
}

}  // namespace

struct fsk_group {
    std::vector<fsk_engine*> member;  // member[0] is the handle the caller holds
    std::vector<int> device;
    std::unique_ptr<Collective> coll;
    std::vector<hipStream_t> xstream;            // the exchange stream of every engine
    std::vector<hipEvent_t> ev_xdone;            // exchange -> compute
    std::vector<std::vector<hipEvent_t>> ev_cb;  // [engine][band]: the band's kernels have run on the compute stream (compute -> exchange)
    std::vector<std::vector<hipEvent_t>> ev_xb;  // [engine][band]: the band's collective has run on the exchange stream
    std::vector<DevBuf<int32_t>> stage;          // a band of the triangle narrowed to int32
    WorkerPool pool;
    HostBarrier bar;
    std::atomic<int> failed{0};
    bool others_hold_total = false;   // engines 1.. hold a REDUCED triangle: zero them before they accumulate again
    int64_t combos_since_reset = 0;   // bounds the cells of engine 0's triangle (narrowing)
    bool bound_unknown = false;       // engine 0's triangle was bound to caller memory of unknown contents: no narrowing until a whole reset
    fsk_multi_info info{};
    // ---- fail fast (fsk_config.deadline_ms / tuning deadline_ms): every host-side wait of the exchange — the engines'
    // barriers, and the wait for a band's collective to have run on the device — gives up after deadline_ms, names
    // the stage it was in, aborts the communicator (so that stuck exchange kernels let go of the streams) and
    // leaves the group POISONED: every later call returns FSK_EDEVICE with the first failure's message.
    int deadline_ms = 0;
    int bands_in_flight = 0;          // bands of the last accumulate whose events are recorded
    std::atomic<int> poisoned{0};
    std::mutex poison_m;
    std::string poison_msg;

    int R() const { return (int)member.size(); }
    void poison(const std::string& msg) {
        {
            std::lock_guard<std::mutex> lk(poison_m);
            if (poisoned.load()) return;
            poison_msg = msg;
            poisoned.store(1);
        }
        // whoever waits for the engine that failed — at the group's barrier, inside the collective — is released at once,
        // with or without a deadline
        bar.release();
        if (coll) coll->abort();
    }
    int poisoned_rc(fsk_engine* e) {
        std::lock_guard<std::mutex> lk(poison_m);
        return e->fail(FSK_EDEVICE, "the multi-GPU group is dead after an earlier failure: %s", poison_msg.c_str());
    }
    // every worker reports its code; all leave together with the first failure (nobody is left waiting
    // inside a collective for a rank that gave up)
    // (timeout_ms: how long an engine waits for the others; 0: as long as it takes)
    bool agree(int rc, int timeout_ms) {
        if (rc) failed.store(rc);
        if (!bar.wait(timeout_ms)) return false;
        const bool ok = failed.load() == 0;
        if (!bar.wait(timeout_ms)) return false;
        return ok;
    }
    // run fn on every engine's thread; the first failing engine's message becomes the handle's
    int run(const std::function<int(int)>& fn) {
        if (poisoned.load()) return poisoned_rc(member[0]);
        failed.store(0);
        pool.run(fn);
        if (poisoned.load()) {  // the FIRST failure is the one to report: the other engines only saw its consequences
            int code = FSK_EDEVICE;
            for (int r = 0; r < R(); ++r)
                if (pool.rc[(size_t)r]) { code = pool.rc[(size_t)r]; break; }
            std::lock_guard<std::mutex> lk(poison_m);
            member[0]->err = poison_msg;
            return code;
        }
        for (int r = 0; r < R(); ++r)
            if (pool.rc[(size_t)r]) {
                if (r != 0) member[0]->err = "device " + std::to_string(device[(size_t)r]) + ": " + member[(size_t)r]->err;
                if (bar.broken) poison(member[0]->err);  // (a broken barrier cannot be used again)
                return pool.rc[(size_t)r];
            }
        return FSK_OK;
    }
};

namespace {

// engine r's share of one accumulate: its combos over every band, each band's all-reduce on the exchange
// stream behind an event, under the next band's kernels
int member_accumulate(fsk_group* g, int r, const std::vector<int32_t>& mine, const std::vector<int64_t>& edges, bool narrow) {
    fsk_engine* e = g->member[(size_t)r];
    FSK_ON_DEVICE(e);
    int rc = FSK_OK;
    if (r > 0 && g->others_hold_total) rc = one_reset_counts(e);
    hipStream_t xs = g->xstream[(size_t)r];
    // After the ranks have agreed to go on, nothing below leaves early: a rank that skipped a collective
    // would leave the others' exchange streams waiting for it for ever. Failures are collected and reported
    // at the end (the exchange then ran on whatever the triangle held); a failed collective poisons the group
    // and aborts the communicator, which releases whatever the other ranks have already enqueued.
    auto note = [&](hipError_t he, const char* what) {
        if (he != hipSuccess && !rc) rc = e->fail(FSK_EDEVICE, "%s failed: %s", what, hipGetErrorString(he));
    };
    const auto waited = [&](size_t b, const char* where) {
        const int code = e->fail(FSK_EDEVICE, "engine %d (device %d) waited more than %d ms for the other engines %s (band %zu of %zu)",
                                 r, e->cfg.device, g->deadline_ms, where, b, edges.size() - 1);
        g->poison(e->err);
        return code;
    };
    for (size_t b = 0; b + 1 < edges.size(); ++b) {
        const int64_t lo = edges[b], hi = edges[b + 1];
        if (!rc) rc = one_accumulate_rows(e, mine.data(), (int32_t)mine.size(), lo, hi);
        if (!g->agree(rc, g->deadline_ms)) {
            if (g->poisoned.load()) return rc ? rc : g->poisoned_rc(e);
            if (g->bar.broken) return waited(b, "before the band's all-reduce");
            return rc ? rc : e->fail(FSK_EDEVICE, "another engine of the group failed");
        }
        const u64 c0 = (u64)cell_of(lo), cells = (u64)cell_of(hi) - c0;
        note(hipEventRecord(g->ev_cb[(size_t)r][b], e->stream), "hipEventRecord");
        note(hipStreamWaitEvent(xs, g->ev_cb[(size_t)r][b], 0), "hipStreamWaitEvent");
#ifdef FSK_TEST_HOOKS
        {   // test builds only: this engine is late
            const fsk_tuning& t = g->member[0]->tune;
            if (t.fault_rank == r && t.fault_band == (int64_t)b && t.fault_ms > 0) {
                if (t.fault_kind == 1) std::this_thread::sleep_for(std::chrono::milliseconds(t.fault_ms));
                else if (t.fault_kind == 2) FSK_LAUNCH(fsk::k_spin_ms, dim3(1), dim3(64), 0, xs, (u64)t.fault_ms);
            }
        }
#endif
        std::string cerr;
        int crc;
        if (narrow) {
            int32_t* st = g->stage[(size_t)r].p;
            const uint32_t blocks = (uint32_t)std::min<u64>((cells + 255) / 256, 16384);
            FSK_LAUNCH(fsk::k_narrow_u64_i32, dim3(blocks), dim3(256), 0, xs, (const u64*)(e->d_K + c0), st, cells);
            crc = g->coll->all_reduce(r, st, (size_t)cells, XType::I32, xs, cerr);
            FSK_LAUNCH(fsk::k_widen_i32_u64, dim3(blocks), dim3(256), 0, xs, (const int32_t*)st, e->d_K + c0, cells);
        } else {
            crc = g->coll->all_reduce(r, e->d_K + c0, (size_t)cells, XType::U64, xs, cerr);
        }
        if (crc) {
            if (!rc) rc = e->fail(crc, "band %zu of %zu: %s", b, edges.size() - 1, cerr.c_str());
            g->poison(e->err);  // the other engines may already have enqueued their half: release them
        }
        note(hipGetLastError(), "exchange kernels");
        note(hipEventRecord(g->ev_xb[(size_t)r][b], xs), "hipEventRecord");
        if (g->poisoned.load()) return rc ? rc : g->poisoned_rc(e);
    }
    // the engine's next work (finalize, getters, another accumulate) starts after the reduced cells are in place
    note(hipEventRecord(g->ev_xdone[(size_t)r], xs), "hipEventRecord");
    note(hipStreamWaitEvent(e->stream, g->ev_xdone[(size_t)r], 0), "hipStreamWaitEvent");
    return rc;
}

// Wait, on the host and with the group's deadline, until every band's collective of the last accumulate has run
// on engine r's exchange stream. A band that does not complete in time is named, the group is poisoned and the
// communicator aborted. (Without a deadline the stream synchronisation that follows does the waiting.)
int member_await_exchange(fsk_group* g, int r) {
    fsk_engine* e = g->member[(size_t)r];
    if (g->deadline_ms <= 0 || g->bands_in_flight <= 0) return FSK_OK;
    FSK_ON_DEVICE(e);
    for (int b = 0; b < g->bands_in_flight && b < (int)g->ev_xb[(size_t)r].size(); ++b) {
        // The deadline bounds a band's EXCHANGE, so its clock starts when the band's own kernels have run (the accumulate is
        // asynchronous: a long one — many combos, the sparse dataflow at large g — is not a peer that does not answer).
        // (That wait has a bound of its own, generous — the band's kernels are this engine's own work, minutes at the largest
        // inputs —: a compute kernel that never finishes must not leave the worker polling for ever.)
        const auto tc0 = std::chrono::steady_clock::now();
        const long long compute_bound_ms = std::max<long long>(600000, 100ll * g->deadline_ms);
        for (;;) {
            const hipError_t q = hipEventQuery(g->ev_cb[(size_t)r][(size_t)b]);
            if (q == hipSuccess) break;
            (void)hipGetLastError();
            if (q != hipErrorNotReady) {
                const int code = e->fail(FSK_EDEVICE, "the kernels of band %d failed on device %d: %s", b, e->cfg.device, hipGetErrorString(q));
                g->poison(e->err);
                return code;
            }
            if (g->poisoned.load()) return g->poisoned_rc(e);
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - tc0).count() > compute_bound_ms) {
                const int code = e->fail(FSK_EDEVICE, "the kernels of band %d of %d did not finish within %lld ms on device %d (engine %d)", b,
                                         g->bands_in_flight, compute_bound_ms, e->cfg.device, r);
                g->poison(e->err);
                return code;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipEventQuery(g->ev_xb[(size_t)r][(size_t)b]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                const int code = e->fail(FSK_EDEVICE, "exchange of band %d failed on device %d: %s", b, e->cfg.device, hipGetErrorString(q));
                g->poison(e->err);
                return code;
            }
            (void)hipGetLastError();
            if (g->poisoned.load()) return g->poisoned_rc(e);
            const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (waited > g->deadline_ms) {
                const int code = e->fail(FSK_EDEVICE, "the all-reduce of band %d of %d did not complete within %d ms on device %d (engine %d): "
                                         "a peer or a link is not answering", b, g->bands_in_flight, g->deadline_ms, e->cfg.device, r);
                g->poison(e->err);
                return code;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    return FSK_OK;
}

int group_sum_combos(fsk_engine* lead, const int32_t* combos, int32_t n) {
    fsk_group* g = lead->group;
    const int R = g->R();
    const int64_t N = lead->N;
    std::vector<std::vector<int32_t>> mine((size_t)R);
    for (int32_t i = 0; i < n; ++i) mine[(size_t)(i % R)].push_back(combos[i]);  // fastsk_kernel.cpp:148,275
    // bands: more, smaller bands leave less of the last band's exchange exposed — as long as a band's tile
    // launch keeps one workgroup per tile (>= 16384 tiles), the form that stores instead of adding
    int n_bands = lead->cfg.bands;
    if (n_bands <= 0) {
        const int64_t tr = (N + fsk::TILE - 1) / fsk::TILE, tiles = tr * (tr + 1) / 2;
        n_bands = (lead->path == FSK_PATH_DENSE && N >= 8192) ? (tiles >= 16 * 16384 ? 16 : 8) : 1;
    }
    const std::vector<int64_t> edges = band_edges(N, std::max(1, std::min(n_bands, MAX_BANDS)));
    g->combos_since_reset += n;
    const double bound = (double)g->combos_since_reset * (double)lead->maxW * (double)lead->maxW;
    // every cell of every engine: <= combos since the reset x max_windows^2 — unless engine 0's triangle is caller
    // memory bound since the last whole reset, whose contents nobody has bounded
    const bool narrow = bound < 2147483648.0 && !g->bound_unknown;
    u64 largest = 0;
    for (size_t b = 0; b + 1 < edges.size(); ++b) largest = std::max<u64>(largest, (u64)(cell_of(edges[b + 1]) - cell_of(edges[b])));
    g->info.bands = (int32_t)edges.size() - 1;
    g->info.narrow = narrow ? 1 : 0;
    g->info.reduce_bytes = (int64_t)((u64)lead->pairs * (narrow ? 4 : 8));
    g->info.comm_ranks = g->coll->ranks;
    for (int r = 0; r < R && r < 16; ++r) g->info.combos_per_engine[r] = (int64_t)mine[(size_t)r].size();
    g->bands_in_flight = (int)edges.size() - 1;
    const int rc = g->run([&](int r) -> int {
        fsk_engine* e = g->member[(size_t)r];
        int rc1 = FSK_OK;
        if (narrow) {
            DeviceScope on(e->cfg.device);
            if (g->stage[(size_t)r].reserve((size_t)largest) != hipSuccess)
                rc1 = e->fail(FSK_ENOMEM, "cannot allocate %llu bytes of exchange staging", (unsigned long long)(largest * 4));
        }
        if (!g->agree(rc1, g->deadline_ms)) {
            if (g->poisoned.load()) return rc1 ? rc1 : g->poisoned_rc(e);
            if (g->bar.broken) {
                const int code = e->fail(FSK_EDEVICE, "engine %d waited more than %d ms for the other engines before the accumulate", r, g->deadline_ms);
                g->poison(e->err);
                return code;
            }
            return rc1 ? rc1 : e->fail(FSK_EDEVICE, "another engine of the group failed");
        }
        return member_accumulate(g, r, mine[(size_t)r], edges, narrow);
    });
    g->others_hold_total = true;
    return rc;
}

}  // namespace

// =============================================================================================
namespace fsk_detail {

int group_load_sequences(fsk_engine* lead, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    fsk_group* g = lead->group;
    g->others_hold_total = false;
    g->combos_since_reset = 0;
    g->bound_unknown = false;  // (one_load_sequences zeroes the triangle, bound or owned)
    g->bands_in_flight = 0;
    return g->run([&](int r) { return one_load_sequences(g->member[(size_t)r], tokens, offsets, n_train, n_test); });
}

int group_reset_counts(fsk_engine* lead, int64_t row_begin, int64_t row_end) {
    fsk_group* g = lead->group;
    const bool whole = row_end < 0;
    if (whole) {
        g->others_hold_total = false;
        g->combos_since_reset = 0;
        g->bound_unknown = false;  // (every cell is zero again)
    }
    return g->run([&](int r) {
        fsk_engine* e = g->member[(size_t)r];
        return whole ? one_reset_counts(e) : one_reset_counts_rows(e, row_begin, row_end);
    });
}

int group_accumulate(fsk_engine* lead, const int32_t* combos, int32_t n) {
    for (int32_t i = 0; i < n; ++i)
        if (combos[i] < 0 || combos[i] >= lead->ncomb) return lead->fail(FSK_EINVAL, "combo id %d out of range [0,%lld)", combos[i], (long long)lead->ncomb);
    if (n == 0) return FSK_OK;
    lead->finalized = false;
    return group_sum_combos(lead, combos, n);
}

int group_synchronize(fsk_engine* lead) {
    fsk_group* g = lead->group;
    const int rc = g->run([&](int r) {
        const int rc1 = member_await_exchange(g, r);
        return rc1 ? rc1 : one_synchronize(g->member[(size_t)r]);
    });
    if (!rc) g->bands_in_flight = 0;
    return rc;
}

void group_note_bound_counts(fsk_engine* lead) { lead->group->bound_unknown = true; }

int group_finalize(fsk_engine* lead) {
    const int rc = group_synchronize(lead);  // every engine's share of the exchange has landed
    return rc ? rc : one_finalize(lead);
}

int group_set_combo_order(fsk_engine* lead, const int32_t* order, int32_t n) {
    for (fsk_engine* e : lead->group->member) {
        const int rc = one_set_combo_order(e, order, n);
        if (rc) { if (e != lead) lead->err = e->err; return rc; }
    }
    return FSK_OK;
}

int group_set_seed(fsk_engine* lead, uint64_t seed) {
    for (fsk_engine* e : lead->group->member) e->seed = seed;
    return FSK_OK;
}

int group_set_skip_test_block(fsk_engine* lead, int32_t skip) {
    for (fsk_engine* e : lead->group->member) {
        e->cfg.skip_test_block = skip ? 1 : 0;
        e->tab_n = 0;
    }
    return FSK_OK;
}

int group_set_tuning(fsk_engine* lead, const char* key, int64_t value, std::string& err) {
    for (fsk_engine* e : lead->group->member) {
        const int rc = tuning_set(e->tune, key, value, err);
        if (rc) return rc;
    }
    return FSK_OK;
}

void group_set_profile(fsk_engine* lead, int profile) {
    for (fsk_engine* e : lead->group->member) {
        if (e->cfg.profile == profile) continue;
        DeviceScope on(e->cfg.device);
        e->harvest_times();
        e->cfg.profile = profile;
    }
}

int group_get_stats(fsk_engine* lead, fsk_stats* out) {
    fsk_group* g = lead->group;
    int rc = one_get_stats(lead, out);
    for (int r = 1; r < g->R() && !rc; ++r) {  // what the group did: sums; HIP-event times stay engine 0's
        fsk_stats s;
        rc = one_get_stats(g->member[(size_t)r], &s);
        out->combos_done += s.combos_done;  // (variance mode: the iterations of every chain)
        out->cell_updates += s.cell_updates;
        out->sort_records += s.sort_records;
        out->launches += s.launches;
        out->dense_macs += s.dense_macs;
        out->panel_bytes += s.panel_bytes;
        out->combos_issued += s.combos_issued;
    }
    // integer modes: engines 1.. start every accumulate from a reset triangle (and a reset counter)
    if (!lead->result_f64) out->combos_done = g->combos_since_reset;
    return rc;
}

int group_compute(fsk_engine* lead, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    fsk_group* g = lead->group;
    int rc = group_load_sequences(lead, tokens, offsets, n_train, n_test);
    if (rc) return rc;
    const fsk_config& c = lead->cfg;
    if (!c.approx) {  // exact: every combination
        std::vector<int32_t> all((size_t)lead->ncomb);
        for (int64_t i = 0; i < lead->ncomb; ++i) all[(size_t)i] = (int32_t)i;
        rc = group_sum_combos(lead, all.data(), (int32_t)all.size());
        return rc ? rc : group_finalize(lead);
    }
    if (!lead->order_set) {  // one seeded order, the same on every engine
        default_order(lead);
        for (fsk_engine* e : g->member) e->order = lead->order;
    }
    if (c.skip_variance) {
        std::vector<int32_t> used;
        skip_variance_combos(lead, used);
        rc = group_sum_combos(lead, used.data(), (int32_t)used.size());
        return rc ? rc : group_finalize(lead);
    }
    // variance mode: chain t on engine t mod R, then the sum of fastsk_kernel.cpp:286-315 over the engines
    const int T = approx_chains(lead), R = g->R();
    rc = g->run([&](int r) -> int {
        fsk_engine* e = g->member[(size_t)r];
        FSK_ON_DEVICE(e);
        e->finalized = false;
        int rc1 = run_variance_mode(e, T, r, R);
#ifdef FSK_TEST_HOOKS
        {   // test builds only (fault_kind 3): this engine's chains take fault_ms longer than the others'
            const fsk_tuning& t = g->member[0]->tune;
            if (t.fault_kind == 3 && t.fault_rank == r && t.fault_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(t.fault_ms));
        }
#endif
        // (no deadline on this barrier: the chains of one engine may legitimately run much longer than another's; a failure
        // elsewhere still releases it, see fsk_group::poison)
        if (!g->agree(rc1, 0)) return rc1 ? rc1 : g->poisoned.load() ? g->poisoned_rc(e) : e->fail(FSK_EDEVICE, "another engine of the group failed");
        std::string cerr;
        rc1 = g->coll->all_reduce(r, e->d_Kf64.p, (size_t)e->pairs, XType::F64, e->stream, cerr);
        if (rc1) {
            const int code = e->fail(rc1, "%s", cerr.c_str());
            g->poison(e->err);  // the other engines have enqueued their half of the collective: release them
            return code;
        }
        if (g->poisoned.load()) return g->poisoned_rc(e);
        FSK_HIP(hipStreamSynchronize(e->stream));
        return FSK_OK;
    });
    g->info.bands = 1;
    g->info.narrow = 0;
    g->info.reduce_bytes = lead->pairs * 8;
    g->info.comm_ranks = g->coll->ranks;
    return rc ? rc : one_finalize(lead);
}

void group_destroy(fsk_engine* lead) {
    fsk_group* g = lead->group;
    lead->group = nullptr;
    for (int r = 0; r < g->R(); ++r) {  // nothing of the exchange may still be in flight
        DeviceScope on(g->device[(size_t)r]);
        if (g->xstream[(size_t)r]) (void)hipStreamSynchronize(g->xstream[(size_t)r]);
        (void)hipStreamSynchronize(g->member[(size_t)r]->stream);
    }
    g->pool.shutdown();
    g->coll.reset();
    for (int r = 0; r < g->R(); ++r) {
        DeviceScope on(g->device[(size_t)r]);
        g->stage[(size_t)r].release();
        if (g->ev_xdone[(size_t)r]) (void)hipEventDestroy(g->ev_xdone[(size_t)r]);
        for (hipEvent_t ev : g->ev_cb[(size_t)r]) if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : g->ev_xb[(size_t)r]) if (ev) (void)hipEventDestroy(ev);
        if (g->xstream[(size_t)r]) (void)hipStreamDestroy(g->xstream[(size_t)r]);
    }
    for (fsk_engine* e : g->member) one_destroy(e);
    delete g;
}

}  // namespace fsk_detail

// =============================================================================================
extern "C" {

int fsk_create_multi(const fsk_config* cfg, const int32_t* devices, int32_t ndev, fsk_engine** out) {
    if (!cfg || !out || !devices) { set_create_error("null argument"); return FSK_EINVAL; }
    *out = nullptr;
    if (ndev < 1 || ndev > 16) { set_create_error("need 1..16 devices"); return FSK_EINVAL; }
    int collective = cfg->collective;
    if (collective != FSK_COLL_AUTO && collective != FSK_COLL_RCCL && collective != FSK_COLL_P2P) {
        set_create_error("collective must be FSK_COLL_AUTO, FSK_COLL_RCCL or FSK_COLL_P2P");
        return FSK_EINVAL;
    }
    bool distinct = true;
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b) distinct = distinct && devices[a] != devices[b];
    std::unique_ptr<fsk_group> g(new fsk_group);
    auto undo = [&](int rc, const std::string& msg) {
        g->coll.reset();
        for (size_t r = 0; r < g->member.size(); ++r) {
            DeviceScope on(g->device[r]);
            if (r < g->ev_xdone.size() && g->ev_xdone[r]) (void)hipEventDestroy(g->ev_xdone[r]);
            if (r < g->ev_cb.size())
                for (hipEvent_t ev : g->ev_cb[r]) if (ev) (void)hipEventDestroy(ev);
            if (r < g->ev_xb.size())
                for (hipEvent_t ev : g->ev_xb[r]) if (ev) (void)hipEventDestroy(ev);
            if (r < g->xstream.size() && g->xstream[r]) (void)hipStreamDestroy(g->xstream[r]);
        }
        for (fsk_engine* e : g->member) one_destroy(e);
        set_create_error(msg);
        return rc;
    };
    for (int r = 0; r < ndev; ++r) {
        fsk_config c = *cfg;
        c.device = devices[r];
        fsk_engine* e = nullptr;
        const int rc = fsk_create(&c, &e);
        if (rc) return undo(rc, std::string("device ") + std::to_string(devices[r]) + ": " + fsk_last_error(nullptr));
        g->member.push_back(e);
        g->device.push_back(devices[r]);
    }
    g->xstream.assign((size_t)ndev, nullptr);
    g->ev_xdone.assign((size_t)ndev, nullptr);
    g->ev_cb.assign((size_t)ndev, std::vector<hipEvent_t>(MAX_BANDS, nullptr));
    g->ev_xb.assign((size_t)ndev, std::vector<hipEvent_t>(MAX_BANDS, nullptr));
    g->stage.resize((size_t)ndev);
    for (int r = 0; r < ndev; ++r) {
        DeviceScope on(devices[r]);
        if (on.err != hipSuccess || hipStreamCreateWithFlags(&g->xstream[(size_t)r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_xdone[(size_t)r], hipEventDisableTiming) != hipSuccess)
            return undo(FSK_EDEVICE, "cannot create the exchange stream / events on device " + std::to_string(devices[r]));
        for (auto* evs : {&g->ev_cb[(size_t)r], &g->ev_xb[(size_t)r]})
            for (hipEvent_t& ev : *evs)
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
                    return undo(FSK_EDEVICE, "cannot create the band events on device " + std::to_string(devices[r]));
    }
    // what fsk_config leaves open comes from engine 0's tuning (FSK_TUNING at its fsk_create): the collective, and the
    // fail-fast bound — fsk_config.deadline_ms, else the tuning's (two minutes by default); negative: no deadline
    const fsk_tuning& tune = g->member[0]->tune;
    if (collective == FSK_COLL_AUTO) collective = (int)tune.collective;
    if (!distinct && collective == FSK_COLL_RCCL)
        return undo(FSK_EINVAL, "RCCL needs distinct devices (a device listed twice runs over FSK_COLL_P2P)");
    g->deadline_ms = cfg->deadline_ms != 0 ? cfg->deadline_ms : (int)tune.deadline_ms;
    if (g->deadline_ms < 0) g->deadline_ms = 0;
    const std::vector<int> devs(devices, devices + ndev);
    std::string why;
#ifdef FSK_EMU
    const bool try_rccl = collective == FSK_COLL_RCCL;  // (the test build's stand-in is taken only when asked for by name)
#else
    const bool try_rccl = collective != FSK_COLL_P2P && distinct;
#endif
    if (try_rccl) {
        std::unique_ptr<RcclCollective> rc(new RcclCollective);
        const int code = rc->init(devs, g->deadline_ms, why);
        if (code == FSK_OK) g->coll = std::move(rc);
        // (an init that timed out may still be running on these devices: no second collective beside it)
        else if (collective == FSK_COLL_RCCL || rc->init_timed_out) return undo(code, why);
    }
    if (!g->coll) {
        std::unique_ptr<P2PCollective> p(new P2PCollective);
        const int code = p->init(devs, why);
        if (code) return undo(code, why);
        g->coll = std::move(p);
    }
    g->coll->deadline_ms = g->deadline_ms;
    g->bar.n = ndev;
    g->info.ndev = ndev;
    for (int r = 0; r < ndev; ++r) g->info.devices[r] = devices[r];
    g->info.collective = g->coll->kind;
    g->info.comm_ranks = g->coll->ranks;
    g->pool.start(ndev);
    fsk_engine* lead = g->member[0];
    lead->group = g.release();
    *out = lead;
    return FSK_OK;
}

int fsk_get_multi_info(fsk_engine* e, fsk_multi_info* out) {
    if (!e || !out) return FSK_EINVAL;
    if (!e->group) {
        *out = fsk_multi_info{};
        return FSK_OK;
    }
    *out = e->group->info;
    return FSK_OK;
}

}  // extern "C"

// fsk_engine.hip — host side of the C ABI declared in include/fastsk_amd.h.
//
// Replaces, for the kernel-construction path only, the reference's host driver
// (FastSK::compute_kernel / compute_train, fastsk.cpp:30-188), its engine
// (KernelFunction::compute_kernel / kernel_build_parallel / get_variance,
// fastsk_kernel.cpp:24-322) and the getters (fastsk.cpp:190-237). No CPU compute fallback lives
// here: every count is produced by the HIP kernels of fsk_kernels.h, and construction fails
// loudly when no device is usable.
#include "fsk_kernels.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <array>
#include <atomic>
#include <thread>
#include <vector>

#include <cstdint>
#include <climits>

#include "../../include/fastsk_amd.h"

namespace {

thread_local std::string g_create_error;

#define FSK_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) return e->fail(FSK_EDEVICE, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t r = hipMalloc((void**)&p, n * sizeof(T));
        if (r == hipSuccess) cap = n;
        return r;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Every entry point runs on the engine's device and puts the calling thread's current device back
// afterwards: the engine shares one HIP runtime with torch, whose current device must not move
// under it (e.g. when an Engine on another GPU is garbage-collected).
struct DeviceScope {
    int prev = -1, dev;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define FSK_ON_DEVICE(e)                                                                      \
    DeviceScope fsk_on_device_((e)->cfg.device);                                             \
    if (fsk_on_device_.err != hipSuccess)                                                    \
        return (e)->fail(FSK_EDEVICE, "hipSetDevice(%d) failed: %s", (e)->cfg.device, hipGetErrorString(fsk_on_device_.err))

}  // namespace

struct fsk_engine {
    fsk_config cfg{};
    std::string err;
    int k = 0;
    int64_t ncomb = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr;       // fsk_stream_wait_engine / fsk_engine_wait_stream
    hipStream_t chain_stream = nullptr;  // variance mode: the sequential sums of a batch, under the next batches' kernels
    // fsk_reset_counts does not fill K when the next accumulate can STORE its sums instead of adding
    // them (dense dataflow, one workgroup per tile): rows [lazy_lo, lazy_hi) are zero by contract
    // but not in memory until a tile launch stores them or materialise_zero() fills them.
    int64_t lazy_lo = -1, lazy_hi = -1;
    int variance_dense_slots = 1;  // FSK_VARIANCE_DENSE_SLOTS=0: zero fill + k_welford per iteration instead (testing)
    bool store_next = false;  // variance mode, dense dataflow: the next (one-combo, whole-triangle) tile launch stores into the K it is given
    uint32_t* h_stage = nullptr;         // pinned: the packed sequences on their way to the device (fsk_load_sequences)
    size_t h_stage_cap = 0;
    bool stage_in_flight = false;
    double* h_prod = nullptr;            // pinned: one sequential sum per iteration in flight (kept across calls)
    size_t h_prod_cap = 0;

    // sequences
    bool loaded = false, finalized = false, result_f64 = false;
    int64_t N = 0, n_train = 0, n_test = 0, nfeat = 0, pairs = 0;
    uint32_t sigma = 0, Lmax = 0, Lmin = 0, maxW = 0, Vq = 0, n_panels = 0;
    int bits = 0;
    u64 V = 0;
    int path = 0;
    DevBuf<uint32_t> d_words, d_wstart, d_len, d_fstart, d_featseq;
    std::vector<uint32_t> h_len, h_fstart;
    bool featseq_ready = false;
    int force_splits = 0;      // FSK_TILE_SPLITS=n: combo splits per tile (tuning)
    uint32_t force_chunk = 0;  // FSK_DENSE_CHUNK=n: cap the count kernel's staging chunk (testing)

    // combos
    std::vector<uint8_t> all_pos;  // [ncomb][k]
    DevBuf<uint8_t> d_pos, d_allpos;  // positions of the batch at hand; of all combos (sparse dataflow, small batches)
    bool allpos_ready = false;
    std::vector<int32_t> order;
    bool order_set = false;
    uint64_t seed = 0;
    std::vector<double> stdevs;

    // counts / results
    u64* d_K = nullptr;
    bool K_owned = false;
    int64_t bound_cells = 0;
    DevBuf<u64> K_store;
    DevBuf<double> d_Kf64, d_Khat, d_prod, d_diag, d_stage, d_bsum;
    DevBuf<unsigned char> d_seqblk;
    DevBuf<u64> d_stage_u64, d_Kslots;
    DevBuf<int64_t> d_cell_idx;

    // dense scratch
    DevBuf<uint32_t> d_C4, d_C4H, d_rowmask, d_flag;  // lo / hi nibble planes, per-row hi masks
    DevBuf<uint32_t> d_keybits;   // key compaction: per-combo bitmap of the keys that occur
    DevBuf<uint16_t> d_lut, d_vc; //                  rank table and key count per combo
    bool compact = false;         // decided at load: the alphabet has a rare symbol
    std::vector<uint16_t> h_vc_cache;
    double vc_sum = 0, vc_n = 0;
    int force_compact = -1;       // FSK_COMPACT=0/1 overrides (testing)
    DevBuf<uint32_t> d_tiletab;
    uint32_t tab_t0 = 0, tab_t1 = 0, tab_n = 0;   // tile-row range the table on the device covers
    std::vector<int32_t> prep_combos;              // combos whose count panels are resident
    bool prep_valid = false, prep_overflow = false;
    // sparse scratch
    DevBuf<unsigned char> d_keys[2];      // packed sort records (u32 or u64), double-buffered
    DevBuf<uint32_t> d_blockhist, d_totals, d_tile_ent, d_ebase, d_Pk, d_Tk, d_owner_r0, d_ucount, d_uchunk, d_utot, d_list_off, d_ulist, d_part_base;
    DevBuf<u64> d_tile_stat;
    DevBuf<uint32_t> d_segc;              // chunk records of the segment scan (batches of many tiles)
    DevBuf<int> d_tile_lrh, d_tile_rs, d_tile_lth, d_tile_ts;
    DevBuf<uint2> d_E;                    // entries: {sequence, multiplicity}
    DevBuf<u64> d_sxstat, d_U;
    std::vector<uint32_t> h_owner_r0;     // owner bands of K: rows [r0[o], r0[o+1])
    uint32_t n_owners = 0, sx_rounds = 1, sx_cap = 0;
    int sx_pb = 16, sx_sb = 1, sx_keybits = 1, sx_own_shift = 13;
    bool sx_lists = false, owner_ready = false;
    // profile mode, dense dataflow: U of the last single-chunk combo list is kept, so that repeating
    // the same pass (bench steps, row bands of later passes) does not re-read every count panel
    DevBuf<u64> d_U2;
    std::vector<int32_t> u_combos;
    bool u_known = false, u_pending = false;
    u64 u_value = 0, u_extra = 0;
    int force_global_pairs = 0;  // FSK_SPARSE_GLOBAL=1: per-pair global atomics (testing)
    u64 sx_max_words = (u64)1 << 31;  // update words per batch beyond which the pairs go to K with atomics (FSK_LIST_MAX_WORDS: testing)
    // Batches are enqueued without waiting for their word counts once one batch of these sequences has
    // been sized: the stream buffer keeps headroom over the largest count seen, the kernels leave a
    // batch that does not fit alone, and the host redoes such a batch (sized exactly) when it reads the
    // counts back — at the end of an exact accumulate, at the hand-over of a variance-mode batch.
    unsigned char* h_sx_pos = nullptr;   // pinned: positions of the batches in flight ([SX_DEFER slots][exact call])
    size_t h_sx_pos_cap = 0;
    u64* h_sx_stat = nullptr;            // pinned: {pairs, words} of the batches in flight (same layout)
    size_t h_sx_stat_cap = 0;
    u64 sx_words_seen = 0;               // largest word count of a batch since the sequences were loaded
    struct SxDefer { bool active = false; u64 cap = 0; } sx_defer[8];
    int force_seg_chunks = 0;            // FSK_SEG_SCAN_CHUNKED=1: the three-launch segment scan whatever the tile count (testing)
    int sx_sync = 0;                     // FSK_SPARSE_SYNC=1: size every batch exactly (testing); also while redoing a batch
    u64 sx_guard_cap = 0;                // FSK_SPARSE_GUARD_CAP=n: pretend the stream buffer holds n words (testing the redo)
    u64 sx_redone = 0;                   // batches redone because they did not fit
    int tile_dma = 1;            // FSK_TILE_DMA=0: register-staged tile kernel instead of the direct-to-LDS one (testing)
    int compact_dma = 0;         // FSK_COMPACT_DMA=1: direct-to-LDS k_dense_tile_dma_compact for key-compacted panels (measured slower on
                                 // config 3: its flagged rows take the generic remainder, not k_dense_tile_compact's side-aware one)

    fsk_stats st{};

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    fsk::SeqView view() const {
        return fsk::SeqView{d_words.p, d_wstart.p, d_len.p, (uint32_t)N, bits};
    }
    void tic() {
        if (cfg.profile) (void)hipEventRecord(ev0, stream);
    }
    void toc(double* acc) {
        if (!cfg.profile) return;
        (void)hipEventRecord(ev1, stream);
        (void)hipEventSynchronize(ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        *acc += ms;
    }
};

namespace {

int64_t n_choose_k(int n, int k) {  // nchoosek, shared.cpp:335-345 (exact in 64 bits)
    if (k < 0 || k > n) return 0;
    if (k * 2 > n) k = n - k;
    int64_t r = 1;
    for (int i = 1; i <= k; ++i) {
        if (r > (INT64_MAX / 2) / (n - k + i)) return INT64_MAX / 2;  // saturate: callers reject >= 2^31
        r = r * (n - k + i) / i;
    }
    return r;
}

// all k-subsets of {0..g-1}, lexicographic (getCombinations, shared.cpp:347-360)
void enumerate_combos(int g, int k, std::vector<uint8_t>& out) {
    std::vector<int> pos(k);
    for (int i = 0; i < k; ++i) pos[i] = i;
    out.clear();
    while (true) {
        for (int i = 0; i < k; ++i) out.push_back((uint8_t)pos[i]);
        int i = k - 1;
        while (i >= 0 && pos[i] == g - k + i) --i;
        if (i < 0) break;
        ++pos[i];
        for (int j = i + 1; j < k; ++j) pos[j] = pos[j - 1] + 1;
    }
}

// host-side helpers of fsk_load_sequences: contiguous ranges of [0, n) on a few threads
int host_threads_for(int64_t work_items) {
    if (work_items < ((int64_t)1 << 18)) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned want = work_items < ((int64_t)1 << 21) ? 4u : 8u;  // (a thread start costs ~50 us)
    return (int)std::max(1u, std::min(want, hw ? hw : 1u));
}
template <typename F>
void parallel_ranges(int64_t n, int nt, F&& fn) {
    if (nt <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(t, n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

// two phases over the same thread team with `between` run by one thread in the middle (one thread start
// per call instead of two; a spinning barrier: the team is a handful of threads for a fraction of a millisecond)
template <typename F1, typename FM, typename F2>
void parallel_two_phase(int nt, F1&& phase1, FM&& between, F2&& phase2) {
    if (nt <= 1) { phase1(0); between(); phase2(0); return; }
    std::atomic<int> arrived{0}, go{0};
    auto body = [&](int t) {
        phase1(t);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {
            between();
            go.store(1, std::memory_order_release);
        } else {
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        }
        phase2(t);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
}

uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr size_t LDS_BUDGET = 150 * 1024;       // of 160 KiB per CU
constexpr u64 DENSE_MAX_KEYS = 16384;           // count panels: alphabet^k <= this (DNA up to k = 7)
constexpr size_t SPARSE_MAX_RECORDS = 1u << 25; // records per sort batch

// k_dense_count LDS plan: (CH + g - 1) staged symbols x 64 sequences + the u16 histogram of one
// key sweep (512 B per key quad). Symbols get what they need up to 64 KiB (all windows in one
// staging pass when possible), the histogram gets the rest (fewer sweeps over large key spaces).
struct DensePlan { uint32_t CH = 0, Vcq = 0; size_t lds = 0; };
DensePlan dense_plan(uint32_t maxW, int g, uint32_t Vq, size_t extra = 0) {
    DensePlan p;
    const size_t sym_cap = (size_t)64 << 10;
    const size_t want_sym = (size_t)(maxW + g - 1) * fsk::PANEL;
    size_t sym = std::min(want_sym, sym_cap);
    if (sym + extra + 1024 > LDS_BUDGET) return p;
    size_t hist_room = LDS_BUDGET - sym - extra;
    uint32_t vcq = (uint32_t)std::min<size_t>(Vq, hist_room / 512);
    if (vcq < Vq) {       // several sweeps: each must start on an 8-key boundary (4-bit panels
        vcq &= ~1u;       // pack 8 keys per dword)
        if (vcq < 2) return p;
    }
    if (sym / fsk::PANEL < (size_t)g) return p;
    p.Vcq = vcq;
    p.CH = (uint32_t)std::min<size_t>(maxW, sym / fsk::PANEL - (size_t)(g - 1));
    p.lds = (size_t)(p.CH + g - 1) * fsk::PANEL + (size_t)p.Vcq * 512 + extra;
    return p;
}

// Which dataflow is cheaper per combo (path = auto)? The dense one multiplies every pair of
// sequences over the whole key space at the v_dot8 rate; the sparse one issues one scattered
// 64-bit atomic per (run, pair). Rates measured on MI355X.
bool dense_is_cheaper(const fsk_engine* e) {
    const double N = (double)e->N, V = (double)e->V;
    const double W = (double)e->nfeat / std::max(1.0, N);                 // windows per sequence
    // dense: every pair of sequences over the whole key space at the tile kernel's rate, plus the
    // count kernel: one pass over the windows per histogram sweep (large key spaces need many)
    // (count kernel, fitted on MI355X: 1e-12 s per window for the first histogram sweep, 2.6e-12 s
    // per window for every further sweep when the window keys are cached in LDS — as
    // accumulate_dense arranges when they fit — and 8e-12 s when they are recomputed; 7e-13 s per
    // (sequence, key) for zeroing and reading out the histograms)
    DensePlan plan = dense_plan(e->maxW, e->cfg.g, e->Vq);
    double sweep_cost = 8e-12;
    if (plan.Vcq && plan.Vcq < e->Vq && plan.CH >= e->maxW) {
        const DensePlan p2 = dense_plan(e->maxW, e->cfg.g, e->Vq, (size_t)e->maxW * fsk::PANEL * sizeof(uint16_t));
        if (p2.CH >= e->maxW && p2.Vcq >= 64) { plan = p2; sweep_cost = 2.6e-12; }
    }
    const double sweeps = plan.Vcq ? std::ceil((double)e->Vq / plan.Vcq) : 1.0;
    // (tile kernel: 3.0e14 count-MAC/s with thousands of tiles, ~2.4e14 with few)
    const double dense = 0.5 * N * N * (double)(((e->Vq + 1) / 2) * 8) / (N < 8192.0 ? 2.4e14 : 3.0e14) +
                         (double)e->nfeat * (1e-12 + (sweeps - 1.0) * sweep_cost) + N * V * 7e-13;
    // sparse: sort + segments per g-mer, then one update per (run, pair). d = sequences holding a
    // given key. Update streams summed in LDS by the owner bands (N up to ~23,000): 1.7e11 updates/s
    // through emit + consume, a band with several LDS rounds re-reads its stream once per round;
    // per-pair global atomics beyond that: 1.6e10/s. Extraction + sort + segments: 3.1e-11 s per g-mer.
    const double d = N * (1.0 - std::exp(-W / V));
    const double U = V * d * (d + 1.0) / 2.0;
    const double rate = e->sx_lists ? 1.7e11 / (1.0 + 0.3 * ((double)e->sx_rounds - 1.0)) : 1.6e10;
    const double sparse = U / rate + (double)e->nfeat * 3.1e-11;
    return dense <= sparse;
}

int choose_path(fsk_engine* e) {
    bool dense_ok = e->V <= DENSE_MAX_KEYS && e->k <= 16 && e->Lmax < 65536 && dense_plan(e->maxW, e->cfg.g, e->Vq).CH > 0;
    if (e->cfg.path == FSK_PATH_DENSE) {
        if (!dense_ok)
            return e->fail(FSK_EUNSUPPORTED, "dense path needs alphabet^k <= %llu and the panel histogram to fit in LDS",
                           (unsigned long long)DENSE_MAX_KEYS);
        e->path = FSK_PATH_DENSE;
    } else if (e->cfg.path == FSK_PATH_SPARSE) {
        e->path = FSK_PATH_SPARSE;
    } else {
        e->path = dense_ok && dense_is_cheaper(e) ? FSK_PATH_DENSE : FSK_PATH_SPARSE;
    }
    return FSK_OK;
}

// ---------------------------------------------------------------------------------------------
// sparse dataflow: owner bands of K. The update stream of a band is summed in LDS by one workgroup,
// so a band is a range of whole rows with about 8192 cells (the LDS budget of k_sx_consume bounds
// it: SX_CAP cells per round, at most SX_MAX_ROUNDS rounds over the band's stream).
constexpr uint32_t SX_CAP = 16384;        // u32 cells of K one k_sx_consume workgroup holds in LDS (64 KiB)
constexpr uint32_t SX_MAX_ROUNDS = 16;
constexpr u64 SX_MAX_LIST_WORDS = (u64)1 << 31;
constexpr int FSK_RETRY_UNGROUPED = 1;  // internal: a per-slot sparse batch has to be redone one combo at a time

void plan_owner_bands(fsk_engine* e) {
    // band o = the rows whose first cell index lies in [o << t, (o + 1) << t): a row's band is a shift
    // of its triangular index, bands hold about 2^t cells (2^t + N at most: the last row of a band is
    // kept whole) and can be empty when a single row is longer than 2^t cells.
    const u64 N = (u64)e->N, cells = N * (N + 1) / 2;
    int t = 13;
    while ((((cells + (((u64)1) << t) - 1) >> t)) > (u64)fsk::SX_MAX_OWNERS) ++t;
    e->sx_own_shift = t;
    e->n_owners = (uint32_t)((cells + (((u64)1) << t) - 1) >> t);
    e->h_owner_r0.assign((size_t)e->n_owners + 1, (uint32_t)N);
    u64 largest = 0;
    {
        uint32_t o = 0;  // r0[o] = first row whose triangular index reaches o << t
        for (u64 i = 0; i < N && o <= e->n_owners; ++i)
            while (o <= e->n_owners && (i * (i + 1) / 2) >= ((u64)o << t)) e->h_owner_r0[o++] = (uint32_t)i;
        for (uint32_t q = 0; q < e->n_owners; ++q) {
            const u64 a = e->h_owner_r0[q], b = e->h_owner_r0[q + 1];
            largest = std::max(largest, b * (b + 1) / 2 - a * (a + 1) / 2);
        }
    }
    int L = 1;
    while (((u64)1 << L) < largest) ++L;
    e->sx_pb = 32 - L;
    e->sx_rounds = (uint32_t)std::max<u64>(1, (largest + SX_CAP - 1) / SX_CAP);
    e->sx_cap = (uint32_t)std::max<u64>(1, std::min<u64>(SX_CAP, largest));
    e->sx_lists = e->n_owners <= (uint32_t)fsk::SX_MAX_OWNERS && e->sx_rounds <= SX_MAX_ROUNDS && e->sx_pb >= 8;
    e->owner_ready = false;
}

// `pos_pin` / `stat_pin`: pinned staging of this batch (positions in, {pairs, words} out), untouched by
// anyone else until the batch's counts have been read. `guard_cap` == 0: the call waits for the
// counts and sizes the streams exactly; else it only enqueues, for streams of at most guard_cap words.
template <typename RecT>
int sparse_batch(fsk_engine* e, const int32_t* combos, int nb, u64* K, int64_t row0, int64_t row1, u64 slot_stride,
                 unsigned char* pos_pin, u64* stat_pin, u64 guard_cap) {
    const uint32_t nfeat = (uint32_t)e->nfeat;
    const size_t nrec = (size_t)nb * nfeat;
    if (nrec == 0) return FSK_OK;
    // Only the k-mer bits are sorted: the records of a slot are generated in sequence order and every
    // LSD pass is stable, so equal k-mers end up contiguous with their sequence ids ascending.
    int keybits = 1;
    while (keybits < 62 && ((u64)1 << keybits) < (u64)e->V) ++keybits;
    const int sb = e->sx_sb;
    const int passes = (keybits + 7) / 8;
    const int nbits = (keybits + passes - 1) / passes;  // the k-mer bits split evenly: 19 bits sort as 7 + 6 + 6, not 8 + 8 + 3
    const uint32_t dmask = (1u << nbits) - 1u;
    const uint32_t tps = (nfeat + fsk::SX_TILE - 1) / fsk::SX_TILE;   // sort tiles per slot
    const uint32_t tpg = (nfeat + fsk::SG_TILE - 1) / fsk::SG_TILE;   // segment tiles per slot
    const uint32_t ntiles = tpg * (uint32_t)nb;
    const bool lists = e->sx_lists && !e->force_global_pairs;
    const uint32_t O = e->n_owners;
    for (int b = 0; b < 2; ++b) FSK_HIP(e->d_keys[b].reserve(nrec * sizeof(RecT)));
    FSK_HIP(e->d_blockhist.reserve((size_t)256 * tps * nb));
    FSK_HIP(e->d_totals.reserve((size_t)256 * nb));
    FSK_HIP(e->d_tile_ent.reserve(ntiles));
    FSK_HIP(e->d_tile_lrh.reserve(ntiles));
    FSK_HIP(e->d_tile_rs.reserve(ntiles));
    FSK_HIP(e->d_ebase.reserve((size_t)ntiles + 1));
    FSK_HIP(e->d_E.reserve(nrec));
    FSK_HIP(e->d_Pk.reserve(nrec));
    // skip_test_block: test rows pair only with the train entries of their runs (and themselves)
    const uint32_t skip_from = e->cfg.skip_test_block && e->n_test > 0 ? (uint32_t)e->n_train : 0xffffffffu;
    const bool skipping = skip_from != 0xffffffffu;
    if (skipping) {
        FSK_HIP(e->d_Tk.reserve(nrec));
        FSK_HIP(e->d_tile_lth.reserve(ntiles));
        FSK_HIP(e->d_tile_ts.reserve(ntiles));
    }
    FSK_HIP(e->d_sxstat.reserve(3));
    FSK_HIP(e->d_tile_stat.reserve((size_t)2 * ntiles));
    FSK_HIP(e->d_pos.reserve((size_t)nb * e->k));
    if (!e->owner_ready) {
        FSK_HIP(e->d_owner_r0.reserve(e->h_owner_r0.size()));
        FSK_HIP(hipMemcpyAsync(e->d_owner_r0.p, e->h_owner_r0.data(), e->h_owner_r0.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        e->owner_ready = true;  // (h_owner_r0 lives as long as the engine: no wait needed)
    }
    const uint32_t nchunks = (ntiles + fsk::UC_CHUNK - 1) / fsk::UC_CHUNK;
    if (lists) {
        FSK_HIP(e->d_ucount.reserve((size_t)O * ntiles));
        FSK_HIP(e->d_uchunk.reserve((size_t)O * nchunks));
        FSK_HIP(e->d_utot.reserve(O));
        FSK_HIP(e->d_list_off.reserve((size_t)O + 1));
        FSK_HIP(e->d_part_base.reserve((size_t)O + 1));
    }
    fsk::SxIds ids{};
    const bool by_id = nb <= 16;  // (variance mode: a handful of combos per batch) positions from the resident table
    if (by_id) {
        if (!e->allpos_ready) {
            FSK_HIP(e->d_allpos.reserve(e->all_pos.size()));
            FSK_HIP(hipMemcpy(e->d_allpos.p, e->all_pos.data(), e->all_pos.size(), hipMemcpyHostToDevice));
            e->allpos_ready = true;
        }
        for (int s = 0; s < nb; ++s) ids.id[s] = combos[s];
    } else {
        for (int s = 0; s < nb; ++s)
            memcpy(pos_pin + (size_t)s * e->k, &e->all_pos[(size_t)combos[s] * e->k], e->k);
        FSK_HIP(hipMemcpyAsync(e->d_pos.p, pos_pin, (size_t)nb * e->k, hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipMemsetAsync(e->d_sxstat.p, 0, 3 * sizeof(u64), e->stream));
    }

    RecT* rec[2] = {(RecT*)e->d_keys[0].p, (RecT*)e->d_keys[1].p};

    e->tic();
    FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_extract<RecT>), dim3(tps, nb), dim3(256), 0, e->stream, e->view(), e->d_featseq.p,
               e->d_fstart.p, nfeat, tps, e->k, e->sigma, sb, by_id ? (const uint8_t*)e->d_allpos.p : (const uint8_t*)e->d_pos.p, rec[0],
               e->d_blockhist.p, dmask, ids, by_id ? e->d_sxstat.p : (u64*)nullptr);
    e->toc(&e->st.ms_extract);
    e->st.launches += 1;

    e->tic();
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = sb + nbits * p;
        if (p > 0)  // (the extraction counted the first pass's digits)
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_hist<RecT>), dim3(tps, nb), dim3(256), 0, e->stream, rec[cur], nfeat, tps, shift, dmask,
                       e->d_blockhist.p);
        FSK_LAUNCH(fsk::k_sx_scan_slot, dim3(nb), dim3(1024), 0, e->stream, e->d_blockhist.p, tps, e->d_totals.p);
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_scatter<RecT>), dim3(tps, nb), dim3(256), 0, e->stream, rec[cur], rec[cur ^ 1], nfeat,
                   tps, shift, nbits, e->d_blockhist.p, e->d_totals.p);
        cur ^= 1;
        e->st.launches += 3;
    }
    e->toc(&e->st.ms_sort);
    e->st.sort_records += nrec;
    e->st.sort_passes = passes;

    e->tic();
    const uint32_t maxprod = (1u << e->sx_pb) - 1u;
    const uint32_t cmax = maxprod / std::max<uint32_t>(1u, e->maxW);  // multiplicities up to here: one word per pair
    FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_seg_count<RecT>), dim3(tpg, nb), dim3(256), 0, e->stream, rec[cur], nfeat, tpg, sb,
               e->d_tile_ent.p, e->d_tile_lrh.p, skip_from, skipping ? e->d_tile_lth.p : (int*)nullptr);
    {
        const int* lth = skipping ? (const int*)e->d_tile_lth.p : (const int*)nullptr;
        int* ts = skipping ? e->d_tile_ts.p : (int*)nullptr;
        if (ntiles <= 4096u && !e->force_seg_chunks) {  // one workgroup walks the tile records
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(1), dim3(1024), 0, e->stream, (const uint32_t*)e->d_tile_ent.p, (const int*)e->d_tile_lrh.p, ntiles,
                       e->d_ebase.p, e->d_tile_rs.p, lth, ts, (const uint32_t*)nullptr, (const int*)nullptr, (const int*)nullptr,
                       (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr);
        } else {  // chunk totals, the same scan over the chunk records, the chunks with their carries
            const uint32_t nch = (ntiles + 1023u) / 1024u;
            FSK_HIP(e->d_segc.reserve((size_t)6 * (nch + 1)));
            uint32_t* c_tot = e->d_segc.p;
            int* c_lrh = reinterpret_cast<int*>(c_tot + (nch + 1));
            int* c_lth = c_lrh + (nch + 1);
            uint32_t* c_ex = reinterpret_cast<uint32_t*>(c_lth + (nch + 1));
            int* c_h = reinterpret_cast<int*>(c_ex + (nch + 1));
            int* c_t = c_h + (nch + 1);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(nch), dim3(1024), 0, e->stream, (const uint32_t*)e->d_tile_ent.p, (const int*)e->d_tile_lrh.p, ntiles,
                       (uint32_t*)nullptr, (int*)nullptr, lth, (int*)nullptr, (const uint32_t*)nullptr, (const int*)nullptr, (const int*)nullptr,
                       c_tot, c_lrh, c_lth);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(1), dim3(1024), 0, e->stream, (const uint32_t*)c_tot, (const int*)c_lrh, nch, c_ex, c_h,
                       skipping ? (const int*)c_lth : (const int*)nullptr, skipping ? c_t : (int*)nullptr, (const uint32_t*)nullptr,
                       (const int*)nullptr, (const int*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(nch), dim3(1024), 0, e->stream, (const uint32_t*)e->d_tile_ent.p, (const int*)e->d_tile_lrh.p, ntiles,
                       e->d_ebase.p, e->d_tile_rs.p, lth, ts, (const uint32_t*)c_ex, (const int*)c_h, (const int*)c_t, (uint32_t*)nullptr,
                       (int*)nullptr, (int*)nullptr);
            e->st.launches += 2;
        }
    }
    FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_seg_write<RecT>), dim3(tpg, nb), dim3(256), 0, e->stream, rec[cur], nfeat, tpg, sb,
               e->d_ebase.p, e->d_tile_rs.p, e->d_E.p, e->d_Pk.p, e->sx_own_shift, O,
               lists ? e->d_ucount.p : (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, e->d_tile_stat.p,
               skip_from, skipping ? (const int*)e->d_tile_ts.p : (const int*)nullptr, skipping ? e->d_Tk.p : (uint32_t*)nullptr);
    stat_pin[0] = stat_pin[1] = 0;
    e->st.launches += 3;
    u64 words = 0;
    if (lists) {  // where every (tile, owner) share of the update streams starts (+ the batch's pair and word totals)
        FSK_LAUNCH(fsk::k_sx_ucol_sum, dim3(nchunks), dim3(256), 0, e->stream, (const uint32_t*)e->d_ucount.p, ntiles, O, e->d_uchunk.p,
                   (const u64*)e->d_tile_stat.p, e->d_sxstat.p, stat_pin);
        FSK_LAUNCH(fsk::k_sx_ucol_scan, dim3(O), dim3(256), 0, e->stream, e->d_uchunk.p, nchunks, O, e->d_utot.p);
        FSK_LAUNCH(fsk::k_sx_ucol_apply, dim3(nchunks), dim3(256), 0, e->stream, e->d_ucount.p, ntiles, O, (const uint32_t*)e->d_uchunk.p,
                   (const uint32_t*)e->d_utot.p, e->d_list_off.p);
        e->st.launches += 3;
    } else {
        FSK_LAUNCH(fsk::k_sx_stat_sum, dim3(32), dim3(256), 0, e->stream, (const u64*)e->d_tile_stat.p, ntiles, e->d_sxstat.p, stat_pin);
        e->st.launches += 1;
    }
    const bool guarded = guard_cap != 0;
    u64 cap_words = ~(u64)0;
    if (guarded) {
        words = lists ? std::max<u64>(1, std::min(e->sx_words_seen, guard_cap)) : 0;  // (sizes the parts; the kernels read the true offsets)
        cap_words = guard_cap;
    } else {
        FSK_HIP(hipStreamSynchronize(e->stream));  // the update streams are sized exactly
        e->u_extra += stat_pin[0];
        words = stat_pin[1];
        e->sx_words_seen = std::max(e->sx_words_seen, words);
    }
    e->toc(&e->st.ms_segment);

    e->tic();
    const bool use_lists = lists && words < e->sx_max_words;
    if (slot_stride != 0 && !use_lists) return FSK_RETRY_UNGROUPED;  // (nothing of this batch has touched K yet)
    if (use_lists) {
        if (words > 0 || slot_stride != 0) {
            if (!guarded && (size_t)words > e->d_ulist.cap)  // (grown with headroom: the batches of a pass differ by a few percent)
                FSK_HIP(e->d_ulist.reserve((size_t)std::max<u64>(1, words + words / 4)));
            // (function pointers: a template-id with a comma cannot pass through the launch macro)
            auto k_emit = skipping ? fsk::k_sx_emit<false, true> : fsk::k_sx_emit<false, false>;
            FSK_LAUNCH(k_emit, dim3(ntiles), dim3(fsk::EM_THREADS), 0, e->stream, (const uint2*)e->d_E.p, (const uint32_t*)e->d_Pk.p,
                       (const uint32_t*)e->d_ebase.p, (const uint32_t*)e->d_owner_r0.p, e->sx_own_shift, O, (const uint32_t*)e->d_list_off.p,
                       (const uint32_t*)e->d_ucount.p, e->d_ulist.p, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, e->sx_pb, K, tpg,
                       slot_stride, skipping ? (const uint32_t*)e->d_Tk.p : (const uint32_t*)nullptr, (const u64*)e->d_sxstat.p, cap_words);
            const size_t lds = (size_t)e->sx_cap * sizeof(uint32_t);
#ifndef FSK_EMU
            FSK_HIP(hipFuncSetAttribute((const void*)fsk::k_sx_consume, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
            // parts of about `target` words: ~1024 workgroups, and never so short that the flush of a
            // part (up to sx_cap cells) outweighs the words it summed
            const uint32_t target = (uint32_t)std::max<u64>((u64)4 * e->sx_cap, (words + 1023) / 1024);
            const uint32_t max_parts = O + (uint32_t)(((guarded ? guard_cap : words) + target - 1) / target);
            if (slot_stride != 0) {  // one triangle per slot: a slot's words of a stream are one contiguous piece
                FSK_LAUNCH(fsk::k_sx_consume, dim3(O, e->sx_rounds, nb), dim3(fsk::CS_THREADS), lds, e->stream, (const uint32_t*)e->d_ulist.p,
                           (const uint32_t*)e->d_list_off.p, (const uint32_t*)e->d_owner_r0.p, (const uint32_t*)nullptr, O, target,
                           e->sx_cap, e->sx_pb, K, (const uint32_t*)e->d_ucount.p, tpg, slot_stride, (const u64*)e->d_sxstat.p, cap_words);
            } else {
                FSK_LAUNCH(fsk::k_sx_parts, dim3(1), dim3(512), 0, e->stream, (const uint32_t*)e->d_list_off.p, O, target, e->d_part_base.p,
                           (const u64*)e->d_sxstat.p, cap_words);
                FSK_LAUNCH(fsk::k_sx_consume, dim3(max_parts, e->sx_rounds), dim3(fsk::CS_THREADS), lds, e->stream, (const uint32_t*)e->d_ulist.p,
                           (const uint32_t*)e->d_list_off.p, (const uint32_t*)e->d_owner_r0.p, (const uint32_t*)e->d_part_base.p, O, target,
                           e->sx_cap, e->sx_pb, K, (const uint32_t*)nullptr, tpg, (u64)0, (const u64*)e->d_sxstat.p, cap_words);
                e->st.launches += 1;
            }
            e->st.launches += 2;
        }
    } else {
        auto k_emit = skipping ? fsk::k_sx_emit<true, true> : fsk::k_sx_emit<true, false>;
        FSK_LAUNCH(k_emit, dim3(ntiles), dim3(fsk::EM_THREADS), 0, e->stream, (const uint2*)e->d_E.p, (const uint32_t*)e->d_Pk.p,
                   (const uint32_t*)e->d_ebase.p, (const uint32_t*)e->d_owner_r0.p, e->sx_own_shift, O, (const uint32_t*)nullptr,
                   (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, e->sx_pb, K, tpg,
                   slot_stride, skipping ? (const uint32_t*)e->d_Tk.p : (const uint32_t*)nullptr, (const u64*)nullptr, ~(u64)0);
        e->st.launches += 1;
    }
    e->toc(&e->st.ms_pairs);
    FSK_HIP(hipGetLastError());
    return FSK_OK;
}

int ensure_featseq(fsk_engine* e) {
    if (e->featseq_ready) return FSK_OK;
    std::vector<uint32_t> fs((size_t)e->nfeat);
    for (int64_t i = 0; i < e->N; ++i)
        for (uint32_t f = e->h_fstart[i]; f < e->h_fstart[i + 1]; ++f) fs[f] = (uint32_t)i;
    FSK_HIP(e->d_featseq.reserve((size_t)e->nfeat));
    FSK_HIP(hipMemcpy(e->d_featseq.p, fs.data(), fs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    e->featseq_ready = true;
    return FSK_OK;
}

int materialise_zero(fsk_engine* e);

constexpr int SX_DEFER = 8;          // variance mode: batches whose counts are read at their hand-over
constexpr int SX_DEFER_COMBOS = 16;  //                combos of such a batch at most

int sx_pinned(fsk_engine* e, size_t pos_bytes, size_t stat_words) {
    if (pos_bytes > e->h_sx_pos_cap) {
        if (e->h_sx_pos) (void)hipHostFree(e->h_sx_pos);
        e->h_sx_pos = nullptr; e->h_sx_pos_cap = 0;
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_pos, pos_bytes + pos_bytes / 2));
        e->h_sx_pos_cap = pos_bytes + pos_bytes / 2;
    }
    if (stat_words > e->h_sx_stat_cap) {
        if (e->h_sx_stat) (void)hipHostFree(e->h_sx_stat);
        e->h_sx_stat = nullptr; e->h_sx_stat_cap = 0;
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_stat, (stat_words + stat_words / 2) * sizeof(u64)));
        e->h_sx_stat_cap = stat_words + stat_words / 2;
    }
    return FSK_OK;
}

// how many words a batch may hold when it is enqueued before its count is known (0: size it exactly)
u64 sx_guard_for(fsk_engine* e) {
    if (e->sx_sync || e->cfg.profile) return 0;
    if (!e->sx_lists || e->force_global_pairs) return ~(u64)0;       // no streams: nothing to size
    if (e->sx_words_seen == 0) return 0;                                // (the first batch of these sequences)
    if (e->sx_guard_cap) return std::min<u64>(e->sx_guard_cap, (u64)e->d_ulist.cap);
    const u64 want = e->sx_words_seen + e->sx_words_seen / 2;
    if (want >= e->sx_max_words) return 0;
    if ((u64)e->d_ulist.cap < want && e->d_ulist.reserve((size_t)want) != hipSuccess) return 0;
    return std::min<u64>((u64)e->d_ulist.cap, e->sx_max_words - 1);
}

// the counts of deferred batch `slot` (its kernels have finished): false when it has to be redone
bool sx_harvest(fsk_engine* e, int slot) {
    if (slot < 0 || !e->sx_defer[slot].active) return true;
    e->sx_defer[slot].active = false;
    const u64 pairs = e->h_sx_stat[2 * slot], words = e->h_sx_stat[2 * slot + 1];
    e->sx_words_seen = std::max(e->sx_words_seen, words);
    if (words > e->sx_defer[slot].cap) { e->sx_redone += 1; return false; }
    e->u_extra += pairs;
    return true;
}

// slot_stride != 0 (variance mode): combo q of the list goes to its own u32 triangle (uint32_t*)K + q * slot_stride,
// written whole; returns FSK_RETRY_UNGROUPED when that form cannot be used for this batch.
// defer >= 0 (variance mode): the call returns with the batch enqueued; the caller passes `defer` to
// sx_harvest() once the batch has finished and redoes the batch (with e->sx_sync set) if that says so.
int accumulate_sparse(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1, u64 slot_stride = 0,
                      int defer = -1) {
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    int rc = ensure_featseq(e);
    if (rc) return rc;
    // batch so that the record count stays below the cap ...
    size_t per = SPARSE_MAX_RECORDS / (size_t)std::max<int64_t>(1, e->nfeat);
    int B = (int)std::max<size_t>(1, std::min<size_t>(per, (size_t)n));
    // ... and the owner bands can sum a batch in u32 LDS cells: per cell and combo <= maxW^2
    B = (int)std::max<u64>(1, std::min<u64>((u64)B, 0xffffffffull / std::max<u64>(1, (u64)e->maxW * e->maxW)));
    B = std::min(B, 65535);  // grid.y
    const int recbits = e->sx_keybits + e->sx_sb;  // (<= 62 + 31: a 128-bit record always holds it)
    const int nbatches = (n + B - 1) / B;
    if (defer >= 0 && (defer >= SX_DEFER || nbatches != 1 || n > SX_DEFER_COMBOS)) defer = -1;
    const size_t pos_head = (size_t)SX_DEFER * SX_DEFER_COMBOS * e->k, stat_head = (size_t)2 * SX_DEFER;
    rc = sx_pinned(e, pos_head + (size_t)n * e->k, stat_head + (size_t)2 * nbatches);
    if (rc) return rc;
    auto one = [&](int s, int nb, unsigned char* pos_pin, u64* stat_pin, u64 guard) {
        // (slot triangles are u32 arrays, slot_stride cells apart)
        u64* Kb = slot_stride ? reinterpret_cast<u64*>(reinterpret_cast<uint32_t*>(K) + (u64)s * slot_stride) : K;
        return recbits <= 32   ? sparse_batch<uint32_t>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard)
               : recbits <= 64 ? sparse_batch<u64>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard)
                               : sparse_batch<u128>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard);
    };
    if (defer >= 0) {
        const u64 guard = sx_guard_for(e);
        e->sx_defer[defer].active = guard != 0;
        e->sx_defer[defer].cap = guard;
        rc = one(0, n, e->h_sx_pos + (size_t)defer * SX_DEFER_COMBOS * e->k, e->h_sx_stat + 2 * defer, guard);
        if (rc) e->sx_defer[defer].active = false;
        return rc;
    }
    std::vector<u64> caps((size_t)nbatches, 0);  // per batch: the guard it was enqueued under (0: sized exactly)
    bool waiting = false;
    for (int s = 0, q = 0; s < n; s += B, ++q) {
        const int nb = std::min(B, n - s);
        caps[q] = sx_guard_for(e);
        waiting |= caps[q] != 0;
        rc = one(s, nb, e->h_sx_pos + pos_head + (size_t)s * e->k, e->h_sx_stat + stat_head + 2 * q, caps[q]);
        if (rc) return rc;
    }
    if (!waiting) return FSK_OK;
    FSK_HIP(hipStreamSynchronize(e->stream));
    for (int s = 0, q = 0; s < n; s += B, ++q) {
        if (!caps[q]) continue;
        const u64 pairs = e->h_sx_stat[stat_head + 2 * q], words = e->h_sx_stat[stat_head + 2 * q + 1];
        e->sx_words_seen = std::max(e->sx_words_seen, words);
        if (words <= caps[q]) { e->u_extra += pairs; continue; }
        // the batch did not fit and has left K alone: once more, sized exactly
        e->sx_redone += 1;
        const int was = e->sx_sync;
        e->sx_sync = 1;
        rc = one(s, std::min(B, n - s), e->h_sx_pos + pos_head + (size_t)s * e->k, e->h_sx_stat + stat_head + 2 * q, 0);
        e->sx_sync = was;
        if (rc) return rc;
    }
    return FSK_OK;
}

// XCD-aware tile order for the tile rows [t0, t1) of the lower-triangular tile grid: 8x8
// super-tiles are dealt to 8 queues (one per XCD, balanced by tile count); block b = 8q + x takes
// the q-th tile of queue x, because the dispatcher is observed to place blocks b, b+8, ... on one
// XCD (placement only changes speed, never results).
// `first_test_tile` (skip_test_block): tiles whose columns are all test sequences and that are not
// on the diagonal hold only test x test cells, which no getter of the reference exposes; they are
// left out (tile granularity: a tile that straddles the train/test boundary is kept).
void build_tile_table(uint32_t t0, uint32_t t1, uint32_t first_test_tile, std::vector<uint32_t>& tab) {
    constexpr uint32_t S = 8;
    std::vector<std::vector<uint32_t>> q(8);
    for (uint32_t si = t0 / S; si * S < t1; ++si)
        for (uint32_t sj = 0; sj <= si; ++sj) {
            size_t best = 0;
            for (size_t x = 1; x < 8; ++x)
                if (q[x].size() < q[best].size()) best = x;
            for (uint32_t ti = std::max(si * S, t0); ti < std::min((si + 1) * S, t1); ++ti)
                for (uint32_t tj = sj * S; tj < (sj + 1) * S && tj <= ti; ++tj)
                    if (tj < first_test_tile || tj == ti) q[best].push_back(ti << 16 | tj);
        }
    size_t total = 0, pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (auto& v : q) total += v.size();
    tab.clear();
    tab.reserve(total);
    while (tab.size() < total)
        for (size_t x = 0; x < 8 && tab.size() < total; ++x) {
            size_t src = x;
            if (pos[src] >= q[src].size()) {  // queue exhausted: steal from the longest remainder
                for (size_t y = 0; y < 8; ++y)
                    if (q[y].size() - pos[y] > q[src].size() - pos[src]) src = y;
            }
            tab.push_back(q[src][pos[src]++]);
        }
}

// rows that fsk_reset_counts left for a storing tile launch get their zeros now (anything but such
// a launch is about to look at K)
int materialise_zero(fsk_engine* e) {
    if (e->lazy_lo < 0) return FSK_OK;
    const u64 c0 = (u64)e->lazy_lo * ((u64)e->lazy_lo + 1) / 2, c1 = (u64)e->lazy_hi * ((u64)e->lazy_hi + 1) / 2;
    e->lazy_lo = e->lazy_hi = -1;
    if (c1 > c0) FSK_HIP(hipMemsetAsync(e->d_K + c0, 0, (size_t)(c1 - c0) * sizeof(u64), e->stream));
    return FSK_OK;
}
bool lazy_zero_possible(const fsk_engine* e) {
    return e->path == FSK_PATH_DENSE && e->tile_dma && !e->compact && !(e->cfg.skip_test_block && e->n_test > 0);
}

// the U of the first launch of a combo list arrives here (profile mode only)
int fetch_pending_u(fsk_engine* e) {
    if (!e->u_pending) return FSK_OK;
    FSK_HIP(hipStreamSynchronize(e->stream));
    FSK_HIP(hipMemcpy(&e->u_value, e->d_U2.p, sizeof(u64), hipMemcpyDeviceToHost));
    e->u_extra += e->u_value;
    e->u_pending = false;
    return FSK_OK;
}

int accumulate_dense(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1) {
    const uint32_t panels_pad = (e->n_panels + 1u) & ~1u;  // tiles are 2x2 panels
    const uint32_t t0 = (uint32_t)(row0 / fsk::TILE), t1 = (uint32_t)((row1 + fsk::TILE - 1) / fsk::TILE);
    if (t1 > 0xffffu) return e->fail(FSK_EUNSUPPORTED, "more than 65535 tile rows");
    if (t1 <= t0) return FSK_OK;
    // skip_test_block: first tile column made of test sequences only (none when not asked for)
    const uint32_t first_test_tile = e->cfg.skip_test_block && e->n_test > 0
                                         ? (uint32_t)((e->n_train + fsk::TILE - 1) / fsk::TILE) : 0xffffffffu;
    const uint32_t Vq8 = (e->Vq + 1u) / 2u;                               // dword rows: 8 keys (nibbles) each
    const uint32_t nst = (Vq8 + fsk::STAGE_KQ - 1) / fsk::STAGE_KQ;      // 32-row stages per combo
    const size_t slot_dwords4 = (size_t)panels_pad * Vq8 * fsk::PANEL;    // dwords of one plane per combo
    // combos per launch: u32 accumulators must not wrap (per cell and combo <= maxW^2), and the
    // count panels (lo + hi plane) must fit in the memory we are willing to take
    const u64 w2 = std::max<u64>(1, (u64)e->maxW * e->maxW);
    u64 by_overflow = 0xffffffffull / w2;
    if (by_overflow == 0) return e->fail(FSK_EUNSUPPORTED, "sequence too long for the dense path");
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t have = (e->d_C4.cap + e->d_C4H.cap) * sizeof(uint32_t);
    // panels for a few thousand combos per launch are plenty (one more launch costs one more
    // flush per tile); larger allocations only cost hipMalloc time
    size_t budget = std::max<size_t>(have, std::min<size_t>((size_t)((double)(free_b + have) * 0.6), (size_t)32 << 30));
    u64 by_memory = std::max<u64>(1, budget / (2 * slot_dwords4 * sizeof(uint32_t)));
    // (32768 combos per launch also keeps grid.y of the count and tile launches within limits)
    const int chunk = (int)std::max<u64>(1, std::min<u64>({(u64)n, by_overflow, by_memory, (u64)32768}));
    FSK_HIP(e->d_C4.reserve(slot_dwords4 * (size_t)chunk));
    FSK_HIP(e->d_C4H.reserve(slot_dwords4 * (size_t)chunk));
    FSK_HIP(e->d_rowmask.reserve((size_t)panels_pad * chunk * nst));
    FSK_HIP(e->d_flag.reserve(2));
    FSK_HIP(e->d_pos.reserve((size_t)chunk * e->k));
    if (e->tab_t0 != t0 || e->tab_t1 != t1 || e->tab_n == 0) {
        std::vector<uint32_t> tab;
        build_tile_table(t0, t1, first_test_tile, tab);
        if (first_test_tile == 0xffffffffu && tab.size() != (u64)t1 * (t1 + 1) / 2 - (u64)t0 * (t0 + 1) / 2)
            return e->fail(FSK_EDEVICE, "internal: tile table size mismatch");
        FSK_HIP(e->d_tiletab.reserve(tab.size()));
        FSK_HIP(hipMemcpyAsync(e->d_tiletab.p, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
        e->tab_t0 = t0; e->tab_t1 = t1; e->tab_n = (uint32_t)tab.size();
    }
    const u64 n_tiles = e->tab_n;
    const bool compact = e->compact;
    const uint32_t Vkeys = (uint32_t)e->V, Vw = (Vkeys + 31u) / 32u;
    DensePlan plan = dense_plan(e->maxW, e->cfg.g, e->Vq, compact ? (size_t)Vkeys * 2 : 0);  // may be re-planned below
    if (plan.CH == 0) return e->fail(FSK_EUNSUPPORTED, "dense path: LDS plan does not fit");
    uint32_t CH = plan.CH;
    if (e->force_chunk) CH = std::max(1u, std::min(CH, e->force_chunk));
    size_t lds = (size_t)(CH + e->cfg.g - 1) * fsk::PANEL + (size_t)plan.Vcq * 512 + (compact ? (size_t)Vkeys * 2 : 0);
    // several histogram sweeps over one staging pass: cache the window keys in LDS (u16 each) when
    // they fit next to everything else, so that only the first sweep computes them
    uint32_t kc_rows = 0;
    if (plan.Vcq < e->Vq && CH >= e->maxW && !e->force_chunk) {
        // re-plan with the cache carved out first
        const size_t cache = (size_t)e->maxW * fsk::PANEL * sizeof(uint16_t);
        DensePlan p2 = dense_plan(e->maxW, e->cfg.g, e->Vq, (compact ? (size_t)Vkeys * 2 : 0) + cache);
        if (p2.CH >= e->maxW && p2.Vcq >= 64) {
            plan = p2;
            CH = plan.CH;
            kc_rows = e->maxW;
            lds = plan.lds;
        }
    }
#ifndef FSK_EMU
    {
        auto k0 = fsk::k_dense_count<false, false>;
        auto k1 = fsk::k_dense_count<false, true>;
        auto k2 = fsk::k_dense_count<true, false>;
        FSK_HIP(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FSK_HIP(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FSK_HIP(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
#endif
    if (compact) {
        FSK_HIP(e->d_keybits.reserve((size_t)chunk * Vw));
        FSK_HIP(e->d_lut.reserve((size_t)chunk * Vkeys));
        FSK_HIP(e->d_vc.reserve((size_t)chunk));
    }
    std::vector<uint16_t> h_vc;
    std::vector<uint8_t> pos;
    for (int s = 0; s < n; s += chunk) {
        const int nb = std::min(chunk, n - s);
        // the count panels of an unchanged single-chunk combo list are reused by the FOLLOWING row
        // bands of one pass (row0 > 0); a call that starts at row 0 always recounts
        const bool cached = row0 > 0 && e->prep_valid && nb == n && (int)e->prep_combos.size() == n &&
                            std::equal(combos, combos + n, e->prep_combos.begin());
        if (!cached) {
            e->prep_valid = false;
            pos.resize((size_t)nb * e->k);
            for (int q = 0; q < nb; ++q)
                memcpy(&pos[(size_t)q * e->k], &e->all_pos[(size_t)combos[s + q] * e->k], e->k);
            FSK_HIP(hipMemcpyAsync(e->d_pos.p, pos.data(), pos.size(), hipMemcpyHostToDevice, e->stream));
            FSK_HIP(hipMemsetAsync(e->d_flag.p, 0, sizeof(uint32_t), e->stream));
            FSK_HIP(hipStreamSynchronize(e->stream));  // `pos` is a pageable temporary
            // ---- segment counts
            // up to 16 combos share one staging of a panel's symbols, fewer when that would leave the
            // launch with less than ~1024 workgroups (few sequences)
            const int slots_per_chunk = std::max(1, std::min({nb, 16, (int)((u64)nb * panels_pad / 1024)}));
            const int n_chunks = (nb + slots_per_chunk - 1) / slots_per_chunk;
            e->tic();
            const dim3 cgrid(panels_pad, n_chunks);
            // (function pointers: a template-id with a comma cannot pass through the launch macro)
            auto k_mark = fsk::k_dense_count<true, false>;
            auto k_count_lut = fsk::k_dense_count<false, true>;
            auto k_count = fsk::k_dense_count<false, false>;
            if (compact) {  // which keys occur per combo -> rank tables -> compacted panels
                FSK_HIP(hipMemsetAsync(e->d_keybits.p, 0, (size_t)nb * Vw * sizeof(uint32_t), e->stream));
                FSK_LAUNCH(k_mark, cgrid, dim3(256), lds, e->stream, e->view(), e->cfg.g,
                           e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, e->d_pos.p, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
                           e->d_rowmask.p, nst, e->d_flag.p, Vkeys, (const uint16_t*)nullptr, (const uint16_t*)nullptr, e->d_keybits.p, kc_rows);
                FSK_LAUNCH(fsk::k_dense_keylut, dim3(nb), dim3(256), 0, e->stream, e->d_keybits.p, Vkeys, e->d_lut.p, e->d_vc.p);
                FSK_LAUNCH(k_count_lut, cgrid, dim3(256), lds, e->stream, e->view(), e->cfg.g,
                           e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, e->d_pos.p, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
                           e->d_rowmask.p, nst, e->d_flag.p, Vkeys, e->d_lut.p, e->d_vc.p, (uint32_t*)nullptr, kc_rows);
                h_vc.resize((size_t)nb);
                FSK_HIP(hipMemcpyAsync(h_vc.data(), e->d_vc.p, (size_t)nb * sizeof(uint16_t), hipMemcpyDeviceToHost, e->stream));
                FSK_HIP(hipStreamSynchronize(e->stream));
                e->st.launches += 2;
                e->h_vc_cache = h_vc;
                {   // running mean of the compacted key counts (stats)
                    double sum = 0;
                    for (uint16_t v : h_vc) sum += v;
                    e->vc_sum += sum; e->vc_n += (double)nb;
                    e->st.compact_keys_avg = e->vc_sum / e->vc_n;
                }
            } else {
                FSK_LAUNCH(k_count, cgrid, dim3(256), lds, e->stream, e->view(), e->cfg.g,
                           e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, e->d_pos.p, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
                           e->d_rowmask.p, nst, e->d_flag.p, Vkeys, (const uint16_t*)nullptr, (const uint16_t*)nullptr, (uint32_t*)nullptr, kc_rows);
            }
            e->toc(&e->st.ms_count);
            e->st.count_launches += 1;
            e->st.launches += 1;
            e->st.panel_bytes += 2 * slot_dwords4 * sizeof(uint32_t) * (u64)nb;
            // a count above 255 does not fit the u8 panels either: take the general dataflow
            // for this batch (only possible when a sequence has more than 255 windows)
            e->prep_overflow = false;
            if (e->maxW > 255) {
                uint32_t flag = 0;
                FSK_HIP(hipMemcpyAsync(&flag, e->d_flag.p, sizeof flag, hipMemcpyDeviceToHost, e->stream));
                FSK_HIP(hipStreamSynchronize(e->stream));
                e->prep_overflow = (flag & 1u) != 0;
            }
            if (e->cfg.profile && !e->prep_overflow) {  // exact algorithmic update count U
                const bool same = nb == n && e->u_known && (int)e->u_combos.size() == n && std::equal(combos, combos + n, e->u_combos.begin());
                if (e->u_pending) {  // value of the previous first-time launch
                    int rc = fetch_pending_u(e);
                    if (rc) return rc;
                }
                if (same) {
                    e->u_extra += e->u_value;  // same sequences, same combos: same U
                } else if (nb == n) {
                    FSK_HIP(e->d_U2.reserve(1));
                    FSK_HIP(hipMemsetAsync(e->d_U2.p, 0, sizeof(u64), e->stream));
                    FSK_LAUNCH(fsk::k_dense_distinct, dim3(Vq8, nb), dim3(64), 0, e->stream, e->d_C4.p, e->d_C4H.p, panels_pad, nb, Vq8,
                               e->d_U2.p, compact ? (const uint16_t*)e->d_vc.p : (const uint16_t*)nullptr);
                    e->u_combos.assign(combos, combos + n);
                    e->u_known = true;
                    e->u_pending = true;
                } else {
                    FSK_LAUNCH(fsk::k_dense_distinct, dim3(Vq8, nb), dim3(64), 0, e->stream, e->d_C4.p, e->d_C4H.p, panels_pad, nb, Vq8,
                               e->d_U.p, compact ? (const uint16_t*)e->d_vc.p : (const uint16_t*)nullptr);
                }
            }
            if (nb == n) {
                e->prep_combos.assign(combos, combos + n);
                e->prep_valid = true;
            }
        }
        if (e->prep_overflow) {
            int rc = accumulate_sparse(e, combos + s, nb, K, row0, row1);
            if (rc) return rc;
            continue;
        }
        // ---- tiled accumulate. With few tiles (small N) the combo range is split over several
        // workgroups per tile (each flushes its partial sums with atomics, which drain under other
        // workgroups' dot products).
        int n_splits = 1;
        if (n_tiles < 16384 && nb >= 2) {
            // Measured (tools/sweep_splits.py, both tile kernels, N = 256 .. 22000): the launch is
            // fastest when a workgroup multiplies about 600 dword rows (20-odd combos of 256 keys)
            // — short enough that the 1024 (compact: 768) resident slots turn over many times and
            // the tail is short, long enough that prologue and flush stay small — with no more
            // than ~16k workgroups in all and never fewer than slots when the combos allow it.
            // The curve is flat around the optimum (+-2 %); one split costs 10-30 %.
            double rows_per_combo = (double)Vq8;
            if (compact && (int)e->h_vc_cache.size() == nb) {
                double sum = 0;
                for (uint16_t v : e->h_vc_cache) sum += (v + 7u) / 8u;
                rows_per_combo = std::max(1.0, sum / nb);
            }
            const int slots = compact ? 768 : 1024;
            int per = std::max(2, (int)std::ceil(600.0 / rows_per_combo));
            n_splits = std::max(1, (nb + per - 1) / per);
            // ... and about 16k workgroups are enough: beyond that more splits only add flushes
            n_splits = std::min(n_splits, (int)((16384 + n_tiles - 1) / n_tiles));
            if ((double)n_tiles * n_splits < slots)
                n_splits = std::max(n_splits, std::min(nb / 2, (int)((slots + n_tiles - 1) / n_tiles)));
            n_splits = std::max(1, std::min({n_splits, nb, 4096}));
        }
        if (e->force_splits > 0) n_splits = std::min({nb, e->force_splits, 4096});
        const int slots_per_split = (nb + n_splits - 1) / n_splits;
        n_splits = (nb + slots_per_split - 1) / slots_per_split;
        // Store instead of add? Only the first launch over rows that are still "zero by contract",
        // starting at their lower edge, with one workgroup per tile and the engine's own triangle.
        int store = 0;
        if (e->store_next) {
            if (!(e->tile_dma && !compact && n_splits == 1 && first_test_tile == 0xffffffffu && row0 == 0 && row1 >= e->N))
                return e->fail(FSK_ESTATE, "internal: a storing tile launch was asked for where none is possible");
            store = 1;
        } else if (e->lazy_lo >= 0) {
            if (K == e->d_K && e->tile_dma && !compact && n_splits == 1 && first_test_tile == 0xffffffffu && row0 == e->lazy_lo &&
                row1 <= e->lazy_hi) {
                store = 1;
                e->lazy_lo = row1 < e->lazy_hi ? row1 : -1;
                if (e->lazy_lo < 0) e->lazy_hi = -1;
            } else {
                int rcz = materialise_zero(e);
                if (rcz) return rcz;
            }
        }
        e->tic();
        if (compact && e->compact_dma)
            FSK_LAUNCH(fsk::k_dense_tile_dma_compact, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p, e->d_C4H.p,
                       e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split, 0, (const uint16_t*)e->d_vc.p);
        else if (compact)
            FSK_LAUNCH(fsk::k_dense_tile_compact, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p,
                       e->d_C4H.p, e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split,
                       (const uint16_t*)e->d_vc.p);
        else if (e->tile_dma)
            FSK_LAUNCH(fsk::k_dense_tile_dma, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p, e->d_C4H.p,
                       e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split, store);
        else
            FSK_LAUNCH(fsk::k_dense_tile, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p, e->d_C4H.p,
                       e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split);
        e->toc(&e->st.ms_tile);
        e->st.n_tile_launches += 1;
        u64 row_sum = (u64)Vq8 * (u64)nb;  // dword rows multiplied per tile (flagged-row remainders not counted)
        if (compact && (int)e->h_vc_cache.size() == nb) {
            row_sum = 0;
            for (uint16_t v : e->h_vc_cache) row_sum += (v + 7u) / 8u;
        }
        e->st.dense_macs += n_tiles * (u64)fsk::TILE * fsk::TILE * row_sum * 8;
        e->st.u4_tile_launches += 1;
        e->st.launches += 1;
        FSK_HIP(hipGetLastError());
    }
    return FSK_OK;
}

int do_accumulate(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0 = 0, int64_t row1 = -1, u64 slot_stride = 0,
                  int defer = -1) {
    if (row1 < 0) row1 = e->N;
    for (int i = 0; i < n; ++i)
        if (combos[i] < 0 || combos[i] >= e->ncomb) return e->fail(FSK_EINVAL, "combo id %d out of range [0,%lld)", combos[i], (long long)e->ncomb);
    hipEvent_t a = nullptr, b = nullptr;
    if (e->cfg.profile) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, e->stream);
    }
    int rc = e->path == FSK_PATH_DENSE ? accumulate_dense(e, combos, n, K, row0, row1) : accumulate_sparse(e, combos, n, K, row0, row1, slot_stride, defer);
    if (e->cfg.profile) {
        (void)hipEventRecord(b, e->stream);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        e->st.ms_total += ms;
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    if (rc == FSK_OK && row1 >= e->N) e->st.combos_done += n;  // a combo is done when its last row band is
    return rc;
}

int make_diag(fsk_engine* e) {
    FSK_HIP(e->d_diag.reserve((size_t)e->N));
    const uint32_t blocks = (uint32_t)((e->N + 255) / 256);
    if (e->result_f64)
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_diag<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, e->d_diag.p, (uint32_t)e->N);
    else
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_diag<u64>), dim3(blocks), dim3(256), 0, e->stream, e->d_K, e->d_diag.p, (uint32_t)e->N);
    FSK_HIP(hipStreamSynchronize(e->stream));
    e->finalized = true;
    return FSK_OK;
}

void default_order(fsk_engine* e) {
    e->order.resize((size_t)e->ncomb);
    for (int64_t i = 0; i < e->ncomb; ++i) e->order[i] = (int32_t)i;
    uint64_t s = e->seed;
    for (int64_t i = e->ncomb - 1; i > 0; --i) {
        int64_t j = (int64_t)(splitmix64(s) % (uint64_t)(i + 1));
        std::swap(e->order[i], e->order[j]);
    }
}

// the reference's `avg` of get_variance (fastsk_kernel.cpp:116-131): sum of n doubles in index order,
// on the device (k_seq_prep on all CUs + k_seq_chain); bsum = approximate sums per SQ_BLOCK values
// (zero on entry, zero again on exit), result to out[0]
// `count` independent sums laid out `stride` values apart (their block sums / block records / results
// follow each other); the chains run on `chain_stream`
// grp: 2 * SQ_GROUPS group records per block, laid out like blk
int enqueue_sequential_sum(fsk_engine* e, const double* d_vals, u64 n, double* bsum, fsk::SeqBlk* blk, fsk::SeqGrp* grp, double* out,
                           int count = 1, u64 stride = 0, hipStream_t chain_stream = nullptr, hipEvent_t handoff = nullptr) {
    const uint32_t nblocks = (uint32_t)((n + fsk::SQ_BLOCK - 1) / fsk::SQ_BLOCK);
    if (nblocks > 0)
        FSK_LAUNCH(fsk::k_seq_prep, dim3(nblocks, count), dim3(256), 0, e->stream, d_vals, n, (const double*)bsum, blk, stride, nblocks);
    hipStream_t cs = chain_stream ? chain_stream : e->stream;
    if (chain_stream) {
        FSK_HIP(hipEventRecord(handoff, e->stream));
        FSK_HIP(hipStreamWaitEvent(chain_stream, handoff, 0));
    }
    if (nblocks > 0)  // (the few blocks that need group records: on the chains' stream, beside the next batch's kernels)
        FSK_LAUNCH(fsk::k_seq_prep_groups, dim3(nblocks, count), dim3(256), 0, cs, d_vals, n, (const fsk::SeqBlk*)blk, stride, nblocks, grp);
    FSK_LAUNCH(fsk::k_seq_chain, dim3(count), dim3(64), 0, cs, d_vals, n, (const fsk::SeqBlk*)blk, nblocks, bsum, out, stride,
               (const fsk::SeqGrp*)grp);
    return FSK_OK;
}

// variance mode: T sequential Welford chains (fastsk_kernel.cpp:188-262, 286-315)
// chains tid = chain_first, chain_first + chain_step, ... < T (all of them: 0, 1); stdevs are chain 0's
int run_variance_mode(fsk_engine* e, int T, int chain_first = 0, int chain_step = 1) {
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const int64_t pairs = e->pairs;
    const int64_t train_pairs = (int64_t)((e->n_train / (double)2) * (e->n_train + 1));
    const size_t tp = (size_t)std::max<int64_t>(1, train_pairs);
    // The stop test of iteration i needs avg_variance, a SEQUENTIAL fp64 sum in triangle-index
    // order (fastsk_kernel.cpp:116-131), to the last bit. It is computed on the device
    // (enqueue_sequential_sum), so only 8 bytes per iteration come back; the engine still runs AHEAD
    // of its stop test: iterations are issued in batches of AHEAD with up to DEPTH batches in flight,
    // the Welford state of every untested iteration is kept in a ring, and whatever lies beyond the
    // stopping iteration is dropped.
    // (two batches in flight: the sums of one run on the second stream under the kernels of the next;
    // a third would only add iterations that are thrown away when the stop test fires)
    constexpr int AHEAD = 4, MAX_DEPTH = 2;
    const int DEPTH = MAX_DEPTH;
    const int RING = DEPTH * AHEAD + 1;
    const bool trace = getenv("FSK_TRACE") != nullptr;  // stderr: where the wall time of this mode goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    double t_wait = 0;
    const auto t_begin = now();
    const size_t nblk = (tp + fsk::SQ_BLOCK - 1) / fsk::SQ_BLOCK;
    const size_t slots = (size_t)DEPTH * AHEAD;
    FSK_HIP(e->d_Kf64.reserve((size_t)pairs));
    FSK_HIP(e->d_Khat.reserve((size_t)pairs * RING));
    FSK_HIP(e->d_prod.reserve(tp * slots));
    FSK_HIP(e->d_bsum.reserve(nblk * slots + slots));
    FSK_HIP(e->d_seqblk.reserve(nblk * slots * (sizeof(fsk::SeqBlk) + 2 * fsk::SQ_GROUPS * sizeof(fsk::SeqGrp))));
    fsk::SeqGrp* const seq_grp = reinterpret_cast<fsk::SeqGrp*>(e->d_seqblk.p + nblk * slots * sizeof(fsk::SeqBlk));
    FSK_HIP(hipMemsetAsync(e->d_Kf64.p, 0, (size_t)pairs * sizeof(double), e->stream));
    FSK_HIP(hipMemsetAsync(e->d_bsum.p, 0, (nblk * slots + slots) * sizeof(double), e->stream));
    if (e->h_prod_cap < slots) {
        if (e->h_prod) (void)hipHostFree(e->h_prod);
        e->h_prod = nullptr; e->h_prod_cap = 0;
        FSK_HIP(hipHostMalloc((void**)&e->h_prod, slots * sizeof(double)));
        e->h_prod_cap = slots;
    }
    double* h_avg = e->h_prod;  // pinned
    if (!e->chain_stream) FSK_HIP(hipStreamCreateWithFlags(&e->chain_stream, hipStreamNonBlocking));
    hipEvent_t ev_done[MAX_DEPTH], ev_hand[MAX_DEPTH];
    for (auto& ev : ev_done) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto& ev : ev_hand) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct Cleanup {
        hipEvent_t* a; hipEvent_t* b; fsk_engine* e;
        ~Cleanup() {
            (void)hipStreamSynchronize(e->stream);  // nothing of this call may still be in flight
            (void)hipStreamSynchronize(e->chain_stream);
            for (int i = 0; i < MAX_DEPTH; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); }
        }
    } cleanup{ev_done, ev_hand, e};
    const double t_alloc = ms_since(t_begin);
    const uint32_t blocks = (uint32_t)((pairs + 255) / 256);                                       // one cell per thread
    const uint32_t wblocks = (uint32_t)((pairs + 256 * fsk::WF_ITEMS - 1) / (256 * fsk::WF_ITEMS)); // k_welford
    const int n_order = (int)e->order.size();
    auto khat = [&](int i) { return e->d_Khat.p + (size_t)(i % RING) * (size_t)pairs; };
    struct Batch { int n = 0, first_iter = 0, first_item = 0, base = 0, part = 0; bool grouped = false; };
    // how many iterations can still follow (end of the work list, max_iters)
    auto plan = [&](int first_iter, int first_item) {
        int n = AHEAD;
        n = std::min(n, first_item < n_order ? (n_order - first_item + T - 1) / T : 0);
        if (e->cfg.max_iters != -1) n = std::min(n, e->cfg.max_iters - first_iter + 1);
        return std::max(n, 0);
    };
    // the batch that would follow B if all of B is accepted
    auto after = [&](const Batch& B) {
        Batch N;
        N.first_iter = B.first_iter + B.n; N.first_item = B.first_item + B.n * T; N.base = B.base + B.n;
        N.part = (B.part + 1) % DEPTH;
        N.n = plan(N.first_iter, N.first_item);
        return N;
    };
    // sparse dataflow: the iterations of a batch are sorted and segmented together (one slot each) and
    // land in AHEAD separate triangles; dense dataflow: one iteration at a time into the engine's triangle
    // (u32 triangles written whole by k_sx_consume; without update streams — huge N — pairs go to K with
    // atomics and the iterations run one at a time like the dense ones)
    bool grouped = e->path == FSK_PATH_SPARSE && e->sx_lists && !e->force_global_pairs;
    // Dense dataflow with a tile kernel that can STORE (one workgroup per tile, the direct-to-LDS kernel, no
    // key compaction, no test-block filter): every iteration's tile launch stores its counts into a u64
    // triangle of its own — no zero fill — and the batch's Welford updates run as one pass like the sparse
    // batches' (while the slot triangles stay a modest share of the memory).
    const bool dense_slots = e->path == FSK_PATH_DENSE && e->tile_dma && !e->compact && !(e->cfg.skip_test_block && e->n_test > 0) &&
                             (u64)pairs * AHEAD * DEPTH * sizeof(u64) <= ((u64)8 << 30) && e->variance_dense_slots;
    // (one set of slot triangles per batch in flight: a stop inside a batch runs its Welford prefix again)
    if (grouped) FSK_HIP(e->d_Kslots.reserve(((size_t)pairs * AHEAD * DEPTH + 1) / 2));
    if (dense_slots) FSK_HIP(e->d_Kslots.reserve((size_t)pairs * AHEAD * DEPTH));
    auto slots_of = [&](int part) { return reinterpret_cast<uint32_t*>(e->d_Kslots.p) + (size_t)part * AHEAD * (size_t)pairs; };
    auto slots64_of = [&](int part) { return e->d_Kslots.p + (size_t)part * AHEAD * (size_t)pairs; };
    static_assert(AHEAD <= fsk::WF_SLOTS, "k_welford_batch carries a batch's iterations in registers");
    auto issue = [&](Batch& B) -> int {
        if (grouped) {
            int32_t combos[AHEAD];
            for (int b = 0; b < B.n; ++b) combos[b] = e->order[B.first_item + b * T];
            int rc = do_accumulate(e, combos, B.n, reinterpret_cast<u64*>(slots_of(B.part)), 0, -1, (u64)pairs, B.part);
            if (rc == FSK_RETRY_UNGROUPED) grouped = false;  // too many updates for one stream: from here on one iteration at a time
            else if (rc) return rc;
        }
        B.grouped = grouped || dense_slots;  // (the batch's counts sit in slot triangles, its Welford update is one pass)
        if (grouped) {  // K_hat through the batch's iterations in one pass; only the state after the batch is written
            const size_t slot0 = (size_t)B.part * AHEAD;
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch<uint32_t>), dim3(wblocks), dim3(256), 0, e->stream, (const uint32_t*)slots_of(B.part), B.n,
                       (const double*)khat(B.base), khat(B.base + B.n), e->d_prod.p + slot0 * tp, (u64)tp, (u64)pairs, (u64)train_pairs,
                       (double)B.first_iter, e->d_bsum.p + slot0 * nblk, (uint32_t)nblk, 1);
        } else if (dense_slots) {
            for (int b = 0; b < B.n; ++b) {
                int32_t combo = e->order[B.first_item + b * T];
                e->store_next = true;
                int rc = do_accumulate(e, &combo, 1, slots64_of(B.part) + (size_t)b * pairs);
                e->store_next = false;
                if (rc) return rc;
            }
            const size_t slot0 = (size_t)B.part * AHEAD;
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch<u64>), dim3(wblocks), dim3(256), 0, e->stream, (const u64*)slots64_of(B.part), B.n,
                       (const double*)khat(B.base), khat(B.base + B.n), e->d_prod.p + slot0 * tp, (u64)tp, (u64)pairs, (u64)train_pairs,
                       (double)B.first_iter, e->d_bsum.p + slot0 * nblk, (uint32_t)nblk, 1);
        }
        for (int b = 0; b < B.n && !B.grouped; ++b) {
            const size_t slot = (size_t)(B.part * AHEAD + b);
            FSK_HIP(hipMemsetAsync(e->d_K, 0, (size_t)pairs * sizeof(u64), e->stream));
            int32_t combo = e->order[B.first_item + b * T];
            int rc = do_accumulate(e, &combo, 1, e->d_K);
            if (rc) return rc;
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford<u64>), dim3(wblocks), dim3(256), 0, e->stream, (const u64*)e->d_K, (const double*)khat(B.base + b),
                       khat(B.base + b + 1), e->d_prod.p + slot * tp, (u64)pairs, (u64)train_pairs, (double)(B.first_iter + b),
                       e->d_bsum.p + slot * nblk);
        }
        // the batch's sums: block totals on all CUs, then one wave per iteration walks its blocks — on a
        // second stream, under the kernels of the batches that follow
        const size_t slot0 = (size_t)B.part * AHEAD;
        int rc = enqueue_sequential_sum(e, e->d_prod.p + slot0 * tp, (u64)train_pairs, e->d_bsum.p + slot0 * nblk,
                                        reinterpret_cast<fsk::SeqBlk*>(e->d_seqblk.p) + slot0 * nblk,
                                        seq_grp + slot0 * nblk * (2 * fsk::SQ_GROUPS), h_avg + slot0, B.n, (u64)tp,
                                        e->chain_stream, ev_hand[B.part]);  // (the sums land in pinned host memory)
        if (rc) return rc;
        FSK_HIP(hipEventRecord(ev_done[B.part], e->chain_stream));
        return FSK_OK;
    };
    e->stdevs.clear();
    for (int tid = chain_first; tid < T; tid += chain_step) {
        int cur = 0;  // ring position of the state after the last accepted iteration
        FSK_HIP(hipMemsetAsync(khat(cur), 0, (size_t)pairs * sizeof(double), e->stream));
        int iter = 1, item = tid;
        std::vector<Batch> q;  // issued, untested batches, oldest first (at most DEPTH)
        {
            Batch A;
            A.first_iter = iter; A.first_item = item; A.base = cur; A.part = 0;
            A.n = std::max(1, plan(iter, item));  // (the reference always runs the first iteration)
            int rc = issue(A);
            if (rc) return rc;
            q.push_back(A);
        }
        bool working = true;
        while (working) {
            while ((int)q.size() < DEPTH) {  // keep the device DEPTH batches ahead of the stop test
                Batch N = after(q.back());
                if (N.n == 0) break;
                int rc = issue(N);
                if (rc) return rc;
                q.push_back(N);
            }
            const Batch A = q.front();
            auto t0 = now();
            FSK_HIP(hipEventSynchronize(ev_done[A.part]));
            t_wait += ms_since(t0);
            if (!sx_harvest(e, A.part)) {
                // A was enqueued ahead of its word count and did not fit the update streams: its slot
                // triangles were not written. Everything issued after it started from A's state: drop
                // it all and run A again, sized exactly.
                FSK_HIP(hipStreamSynchronize(e->stream));
                FSK_HIP(hipStreamSynchronize(e->chain_stream));
                for (const Batch& B : q) { e->st.combos_done -= B.n; e->sx_defer[B.part].active = false; }
                q.clear();
                const int was = e->sx_sync;
                e->sx_sync = 1;
                Batch R = A;
                int rc = issue(R);
                e->sx_sync = was;
                if (rc) return rc;
                q.push_back(R);
                continue;
            }
            int accepted = 0;
            for (int b = 0; b < A.n && working; ++b) {
                double v = h_avg[(size_t)A.part * AHEAD + b] / (double)train_pairs;
                if (iter == 1) v = 9999999;
                else v /= iter - 1;
                const double sd = std::sqrt(v / iter);
                if (tid == 0) e->stdevs.push_back(sd);
                if (e->cfg.delta / sd > 1.96) working = false;
                if (e->cfg.max_iters != -1 && iter >= e->cfg.max_iters) working = false;
                item += T;
                if (item >= n_order) working = false;
                iter++;
                accepted = b + 1;
            }
            cur = A.base + accepted;
            e->st.combos_done -= A.n - accepted;  // iterations run ahead of the stop are dropped
            q.erase(q.begin());
            if (!working) {
                for (const Batch& B : q) e->st.combos_done -= B.n;
                if (!q.empty()) {  // dropped batches drain before their buffers are reused
                    FSK_HIP(hipStreamSynchronize(e->stream));
                    FSK_HIP(hipStreamSynchronize(e->chain_stream));
                }
                for (const Batch& B : q) (void)sx_harvest(e, B.part);
                if (A.grouped && accepted < A.n) {  // the stop fell inside the batch: the state after its accepted prefix
                    if (dense_slots)
                        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch<u64>), dim3(wblocks), dim3(256), 0, e->stream, (const u64*)slots64_of(A.part),
                                   accepted, (const double*)khat(A.base), khat(A.base + accepted), (double*)nullptr, (u64)0, (u64)pairs,
                                   (u64)train_pairs, (double)A.first_iter, (double*)nullptr, (uint32_t)0, 0);
                    else
                        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch<uint32_t>), dim3(wblocks), dim3(256), 0, e->stream,
                                   (const uint32_t*)slots_of(A.part), accepted, (const double*)khat(A.base), khat(A.base + accepted),
                                   (double*)nullptr, (u64)0, (u64)pairs, (u64)train_pairs, (double)A.first_iter, (double*)nullptr, (uint32_t)0, 0);
                }
                break;
            }
            // (working implies more items and iterations: the queue is not empty)
        }
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_add_nonzero<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, (const double*)khat(cur), (u64)pairs);
    }
    e->result_f64 = true;
    FSK_HIP(hipStreamSynchronize(e->stream));
    if (trace)
        fprintf(stderr, "[fsk] variance mode: setup %.2f ms, waiting for the GPU %.2f ms, total %.2f ms (%lld cells/iteration, %d batches in flight)\n",
                t_alloc, t_wait, ms_since(t_begin), (long long)train_pairs, DEPTH);
    return FSK_OK;
}

int fetch_block(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double* out) {
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel: call fsk_compute or fsk_finalize first");
    if (i0 < 0 || j0 < 0 || i1 > e->N || j1 > e->N || i0 > i1 || j0 > j1) return e->fail(FSK_EINVAL, "block out of range");
    const int64_t rows = i1 - i0, cols = j1 - j0;
    if (rows == 0 || cols == 0) return FSK_OK;
    if (!out) return e->fail(FSK_EINVAL, "null output");
    const int64_t max_cells = (int64_t)32 << 20;  // 256 MB of doubles per staging round
    const int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>(rows, max_cells / cols));
    FSK_HIP(e->d_stage.reserve((size_t)(rows_per * cols)));
    for (int64_t r = 0; r < rows; r += rows_per) {
        const int64_t nr = std::min(rows_per, rows - r);
        const u64 cells = (u64)nr * cols;
        const uint32_t blocks = (uint32_t)((cells + 255) / 256);
        if (e->result_f64)
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_block<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, e->d_diag.p,
                       (u64)(i0 + r), (u64)nr, (u64)j0, (u64)cols, e->d_stage.p);
        else
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_block<u64>), dim3(blocks), dim3(256), 0, e->stream, e->d_K, e->d_diag.p,
                       (u64)(i0 + r), (u64)nr, (u64)j0, (u64)cols, e->d_stage.p);
        FSK_HIP(hipMemcpyAsync(out + r * cols, e->d_stage.p, cells * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    return FSK_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int fsk_abi_version(void) { return FSK_ABI_VERSION; }

int fsk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int64_t fsk_num_combos(int32_t g, int32_t m) { return n_choose_k(g, m); }

int fsk_combo_positions(int32_t g, int32_t k, int64_t combo, int32_t* out) {
    if (!out || k <= 0 || k > g || combo < 0 || combo >= n_choose_k(g, k)) return FSK_EINVAL;
    int next = 0;
    for (int d = 0; d < k; ++d)
        for (int p = next; p < g; ++p) {
            int64_t below = n_choose_k(g - p - 1, k - d - 1);
            if (combo < below) { out[d] = p; next = p + 1; break; }
            combo -= below;
        }
    return FSK_OK;
}

const char* fsk_last_error(const fsk_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int fsk_create(const fsk_config* cfg, fsk_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return FSK_EINVAL; }
    *out = nullptr;
    if (cfg->g <= 0 || cfg->m < 0 || cfg->m >= cfg->g) {
        g_create_error = "need 0 <= m < g";
        return FSK_EINVAL;
    }
    if (cfg->g > 255) { g_create_error = "g > 255 unsupported"; return FSK_EUNSUPPORTED; }
    if (cfg->t == 0 || cfg->t < -1) { g_create_error = "t must be -1 or >= 1"; return FSK_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device visible: the MI355X engine has no CPU fallback";
        return FSK_EDEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return FSK_EINVAL; }
    DeviceScope on_device(cfg->device);  // the caller's current device is restored on return
    if (on_device.err != hipSuccess) { g_create_error = "hipSetDevice failed"; return FSK_EDEVICE; }
    fsk_engine* e = new fsk_engine;
    e->cfg = *cfg;
    e->k = cfg->g - cfg->m;
    e->ncomb = n_choose_k(cfg->g, cfg->m);
    if (e->ncomb > 0x7fffffff) { delete e; g_create_error = "C(g,m) >= 2^31 unsupported"; return FSK_EUNSUPPORTED; }
    enumerate_combos(cfg->g, e->k, e->all_pos);
    { const char* f = getenv("FSK_COMPACT"); e->force_compact = f ? atoi(f) : -1; }
    { const char* f = getenv("FSK_SPARSE_GLOBAL"); e->force_global_pairs = f ? atoi(f) : 0; }
    { const char* f = getenv("FSK_VARIANCE_DENSE_SLOTS"); if (f) e->variance_dense_slots = atoi(f); }
    { const char* f = getenv("FSK_SEG_SCAN_CHUNKED"); if (f) e->force_seg_chunks = atoi(f); }
    { const char* f = getenv("FSK_SPARSE_SYNC"); if (f) e->sx_sync = atoi(f); }
    { const char* f = getenv("FSK_SPARSE_GUARD_CAP"); if (f && atoll(f) > 0) e->sx_guard_cap = (u64)atoll(f); }
    { const char* f = getenv("FSK_LIST_MAX_WORDS"); if (f && atoll(f) > 0) e->sx_max_words = std::min<u64>(SX_MAX_LIST_WORDS, (u64)atoll(f)); }
    { const char* f = getenv("FSK_TILE_SPLITS"); e->force_splits = f ? atoi(f) : 0; }
    { const char* f = getenv("FSK_TILE_DMA"); e->tile_dma = f ? atoi(f) : 1; }
    { const char* f = getenv("FSK_COMPACT_DMA"); e->compact_dma = f ? atoi(f) : 0; }
    { const char* f = getenv("FSK_DENSE_CHUNK"); e->force_chunk = f ? (uint32_t)atoi(f) : 0u; }
    if (hipStreamCreate(&e->stream) != hipSuccess || hipEventCreate(&e->ev0) != hipSuccess ||
        hipEventCreate(&e->ev1) != hipSuccess) {
        delete e;
        g_create_error = "cannot create HIP stream/events";
        return FSK_EDEVICE;
    }
    e->st.n_combos_total = (int32_t)e->ncomb;
    *out = e;
    return FSK_OK;
}

void fsk_destroy(fsk_engine* e) {
    if (!e) return;
    DeviceScope on_device(e->cfg.device);
    (void)hipStreamSynchronize(e->stream);
    e->d_words.release(); e->d_wstart.release(); e->d_len.release(); e->d_fstart.release(); e->d_featseq.release();
    e->d_pos.release(); e->d_allpos.release(); e->d_bsum.release(); e->d_seqblk.release(); e->K_store.release(); e->d_Kf64.release(); e->d_Khat.release(); e->d_prod.release();
    e->d_diag.release(); e->d_stage.release(); e->d_stage_u64.release(); e->d_Kslots.release(); e->d_cell_idx.release(); e->d_C4.release(); e->d_C4H.release(); e->d_rowmask.release(); e->d_flag.release(); e->d_tiletab.release(); e->d_keybits.release(); e->d_lut.release(); e->d_vc.release();
    for (int b = 0; b < 2; ++b) e->d_keys[b].release();
    e->d_blockhist.release(); e->d_totals.release(); e->d_tile_ent.release(); e->d_ebase.release(); e->d_Pk.release();
    e->d_owner_r0.release(); e->d_ucount.release(); e->d_uchunk.release(); e->d_part_base.release(); e->d_tile_stat.release(); e->d_utot.release(); e->d_list_off.release(); e->d_ulist.release();
    e->d_tile_lrh.release(); e->d_tile_rs.release(); e->d_tile_lth.release(); e->d_tile_ts.release(); e->d_Tk.release(); e->d_E.release(); e->d_sxstat.release(); e->d_segc.release(); e->d_U.release(); e->d_U2.release();
    if (e->h_prod) (void)hipHostFree(e->h_prod);
    if (e->h_sx_pos) (void)hipHostFree(e->h_sx_pos);
    if (e->h_sx_stat) (void)hipHostFree(e->h_sx_stat);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->ev_order) (void)hipEventDestroy(e->ev_order);
    if (e->chain_stream) { (void)hipStreamSynchronize(e->chain_stream); (void)hipStreamDestroy(e->chain_stream); }
    (void)hipEventDestroy(e->ev0);
    (void)hipEventDestroy(e->ev1);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

int fsk_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n) {
    if (!e) return FSK_EINVAL;
    if (!order || n <= 0 || n > e->ncomb) return e->fail(FSK_EINVAL, "combo order must hold 1..C(g,m) ids");
    std::vector<char> seen((size_t)e->ncomb, 0);
    for (int i = 0; i < n; ++i) {
        if (order[i] < 0 || order[i] >= e->ncomb || seen[order[i]]) return e->fail(FSK_EINVAL, "combo order: id %d invalid or repeated", order[i]);
        seen[order[i]] = 1;
    }
    e->order.assign(order, order + n);
    e->order_set = true;
    return FSK_OK;
}

int fsk_set_seed(fsk_engine* e, uint64_t seed) {
    if (!e) return FSK_EINVAL;
    e->seed = seed;
    return FSK_OK;
}

int fsk_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    if (!e) return FSK_EINVAL;
    if (!offsets || n_train <= 0 || n_test < 0) return e->fail(FSK_EINVAL, "need n_train >= 1, n_test >= 0 and offsets");
    FSK_ON_DEVICE(e);
    const int64_t N = n_train + n_test;
    const int g = e->cfg.g;
    if (N >= ((int64_t)1 << 31)) return e->fail(FSK_EUNSUPPORTED, "more than 2^31 sequences");
    if (offsets[0] < 0) return e->fail(FSK_EINVAL, "offsets[0] must be >= 0");
    if (offsets[N] > offsets[0] && !tokens) return e->fail(FSK_EINVAL, "null tokens");
    if (e->stage_in_flight) {  // the previous call's upload reads the pinned staging this call is about to refill
        FSK_HIP(hipStreamSynchronize(e->stream));
        e->stage_in_flight = false;
    }
    // only tokens[offsets[0] .. offsets[N]) belong to the call: everything below works on that window
    tokens = tokens ? tokens + offsets[0] : tokens;
    const int64_t off0 = offsets[0];
    // ---- lengths (fastsk.cpp:32-58)
    int64_t shortest_train = INT64_MAX, shortest_test = INT64_MAX, longest = 0, nfeat = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (len < 0) return e->fail(FSK_EINVAL, "offsets must be non-decreasing");
        if (i < n_train) shortest_train = std::min(shortest_train, len);
        else shortest_test = std::min(shortest_test, len);
        longest = std::max(longest, len);
        nfeat += len >= g ? len - g + 1 : 0;
    }
    if (g > shortest_train)
        return e->fail(FSK_ESHORT, "g cannot be longer than the shortest sequence in a dataset. g = %d, but shortest train sequence has length %lld", g, (long long)shortest_train);
    if (n_test > 0 && g > shortest_test)
        return e->fail(FSK_ESHORT, "g cannot be longer than the shortest sequence in a dataset. g = %d, but shortest test sequence has length %lld", g, (long long)shortest_test);
    if (nfeat >= ((int64_t)1 << 31) || longest >= ((int64_t)1 << 24)) return e->fail(FSK_EUNSUPPORTED, "input too large (g-mers >= 2^31 or a sequence >= 2^24)");
    // ---- alphabet: rank-remap the tokens that occur (equality preserving; the reference's
    // dict_size = |{0} U tokens|, fastsk.cpp:70-85, only serves as its counting-sort radix), then pack,
    // every sequence word-aligned.
    const int64_t total = offsets[N] - off0;
    std::vector<int32_t> distinct;
    std::vector<int64_t> sym_freq(256, 0);
    uint32_t sigma = 1;
    int bits = 2;
    u64 V = 1;
    std::vector<uint32_t> len32((size_t)N), fstart((size_t)N + 1);
    std::vector<uint32_t> wstart_v, words_v;           // general path only
    const uint32_t *p_words = nullptr, *p_wstart = nullptr;
    size_t n_words_alloc = 0;
    bool staged = false;                               // packed into the engine's pinned staging: asynchronous upload
    {
        uint32_t fcount = 0;
        for (int64_t i = 0; i < N; ++i) {
            const int64_t len = offsets[i + 1] - offsets[i];
            len32[i] = (uint32_t)len;
            fstart[i] = fcount;
            fcount += (uint32_t)(len - g + 1);
        }
        fstart[N] = fcount;
    }
    // sigma, bits, V from `distinct`; where every sequence's words start
    auto plan_words = [&](uint32_t* wstart, uint64_t* nwords_out) -> int {
        if (distinct.size() > 256) return e->fail(FSK_EUNSUPPORTED, "alphabet of %zu symbols (> 256)", distinct.size());
        sigma = (uint32_t)std::max<size_t>(1, distinct.size());
        bits = sigma <= 4 ? 2 : sigma <= 16 ? 4 : 8;
        V = 1;
        for (int c = 0; c < e->k; ++c) {
            if (V > (((u64)1 << 62) / sigma)) return e->fail(FSK_EUNSUPPORTED, "alphabet^(g-m) does not fit in 62 bits");
            V *= sigma;
        }
        uint64_t nwords = 0;
        for (int64_t i = 0; i < N; ++i) {
            wstart[i] = (uint32_t)nwords;
            nwords += ((uint64_t)len32[i] * bits + 31) / 32;
            if (nwords >= ((uint64_t)1 << 32)) return e->fail(FSK_EUNSUPPORTED, "packed sequences exceed 2^32 words");
        }
        *nwords_out = nwords;
        return FSK_OK;
    };
    {
        // Fast path — token ids below 256, every vocabulary in practice: ONE team of host threads counts the
        // tokens of its share of the sequences (256 counters each), one of them turns the counts into the
        // rank table and lays out the words in the engine's pinned staging, the team packs its sequences
        // there; the upload is asynchronous. (At 100k x 300 tokens this is the load time.)
        const int nt = host_threads_for(total);
        std::vector<int64_t> bound((size_t)nt + 1, N);  // sequence ranges with about equal token counts
        bound[0] = 0;
        for (int t = 1; t < nt; ++t)
            bound[(size_t)t] = std::lower_bound(offsets, offsets + N, off0 + total * t / nt) - offsets;
        std::vector<std::array<int64_t, 256>> hist((size_t)nt);
        std::vector<char> small((size_t)nt, 1);
        uint8_t lut[256] = {0};
        int rc_mid = FSK_OK;
        bool all_small = true;
        uint32_t* st_words = nullptr;
        uint32_t* st_wstart = nullptr;
        parallel_two_phase(
            nt,
            [&](int t) {
                std::array<int64_t, 256>& h = hist[(size_t)t];
                h.fill(0);
                const int64_t lo = offsets[bound[(size_t)t]] - off0, hi = offsets[bound[(size_t)t + 1]] - off0;
                for (int64_t i = lo; i < hi; ++i) {
                    const uint32_t v = (uint32_t)tokens[i];
                    if (v < 256u) h[v]++;
                    else { small[(size_t)t] = 0; break; }
                }
            },
            [&] {
                for (int t = 0; t < nt; ++t) all_small = all_small && small[(size_t)t];
                if (!all_small) return;
                for (int v = 0; v < 256; ++v) {
                    int64_t c = 0;
                    for (int t = 0; t < nt; ++t) c += hist[(size_t)t][(size_t)v];
                    if (c) { lut[v] = (uint8_t)distinct.size(); sym_freq[distinct.size()] = c; distinct.push_back(v); }
                }
                // staging: [words + 4][wstart N][len N][fstart N + 1]
                std::vector<uint32_t> ws((size_t)N);
                uint64_t nwords = 0;
                rc_mid = plan_words(ws.data(), &nwords);
                if (rc_mid) return;
                const size_t need = (size_t)nwords + 4 + 3 * (size_t)N + 1;
                if (need > e->h_stage_cap) {
                    if (e->h_stage) (void)hipHostFree(e->h_stage);
                    e->h_stage = nullptr; e->h_stage_cap = 0;
                    if (hipHostMalloc((void**)&e->h_stage, (need + need / 4) * sizeof(uint32_t)) != hipSuccess) {
                        rc_mid = e->fail(FSK_ENOMEM, "cannot allocate %zu bytes of pinned staging", need * sizeof(uint32_t));
                        return;
                    }
                    e->h_stage_cap = need + need / 4;
                }
                st_words = e->h_stage;
                st_wstart = e->h_stage + (size_t)nwords + 4;
                memcpy(st_wstart, ws.data(), (size_t)N * sizeof(uint32_t));
                memcpy(st_wstart + N, len32.data(), (size_t)N * sizeof(uint32_t));
                memcpy(st_wstart + 2 * N, fstart.data(), ((size_t)N + 1) * sizeof(uint32_t));
                memset(st_words + nwords, 0, 4 * sizeof(uint32_t));
                n_words_alloc = (size_t)nwords + 4;
            },
            [&](int t) {
                if (!all_small || rc_mid) return;
                const uint32_t per_word = 32u / (uint32_t)bits;
                for (int64_t i = bound[(size_t)t]; i < bound[(size_t)t + 1]; ++i) {  // a sequence's words are its own
                    const int32_t* sq = tokens + (offsets[i] - off0);
                    uint32_t* w = st_words + st_wstart[i];
                    const uint32_t len = len32[i];
                    uint32_t p = 0;
                    for (; p + per_word <= len; p += per_word) {
                        uint32_t word = 0;
                        for (uint32_t q = 0; q < per_word; ++q) word |= (uint32_t)lut[sq[p + q]] << (q * (uint32_t)bits);
                        *w++ = word;
                    }
                    if (p < len) {
                        uint32_t word = 0;
                        for (uint32_t q = 0; p + q < len; ++q) word |= (uint32_t)lut[sq[p + q]] << (q * (uint32_t)bits);
                        *w = word;
                    }
                }
            });
        if (rc_mid) return rc_mid;
        if (all_small) {
            staged = true;
            p_words = st_words;
            p_wstart = st_wstart;
        }
    }
    if (!staged) {
        // General path (token values anywhere in int32): distinct values by sort or by a seen-table, ranks by
        // table or binary search, one thread.
        int32_t lo = INT32_MAX, hi = INT32_MIN;
        for (int64_t i = 0; i < total; ++i) { lo = std::min(lo, tokens[i]); hi = std::max(hi, tokens[i]); }
        if (total > 0 && lo >= 0 && hi < (1 << 20)) {
            std::vector<char> seen((size_t)hi + 1, 0);
            for (int64_t i = 0; i < total; ++i) seen[tokens[i]] = 1;
            for (int32_t v = 0; v <= hi; ++v) if (seen[v]) distinct.push_back(v);
        } else {
            distinct.assign(tokens, tokens + total);
            std::sort(distinct.begin(), distinct.end());
            distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        }
        wstart_v.resize((size_t)N);
        uint64_t nwords = 0;
        int rc_plan = plan_words(wstart_v.data(), &nwords);
        if (rc_plan) return rc_plan;
        words_v.assign((size_t)nwords + 4, 0u);
        const int32_t base = distinct.empty() ? 0 : distinct.front();
        const bool direct = !distinct.empty() && (int64_t)distinct.back() - base < (1 << 20);
        std::vector<uint8_t> lut;
        if (direct) {
            lut.assign((size_t)(distinct.back() - base) + 1, 0);
            for (size_t r = 0; r < distinct.size(); ++r) lut[(size_t)(distinct[r] - base)] = (uint8_t)r;
        }
        for (int64_t i = 0; i < N; ++i) {
            const int32_t* sq = tokens + (offsets[i] - off0);
            uint32_t* w = words_v.data() + wstart_v[i];
            for (uint32_t p = 0; p < len32[i]; ++p) {
                uint32_t r = direct ? lut[(size_t)(sq[p] - base)]
                                    : (uint32_t)(std::lower_bound(distinct.begin(), distinct.end(), sq[p]) - distinct.begin());
                const uint32_t bitpos = p * (uint32_t)bits;
                w[bitpos >> 5] |= r << (bitpos & 31u);
                sym_freq[r]++;
            }
        }
        p_words = words_v.data();
        p_wstart = wstart_v.data();
        n_words_alloc = words_v.size();
    }
    // ---- commit
    e->N = N; e->n_train = n_train; e->n_test = n_test; e->nfeat = nfeat;
    e->pairs = N * (N + 1) / 2;
    e->sigma = sigma; e->bits = bits; e->V = V; e->Vq = (uint32_t)((V + 3) / 4);
    e->Lmax = (uint32_t)longest; e->Lmin = (uint32_t)std::min(shortest_train, n_test > 0 ? shortest_test : shortest_train);
    e->maxW = (uint32_t)(longest - g + 1);
    e->n_panels = (uint32_t)((N + fsk::PANEL - 1) / fsk::PANEL);
    e->h_len = len32; e->h_fstart = fstart; e->featseq_ready = false;
    e->prep_valid = false; e->tab_n = 0; e->vc_sum = 0; e->vc_n = 0;
    e->lazy_lo = e->lazy_hi = -1;  // (the triangle is zeroed, or promised to be, below)
    e->u_known = false; e->u_pending = false; e->u_extra = 0; e->u_value = 0;
    e->sx_words_seen = 0;
    for (auto& d : e->sx_defer) d.active = false;
    if (e->V > DENSE_MAX_KEYS) e->Vq = 1;  // unused on the sparse path
    {   // sparse dataflow: sort record = (k-mer << sx_sb) | sequence id; owner bands of K
        e->sx_sb = 1;
        while (((int64_t)1 << e->sx_sb) < N) ++e->sx_sb;
        e->sx_keybits = 1;
        while (e->sx_keybits < 62 && ((u64)1 << e->sx_keybits) < V) ++e->sx_keybits;
        plan_owner_bands(e);
    }
    int rc = choose_path(e);
    if (rc) return rc;
    {   // key compaction pays when a symbol is rare (DNA with a few 'n'): most of the sigma^k key
        // space is then empty and need not be multiplied
        int64_t rarest = INT64_MAX;
        for (uint32_t r = 0; r < sigma; ++r) rarest = std::min(rarest, sym_freq[r]);
        e->compact = sigma >= 3 && V >= 64 && V <= 4096 && rarest * 50 < total;
        if (e->force_compact >= 0) e->compact = e->force_compact != 0 && V <= 4096;
    }
    FSK_HIP(e->d_words.reserve(n_words_alloc));
    FSK_HIP(e->d_wstart.reserve((size_t)N));
    FSK_HIP(e->d_len.reserve((size_t)N));
    FSK_HIP(e->d_fstart.reserve((size_t)N + 1));
    if (staged) {  // everything sits in pinned memory: four copies on the stream, nothing to wait for
        FSK_HIP(hipMemcpyAsync(e->d_words.p, p_words, n_words_alloc * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipMemcpyAsync(e->d_wstart.p, p_wstart, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipMemcpyAsync(e->d_len.p, p_wstart + N, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipMemcpyAsync(e->d_fstart.p, p_wstart + 2 * N, ((size_t)N + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        e->stage_in_flight = true;
    } else {
        FSK_HIP(hipMemcpy(e->d_words.p, p_words, n_words_alloc * sizeof(uint32_t), hipMemcpyHostToDevice));
        FSK_HIP(hipMemcpy(e->d_wstart.p, p_wstart, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice));
        FSK_HIP(hipMemcpy(e->d_len.p, e->h_len.data(), (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice));
        FSK_HIP(hipMemcpy(e->d_fstart.p, e->h_fstart.data(), ((size_t)N + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (e->d_K && !e->K_owned) {
        if (e->bound_cells != e->pairs) return e->fail(FSK_EINVAL, "bound counts buffer holds %lld cells, need %lld", (long long)e->bound_cells, (long long)e->pairs);
    } else {
        hipError_t r = e->K_store.reserve((size_t)e->pairs);
        if (r != hipSuccess) return e->fail(FSK_ENOMEM, "cannot allocate the %lld-cell integer triangle", (long long)e->pairs);
        e->d_K = e->K_store.p;
        e->K_owned = true;
    }
    if (lazy_zero_possible(e)) {  // (as after fsk_reset_counts: the first tile launch stores its sums)
        e->lazy_lo = 0; e->lazy_hi = N;
    } else {
        FSK_HIP(hipMemsetAsync(e->d_K, 0, (size_t)e->pairs * sizeof(u64), e->stream));
    }
    FSK_HIP(e->d_U.reserve(1));
    FSK_HIP(hipMemsetAsync(e->d_U.p, 0, sizeof(u64), e->stream));
    e->loaded = true; e->finalized = false; e->result_f64 = false;
    e->stdevs.clear();
    fsk_stats& st = e->st;
    st = fsk_stats{};
    st.n_seq = N; st.n_train = n_train; st.n_test = n_test; st.n_feat = nfeat; st.n_pairs = e->pairs;
    st.alphabet = (int32_t)sigma; st.bits_per_symbol = bits; st.key_space = (int64_t)V; st.path_used = e->path;
    st.n_combos_total = (int32_t)e->ncomb;
    st.max_windows = (double)e->maxW;
    return FSK_OK;
}

int fsk_bind_counts(fsk_engine* e, void* device_u64, int64_t n_cells) {
    if (!e) return FSK_EINVAL;
    if (!device_u64 || n_cells <= 0) return e->fail(FSK_EINVAL, "bad counts buffer");
    if (e->loaded && n_cells != e->pairs) return e->fail(FSK_EINVAL, "counts buffer holds %lld cells, need %lld", (long long)n_cells, (long long)e->pairs);
    if (e->lazy_lo >= 0) {
        FSK_ON_DEVICE(e);
        int rcz = materialise_zero(e);
        if (rcz) return rcz;
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    if (e->K_owned) e->K_store.release();
    e->bound_cells = n_cells;
    e->d_K = (u64*)device_u64;
    e->K_owned = false;
    e->finalized = false;
    return FSK_OK;
}

int fsk_counts_device_ptr(fsk_engine* e, void** out) {
    if (!e || !out) return FSK_EINVAL;
    if (!e->d_K) return e->fail(FSK_ESTATE, "no counts buffer yet");
    *out = e->d_K;
    return FSK_OK;
}

int fsk_reset_counts(fsk_engine* e) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    FSK_ON_DEVICE(e);
    if (lazy_zero_possible(e)) {
        e->lazy_lo = 0; e->lazy_hi = e->N;  // (whatever was pending is covered by this range)
    } else {
        e->lazy_lo = e->lazy_hi = -1;
        FSK_HIP(hipMemsetAsync(e->d_K, 0, (size_t)e->pairs * sizeof(u64), e->stream));
    }
    e->finalized = false; e->result_f64 = false;
    e->st.combos_done = 0;
    e->prep_valid = false;  // a new pass recounts its panels even when its first band starts at row > 0
    return FSK_OK;
}

int fsk_reset_counts_rows(fsk_engine* e, int64_t row_begin, int64_t row_end) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (row_begin < 0 || row_end > e->N || row_begin > row_end) return e->fail(FSK_EINVAL, "bad row range");
    FSK_ON_DEVICE(e);
    const u64 c0 = (u64)row_begin * ((u64)row_begin + 1) / 2, c1 = (u64)row_end * ((u64)row_end + 1) / 2;
    { int rcz = materialise_zero(e); if (rcz) return rcz; }  // an earlier, different reset
    if (lazy_zero_possible(e) && row_begin % fsk::TILE == 0 && row_end > row_begin) {
        e->lazy_lo = row_begin; e->lazy_hi = row_end;
    } else if (c1 > c0) {
        FSK_HIP(hipMemsetAsync(e->d_K + c0, 0, (size_t)(c1 - c0) * sizeof(u64), e->stream));
    }
    e->finalized = false; e->result_f64 = false;
    e->st.combos_done = 0;
    e->prep_valid = false;
    return FSK_OK;
}

int fsk_accumulate(fsk_engine* e, const int32_t* combos, int32_t n) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (n < 0 || (n > 0 && !combos)) return e->fail(FSK_EINVAL, "bad combo list");
    if (n == 0) return FSK_OK;
    FSK_ON_DEVICE(e);
    e->finalized = false;
    return do_accumulate(e, combos, n, e->d_K);
}

int fsk_accumulate_rows(fsk_engine* e, const int32_t* combos, int32_t n, int64_t row_begin, int64_t row_end) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (n < 0 || (n > 0 && !combos)) return e->fail(FSK_EINVAL, "bad combo list");
    if (row_begin < 0 || row_end > e->N || row_begin > row_end || row_begin % fsk::TILE != 0 ||
        (row_end % fsk::TILE != 0 && row_end != e->N))
        return e->fail(FSK_EINVAL, "row band must be [a,b) with a, b multiples of %d (b may be N)", fsk::TILE);
    if (n == 0 || row_begin == row_end) return FSK_OK;
    FSK_ON_DEVICE(e);
    e->finalized = false;
    return do_accumulate(e, combos, n, e->d_K, row_begin, row_end);
}

int fsk_synchronize(fsk_engine* e) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    FSK_HIP(hipStreamSynchronize(e->stream));
    FSK_HIP(hipGetLastError());
    return FSK_OK;
}

// Ordering against a stream of the caller (torch's current stream, which RCCL collectives are
// ordered after) without blocking the host: an event recorded on one stream, waited for by the other.
int fsk_stream_wait_engine(fsk_engine* e, void* hip_stream) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    if (!e->ev_order) FSK_HIP(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));
    FSK_HIP(hipEventRecord(e->ev_order, e->stream));
    FSK_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, e->ev_order, 0));
    return FSK_OK;
}

int fsk_engine_wait_stream(fsk_engine* e, void* hip_stream) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    if (!e->ev_order) FSK_HIP(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));
    FSK_HIP(hipEventRecord(e->ev_order, (hipStream_t)hip_stream));
    FSK_HIP(hipStreamWaitEvent(e->stream, e->ev_order, 0));
    return FSK_OK;
}

int fsk_finalize(fsk_engine* e) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    return make_diag(e);
}

int fsk_compute(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    int rc = fsk_load_sequences(e, tokens, offsets, n_train, n_test);
    if (rc) return rc;
    const fsk_config& c = e->cfg;
    if (!c.approx) {  // exact: every combination, order irrelevant (integer sum)
        std::vector<int32_t> all((size_t)e->ncomb);
        for (int64_t i = 0; i < e->ncomb; ++i) all[i] = (int32_t)i;
        rc = do_accumulate(e, all.data(), (int)all.size(), e->d_K);
        if (rc) return rc;
        return make_diag(e);
    }
    if (!e->order_set) default_order(e);
    int T = c.t == -1 ? 20 : c.t;  // fastsk_kernel.cpp:54-61
    T = std::max(1, std::min<int>(T, (int)e->order.size()));
    if (c.skip_variance) {  // chains only select which combos enter the integer sum
        std::vector<int32_t> used;
        for (int tid = 0; tid < T; ++tid) {
            int iters = 0;
            for (size_t item = tid; item < e->order.size(); item += T) {
                used.push_back(e->order[item]);
                if (c.max_iters != -1 && ++iters >= c.max_iters) break;
            }
        }
        rc = do_accumulate(e, used.data(), (int)used.size(), e->d_K);
        if (rc) return rc;
        return make_diag(e);
    }
    rc = run_variance_mode(e, T);
    if (rc) return rc;
    return make_diag(e);
}

int fsk_run_chains(fsk_engine* e, int32_t first, int32_t step) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    const fsk_config& c = e->cfg;
    if (!c.approx || c.skip_variance) return e->fail(FSK_ESTATE, "fsk_run_chains is the variance mode (approx=1, skip_variance=0)");
    if (first < 0 || step < 1) return e->fail(FSK_EINVAL, "need first >= 0 and step >= 1");
    FSK_ON_DEVICE(e);
    if (!e->order_set) default_order(e);
    int T = c.t == -1 ? 20 : c.t;  // fastsk_kernel.cpp:54-61
    T = std::max(1, std::min<int>(T, (int)e->order.size()));
    e->finalized = false;
    return run_variance_mode(e, T, first, step);
}

int fsk_get_kernel_sum_device(fsk_engine* e, double* device_out) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded || !e->result_f64) return e->fail(FSK_ESTATE, "no Welford chains have run");
    if (!device_out) return e->fail(FSK_EINVAL, "null output");
    FSK_ON_DEVICE(e);
    FSK_HIP(hipMemcpyAsync(device_out, e->d_Kf64.p, (size_t)e->pairs * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    FSK_HIP(hipStreamSynchronize(e->stream));
    return FSK_OK;
}

int fsk_set_kernel_sum_device(fsk_engine* e, const double* device_in) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded || !e->result_f64) return e->fail(FSK_ESTATE, "no Welford chains have run");
    if (!device_in) return e->fail(FSK_EINVAL, "null input");
    FSK_ON_DEVICE(e);
    FSK_HIP(hipMemcpyAsync(e->d_Kf64.p, device_in, (size_t)e->pairs * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    FSK_HIP(hipStreamSynchronize(e->stream));
    e->finalized = false;
    return FSK_OK;
}

int fsk_get_block(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double* out) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    return fetch_block(e, i0, i1, j0, j1, out);
}
int fsk_get_block_device(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double* device_out) {
    if (!e) return FSK_EINVAL;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel: call fsk_compute or fsk_finalize first");
    if (i0 < 0 || j0 < 0 || i1 > e->N || j1 > e->N || i0 > i1 || j0 > j1) return e->fail(FSK_EINVAL, "block out of range");
    const u64 rows = (u64)(i1 - i0), cols = (u64)(j1 - j0);
    if (rows == 0 || cols == 0) return FSK_OK;
    if (!device_out) return e->fail(FSK_EINVAL, "null output");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const u64 max_cells = (u64)1 << 31;  // grid.x limit: launch in row chunks
    const u64 rows_per = std::max<u64>(1, std::min<u64>(rows, max_cells / cols));
    for (u64 r = 0; r < rows; r += rows_per) {
        const u64 nr = std::min(rows_per, rows - r), cells = nr * cols;
        const uint32_t blocks = (uint32_t)((cells + 255) / 256);
        if (e->result_f64)
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_block<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, e->d_diag.p,
                       (u64)i0 + r, nr, (u64)j0, cols, device_out + r * cols);
        else
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_block<u64>), dim3(blocks), dim3(256), 0, e->stream, e->d_K, e->d_diag.p,
                       (u64)i0 + r, nr, (u64)j0, cols, device_out + r * cols);
    }
    FSK_HIP(hipStreamSynchronize(e->stream));
    return FSK_OK;
}

int fsk_get_train(fsk_engine* e, double* out) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    return fetch_block(e, 0, e->n_train, 0, e->n_train, out);
}
int fsk_get_test(fsk_engine* e, double* out) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    return fetch_block(e, e->n_train, e->N, 0, e->n_train, out);
}

int fsk_get_triangle(fsk_engine* e, double* out) {
    if (!e) return FSK_EINVAL;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel");
    if (!out) return e->fail(FSK_EINVAL, "null output");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const u64 chunk = (u64)32 << 20;
    FSK_HIP(e->d_stage.reserve((size_t)std::min<u64>(chunk, (u64)e->pairs)));
    for (u64 c0 = 0; c0 < (u64)e->pairs; c0 += chunk) {
        const u64 cnt = std::min<u64>(chunk, (u64)e->pairs - c0);
        const uint32_t blocks = (uint32_t)((cnt + 255) / 256);
        if (e->result_f64)
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_triangle<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, e->d_diag.p, c0, cnt, e->d_stage.p);
        else
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_triangle<u64>), dim3(blocks), dim3(256), 0, e->stream, e->d_K, e->d_diag.p, c0, cnt, e->d_stage.p);
        FSK_HIP(hipMemcpyAsync(out + c0, e->d_stage.p, cnt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    return FSK_OK;
}

int fsk_get_counts(fsk_engine* e, uint64_t* out) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (e->result_f64) return e->fail(FSK_ESTATE, "variance mode keeps a floating-point mean, not integer counts");
    if (!out) return e->fail(FSK_EINVAL, "null output");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    FSK_HIP(hipStreamSynchronize(e->stream));
    FSK_HIP(hipMemcpy(out, e->d_K, (size_t)e->pairs * sizeof(u64), hipMemcpyDeviceToHost));
    return FSK_OK;
}

int fsk_get_counts_block(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint64_t* out) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (e->result_f64) return e->fail(FSK_ESTATE, "variance mode keeps a floating-point mean, not integer counts");
    if (i0 < 0 || j0 < 0 || i1 > e->N || j1 > e->N || i0 > i1 || j0 > j1) return e->fail(FSK_EINVAL, "block out of range");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const int64_t rows = i1 - i0, cols = j1 - j0;
    if (rows == 0 || cols == 0) return FSK_OK;
    if (!out) return e->fail(FSK_EINVAL, "null output");
    const int64_t max_cells = (int64_t)32 << 20;
    const int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>(rows, max_cells / cols));
    FSK_HIP(e->d_stage_u64.reserve((size_t)(rows_per * cols)));
    for (int64_t r = 0; r < rows; r += rows_per) {
        const int64_t nr = std::min(rows_per, rows - r);
        const u64 cells = (u64)nr * cols;
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_block_raw<u64>), dim3((uint32_t)((cells + 255) / 256)), dim3(256), 0, e->stream, e->d_K,
                   (u64)(i0 + r), (u64)nr, (u64)j0, (u64)cols, e->d_stage_u64.p);
        FSK_HIP(hipMemcpyAsync(out + r * cols, e->d_stage_u64.p, cells * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    return FSK_OK;
}

int fsk_get_counts_cells(fsk_engine* e, const int64_t* rows, const int64_t* cols, int64_t n, uint64_t* out) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (e->result_f64) return e->fail(FSK_ESTATE, "variance mode keeps a floating-point mean, not integer counts");
    if (n < 0 || (n > 0 && (!rows || !cols || !out))) return e->fail(FSK_EINVAL, "bad cell list");
    for (int64_t q = 0; q < n; ++q)
        if (rows[q] < 0 || rows[q] >= e->N || cols[q] < 0 || cols[q] >= e->N) return e->fail(FSK_EINVAL, "cell (%lld, %lld) out of range", (long long)rows[q], (long long)cols[q]);
    if (n == 0) return FSK_OK;
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const int64_t chunk = (int64_t)16 << 20;
    FSK_HIP(e->d_cell_idx.reserve((size_t)std::min(n, chunk) * 2));
    FSK_HIP(e->d_stage_u64.reserve((size_t)std::min(n, chunk)));
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
        const int64_t cnt = std::min(chunk, n - c0);
        FSK_HIP(hipMemcpyAsync(e->d_cell_idx.p, rows + c0, (size_t)cnt * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipMemcpyAsync(e->d_cell_idx.p + cnt, cols + c0, (size_t)cnt * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
        FSK_LAUNCH(fsk::k_cells_raw, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, e->stream, e->d_K, (const int64_t*)e->d_cell_idx.p,
                   (const int64_t*)(e->d_cell_idx.p + cnt), (u64)cnt, e->d_stage_u64.p);
        FSK_HIP(hipMemcpyAsync(out + c0, e->d_stage_u64.p, (size_t)cnt * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    return FSK_OK;
}

int fsk_sequential_sum(fsk_engine* e, const double* values, int64_t n, double* out) {
    if (!e) return FSK_EINVAL;
    if (n < 0 || (n > 0 && !values) || !out) return e->fail(FSK_EINVAL, "bad arguments");
    FSK_ON_DEVICE(e);
    const size_t nblk = ((size_t)n + fsk::SQ_BLOCK - 1) / fsk::SQ_BLOCK;
    DevBuf<double> vals, bsum;
    DevBuf<unsigned char> blk;
    struct Free { DevBuf<double>&a, &b; DevBuf<unsigned char>& c; ~Free() { a.release(); b.release(); c.release(); } } guard{vals, bsum, blk};
    FSK_HIP(vals.reserve((size_t)std::max<int64_t>(1, n)));
    FSK_HIP(bsum.reserve(nblk + 1));
    FSK_HIP(blk.reserve((nblk + 1) * (sizeof(fsk::SeqBlk) + 2 * fsk::SQ_GROUPS * sizeof(fsk::SeqGrp))));
    FSK_HIP(hipMemcpyAsync(vals.p, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, e->stream));
    FSK_HIP(hipMemsetAsync(bsum.p, 0, (nblk + 1) * sizeof(double), e->stream));
    if (n > 0)  // approximate block sums (what k_welford accumulates on the way in variance mode)
        FSK_LAUNCH(fsk::k_block_sums, dim3((uint32_t)nblk), dim3(256), 0, e->stream, (const double*)vals.p, (u64)n, bsum.p);
    int rc = enqueue_sequential_sum(e, vals.p, (u64)n, bsum.p, reinterpret_cast<fsk::SeqBlk*>(blk.p),
                                    reinterpret_cast<fsk::SeqGrp*>(blk.p + (nblk + 1) * sizeof(fsk::SeqBlk)), bsum.p + nblk);
    if (rc) return rc;
    FSK_HIP(hipMemcpyAsync(out, bsum.p + nblk, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    FSK_HIP(hipStreamSynchronize(e->stream));
    return FSK_OK;
}

int fsk_get_stdevs(fsk_engine* e, double* out, int32_t cap, int32_t* n) {
    if (!e || !n) return FSK_EINVAL;
    *n = (int32_t)e->stdevs.size();
    for (int32_t i = 0; i < *n && i < cap && out; ++i) out[i] = e->stdevs[i];
    return FSK_OK;
}

int fsk_save_kernel(fsk_engine* e, const char* path) {
    if (!e) return FSK_EINVAL;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel");
    if (!path || !*path) return FSK_OK;  // the reference silently does nothing for an empty name
    FILE* f = fopen(path, "w");
    if (!f) return e->fail(FSK_EINVAL, "cannot open %s", path);
    std::vector<double> row((size_t)e->N);
    for (int64_t i = 0; i < e->N; ++i) {
        int rc = fsk_get_block(e, i, i + 1, 0, e->N, row.data());
        if (rc) { fclose(f); return rc; }
        for (int64_t j = 0; j < e->N; ++j) fprintf(f, "%d:%e ", (int)(j + 1), row[(size_t)j]);
        fprintf(f, "\n");
    }
    fclose(f);
    return FSK_OK;
}

int fsk_get_stats(fsk_engine* e, fsk_stats* out) {
    if (!e || !out) return FSK_EINVAL;
    if (e->d_U.p && e->loaded) {
        DeviceScope on_device(e->cfg.device);
        u64 U = 0;
        if (hipStreamSynchronize(e->stream) == hipSuccess && fetch_pending_u(e) == FSK_OK &&
            hipMemcpy(&U, e->d_U.p, sizeof U, hipMemcpyDeviceToHost) == hipSuccess)
            e->st.cell_updates = U + e->u_extra;
    }
    e->st.batches_redone = (double)e->sx_redone;
    *out = e->st;
    return FSK_OK;
}

}  // extern "C"

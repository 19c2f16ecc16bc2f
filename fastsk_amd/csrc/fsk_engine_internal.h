// fsk_engine_internal.h — the engine object behind the opaque `fsk_engine` handle of include/fastsk_amd.h,
// shared by the translation units of libfastsk_amd.so (fsk_engine.hip: one device; fsk_multi.hip: a group
// of engines on several devices of one process). Not installed, not part of the ABI.
#pragma once
#include "fsk_common.h"

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
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>

struct fsk_group;  // fsk_multi.hip: the engines of fsk_create_multi (this one leads them), their worker threads and collective

namespace fsk_detail {

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

}  // namespace fsk_detail

using fsk_detail::DevBuf;

// ---------------------------------------------------------------------------------------------
// Every knob of the engine that is not part of fsk_config, in ONE place. A key is set with fsk_set_tuning(handle,
// "key", value) — tests that force a branch, A/B tools — or through the FSK_TUNING environment variable
// ("key=value,key=value"), which fsk_create parses ONCE; nothing else in the library reads the environment, and the
// values an engine runs with are printed under trace=1. X(name, default, lowest, highest, what it does).
#define FSK_TUNING_KEYS(X)                                                                                                          \
    X(trace, 0, 0, 1, "stderr: the tuning in force, where the host time of a load and of a variance-mode call goes")                 \
    X(profile, -1, -1, 2, "fsk_config.profile from here on (-1: as the engine was created)")                                          \
    X(compact, -1, -1, 1, "dense: key compaction off / on whatever the alphabet says (-1: a rare symbol decides)")                   \
    X(compact_rare, -1, -1, 1, "dense: keys that occur from the places of the rare symbols (1) or a marking pass over every window (0)") \
    X(tile_splits, 0, 0, 4096, "dense: combo splits per tile (0: by the size of the launch)")                                        \
    X(dense_small, 0, 0, 1, "dense: 1 = a split tile launch (small N) leaves its sums as 32-bit staging blocks that k_dense_widen adds into K (0: one 64-bit atomic a cell and split — measured faster: the atomics drain under other workgroups' dot products)")                                        \
    X(dense_chunk, 0, 0, 1 << 24, "dense: cap of the count kernel's staging chunk, in windows (0: none)")                             \
    X(variance_dense_slots, 1, 0, 1, "variance mode, dense: 0 = zero fill + k_welford per iteration instead of storing slot triangles") \
    X(var_slots16, 1, 0, 1, "variance mode, sparse: 0 = u32 slot triangles from the start")                                          \
    X(sparse_global, 0, 0, 1, "sparse: every += as a 64-bit atomicAdd (k_sx_emit<DIRECT>) whatever N is")                             \
    X(sparse_unpacked, 0, 0, 1, "sparse: entries in the general 8 + 4 (+ 4) byte format whatever N is")                               \
    X(sparse_pairs, 1, 0, 1, "sparse: 0 = every update word 32 bits wide (1: unit products of clean runs travel two to a word)")      \
    X(sparse_sync, 0, 0, 1, "sparse: wait for every batch's word count and size its stream exactly")                                 \
    X(sparse_hint, 1, 0, 1, "sparse: 0 = the words per record of the previous set of sequences are never kept as a hint")             \
    X(sparse_exact_lanes, 0, 0, 2, "sparse: the batches of an exact accumulate never (1) / always (2) in two lanes (0: from six batches on)") \
    X(sparse_batch_records, 0, 0, (int64_t)1 << 31, "sparse: records per batch at most (0: 2^27)")                                   \
    X(guard_cap, 0, 0, (int64_t)1 << 40, "sparse: pretend the stream buffer holds this many words (0: its real size)")                \
    X(list_max_words, 0, 0, (int64_t)1 << 31, "sparse: update words of one batch beyond which its pairs go to K with atomics (0: 2^31)") \
    X(seg_scan_chunked, 0, 0, 1, "sparse: the three-launch segment scan whatever the tile count")                                    \
    X(extract_slots, 0, 0, 4, "sparse: slots per k_sx_extract_win workgroup, 1 or 4 (0: by the size of the launch)")                  \
    X(sparse_form, 0, 0, 3, "sparse: the update stage — 1 = owner bands whenever they exist, 2 = two-level blocks, 3 = 64-bit atomics (0: bands up to four LDS rounds a band, blocks beyond)") \
    X(blocks_sub_shift, 14, 4, 14, "sparse, blocks: log2 of the cells one k_sxb_consume workgroup sums in LDS (tests: small blocks on small inputs)") \
    X(blocks_max_bands, 512, 2, 512, "sparse, blocks: bands of one pass at most (tests: several passes on small inputs)")            \
    X(blocks_band_shift_max, 23, 4, 23, "sparse, blocks: log2 of a band's cells at most (tests)")                                    \
    X(blocks_scatter_threads, 0, 0, 1024, "sparse, blocks: threads of a k_sxb_scatter workgroup, 256 / 512 / 1024 (0: 256)")          \
    X(blocks_pass_words, 0, 0, (int64_t)1 << 32, "sparse, blocks: update words of one pass at most (0: 2^31)")                       \
    X(sparse_desc, 0, -1, 1, "sparse, owner bands and two-level blocks: entries of many partners leave k_sx_emit as descriptors (one an entry; blocks: one per sub-band its partners fall into) that k_sx_consume / k_sxb_consume expand in LDS — 1 = always, -1 = never (0: once a batch of these sequences has shown sparse_desc_from pairs per record)") \
    X(sparse_desc_blocks, 1, 0, 1, "sparse, two-level blocks: 0 = never descriptors there (1: as sparse_desc says — one record per sub-band an entry's partners fall into)") \
    X(sparse_desc_cols, 1, 0, 3, "sparse, descriptors: where the partners are read from — 0 = the entries themselves; 1 = a column array (sequence id | multiplicity, 4 bytes) beside 8-byte entries, the entries themselves when they are packed (measured: 2-byte columns beside packed entries are no faster, 0.88 against 0.84 s in the large-g regime; 4-byte columns beside 8-byte entries 4.69 -> 4.44 ms a combo at N = 100k); tests: 2 = 4-byte columns always, 3 = 2-byte columns when N < 32768") \
    X(sparse_desc_min, 16, 1, 48, "sparse, descriptors: entries of more partners than this become descriptors (48: everything k_sx_emit does not bin in LDS; measured, large-g regime: 48 0.92 s, 32 0.85, 16 0.83-0.85, 8 0.88, 4 0.90)") \
    X(sparse_desc_from, 12, 1, 1 << 20, "sparse, descriptors: the pairs per sort record of a batch from which on the following batches use them (sparse_desc = 0)") \
    X(sparse_parts_target, 0, 0, 1 << 30, "sparse, owner bands: words of one k_sx_consume part (0: four LDS rounds' worth at least, 1/1024 of the batch's words; tests: several parts a band on small inputs)") \
    X(sparse_desc_parts, 2048, 1, 1 << 16, "sparse, descriptors: parts (k_sx_consume workgroups) the bands' streams are cut into, about") \
    X(sparse_share, 0, -1, 254, "sparse: leading kept positions sorted once per group of consecutive combos that share them (0: by cost; -1: never); batches of more than 16 slots only (smaller ones read their positions by id), and never when the presort's scratch passes a quarter of the free memory") \
    X(seed_splitmix, 0, 0, 1, "approx modes: 1 = fsk_set_seed draws the engine's older splitmix64 Fisher-Yates order (0: the reference's std::shuffle of minstd_rand0)") \
    X(collective, 0, 0, 2, "fsk_create_multi: FSK_COLL_* when fsk_config.collective is FSK_COLL_AUTO")                                \
    X(deadline_ms, 120000, -1, 86400000, "fsk_create_multi: the fail-fast bound when fsk_config.deadline_ms is 0 (negative: none)")   \
    FSK_TUNING_TEST_KEYS(X)
// keys that exist in test builds only (-DFSK_TEST_HOOKS: the CPU emulation and tests/hooks' library; never the product):
// engine fault_rank of a group is late by fault_ms before the collective of band fault_band — fault_kind 1: its worker
// thread sleeps; 2: a bounded spin kernel holds its exchange stream
#ifdef FSK_TEST_HOOKS
#define FSK_TUNING_TEST_KEYS(X)                                            \
    X(fault_kind, 0, 0, 3, "test hook: 1 = host sleep, 2 = device spin before a band's collective; 3 = host sleep after the variance chains")   \
    X(fault_rank, -1, -1, 15, "test hook: the engine that is late")        \
    X(fault_band, -1, -1, 63, "test hook: before this band's collective")  \
    X(fault_ms, 0, 0, 60000, "test hook: by this many milliseconds")
#else
#define FSK_TUNING_TEST_KEYS(X)
#endif
struct fsk_tuning {
#define FSK_X(name, def, lo, hi, doc) int64_t name = (def);
    FSK_TUNING_KEYS(FSK_X)
#undef FSK_X
};

// What one batch of the sparse dataflow works in: sort records, tile records, entries, update streams. Two sets
// ("lanes"): variance mode keeps two batches in flight and runs them on two streams, so that the many short
// kernels of one batch's sort and segmentation fill the gaps of the other's emit / consume / Welford kernels.
struct SxScratch {
    DevBuf<unsigned char> d_keys[2];      // packed sort records (u32 or u64), double-buffered
    DevBuf<uint32_t> d_blockhist, d_totals, d_tile_ent, d_ebase, d_Pk, d_Tk, d_ucount, d_uchunk, d_utot, d_list_off, d_ulist, d_part_base;
    DevBuf<u64> d_tile_stat;
    DevBuf<uint32_t> d_segc;              // chunk records of the segment scan (batches of many tiles)
    DevBuf<int> d_tile_lrh, d_tile_rs, d_tile_lth, d_tile_ts;
    DevBuf<uint2> d_E;                    // entries: {sequence, multiplicity} (or the packed format)
    DevBuf<u64> d_sxstat;
    DevBuf<uint32_t> d_group_of, d_group_head, d_winp;  // shared prefixes (k_sx_group_tables): slot -> group, group -> first slot,
    DevBuf<unsigned char> d_part;                        // the windows and part records of every group in presorted order
    DevBuf<uint32_t> d_ulist2, d_subcnt, d_suboff, d_subcur;  // blocks form: the stream split by sub-band, words / start / cursor per (band, sub-band)
    DevBuf<uint32_t> d_dsubcnt, d_dsuboff, d_dsubcur;         //              the same for the descriptor records
    DevBuf<uint32_t> d_cols;                                  // descriptors: the entries' column array (2 or 4 bytes an entry)
    void release() {
        d_ulist2.release(); d_subcnt.release(); d_suboff.release(); d_subcur.release();
        d_dsubcnt.release(); d_dsuboff.release(); d_dsubcur.release(); d_cols.release();
        d_group_of.release(); d_group_head.release(); d_winp.release(); d_part.release();
        for (auto& k : d_keys) k.release();
        d_blockhist.release(); d_totals.release(); d_tile_ent.release(); d_ebase.release(); d_Pk.release(); d_Tk.release();
        d_ucount.release(); d_uchunk.release(); d_utot.release(); d_list_off.release(); d_ulist.release(); d_part_base.release();
        d_tile_stat.release(); d_segc.release(); d_tile_lrh.release(); d_tile_rs.release(); d_tile_lth.release(); d_tile_ts.release();
        d_E.release(); d_sxstat.release();
    }
};

struct fsk_engine {
    fsk_config cfg{};
    int cfg_profile0 = 0;  // fsk_config.profile as the engine was created (tuning key profile = -1 restores it)
    fsk_tuning tune{};  // (fsk_set_tuning / FSK_TUNING at fsk_create; see FSK_TUNING_KEYS)
    std::string err;
    int k = 0;
    int64_t ncomb = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_out = nullptr, ev_in = nullptr;  // fsk_stream_wait_engine / fsk_engine_wait_stream: one event per direction
    fsk_group* group = nullptr;          // fsk_create_multi: this engine leads a group (fsk_multi.hip); its other engines have none
    hipStream_t chain_stream = nullptr;  // variance mode: the sequential sums of a batch, under the next batches' kernels
    // fsk_reset_counts does not fill K when the next accumulate can STORE its sums instead of adding
    // them (dense dataflow, one workgroup per tile): rows [lazy_lo, lazy_hi) are zero by contract
    // but not in memory until a tile launch stores them or materialise_zero() fills them.
    int64_t lazy_lo = -1, lazy_hi = -1;
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
    DevBuf<uint32_t> d_win;  // sparse dataflow: the g-mer windows, win_words 32-bit words each (0: g * bits > 128, symbols are gathered)
    int win_words = 0;
    std::vector<uint32_t> h_len, h_fstart;
    bool featseq_ready = false;

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
    DevBuf<uint32_t> d_stage32;   // small N: the split tile launch's staging blocks (fsk_engine_dense_small.hip)
    DevBuf<uint32_t> d_keybits;   // key compaction: per-combo bitmap of the keys that occur
    DevBuf<uint16_t> d_lut, d_vc; //                  rank table and key count per combo
    bool compact = false;         // decided at load: the alphabet has a rare symbol
    // key compaction from the places of the rare symbols (k_dense_rare_scan / k_dense_mark_rare) instead of a marking pass
    // over every window: decided at load (few places, plenty of windows per common key); tuning compact_rare = 0: never
    bool compact_rare = false, rare_ready = false;
    uint32_t rare_mask = 0, rare_places = 0;
    DevBuf<u64> d_rare;
    DevBuf<uint32_t> d_rare_n, d_common;  // (d_common: the bitmap of the keys made of common symbols alone)
    std::vector<uint16_t> h_vc_cache;
    double vc_sum = 0, vc_n = 0;
    DevBuf<uint32_t> d_tiletab;
    uint32_t tab_t0 = 0, tab_t1 = 0, tab_n = 0;   // tile-row range the table on the device covers
    uint32_t tab_ftt = 0xffffffffu;               // ... and the first all-test tile column it was built for (skip_test_block)
    std::vector<int32_t> prep_combos;              // combos whose count panels are resident
    bool prep_valid = false, prep_overflow = false;
    // sparse scratch
    SxScratch sxs[2];                     // lane 0: every exact accumulate; lanes 0 and 1: variance mode's batches in flight
    hipStream_t lane_stream = nullptr;    // lane 1's stream (lane 0 runs on `stream`)
    hipEvent_t ev_lane[4] = {nullptr, nullptr, nullptr, nullptr};  // exact accumulate in two lanes: fork, join, K handed on by lane 0 / 1
    DevBuf<uint32_t> d_owner_r0;
    DevBuf<u64> d_U;
    std::vector<uint32_t> h_owner_r0;     // owner bands of K: rows [r0[o], r0[o+1])
    uint32_t n_owners = 0, sx_rounds = 1, sx_cap = 0;
    uint32_t sx_rounds_slot = 1, sx_cap_slot = 0;  // the same for the by-slot form of k_sx_consume (variance mode)
    int sx_pb = 16, sx_sb = 1, sx_keybits = 1, sx_own_shift = 13;
    int sx_symbits = 0;  // != 0: the k-mer space passes 2^62 and a key is the symbols' sx_symbits-bit fields side by side, not a mixed-radix number
    bool sx_lists = false, owner_ready = false;
    int n_cu = 256;   // compute units of the device (persistent launches)
    int sx_form = 0;  // the update stage of the sparse dataflow for these sequences: 0 = owner bands, 1 = 64-bit atomics, 2 = two-level blocks
    int sx_form_used = -1;  // ... what the last batch really took (tuning key sparse_form_used reads it)
    u64 sx_passes = 0;      // blocks form: passes run since the sequences were loaded
    int sx_share_used = 0;  // shared prefixes: leading positions / groups of the last batch (fsk_stats)
    uint32_t sx_share_groups = 0;
    DevBuf<uint32_t> d_blk_r0;  // blocks form: the band table of the pass at hand
    bool sx_pairs_asked = true;
    bool sx_pairs = false;  // update streams: unit products travel as 15-bit cells, two to a 32-bit container (bands of < 32767 cells)
    int64_t owner_N = -1;                 // the number of sequences the owner bands were planned for
    // profile mode, dense dataflow: U of the last single-chunk combo list is kept, so that repeating
    // the same pass (bench steps, row bands of later passes) does not re-read every count panel
    DevBuf<u64> d_U2;
    std::vector<int32_t> u_combos;
    bool u_known = false, u_pending = false;
    u64 u_value = 0, u_extra = 0;
    // Batches are enqueued without waiting for their word counts once one batch of these sequences has
    // been sized: the stream buffer keeps headroom over the largest count seen, the kernels leave a
    // batch that does not fit alone, and the host redoes such a batch (sized exactly) when it reads the
    // counts back — at the end of an exact accumulate, at the hand-over of a variance-mode batch.
    unsigned char* h_sx_pos = nullptr;   // pinned: positions of the batches of ONE exact accumulate (read back before it returns)
    size_t h_sx_pos_cap = 0;
    u64* h_sx_stat = nullptr;            // pinned: {pairs, words} of those batches
    size_t h_sx_stat_cap = 0;
    unsigned char* h_sx_head_pos = nullptr;  // pinned, fixed size, never moved: positions of variance mode's deferred batches (in flight across calls)
    u64* h_sx_head_stat = nullptr;           //                                   their {pairs, words}
    // variance mode, grouped sparse batches: u16 slot triangles (half the bytes between k_sx_consume and the Welford pass); a
    // sum above 65535 raises the batch's flag in pinned memory and the batch is redone with u32 triangles, which the
    // sequences then keep (slots16_ok). Tuning var_slots16 = 0: never.
    bool sx_slot16 = false;              // set by run_variance_mode around the accumulate of a deferred batch
    bool sx_slot16_used = false;         // what accumulate_sparse really did
    bool slots16_ok = true;
    uint32_t* h_sx_head_flag = nullptr;  // pinned, fixed size: one overflow flag per deferred batch
    uint32_t* sx_ovf_now = nullptr;      // the flag of the batch being enqueued
    int sx_last_lane = 0;                // the lane (scratch + stream) the last accumulate_sparse ran in
    u64 sx_shape[4] = {0, 0, 0, 0};      // sequences, windows, alphabet, longest sequence of the set at hand
    double sx_ppr = 0;                   // most pairs (the reference's +=) per sort record of a batch since the sequences were loaded
    bool sx_desc_used = false;           // the last batch of the owner-band form sent descriptors (fsk_stats.sparse_desc)
    double sx_wpr = 0;                   // most update words per sort record of a batch since the sequences were loaded (0: none seen)
    u64 sx_words_of(u64 nrec) const { return (u64)(sx_wpr * (double)nrec) + 1; }  // what a batch of nrec records is expected to emit
    void sx_saw(u64 words, u64 nrec) { if (nrec) sx_wpr = std::max(sx_wpr, std::max(1e-9, (double)words / (double)nrec)); }
    // descriptors (tuning sparse_desc): forced, or once the sequences have shown long runs. The batch that tips the
    // decision forgets the words per record seen so far — with descriptors a batch emits a fraction of them — so the
    // next batch is sized exactly again.
    bool sx_desc_now() const { return tune.sparse_desc > 0 || (tune.sparse_desc == 0 && sx_ppr >= (double)tune.sparse_desc_from); }
    void sx_saw_pairs(u64 pairs, u64 nrec) {
        if (!nrec) return;
        const bool was = sx_desc_now();
        sx_ppr = std::max(sx_ppr, (double)pairs / (double)nrec);
        if (!was && sx_desc_now()) sx_wpr = 0;
    }
    struct SxDefer { bool active = false; u64 cap = 0, nrec = 0; } sx_defer[8];
    u64 sx_redone = 0;                   // batches redone because they did not fit
    bool sx_redoing = false;             // set around such a redo: the batch waits for its word count and is sized exactly
    bool sx_exactly() const { return tune.sparse_sync != 0 || sx_redoing; }
    // update words per batch beyond which the pairs go to K with atomics
    u64 sx_max_words() const { return tune.list_max_words > 0 ? std::min<u64>((u64)1 << 31, (u64)tune.list_max_words) : (u64)1 << 31; }
    bool trace() const { return tune.trace != 0; }

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
    // ---- HIP-event timing of the kernel families (fsk_stats.ms_*). fsk_config.profile = 1: measurement mode — every interval
    // is waited for on the spot (and the exact update count U is computed; the sparse dataflow runs on one stream, every batch
    // sized exactly). profile = 2: the PRODUCT dataflow, untouched — the events are only recorded (on the stream the kernels
    // are launched on) and their times are harvested later, by fsk_get_stats or when some hundred intervals are pending.
    struct LazyTime { hipEvent_t a, b; double* acc; };
    std::vector<LazyTime> lazy_pending;
    std::vector<hipEvent_t> lazy_free;
    hipEvent_t lazy_open = nullptr;
    bool profile_sync() const { return cfg.profile == 1; }
    hipEvent_t lazy_event() {
        hipEvent_t ev = nullptr;
        if (!lazy_free.empty()) { ev = lazy_free.back(); lazy_free.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
        return ev;
    }
    void harvest_times() {
        for (const LazyTime& t : lazy_pending) {
            float ms = 0;
            if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) *t.acc += ms;
            lazy_free.push_back(t.a);
            lazy_free.push_back(t.b);
        }
        lazy_pending.clear();
    }
    // the intervals whose end event has been reached already (never a wait: profile = 2 leaves the dataflow alone)
    void harvest_finished() {
        size_t keep = 0;
        for (size_t i = 0; i < lazy_pending.size(); ++i) {
            const LazyTime t = lazy_pending[i];
            float ms = 0;
            if (hipEventQuery(t.b) == hipSuccess) {
                if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) *t.acc += ms;
                lazy_free.push_back(t.a);
                lazy_free.push_back(t.b);
            } else {
                (void)hipGetLastError();
                lazy_pending[keep++] = t;
            }
        }
        lazy_pending.resize(keep);
    }
    void lazy_interval(hipEvent_t a, hipEvent_t b, double* acc) {
        if (!a || !b) { if (a) lazy_free.push_back(a); if (b) lazy_free.push_back(b); return; }
        lazy_pending.push_back(LazyTime{a, b, acc});
        // (a long accumulate: collect what has finished; what has not stays pending — the vector grows, nothing blocks.
        // Only fsk_get_stats, a change of mode and fsk_destroy wait: harvest_times)
        if (lazy_pending.size() >= 512 && lazy_pending.size() % 128 == 0) harvest_finished();
    }
    void tic(hipStream_t s = nullptr) {
        if (cfg.profile == 1) (void)hipEventRecord(ev0, s ? s : stream);
        else if (cfg.profile == 2) {
            if (lazy_open) lazy_free.push_back(lazy_open);
            lazy_open = lazy_event();
            if (lazy_open) (void)hipEventRecord(lazy_open, s ? s : stream);
        }
    }
    void toc(double* acc, hipStream_t s = nullptr) {
        if (cfg.profile == 1) {
            (void)hipEventRecord(ev1, s ? s : stream);
            (void)hipEventSynchronize(ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            *acc += ms;
        } else if (cfg.profile == 2 && lazy_open) {
            hipEvent_t b = lazy_event();
            if (b) (void)hipEventRecord(b, s ? s : stream);
            lazy_interval(lazy_open, b, acc);
            lazy_open = nullptr;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// What the translation units of the library call in each other (all on the engine's device, which
// the C-ABI entry point has made current).
namespace fsk_detail {

constexpr size_t LDS_BUDGET = 150 * 1024;       // of 160 KiB per CU
constexpr u64 DENSE_MAX_KEYS = 16384;           // count panels: alphabet^k <= this (DNA up to k = 7)
constexpr size_t SPARSE_MAX_RECORDS = 1u << 27; // records per sort batch (0.5 GB of 4-byte records; each batch pays ~20 launches)
constexpr int FSK_RETRY_UNGROUPED = 1;  // internal: a per-slot sparse batch has to be redone one combo at a time

// fsk_engine.hip
int tuning_set(fsk_tuning& t, const char* key, int64_t value, std::string& err);  // FSK_EINVAL: unknown key / out of range
int tuning_parse(fsk_tuning& t, const char* text, std::string& err);             // "key=value,key=value"
std::string tuning_in_force(const fsk_tuning& t);                                // the keys that differ from their defaults
void set_create_error(const std::string& msg);  // what fsk_last_error(NULL) reports (per thread)
int64_t n_choose_k(int n, int k);
int materialise_zero(fsk_engine* e);
bool lazy_zero_possible(const fsk_engine* e);
int do_accumulate(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0 = 0, int64_t row1 = -1, u64 slot_stride = 0,
                  int defer = -1);
int make_diag(fsk_engine* e);
void default_order(fsk_engine* e);
void libstdcxx_shuffle_order(uint64_t seed, int64_t n, int32_t* out);

// fsk_engine_dense.hip
struct DensePlan { uint32_t CH = 0, Vcq = 0; size_t lds = 0; };
DensePlan dense_plan(uint32_t maxW, int g, uint32_t Vq, size_t extra = 0);
int fetch_pending_u(fsk_engine* e);
int accumulate_dense(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1);
// fsk_engine_dense_small.hip: the split tile launch at small N through 32-bit staging blocks
size_t dense_small_stage_bytes(u64 n_tiles, int n_splits);
int dense_tile_small(fsk_engine* e, bool compact, u64 n_tiles, int n_splits, int nb, uint32_t Vq8, uint32_t nst, u64* K, int slots_per_split);

// fsk_engine_sparse.hip
// which lane (scratch set + stream) the deferred batch `defer` of variance mode runs in
inline int sx_lane_of(const fsk_engine* e, int defer) { return (defer >= 0 && !e->profile_sync()) ? (defer & 1) : 0; }
// the bands of one pass of the two-level form over rows [ra, rb) (fsk_sparse_blocks.inc)
struct SxPass { int64_t ra = 0, rb = 0; int t = 14, sub_shift = 14, pb = 8; uint32_t n_owners = 0, submax = 0, own_base = 0; std::vector<uint32_t> r0; };
bool blocks_plan_pass(fsk_engine* e, int64_t ra, int64_t rb, SxPass* out);
void plan_owner_bands(fsk_engine* e);
bool sx_harvest(fsk_engine* e, int slot);
int accumulate_sparse(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1, u64 slot_stride = 0,
                      int defer = -1);

// fsk_engine_variance.hip
int run_variance_mode(fsk_engine* e, int T, int chain_first = 0, int chain_step = 1);

// The single-engine bodies of the C-ABI calls that a group (fsk_create_multi) spreads over its engines;
// the extern "C" entry points dispatch on fsk_engine::group. fsk_engine.hip.
int one_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test);
int one_reset_counts(fsk_engine* e);
int one_reset_counts_rows(fsk_engine* e, int64_t row_begin, int64_t row_end);
int one_accumulate_rows(fsk_engine* e, const int32_t* combos, int32_t n, int64_t row_begin, int64_t row_end);
int one_synchronize(fsk_engine* e);
int one_finalize(fsk_engine* e);
int one_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n);
int one_get_stats(fsk_engine* e, fsk_stats* out);
void one_destroy(fsk_engine* e);
// the combos the approx modes accumulate as plain integer sums (skip_variance): fastsk_kernel.cpp:148,275
void skip_variance_combos(fsk_engine* e, std::vector<int32_t>& used);
int approx_chains(const fsk_engine* e);  // T of fastsk_kernel.cpp:54-61

// fsk_multi.hip: the same calls on a group (e->group != nullptr)
int group_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test);
int group_reset_counts(fsk_engine* e, int64_t row_begin, int64_t row_end);
int group_accumulate(fsk_engine* e, const int32_t* combos, int32_t n);
int group_synchronize(fsk_engine* e);
int group_finalize(fsk_engine* e);
int group_compute(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test);
int group_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n);
int group_set_seed(fsk_engine* e, uint64_t seed);
int group_get_stats(fsk_engine* e, fsk_stats* out);
int group_set_skip_test_block(fsk_engine* e, int32_t skip);
int group_set_tuning(fsk_engine* e, const char* key, int64_t value, std::string& err);
void group_set_profile(fsk_engine* e, int profile);
void group_destroy(fsk_engine* e);
void group_note_bound_counts(fsk_engine* e);  // fsk_bind_counts on a group handle: the bound cells are not bounded by the combos since a reset

}  // namespace fsk_detail

// fsk_engine.hip — host side of the C ABI declared in include/fastsk_amd.h.
//
// Replaces, for the kernel-construction path only, the reference's host driver
// (FastSK::compute_kernel / compute_train, fastsk.cpp:30-188), its engine
// (KernelFunction::compute_kernel / kernel_build_parallel / get_variance,
// fastsk_kernel.cpp:24-322) and the getters (fastsk.cpp:190-237). No CPU compute fallback lives
// here: every count is produced by the HIP kernels of fsk_kernels_*.h, and construction fails
// loudly when no device is usable.
#include "fsk_engine_internal.h"
#include "fsk_kernels_result.h"

using namespace fsk_detail;

namespace {

thread_local std::string g_create_error;

}  // namespace

namespace fsk_detail {

void set_create_error(const std::string& msg) { g_create_error = msg; }

// ---- tuning (fsk_tuning, fsk_engine_internal.h): one table of keys, defaults and ranges
namespace {
struct TuneKey { const char* name; int64_t fsk_tuning::*field; int64_t def, lo, hi; const char* doc; };
const TuneKey TUNE_KEYS[] = {
#define FSK_X(name, def, lo, hi, doc) {#name, &fsk_tuning::name, (int64_t)(def), (int64_t)(lo), (int64_t)(hi), doc},
    FSK_TUNING_KEYS(FSK_X)
#undef FSK_X
};
}  // namespace

int tuning_set(fsk_tuning& t, const char* key, int64_t value, std::string& err) {
    for (const TuneKey& k : TUNE_KEYS)
        if (key && !strcmp(k.name, key)) {
            if (value < k.lo || value > k.hi) {
                err = std::string("tuning key ") + key + " takes " + std::to_string(k.lo) + " .. " + std::to_string(k.hi) + ", not " + std::to_string(value);
                return FSK_EINVAL;
            }
            t.*(k.field) = value;
            return FSK_OK;
        }
    err = std::string("unknown tuning key '") + (key ? key : "(null)") + "'";
    return FSK_EINVAL;
}

// "key=value,key=value" (commas or blanks between the pairs)
int tuning_parse(fsk_tuning& t, const char* text, std::string& err) {
    std::string s(text ? text : "");
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ',' || s[i] == ' ' || s[i] == ';')) ++i;
        size_t j = i;
        while (j < s.size() && s[j] != ',' && s[j] != ' ' && s[j] != ';') ++j;
        if (j > i) {
            const std::string pair = s.substr(i, j - i);
            const size_t eq = pair.find('=');
            char* end = nullptr;
            const long long v = eq == std::string::npos ? 0 : strtoll(pair.c_str() + eq + 1, &end, 0);
            if (eq == std::string::npos || eq == 0 || eq + 1 == pair.size() || (end && *end)) {
                err = "expected key=integer, got '" + pair + "'";
                return FSK_EINVAL;
            }
            const int rc = tuning_set(t, pair.substr(0, eq).c_str(), (int64_t)v, err);
            if (rc) return rc;
        }
        i = j;
    }
    return FSK_OK;
}

// the keys that differ from their defaults ("defaults" when none does)
std::string tuning_in_force(const fsk_tuning& t) {
    std::string out;
    for (const TuneKey& k : TUNE_KEYS)
        if (t.*(k.field) != k.def) out += (out.empty() ? "" : ",") + std::string(k.name) + "=" + std::to_string(t.*(k.field));
    return out.empty() ? "defaults" : out;
}

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
    // (a thread start costs ~50 us; the 3 * 10^7 tokens of the 100,000 x 300 workload are worth 32 of them)
    const unsigned want = work_items < ((int64_t)1 << 20) ? 4u : work_items < ((int64_t)1 << 23) ? 8u : work_items < ((int64_t)1 << 24) ? 16u : 32u;
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

// one sequence's tokens, rank-remapped through `lut`, BITS per symbol into its own words
template <int BITS>
inline void pack_sequence(const int32_t* sq, uint32_t len, const uint8_t* lut, uint32_t* w) {
    constexpr uint32_t PER = 32u / (uint32_t)BITS;
    uint32_t p = 0;
    for (; p + PER <= len; p += PER) {
        uint32_t word = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) word |= (uint32_t)lut[sq[p + q]] << (q * (uint32_t)BITS);
        *w++ = word;
    }
    if (p < len) {
        uint32_t word = 0;
        for (uint32_t q = 0; p + q < len; ++q) word |= (uint32_t)lut[sq[p + q]] << (q * (uint32_t)BITS);
        *w = word;
    }
}

uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
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
    // sparse: sort + segments per g-mer (3.1e-11 s each), then one update per (run, pair). d = sequences holding a given key.
    // SHORT runs — update words through emit + consume: 1.7e11 updates/s on owner bands of one LDS round, a band of several
    // rounds re-reads its stream once per round; beyond four rounds the two-level blocks, ~1.0e11/s (N = 100k: 9.3e10); where
    // neither exists per-pair global atomics, 1.6e10/s. LONG runs (12 += a sort record and more, i.e. d >= ~24) — DESCRIPTORS,
    // bands of one round or blocks: the longer the run the cheaper a pair (a descriptor's fixed cost, then four partners a
    // 16-byte load): 2.2e12 d / (d + 300) updates/s fits profiles/dense_vs_sparse_regimes.jsonl (DNA k = 4 .. 7, N = 4000 and
    // 16000: 0.2e12 at d = 22, 0.5 at 88, 1.2 at 300, 1.6 at 1000, 2.0 at 2700) within a fifth. The blocks sweep K once a batch
    // (16 B a cell at ~5 TB/s, a batch = 2^27 records). The approx modes' variance form (slot triangles, owner bands only)
    // does not have the blocks and expands a slot's few descriptors in workgroups of their own: priced as the word streams.
    const double d = N * (1.0 - std::exp(-W / V));
    const double U = V * d * (d + 1.0) / 2.0;
    const bool variance_form = e->cfg.approx && !e->cfg.skip_variance;
    const bool blocks_ok = !variance_form && blocks_plan_pass(const_cast<fsk_engine*>(e), 0, e->N, nullptr);
    const bool bands_few_rounds = e->sx_lists && e->sx_rounds <= 4;
    double rate = bands_few_rounds ? 1.7e11 / (1.0 + 0.3 * ((double)e->sx_rounds - 1.0))
                  : blocks_ok      ? 1.0e11
                  : e->sx_lists    ? 1.7e11 / (1.0 + 0.3 * ((double)e->sx_rounds - 1.0))
                                   : 1.6e10;
    const bool descriptors = !variance_form && e->tune.sparse_desc >= 0 && d >= 24.0 && ((e->sx_lists && e->sx_rounds <= 1) || blocks_ok);
    if (descriptors) rate = std::max(rate, 2.2e12 * d / (d + 300.0));
    double sparse = U / rate + (double)e->nfeat * 3.1e-11;
    if (!bands_few_rounds && blocks_ok) {
        const double per_batch = std::max(1.0, std::floor((double)SPARSE_MAX_RECORDS / std::max(1.0, (double)e->nfeat)));
        sparse += 0.5 * N * N * 16.0 / 5e12 / per_batch;
    }
    return dense <= sparse;
}

int choose_path(fsk_engine* e) {
    bool dense_ok = e->V <= DENSE_MAX_KEYS && e->bits <= 8 && e->k <= 16 && e->Lmax < 65536 && dense_plan(e->maxW, e->cfg.g, e->Vq).CH > 0;
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
    return e->path == FSK_PATH_DENSE && !e->compact && !(e->cfg.skip_test_block && e->n_test > 0);
}



int do_accumulate(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1, u64 slot_stride,
                  int defer) {
    if (row1 < 0) row1 = e->N;
    for (int i = 0; i < n; ++i)
        if (combos[i] < 0 || combos[i] >= e->ncomb) return e->fail(FSK_EINVAL, "combo id %d out of range [0,%lld)", combos[i], (long long)e->ncomb);
    hipEvent_t a = nullptr, b = nullptr;
    if (e->cfg.profile) {  // (the whole call on the engine's stream; profile = 2: harvested later, see fsk_engine::tic)
        a = e->lazy_event();
        if (a) (void)hipEventRecord(a, e->stream);
    }
    int rc = e->path == FSK_PATH_DENSE ? accumulate_dense(e, combos, n, K, row0, row1) : accumulate_sparse(e, combos, n, K, row0, row1, slot_stride, defer);
    if (e->cfg.profile) {
        b = e->lazy_event();
        if (b) (void)hipEventRecord(b, e->stream);
        e->lazy_interval(a, b, &e->st.ms_total);
        if (e->profile_sync()) e->harvest_times();
    }
    if (rc == FSK_OK && row1 >= e->N) {  // a combo is done when its last row band is
        e->st.combos_done += n;
        e->st.combos_issued += (double)n;
    }
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

// The order the reference draws for this seed: std::shuffle(indexes, std::default_random_engine{seed}) of libstdc++
// (fastsk_kernel.cpp:31-38 seeds it with time(0)), restated. default_random_engine is minstd_rand0, x <- 16807 x mod
// (2^31 - 1), values in [1, 2^31 - 2]; uniform_int_distribution over [0, r] takes its "downscaling" branch (reject draws
// from urange / (r + 1) * (r + 1) on, divide by the scaling); std::shuffle swaps element i with a draw from [0, i] —
// two elements per draw (one number below i * (i + 1) split by / and %) while n * n fits the generator's range, after
// a single swap up front when n is even.
void libstdcxx_shuffle_order(uint64_t seed, int64_t n, int32_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (int32_t)i;
    if (n < 2) return;
    const uint64_t M = 2147483647ull, RANGE = M - 2;  // max() - min()
    uint64_t x = seed % M;
    if (x == 0) x = 1;  // linear_congruential_engine::seed with c = 0
    auto below = [&](uint64_t count) {  // uniform in [0, count): uniform_int_distribution{0, count - 1}
        if (count - 1 == RANGE) { x = x * 16807ull % M; return x - 1; }
        const uint64_t scaling = RANGE / count, past = count * scaling;
        uint64_t r;
        do { x = x * 16807ull % M; r = x - 1; } while (r >= past);
        return r / scaling;
    };
    const uint64_t un = (uint64_t)n;
    if (RANGE / un >= un) {
        int64_t i = 1;
        if (un % 2 == 0) { std::swap(out[i], out[below(2)]); ++i; }
        while (i < n) {
            const uint64_t r = (uint64_t)i + 1, both = below(r * (r + 1));
            std::swap(out[i], out[both / (r + 1)]); ++i;
            std::swap(out[i], out[both % (r + 1)]); ++i;
        }
        return;
    }
    for (int64_t i = 1; i < n; ++i) std::swap(out[i], out[below((uint64_t)i + 1)]);
}

void default_order(fsk_engine* e) {
    e->order.resize((size_t)e->ncomb);
    if (e->tune.seed_splitmix == 0) {  // the reference's own sample for this seed
        libstdcxx_shuffle_order(e->seed, e->ncomb, e->order.data());
        return;
    }
    for (int64_t i = 0; i < e->ncomb; ++i) e->order[i] = (int32_t)i;
    uint64_t s = e->seed;
    for (int64_t i = e->ncomb - 1; i > 0; --i) {
        int64_t j = (int64_t)(splitmix64(s) % (uint64_t)(i + 1));
        std::swap(e->order[i], e->order[j]);
    }
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

}  // namespace fsk_detail

// =============================================================================================
extern "C" {

int fsk_abi_version(void) { return FSK_ABI_VERSION; }

int fsk_seed_order(uint64_t seed, int64_t n, int32_t* out) {
    if (n < 0 || n > 0x7fffffff || (n > 0 && !out)) return FSK_EINVAL;
    fsk_detail::libstdcxx_shuffle_order(seed, n, out);
    return FSK_OK;
}

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
    if (cfg->profile < 0 || cfg->profile > 2) { g_create_error = "profile must be 0, 1 or 2"; return FSK_EINVAL; }
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
    // the one place the library reads the environment: FSK_TUNING="key=value,key=value" (see FSK_TUNING_KEYS)
    if (const char* env = getenv("FSK_TUNING")) {
        std::string why;
        if (tuning_parse(e->tune, env, why)) {
            delete e;
            g_create_error = "FSK_TUNING: " + why;
            return FSK_EINVAL;
        }
    }
    e->cfg_profile0 = e->cfg.profile;  // (what tuning key profile = -1 goes back to)
    if (e->tune.profile >= 0) e->cfg.profile = (int)e->tune.profile;
    if (e->trace()) fprintf(stderr, "[fsk] tuning: %s\n", tuning_in_force(e->tune).c_str());
    if (hipStreamCreate(&e->stream) != hipSuccess || hipEventCreate(&e->ev0) != hipSuccess ||
        hipEventCreate(&e->ev1) != hipSuccess) {
        delete e;
        g_create_error = "cannot create HIP stream/events";
        return FSK_EDEVICE;
    }
#ifndef FSK_EMU
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device) == hipSuccess && cus > 0) e->n_cu = cus;
    }
#else
    e->n_cu = 2;  // (the emulation: a few persistent workgroups, so that chunks of several tiles and several bands occur)
#endif
    e->st.n_combos_total = (int32_t)e->ncomb;
    *out = e;
    return FSK_OK;
}

void fsk_destroy(fsk_engine* e) {
    if (!e) return;
    if (e->group) { group_destroy(e); return; }
    one_destroy(e);
}

}  // extern "C"

void fsk_detail::one_destroy(fsk_engine* e) {
    DeviceScope on_device(e->cfg.device);
    (void)hipStreamSynchronize(e->stream);
    if (e->lane_stream) (void)hipStreamSynchronize(e->lane_stream);
    if (e->chain_stream) (void)hipStreamSynchronize(e->chain_stream);  // (variance mode may leave a dropped batch's sums running)
    e->d_words.release(); e->d_wstart.release(); e->d_len.release(); e->d_fstart.release(); e->d_featseq.release(); e->d_win.release();
    e->d_pos.release(); e->d_allpos.release(); e->d_bsum.release(); e->d_seqblk.release(); e->K_store.release(); e->d_Kf64.release(); e->d_Khat.release(); e->d_prod.release();
    e->d_diag.release(); e->d_stage.release(); e->d_stage_u64.release(); e->d_Kslots.release(); e->d_cell_idx.release(); e->d_C4.release(); e->d_C4H.release(); e->d_rowmask.release(); e->d_flag.release(); e->d_tiletab.release(); e->d_rare.release(); e->d_rare_n.release(); e->d_common.release(); e->d_keybits.release(); e->d_lut.release(); e->d_vc.release();
    if (e->lane_stream) { (void)hipStreamSynchronize(e->lane_stream); (void)hipStreamDestroy(e->lane_stream); }
    e->harvest_times();
    if (e->lazy_open) (void)hipEventDestroy(e->lazy_open);
    for (hipEvent_t ev : e->lazy_free) (void)hipEventDestroy(ev);
    for (auto& lane : e->sxs) lane.release();
    e->d_owner_r0.release(); e->d_U.release(); e->d_U2.release(); e->d_stage32.release(); e->d_blk_r0.release();
    if (e->h_prod) (void)hipHostFree(e->h_prod);
    if (e->h_sx_pos) (void)hipHostFree(e->h_sx_pos);
    if (e->h_sx_stat) (void)hipHostFree(e->h_sx_stat);
    if (e->h_sx_head_pos) (void)hipHostFree(e->h_sx_head_pos);
    if (e->h_sx_head_stat) (void)hipHostFree(e->h_sx_head_stat);
    if (e->h_sx_head_flag) (void)hipHostFree(e->h_sx_head_flag);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    for (auto& ev : e->ev_lane) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->chain_stream) { (void)hipStreamSynchronize(e->chain_stream); (void)hipStreamDestroy(e->chain_stream); }
    (void)hipEventDestroy(e->ev0);
    (void)hipEventDestroy(e->ev1);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" {

int fsk_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n) {
    if (!e) return FSK_EINVAL;
    return e->group ? group_set_combo_order(e, order, n) : one_set_combo_order(e, order, n);
}

}  // extern "C"

int fsk_detail::one_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n) {
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

extern "C" {

int fsk_set_seed(fsk_engine* e, uint64_t seed) {
    if (!e) return FSK_EINVAL;
    if (e->group) return group_set_seed(e, seed);
    e->seed = seed;
    return FSK_OK;
}

int fsk_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    if (!e) return FSK_EINVAL;
    return e->group ? group_load_sequences(e, tokens, offsets, n_train, n_test) : one_load_sequences(e, tokens, offsets, n_train, n_test);
}

}  // extern "C"

int fsk_detail::one_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
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
    const bool trace_load = e->trace();  // stderr: where the host time of the load goes
    const auto tl0 = std::chrono::steady_clock::now();
    auto tl_ms = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double tl_lengths = 0, tl_pack = 0;
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
    tl_lengths = tl_ms(tl0);
    // ---- alphabet: rank-remap the tokens that occur (equality preserving; the reference's
    // dict_size = |{0} U tokens|, fastsk.cpp:70-85, only serves as its counting-sort radix), then pack,
    // every sequence word-aligned.
    const int64_t total = offsets[N] - off0;
    std::vector<int32_t> distinct;
    std::vector<int64_t> sym_freq(256, 0);
    uint32_t sigma = 1;
    int bits = 2, key_symbits = 0;
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
        // (cntsrtna's radix is the dictionary size, whatever it is — shared.cpp:156-191 — and its keys are tuples, never a
        // number: up to 65536 symbols travel as 16-bit fields, and a k-mer space beyond 2^62 is keyed by the symbols' bit
        // fields side by side instead of a mixed-radix number, see key_symbits)
        if (distinct.size() > 65536) return e->fail(FSK_EUNSUPPORTED, "alphabet of %zu symbols (> 65536)", distinct.size());
        sigma = (uint32_t)std::max<size_t>(1, distinct.size());
        bits = sigma <= 4 ? 2 : sigma <= 16 ? 4 : sigma <= 256 ? 8 : 16;
        V = 1;
        key_symbits = 0;
        for (int c = 0; c < e->k; ++c) {
            if (V > (((u64)1 << 62) / sigma)) { key_symbits = 1; break; }
            V *= sigma;
        }
        if (key_symbits) {
            while (((u64)1 << key_symbits) < sigma) ++key_symbits;
            V = (u64)1 << 62;  // (at least: the dense dataflow and the 24-bit fast paths are out of the question)
            if ((int64_t)key_symbits * e->k > 96)
                return e->fail(FSK_EUNSUPPORTED, "a k-mer of %d symbols of %d bits does not fit the 128-bit sort records", e->k, key_symbits);
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
                // (four interleaved sets of counters: neighbouring tokens are often equal — DNA has four symbols —, and one
                // counter incremented twice in a row waits for its own store: ~1.1 -> ~0.5 ns a token)
                uint32_t h4[4][256];
                memset(h4, 0, sizeof h4);
                uint32_t big = 0;
                int64_t i = lo;
                for (; i + 4 <= hi; i += 4) {
                    const uint32_t v0 = (uint32_t)tokens[i], v1 = (uint32_t)tokens[i + 1], v2 = (uint32_t)tokens[i + 2], v3 = (uint32_t)tokens[i + 3];
                    big |= v0 | v1 | v2 | v3;
                    h4[0][v0 & 255u]++; h4[1][v1 & 255u]++; h4[2][v2 & 255u]++; h4[3][v3 & 255u]++;
                }
                for (; i < hi; ++i) { const uint32_t v = (uint32_t)tokens[i]; big |= v; h4[0][v & 255u]++; }
                if (big >= 256u) small[(size_t)t] = 0;  // (a token id beyond 255 somewhere: the general path below)
                for (int v = 0; v < 256; ++v) h[(size_t)v] = (int64_t)h4[0][v] + h4[1][v] + h4[2][v] + h4[3][v];
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
                for (int64_t i = bound[(size_t)t]; i < bound[(size_t)t + 1]; ++i) {  // a sequence's words are its own
                    const int32_t* sq = tokens + (offsets[i] - off0);
                    uint32_t* w = st_words + st_wstart[i];
                    // (the symbol width as a compile-time constant: the 16 / 8 / 4 symbols of a word unroll)
                    if (bits == 2) pack_sequence<2>(sq, len32[i], lut, w);
                    else if (bits == 4) pack_sequence<4>(sq, len32[i], lut, w);
                    else pack_sequence<8>(sq, len32[i], lut, w);
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
        if (distinct.size() > sym_freq.size()) sym_freq.assign(distinct.size(), 0);
        wstart_v.resize((size_t)N);
        uint64_t nwords = 0;
        int rc_plan = plan_words(wstart_v.data(), &nwords);
        if (rc_plan) return rc_plan;
        words_v.assign((size_t)nwords + 4, 0u);
        const int32_t base = distinct.empty() ? 0 : distinct.front();
        const bool direct = !distinct.empty() && (int64_t)distinct.back() - base < (1 << 20);
        std::vector<uint16_t> lut;
        if (direct) {
            lut.assign((size_t)(distinct.back() - base) + 1, 0);
            for (size_t r = 0; r < distinct.size(); ++r) lut[(size_t)(distinct[r] - base)] = (uint16_t)r;
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
    tl_pack = tl_ms(tl0);
    // ---- commit
    // (the dense dataflow's tile table depends on the number of sequences and the train / test split alone: a set of the same
    // shape keeps the one on the device)
    if (e->N != N || e->n_train != n_train) e->tab_n = 0;
    e->N = N; e->n_train = n_train; e->n_test = n_test; e->nfeat = nfeat;
    e->pairs = N * (N + 1) / 2;
    e->sigma = sigma; e->bits = bits; e->V = V; e->Vq = (uint32_t)((V + 3) / 4);
    e->Lmax = (uint32_t)longest; e->Lmin = (uint32_t)std::min(shortest_train, n_test > 0 ? shortest_test : shortest_train);
    e->maxW = (uint32_t)(longest - g + 1);
    e->n_panels = (uint32_t)((N + fsk::PANEL - 1) / fsk::PANEL);
    e->h_len = len32; e->h_fstart = fstart; e->featseq_ready = false;
    e->prep_valid = false; e->vc_sum = 0; e->vc_n = 0;

    e->lazy_lo = e->lazy_hi = -1;  // (the triangle is zeroed, or promised to be, below)
    e->u_known = false; e->u_pending = false; e->u_extra = 0; e->u_value = 0;
    {
        // The update words per record seen on the previous set of sequences stay as a HINT when the new set has the same shape
        // (sequences, windows, alphabet, longest sequence: the same data loaded again, or its next fold): the first batch is then
        // enqueued under a guard like every later one instead of being capped at 2^25 records and waited for. A hint that is
        // too low costs what any overflowing guard costs — the batch leaves K alone and is redone sized exactly —, never a
        // result. Tuning sparse_hint = 0: every set of sequences starts from nothing (testing).
        const bool same = e->tune.sparse_hint && e->loaded && e->sx_shape[0] == (u64)N && e->sx_shape[1] == (u64)nfeat && e->sx_shape[2] == (u64)sigma &&
                          e->sx_shape[3] == (u64)longest;
        if (!same) { e->sx_wpr = 0; e->sx_ppr = 0; e->slots16_ok = true; }
        e->sx_shape[0] = (u64)N; e->sx_shape[1] = (u64)nfeat; e->sx_shape[2] = (u64)sigma; e->sx_shape[3] = (u64)longest;
    }
    for (auto& d : e->sx_defer) d.active = false;
    if (e->V > DENSE_MAX_KEYS) e->Vq = 1;  // unused on the sparse path
    {   // sparse dataflow: sort record = (k-mer << sx_sb) | sequence id; owner bands of K
        e->sx_sb = 1;
        while (((int64_t)1 << e->sx_sb) < N) ++e->sx_sb;
        e->sx_symbits = key_symbits;
        e->sx_keybits = 1;
        if (key_symbits) e->sx_keybits = key_symbits * e->k;
        else while (e->sx_keybits < 62 && ((u64)1 << e->sx_keybits) < V) ++e->sx_keybits;
        plan_owner_bands(e);
    }
    int rc = choose_path(e);
    if (rc) return rc;
    {   // key compaction pays when a symbol is rare (DNA with a few 'n'): most of the sigma^k key
        // space is then empty and need not be multiplied
        int64_t rarest = INT64_MAX;
        for (uint32_t r = 0; r < sigma; ++r) rarest = std::min(rarest, sym_freq[r]);
        e->compact = sigma >= 3 && V >= 64 && V <= 4096 && rarest * 50 < total;
        if (e->tune.compact >= 0) e->compact = e->tune.compact != 0 && V <= 4096;
        // ... and the keys that occur follow from the places of the rare symbols when those are few (g windows per place
        // and combo are marked instead of every window) and every key of common symbols can be taken as present (16
        // windows per such key at least: one that is missing only costs an empty panel row)
        e->rare_mask = 0; e->rare_places = 0; e->rare_ready = false;
        int64_t places = 0;
        uint32_t common = 0;
        for (uint32_t r = 0; r < sigma && r < 32u; ++r) {
            if (sym_freq[r] * 50 < total) { e->rare_mask |= 1u << r; places += sym_freq[r]; }
            else ++common;
        }
        double common_keys = 1;
        for (int c = 0; c < e->k; ++c) common_keys *= (double)common;
        e->compact_rare = e->compact && sigma <= 32 && common >= 1 && places > 0 && places < ((int64_t)1 << 24) &&
                          places * (int64_t)g * 8 < std::max<int64_t>(1, nfeat) && (double)nfeat >= 16.0 * common_keys;
        if (e->tune.compact_rare >= 0) e->compact_rare = e->compact && e->tune.compact_rare != 0 && sigma <= 32 && places < ((int64_t)1 << 24);
        e->rare_places = (uint32_t)places;
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
    FSK_HIP(e->d_U.reserve(2));  // [0] update count U (multi-chunk dense launches), [1] remainder row products of the dense tile launches
    FSK_HIP(hipMemsetAsync(e->d_U.p, 0, 2 * sizeof(u64), e->stream));
    e->loaded = true; e->finalized = false; e->result_f64 = false;
    e->stdevs.clear();
    fsk_stats& st = e->st;
    st = fsk_stats{};
    st.n_seq = N; st.n_train = n_train; st.n_test = n_test; st.n_feat = nfeat; st.n_pairs = e->pairs;
    st.alphabet = (int32_t)sigma; st.bits_per_symbol = bits; st.key_space = (int64_t)V; st.path_used = e->path;
    st.n_combos_total = (int32_t)e->ncomb;
    st.max_windows = (double)e->maxW;
    if (trace_load)
        fprintf(stderr, "[fsk] load: %lld tokens, lengths %.3f ms, alphabet + packing %.3f ms, plan + uploads enqueued %.3f ms\n", (long long)total,
                tl_lengths, tl_pack - tl_lengths, tl_ms(tl0) - tl_pack);
    return FSK_OK;
}

extern "C" {

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
    // a group narrows its exchange to int32 when combos-since-reset x max_windows^2 < 2^31 bounds every cell: the
    // caller's memory holds whatever it holds, so no narrowing until fsk_reset_counts / fsk_load_sequences zero it
    if (e->group) group_note_bound_counts(e);
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
    return e->group ? group_reset_counts(e, 0, -1) : one_reset_counts(e);
}

int fsk_reset_counts_rows(fsk_engine* e, int64_t row_begin, int64_t row_end) {
    if (!e) return FSK_EINVAL;
    if (e->group && (row_begin < 0 || row_end < row_begin)) return e->fail(FSK_EINVAL, "bad row range");
    return e->group ? group_reset_counts(e, row_begin, row_end) : one_reset_counts_rows(e, row_begin, row_end);
}

int fsk_accumulate(fsk_engine* e, const int32_t* combos, int32_t n) {
    if (!e) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (n < 0 || (n > 0 && !combos)) return e->fail(FSK_EINVAL, "bad combo list");
    if (e->group) return group_accumulate(e, combos, n);
    return one_accumulate_rows(e, combos, n, 0, e->N);
}

int fsk_accumulate_rows(fsk_engine* e, const int32_t* combos, int32_t n, int64_t row_begin, int64_t row_end) {
    if (!e) return FSK_EINVAL;
    if (e->group) return e->fail(FSK_ESTATE, "fsk_accumulate_rows is a single-engine call: a group bands its accumulate itself");
    return one_accumulate_rows(e, combos, n, row_begin, row_end);
}

int fsk_synchronize(fsk_engine* e) {
    if (!e) return FSK_EINVAL;
    return e->group ? group_synchronize(e) : one_synchronize(e);
}

// Ordering against a stream of the caller (torch's current stream, which RCCL collectives are
// ordered after) without blocking the host: an event recorded on one stream, waited for by the other
// (one event per direction: a wait captures the record that precedes it).
int fsk_stream_wait_engine(fsk_engine* e, void* hip_stream) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    if (!e->ev_out) FSK_HIP(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    FSK_HIP(hipEventRecord(e->ev_out, e->stream));
    FSK_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, e->ev_out, 0));
    return FSK_OK;
}

int fsk_engine_wait_stream(fsk_engine* e, void* hip_stream) {
    if (!e) return FSK_EINVAL;
    FSK_ON_DEVICE(e);
    if (!e->ev_in) FSK_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    FSK_HIP(hipEventRecord(e->ev_in, (hipStream_t)hip_stream));
    FSK_HIP(hipStreamWaitEvent(e->stream, e->ev_in, 0));
    return FSK_OK;
}

int fsk_finalize(fsk_engine* e) {
    if (!e) return FSK_EINVAL;
    return e->group ? group_finalize(e) : one_finalize(e);
}

int fsk_compute(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
    if (!e) return FSK_EINVAL;
    if (e->group) return group_compute(e, tokens, offsets, n_train, n_test);
    FSK_ON_DEVICE(e);
    int rc = one_load_sequences(e, tokens, offsets, n_train, n_test);
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
    if (c.skip_variance) {  // chains only select which combos enter the integer sum
        std::vector<int32_t> used;
        skip_variance_combos(e, used);
        rc = do_accumulate(e, used.data(), (int)used.size(), e->d_K);
        if (rc) return rc;
        return make_diag(e);
    }
    rc = run_variance_mode(e, approx_chains(e));
    if (rc) return rc;
    return make_diag(e);
}

int fsk_run_chains(fsk_engine* e, int32_t first, int32_t step) {
    if (!e) return FSK_EINVAL;
    if (e->group) return e->fail(FSK_ESTATE, "fsk_run_chains is a single-engine call: a group deals its chains itself (fsk_compute)");
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    const fsk_config& c = e->cfg;
    if (!c.approx || c.skip_variance) return e->fail(FSK_ESTATE, "fsk_run_chains is the variance mode (approx=1, skip_variance=0)");
    if (first < 0 || step < 1) return e->fail(FSK_EINVAL, "need first >= 0 and step >= 1");
    FSK_ON_DEVICE(e);
    if (!e->order_set) default_order(e);
    e->finalized = false;
    return run_variance_mode(e, approx_chains(e), first, step);
}

}  // extern "C"

namespace fsk_detail {

int approx_chains(const fsk_engine* e) {
    const int T = e->cfg.t == -1 ? 20 : e->cfg.t;  // fastsk_kernel.cpp:54-61
    return std::max(1, std::min<int>(T, (int)e->order.size()));
}

void skip_variance_combos(fsk_engine* e, std::vector<int32_t>& used) {
    const int T = approx_chains(e);
    used.clear();
    for (int tid = 0; tid < T; ++tid) {
        int iters = 0;
        for (size_t item = tid; item < e->order.size(); item += T) {
            used.push_back(e->order[item]);
            if (e->cfg.max_iters != -1 && ++iters >= e->cfg.max_iters) break;
        }
    }
}

int one_reset_counts(fsk_engine* e) {
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

int one_reset_counts_rows(fsk_engine* e, int64_t row_begin, int64_t row_end) {
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

int one_accumulate_rows(fsk_engine* e, const int32_t* combos, int32_t n, int64_t row_begin, int64_t row_end) {
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (n < 0 || (n > 0 && !combos)) return e->fail(FSK_EINVAL, "bad combo list");
    if (row_begin < 0 || row_end > e->N || row_begin > row_end || row_begin % fsk::TILE != 0 ||
        (row_end % fsk::TILE != 0 && row_end != e->N))
        return e->fail(FSK_EINVAL, "row band must be [a,b) with a, b multiples of %d (b may be N)", fsk::TILE);
    if (row_begin == row_end) return FSK_OK;
    FSK_ON_DEVICE(e);
    if (n == 0) {
        // Nothing to add — but rows that a reset left to a storing launch get their zeros now: the caller
        // may hand these rows to another stream next (fsk_stream_wait_engine), e.g. a rank of a job with
        // more ranks than combos, whose share of the all-reduce must be zeros, not the previous pass.
        return materialise_zero(e);
    }
    e->finalized = false;
    return do_accumulate(e, combos, n, e->d_K, row_begin, row_end);
}

int one_synchronize(fsk_engine* e) {
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    FSK_HIP(hipStreamSynchronize(e->stream));
    FSK_HIP(hipGetLastError());
    return FSK_OK;
}

int one_finalize(fsk_engine* e) {
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    return make_diag(e);
}

}  // namespace fsk_detail

extern "C" {

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

// cells [c0, c0 + cnt) of the normalised triangle into device memory at `dst` (dst[0] = cell c0)
static int launch_triangle(fsk_engine* e, u64 c0, u64 cnt, double* dst) {
    const u64 per_block = (u64)256 * fsk::TR_ITEMS;
    const uint32_t blocks = (uint32_t)((cnt + per_block - 1) / per_block);
    if (e->result_f64)
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_triangle<double>), dim3(blocks), dim3(256), 0, e->stream, e->d_Kf64.p, e->d_diag.p, c0, cnt, dst);
    else
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_triangle<u64>), dim3(blocks), dim3(256), 0, e->stream, e->d_K, e->d_diag.p, c0, cnt, dst);
    return FSK_OK;
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
        { int rc = launch_triangle(e, c0, cnt, e->d_stage.p); if (rc) return rc; }
        FSK_HIP(hipMemcpyAsync(out + c0, e->d_stage.p, cnt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
    }
    return FSK_OK;
}

int fsk_get_triangle_device(fsk_engine* e, double* device_out) {
    if (!e) return FSK_EINVAL;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel: call fsk_compute or fsk_finalize first");
    if (!device_out) return e->fail(FSK_EINVAL, "null output");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const u64 chunk = (u64)1 << 33;  // (grid.x stays far below its limit)
    for (u64 c0 = 0; c0 < (u64)e->pairs; c0 += chunk) {
        const int rc = launch_triangle(e, c0, std::min<u64>(chunk, (u64)e->pairs - c0), device_out + c0);
        if (rc) return rc;
    }
    FSK_HIP(hipStreamSynchronize(e->stream));
    return FSK_OK;
}

int fsk_alloc_triangle_device(fsk_engine* e, double** device_out) {
    if (!e || !device_out) return FSK_EINVAL;
    *device_out = nullptr;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel: call fsk_compute or fsk_finalize first");
    FSK_ON_DEVICE(e);
    double* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)e->pairs * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        return e->fail(FSK_ENOMEM, "cannot allocate the %lld-cell normalised triangle (%.1f GB) on device %d", (long long)e->pairs,
                       (double)e->pairs * 8e-9, e->cfg.device);
    }
    const int rc = fsk_get_triangle_device(e, p);
    if (rc) { (void)hipFree(p); return rc; }
    *device_out = p;
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
    return e->group ? group_get_stats(e, out) : one_get_stats(e, out);
}

int fsk_counts_digest(fsk_engine* e, int64_t row_begin, int64_t row_end, uint64_t out[2]) {
    if (!e || !out) return FSK_EINVAL;
    if (!e->loaded) return e->fail(FSK_ESTATE, "load sequences first");
    if (e->result_f64) return e->fail(FSK_ESTATE, "variance mode keeps a floating-point mean, not integer counts");
    if (row_begin < 0 || row_end > e->N || row_begin > row_end) return e->fail(FSK_EINVAL, "bad row range");
    FSK_ON_DEVICE(e);
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    const u64 c0 = (u64)row_begin * ((u64)row_begin + 1) / 2, c1 = (u64)row_end * ((u64)row_end + 1) / 2;
    FSK_HIP(e->d_stage_u64.reserve(2));
    FSK_HIP(hipMemsetAsync(e->d_stage_u64.p, 0, 2 * sizeof(u64), e->stream));
    if (c1 > c0) {
        const u64 per_block = (u64)256 * fsk::DG_ITEMS;
        const uint32_t blocks = (uint32_t)std::min<u64>((c1 - c0 + per_block - 1) / per_block, (u64)1 << 16);
        FSK_LAUNCH(fsk::k_digest, dim3(blocks), dim3(256), 0, e->stream, (const u64*)e->d_K, c0, c1 - c0, e->d_stage_u64.p);
    }
    FSK_HIP(hipMemcpyAsync(out, e->d_stage_u64.p, 2 * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
    FSK_HIP(hipStreamSynchronize(e->stream));
    return FSK_OK;
}

int fsk_alloc_block_device(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double** device_out) {
    if (!e || !device_out) return FSK_EINVAL;
    *device_out = nullptr;
    if (!e->finalized) return e->fail(FSK_ESTATE, "no finalized kernel: call fsk_compute or fsk_finalize first");
    if (i0 < 0 || j0 < 0 || i1 > e->N || j1 > e->N || i0 > i1 || j0 > j1) return e->fail(FSK_EINVAL, "block out of range");
    FSK_ON_DEVICE(e);
    const size_t cells = (size_t)(i1 - i0) * (size_t)(j1 - j0);
    double* p = nullptr;
    if (hipMalloc((void**)&p, std::max<size_t>(1, cells) * sizeof(double)) != hipSuccess)
        return e->fail(FSK_ENOMEM, "cannot allocate a %lld x %lld block on device %d", (long long)(i1 - i0), (long long)(j1 - j0), e->cfg.device);
    const int rc = fsk_get_block_device(e, i0, i1, j0, j1, p);
    if (rc) { (void)hipFree(p); return rc; }
    *device_out = p;
    return FSK_OK;
}

int fsk_free_device(fsk_engine* e, void* device_ptr) {
    if (!device_ptr) return FSK_OK;
    if (!e) return hipFree(device_ptr) == hipSuccess ? FSK_OK : FSK_EDEVICE;  // (the engine may be gone before its block)
    FSK_ON_DEVICE(e);
    FSK_HIP(hipFree(device_ptr));
    return FSK_OK;
}

}  // extern "C"

int fsk_detail::one_get_stats(fsk_engine* e, fsk_stats* out) {
    u64 rem_rows = 0;
    if (e->d_U.p && e->loaded) {
        DeviceScope on_device(e->cfg.device);
        u64 U[2] = {0, 0};
        e->harvest_times();
        if (hipStreamSynchronize(e->stream) == hipSuccess && fetch_pending_u(e) == FSK_OK &&
            hipMemcpy(U, e->d_U.p, sizeof U, hipMemcpyDeviceToHost) == hipSuccess) {
            e->st.cell_updates = U[0] + e->u_extra;
            rem_rows = U[1];
        }
    }
    e->st.batches_redone = (double)e->sx_redone;
    e->st.sparse_form = (double)e->sx_form_used;
    e->st.sparse_passes = (double)e->sx_passes;
    e->st.sparse_desc = e->sx_desc_used ? 1.0 : 0.0;
    e->st.share_positions = (double)e->sx_share_used;
    e->st.share_groups = (double)e->sx_share_groups;
    *out = e->st;
    // (every flagged-row remainder product is one more dot8 per cell of its tile: 8 count-MACs x 128 x 128)
    out->dense_macs += rem_rows * (u64)fsk::TILE * fsk::TILE * 8;
    return FSK_OK;
}

extern "C" int fsk_set_skip_test_block(fsk_engine* e, int32_t skip) {
    if (!e) return FSK_EINVAL;
    if (e->group) return fsk_detail::group_set_skip_test_block(e, skip);
    e->cfg.skip_test_block = skip ? 1 : 0;
    e->tab_n = 0;  // (the tile table leaves out test x test tiles)
    return FSK_OK;
}

// ---- tuning through the ABI (include/fastsk_amd.h)
extern "C" int fsk_set_tuning(fsk_engine* e, const char* key, int64_t value) {
    if (!e) return FSK_EINVAL;
    std::string why;
    // (a group: every engine runs with the same tuning; the group's own keys are read from engine 0 by fsk_multi.hip)
    const int rc = e->group ? fsk_detail::group_set_tuning(e, key, value, why) : fsk_detail::tuning_set(e->tune, key, value, why);
    if (rc) return e->fail(rc, "%s", why.c_str());
    {   // the profile key: 0 .. 2 from here on, -1 = as the engine was created (a group: every engine)
        const int want = e->tune.profile >= 0 ? (int)e->tune.profile : e->cfg_profile0;
        if (e->group) fsk_detail::group_set_profile(e, want);
        else if (want != e->cfg.profile) { e->harvest_times(); e->cfg.profile = want; }
    }
    if (e->trace()) fprintf(stderr, "[fsk] tuning: %s\n", fsk_detail::tuning_in_force(e->tune).c_str());
    return FSK_OK;
}

extern "C" int fsk_get_tuning(fsk_engine* e, const char* key, int64_t* value) {
    if (!e || !key || !value) return FSK_EINVAL;
    for (const TuneKey& k : TUNE_KEYS)
        if (!strcmp(k.name, key)) { *value = e->tune.*(k.field); return FSK_OK; }
    return e->fail(FSK_EINVAL, "unknown tuning key '%s'", key);
}

extern "C" const char* fsk_tuning_keys(void) {
    static const std::string text = [] {
        std::string t;
        for (const TuneKey& k : TUNE_KEYS)
            t += std::string(k.name) + "=" + std::to_string(k.def) + " [" + std::to_string(k.lo) + ".." + std::to_string(k.hi) + "] " + k.doc + "\n";
        return t;
    }();
    return text.c_str();
}

#ifdef FSK_TEST_HOOKS
// ---- test builds only (tests/hooks, the CPU emulation; never the product): the hand-written wave primitives of fsk_gfx950.h,
// one call each per element, for a direct comparison with their definitions (tests/test_gpu_parity.py::test_wave_primitives) —
// the emulator replaces them with shuffle loops, so the kernels' own tests never run the inline asm on the CPU.
namespace fsk {
__global__ __launch_bounds__(256) void k_test_wave_ops(const int32_t* in, const uint32_t* aux, int32_t n, int32_t* out_max, uint32_t* out_sum,
                                                       u64* out_sum64, uint32_t* out_xnor, int32_t* out_sbfe, uint32_t* out_mbcnt) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    const int32_t v = i < n ? in[i] : 0;  // (whole waves call the primitives: n is a multiple of 64)
    const uint32_t a = i < n ? aux[i] : 0u;
    const int32_t mx = fsk_hw::wave_incl_max_i32(v);
    const uint32_t sm = fsk_hw::wave_incl_sum_u32((uint32_t)v & 0xffffu);
    const u64 s64 = fsk_hw::wave_incl_sum_u64(((u64)a << 20) | ((uint32_t)v & 0xfffffu));
    const uint32_t xn = fsk_hw::and_xnor((uint32_t)v, a, a * 2654435761u);
    const int32_t sb = fsk_hw::sbfe1((uint32_t)v, (int)(a & 31u));
    const unsigned long long bal = __ballot((a & 1u) != 0u);
    const uint32_t mb = fsk_mbcnt((uint32_t)bal, (uint32_t)(bal >> 32));
    if (i < n) { out_max[i] = mx; out_sum[i] = sm; out_sum64[i] = s64; out_xnor[i] = xn; out_sbfe[i] = sb; out_mbcnt[i] = mb; }
}
}  // namespace fsk
extern "C" int fsk_test_wave_ops(const int32_t* in, const uint32_t* aux, int32_t n, int32_t* out_max, uint32_t* out_sum, unsigned long long* out_sum64,
                                 uint32_t* out_xnor, int32_t* out_sbfe, uint32_t* out_mbcnt) {
    if (n <= 0 || n % 64) return FSK_EINVAL;
    void* d[8] = {nullptr};
    const size_t sz[8] = {4, 4, 4, 4, 8, 4, 4, 4};
    bool ok = true;
    for (int q = 0; q < 8 && ok; ++q) ok = hipMalloc(&d[q], sz[q] * (size_t)n) == hipSuccess;
    if (ok) ok = hipMemcpy(d[0], in, 4 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d[1], aux, 4 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        FSK_LAUNCH(fsk::k_test_wave_ops, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t) nullptr, (const int32_t*)d[0], (const uint32_t*)d[1], n,
                   (int32_t*)d[2], (uint32_t*)d[3], (u64*)d[4], (uint32_t*)d[5], (int32_t*)d[6], (uint32_t*)d[7]);
        ok = hipDeviceSynchronize() == hipSuccess;
    }
    void* host[8] = {nullptr, nullptr, out_max, out_sum, out_sum64, out_xnor, out_sbfe, out_mbcnt};
    for (int q = 2; q < 8 && ok; ++q) ok = hipMemcpy(host[q], d[q], sz[q] * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    for (int q = 0; q < 8; ++q) if (d[q]) (void)hipFree(d[q]);
    return ok ? FSK_OK : FSK_EDEVICE;
}
#endif

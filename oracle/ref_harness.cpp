// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never imported by the product path).
//
// Thin extern "C" harness around the *compiled reference* (QData/FastSK). The reference's own
// translation units are compiled where they lie under /root/reference by oracle/Makefile and
// linked with this file into oracle/_ref/libfastsk_ref.so. Nothing from the reference is copied
// into this repository: this file only #includes its headers by path and calls its functions.
//
// What it exposes:
//   ref_shuffle_order     the combo order fastsk_kernel.cpp:29-38 produces for a given seed
//   ref_compute           FastSK::compute_kernel / compute_train end to end (fastsk.cpp:30-188)
//                         with std::time() pinned (fastsk_kernel.cpp:37 seeds the shuffle with it)
//   ref_full_triangle     KernelFunction::compute_kernel (fastsk_kernel.cpp:24-106) returning the
//                         whole normalised triangle incl. the never-exposed test x test block
//   ref_raw_counts        replay of fastsk_kernel.cpp:216-241 per combo with the reference's own
//                         extractFeatures / getCombinations / cntsrtna / countAndUpdateTri,
//                         summing the uint32 partial kernels (the API itself only returns fp64);
//                         T threads round-robin (fastsk_kernel.cpp:148,275), returns the seconds
//   ref_save_kernel       FastSK::compute_kernel / compute_train followed by FastSK::save_kernel
//                         (fastsk.cpp:223-237): the reference's own text dump of the whole N x N matrix
// (bench.py's cpu_baseline times ref_full_triangle: the reference's own thread pool and reduce)
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <random>
#include <thread>
#include <vector>
#include <unistd.h>
#include <fcntl.h>

#include "fastsk.hpp"         // -I /root/reference/src/fastsk/_fastsk
#include "fastsk_kernel.hpp"
#include "shared.h"

// ---- pinned clock -------------------------------------------------------------------------
// The library is linked with -Wl,-Bsymbolic so the reference's call to std::time(0) binds to
// this definition: approx-mode sampling becomes a function of `g_fake_time` alone.
static time_t g_fake_time = 0;
extern "C" time_t time(time_t* t) {
    if (t) *t = g_fake_time;
    return g_fake_time;
}

namespace {
struct Quiet {  // the reference prints progress unconditionally; silence it while we call it
    int saved = -1;
    explicit Quiet(bool on) {
        if (!on) return;
        fflush(stdout);
        saved = dup(1);
        int devnull = open("/dev/null", O_WRONLY);
        dup2(devnull, 1);
        close(devnull);
    }
    ~Quiet() {
        if (saved < 0) return;
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};

std::vector<std::vector<int>> rows(const int32_t* tokens, const int64_t* offsets, int64_t lo, int64_t hi) {
    std::vector<std::vector<int>> X;
    X.reserve(hi - lo);
    for (int64_t i = lo; i < hi; ++i) X.emplace_back(tokens + offsets[i], tokens + offsets[i + 1]);
    return X;
}

int dict_size_of(const std::vector<std::vector<int>>& X) {  // fastsk.cpp:70-85
    std::vector<int> v{0};
    for (auto& r : X) v.insert(v.end(), r.begin(), r.end());
    std::sort(v.begin(), v.end());
    return (int)(std::unique(v.begin(), v.end()) - v.begin());
}
}  // namespace

extern "C" {

void ref_shuffle_order(long seed, int n, int32_t* out) {
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    auto rng = std::default_random_engine{};
    rng.seed(seed);
    std::shuffle(idx.begin(), idx.end(), rng);
    for (int i = 0; i < n; ++i) out[i] = idx[i];
}

// End-to-end through the reference's FastSK class. train_out: n_train*n_train, test_out:
// n_test*n_train (row-major), stdevs_out: capacity stdev_cap. Returns number of stdevs.
int ref_compute(const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test,
                int g, int m, int t, int approx, double delta, int max_iters, int skip_variance,
                long seed, double* train_out, double* test_out, double* stdevs_out, int stdev_cap,
                int quiet) {
    g_fake_time = (time_t)seed;
    Quiet q(quiet != 0);
    FastSK fsk(g, m, t, approx != 0, delta, max_iters, skip_variance != 0);
    auto Xtr = rows(tokens, offsets, 0, n_train);
    if (n_test > 0) {
        auto Xte = rows(tokens, offsets, n_train, n_train + n_test);
        fsk.compute_kernel(Xtr, Xte);
    } else {
        fsk.compute_train(Xtr);
    }
    auto Ktr = fsk.get_train_kernel();
    for (int64_t i = 0; i < n_train; ++i)
        memcpy(train_out + i * n_train, Ktr[i].data(), n_train * sizeof(double));
    if (n_test > 0) {
        auto Kte = fsk.get_test_kernel();
        for (int64_t i = 0; i < n_test; ++i)
            memcpy(test_out + i * n_train, Kte[i].data(), n_train * sizeof(double));
    }
    auto sd = fsk.get_stdevs();
    int n = (int)sd.size();
    for (int i = 0; i < n && i < stdev_cap; ++i) stdevs_out[i] = sd[i];
    return n;
}

// The reference's own on-disk format: compute, then FastSK::save_kernel(path) (fastsk.cpp:223-237).
int ref_save_kernel(const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test,
                    int g, int m, int t, int approx, double delta, int max_iters, int skip_variance,
                    long seed, const char* path, int quiet) {
    g_fake_time = (time_t)seed;
    Quiet q(quiet != 0);
    FastSK fsk(g, m, t, approx != 0, delta, max_iters, skip_variance != 0);
    auto Xtr = rows(tokens, offsets, 0, n_train);
    if (n_test > 0) {
        auto Xte = rows(tokens, offsets, n_train, n_train + n_test);
        fsk.compute_kernel(Xtr, Xte);
    } else {
        fsk.compute_train(Xtr);
    }
    fsk.save_kernel(std::string(path));
    return 0;
}

// Whole normalised triangle via the reference engine (KernelFunction). tri_out: N(N+1)/2.
int ref_full_triangle(const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test,
                      int g, int m, int t, int approx, double delta, int max_iters,
                      int skip_variance, long seed, double* tri_out, double* stdevs_out,
                      int stdev_cap, int quiet) {
    g_fake_time = (time_t)seed;
    Quiet q(quiet != 0);
    int64_t N = n_train + n_test;
    auto X = rows(tokens, offsets, 0, N);
    std::vector<int> lengths;
    int** S = (int**)malloc(N * sizeof(int*));
    for (int64_t i = 0; i < N; ++i) { S[i] = X[i].data(); lengths.push_back((int)X[i].size()); }
    Features* features = extractFeatures(S, lengths, (int)N, g);
    kernel_params params;
    params.g = g; params.k = g - m; params.m = m;
    params.n_str_train = n_train; params.n_str_test = n_test; params.total_str = N;
    params.n_str_pairs = (N / (double)2) * (N + 1);
    params.features = features;
    params.dict_size = dict_size_of(X);
    params.num_threads = t; params.num_mutex = -1;
    params.quiet = true; params.approx = approx != 0; params.delta = delta;
    params.max_iters = max_iters; params.skip_variance = skip_variance != 0;
    KernelFunction kf(&params);
    double* K = kf.compute_kernel();
    memcpy(tri_out, K, params.n_str_pairs * sizeof(double));
    int n = (int)kf.stdevs.size();
    for (int i = 0; i < n && i < stdev_cap; ++i) stdevs_out[i] = kf.stdevs[i];
    free(K);
    free(S);
    return n;
}

// One worker: replay of the per-combo body (fastsk_kernel.cpp:216-241) with the reference's
// primitives, into a private uint32 triangle.
static void replay(const Features* F, int g, int k, int dict_size, int N, const int32_t* combos,
                   int n_combos, int tid, int T, unsigned int* Ks) {
    int nfeat = F->n;
    int* feat = F->features;
    int num_comb = nchoosek(g, k);
    unsigned int* out = (unsigned int*)malloc((size_t)k * num_comb * sizeof(unsigned int));
    int* pos = (int*)calloc(nfeat > g ? nfeat : g, sizeof(int));
    unsigned int cnt_comb[2] = {0, 0};
    getCombinations(g, k, pos, 0, 0, cnt_comb, out, num_comb);
    unsigned int* sortIdx = (unsigned int*)malloc((size_t)nfeat * sizeof(unsigned int));
    unsigned int* features_srt = (unsigned int*)malloc((size_t)nfeat * k * sizeof(unsigned int));
    unsigned int* group_srt = (unsigned int*)malloc((size_t)nfeat * sizeof(unsigned int));
    unsigned int* feat1 = (unsigned int*)malloc((size_t)nfeat * k * sizeof(unsigned int));
    for (int it = tid; it < n_combos; it += T) {
        int combo = combos[it];
        for (int j1 = 0; j1 < nfeat; ++j1)
            for (int j2 = 0; j2 < k; ++j2)
                feat1[j1 + (size_t)j2 * nfeat] = feat[j1 + (size_t)out[combo + j2 * num_comb] * nfeat];
        cntsrtna(sortIdx, feat1, k, nfeat, dict_size);
        for (int j1 = 0; j1 < nfeat; ++j1) {
            for (int j2 = 0; j2 < k; ++j2)
                features_srt[j1 + (size_t)j2 * nfeat] = feat1[sortIdx[j1] + (size_t)j2 * nfeat];
            group_srt[j1] = F->group[sortIdx[j1]];
        }
        countAndUpdateTri(Ks, features_srt, group_srt, k, nfeat, N);
    }
    free(out); free(pos); free(sortIdx); free(features_srt); free(group_srt); free(feat1);
}

// Raw integer counts summed over `combos`, T host threads. counts_out: uint64[N(N+1)/2].
// Returns seconds spent in the per-combo workers + reduce (feature extraction excluded).
double ref_raw_counts(const int32_t* tokens, const int64_t* offsets, int64_t N, int g, int m,
                      const int32_t* combos, int n_combos, int T, uint64_t* counts_out) {
    auto X = rows(tokens, offsets, 0, N);
    std::vector<int> lengths;
    int** S = (int**)malloc(N * sizeof(int*));
    for (int64_t i = 0; i < N; ++i) { S[i] = X[i].data(); lengths.push_back((int)X[i].size()); }
    Features* F = extractFeatures(S, lengths, (int)N, g);
    int dict_size = dict_size_of(X);
    int64_t pairs = N * (N + 1) / 2;
    if (T < 1) T = 1;
    if (T > n_combos) T = n_combos > 0 ? n_combos : 1;
    std::vector<unsigned int*> Ks(T);
    for (int t = 0; t < T; ++t) Ks[t] = (unsigned int*)calloc(pairs, sizeof(unsigned int));
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back(replay, F, g, g - m, dict_size, (int)N, combos, n_combos, t, T, Ks[t]);
    for (auto& x : th) x.join();
    if (counts_out) {
        memset(counts_out, 0, pairs * sizeof(uint64_t));
        for (int t = 0; t < T; ++t)
            for (int64_t i = 0; i < pairs; ++i) counts_out[i] += Ks[t][i];
    }
    auto t1 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; ++t) free(Ks[t]);
    free(F->features); free(F->group); free(F);
    free(S);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"

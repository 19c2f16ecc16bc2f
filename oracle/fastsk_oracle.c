/*
 * oracle/fastsk_oracle.c — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the
 * product path (fastsk_amd); used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
 *
 * A plain-C restatement of the reference's gapped-k-mer kernel path (QData/FastSK, paths relative
 * to /root/reference/src/fastsk/_fastsk). It follows the reference's algorithm step by step —
 * column-major g-mer table, lexicographic kept-position combinations, stable LSD counting sort,
 * run-length co-occurrence count into a uint32 lower triangle, round-robin combos over T workers,
 * Welford variance chain, two-pass cosine normalisation — but is written from scratch.
 *
 * Parity status: PINNED. tests/test_oracle.py checks this file bit-for-bit against the compiled
 * reference (oracle/_ref/libfastsk_ref.so, when present) and against tests/golden/*.npz, which
 * tests/make_golden.py generated from that compiled reference.
 *
 * Deviations from the reference, all outside its defined behaviour:
 *   - tokens outside [0, dict_size) are rank-remapped first (the reference indexes a
 *     dict_size-long histogram with the raw token, shared.cpp:170-173: undefined there);
 *   - the triangle index is computed in 64 bits (shared.cpp:97-117 uses int: overflows N>46340);
 *   - approx/variance mode with T>1 adds the per-worker means in worker order 0..T-1 (the
 *     reference adds them in thread-completion order, fastsk_kernel.cpp:286-315);
 *   - the combo order is an explicit argument (the reference shuffles with a time(0) seed,
 *     fastsk_kernel.cpp:36-38).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int32_t *feat;  /* column-major: feat[c + p*nfeat] = symbol p of g-mer c */
    int32_t *group; /* owning sequence of g-mer c */
    int64_t nfeat;
    int dict_size;
} gmers_t;

static inline int64_t tri(int64_t i, int64_t j) { /* shared.cpp:97-117, 64-bit */
    if (j > i) { int64_t t = i; i = j; j = t; }
    return i * (i + 1) / 2 + j;
}

/* nchoosek, shared.cpp:335-345 (int accumulation, multiply before divide) */
int64_t orc_num_combos(int g, int m) {
    unsigned n = (unsigned)g, k = (unsigned)m;
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int result = (int)n;
    for (unsigned i = 2; i <= k; ++i) {
        result *= (int)(n - i + 1);
        result /= (int)i;
    }
    return result;
}

/* c-th k-subset of {0..g-1} in lexicographic order == row c of getCombinations' table
 * (shared.cpp:347-360: depth-first, ascending positions). Returns 0 on success. */
int orc_combo_positions(int g, int k, int64_t combo, int32_t *out) {
    int64_t total = orc_num_combos(g, k);
    if (combo < 0 || combo >= total) return -1;
    int next = 0;
    for (int d = 0; d < k; ++d) {
        for (int p = next; p < g; ++p) {
            /* subsets that put p at depth d: choose the remaining k-d-1 from g-p-1 */
            int64_t below = orc_num_combos(g - p - 1, k - d - 1);
            if (k - d - 1 > g - p - 1) below = 0;
            if (combo < below) { out[d] = p; next = p + 1; break; }
            combo -= below;
        }
    }
    return 0;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* extractFeatures, shared.cpp:55-91 + dict_size, fastsk.cpp:70-85 */
static int build_gmers(const int32_t *tokens, const int64_t *offsets, int64_t N, int g, gmers_t *G) {
    int64_t total = offsets[N], nfeat = 0;
    for (int64_t i = 0; i < N; ++i) {
        int64_t len = offsets[i + 1] - offsets[i];
        if (len < g) return -2; /* reference exit(1)s here, fastsk.cpp:53-58 */
        nfeat += len - g + 1;
    }
    /* distinct values of {0} U tokens */
    int32_t *sorted = (int32_t *)malloc((total + 1) * sizeof(int32_t));
    memcpy(sorted, tokens, total * sizeof(int32_t));
    sorted[total] = 0;
    qsort(sorted, total + 1, sizeof(int32_t), cmp_i32);
    int64_t nd = 0;
    for (int64_t i = 0; i <= total; ++i)
        if (i == 0 || sorted[i] != sorted[i - 1]) sorted[nd++] = sorted[i];
    int dict_size = (int)nd;
    int in_range = sorted[0] >= 0 && sorted[nd - 1] < dict_size;

    G->nfeat = nfeat;
    G->dict_size = dict_size;
    G->feat = (int32_t *)malloc((size_t)nfeat * g * sizeof(int32_t));
    G->group = (int32_t *)malloc((size_t)nfeat * sizeof(int32_t));
    int64_t c = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int32_t *s = tokens + offsets[i];
        int64_t len = offsets[i + 1] - offsets[i];
        for (int64_t j = 0; j + g <= len; ++j, ++c) {
            for (int p = 0; p < g; ++p) {
                int32_t v = s[j + p];
                if (!in_range) { /* rank remap (equality preserving) */
                    int32_t *hit = (int32_t *)bsearch(&v, sorted, nd, sizeof(int32_t), cmp_i32);
                    v = (int32_t)(hit - sorted);
                }
                G->feat[c + (int64_t)p * nfeat] = v;
            }
            G->group[c] = (int32_t)i;
        }
    }
    free(sorted);
    return 0;
}

/* cntsrtna, shared.cpp:156-191: stable LSD counting sort of the nfeat k-tuples, last kept
 * position first, radix = dict_size; yields the permutation idx. */
static void lsd_sort(uint32_t *idx, const uint32_t *cols, int k, int64_t n, int radix,
                     uint32_t *tmp, int32_t *digit) {
    int64_t *start = (int64_t *)malloc((size_t)radix * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    for (int p = k - 1; p >= 0; --p) {
        memset(start, 0, (size_t)radix * sizeof(int64_t));
        const uint32_t *col = cols + (int64_t)p * n;
        for (int64_t i = 0; i < n; ++i) { digit[i] = (int32_t)col[idx[i]]; start[digit[i]]++; }
        int64_t run = 0;
        for (int d = 0; d < radix; ++d) { int64_t c = start[d]; start[d] = run; run += c; }
        for (int64_t i = 0; i < n; ++i) tmp[start[digit[i]]++] = idx[i];
        memcpy(idx, tmp, (size_t)n * sizeof(uint32_t));
    }
    free(start);
}

/* countAndUpdateTri, shared.cpp:268-333. Ks += sum over runs of equal k-mers of the outer
 * product of per-sequence multiplicities (lower triangle). *U counts the `+=` issued. */
static void count_runs(uint32_t *Ks, const uint32_t *cols, const uint32_t *grp, int k, int64_t n,
                       int64_t N, int32_t *ucnt, int32_t *upd, uint64_t *U) {
    int64_t i = 0;
    while (i < n) {
        int64_t lo = i;
        for (++i; i < n; ++i) {
            int same = 1;
            for (int p = 0; p < k; ++p)
                if (cols[i + (int64_t)p * n] != cols[lo + (int64_t)p * n]) { same = 0; break; }
            if (!same) break;
        }
        int64_t hi = i; /* run is [lo, hi) */
        if (hi - lo > 1) {
            memset(ucnt, 0, (size_t)N * sizeof(int32_t));
            for (int64_t j = lo; j < hi; ++j) ucnt[grp[j]]++;
            int64_t cu = 0;
            for (int64_t s = 0; s < N; ++s) /* the O(N) sweep per run, shared.cpp:310-315 */
                if (ucnt[s] > 0) upd[cu++] = (int32_t)s;
            for (int64_t a = 0; a < cu; ++a)
                for (int64_t b = a; b < cu; ++b)
                    Ks[tri(upd[b], upd[a])] += (uint32_t)(ucnt[upd[a]] * ucnt[upd[b]]);
            *U += (uint64_t)(cu * (cu + 1) / 2);
        } else {
            Ks[tri(grp[lo], grp[lo])] += 1u;
            *U += 1;
        }
    }
}

typedef struct {
    uint32_t *gathered, *sorted_cols, *sorted_grp, *idx, *tmp;
    int32_t *digit, *ucnt, *upd, *pos;
} scratch_t;

static void scratch_init(scratch_t *s, int64_t nfeat, int k, int64_t N) {
    s->gathered = (uint32_t *)malloc((size_t)nfeat * k * sizeof(uint32_t));
    s->sorted_cols = (uint32_t *)malloc((size_t)nfeat * k * sizeof(uint32_t));
    s->sorted_grp = (uint32_t *)malloc((size_t)nfeat * sizeof(uint32_t));
    s->idx = (uint32_t *)malloc((size_t)nfeat * sizeof(uint32_t));
    s->tmp = (uint32_t *)malloc((size_t)nfeat * sizeof(uint32_t));
    s->digit = (int32_t *)malloc((size_t)nfeat * sizeof(int32_t));
    s->ucnt = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    s->upd = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    s->pos = (int32_t *)malloc(64 * sizeof(int32_t));
}
static void scratch_free(scratch_t *s) {
    free(s->gathered); free(s->sorted_cols); free(s->sorted_grp); free(s->idx); free(s->tmp);
    free(s->digit); free(s->ucnt); free(s->upd); free(s->pos);
}

/* One mismatch combination: the loop body of kernel_build_parallel, fastsk_kernel.cpp:216-241 */
static void one_combo(const gmers_t *G, int g, int k, int64_t N, int64_t combo, uint32_t *Ks,
                      scratch_t *s, uint64_t *U) {
    int64_t n = G->nfeat;
    orc_combo_positions(g, k, combo, s->pos);
    for (int p = 0; p < k; ++p) /* gather kept positions, :224-228 */
        for (int64_t c = 0; c < n; ++c)
            s->gathered[c + (int64_t)p * n] = (uint32_t)G->feat[c + (int64_t)s->pos[p] * n];
    lsd_sort(s->idx, s->gathered, k, n, G->dict_size, s->tmp, s->digit); /* :231 */
    for (int64_t c = 0; c < n; ++c) { /* permute, :233-238 */
        for (int p = 0; p < k; ++p)
            s->sorted_cols[c + (int64_t)p * n] = s->gathered[s->idx[c] + (int64_t)p * n];
        s->sorted_grp[c] = (uint32_t)G->group[s->idx[c]];
    }
    count_runs(Ks, s->sorted_cols, s->sorted_grp, k, n, N, s->ucnt, s->upd, U); /* :241 */
}

/* ------------------------------------------------------------------ raw integer counts */
typedef struct {
    const gmers_t *G;
    int g, k, tid, T;
    int64_t N, pairs;
    const int32_t *combos;
    int n_combos;
    uint32_t *Ks;
    uint64_t U;
} raw_job_t;

static void *raw_worker(void *arg) {
    raw_job_t *J = (raw_job_t *)arg;
    scratch_t s;
    scratch_init(&s, J->G->nfeat, J->k, J->N);
    for (int it = J->tid; it < J->n_combos; it += J->T) /* round robin, fastsk_kernel.cpp:148,275 */
        one_combo(J->G, J->g, J->k, J->N, J->combos[it], J->Ks, &s, &J->U);
    scratch_free(&s);
    return NULL;
}

/* Sum of the partial kernels of `combos` as a uint64 lower triangle (may be NULL: timing only).
 * T workers, each with a private uint32 triangle (fastsk_kernel.cpp:175-176). Returns seconds
 * in workers + reduce, or a negative error code. */
double orc_raw_counts(const int32_t *tokens, const int64_t *offsets, int64_t N, int g, int m,
                      const int32_t *combos, int n_combos, int T, uint64_t *counts_out,
                      uint64_t *U_out) {
    gmers_t G;
    int k = g - m;
    if (k <= 0 || k > 63) return -1.0;
    if (build_gmers(tokens, offsets, N, g, &G) != 0) return -2.0;
    int64_t pairs = N * (N + 1) / 2;
    if (T < 1) T = 1;
    if (T > n_combos) T = n_combos > 0 ? n_combos : 1;
    raw_job_t *jobs = (raw_job_t *)calloc((size_t)T, sizeof(raw_job_t));
    pthread_t *th = (pthread_t *)malloc((size_t)T * sizeof(pthread_t));
    struct timespec t0, t1;
    for (int t = 0; t < T; ++t) {
        jobs[t] = (raw_job_t){&G, g, k, t, T, N, pairs, combos, n_combos,
                              (uint32_t *)calloc((size_t)pairs, sizeof(uint32_t)), 0};
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < T; ++t) pthread_create(&th[t], NULL, raw_worker, &jobs[t]);
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    uint64_t U = 0;
    if (counts_out) memset(counts_out, 0, (size_t)pairs * sizeof(uint64_t));
    for (int t = 0; t < T; ++t) {
        if (counts_out)
            for (int64_t i = 0; i < pairs; ++i) counts_out[i] += jobs[t].Ks[i];
        U += jobs[t].U;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < T; ++t) free(jobs[t].Ks);
    if (U_out) *U_out = U;
    free(jobs); free(th); free(G.feat); free(G.group);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Normalisation, fastsk_kernel.cpp:96-103: off-diagonals first (raw diagonals), then diagonals. */
void orc_normalise(double *K, int64_t N) {
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < i; ++j)
            K[tri(i, j)] = K[tri(i, j)] / sqrt(K[tri(i, i)] * K[tri(j, j)]);
    for (int64_t i = 0; i < N; ++i)
        K[tri(i, i)] = K[tri(i, i)] / sqrt(K[tri(i, i)] * K[tri(i, i)]);
}

/* get_variance, fastsk_kernel.cpp:108-143 (the `variances`/max_variance bookkeeping only feeds a
 * value the reference discards; the returned average is what matters). */
static double welford_step(const uint32_t *Ks, double *K_hat, int64_t pairs, int64_t train_pairs,
                           int iter) {
    double avg = 0;
    int64_t count = 0;
    for (int64_t i = 0; i < pairs; ++i) {
        double delta = Ks[i] - K_hat[i];
        K_hat[i] += delta / iter;
        if (i < train_pairs) {
            double delta2 = Ks[i] - K_hat[i];
            avg += delta * delta2;
            count++;
        }
    }
    avg /= count;
    if (iter == 1) avg = 9999999;
    else avg /= iter - 1;
    return avg;
}

/* The whole path for an explicit combo order, exact and approx modes; workers run one after the
 * other (the result does not depend on the interleaving except as noted in the header).
 *   tri_out     double[N(N+1)/2] normalised kernel (reference layout)
 *   stdevs_out  worker 0's convergence trace (fastsk_kernel.cpp:248-250); returns its length
 *   iters_out   int32[T] iterations each worker ran (may be NULL)
 */
int orc_compute(const int32_t *tokens, const int64_t *offsets, int64_t n_train, int64_t n_test,
                int g, int m, int t, int approx, double delta, int max_iters, int skip_variance,
                const int32_t *order, int n_order, double *tri_out, double *stdevs_out,
                int stdev_cap, int32_t *iters_out) {
    int64_t N = n_train + n_test;
    int k = g - m;
    if (k <= 0 || k > 63) return -1;
    gmers_t G;
    if (build_gmers(tokens, offsets, N, g, &G) != 0) return -2;
    int64_t pairs = (int64_t)((N / (double)2) * (N + 1));                 /* fastsk.cpp:101 */
    int64_t train_pairs = (int64_t)((n_train / (double)2) * (n_train + 1)); /* kernel.cpp:170 */
    int T = t == -1 ? 20 : t;                                              /* kernel.cpp:54-61 */
    if (T > n_order) T = n_order;
    if (T < 1) T = 1;
    int variance = approx && !skip_variance;
    double *K = (double *)calloc((size_t)pairs, sizeof(double));
    uint32_t *Ks = (uint32_t *)malloc((size_t)pairs * sizeof(uint32_t));
    double *K_hat = variance ? (double *)malloc((size_t)pairs * sizeof(double)) : NULL;
    scratch_t s;
    scratch_init(&s, G.nfeat, k, N);
    int n_sd = 0;
    uint64_t U = 0;
    for (int tid = 0; tid < T; ++tid) {
        memset(Ks, 0, (size_t)pairs * sizeof(uint32_t));
        if (variance) memset(K_hat, 0, (size_t)pairs * sizeof(double));
        int iter = 1, working = 1, item = tid;
        while (working) {
            if (variance) memset(Ks, 0, (size_t)pairs * sizeof(uint32_t)); /* :192-194 */
            one_combo(&G, g, k, N, order[item], Ks, &s, &U);
            if (variance) {
                double sd = welford_step(Ks, K_hat, pairs, train_pairs, iter); /* :244 */
                sd = sqrt(sd / iter);                                          /* :247 */
                if (tid == 0) { if (n_sd < stdev_cap) stdevs_out[n_sd] = sd; n_sd++; }
                if (delta / sd > 1.96) working = 0;                            /* :251-254 */
            }
            if (approx && max_iters != -1 && iter >= max_iters) working = 0;   /* :257-262 */
            item += T;
            if (item >= n_order) working = 0;                                  /* :275-278 */
            iter++;
        }
        if (iters_out) iters_out[tid] = iter - 1;
        for (int64_t i = 0; i < pairs; ++i) { /* reduce, :286-315 */
            double val = variance ? K_hat[i] : (double)Ks[i];
            if (val != 0) K[i] += val;
        }
    }
    orc_normalise(K, N);
    memcpy(tri_out, K, (size_t)pairs * sizeof(double));
    scratch_free(&s);
    free(K); free(Ks); free(K_hat); free(G.feat); free(G.group);
    return n_sd;
}

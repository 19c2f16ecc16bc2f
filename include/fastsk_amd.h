/*
 * fastsk_amd.h — C ABI of the MI355X-native gapped-k-mer kernel engine (libfastsk_amd.so).
 *
 * This is the drop-in boundary for ONE path of QData/FastSK: the per-mismatch-combination
 * partial-kernel worker behind fastsk.FastSK(g,m,...).compute_kernel(Xtrain,Xtest). Every entry
 * point names the reference interface it replaces (paths relative to
 * /root/reference/src/fastsk/_fastsk). Plain pointers and sizes only: no C++ types, no torch
 * types, no exceptions across the boundary. All functions return FSK_OK (0) or a negative
 * FSK_E* code; fsk_last_error() holds the message. INTEGRATION.md shows the pybind11 binding a
 * reference maintainer would write against this header.
 *
 * Data conventions (same as the reference):
 *   - sequences arrive as one int32 token array + int64 offsets (n_train rows first, then
 *     n_test rows), i.e. the flattened form of the vector<vector<int>> arguments of
 *     FastSK::compute_kernel (fastsk.cpp:30); buffers are caller-owned and only read during
 *     the call;
 *   - the kernel triangle is row-major lower-triangular, cell (i,j), j<=i, at i(i+1)/2+j
 *     (tri_access, shared.cpp:97-117), N = n_train + n_test;
 *   - combination id c is the c-th (g-m)-subset of {0..g-1} in lexicographic order
 *     (getCombinations, shared.cpp:347-360).
 */
#ifndef FASTSK_AMD_H
#define FASTSK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSK_ABI_VERSION 5  /* 2: fsk_create_multi + fsk_config.collective/bands, fsk_counts_digest, device-block allocation;
                              3: fsk_get_triangle_device / fsk_alloc_triangle_device, fsk_config.deadline_ms;
                              4: fsk_set_tuning / fsk_get_tuning / fsk_tuning_keys (one FSK_TUNING variable instead of
                                 two dozen FSK_* switches);
                              5: fsk_seed_order; fsk_set_seed draws the reference's own std::shuffle order */

enum {
    FSK_OK = 0,
    FSK_EINVAL = -1,   /* bad argument (g<=m, m<0, null pointer, bad combo id, ...)              */
    FSK_ESHORT = -2,   /* g > shortest sequence: the reference printf+exit(1)s, fastsk.cpp:53-58 */
    FSK_ESTATE = -3,   /* call out of order (e.g. getter before compute)                          */
    FSK_EDEVICE = -4,  /* HIP runtime error / no device                                           */
    FSK_ENOMEM = -5,   /* device or host allocation failed                                        */
    FSK_EUNSUPPORTED = -6 /* alphabet > 65536 symbols, (g-m) * ceil(log2(alphabet)) > 96 bits, g > 255 ...   */
};

/* accumulate dataflow */
enum {
    FSK_PATH_AUTO = 0,
    FSK_PATH_DENSE = 1,  /* k_dense_count: per-sequence LDS counting sort -> 4-bit count panels (lo + 16*hi);
                            k_dense_tile_dma: 128 x 128 tiles of K, panels DMA'd into LDS, v_dot8_u32_u4
                            co-occurrence sums kept in registers over all combos of the launch, then one
                            64-bit store (first launch after a reset) or atomicAdd per cell            */
    FSK_PATH_SPARSE = 2  /* the reference's dataflow (cntsrtna + countAndUpdateTri) as streams: packed
                            (k-mer, seq) records -> LSD radix sort in LDS-staged passes -> run-length
                            entries -> one 32-bit update word per += binned by owner band of K -> the
                            band's words summed in LDS, one 64-bit add per touched cell; beyond a few
                            LDS rounds a band (N > ~9,000) K is owned in two levels (bands, then blocks of
                            2^14 cells); where runs are long an entry of many partners travels as one
                            16-byte descriptor that the owning workgroup expands itself; a 64-bit
                            atomicAdd per (run, pair) only on request or for sequences in the millions   */
};

/* how the engines of fsk_create_multi sum their partial triangles (fastsk_kernel.cpp:286-315) */
enum {
    FSK_COLL_AUTO = 0,   /* RCCL when the listed devices are distinct and librccl can be loaded, else P2P  */
    FSK_COLL_RCCL = 1,   /* ncclAllReduce over xGMI, one communicator rank per listed device               */
    FSK_COLL_P2P = 2     /* the engine's own reduce-scatter + all-gather kernels over peer mappings (all
                            devices live in this process); also what lets a device be listed twice, i.e.
                            the multi-device host code exercised on a single GPU                          */
};

typedef struct fsk_engine fsk_engine; /* opaque; replaces class FastSK + KernelFunction state */

/* Constructor arguments of FastSK (bindings.cpp:14-22, fastsk.cpp:19-28), plus placement. */
typedef struct fsk_config {
    int32_t g;             /* g-mer length                                                      */
    int32_t m;             /* mismatch positions; k = g - m kept positions (fastsk.cpp:22)      */
    int32_t t;             /* host threads in the reference; here: number of approx-mode chains
                              (-1 -> 20, capped at #combos: fastsk_kernel.cpp:54-61)            */
    int32_t approx;        /* 0 exact, 1 approx (fastsk_kernel.cpp:188-262)                     */
    double delta;          /* approx convergence threshold (default 0.025)                      */
    int32_t max_iters;     /* approx: max combos per chain, -1 = unlimited                      */
    int32_t skip_variance; /* approx: integer sums, no Welford chain                            */
    int32_t device;        /* HIP device ordinal                                                */
    int32_t path;          /* FSK_PATH_*                                                        */
    int32_t profile;       /* 1: measurement mode — every kernel family timed with HIP events that are waited for on
                              the spot, the exact update count U computed (fsk_get_stats); the sparse dataflow on
                              one stream, every batch sized exactly. 2: the product dataflow untouched — the events
                              are only recorded (on the stream the kernels run on), fsk_get_stats harvests them */
    int32_t skip_test_block; /* 1: cells with both sequences in the test set (other than the diagonal)
                                may be left at zero — no getter of the reference exposes them
                                (fastsk.cpp:190-217). Dense dataflow: whole tiles of such cells
                                are not computed; sparse dataflow: a test row pairs only with the
                                train sequences of its k-mer runs (and itself).                 */
    /* fsk_create_multi only (ignored by fsk_create): */
    int32_t collective;    /* FSK_COLL_*                                                          */
    int32_t bands;         /* row bands of the overlapped all-reduce; 0 = automatic               */
    int32_t deadline_ms;   /* fail fast: bound, in milliseconds, on every host-side wait of the multi-GPU exchange —
                              ncclCommInitAll, the engines' barriers, a band's all-reduce having run on the device.
                              A wait that exceeds it returns FSK_EDEVICE naming the stage and the band, the
                              communicator is aborted (ncclCommAbort) and the group is dead: every later call
                              returns FSK_EDEVICE with that first message. 0 = the tuning key deadline_ms
                              (120000 unless set); negative = no deadline                                     */
    int32_t reserved[1];
} fsk_config;

/* Measured and algorithmic quantities of the work done so far. */
typedef struct fsk_stats {
    int64_t n_seq, n_train, n_test, n_feat, n_pairs;
    int32_t alphabet;        /* distinct tokens actually present (rank-remapped 0..alphabet-1)  */
    int32_t bits_per_symbol; /* packed width in HBM                                             */
    int64_t key_space;       /* alphabet^k                                                      */
    int32_t path_used;       /* FSK_PATH_DENSE / FSK_PATH_SPARSE                                */
    int32_t n_combos_total;  /* C(g,m)                                                          */
    int64_t combos_done;     /* combos accumulated so far                                       */
    uint64_t cell_updates;   /* U = sum over runs d(d+1)/2, exact on both dataflows (dense: counted from the
                                count panels by k_dense_distinct when profile = 1, else 0)                */
    uint64_t sort_records;   /* records pushed through the radix sort                           */
    int32_t sort_passes;     /* 8-bit LSD passes per batch                                      */
    int32_t launches;        /* kernel launches in accumulate                                   */
    /* HIP-event milliseconds on the engine's stream (profile=1), summed over launches         */
    double ms_count;         /* dense: k-mer extraction + LDS counting sort -> count panels     */
    double ms_tile;          /* dense: tiled co-occurrence accumulate + flush                   */
    double ms_extract;       /* sparse: key extraction                                          */
    double ms_sort;          /* sparse: radix sort                                              */
    double ms_segment;       /* sparse: run/segment detection + compaction                      */
    double ms_pairs;         /* sparse: per-run pair atomics                                    */
    double ms_total;         /* whole accumulate calls                                          */
    int64_t n_tile_launches; /* launches of the tile kernel (for per-launch averages)           */
    uint64_t dense_macs;     /* count multiply-adds issued by the tile kernel: 8 per dword row and cell, the exact
                                remainder products of the rows with counts above 15 included (profile = 1)  */
    uint64_t panel_bytes;    /* bytes of count panels written (= read at least once)            */
    double u4_tile_launches; /* launches of the 4-bit tile kernel (v_dot8_u32_u4)                */
    double max_windows;      /* max over sequences of (length - g + 1): bounds a cell per combo  */
    double count_launches;   /* launches of the segment-count kernel (panel cache misses)        */
    double compact_keys_avg; /* key compaction on: mean keys per combo that really occur (else 0) */
    double batches_redone;   /* sparse: batches enqueued ahead of their word count that did not fit */
    double combos_issued;    /* combos whose kernels ran: combos_done + the iterations variance mode ran ahead of
                                its stop test and dropped (cell_updates and the ms_* cover all of them)      */
    /* ABI 5 */
    double sparse_form;      /* sparse: the update stage the last batch took — 0 owner bands (k_sx_consume), 1 one 64-bit
                                atomic per +=, 2 two-level blocks (k_sxb_*); -1: no sparse batch yet               */
    double sparse_passes;    /* sparse, blocks: passes over disjoint row ranges run since the sequences were loaded */
    double share_positions;  /* sparse: leading kept positions the last batch sorted once per group of slots (0: none) */
    double share_groups;     /* ... and the groups it had                                                          */
    double sparse_desc;      /* sparse, owner bands: 1 when the last batch sent its long entries as descriptors that
                                k_sx_consume expands (tuning sparse_desc), else 0                                   */
} fsk_stats;

/* ---- lifecycle: replaces FastSK::FastSK (fastsk.cpp:19-28) and ~nothing (the reference leaks) */
int fsk_create(const fsk_config* cfg, fsk_engine** out);
void fsk_destroy(fsk_engine* e);
/* One engine over SEVERAL GPUs of this process — what FastSK(..., devices=[0,1,...]) creates. The reference
 * fans one compute_kernel call out over t host threads (fastsk_kernel.cpp:54-93: thread r takes work items
 * r, r+T, ...) and sum-reduces their private triangles (fastsk_kernel.cpp:286-315); here engine r of R lives
 * on devices[r], holds a replica of the packed sequences, takes combos r, r+R, ... of every accumulate and
 * the partial triangles are summed by ONE logical all-reduce (RCCL over xGMI, or the P2P kernels), issued per
 * row band on a communication stream under the next band's kernels, as int32 when every reduced cell provably
 * fits (C(g,m) * max_windows^2 < 2^31). Variance mode: Welford chain c runs on engine c mod R, one fp64
 * all-reduce of the chains' sums. Integer sums do not depend on their order: results are bit-identical to
 * fsk_create's for every R. The returned handle is engine 0 and takes every call of this header: load /
 * reset / accumulate / synchronize / finalize / compute act on the whole group (one host thread per
 * device), getters read engine 0's copy of the reduced triangle, fsk_bind_counts binds engine 0's triangle,
 * fsk_accumulate_rows and fsk_run_chains are single-engine calls and return FSK_ESTATE. ndev = 1 runs the
 * same banded, collective code with a world of one. */
int fsk_create_multi(const fsk_config* cfg, const int32_t* devices, int32_t ndev, fsk_engine** out);
typedef struct fsk_multi_info {
    int32_t ndev;            /* engines in the group (0: `e` came from fsk_create)                        */
    int32_t devices[16];
    int32_t collective;      /* FSK_COLL_RCCL or FSK_COLL_P2P: the one in use                              */
    int32_t comm_ranks;      /* ranks of the communicator the last collective ran over                     */
    int32_t bands;           /* row bands of the last accumulate                                           */
    int32_t narrow;          /* 1: the last accumulate exchanged int32                                     */
    int64_t reduce_bytes;    /* payload of the last accumulate's all-reduce, per engine                    */
    int64_t combos_per_engine[16]; /* combos of the last accumulate                                        */
    double reserved[4];
} fsk_multi_info;
int fsk_get_multi_info(fsk_engine* e, fsk_multi_info* out);
/* Tuning: every knob of the engine that is not a constructor argument of the reference — forcing one of two
 * equivalent code paths (tests), sizes of batches and launches (A/B measurements) — is a named integer key.
 * fsk_tuning_keys() lists them, one "key=default [lowest..highest] what it does" per line. A key is set per handle
 * with fsk_set_tuning (a group: on all its engines; takes effect from the next fsk_load_sequences / fsk_compute on),
 * or for every engine a process creates through ONE environment variable, FSK_TUNING="key=value,key=value", parsed
 * once by fsk_create — the only variable the library reads; an unknown key or a value out of range fails the call
 * (FSK_EINVAL). No key changes a result. trace=1 prints the keys in force (stderr). */
int fsk_set_tuning(fsk_engine* e, const char* key, int64_t value);
int fsk_get_tuning(fsk_engine* e, const char* key, int64_t* value);
const char* fsk_tuning_keys(void);
/* message of the last failure on `e` (or of the last failed fsk_create when e == NULL) */
const char* fsk_last_error(const fsk_engine* e);
int fsk_abi_version(void);
/* number of visible HIP devices (<0 on runtime error) */
int fsk_device_count(void);

/* ---- one-call path: replaces FastSK::compute_kernel (fastsk.cpp:30-118, n_test > 0) and
 * FastSK::compute_train (fastsk.cpp:120-188, n_test == 0) including
 * KernelFunction::compute_kernel (fastsk_kernel.cpp:24-106): exact / skip-variance / variance
 * modes per the config. Blocks until the result is resident on the device. */
int fsk_compute(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train,
                int64_t n_test);

/* Combo order used by the approx modes. Replaces the time(0)-seeded std::shuffle of
 * fastsk_kernel.cpp:29-38 with an explicit permutation (or prefix of one) of 0..C(g,m)-1.
 * Without it approx mode draws the order of fsk_set_seed's seed (0 when none was set). */
int fsk_set_combo_order(fsk_engine* e, const int32_t* order, int32_t n);
/* The seed of that shuffle: the order is the one the reference draws when time(0) == seed —
 * libstdc++'s std::shuffle over std::default_random_engine (minstd_rand0), fastsk_kernel.cpp:31-38,
 * restated in the engine (fsk_seed_order returns it). So FastSK(approx=True, seed=S) samples the
 * combos the reference sampled in the second S. */
int fsk_set_seed(fsk_engine* e, uint64_t seed);

/* ---- staged path (multi-GPU sharding, benchmarking with inputs resident in HBM) ------------ */
/* lengths check + dictionary + packing + H2D: fastsk.cpp:32-88 (extractFeatures is replaced by
 * bit-packed sequences that every combo re-reads) */
int fsk_load_sequences(fsk_engine* e, const int32_t* tokens, const int64_t* offsets, int64_t n_train,
                       int64_t n_test);
/* Use caller-provided device memory (uint64[n_pairs], e.g. a torch tensor that RCCL will
 * all-reduce) for the integer triangle instead of an engine-owned allocation. The buffer's contents are
 * taken as they are (call fsk_reset_counts for zeros); on a group handle the exchange stays 64 bits wide
 * until the next whole reset, because nothing bounds the cells the caller brought. */
int fsk_bind_counts(fsk_engine* e, void* device_u64, int64_t n_cells);
/* device address of the integer triangle (engine-owned or bound) */
int fsk_counts_device_ptr(fsk_engine* e, void** out);
/* Zero the integer triangle before another pass. The zeros may be written by the next tile launch
 * itself (it then stores its sums instead of adding them); every other reader of the triangle,
 * including fsk_synchronize — after which a caller may read a bound buffer — sees them. */
int fsk_reset_counts(fsk_engine* e);
/* zero only rows [row_begin, row_end) of the triangle (a rank that owns a band of rows) */
int fsk_reset_counts_rows(fsk_engine* e, int64_t row_begin, int64_t row_end);
/* THE HOT PATH. Adds the partial kernels of the listed combos into the integer triangle:
 * the loop body of kernel_build_parallel (fastsk_kernel.cpp:188-281) for each combo, and the
 * K += Ks reduce (fastsk_kernel.cpp:286-315). Asynchronous on the engine's stream. */
int fsk_accumulate(fsk_engine* e, const int32_t* combos, int32_t n);
/* The same, restricted to the cells (i, j<=i) with row_begin <= i < row_end (row bounds multiples
 * of 128, or N). Lets the host all-reduce finished row bands of the triangle while the next band
 * is still being accumulated; the count panels of an unchanged combo list are reused between
 * calls. fsk_accumulate(e, c, n) == fsk_accumulate_rows(e, c, n, 0, N). */
int fsk_accumulate_rows(fsk_engine* e, const int32_t* combos, int32_t n, int64_t row_begin, int64_t row_end);
/* wait for the engine's stream */
int fsk_synchronize(fsk_engine* e);
/* Order the engine's stream against a HIP stream of the caller (hipStream_t passed as void*; NULL =
 * the default stream) without blocking the host. fsk_stream_wait_engine: work enqueued on
 * `hip_stream` after the call waits for everything the engine has enqueued so far (e.g. RCCL's
 * all-reduce of a finished row band, issued on torch's stream, while the engine accumulates the next
 * band). Rows that an fsk_reset_counts left for a later storing launch are NOT filled by this call:
 * the other stream may read only rows already accumulated (fsk_synchronize fills the rest; an accumulate
 * with an empty combo list fills the rows it was given).
 * fsk_engine_wait_stream: the engine's later work waits for what `hip_stream` holds now.
 * `hip_stream` must be a stream of the engine's device. */
int fsk_stream_wait_engine(fsk_engine* e, void* hip_stream);
int fsk_engine_wait_stream(fsk_engine* e, void* hip_stream);
/* extract the raw diagonal for normalisation (fastsk_kernel.cpp:96-103); call after the last
 * accumulate (and after any cross-GPU all-reduce of the triangle) */
int fsk_finalize(fsk_engine* e);

/* ---- results: replace the getters of fastsk.cpp:190-221 ---------------------------------- */
/* normalised K[i0:i1, j0:j1] as row-major doubles, any sub-block of the symmetric N x N matrix */
int fsk_get_block(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double* out);
/* the same block written to DEVICE memory owned by the caller (e.g. a torch tensor): no host copy,
 * for consumers that keep the kernel matrix on the GPU */
int fsk_get_block_device(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double* device_out);
/* the same block in device memory the ENGINE allocates on its device (*device_out; release with
 * fsk_free_device, whose `e` may be NULL once the engine is destroyed): for host code that has no allocator of its own for the GPU — the pybind11 class wraps
 * it in a DLPack capsule, so the SVM stage can take the kernel matrix without a host bounce */
int fsk_alloc_block_device(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, double** device_out);
int fsk_free_device(fsk_engine* e, void* device_ptr);
/* change fsk_config.skip_test_block for later computes (a caller that finds it needs test x test cells after all) */
int fsk_set_skip_test_block(fsk_engine* e, int32_t skip);
int fsk_get_train(fsk_engine* e, double* out);      /* n_train x n_train, get_train_kernel()   */
int fsk_get_test(fsk_engine* e, double* out);       /* n_test  x n_train, get_test_kernel()    */
int fsk_get_triangle(fsk_engine* e, double* out);   /* double[N(N+1)/2], the reference's K     */
/* The whole normalised triangle (fastsk_kernel.cpp:96-103 applied to every cell) written to DEVICE memory:
 * N(N+1)/2 doubles at `device_out` (caller-owned), or in a buffer the engine allocates on its device
 * (*device_out; release with fsk_free_device). One streaming pass: 16 bytes per cell. */
int fsk_get_triangle_device(fsk_engine* e, double* device_out);
int fsk_alloc_triangle_device(fsk_engine* e, double** device_out);
int fsk_get_counts(fsk_engine* e, uint64_t* out);   /* raw integer triangle (exact/skip-var)   */
int fsk_get_counts_block(fsk_engine* e, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint64_t* out);
/* raw integer cells (rows[q], cols[q]), q < n, of the symmetric matrix: scattered spot checks of a
 * triangle too large to copy out (tri_access of arbitrary pairs, shared.cpp:97-117) */
int fsk_get_counts_cells(fsk_engine* e, const int64_t* rows, const int64_t* cols, int64_t n, uint64_t* out);
/* Order-free digest of the integer cells of rows [row_begin, row_end): out[0] = sum of the cells (mod 2^64),
 * out[1] = xor over the cells of cell * (index | 1) (mod 2^64), index = i(i+1)/2 + j. One pass on the device.
 * Digests of disjoint row ranges combine (add / xor), so a triangle held in row bands by several ranks — or
 * reduced over several GPUs — can be compared with the single-GPU one without copying it out: the
 * "bit-identical at 1/2/4/8 GPUs" gate of the exact mode (integer sums, fastsk_kernel.cpp:286-315). */
int fsk_counts_digest(fsk_engine* e, int64_t row_begin, int64_t row_end, uint64_t out[2]);
/* The reduction inside get_variance (fastsk_kernel.cpp:116-131): the sum of n doubles in INDEX ORDER,
 * s = fl(s + values[i]) for i = 0 .. n-1, to the last bit, computed on the device (the stop test of
 * approx mode depends on it). `values` is a host array. Exposed so that the summation can be verified
 * on its own. */
int fsk_sequential_sum(fsk_engine* e, const double* values, int64_t n, double* out);
/* Variance mode over several GPUs (the T Welford chains are the units): after
 * fsk_load_sequences, run the chains first, first + step, ... < T of this engine — one worker thread's
 * share of fastsk_kernel.cpp:188-281 — leaving the sum of their K_hat (fastsk_kernel.cpp:286-315) in the
 * engine's fp64 triangle; chain 0's engine holds get_stdevs(). The caller sums the triangles of all
 * engines (fsk_get_kernel_sum_device -> e.g. an RCCL all-reduce -> fsk_set_kernel_sum_device; device
 * buffers of N(N+1)/2 doubles) and calls fsk_finalize. With T > 1 the fp64 sum over chains depends on
 * its order, as it does between the reference's threads. */
int fsk_run_chains(fsk_engine* e, int32_t first, int32_t step);
int fsk_get_kernel_sum_device(fsk_engine* e, double* device_out);
int fsk_set_kernel_sum_device(fsk_engine* e, const double* device_in);
/* approx/variance mode: thread 0's convergence trace, get_stdevs() (fastsk.cpp:219-221) */
int fsk_get_stdevs(fsk_engine* e, double* out, int32_t cap, int32_t* n);
/* "%d:%e " text dump, one row per line, 1-based column ids: save_kernel (fastsk.cpp:223-237) */
int fsk_save_kernel(fsk_engine* e, const char* path);
int fsk_get_stats(fsk_engine* e, fsk_stats* out);

/* ---- helpers shared with the host side --------------------------------------------------- */
int64_t fsk_num_combos(int32_t g, int32_t m);                              /* nchoosek */
int fsk_combo_positions(int32_t g, int32_t k, int64_t combo, int32_t* out); /* getCombinations */
/* the permutation of 0..n-1 that std::shuffle(begin, end, std::default_random_engine{seed}) of libstdc++ produces
 * (fastsk_kernel.cpp:31-38 with n = C(g,m) and seed = time(0)); host only */
int fsk_seed_order(uint64_t seed, int64_t n, int32_t* out);


/* ---- input: native counterpart of FastaUtility.read_data + Vocabulary (src/fastsk/utils.py:5-96)
 * Alternating ">label" / sequence lines; every line stripped and lower-cased; labels in {-1,0,1};
 * token ids in first-seen order from *next_id (1 for a fresh vocabulary; id 0 stays reserved),
 * kept in vocab256[byte] (0 = not seen yet) so that successive files share ids as the files read
 * through one FastaUtility do. Host only, no device needed. Call once with tokens = offsets =
 * labels = NULL to learn *n_seq and *n_tokens (the vocabulary is already updated; the second call
 * finds every symbol assigned), then with buffers tokens[n_tokens], offsets[n_seq + 1],
 * labels[n_seq]. Returns FSK_EINVAL on a malformed file (where the reference's asserts fire) and
 * FSK_EUNSUPPORTED on non-ASCII bytes (the caller's text-mode fallback handles those); the message
 * goes to err[err_cap]. */
int fsk_read_fasta(const char* path, int32_t* vocab256, int32_t* next_id, int32_t* tokens, int64_t tokens_cap,
                   int64_t* offsets, int32_t* labels, int64_t seq_cap, int64_t* n_seq, int64_t* n_tokens, char* err,
                   int32_t err_cap);

#ifdef __cplusplus
}
#endif
#endif /* FASTSK_AMD_H */

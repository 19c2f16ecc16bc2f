// fsk_engine_variance.hip — approx / variance mode (fastsk_kernel.cpp:108-143, 188-262): T Welford chains,
// batches of iterations run ahead of the stop test, the exact sequential fp64 sum on the device.
#include "fsk_engine_internal.h"
#include "fsk_kernels_variance.h"

using namespace fsk_detail;

namespace fsk_detail {

// the reference's `avg` of get_variance (fastsk_kernel.cpp:116-131): sum of n doubles in index order,
// on the device (k_seq_prep on all CUs + k_seq_chain); bsum = approximate sums per SQ_BLOCK values
// (zero on entry, zero again on exit), result to out[0]
// `count` independent sums laid out `stride` values apart (their block sums / block records / results
// follow each other); the chains run on `chain_stream`
// grp: 2 * SQ_GROUPS group records per block, laid out like blk
int enqueue_sequential_sum(fsk_engine* e, const double* d_vals, u64 n, double* bsum, fsk::SeqBlk* blk, fsk::SeqGrp* grp, double* out,
                           int count = 1, u64 stride = 0, hipStream_t chain_stream = nullptr, hipEvent_t handoff = nullptr,
                           hipStream_t main = nullptr) {
    if (!main) main = e->stream;  // (the stream the values were produced on)
    const uint32_t nblocks = (uint32_t)((n + fsk::SQ_BLOCK - 1) / fsk::SQ_BLOCK);
    if (nblocks > 0)
        FSK_LAUNCH(fsk::k_seq_prep, dim3(nblocks, count), dim3(256), 0, main, d_vals, n, (const double*)bsum, blk, stride, nblocks);
    hipStream_t cs = chain_stream ? chain_stream : main;
    if (chain_stream) {
        FSK_HIP(hipEventRecord(handoff, main));
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
int run_variance_mode(fsk_engine* e, int T, int chain_first, int chain_step) {
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    // (a previous call may have left the sequential sums of a batch beyond its stop running: see the end of this function)
    if (e->chain_stream) FSK_HIP(hipStreamSynchronize(e->chain_stream));
    const int64_t pairs = e->pairs;
    const int64_t train_pairs = (int64_t)((e->n_train / (double)2) * (e->n_train + 1));
    const size_t tp = (size_t)std::max<int64_t>(1, train_pairs);
    // strides of the per-iteration arrays — slot triangles, the K_hat ring, the products — padded to multiples of 4 cells:
    // every array then starts 16-byte aligned and k_welford_batch_v moves 16 bytes per access (cells in [pairs, ps) of a
    // slot hold nothing anybody reads)
    const size_t ps = ((size_t)pairs + 3) & ~(size_t)3, tps = (tp + 3) & ~(size_t)3;
    // The stop test of iteration i needs avg_variance, a SEQUENTIAL fp64 sum in triangle-index
    // order (fastsk_kernel.cpp:116-131), to the last bit. It is computed on the device
    // (enqueue_sequential_sum), so only 8 bytes per iteration come back; the engine still runs AHEAD
    // of its stop test: iterations are issued in batches of AHEAD with up to DEPTH batches in flight,
    // the Welford state of every untested iteration is kept in a ring, and whatever lies beyond the
    // stopping iteration is dropped.
    // (the sums of a batch run on a stream of their own under the kernels of the batches that follow)
    // (Tried as well: deciding "no stop in this batch" from the approximate block sums, a safe margin below 1.96, so
    // that the next issue does not wait for the oldest batch's exact sums — the lanes then run in lockstep and the
    // GPU's throughput bounds the batch rate just the same: 6.2 -> 6.3 ms on config 1.)
    // (batches of 8 iterations: 4 / 6 / 8 give 7.5 / 6.55 / 5.75 ms on config 1 — the two dozen launches of a batch are
    // paid per batch —; 10 / 12 / 16, as two Welford passes of up to 8 over one sparse batch: 6.3 / 5.9 / 6.0 ms there, and
    // 18.2 against 18.6 ms on the 210 iterations of the reference's CI configuration: not worth twice the buffers)
    // (two batches in flight, one per lane. Measured with three and four — the issue of a batch waits for the sums
    // of the oldest one in flight, a latency a third batch would cover —: config 1 6.4 -> 6.8 -> 7.4 ms; the GPU is
    // busy as it is, and what the extra batches add are iterations beyond the stop.)
    constexpr int MAX_AHEAD = 16, MAX_DEPTH = 2, LANES = 2;
    const int AHEAD = 8;  // iterations per batch (see above)
    const int DEPTH = MAX_DEPTH, INFLIGHT = MAX_DEPTH;
    const int RING = DEPTH * AHEAD + 1;
    const bool trace = e->trace();  // stderr: where the wall time of this mode goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    double t_wait = 0;
    const auto t_begin = now();
    const size_t nblk = (tp + fsk::SQ_BLOCK - 1) / fsk::SQ_BLOCK;
    const size_t slots = (size_t)DEPTH * AHEAD;
    FSK_HIP(e->d_Kf64.reserve((size_t)pairs));
    FSK_HIP(e->d_Khat.reserve(ps * RING));
    FSK_HIP(e->d_prod.reserve(tps * slots));
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
    hipEvent_t ev_done[MAX_DEPTH], ev_hand[MAX_DEPTH], ev_wf[MAX_DEPTH];  // (ev_wf: one per lane)
    for (auto& ev : ev_done) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto& ev : ev_hand) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto& ev : ev_wf) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // Sparse batches in flight alternate between two lanes (scratch + stream, sx_lane_of): the short kernels of one
    // batch's sort and segmentation run in the gaps of the other's emit / consume / Welford. What orders them: a
    // batch's Welford pass waits for the previous batch's (K_hat is handed from one to the next) through ev_wf.
    if (!e->profile_sync() && !e->lane_stream) FSK_HIP(hipStreamCreateWithFlags(&e->lane_stream, hipStreamNonBlocking));
    auto sync_all = [e]() {
        (void)hipStreamSynchronize(e->stream);
        if (e->lane_stream) (void)hipStreamSynchronize(e->lane_stream);
        (void)hipStreamSynchronize(e->chain_stream);
    };
    struct Cleanup {
        hipEvent_t* a; hipEvent_t* b; hipEvent_t* c; fsk_engine* e;
        bool leave_sums = false;  // (a regular end: only a dropped batch's sequential sums may still run, see below)
        ~Cleanup() {
            (void)hipStreamSynchronize(e->stream);  // nothing of this call may still be in flight
            if (e->lane_stream) (void)hipStreamSynchronize(e->lane_stream);
            if (!leave_sums) (void)hipStreamSynchronize(e->chain_stream);
            for (int i = 0; i < MAX_DEPTH; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); (void)hipEventDestroy(c[i]); }
        }
    } cleanup{ev_done, ev_hand, ev_wf, e};
    if (e->lane_stream) {  // lane 1 starts after the uploads and fills enqueued on the engine's stream so far
        FSK_HIP(hipEventRecord(ev_wf[0], e->stream));
        FSK_HIP(hipStreamWaitEvent(e->lane_stream, ev_wf[0], 0));
    }
    bool wf_set[LANES] = {false, false};  // ev_wf[lane] holds a Welford pass of the chain at hand
    const double t_alloc = ms_since(t_begin);
    const uint32_t blocks = (uint32_t)((pairs + 255) / 256);                                       // one cell per thread
    const uint32_t wblocks = (uint32_t)((pairs + 256 * fsk::WF_ITEMS - 1) / (256 * fsk::WF_ITEMS)); // k_welford
    const int n_order = (int)e->order.size();
    auto khat = [&](int i) { return e->d_Khat.p + (size_t)(i % RING) * ps; };
    struct Batch { int n = 0, first_iter = 0, first_item = 0, base = 0, part = 0, lane = 0; bool grouped = false, s16 = false; };
    // Where the stop test is expected to fire: the variance estimate settles, so sd falls like 1 / sqrt(iter) and
    // delta / sd > 1.96 is reached near iter * (1.96 sd / delta)^2. Only the SIZE of the batches issued ahead follows
    // from it (what lies beyond the stop is thrown away: a full batch there is an eighth of config 1's work); a
    // wrong guess costs small batches, never a result.
    int pred_stop = INT32_MAX;
    constexpr int HEDGE = 2;  // iterations issued ahead once the stop is expected to have fired already
    // how many iterations can still follow (end of the work list, max_iters)
    auto plan = [&](int first_iter, int first_item) {
        // (sparse batches: the first of a set of sequences is sized for what is known to fit, later ones by the update
        // words per record seen — a batch beyond the stream limit would fall back to one iteration at a time for good)
        int most = AHEAD;
        if (e->path == FSK_PATH_SPARSE) {
            if (e->sx_wpr == 0) most = std::min(most, (int)fsk::WF_SLOTS);
            else most = std::max(1, std::min(most, (int)((double)(e->sx_max_words() / 2) / (e->sx_wpr * (double)std::max<int64_t>(1, e->nfeat)))));
        }
        int n = most;
        if (pred_stop != INT32_MAX) {
            const int64_t left = (int64_t)pred_stop + 1 - first_iter;  // (one to spare)
            n = left >= most ? most : left > 0 ? (int)left : HEDGE;
        }
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
    bool grouped = e->path == FSK_PATH_SPARSE && e->sx_lists && !e->tune.sparse_global && e->tune.sparse_form != 3;
    // Dense dataflow with a tile kernel that can STORE (one workgroup per tile, the direct-to-LDS kernel, no
    // key compaction, no test-block filter): every iteration's tile launch stores its counts into a u64
    // triangle of its own — no zero fill — and the batch's Welford updates run as one pass like the sparse
    // batches' (while the slot triangles stay a modest share of the memory).
    const bool dense_slots = e->path == FSK_PATH_DENSE && !e->compact && !(e->cfg.skip_test_block && e->n_test > 0) &&
                             (u64)pairs * AHEAD * DEPTH * sizeof(u64) <= ((u64)8 << 30) && e->tune.variance_dense_slots;
    // (one set of slot triangles per batch in flight: a stop inside a batch runs its Welford prefix again)
    if (grouped) FSK_HIP(e->d_Kslots.reserve((ps * AHEAD * DEPTH + 1) / 2));
    if (dense_slots) FSK_HIP(e->d_Kslots.reserve(ps * AHEAD * DEPTH));
    auto slots_of = [&](int part) { return reinterpret_cast<uint32_t*>(e->d_Kslots.p) + (size_t)part * AHEAD * ps; };
    auto slots64_of = [&](int part) { return e->d_Kslots.p + (size_t)part * AHEAD * ps; };
    // The Welford pass of a batch: k_welford_batch carries up to WF_SLOTS iterations in registers, a larger batch takes
    // several passes (each leaves the state after its last iteration in the ring). `first` .. `first + count`: the
    // batch's slots to fold, from the state after slot first - 1; with_sums = 0: states only (a stop inside the batch).
    auto welford_passes = [&](const Batch& B, bool u64_slots, int first, int count, int with_sums, hipStream_t st) -> int {
        for (int c = first; c < first + count; c += fsk::WF_SLOTS) {
            const int nc = std::min((int)fsk::WF_SLOTS, first + count - c);
            const size_t slot = (size_t)B.part * AHEAD + c;
            fsk::WfRecip rr;  // 1 / iteration number of every slot of the pass (the IEEE quotient, as the kernels compute it)
            for (int q = 0; q < fsk::WF_SLOTS; ++q) rr.r[q] = 1.0 / ((double)(B.first_iter + c) + (double)q);
            if (B.s16 && !u64_slots)  // (u16 slot triangles: the same cells, `ps` apart, in half the bytes)
                FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch_v<uint16_t>), dim3(wblocks), dim3(256), 0, st,
                           reinterpret_cast<const uint16_t*>(slots_of(B.part)) + (size_t)c * ps, (u64)ps, nc, (const double*)khat(B.base + c), khat(B.base + c + nc),
                           with_sums ? e->d_prod.p + slot * tps : (double*)nullptr, (u64)tps, (u64)pairs, (u64)train_pairs,
                           (double)(B.first_iter + c), rr, with_sums ? e->d_bsum.p + slot * nblk : (double*)nullptr, (uint32_t)nblk, with_sums);
            else if (u64_slots)
                FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch_v<u64>), dim3(wblocks), dim3(256), 0, st, (const u64*)slots64_of(B.part) + (size_t)c * ps,
                           (u64)ps, nc, (const double*)khat(B.base + c), khat(B.base + c + nc), with_sums ? e->d_prod.p + slot * tps : (double*)nullptr,
                           (u64)tps, (u64)pairs, (u64)train_pairs, (double)(B.first_iter + c), rr,
                           with_sums ? e->d_bsum.p + slot * nblk : (double*)nullptr, (uint32_t)nblk, with_sums);
            else
                FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford_batch_v<uint32_t>), dim3(wblocks), dim3(256), 0, st,
                           (const uint32_t*)slots_of(B.part) + (size_t)c * ps, (u64)ps, nc, (const double*)khat(B.base + c), khat(B.base + c + nc),
                           with_sums ? e->d_prod.p + slot * tps : (double*)nullptr, (u64)tps, (u64)pairs, (u64)train_pairs,
                           (double)(B.first_iter + c), rr, with_sums ? e->d_bsum.p + slot * nblk : (double*)nullptr, (uint32_t)nblk, with_sums);
        }
        return FSK_OK;
    };
    // (Tried: the Welford pass INSIDE the by-slot pass over the update streams — one workgroup per owner band goes
    // through the batch's slots in order, K_hat in registers, no slot triangles written and read back (2 x 204 MB
    // of a config-1 batch's 1.3 GB). The slots then come one after the other inside a workgroup, a few tens of words
    // per thread each, and every memory latency is paid eight times: 390 us a batch against 190 + 130 us for
    // k_sx_consume over (band, slot) in parallel + k_welford_batch.)
    auto issue = [&](Batch& B) -> int {
        const bool was_grouped = grouped;
        if (grouped) {
            int32_t combos[MAX_AHEAD];
            for (int b = 0; b < B.n; ++b) combos[b] = e->order[B.first_item + b * T];
            e->sx_slot16 = e->tune.var_slots16 && e->slots16_ok;  // (u16 slot triangles until a sum of these sequences has not fit one)
            int rc = do_accumulate(e, combos, B.n, reinterpret_cast<u64*>(slots_of(B.part)), 0, -1, (u64)ps, B.part);
            e->sx_slot16 = false;
            B.s16 = e->sx_slot16_used;
            if (rc == FSK_RETRY_UNGROUPED) grouped = false;  // too many updates for one stream: from here on one iteration at a time
            else if (rc) return rc;
        }
        B.grouped = grouped || dense_slots;  // (the batch's counts sit in slot triangles, its Welford update is one pass)
        // the stream this batch's passes over the triangle run on: the lane accumulate_sparse really ran the batch in
        // (it drops the deferral, and with it lane 1, for a batch it has to split), else the engine's
        const int lane = grouped ? e->sx_last_lane : 0;
        B.lane = lane;
        hipStream_t bs = lane ? e->lane_stream : e->stream;
        if (was_grouped && !grouped) sync_all();  // (leaving the two-lane form: everything drains first, once)
        for (int l = 0; l < LANES; ++l)  // K_hat comes from the previous batch's Welford pass, wherever that ran
            if (l != lane && wf_set[l]) FSK_HIP(hipStreamWaitEvent(bs, ev_wf[l], 0));
        if (grouped) {  // K_hat through the batch's iterations, WF_SLOTS per pass
            int rc = welford_passes(B, false, 0, B.n, 1, bs);
            if (rc) return rc;
        } else if (dense_slots) {
            for (int b = 0; b < B.n; ++b) {
                int32_t combo = e->order[B.first_item + b * T];
                e->store_next = true;
                int rc = do_accumulate(e, &combo, 1, slots64_of(B.part) + (size_t)b * ps);
                e->store_next = false;
                if (rc) return rc;
            }
            int rc = welford_passes(B, true, 0, B.n, 1, e->stream);
            if (rc) return rc;
        }
        for (int b = 0; b < B.n && !B.grouped; ++b) {
            const size_t slot = (size_t)(B.part * AHEAD + b);
            FSK_HIP(hipMemsetAsync(e->d_K, 0, (size_t)pairs * sizeof(u64), e->stream));
            int32_t combo = e->order[B.first_item + b * T];
            int rc = do_accumulate(e, &combo, 1, e->d_K);
            if (rc) return rc;
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_welford<u64>), dim3(wblocks), dim3(256), 0, e->stream, (const u64*)e->d_K, (const double*)khat(B.base + b),
                       khat(B.base + b + 1), e->d_prod.p + slot * tps, (u64)pairs, (u64)train_pairs, (double)(B.first_iter + b),
                       e->d_bsum.p + slot * nblk);
        }
        // the batch's sums: block totals on all CUs, then one wave per iteration walks its blocks — on a
        // second stream, under the kernels of the batches that follow
        const size_t slot0 = (size_t)B.part * AHEAD;
        FSK_HIP(hipEventRecord(ev_wf[lane], bs));  // the state after this batch is on its way
        wf_set[lane] = true;
        int rc = enqueue_sequential_sum(e, e->d_prod.p + slot0 * tps, (u64)train_pairs, e->d_bsum.p + slot0 * nblk,
                                        reinterpret_cast<fsk::SeqBlk*>(e->d_seqblk.p) + slot0 * nblk,
                                        seq_grp + slot0 * nblk * (2 * fsk::SQ_GROUPS), h_avg + slot0, B.n, (u64)tps,
                                        e->chain_stream, ev_hand[B.part], bs);  // (the sums land in pinned host memory)
        if (rc) return rc;
        FSK_HIP(hipEventRecord(ev_done[B.part], e->chain_stream));
        return FSK_OK;
    };
    e->stdevs.clear();
    for (int tid = chain_first; tid < T; tid += chain_step) {
        int cur = 0;  // ring position of the state after the last accepted iteration
        pred_stop = INT32_MAX;
        FSK_HIP(hipMemsetAsync(khat(cur), 0, (size_t)pairs * sizeof(double), e->stream));
        int iter = 1, item = tid;
        std::vector<Batch> q;  // issued, untested batches, oldest first (at most INFLIGHT)
        std::vector<int> dropped;        // sets of buffers of the batches issued beyond the stop
        hipStream_t fin = e->stream;     // the stream the chain's last passes run on
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
            while ((int)q.size() < INFLIGHT) {  // keep the device INFLIGHT batches ahead of the stop test
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
                sync_all();
                for (const Batch& B : q) { e->st.combos_done -= B.n; e->sx_defer[B.part].active = false; }
                q.clear();
                const bool was = e->sx_redoing;
                e->sx_redoing = true;
                Batch R = A;
                int rc = issue(R);
                e->sx_redoing = was;
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
                if (iter >= 2 && sd > 0.0 && e->cfg.delta > 0.0) {
                    const double at = (double)iter * (1.96 * sd / e->cfg.delta) * (1.96 * sd / e->cfg.delta);
                    pred_stop = at < 1e9 ? std::max(iter, (int)std::ceil(at)) : INT32_MAX;
                }
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
                // What was issued beyond the stop is dropped — but not waited for here: the chain's result needs A's
                // buffers and the K_hat ring only (a dropped batch reads the state after A and writes the one after
                // itself), so the last passes run on A's own, idle stream beside the dropped kernels' tail; everything
                // drains, and the dropped batches' counts are read, before the chain's buffers are used again (below).
                for (const Batch& B : q) { e->st.combos_done -= B.n; dropped.push_back(B.part); }
                const int lane_a = (A.grouped && !dense_slots) ? A.lane : 0;
                fin = lane_a ? e->lane_stream : e->stream;
                // the stop fell inside the batch: the state after its accepted prefix (the passes before the one the stop
                // fell in have left theirs in the ring)
                const int done = accepted / (int)fsk::WF_SLOTS * (int)fsk::WF_SLOTS;
                if (A.grouped && accepted < A.n && accepted > done) {
                    int rc = welford_passes(A, dense_slots, done, accepted - done, 0, fin);
                    if (rc) return rc;
                }
                break;
            }
            // (working implies more items and iterations: the queue is not empty)
        }
        // (the chain's last state was written on `fin`, or by a batch whose sums have been waited for)
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_add_nonzero<double>), dim3(blocks), dim3(256), 0, fin, e->d_Kf64.p, (const double*)khat(cur), (u64)pairs);
        // Everything drains before the next chain starts over with these buffers. After the LAST chain the sequential
        // sums of a batch beyond the stop (one wave per iteration, 200 us, nobody reads them) are left to finish on their
        // own stream: they touch the sums' buffers only, which the next variance-mode call, and fsk_destroy, wait for.
        const bool last_chain = tid + chain_step >= T;
        if (last_chain) {
            FSK_HIP(hipStreamSynchronize(e->stream));
            if (e->lane_stream) FSK_HIP(hipStreamSynchronize(e->lane_stream));
        } else {
            sync_all();
        }
        for (int part : dropped) (void)sx_harvest(e, part);
        for (bool& w : wf_set) w = false;
    }
    e->result_f64 = true;
    FSK_HIP(hipStreamSynchronize(e->stream));
    cleanup.leave_sums = true;
    if (trace)
        fprintf(stderr, "[fsk] variance mode: setup %.2f ms, waiting for the GPU %.2f ms, total %.2f ms (%lld cells/iteration, %d batches in flight)\n",
                t_alloc, t_wait, ms_since(t_begin), (long long)train_pairs, INFLIGHT);
    return FSK_OK;
}

}  // namespace fsk_detail

extern "C" {

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

}  // extern "C"

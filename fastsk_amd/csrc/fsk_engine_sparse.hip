// fsk_engine_sparse.hip — host side of the SPARSE dataflow (configs 1, 4; the fallback for everything): the
// reference's own pipeline per combo — gather, cntsrtna, permute, countAndUpdateTri (fastsk_kernel.cpp:224-241,
// shared.cpp:156-191, 268-333) — as batched streams: extract -> LSD radix sort -> segments -> update streams ->
// owner bands.
#include "fsk_engine_internal.h"
#include "fsk_kernels_sparse.h"

using namespace fsk_detail;

namespace fsk_detail {

// ---------------------------------------------------------------------------------------------
// sparse dataflow: owner bands of K. The update stream of a band is summed in LDS by one workgroup,
// so a band is a range of whole rows with 8192 or 16384 cells + up to a row (the LDS budget of k_sx_consume
// bounds it: SX_CAP cells per round, at most SX_MAX_ROUNDS rounds over the band's stream).
constexpr uint32_t SX_CAP = 20480;        // u32 cells of K one k_sx_consume workgroup holds in LDS (80 KiB: two workgroups per CU; its parts table stays in global memory)
constexpr uint32_t SX_MAX_ROUNDS = 16;
constexpr uint32_t SX_CAP_SLOT = 20480;   // by-slot form (no parts table in LDS): 80 KiB, two workgroups per CU

// ---- two-level form (fsk_sparse_blocks.inc): the bands of ONE pass over rows [ra, rb) ------------------------------
constexpr uint32_t SX_BLOCKS_FROM_ROUNDS = 4;  // the owner bands up to this many LDS rounds a band, blocks beyond (measured: protein-like N = 8,000, four rounds: bands 0.041 ms a combo, blocks 0.046; N = 12,000, fourteen rounds: 0.133 against 0.075)
// Bands of 2^t cells counted from the pass's first row, t the smallest that leaves at most `blocks_max_bands` bands; a band's
// cell offsets (2^t + a row at most) and an 8-bit product at least share the 32-bit word, so t <= 23 and a pass covers
// fewer than 2^32 cells. false: rows [ra, rb) do not fit one pass (the caller halves the range) or the sub-bands of a band
// outnumber the scatter's histogram (sequences in the millions: 64-bit atomics then).
bool blocks_plan_pass(fsk_engine* e, int64_t ra, int64_t rb, SxPass* out) {
    const u64 N = (u64)e->N;
    const u64 c_lo = (u64)ra * ((u64)ra + 1) / 2, c_hi = (u64)rb * ((u64)rb + 1) / 2, cells = c_hi - c_lo;
    const int sub_shift = (int)e->tune.blocks_sub_shift;
    const u64 max_bands = (u64)e->tune.blocks_max_bands;
    int t_max = (int)e->tune.blocks_band_shift_max;
    while (t_max > sub_shift && ((u64)1 << t_max) + N > ((u64)1 << 24)) --t_max;  // (pb >= 8)
    if (((u64)1 << t_max) + N > ((u64)1 << 24)) return false;
    if ((((((u64)1 << t_max) + N) >> sub_shift) + 1) > (u64)fsk::SXB_MAX_SUB) return false;
    if (!out) return true;  // (the form exists for these sequences)
    if (rb - ra > 1 && (cells > ((max_bands - 1) << t_max) || cells >= ((u64)1 << 32))) return false;  // (one row is always a pass)
    int t = sub_shift;
    while (t < t_max && ((cells + (((u64)1) << t) - 1) >> t) > max_bands) ++t;
    u64 nb = std::max<u64>(1, (cells + (((u64)1) << t) - 1) >> t);
    if (nb > (u64)fsk::SX_MAX_OWNERS) return false;  // (a single row longer than the bands allow: N beyond 2^23 — excluded above)
    out->ra = ra; out->rb = rb; out->t = t; out->sub_shift = sub_shift;
    out->n_owners = (uint32_t)nb;
    out->own_base = (uint32_t)c_lo;  // (mod 2^32, as sx_tri32 computes it)
    out->r0.assign((size_t)nb + 1, (uint32_t)rb);
    u64 largest = 0;
    {
        uint32_t o = 0;  // r0[o] = first row whose first cell, counted from the pass's first cell, reaches o << t
        for (u64 i = (u64)ra; i < (u64)rb && o <= nb; ++i)
            while (o <= nb && (i * (i + 1) / 2 - c_lo) >= ((u64)o << t)) out->r0[o++] = (uint32_t)i;
        for (u64 q = 0; q < nb; ++q) {
            const u64 a = out->r0[q], b = out->r0[q + 1];
            largest = std::max(largest, b * (b + 1) / 2 - a * (a + 1) / 2);
        }
    }
    int L = 1;
    while (((u64)1 << L) < largest) ++L;
    out->pb = 32 - L;
    out->submax = (uint32_t)(((largest + (((u64)1) << sub_shift) - 1) >> sub_shift) + 1);
    return out->pb >= 8 && out->submax <= fsk::SXB_MAX_SUB;
}

void plan_owner_bands(fsk_engine* e) {
    // band o = the rows whose first cell index lies in [o << t, (o + 1) << t): a row's band is a shift
    // of its triangular index, bands hold about 2^t cells (2^t + N at most: the last row of a band is
    // kept whole) and can be empty when a single row is longer than 2^t cells.
    const u64 N = (u64)e->N, cells = N * (N + 1) / 2;
    const bool want_pairs = e->tune.sparse_pairs != 0;
    if (e->owner_N == e->N && e->n_owners != 0 && e->sx_pairs_asked == want_pairs) return;  // (the bands depend on the number of sequences alone: the same plan, and its table stays on the device)
    e->owner_N = e->N;
    e->sx_pairs_asked = want_pairs;
    // (as large as ONE round of k_sx_consume takes: with half the bands k_sx_emit bins, scans and offsets half as much per
    // tile and the per-(tile, band) word counts are half the matrix — config 4, N = 2560: 400 -> 200 bands, emit 3.75 ->
    // 3.57 ms, the column scans 0.37 -> 0.26, consume 1.04 -> 1.13)
    int t = 13;
    while (t < 20 && ((u64)1 << (t + 1)) + N <= (u64)SX_CAP) ++t;
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
    // PAIRS: a band of fewer than 32767 cells (every workload the streams are fast on) takes its unit products — both
    // multiplicities 1: 94 % of the words on protein data — as bare 15-bit cells, two to a 32-bit container with bit 31 set
    // (0x7fff: no cell); every other word is {0, cell: 15 bits, product: 16 bits}. Half the bytes of those words between
    // k_sx_emit and k_sx_consume.
    e->sx_pairs = want_pairs && largest <= 32766;
    if (e->sx_pairs) e->sx_pb = 16;
    e->sx_rounds = (uint32_t)std::max<u64>(1, (largest + SX_CAP - 1) / SX_CAP);
    e->sx_cap = (uint32_t)std::max<u64>(1, std::min<u64>(SX_CAP, largest));
    e->sx_rounds_slot = (uint32_t)std::max<u64>(1, (largest + SX_CAP_SLOT - 1) / SX_CAP_SLOT);
    e->sx_cap_slot = (uint32_t)std::max<u64>(1, std::min<u64>(SX_CAP_SLOT, largest));
    e->sx_lists = e->n_owners <= (uint32_t)fsk::SX_MAX_OWNERS && e->sx_rounds <= SX_MAX_ROUNDS && e->sx_pb >= 8;
    e->owner_ready = false;
}

// Which form the update stage takes (fsk_sparse_blocks.inc): the owner bands while a band is a few LDS rounds (every
// round re-reads the band's stream), the two-level blocks beyond — and wherever the bands do not exist at all.
// Tuning sparse_form: 1 = bands whenever they exist, 2 = blocks always, 3 = one 64-bit atomic per += (as sparse_global).
void sx_choose_form(fsk_engine* e) {
    const int64_t want = e->tune.sparse_form;
    const bool blocks_ok = blocks_plan_pass(e, 0, e->N, nullptr);
    // (with descriptors a round of the bands walks every descriptor's partners again: the blocks from two rounds a band on —
    // measured, DNA k = 8, N = 6,000 / 8,000 / 12,000 / 16,000, 495 combos: bands 0.09 / 0.14 / 0.86 / 1.3 s, blocks 0.07 / 0.10 / 0.16 / 0.22;
    // one round, N = 4,000, the large-g regime: bands 0.83 s, blocks 1.38)
    const uint32_t max_rounds = e->sx_desc_now() && e->tune.sparse_desc_blocks ? 1u : SX_BLOCKS_FROM_ROUNDS;
    e->sx_form = (want == 3 || e->tune.sparse_global) ? 1
                 : (want == 2 && blocks_ok)          ? 2
                 : (want == 1 && e->sx_lists)        ? 0
                 : e->sx_lists && e->sx_rounds <= max_rounds ? 0
                 : blocks_ok                                            ? 2
                 : e->sx_lists                                          ? 0
                                                                        : 1;
}

// descriptors for this batch? (owner bands only; see fsk_engine::sx_desc_now)
inline bool sx_desc_wanted(const fsk_engine* e) { return e->sx_desc_now(); }

// The LSD passes over `bits` bits from bit `shift0` of the records of n_slots slots (the digits of the first pass have been
// counted by whoever wrote the records). *cur: which of rec[0] / rec[1] holds the result.
template <typename T>
int sx_sort(fsk_engine* e, SxScratch& S, hipStream_t stream, T* const rec[2], uint32_t nfeat, uint32_t tps, uint32_t n_slots, int shift0,
            int bits, int* cur_out) {
    const int passes = (bits + 7) / 8;
    // the bits split evenly over the passes: 19 bits sort as 7 + 6 + 6, not 8 + 8 + 3 (a ballot per bit and record)
    auto pass_bits = [&](int p) { return bits / passes + (p < bits % passes ? 1 : 0); };
    int cur = 0;
    for (int p = 0, shift = shift0; p < passes; shift += pass_bits(p), ++p) {
        const int nbits = pass_bits(p);
        if (p > 0)
            FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_hist<T>), dim3(tps, n_slots), dim3(256), 0, stream, (const T*)rec[cur], nfeat, tps, shift,
                       (1u << nbits) - 1u, S.d_blockhist.p);
        FSK_LAUNCH(fsk::k_sx_scan_slot, dim3(n_slots), dim3(1024), 0, stream, S.d_blockhist.p, tps, S.d_totals.p);
        {   // (function pointers: a template-id with a comma cannot pass through the launch macro)
            auto k_scatter = nbits <= 4 ? fsk::k_sx_scatter<T, 4> : nbits == 5 ? fsk::k_sx_scatter<T, 5>
                             : nbits == 6 ? fsk::k_sx_scatter<T, 6> : nbits == 7 ? fsk::k_sx_scatter<T, 7>
                                                                                 : fsk::k_sx_scatter<T, 8>;
            FSK_LAUNCH(k_scatter, dim3(fsk::xcd_grid(tps * n_slots)), dim3(256), 0, stream, (const T*)rec[cur], rec[cur ^ 1], nfeat, tps, n_slots,
                       shift, nbits, (const uint32_t*)S.d_blockhist.p, (const uint32_t*)S.d_totals.p);
        }
        cur ^= 1;
        e->st.launches += 3;
    }
    *cur_out = cur;
    return FSK_OK;
}
inline int sx_first_pass_bits(int bits) { const int passes = (bits + 7) / 8; return bits / passes + (bits % passes ? 1 : 0); }
inline int sx_bits_below(u64 v) { int b = 0; while (b < 64 && ((u64)1 << b) < v) ++b; return b; }  // bits that hold 0 .. v - 1

// Shared prefixes (k_sx_group_tables in fsk_sparse_kernels.inc): how many leading kept positions the slots of this batch
// sort once per group. 0: none (every slot sorts its whole key).
struct SxShare { int share = 0, topbits = 0, lowbits = 0, wb = 0; uint32_t groups = 0; bool pre64 = false; };
SxShare sx_plan_share(fsk_engine* e, const int32_t* combos, int nb, int recbits_max) {
    SxShare best;
    const int k = e->k, want = (int)e->tune.sparse_share;
    if (want < 0 || k < 2 || nb < 2 || !e->win_words || e->nfeat < 2) return best;
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0;
    // (the presort is half a dozen launches: a batch below 2^24 records does not win them back)
    if (want == 0 && (u64)nb * (u64)e->nfeat < ((u64)1 << 24)) return best;
    // groups[s] = runs of consecutive slots with the same first s kept positions
    std::vector<uint32_t> groups((size_t)k, 1u);
    for (int i = 1; i < nb; ++i) {
        const unsigned char *a = &e->all_pos[(size_t)combos[i] * k], *b = &e->all_pos[(size_t)combos[i - 1] * k];
        int lcp = 0;
        while (lcp < k && a[lcp] == b[lcp]) ++lcp;
        for (int s = lcp + 1; s < k; ++s) groups[s] += 1u;
    }
    const int wb = std::max(1, sx_bits_below((u64)e->nfeat));
    auto passes = [](int bits) { return (bits + 7) / 8; };
    // In units of one scatter pass over a slot's records. Measured on config 4 (one MI355X, 250 slots of 460 K windows a batch,
    // profiles/r05_shared_prefix_ab.txt): a slot's scatter pass 0.74 us, its histogram 0.33, its extraction 0.6; a GROUP costs
    // its gather (20 bytes written, 24 read at random per window: 9 passes' worth), its extraction (2.2; twice that with 8-byte
    // presort records), ~1.65 per presort pass, and 2.4 for the window loads its slots no longer share with their neighbours.
    auto cost_slot = [&](int bits) { return 0.8 + passes(bits) + 0.45 * (passes(bits) - 1); };
    const double plain = nb * cost_slot(e->sx_keybits);
    double best_cost = want > 0 ? 1e300 : 0.93 * plain;
    for (int s = 1; s < k; ++s) {
        if (want > 0 && s != std::min(want, k - 1)) continue;
        int tb, lb;
        if (e->sx_symbits) { tb = s * e->sx_symbits; lb = (k - s) * e->sx_symbits; }
        else {
            u64 tv = 1, lv = 1;
            for (int c = 0; c < s; ++c) tv *= e->sigma;
            for (int c = s; c < k; ++c) lv *= e->sigma;
            tb = sx_bits_below(tv); lb = sx_bits_below(lv);
        }
        if (tb < 1 || lb < 1 || tb + lb + e->sx_sb > recbits_max || tb + wb > 64) continue;
        // (the presort's scratch: the windows and a part record of every group — never more than a quarter of what is free)
        if (mem_free && (double)groups[s] * (double)e->nfeat * (4.0 * e->win_words + recbits_max / 8.0) > 0.25 * (double)mem_free) continue;
        const bool pre64 = tb + wb > 32;
        const double c = nb * cost_slot(lb) + groups[s] * (13.6 + (pre64 ? 2.2 : 0.0) + 1.65 * passes(tb));
        if (c < best_cost) {
            best_cost = c;
            best.share = s; best.topbits = tb; best.lowbits = lb; best.wb = wb; best.groups = groups[s]; best.pre64 = pre64;
        }
    }
    return best;
}

// `pos_pin` / `stat_pin`: pinned staging of this batch (positions in, {pairs, words} out), untouched by
// anyone else until the batch's counts have been read. `guard_cap` == 0: the call waits for the
// counts and sizes the streams exactly; else it only enqueues, for streams of at most guard_cap words.
template <typename RecT>
int sparse_batch(fsk_engine* e, const int32_t* combos, int nb, u64* K, int64_t row0, int64_t row1, u64 slot_stride,
                 unsigned char* pos_pin, u64* stat_pin, u64 guard_cap, int lane, size_t pos_off = 0, hipEvent_t k_wait = nullptr,
                 hipEvent_t k_done = nullptr) {
    // pos_off: where this batch's positions lie in e->d_pos (the batches of one exact accumulate, in flight in two lanes, each
    // have their own piece); k_wait / k_done: the kernels that touch K wait for the first and are followed by the second (a
    // band with one part adds into K with a plain read-modify-write: the consume passes of the two lanes may not overlap)
    SxScratch& S = e->sxs[lane];
    hipStream_t stream = lane ? e->lane_stream : e->stream;
    const uint32_t nfeat = (uint32_t)e->nfeat;
    const size_t nrec = (size_t)nb * nfeat;
    if (nrec == 0) return FSK_OK;
    // Only the k-mer bits are sorted: the records of a slot are generated in sequence order and every
    // LSD pass is stable, so equal k-mers end up contiguous with their sequence ids ascending.
    const int sb = e->sx_sb;
    // (consecutive combos keep the same leading positions for long stretches: those are sorted once per group)
    const SxShare sh = nb > 16 ? sx_plan_share(e, combos, nb, 8 * (int)sizeof(RecT)) : SxShare();
    const int keybits = sh.share ? sh.lowbits : e->sx_keybits;  // what every slot sorts
    e->sx_share_used = sh.share;
    e->sx_share_groups = sh.groups;
    if (e->trace())
        fprintf(stderr, "[fsk] sparse batch: %d slots, shared leading positions %d (%u groups, %d + %d key bits; plain %d)\n", nb, sh.share, sh.groups,
                sh.topbits, sh.lowbits, e->sx_keybits);
    const int passes = (keybits + 7) / 8;
    const uint32_t dmask = (1u << sx_first_pass_bits(keybits)) - 1u;  // (the extraction counts the first pass's digits)
    const uint32_t tps = (nfeat + fsk::SX_TILE - 1) / fsk::SX_TILE;   // sort tiles per slot
    const uint32_t tpg = (nfeat + fsk::SG_TILE - 1) / fsk::SG_TILE;   // segment tiles per slot
    const uint32_t ntiles = tpg * (uint32_t)nb;
    // the form of the update stage (sx_choose_form); a batch of slot triangles (variance mode) knows the owner bands only
    const int form = slot_stride != 0 ? ((e->sx_lists && e->sx_form != 1) ? 0 : 1) : e->sx_form;
    const bool blocks = form == 2;
    const bool lists = form == 0;
    e->sx_form_used = form;
    const int pairs = lists && e->sx_pairs ? 1 : 0;  // (unit products as bare cells, two to a word: plan_owner_bands)
    const bool slot16 = slot_stride != 0 && e->sx_slot16_used;  // (u16 slot triangles: set by accumulate_sparse for a deferred batch)
    const uint32_t O = blocks ? (uint32_t)e->tune.blocks_max_bands : e->n_owners;  // (blocks: the most bands a pass can have)
    // descriptors (owner bands only): entries of more than short_max partners leave k_sx_emit as one descriptor each and
    // k_sx_consume walks their partners; the count matrix and the stream offsets then have two columns a band
    const uint32_t desc = !sx_desc_wanted(e) ? 0u : lists ? 1u : (blocks && e->tune.sparse_desc_blocks) ? 2u : 0u;  // (1: one descriptor an entry; 2: one per sub-band)
    const uint32_t short_max = desc && e->tune.sparse_desc_min > 0 ? (uint32_t)std::min<int64_t>(e->tune.sparse_desc_min, (int64_t)fsk::SX_SHORT) : fsk::SX_SHORT;
    const uint32_t OC = desc ? 2u * O : O;  // columns
    e->sx_desc_used = desc != 0;
    // (the presort's records, 4 or 8 bytes a window and group, go through the same two buffers first)
    const size_t pre_bytes = sh.share ? (size_t)sh.groups * nfeat * (sh.pre64 ? 8 : 4) : 0;
    for (int b = 0; b < 2; ++b) FSK_HIP(S.d_keys[b].reserve(std::max(pre_bytes, nrec * sizeof(RecT))));
    FSK_HIP(S.d_blockhist.reserve((size_t)256 * tps * nb));
    FSK_HIP(S.d_totals.reserve((size_t)256 * nb));
    FSK_HIP(S.d_tile_ent.reserve(ntiles));
    FSK_HIP(S.d_tile_lrh.reserve(ntiles));
    FSK_HIP(S.d_tile_rs.reserve(ntiles));
    FSK_HIP(S.d_ebase.reserve((size_t)ntiles + 1));
    FSK_HIP(S.d_E.reserve(nrec + 2));  // (+ 16 bytes: the descriptors' partners are read in whole 16-byte pieces)
    FSK_HIP(S.d_Pk.reserve(nrec));
    // skip_test_block: test rows pair only with the train entries of their runs (and themselves)
    const uint32_t skip_from = e->cfg.skip_test_block && e->n_test > 0 ? (uint32_t)e->n_train : 0xffffffffu;
    const bool skipping = skip_from != 0xffffffffu;
    if (skipping) {
        FSK_HIP(S.d_Tk.reserve(nrec));
        FSK_HIP(S.d_tile_lth.reserve(ntiles));
        FSK_HIP(S.d_tile_ts.reserve(ntiles));
    }
    FSK_HIP(S.d_sxstat.reserve(3));
    FSK_HIP(S.d_tile_stat.reserve((size_t)2 * ntiles));
    FSK_HIP(e->d_pos.reserve(pos_off + (size_t)nb * e->k));
    if (!e->owner_ready) {
        FSK_HIP(e->d_owner_r0.reserve(e->h_owner_r0.size()));
        // (once per set of sequences, and every lane's kernels read it: a synchronous copy)
        FSK_HIP(hipMemcpy(e->d_owner_r0.p, e->h_owner_r0.data(), e->h_owner_r0.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        e->owner_ready = true;
    }
    // tiles per chunk of the column scans: 64, fewer for a batch of few tiles (variance mode's: a few thousand) so that the
    // chunk kernels have a few hundred workgroups and short chains of dependent steps
    const uint32_t uc = ntiles >= 16384u ? (uint32_t)fsk::UC_CHUNK : ntiles >= 4096u ? 16u : 8u;
    const uint32_t nchunks = (ntiles + uc - 1) / uc;
    if (lists || blocks) {
        FSK_HIP(S.d_ucount.reserve((size_t)OC * ntiles));
        FSK_HIP(S.d_uchunk.reserve((size_t)OC * nchunks));
        FSK_HIP(S.d_utot.reserve(OC));
        FSK_HIP(S.d_list_off.reserve((size_t)OC + 1));
        FSK_HIP(S.d_part_base.reserve((size_t)O + 2));
    }
    fsk::SxIds ids{};
    const bool by_id = nb <= 16;  // (variance mode: a handful of combos per batch) positions from the resident table
    bool consecutive = !by_id;
    if (by_id) {
        if (!e->allpos_ready) {
            FSK_HIP(e->d_allpos.reserve(e->all_pos.size()));
            FSK_HIP(hipMemcpy(e->d_allpos.p, e->all_pos.data(), e->all_pos.size(), hipMemcpyHostToDevice));
            e->allpos_ready = true;
        }
        for (int s = 0; s < nb; ++s) ids.id[s] = combos[s];
    } else {
        // (a run of consecutive combo ids — every exact call — reads its kept positions from the resident table of all
        // combos; the batch's statistics are zeroed by the extraction kernel: no copy or fill command in front of a batch)
        for (int s = 1; s < nb && consecutive; ++s) consecutive = combos[s] == combos[0] + s;
        if (consecutive) {
            if (!e->allpos_ready) {
                FSK_HIP(e->d_allpos.reserve(e->all_pos.size()));
                FSK_HIP(hipMemcpy(e->d_allpos.p, e->all_pos.data(), e->all_pos.size(), hipMemcpyHostToDevice));
                e->allpos_ready = true;
            }
        } else {
            for (int s = 0; s < nb; ++s)
                memcpy(pos_pin + (size_t)s * e->k, &e->all_pos[(size_t)combos[s] * e->k], e->k);
            FSK_HIP(hipMemcpyAsync(e->d_pos.p + pos_off, pos_pin, (size_t)nb * e->k, hipMemcpyHostToDevice, stream));
        }
    }

    RecT* rec[2] = {(RecT*)S.d_keys[0].p, (RecT*)S.d_keys[1].p};

    e->tic(stream);
    const uint8_t* const pos_tab = by_id ? (const uint8_t*)e->d_allpos.p
                                   : consecutive ? (const uint8_t*)e->d_allpos.p + (size_t)combos[0] * e->k : (const uint8_t*)e->d_pos.p + pos_off;
    u64* const zeroed_stats = S.d_sxstat.p;
    const int ww = e->win_words;
    constexpr bool R32 = sizeof(RecT) == 4;
    const bool small = R32 && e->V <= ((u64)1 << 24) && e->sigma < (1u << 24);  // (the k-mer and every prefix of it fit 24 bits)
    fsk::SxSrc src{};
    src.win = e->d_win.p; src.feat_seq = e->d_featseq.p; src.combo_pos = pos_tab;
    src.k = e->k; src.sb = sb; src.bits = e->bits; src.by_id = by_id ? 1 : 0; src.sigma = e->sigma; src.symbits = e->sx_symbits; src.ids = ids;
    src.c0 = 0;
    if (sh.share) {
        // the presort: the windows in the order of their leading part, once per group of slots; every slot then extracts from
        // its group's copy and sorts by the rest of its key alone
        const uint32_t G = sh.groups;
        FSK_HIP(S.d_group_of.reserve((size_t)nb));
        FSK_HIP(S.d_group_head.reserve((size_t)nb));
        FSK_HIP(S.d_winp.reserve((size_t)G * nfeat * ww));
        FSK_HIP(S.d_part.reserve((size_t)G * nfeat * sizeof(RecT)));
        FSK_LAUNCH(fsk::k_sx_group_tables, dim3(1), dim3(1024), 0, stream, src, (uint32_t)nb, sh.share, S.d_group_of.p, S.d_group_head.p);
        const uint32_t dmask_top = (1u << sx_first_pass_bits(sh.topbits)) - 1u;
        const dim3 ggrid((nfeat + 255u) / 256u, G);
        RecT* const part = reinterpret_cast<RecT*>(S.d_part.p);
        int rcs = FSK_OK, pcur = 0;
        if (!sh.pre64) {
            uint32_t* pre[2] = {(uint32_t*)S.d_keys[0].p, (uint32_t*)S.d_keys[1].p};
            auto k_ge = ww == 2 ? (small ? fsk::k_sx_group_extract<uint32_t, 2, true> : fsk::k_sx_group_extract<uint32_t, 2, false>)
                                : (small ? fsk::k_sx_group_extract<uint32_t, 4, true> : fsk::k_sx_group_extract<uint32_t, 4, false>);
            FSK_LAUNCH(k_ge, dim3(tps, G), dim3(256), 0, stream, src, (const uint32_t*)S.d_group_head.p, sh.share, sh.wb, nfeat, tps, pre[0],
                       S.d_blockhist.p, dmask_top);
            rcs = sx_sort<uint32_t>(e, S, stream, pre, nfeat, tps, G, sh.wb, sh.topbits, &pcur);
            if (rcs) return rcs;
            auto k_gg = ww == 2 ? fsk::k_sx_group_gather<uint32_t, RecT, 2> : fsk::k_sx_group_gather<uint32_t, RecT, 4>;
            FSK_LAUNCH(k_gg, ggrid, dim3(256), 0, stream, (const uint32_t*)pre[pcur], (const uint32_t*)e->d_win.p, (const uint32_t*)e->d_featseq.p,
                       nfeat, sh.wb, sh.lowbits + sb, S.d_winp.p, part);
        } else {
            u64* pre[2] = {(u64*)S.d_keys[0].p, (u64*)S.d_keys[1].p};
            auto k_ge = ww == 2 ? fsk::k_sx_group_extract<u64, 2, false> : fsk::k_sx_group_extract<u64, 4, false>;
            FSK_LAUNCH(k_ge, dim3(tps, G), dim3(256), 0, stream, src, (const uint32_t*)S.d_group_head.p, sh.share, sh.wb, nfeat, tps, pre[0],
                       S.d_blockhist.p, dmask_top);
            rcs = sx_sort<u64>(e, S, stream, pre, nfeat, tps, G, sh.wb, sh.topbits, &pcur);
            if (rcs) return rcs;
            auto k_gg = ww == 2 ? fsk::k_sx_group_gather<u64, RecT, 2> : fsk::k_sx_group_gather<u64, RecT, 4>;
            FSK_LAUNCH(k_gg, ggrid, dim3(256), 0, stream, (const u64*)pre[pcur], (const uint32_t*)e->d_win.p, (const uint32_t*)e->d_featseq.p,
                       nfeat, sh.wb, sh.lowbits + sb, S.d_winp.p, part);
        }
        e->st.launches += 3;
        src.win = S.d_winp.p;
        src.c0 = sh.share;
        const bool four = e->tune.extract_slots ? e->tune.extract_slots == 4 : (u64)tps * (u64)nb >= 8192;
#define FSK_EXTRACT_SHARED(SPW)                                                                                              \
    (ww == 2 ? (small ? fsk::k_sx_extract_shared<RecT, 2, R32, SPW> : fsk::k_sx_extract_shared<RecT, 2, false, SPW>)          \
             : (small ? fsk::k_sx_extract_shared<RecT, 4, R32, SPW> : fsk::k_sx_extract_shared<RecT, 4, false, SPW>))
        auto k_ex = four ? FSK_EXTRACT_SHARED(4) : FSK_EXTRACT_SHARED(1);
#undef FSK_EXTRACT_SHARED
        FSK_LAUNCH(k_ex, dim3(tps, four ? ((uint32_t)nb + 3u) / 4u : (uint32_t)nb), dim3(256), 0, stream, src, (const uint32_t*)S.d_group_of.p,
                   (const RecT*)part, nfeat, tps, (uint32_t)nb, rec[0], S.d_blockhist.p, dmask, zeroed_stats);
    } else if (ww) {
        // (four slots per workgroup share the window loads when that still leaves a few thousand workgroups)
        const bool four = e->tune.extract_slots ? e->tune.extract_slots == 4 : (u64)tps * (u64)nb >= 8192;
#define FSK_EXTRACT_WIN(SPW)                                                                                          \
    (ww == 2 ? (small ? fsk::k_sx_extract_win<RecT, 2, R32, SPW> : fsk::k_sx_extract_win<RecT, 2, false, SPW>)         \
             : (small ? fsk::k_sx_extract_win<RecT, 4, R32, SPW> : fsk::k_sx_extract_win<RecT, 4, false, SPW>))
        auto k_ex = four ? FSK_EXTRACT_WIN(4) : FSK_EXTRACT_WIN(1);
#undef FSK_EXTRACT_WIN
        FSK_LAUNCH(k_ex, dim3(tps, four ? ((uint32_t)nb + 3u) / 4u : (uint32_t)nb), dim3(256), 0, stream, src, nfeat, tps, (uint32_t)nb, rec[0],
                   S.d_blockhist.p, dmask, zeroed_stats);
    } else {
        FSK_LAUNCH(HIP_KERNEL_NAME(fsk::k_sx_extract<RecT>), dim3(tps, nb), dim3(256), 0, stream, e->view(), e->d_featseq.p,
                   e->d_fstart.p, nfeat, tps, e->k, e->sigma, sb, pos_tab, rec[0], S.d_blockhist.p, dmask, ids, zeroed_stats, by_id ? 1 : 0, e->sx_symbits);
    }
    e->toc(&e->st.ms_extract, stream);
    e->st.launches += 1;

    e->tic(stream);
    int cur = 0;
    {
        const int rcs = sx_sort<RecT>(e, S, stream, rec, nfeat, tps, (uint32_t)nb, sb, keybits, &cur);
        if (rcs) return rcs;
    }
    e->toc(&e->st.ms_sort, stream);
    e->st.sort_records += nrec;
    e->st.sort_passes = passes;

    e->tic(stream);
    const uint32_t maxprod = (1u << e->sx_pb) - 1u;
    const uint32_t cmax = maxprod / std::max<uint32_t>(1u, e->maxW);  // multiplicities up to here: one word per pair
    {
        auto k_cnt = skipping ? fsk::k_sx_seg_count<RecT, true> : fsk::k_sx_seg_count<RecT, false>;
        FSK_LAUNCH(k_cnt, dim3(tpg, nb), dim3(256), 0, stream, (const RecT*)rec[cur], nfeat, tpg, sb, S.d_tile_ent.p, S.d_tile_lrh.p, skip_from,
                   skipping ? S.d_tile_lth.p : (int*)nullptr);
    }
    {
        const int* lth = skipping ? (const int*)S.d_tile_lth.p : (const int*)nullptr;
        int* ts = skipping ? S.d_tile_ts.p : (int*)nullptr;
        if (ntiles <= 4096u && !e->tune.seg_scan_chunked) {  // one workgroup walks the tile records
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(1), dim3(1024), 0, stream, (const uint32_t*)S.d_tile_ent.p, (const int*)S.d_tile_lrh.p, ntiles,
                       S.d_ebase.p, S.d_tile_rs.p, lth, ts, (const uint32_t*)nullptr, (const int*)nullptr, (const int*)nullptr,
                       (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr);
        } else {  // chunk totals, the same scan over the chunk records, the chunks with their carries
            const uint32_t nch = (ntiles + 1023u) / 1024u;
            FSK_HIP(S.d_segc.reserve((size_t)6 * (nch + 1)));
            uint32_t* c_tot = S.d_segc.p;
            int* c_lrh = reinterpret_cast<int*>(c_tot + (nch + 1));
            int* c_lth = c_lrh + (nch + 1);
            uint32_t* c_ex = reinterpret_cast<uint32_t*>(c_lth + (nch + 1));
            int* c_h = reinterpret_cast<int*>(c_ex + (nch + 1));
            int* c_t = c_h + (nch + 1);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(nch), dim3(1024), 0, stream, (const uint32_t*)S.d_tile_ent.p, (const int*)S.d_tile_lrh.p, ntiles,
                       (uint32_t*)nullptr, (int*)nullptr, lth, (int*)nullptr, (const uint32_t*)nullptr, (const int*)nullptr, (const int*)nullptr,
                       c_tot, c_lrh, c_lth);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(1), dim3(1024), 0, stream, (const uint32_t*)c_tot, (const int*)c_lrh, nch, c_ex, c_h,
                       skipping ? (const int*)c_lth : (const int*)nullptr, skipping ? c_t : (int*)nullptr, (const uint32_t*)nullptr,
                       (const int*)nullptr, (const int*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr);
            FSK_LAUNCH(fsk::k_sx_seg_scan, dim3(nch), dim3(1024), 0, stream, (const uint32_t*)S.d_tile_ent.p, (const int*)S.d_tile_lrh.p, ntiles,
                       S.d_ebase.p, S.d_tile_rs.p, lth, ts, (const uint32_t*)c_ex, (const int*)c_h, (const int*)c_t, (uint32_t*)nullptr,
                       (int*)nullptr, (int*)nullptr);
            e->st.launches += 2;
        }
    }
    // entries in the packed format (4 + 2 + 2 bytes) when sequence ids, multiplicities and ranks fit 16 bits
    const bool packed = e->N < 65535 && e->maxW < 65536u && !e->tune.sparse_unpacked;
    // descriptors: the entries' column array (what the partners are read from: 2 bytes when N < 32768, else 4)
    // (col16 = bits of a sequence id << 1 | two-byte columns: what is left of the 16 or 32 bits holds the multiplicity)
    const int colbits = std::max(1, sx_bits_below((u64)e->N));
    const int col16 = (colbits << 1) | (e->N < 32768 && e->tune.sparse_desc_cols == 3 ? 1 : 0);
    void* colp = nullptr;
    if (desc && (e->tune.sparse_desc_cols >= 2 || (e->tune.sparse_desc_cols == 1 && !packed))) {
        FSK_HIP(S.d_cols.reserve(((col16 & 1) ? (nrec + 1) / 2 : nrec) + 4));  // (+ 16 bytes: a lane's last load reads whole 16-byte pieces)
        colp = (void*)S.d_cols.p;
    }
    if (blocks) {
        // ---- the two-level form (fsk_sparse_blocks.inc): passes over disjoint row ranges, each sized exactly (a pass is
        // milliseconds of work: the wait for its word count does not show). A range that does not fit one pass — more cells
        // than 2^32 or than its bands cover, more words than 32-bit offsets address — is halved by cells.
        const u64 pass_words = e->tune.blocks_pass_words > 0 ? (u64)e->tune.blocks_pass_words : ((u64)1 << 31);
        const u64 total_cells = (u64)e->N * ((u64)e->N + 1) / 2;
        stat_pin[0] = stat_pin[1] = 0;
        e->toc(&e->st.ms_segment, stream);
        e->tic(stream);
        std::vector<std::pair<int64_t, int64_t>> todo;
        todo.emplace_back(row0, row1);
        u64 batch_pairs = 0;  // (the passes' += in all: what decides about descriptors for the batches that follow)
        bool first_pass = true;
        while (!todo.empty()) {
            const int64_t ra = todo.back().first, rb = todo.back().second;
            todo.pop_back();
            if (rb <= ra) continue;
            auto halve = [&]() {  // by cells: the row whose first cell is the middle one
                const u64 ca = (u64)ra * ((u64)ra + 1) / 2, cb = (u64)rb * ((u64)rb + 1) / 2, mid = ca + (cb - ca) / 2;
                int64_t lo = ra + 1, hi = rb - 1;
                while (lo < hi) {
                    const int64_t m = (lo + hi) / 2;
                    if ((u64)m * ((u64)m + 1) / 2 < mid) lo = m + 1; else hi = m;
                }
                todo.emplace_back(lo, rb);  // (the lower rows first: the stack pops them next)
                todo.emplace_back(ra, lo);
            };
            SxPass P;
            // (what the range is expected to emit, by its share of the triangle: a range that would overflow a pass is not tried)
            const u64 cells = (u64)rb * ((u64)rb + 1) / 2 - (u64)ra * ((u64)ra + 1) / 2;
            const bool too_many = e->sx_wpr != 0 && rb - ra > 1 &&
                                  (double)e->sx_words_of(nrec) * ((double)cells / (double)std::max<u64>(1, total_cells)) > 0.9 * (double)pass_words;
            if (too_many || !blocks_plan_pass(e, ra, rb, &P)) {
                if (rb - ra <= 1) return e->fail(FSK_EUNSUPPORTED, "sparse dataflow: row %lld does not fit one pass of the two-level form", (long long)ra);
                halve();
                continue;
            }
            const uint32_t Op = P.n_owners;
            {   // (a pass of ONE row longer than blocks_max_bands bands cover has more bands than that — up to SX_MAX_OWNERS —: the
                // count matrix and the offsets by this pass's own bands; found by tools/stress_parity.py's blocks cases as a
                // memory fault at N = 6500 with seven bands of 2^9 cells a pass. The previous pass has been waited for.)
                const size_t cols = (size_t)(desc ? 2u : 1u) * std::max(Op, O);
                FSK_HIP(S.d_ucount.reserve(cols * ntiles));
                FSK_HIP(S.d_uchunk.reserve(cols * nchunks));
                FSK_HIP(S.d_utot.reserve(cols));
                FSK_HIP(S.d_list_off.reserve(cols + 1));
                FSK_HIP(S.d_part_base.reserve((size_t)std::max(Op, O) + 2));
            }
            FSK_HIP(e->d_blk_r0.reserve((size_t)fsk::SX_MAX_OWNERS + 1));
            // (the previous pass has been waited for: nothing reads the table any more)
            FSK_HIP(hipMemcpy(e->d_blk_r0.p, P.r0.data(), P.r0.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            if (!first_pass) FSK_HIP(hipMemsetAsync(S.d_sxstat.p, 0, 3 * sizeof(u64), stream));  // (the first pass: zeroed by the extraction kernel)
            first_pass = false;
            const uint32_t maxprod_p = (1u << P.pb) - 1u, cmax_p = maxprod_p / std::max<uint32_t>(1u, e->maxW);
            u64 pass_stat[2] = {0, 0};
            // stat_pin is pinned host memory the device writes: this pass's totals land there, the batch's are summed below
            u64* const pin = stat_pin;
            pin[0] = pin[1] = 0;
            if (packed) {
                auto k_seg = desc ? (skipping ? fsk::k_sx_seg_write<RecT, true, false, true, true> : fsk::k_sx_seg_write<RecT, true, false, false, true>)
                                  : (skipping ? fsk::k_sx_seg_write<RecT, true, false, true> : fsk::k_sx_seg_write<RecT, true, false, false>);
                FSK_LAUNCH(k_seg, dim3(tpg, nb), dim3(256), 0, stream, (const RecT*)rec[cur], nfeat, tpg, sb, (const uint32_t*)S.d_ebase.p,
                           (const int*)S.d_tile_rs.p, reinterpret_cast<uint32_t*>(S.d_E.p), reinterpret_cast<uint16_t*>(S.d_Pk.p), P.t, Op,
                           S.d_ucount.p, (uint32_t)ra, (uint32_t)rb, e->maxW, maxprod_p, cmax_p, S.d_tile_stat.p, skip_from,
                           skipping ? (const int*)S.d_tile_ts.p : (const int*)nullptr,
                           skipping ? reinterpret_cast<uint16_t*>(S.d_Tk.p) : (uint16_t*)nullptr, P.own_base, short_max, desc, P.sub_shift, colp, col16);
            } else {
                auto k_seg = desc ? (skipping ? fsk::k_sx_seg_write<RecT, false, false, true, true> : fsk::k_sx_seg_write<RecT, false, false, false, true>)
                                  : (skipping ? fsk::k_sx_seg_write<RecT, false, false, true> : fsk::k_sx_seg_write<RecT, false, false, false>);
                FSK_LAUNCH(k_seg, dim3(tpg, nb), dim3(256), 0, stream, (const RecT*)rec[cur], nfeat, tpg, sb, (const uint32_t*)S.d_ebase.p,
                           (const int*)S.d_tile_rs.p, S.d_E.p, S.d_Pk.p, P.t, Op, S.d_ucount.p, (uint32_t)ra, (uint32_t)rb, e->maxW, maxprod_p,
                           cmax_p, S.d_tile_stat.p, skip_from, skipping ? (const int*)S.d_tile_ts.p : (const int*)nullptr,
                           skipping ? S.d_Tk.p : (uint32_t*)nullptr, P.own_base, short_max, desc, P.sub_shift, colp, col16);
            }
            const uint32_t OCp = desc ? 2u * Op : Op, wcol = desc ? Op : 0u;  // (columns: descriptor streams first, then the words')
            FSK_LAUNCH(fsk::k_sx_ucol_sum, dim3(nchunks), dim3(256), 0, stream, (const uint32_t*)S.d_ucount.p, ntiles, OCp, S.d_uchunk.p,
                       (const u64*)S.d_tile_stat.p, S.d_sxstat.p, pin, uc);
            FSK_LAUNCH(fsk::k_sx_ucol_scan, dim3(OCp), dim3(256), 0, stream, S.d_uchunk.p, nchunks, OCp, S.d_utot.p);
            FSK_LAUNCH(fsk::k_sx_ucol_apply, dim3(nchunks), dim3(256), 0, stream, S.d_ucount.p, ntiles, OCp, (const uint32_t*)S.d_uchunk.p,
                       (const uint32_t*)S.d_utot.p, S.d_list_off.p, uc);
            e->st.launches += 4;
            FSK_HIP(hipStreamSynchronize(stream));
            pass_stat[0] = pin[0]; pass_stat[1] = pin[1];
            const u64 words = pass_stat[1];
            if (words >= pass_words && rb - ra > 1) {  // (does not fit 32-bit offsets with room to spare: the halves, each from its own count)
                halve();
                first_pass = false;
                continue;
            }
            if (words >= ((u64)1 << 32))
                return e->fail(FSK_EUNSUPPORTED, "sparse dataflow: row %lld alone emits %llu update words a batch", (long long)ra, (unsigned long long)words);
            e->u_extra += pass_stat[0];
            batch_pairs += pass_stat[0];
            e->sx_passes += 1;
            if (e->trace())
                fprintf(stderr, "[fsk] sparse blocks: pass rows [%lld, %lld), %u bands of 2^%d cells, %u sub-bands a band, %d product bits, %llu words\n",
                        (long long)ra, (long long)rb, Op, P.t, P.submax, P.pb, (unsigned long long)words);
            if (words == 0) continue;
            if ((size_t)words > S.d_ulist.cap) FSK_HIP(S.d_ulist.reserve((size_t)(words + words / 8)));
            if ((size_t)words > S.d_ulist2.cap) FSK_HIP(S.d_ulist2.reserve((size_t)(words + words / 8)));
            const size_t nsub = (size_t)Op * P.submax;
            FSK_HIP(S.d_subcnt.reserve(nsub));
            FSK_HIP(S.d_suboff.reserve(nsub));
            FSK_HIP(S.d_subcur.reserve(nsub));
            FSK_HIP(hipMemsetAsync(S.d_subcnt.p, 0, nsub * sizeof(uint32_t), stream));
            if (k_wait) { FSK_HIP(hipStreamWaitEvent(stream, k_wait, 0)); k_wait = nullptr; }
            if (packed) {
                auto k_emit = desc ? (skipping ? fsk::k_sx_emit<false, true, true, true> : fsk::k_sx_emit<false, false, true, true>)
                                   : (skipping ? fsk::k_sx_emit<false, true, true> : fsk::k_sx_emit<false, false, true>);
                FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, reinterpret_cast<const uint32_t*>(S.d_E.p),
                           reinterpret_cast<const uint16_t*>(S.d_Pk.p), (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_blk_r0.p, P.t, Op,
                           (const uint32_t*)S.d_list_off.p, (const uint32_t*)S.d_ucount.p, S.d_ulist.p, (uint32_t)ra, (uint32_t)rb, e->maxW,
                           maxprod_p, cmax_p, P.pb, K, tpg, (u64)0,
                           skipping ? reinterpret_cast<const uint16_t*>(S.d_Tk.p) : (const uint16_t*)nullptr, (const u64*)S.d_sxstat.p, ~(u64)0,
                           ntiles, 0, P.own_base, short_max, desc, P.sub_shift);
            } else {
                auto k_emit = desc ? (skipping ? fsk::k_sx_emit<false, true, false, true> : fsk::k_sx_emit<false, false, false, true>)
                                   : (skipping ? fsk::k_sx_emit<false, true, false> : fsk::k_sx_emit<false, false, false>);
                FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, (const uint2*)S.d_E.p, (const uint32_t*)S.d_Pk.p,
                           (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_blk_r0.p, P.t, Op, (const uint32_t*)S.d_list_off.p,
                           (const uint32_t*)S.d_ucount.p, S.d_ulist.p, (uint32_t)ra, (uint32_t)rb, e->maxW, maxprod_p, cmax_p, P.pb, K, tpg, (u64)0,
                           skipping ? (const uint32_t*)S.d_Tk.p : (const uint32_t*)nullptr, (const u64*)S.d_sxstat.p, ~(u64)0, ntiles, 0, P.own_base, short_max, desc, P.sub_shift);
            }
            // (persistent launches: two workgroups of 1024 threads a CU walk the tiles of the bands' streams in contiguous chunks)
            const uint32_t n_tiles_max = Op + (uint32_t)((words + fsk::SXB_TILE - 1) / fsk::SXB_TILE);
            const uint32_t n_split = std::min<uint32_t>(n_tiles_max, 2u * (uint32_t)std::max(1, e->n_cu));
            const uint32_t* const w_off = (const uint32_t*)S.d_list_off.p + wcol;  // where the bands' word streams start
            FSK_LAUNCH(fsk::k_sx_parts, dim3(1), dim3(512), 0, stream, w_off, Op, (uint32_t)fsk::SXB_TILE, S.d_part_base.p,
                       (const u64*)S.d_sxstat.p, ~(u64)0, 0u, (uint32_t*)nullptr, 0xffffffffu, 1u);
            FSK_LAUNCH(fsk::k_sxb_count, dim3(n_split), dim3(fsk::SXB_THREADS), 0, stream, (const uint32_t*)S.d_ulist.p, w_off,
                       (const uint32_t*)S.d_part_base.p, Op, P.pb, P.sub_shift, P.submax, S.d_subcnt.p);
            FSK_LAUNCH(fsk::k_sxb_scan, dim3(Op), dim3(fsk::SXB_THREADS), 0, stream, (const uint32_t*)S.d_subcnt.p, w_off, P.submax,
                       S.d_suboff.p, S.d_subcur.p, 0);
            if (desc) {  // the descriptor records by (band, sub-band): count, scan, scatter (into the head of the second buffer, as in the first)
                FSK_HIP(S.d_dsubcnt.reserve(nsub));
                FSK_HIP(S.d_dsuboff.reserve(nsub));
                FSK_HIP(S.d_dsubcur.reserve(nsub));
                FSK_HIP(hipMemsetAsync(S.d_dsubcnt.p, 0, nsub * sizeof(uint32_t), stream));
                const dim3 dgrid((uint32_t)std::max(1, 4 * e->n_cu / (int)std::max(1u, Op)) + 1u, Op);
                FSK_LAUNCH(fsk::k_sxb_drecords<false>, dgrid, dim3(1024), 0, stream, (const uint32_t*)S.d_ulist.p, (const uint32_t*)S.d_list_off.p, P.submax,
                           S.d_dsubcnt.p, (uint4*)nullptr);
                FSK_LAUNCH(fsk::k_sxb_scan, dim3(Op), dim3(fsk::SXB_THREADS), 0, stream, (const uint32_t*)S.d_dsubcnt.p, (const uint32_t*)S.d_list_off.p,
                           P.submax, S.d_dsuboff.p, S.d_dsubcur.p, 2);
                FSK_LAUNCH(fsk::k_sxb_drecords<true>, dgrid, dim3(1024), 0, stream, (const uint32_t*)S.d_ulist.p, (const uint32_t*)S.d_list_off.p, P.submax,
                           S.d_dsubcur.p, reinterpret_cast<uint4*>(S.d_ulist2.p));
                e->st.launches += 3;
            }
            {   // (workgroups of 256 threads, three a CU by their LDS; tuning blocks_scatter_threads: 512 / 1024 for the A/B)
                const int nt = e->tune.blocks_scatter_threads ? (int)e->tune.blocks_scatter_threads : 256;
                const uint32_t per_cu = nt == 1024 ? 2u : 3u;
                const uint32_t grid = std::min<uint32_t>(n_tiles_max, per_cu * (uint32_t)std::max(1, e->n_cu));
                auto k_sc = nt == 1024 ? fsk::k_sxb_scatter<1024> : nt == 512 ? fsk::k_sxb_scatter<512> : fsk::k_sxb_scatter<256>;
                FSK_LAUNCH(k_sc, dim3(grid), dim3((uint32_t)nt), 0, stream, (const uint32_t*)S.d_ulist.p, w_off,
                           (const uint32_t*)S.d_part_base.p, Op, P.pb, P.sub_shift, P.submax, S.d_subcur.p, S.d_ulist2.p);
            }
            const size_t lds_sub = sizeof(uint32_t) << P.sub_shift;
            {   // (a block of 2^13 cells and fewer: workgroups of 512 threads, four a CU)
                auto k_cs = P.sub_shift <= 13 ? fsk::k_sxb_consume<512> : fsk::k_sxb_consume<1024>;
                FSK_HIP(fsk_hw::allow_dynamic_lds(k_cs, lds_sub));
                FSK_LAUNCH(k_cs, dim3(P.submax, Op), dim3(P.sub_shift <= 13 ? 512u : 1024u), lds_sub, stream, (const uint32_t*)S.d_ulist2.p,
                           (const uint32_t*)S.d_suboff.p, (const uint32_t*)S.d_subcnt.p, (const uint32_t*)e->d_blk_r0.p, P.pb, P.sub_shift, P.submax, K,
                           desc ? reinterpret_cast<const uint4*>(S.d_ulist2.p) : (const uint4*)nullptr, (const uint32_t*)S.d_dsuboff.p,
                           (const uint32_t*)S.d_dsubcnt.p, (const void*)S.d_E.p, packed ? 1 : 0, (const void*)colp, col16);
            }
            e->st.launches += 6;
            FSK_HIP(hipStreamSynchronize(stream));  // (the next pass overwrites the band table, the entries' unit marks and the streams)
            e->sx_saw(words * std::max<u64>(1, total_cells / std::max<u64>(1, cells)), nrec);  // (words per record as if the whole triangle emitted at this rate)
        }
        if (k_wait) FSK_HIP(hipStreamWaitEvent(stream, k_wait, 0));
        if (row1 > row0)
            FSK_LAUNCH(fsk::k_sx_diag_windows, dim3((uint32_t)((row1 - row0 + 255) / 256), 1), dim3(256), 0, stream, (const uint32_t*)e->d_fstart.p,
                       (uint32_t)row0, (uint32_t)row1, (uint32_t)nb, K, (u64)0, (const u64*)nullptr, ~(u64)0, 0, (uint32_t*)nullptr);
        e->st.launches += 1;
        if (k_done) FSK_HIP(hipEventRecord(k_done, stream));
        e->toc(&e->st.ms_pairs, stream);
        FSK_HIP(hipGetLastError());
        stat_pin[0] = stat_pin[1] = 0;  // (every pass has been added to u_extra already)
        e->sx_saw_pairs(batch_pairs, nrec);
        return FSK_OK;
    }
    if (packed) {
        auto k_seg = desc ? (skipping ? (pairs ? fsk::k_sx_seg_write<RecT, true, true, true, true> : fsk::k_sx_seg_write<RecT, true, false, true, true>)
                                      : (pairs ? fsk::k_sx_seg_write<RecT, true, true, false, true> : fsk::k_sx_seg_write<RecT, true, false, false, true>))
                          : (skipping ? (pairs ? fsk::k_sx_seg_write<RecT, true, true, true> : fsk::k_sx_seg_write<RecT, true, false, true>)
                                      : (pairs ? fsk::k_sx_seg_write<RecT, true, true, false> : fsk::k_sx_seg_write<RecT, true, false, false>));
        FSK_LAUNCH(k_seg, dim3(tpg, nb), dim3(256), 0, stream, (const RecT*)rec[cur], nfeat, tpg, sb, (const uint32_t*)S.d_ebase.p,
                   (const int*)S.d_tile_rs.p, reinterpret_cast<uint32_t*>(S.d_E.p), reinterpret_cast<uint16_t*>(S.d_Pk.p), e->sx_own_shift, O,
                   lists ? S.d_ucount.p : (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, S.d_tile_stat.p,
                   skip_from, skipping ? (const int*)S.d_tile_ts.p : (const int*)nullptr,
                   skipping ? reinterpret_cast<uint16_t*>(S.d_Tk.p) : (uint16_t*)nullptr, 0u, short_max, desc, 0, colp, col16);
    } else {
        auto k_seg = desc ? (skipping ? (pairs ? fsk::k_sx_seg_write<RecT, false, true, true, true> : fsk::k_sx_seg_write<RecT, false, false, true, true>)
                                      : (pairs ? fsk::k_sx_seg_write<RecT, false, true, false, true> : fsk::k_sx_seg_write<RecT, false, false, false, true>))
                          : (skipping ? (pairs ? fsk::k_sx_seg_write<RecT, false, true, true> : fsk::k_sx_seg_write<RecT, false, false, true>)
                                      : (pairs ? fsk::k_sx_seg_write<RecT, false, true, false> : fsk::k_sx_seg_write<RecT, false, false, false>));
        FSK_LAUNCH(k_seg, dim3(tpg, nb), dim3(256), 0, stream, (const RecT*)rec[cur], nfeat, tpg, sb, (const uint32_t*)S.d_ebase.p,
                   (const int*)S.d_tile_rs.p, S.d_E.p, S.d_Pk.p, e->sx_own_shift, O,
                   lists ? S.d_ucount.p : (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, S.d_tile_stat.p,
                   skip_from, skipping ? (const int*)S.d_tile_ts.p : (const int*)nullptr, skipping ? S.d_Tk.p : (uint32_t*)nullptr, 0u, short_max, desc, 0, colp, col16);
    }
    stat_pin[0] = stat_pin[1] = 0;
    e->st.launches += 3;
    u64 words = 0;
    if (lists) {  // where every (tile, owner) share of the update streams starts (+ the batch's pair and word totals)
        FSK_LAUNCH(fsk::k_sx_ucol_sum, dim3(nchunks), dim3(256), 0, stream, (const uint32_t*)S.d_ucount.p, ntiles, OC, S.d_uchunk.p,
                   (const u64*)S.d_tile_stat.p, S.d_sxstat.p, stat_pin, uc);
        FSK_LAUNCH(fsk::k_sx_ucol_scan, dim3(OC), dim3(256), 0, stream, S.d_uchunk.p, nchunks, OC, S.d_utot.p);
        FSK_LAUNCH(fsk::k_sx_ucol_apply, dim3(nchunks), dim3(256), 0, stream, S.d_ucount.p, ntiles, OC, (const uint32_t*)S.d_uchunk.p,
                   (const uint32_t*)S.d_utot.p, S.d_list_off.p, uc);
        e->st.launches += 3;
    } else {
        FSK_LAUNCH(fsk::k_sx_stat_sum, dim3(32), dim3(256), 0, stream, (const u64*)S.d_tile_stat.p, ntiles, S.d_sxstat.p, stat_pin);
        e->st.launches += 1;
    }
    const bool guarded = guard_cap != 0;
    u64 cap_words = ~(u64)0;
    if (guarded) {
        words = lists ? std::max<u64>(1, std::min(e->sx_words_of(nrec), guard_cap)) : 0;  // (sizes the parts; the kernels read the true offsets)
        cap_words = guard_cap;
    } else {
        FSK_HIP(hipStreamSynchronize(stream));  // the update streams are sized exactly
        e->u_extra += stat_pin[0];
        words = stat_pin[1];
        e->sx_saw(words, nrec);
        e->sx_saw_pairs(stat_pin[0], nrec);
    }
    e->toc(&e->st.ms_segment, stream);

    e->tic(stream);
    const bool use_lists = lists && words < e->sx_max_words();
    if (slot_stride != 0 && !use_lists) return FSK_RETRY_UNGROUPED;  // (nothing of this batch has touched K yet)
    if (use_lists) {
        if (words > 0 || slot_stride != 0) {
            if (!guarded && (size_t)words > S.d_ulist.cap)  // (grown with headroom: the batches of a pass differ by a few percent)
                FSK_HIP(S.d_ulist.reserve((size_t)std::max<u64>(1, words + words / 4)));
            // (function pointers: a template-id with a comma cannot pass through the launch macro)
            if (packed) {
                auto k_emit = desc ? (skipping ? fsk::k_sx_emit<false, true, true, true> : fsk::k_sx_emit<false, false, true, true>)
                                   : (skipping ? fsk::k_sx_emit<false, true, true> : fsk::k_sx_emit<false, false, true>);
                FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, reinterpret_cast<const uint32_t*>(S.d_E.p),
                           reinterpret_cast<const uint16_t*>(S.d_Pk.p), (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_owner_r0.p,
                           e->sx_own_shift, O, (const uint32_t*)S.d_list_off.p, (const uint32_t*)S.d_ucount.p, S.d_ulist.p, (uint32_t)row0,
                           (uint32_t)row1, e->maxW, maxprod, cmax, e->sx_pb, K, tpg, slot_stride,
                           skipping ? reinterpret_cast<const uint16_t*>(S.d_Tk.p) : (const uint16_t*)nullptr, (const u64*)S.d_sxstat.p, cap_words, ntiles, pairs, 0u, short_max, desc, 0);
            } else {
                auto k_emit = desc ? (skipping ? fsk::k_sx_emit<false, true, false, true> : fsk::k_sx_emit<false, false, false, true>)
                                   : (skipping ? fsk::k_sx_emit<false, true, false> : fsk::k_sx_emit<false, false, false>);
                FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, (const uint2*)S.d_E.p, (const uint32_t*)S.d_Pk.p,
                           (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_owner_r0.p, e->sx_own_shift, O, (const uint32_t*)S.d_list_off.p,
                           (const uint32_t*)S.d_ucount.p, S.d_ulist.p, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, e->sx_pb, K, tpg,
                           slot_stride, skipping ? (const uint32_t*)S.d_Tk.p : (const uint32_t*)nullptr, (const u64*)S.d_sxstat.p, cap_words, ntiles, pairs, 0u, short_max, desc, 0);
            }
            const size_t lds = (size_t)e->sx_cap * sizeof(uint32_t), lds_slot = (size_t)e->sx_cap_slot * sizeof(uint32_t);
            FSK_HIP(fsk_hw::allow_dynamic_lds(fsk::k_sx_consume<false>, lds));
            FSK_HIP(fsk_hw::allow_dynamic_lds(fsk::k_sx_consume<true>, lds_slot));
            // parts of about `target` words: ~1024 workgroups, and never so short that the flush of a
            // part (up to sx_cap cells) outweighs the words it summed
            // (tuning sparse_parts_target: tests cut small inputs into several parts a band)
            const uint32_t target = e->tune.sparse_parts_target > 0 ? (uint32_t)e->tune.sparse_parts_target
                                                                    : (uint32_t)std::max<u64>((u64)4 * e->sx_cap, (words + 1023) / 1024);
            // (with descriptors k_sx_parts sets the target itself: about sparse_desc_parts parts, at most one more a band)
            const uint32_t desc_parts = (uint32_t)std::max<int64_t>(1, e->tune.sparse_desc_parts);
            const uint32_t max_parts = desc ? O + desc_parts + 9u : O + (uint32_t)(((guarded ? guard_cap : words) + target - 1) / target);
            const void* const Ep = (const void*)S.d_E.p;
            if (slot_stride == 0) FSK_HIP(S.d_part_base.reserve((size_t)O + 8 + (size_t)4 * max_parts));  // (bases, total, target, then 16 bytes a part)
            if (k_wait) FSK_HIP(hipStreamWaitEvent(stream, k_wait, 0));
            if (slot_stride != 0) {  // one triangle per slot: a slot's words of a stream are one contiguous piece
                if (slot16) {
                    auto k_cs16 = fsk::k_sx_consume<true, true>;
                    FSK_HIP(fsk_hw::allow_dynamic_lds(k_cs16, lds_slot));
                    FSK_LAUNCH(k_cs16, dim3(O, e->sx_rounds_slot, nb), dim3(fsk::CS_THREADS), lds_slot, stream, (const uint32_t*)S.d_ulist.p,
                               (const uint32_t*)S.d_list_off.p, (const uint32_t*)e->d_owner_r0.p, (const uint32_t*)nullptr, O, target,
                               e->sx_cap_slot, e->sx_pb, K, (const uint32_t*)S.d_ucount.p, tpg, slot_stride, (const u64*)S.d_sxstat.p, cap_words,
                               e->sx_ovf_now, pairs, desc, Ep, packed ? 1 : 0, (const void*)colp, col16);
                } else {
                    FSK_LAUNCH(fsk::k_sx_consume<true>, dim3(O, e->sx_rounds_slot, nb), dim3(fsk::CS_THREADS), lds_slot, stream, (const uint32_t*)S.d_ulist.p,
                               (const uint32_t*)S.d_list_off.p, (const uint32_t*)e->d_owner_r0.p, (const uint32_t*)nullptr, O, target,
                               e->sx_cap_slot, e->sx_pb, K, (const uint32_t*)S.d_ucount.p, tpg, slot_stride, (const u64*)S.d_sxstat.p, cap_words,
                               (uint32_t*)nullptr, pairs, desc, Ep, packed ? 1 : 0, (const void*)colp, col16);
                }
            } else {
                FSK_LAUNCH(fsk::k_sx_parts, dim3(1), dim3(512), 0, stream, (const uint32_t*)S.d_list_off.p, O, target, S.d_part_base.p,
                           (const u64*)S.d_sxstat.p, cap_words, desc, S.d_part_base.p + ((O + 2 + 3) & ~3u), max_parts, desc_parts);
                FSK_LAUNCH(fsk::k_sx_consume<false>, dim3(max_parts, e->sx_rounds), dim3(fsk::CS_THREADS), lds, stream, (const uint32_t*)S.d_ulist.p,
                           (const uint32_t*)S.d_list_off.p, (const uint32_t*)e->d_owner_r0.p, (const uint32_t*)S.d_part_base.p, O, target,
                           e->sx_cap, e->sx_pb, K, (const uint32_t*)nullptr, tpg, (u64)0, (const u64*)S.d_sxstat.p, cap_words, (uint32_t*)nullptr, pairs,
                           desc, Ep, packed ? 1 : 0, (const void*)colp, col16);
                e->st.launches += 1;
            }
            e->st.launches += 2;
        }
    } else {
        if (k_wait) FSK_HIP(hipStreamWaitEvent(stream, k_wait, 0));
        if (packed) {
            auto k_emit = skipping ? fsk::k_sx_emit<true, true, true> : fsk::k_sx_emit<true, false, true>;
            FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, reinterpret_cast<const uint32_t*>(S.d_E.p),
                       reinterpret_cast<const uint16_t*>(S.d_Pk.p), (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_owner_r0.p, e->sx_own_shift,
                       O, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod,
                       cmax, e->sx_pb, K, tpg, slot_stride, skipping ? reinterpret_cast<const uint16_t*>(S.d_Tk.p) : (const uint16_t*)nullptr,
                       (const u64*)nullptr, ~(u64)0, ntiles, 0, 0u, fsk::SX_SHORT, 0u, 0);
        } else {
            auto k_emit = skipping ? fsk::k_sx_emit<true, true, false> : fsk::k_sx_emit<true, false, false>;
            FSK_LAUNCH(k_emit, dim3(fsk::xcd_grid(ntiles)), dim3(fsk::EM_THREADS), 0, stream, (const uint2*)S.d_E.p, (const uint32_t*)S.d_Pk.p,
                       (const uint32_t*)S.d_ebase.p, (const uint32_t*)e->d_owner_r0.p, e->sx_own_shift, O, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t)row0, (uint32_t)row1, e->maxW, maxprod, cmax, e->sx_pb, K, tpg,
                       slot_stride, skipping ? (const uint32_t*)S.d_Tk.p : (const uint32_t*)nullptr, (const u64*)nullptr, ~(u64)0, ntiles, 0, 0u, fsk::SX_SHORT, 0u, 0);
        }
        e->st.launches += 1;
    }
    // the diagonal's combo-independent part (the streams and the atomics above carry the rest)
    if (k_wait && use_lists && !(words > 0 || slot_stride != 0)) FSK_HIP(hipStreamWaitEvent(stream, k_wait, 0));
    if (row1 > row0)
        FSK_LAUNCH(fsk::k_sx_diag_windows, dim3((uint32_t)((row1 - row0 + 255) / 256), slot_stride ? nb : 1), dim3(256), 0, stream,
                   (const uint32_t*)e->d_fstart.p, (uint32_t)row0, (uint32_t)row1, (uint32_t)nb, K, slot_stride,
                   use_lists ? (const u64*)S.d_sxstat.p : (const u64*)nullptr, cap_words, slot16 ? 1 : 0, slot16 ? e->sx_ovf_now : (uint32_t*)nullptr);
    e->st.launches += 1;
    if (k_done) FSK_HIP(hipEventRecord(k_done, stream));
    e->toc(&e->st.ms_pairs, stream);
    FSK_HIP(hipGetLastError());
    return FSK_OK;
}

int ensure_featseq(fsk_engine* e) {
    if (e->featseq_ready) return FSK_OK;
    // (on the device, from the window offsets that are there already: a host loop over the features and its upload
    // were 0.2 ms of idle stream at the head of every first call)
    FSK_HIP(e->d_featseq.reserve((size_t)std::max<int64_t>(1, e->nfeat)));
    if (e->N > 0)
        FSK_LAUNCH(fsk::k_sx_featseq, dim3((uint32_t)((e->N + 3) / 4)), dim3(256), 0, e->stream, (const uint32_t*)e->d_fstart.p, (uint32_t)e->N,
                   e->d_featseq.p);
    // the g-mer windows, packed once: every record of every slot is then one 8- or 16-byte load (k_sx_extract_win)
    const int wbits = e->cfg.g * e->bits;
    e->win_words = wbits <= 64 ? 2 : wbits <= 128 ? 4 : 0;
    if (e->win_words && e->nfeat > 0) {
        FSK_HIP(e->d_win.reserve((size_t)e->nfeat * e->win_words));
        const dim3 grid((uint32_t)((e->nfeat + 255) / 256));
        if (e->win_words == 2)
            FSK_LAUNCH(fsk::k_sx_windows<2>, grid, dim3(256), 0, e->stream, e->view(), (const uint32_t*)e->d_featseq.p,
                       (const uint32_t*)e->d_fstart.p, (uint32_t)e->nfeat, e->cfg.g, e->d_win.p);
        else
            FSK_LAUNCH(fsk::k_sx_windows<4>, grid, dim3(256), 0, e->stream, e->view(), (const uint32_t*)e->d_featseq.p,
                       (const uint32_t*)e->d_fstart.p, (uint32_t)e->nfeat, e->cfg.g, e->d_win.p);
    }
    FSK_HIP(hipStreamSynchronize(e->stream));  // (every lane's kernels read both arrays)
    e->featseq_ready = true;
    return FSK_OK;
}

constexpr int SX_DEFER = 8;          // variance mode: batches whose counts are read at their hand-over
constexpr int SX_DEFER_COMBOS = 16;  //                combos of such a batch at most

// Pinned staging of the batches. The HEAD — positions and {pairs, words} of variance mode's deferred batches, which
// stay in flight across calls — is allocated once at its fixed size and never moved; only the per-call TAIL (the
// batches of one exact accumulate, all read back before the call returns) grows.
constexpr size_t SX_HEAD_POS = (size_t)SX_DEFER * SX_DEFER_COMBOS * 255;  // (k <= g <= 255)
constexpr size_t SX_HEAD_STAT = (size_t)2 * SX_DEFER;
int sx_pinned(fsk_engine* e, size_t tail_pos_bytes, size_t tail_stat_words) {
    if (!e->h_sx_head_pos) {
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_head_pos, SX_HEAD_POS));
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_head_stat, SX_HEAD_STAT * sizeof(u64)));
        memset(e->h_sx_head_stat, 0, SX_HEAD_STAT * sizeof(u64));
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_head_flag, SX_DEFER * sizeof(uint32_t)));
        memset(e->h_sx_head_flag, 0, SX_DEFER * sizeof(uint32_t));
    }
    if (tail_pos_bytes > e->h_sx_pos_cap) {
        if (e->h_sx_pos) (void)hipHostFree(e->h_sx_pos);
        e->h_sx_pos = nullptr; e->h_sx_pos_cap = 0;
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_pos, tail_pos_bytes + tail_pos_bytes / 2));
        e->h_sx_pos_cap = tail_pos_bytes + tail_pos_bytes / 2;
    }
    if (tail_stat_words > e->h_sx_stat_cap) {
        if (e->h_sx_stat) (void)hipHostFree(e->h_sx_stat);
        e->h_sx_stat = nullptr; e->h_sx_stat_cap = 0;
        FSK_HIP(hipHostMalloc((void**)&e->h_sx_stat, (tail_stat_words + tail_stat_words / 2) * sizeof(u64)));
        e->h_sx_stat_cap = tail_stat_words + tail_stat_words / 2;
    }
    return FSK_OK;
}

// how many words a batch may hold when it is enqueued before its count is known (0: size it exactly)
u64 sx_guard_for(fsk_engine* e, int lane, u64 nrec, bool by_slot = false) {
    DevBuf<uint32_t>& ulist = e->sxs[lane].d_ulist;
    if (e->sx_exactly() || e->profile_sync()) return 0;
    if (!by_slot && e->sx_form == 2) return 0;                          // the two-level form sizes every pass exactly
    if (by_slot ? (!e->sx_lists || e->sx_form == 1) : e->sx_form != 0) return ~(u64)0;  // no streams: nothing to size
    if (e->sx_wpr == 0) return 0;                                       // (the first batch of these sequences)
    if (e->tune.guard_cap)  // (testing: pretend the stream buffer holds this many words)
        return ulist.reserve((size_t)e->tune.guard_cap) == hipSuccess ? (u64)e->tune.guard_cap : 0;
    const u64 expect = e->sx_words_of(nrec), want = expect + expect / 2;  // (words per record differ by a few percent between batches)
    if (want >= e->sx_max_words()) return 0;
    if ((u64)ulist.cap < want && ulist.reserve((size_t)want) != hipSuccess) return 0;
    return std::min<u64>((u64)ulist.cap, e->sx_max_words() - 1);
}

// the counts of deferred batch `slot` (its kernels have finished): false when it has to be redone
bool sx_harvest(fsk_engine* e, int slot) {
    if (slot < 0) return true;
    if (e->h_sx_head_flag && e->h_sx_head_flag[slot]) {  // a sum did not fit its u16 slot triangle (k_sx_consume): u32 from here on
        e->h_sx_head_flag[slot] = 0u;
        e->slots16_ok = false;
        e->sx_redone += 1;
        if (e->sx_defer[slot].active) {
            e->sx_defer[slot].active = false;
            e->sx_saw(e->h_sx_head_stat[2 * slot + 1], e->sx_defer[slot].nrec);
        }
        return false;
    }
    if (!e->sx_defer[slot].active) return true;
    e->sx_defer[slot].active = false;
    const u64 pairs = e->h_sx_head_stat[2 * slot], words = e->h_sx_head_stat[2 * slot + 1];
    e->sx_saw(words, e->sx_defer[slot].nrec);
    e->sx_saw_pairs(pairs, e->sx_defer[slot].nrec);
    if (words > e->sx_defer[slot].cap) { e->sx_redone += 1; return false; }
    e->u_extra += pairs;
    return true;
}

// slot_stride != 0 (variance mode): combo q of the list goes to its own u32 triangle (uint32_t*)K + q * slot_stride,
// written whole; returns FSK_RETRY_UNGROUPED when that form cannot be used for this batch.
// defer >= 0 (variance mode): the call returns with the batch enqueued; the caller passes `defer` to
// sx_harvest() once the batch has finished and redoes the batch (with e->sx_redoing set) if that says so.
int accumulate_sparse(fsk_engine* e, const int32_t* combos, int n, u64* K, int64_t row0, int64_t row1, u64 slot_stride,
                      int defer) {
    { int rcz = materialise_zero(e); if (rcz) return rcz; }
    int rc = ensure_featseq(e);
    if (rc) return rc;
    sx_choose_form(e);
    const bool blocks_form = e->sx_form == 2 && slot_stride == 0;
    // Batch so that (i) the record count stays below the cap, (ii) the owner bands can sum a batch in u32 LDS cells
    // (per cell and combo <= maxW^2), (iii) a batch's update words stay well inside what one stream addresses
    // (2^31: beyond it the pairs go to K with atomics, an order of magnitude slower) — judged by the most words per
    // record seen so far on these sequences; while none has been seen, a first batch of at most 2^25 records.
    const size_t nfeat = (size_t)std::max<int64_t>(1, e->nfeat);
    const u64 by_cells = std::max<u64>(1, 0xffffffffull / std::max<u64>(1, (u64)e->maxW * e->maxW));
    // (owner bands with descriptors: batches of 2^25 records — the entries of a batch, which every descriptor's partners are read
    // from again and again, then stay in the 256 MB Infinity Cache: large-g regime 0.82 -> 0.795 s; 2^26: 0.805; 2^24: 0.826)
    // (a call of at least sixteen such batches: g = 12, 924 combos, three batches of 2^27: 0.025 s, nine of 2^25: 0.031)
    const size_t max_records = e->sx_desc_now() && !blocks_form && slot_stride == 0 && (u64)n * nfeat >= ((u64)1 << 29) ? (size_t)1 << 25 : SPARSE_MAX_RECORDS;
    auto batch_combos = [&](int left) {
        size_t recs = e->tune.sparse_batch_records ? (size_t)e->tune.sparse_batch_records : max_records;
        // (the two-level form takes a batch of any word count in passes, and sweeps K once a batch: as many records as the cap allows)
        if (e->sx_wpr == 0) recs = std::min<size_t>(recs, (size_t)1 << 25);
        else if (!blocks_form) recs = std::min<size_t>(recs, (size_t)std::max(1.0, (double)(e->sx_max_words() / 2) / e->sx_wpr));
        const u64 B = std::max<u64>(1, std::min<u64>({(u64)(recs / nfeat), (u64)left, by_cells, (u64)65535}));  // (65535: grid.y)
        return (int)B;
    };
    const int recbits = e->sx_keybits + e->sx_sb;  // (<= 96 + 31: a 128-bit record always holds it)
    if (defer >= 0 && (defer >= SX_DEFER || batch_combos(n) != n || n > SX_DEFER_COMBOS)) defer = -1;
    // variance mode's batches in flight alternate between two lanes of scratch and two streams
    const int lane = sx_lane_of(e, defer);
    e->sx_last_lane = lane;  // (what the caller orders its own passes over the batch's triangles against)
    if (lane && !e->lane_stream) FSK_HIP(hipStreamCreateWithFlags(&e->lane_stream, hipStreamNonBlocking));
    rc = sx_pinned(e, (size_t)n * e->k, (size_t)2 * n);  // (at most one batch per combo)
    if (rc) return rc;
    auto one = [&](int s, int nb, unsigned char* pos_pin, u64* stat_pin, u64 guard, int ln, size_t pos_off = 0, hipEvent_t k_wait = nullptr,
                   hipEvent_t k_done = nullptr) {
        // (slot triangles are u32 arrays, slot_stride cells apart)
        u64* Kb = slot_stride ? reinterpret_cast<u64*>(reinterpret_cast<uint32_t*>(K) + (u64)s * slot_stride) : K;
        return recbits <= 32   ? sparse_batch<uint32_t>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard, ln, pos_off, k_wait, k_done)
               : recbits <= 64 ? sparse_batch<u64>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard, ln, pos_off, k_wait, k_done)
                               : sparse_batch<u128>(e, combos + s, nb, Kb, row0, row1, slot_stride, pos_pin, stat_pin, guard, ln, pos_off, k_wait, k_done);
    };
    e->sx_slot16_used = false;
    if (defer >= 0) {
        e->sx_slot16_used = e->sx_slot16 && slot_stride != 0 && e->sx_lists && !e->tune.sparse_global;
        e->sx_ovf_now = e->h_sx_head_flag + defer;
        e->h_sx_head_flag[defer] = 0u;
        const u64 guard = sx_guard_for(e, lane, (u64)n * nfeat, slot_stride != 0);
        e->sx_defer[defer].active = guard != 0;
        e->sx_defer[defer].cap = guard;
        e->sx_defer[defer].nrec = (u64)n * nfeat;
        rc = one(0, n, e->h_sx_head_pos + (size_t)defer * SX_DEFER_COMBOS * e->k, e->h_sx_head_stat + 2 * defer, guard, lane);
        if (rc) e->sx_defer[defer].active = false;
        return rc;
    }
    // An exact accumulate of MANY batches runs them in TWO LANES (a scratch set and a stream each, as variance mode's): the
    // batches enqueued under a guard alternate between the lanes, so that one batch's sort and segment kernels (bound by the
    // instructions they issue) share the GPU with the other's emit / consume (bound by latency and bandwidth). What orders
    // the lanes is K: a batch's consume pass waits for the previous batch's (ev_lane), everything before it is per lane.
    // Measured (profiles/r04_exact_lanes_ab.txt): the large-g regime (EP300, g = 20: ~40 batches of 2^24.7 records, sized by
    // their update words) 1.735 -> 1.66-1.68 s; config 4 (three batches of 2^27 records) 10.1 -> 10.4-10.8 ms — big batches
    // fill the GPU by themselves and two of them at once only share its caches — hence: from six batches on.
    const int nb0 = batch_combos(n);
    // (how many batches the call takes once its first batch has been sized: the 2^25-record cap of a first batch — every
    // fsk_compute starts with one — says nothing about the batches that follow)
    int nb_steady = nb0;
    if (e->sx_wpr == 0) {
        const size_t recs = e->tune.sparse_batch_records ? (size_t)e->tune.sparse_batch_records : max_records;
        nb_steady = (int)std::max<u64>(1, std::min<u64>({(u64)(recs / nfeat), (u64)n, by_cells, (u64)65535}));
    }
    const bool many = (n + nb_steady - 1) / nb_steady >= 6;
    const bool two = (e->tune.sparse_exact_lanes >= 2 || (e->tune.sparse_exact_lanes == 0 && many)) && !e->profile_sync() && !e->sx_exactly() && slot_stride == 0 && nb0 < n && !blocks_form;
    if (two) {
        if (!e->lane_stream) FSK_HIP(hipStreamCreateWithFlags(&e->lane_stream, hipStreamNonBlocking));
        for (auto& ev : e->ev_lane)
            if (!ev) FSK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        FSK_HIP(e->d_pos.reserve((size_t)n * e->k));  // (never grown while a batch in flight reads its piece)
    }
    struct Enq { int s, nb; u64 cap; };  // batches enqueued under a guard (not sized exactly): checked below
    std::vector<Enq> enq;
    int q = 0, next_lane = 0, prev_lane = -1;
    bool lane1_started = false;
    for (int s = 0; s < n; ++q) {
        int nb = batch_combos(n - s);  // (grows after the first batch of a set of sequences has been sized)
        if (two && e->sx_wpr != 0) {
            // (two lanes want batches to alternate: what is left goes in an even number of equal batches, four at least
            // while a batch still has a few dozen combos)
            const int left = n - s;
            int parts = (left + nb - 1) / nb;
            nb = std::min(nb, (left + parts - 1) / parts);
        }
        int ln = two ? next_lane : 0;
        u64 cap = sx_guard_for(e, ln, (u64)nb * nfeat);
        if (cap == 0 && ln != 0) { ln = 0; cap = sx_guard_for(e, 0, (u64)nb * nfeat); }  // (a batch that waits for its size: lane 0)
        hipEvent_t k_wait = nullptr, k_done = nullptr;
        if (two) {
            if (ln == 1 && !lane1_started) {  // lane 1 starts after what the engine's stream holds so far
                FSK_HIP(hipEventRecord(e->ev_lane[0], e->stream));
                FSK_HIP(hipStreamWaitEvent(e->lane_stream, e->ev_lane[0], 0));
                lane1_started = true;
            }
            if (prev_lane >= 0 && prev_lane != ln) k_wait = e->ev_lane[2 + prev_lane];
            k_done = e->ev_lane[2 + ln];
        }
        rc = one(s, nb, e->h_sx_pos + (size_t)s * e->k, e->h_sx_stat + 2 * q, cap, ln, two ? (size_t)s * e->k : 0, k_wait, k_done);
        if (rc) {
            if (lane1_started) (void)hipStreamSynchronize(e->lane_stream);
            return rc;
        }
        enq.push_back(Enq{s, nb, cap});
        s += nb;
        prev_lane = ln;
        next_lane = ln ^ 1;
    }
    if (lane1_started) {  // the engine's stream continues after lane 1's last batch
        FSK_HIP(hipEventRecord(e->ev_lane[1], e->lane_stream));
        FSK_HIP(hipStreamWaitEvent(e->stream, e->ev_lane[1], 0));
    }
    bool waiting = false;
    for (const Enq& b : enq) waiting |= b.cap != 0;
    if (!waiting) return FSK_OK;
    FSK_HIP(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < enq.size(); ++i) {
        const Enq& b = enq[i];
        if (!b.cap) continue;
        const u64 pairs = e->h_sx_stat[2 * i], words = e->h_sx_stat[2 * i + 1];
        e->sx_saw(words, (u64)b.nb * nfeat);
        e->sx_saw_pairs(pairs, (u64)b.nb * nfeat);
        if (words <= b.cap) { e->u_extra += pairs; continue; }
        // the batch did not fit and has left K alone: once more, sized exactly
        e->sx_redone += 1;
        const bool was = e->sx_redoing;
        e->sx_redoing = true;
        rc = one(b.s, b.nb, e->h_sx_pos + (size_t)b.s * e->k, e->h_sx_stat + 2 * i, 0, 0);
        e->sx_redoing = was;
        if (rc) return rc;
    }
    return FSK_OK;
}

}  // namespace fsk_detail

#if defined(FSK_EM_CLOCKS) && FSK_EM_CLOCKS
// measurement builds only (tools/emit_phases.py): the cycle sums of k_sx_emit's phases since the last call, then zeroed
extern "C" int fsk_debug_emit_clocks(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fsk::g_em_clk), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(fsk::g_em_clk), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif


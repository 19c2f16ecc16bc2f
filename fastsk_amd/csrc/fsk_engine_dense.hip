// fsk_engine_dense.hip — host side of the DENSE dataflow (configs 2, 3, 5): LDS plan of the count kernel,
// XCD-aware tile table, combos per launch, storing / adding flush. The reference work it stands for:
// cntsrtna + countAndUpdateTri per combo (shared.cpp:156-191, 268-333) and the K += Ks reduce
// (fastsk_kernel.cpp:286-315).
#include "fsk_engine_internal.h"
#include "fsk_kernels_dense.h"

using namespace fsk_detail;

namespace fsk_detail {

// k_dense_count LDS plan: (CH + g - 1) staged symbols x 64 sequences + the u16 histogram of one
// key sweep (512 B per key quad). Symbols get what they need up to 64 KiB (all windows in one
// staging pass when possible), the histogram gets the rest (fewer sweeps over large key spaces).
DensePlan dense_plan(uint32_t maxW, int g, uint32_t Vq, size_t extra) {
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

namespace {

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

}  // namespace

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
    if (e->tab_t0 != t0 || e->tab_t1 != t1 || e->tab_n == 0 || e->tab_ftt != first_test_tile) {
        std::vector<uint32_t> tab;
        build_tile_table(t0, t1, first_test_tile, tab);
        if (first_test_tile == 0xffffffffu && tab.size() != (u64)t1 * (t1 + 1) / 2 - (u64)t0 * (t0 + 1) / 2)
            return e->fail(FSK_EDEVICE, "internal: tile table size mismatch");
        FSK_HIP(e->d_tiletab.reserve(tab.size()));
        FSK_HIP(hipMemcpyAsync(e->d_tiletab.p, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        FSK_HIP(hipStreamSynchronize(e->stream));
        e->tab_t0 = t0; e->tab_t1 = t1; e->tab_n = (uint32_t)tab.size(); e->tab_ftt = first_test_tile;
    }
    const u64 n_tiles = e->tab_n;
    const bool compact = e->compact;
    const uint32_t Vkeys = (uint32_t)e->V, Vw = (Vkeys + 31u) / 32u;
    DensePlan plan = dense_plan(e->maxW, e->cfg.g, e->Vq, compact ? (size_t)Vkeys * 2 : 0);  // may be re-planned below
    if (plan.CH == 0) return e->fail(FSK_EUNSUPPORTED, "dense path: LDS plan does not fit");
    uint32_t CH = plan.CH;
    if (e->tune.dense_chunk) CH = std::max(1u, std::min(CH, (uint32_t)e->tune.dense_chunk));
    size_t lds = (size_t)(CH + e->cfg.g - 1) * fsk::PANEL + (size_t)plan.Vcq * 512 + (compact ? (size_t)Vkeys * 2 : 0);
    // several histogram sweeps over one staging pass: cache the window keys in LDS (u16 each) when
    // they fit next to everything else, so that only the first sweep computes them
    uint32_t kc_rows = 0;
    if (plan.Vcq < e->Vq && CH >= e->maxW && !e->tune.dense_chunk) {
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
    {
        auto k0 = fsk::k_dense_count<false, false>;
        auto k1 = fsk::k_dense_count<false, true>;
        auto k2 = fsk::k_dense_count<true, false>;
        FSK_HIP(fsk_hw::allow_dynamic_lds(k0, lds));
        FSK_HIP(fsk_hw::allow_dynamic_lds(k1, lds));
        FSK_HIP(fsk_hw::allow_dynamic_lds(k2, lds));
    }
    if (compact) {
        FSK_HIP(e->d_keybits.reserve((size_t)chunk * Vw));
        FSK_HIP(e->d_lut.reserve((size_t)chunk * Vkeys));
        FSK_HIP(e->d_vc.reserve((size_t)chunk));
    }
    std::vector<uint16_t> h_vc;
    std::vector<uint8_t> pos;
    const uint8_t* chunk_pos = e->d_pos.p;  // the kept positions of the chunk at hand, on the device
    for (int s = 0; s < n; s += chunk) {
        const int nb = std::min(chunk, n - s);
        // the count panels of an unchanged single-chunk combo list are reused by the FOLLOWING row
        // bands of one pass (row0 > 0); a call that starts at row 0 always recounts
        const bool cached = row0 > 0 && e->prep_valid && nb == n && (int)e->prep_combos.size() == n &&
                            std::equal(combos, combos + n, e->prep_combos.begin());
        if (!cached) {
            e->prep_valid = false;
            // the chunk's kept positions: a run of consecutive combo ids (every exact call: 0, 1, 2, ...) reads them from the
            // resident table of all combos, uploaded once per engine; any other list is gathered and uploaded
            bool consecutive = true;
            for (int q = 1; q < nb && consecutive; ++q) consecutive = combos[s + q] == combos[s] + q;
            if (consecutive) {
                if (!e->allpos_ready) {
                    FSK_HIP(e->d_allpos.reserve(e->all_pos.size()));
                    FSK_HIP(hipMemcpy(e->d_allpos.p, e->all_pos.data(), e->all_pos.size(), hipMemcpyHostToDevice));
                    e->allpos_ready = true;
                }
                chunk_pos = e->d_allpos.p + (size_t)combos[s] * e->k;
                FSK_HIP(hipMemsetAsync(e->d_flag.p, 0, sizeof(uint32_t), e->stream));
            } else {
                pos.resize((size_t)nb * e->k);
                for (int q = 0; q < nb; ++q)
                    memcpy(&pos[(size_t)q * e->k], &e->all_pos[(size_t)combos[s + q] * e->k], e->k);
                FSK_HIP(hipMemcpyAsync(e->d_pos.p, pos.data(), pos.size(), hipMemcpyHostToDevice, e->stream));
                FSK_HIP(hipMemsetAsync(e->d_flag.p, 0, sizeof(uint32_t), e->stream));
                FSK_HIP(hipStreamSynchronize(e->stream));  // `pos` is a pageable temporary
                chunk_pos = e->d_pos.p;
            }
            // ---- segment counts
            // up to 16 combos share one staging of a panel's symbols, fewer when that would leave the
            // launch with less than ~512 workgroups (few sequences; with 1024 as the floor config 3 ran 11 combos a
            // staging in 1130 workgroups: 16 a staging in 791 is 0.1-0.15 ms faster per count pass)
            int slots_per_chunk = std::max(1, std::min({nb, 16, (int)((u64)nb * panels_pad / 512)}));
            const int n_chunks = (nb + slots_per_chunk - 1) / slots_per_chunk;
            e->tic();
            const dim3 cgrid(panels_pad, n_chunks);
            // (function pointers: a template-id with a comma cannot pass through the launch macro)
            auto k_mark = fsk::k_dense_count<true, false>;
            auto k_count_lut = fsk::k_dense_count<false, true>;
            auto k_count = fsk::k_dense_count<false, false>;
            if (compact) {  // which keys occur per combo -> rank tables -> compacted panels
                if (e->compact_rare) {  // from the places of the rare symbols (listed once per set of sequences)
                    const uint32_t cap = std::max(1u, e->rare_places);
                    if (!e->rare_ready) {
                        FSK_HIP(e->d_rare.reserve(cap));
                        FSK_HIP(e->d_rare_n.reserve(1));
                        FSK_HIP(hipMemsetAsync(e->d_rare_n.p, 0, sizeof(uint32_t), e->stream));
                        FSK_LAUNCH(fsk::k_dense_rare_scan, dim3((uint32_t)((e->N + 3) / 4)), dim3(256), 0, e->stream, e->view(), e->rare_mask,
                                   e->d_rare.p, cap, e->d_rare_n.p);
                        FSK_HIP(e->d_common.reserve(Vw));
                        FSK_LAUNCH(fsk::k_dense_common_keys, dim3((Vw + 255u) / 256u), dim3(256), 0, e->stream, e->d_common.p, Vkeys, e->k, e->sigma,
                                   e->rare_mask);
                        e->rare_ready = true;
                    }
                    FSK_LAUNCH(fsk::k_dense_keybits_init, dim3((Vw + 255u) / 256u, (uint32_t)nb), dim3(256), 0, e->stream, e->d_keybits.p,
                               (const uint32_t*)e->d_common.p, Vkeys);
                    const uint32_t mblocks = (uint32_t)std::max<u64>(1, std::min<u64>(((u64)cap * (u64)e->cfg.g + 255) / 256, 64));
                    FSK_LAUNCH(fsk::k_dense_mark_rare, dim3(mblocks, (uint32_t)nb), dim3(256), 0, e->stream, e->view(), (const u64*)e->d_rare.p,
                               (const uint32_t*)e->d_rare_n.p, cap, e->cfg.g, e->k, e->sigma, (const uint8_t*)chunk_pos, Vkeys, e->d_keybits.p);
                    e->st.launches += 1;
                } else {
                    FSK_HIP(hipMemsetAsync(e->d_keybits.p, 0, (size_t)nb * Vw * sizeof(uint32_t), e->stream));
                    FSK_LAUNCH(k_mark, cgrid, dim3(256), lds, e->stream, e->view(), e->cfg.g,
                               e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, chunk_pos, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
                               e->d_rowmask.p, nst, e->d_flag.p, Vkeys, (const uint16_t*)nullptr, (const uint16_t*)nullptr, e->d_keybits.p, kc_rows);
                }
                FSK_LAUNCH(fsk::k_dense_keylut, dim3(nb), dim3(256), 0, e->stream, e->d_keybits.p, Vkeys, e->d_lut.p, e->d_vc.p);
                FSK_LAUNCH(k_count_lut, cgrid, dim3(256), lds, e->stream, e->view(), e->cfg.g,
                           e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, chunk_pos, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
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
                           e->k, e->sigma, e->Vq, plan.Vcq, e->maxW, CH, chunk_pos, nb, slots_per_chunk, e->d_C4.p, e->d_C4H.p,
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
            if (e->profile_sync() && !e->prep_overflow) {  // exact algorithmic update count U
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
            // (variance mode's slot triangle: the storing launch that would have written every cell is
            // not coming, and the general dataflow ADDS — the slot still holds an earlier iteration)
            if (e->store_next) FSK_HIP(hipMemsetAsync(K, 0, (size_t)e->pairs * sizeof(u64), e->stream));
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
            // (key-compacted panels — the direct-to-LDS compact kernel, 768 resident slots: about 350 rows; config 3
            // 6 -> 10 splits, 5.6 -> 5.5 ms)
            int per = std::max(2, (int)std::ceil((compact ? 350.0 : 600.0) / rows_per_combo));
            n_splits = std::max(1, (nb + per - 1) / per);
            // ... and about 16k workgroups are enough: beyond that more splits only add flushes
            n_splits = std::min(n_splits, (int)((16384 + n_tiles - 1) / n_tiles));
            if ((double)n_tiles * n_splits < slots)
                n_splits = std::max(n_splits, std::min(nb / 2, (int)((slots + n_tiles - 1) / n_tiles)));
            n_splits = std::max(1, std::min({n_splits, nb, 4096}));
        }
        if (e->tune.tile_splits > 0) n_splits = std::min({nb, (int)e->tune.tile_splits, 4096});
        const int slots_per_split = (nb + n_splits - 1) / n_splits;
        n_splits = (nb + slots_per_split - 1) / slots_per_split;
        // Store instead of add? Only the first launch over rows that are still "zero by contract",
        // starting at their lower edge, with one workgroup per tile and the engine's own triangle.
        int store = 0;
        if (e->store_next) {
            if (!(!compact && n_splits == 1 && first_test_tile == 0xffffffffu && row0 == 0 && row1 >= e->N))
                return e->fail(FSK_ESTATE, "internal: a storing tile launch was asked for where none is possible");
            store = 1;
        } else if (e->lazy_lo >= 0) {
            if (K == e->d_K && !compact && n_splits == 1 && first_test_tile == 0xffffffffu && row0 == e->lazy_lo &&
                row1 <= e->lazy_hi) {
                store = 1;
                e->lazy_lo = row1 < e->lazy_hi ? row1 : -1;
                if (e->lazy_lo < 0) e->lazy_hi = -1;
            } else {
                int rcz = materialise_zero(e);
                if (rcz) return rcz;
            }
        }
        if (e->profile_sync())  // the flagged rows' remainder products of this launch, for fsk_stats.dense_macs (read by fsk_get_stats)
            FSK_LAUNCH(fsk::k_dense_remainder_rows, dim3((uint32_t)((n_tiles + 255) / 256), (uint32_t)nb), dim3(256), 0, e->stream,
                       (const uint32_t*)e->d_rowmask.p, (const uint32_t*)e->d_tiletab.p, (uint32_t)n_tiles, (uint32_t)nb, nst, compact ? 1 : 0,
                       e->d_U.p + 1);
        e->tic();
        // (small N, the combo range split over several workgroups a tile: their sums through 32-bit staging blocks, not atomics)
        const bool small = n_splits >= 2 && e->tune.dense_small != 0 && dense_small_stage_bytes(n_tiles, n_splits) <= ((size_t)2 << 30);
        if (small) {
            const int rcs = dense_tile_small(e, compact, n_tiles, n_splits, nb, Vq8, nst, K, slots_per_split);
            if (rcs) return rcs;
        } else if (compact)
            FSK_LAUNCH(fsk::k_dense_tile_dma_compact, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p, e->d_C4H.p,
                       e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split, 0, (const uint16_t*)e->d_vc.p);
        else
            FSK_LAUNCH(fsk::k_dense_tile_dma, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, e->d_C4.p, e->d_C4H.p,
                       e->d_rowmask.p, e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N, K, slots_per_split, store);
        e->toc(&e->st.ms_tile);
        e->st.n_tile_launches += 1;
        u64 row_sum = (u64)Vq8 * (u64)nb;  // dword rows multiplied per tile (the flagged rows' remainder products are added by fsk_get_stats)
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

}  // namespace fsk_detail

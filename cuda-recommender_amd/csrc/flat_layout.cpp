#include "flat_layout.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

namespace mfx {

uint32_t pick_tiles_per_span(uint64_t nnz, bool panels) {
    // Aim for ~48k spans (256 CUs x 32 resident waves x ~6 rounds) but keep a span between
    // 2 and 16 tiles: shorter spans waste the per-span prologue, longer ones leave the tail of
    // the grid unbalanced.  With LDS panels a workgroup also pays one slice load per chunk, so
    // spans are at least 8 tiles there -- and at most 8: measured at Z = 9.9e8 (4.8 M x 17 770), CSC / CSR pass per
    // launch: 6 tiles 1847 / 1796 us, 8 tiles 1725 / 1731, 10 (CSR) and 16 (CSC) tiles 1823 / 2054-2116; the outer
    // iteration 263 -> 237 ms.  (A chunk of 16 longer spans touches more virtual segments than the 1024-entry LDS
    // window of per-segment operands holds; the Netflix shape gets 8 from its size anyway.)
    if (panels) {
        // Mid-sized matrices: as few tiles as still put all workgroups (16 spans each) into ONE round of the 512
        // resident ones -- ML-10M shape (1e7 ratings, k = 64): 8 tiles = 305 workgroups 5.41 ms per outer iteration,
        // 6 tiles = 407 workgroups 4.38 ms, 4 tiles = 610 workgroups (a second, thin round) 4.68 ms.
        const uint64_t per_tile_round = 16ull * kTileElems * 512;
        uint64_t t = (nnz + per_tile_round - 1) / per_tile_round;
        t = (t + 1) & ~uint64_t(1);
        return (uint32_t) std::min<uint64_t>(8, std::max<uint64_t>(4, t));
    }
    const uint64_t target_spans = 49152;
    uint64_t t = (nnz / kTileElems + target_spans - 1) / target_spans;
    t = (t + 1) & ~uint64_t(1);
    return (uint32_t) std::min<uint64_t>(16, std::max<uint64_t>(2, t));
}

namespace {

// Runs fn(begin, end) over [0, n) on a few host threads (plain std::thread: libmfx must not drag
// a second OpenMP runtime into a process that already hosts torch's).
// Like a range split over segments, but the cuts balance the non-zeros (ptr is the prefix sum), so a
// few heavy segments do not serialise the pass.
template <typename F>
void parallel_segments(const uint32_t* ptr, uint32_t nseg, F fn) {
    const unsigned nt = ThreadGang::width();
    const uint64_t nnz = ptr[nseg];
    if (nseg < 64 || nnz < (1u << 16) || nt == 1) { fn(0u, nseg); return; }
    std::vector<uint32_t> cut(nt + 1, nseg);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; ++t)
        cut[t] = (uint32_t) (std::lower_bound(ptr, ptr + nseg + 1, (uint32_t) (nnz * t / nt)) - ptr);
    for (unsigned t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    cut[nt] = nseg;
    ThreadGang gang;
    for (unsigned t = 0; t < nt; ++t)
        if (cut[t + 1] > cut[t]) gang.run([=] { fn(cut[t], cut[t + 1]); });
    gang.wait();
}

}  // namespace

void parallel_ranges_u64(uint64_t n, void (*fn)(uint64_t, uint64_t, void*), void* ctx) {
    const unsigned nt = ThreadGang::width();
    if (n < (1u << 16) || nt == 1) { fn(0, n, ctx); return; }
    ThreadGang gang;
    for (unsigned t = 0; t < nt; ++t) {
        const uint64_t b = n * t / nt, e = n * (t + 1) / nt;
        gang.run([=] { fn(b, e, ctx); });
    }
    gang.wait();
}

void build_flat_layout(const uint32_t* ptr, const uint32_t* idx, uint32_t nseg, uint64_t nnz, uint32_t G,
                       const FlatLayoutOptions& opt, FlatLayoutHost* out) {
    FlatLayoutHost& L = *out;
    L = FlatLayoutHost();
    L.nseg = nseg;
    L.gather_len = G;
    L.nnz = nnz;
    L.panel_rows = opt.panel_rows;
    L.lds = opt.panel_rows ? opt.lds : true;
    L.npanels = opt.panel_rows ? std::max(1u, (G + opt.panel_rows - 1) / opt.panel_rows) : 1u;
    L.spans_per_wg = (opt.panel_rows && opt.lds) ? std::max(1u, opt.spans_per_wg) : 1u;
    uint32_t tps = opt.tiles_per_span ? opt.tiles_per_span : pick_tiles_per_span(nnz, opt.panel_rows != 0 && opt.lds);
    if (tps & 1) ++tps;  // the kernel consumes tiles in pairs
    L.tiles_per_span = tps;
    const uint64_t span = (uint64_t) tps * kTileElems;
    const uint64_t chunk = span * L.spans_per_wg;
    const uint32_t P = L.npanels, PR = L.panel_rows;
    const size_t nv = (size_t) P * nseg;

    // 1. entries per (panel, segment); with compact_perm also where each pair's run starts in the input
    //    and whether every pair is ONE run there (each panel visited in one stretch per segment)
    std::vector<uint32_t> cnt(nv, 0);
    L.seg_cnt.resize(nseg);
    if (opt.compact_perm) L.first_q.resize(nv);
    uint32_t* const fq = opt.compact_perm ? L.first_q.data() : nullptr;
    std::atomic<bool> grouped{true};
    parallel_segments(ptr, nseg, [&](uint32_t b, uint32_t e) {
        bool ok = true;
        for (uint32_t c = b; c < e; ++c) {
            L.seg_cnt[c] = ptr[c + 1] - ptr[c];
            if (P == 1) { cnt[c] = ptr[c + 1] - ptr[c]; if (fq) fq[c] = ptr[c]; continue; }
            // panel of an index: the division is only taken when the index leaves the current panel,
            // i.e. a handful of times per segment when the indices are ascending (the usual input)
            uint32_t pp = 0, lo = 0, hi = 0;
            for (uint32_t q = ptr[c]; q < ptr[c + 1]; ++q) {
                const uint32_t i = idx[q];
                if (i < lo || i >= hi) {
                    pp = i / PR; lo = pp * PR; hi = lo + PR;
                    const size_t v = (size_t) pp * nseg + c;
                    if (cnt[v] != 0) ok = false;  // second visit of this panel
                    else if (fq) fq[v] = q;
                }
                ++cnt[(size_t) pp * nseg + c];
            }
        }
        if (!ok) grouped = false;
    });
    // 2. panel-major exclusive scan; every panel padded to whole workgroup chunks, the padding
    //    folded into the panel's last virtual segment
    L.ptr_v.resize(nv + 1);
    std::vector<uint64_t> panel_real_end(P);
    uint64_t pos = 0;
    for (uint32_t p = 0; p < P; ++p) {
        for (uint32_t c = 0; c < nseg; ++c) {
            L.ptr_v[(size_t) p * nseg + c] = (uint32_t) pos;
            pos += cnt[(size_t) p * nseg + c];
        }
        panel_real_end[p] = pos;
        pos = (pos + chunk - 1) / chunk * chunk;
    }
    if (pos == 0) pos = chunk;  // nnz == 0: keep one (all padding) chunk so that grids are never empty
    L.ptr_v[nv] = (uint32_t) pos;
    L.padded_nnz = pos;
    L.nspans = (uint32_t) (pos / span);
    L.panel_real_end.assign(panel_real_end.begin(), panel_real_end.end());
    L.perm_is_runs = opt.compact_perm && grouped.load();
    if (!L.perm_is_runs) { L.first_q.clear(); L.first_q.shrink_to_fit(); }

    // 3. stored order: panel-local indices + where every stored element came from (+ the values and
    //    the 16-bit form of the indices if asked for).  The arrays are left uninitialised: every real
    //    position is written by the placement below, the padding ranges right after.
    const bool want16 = opt.emit_idx16 && PR != 0 && L.lds;
    if (want16) L.idx16.resize(L.padded_nnz); else L.idx_local.resize(L.padded_nnz);
    if (!L.perm_is_runs) L.perm.resize(L.padded_nnz);
    if (opt.emit_val) L.val_st.resize(L.padded_nnz);
    uint32_t* const d_idx = want16 ? nullptr : L.idx_local.data();
    uint16_t* const d_idx16 = want16 ? L.idx16.data() : nullptr;
    uint32_t* const d_perm = L.perm_is_runs ? nullptr : L.perm.data();
    float* const d_val = opt.emit_val ? L.val_st.data() : nullptr;
    const float* const s_val = opt.val;
    const bool local = L.lds;
    parallel_segments(ptr, nseg, [&](uint32_t b, uint32_t e) {
        std::vector<uint32_t> cur(P);
        for (uint32_t c = b; c < e; ++c) {
            if (ptr[c + 1] == ptr[c]) continue;
            for (uint32_t p = 0; p < P; ++p) cur[p] = L.ptr_v[(size_t) p * nseg + c];
            uint32_t p = 0, lo = 0, hi = P == 1 ? 0xFFFFFFFFu : PR;
            for (uint32_t q = ptr[c]; q < ptr[c + 1]; ++q) {
                const uint32_t i = idx[q];
                if (i < lo || i >= hi) { p = i / PR; lo = p * PR; hi = lo + PR; }
                const uint32_t d = cur[p]++;
                const uint32_t li = local ? i - lo : i;
                if (d_idx16) d_idx16[d] = (uint16_t) li; else d_idx[d] = li;
                if (d_perm) d_perm[d] = q;
                if (d_val) d_val[d] = s_val ? s_val[q] : 0.f;
            }
        }
    });
    {   // padding: the tail of every panel (and of the whole stream)
        const uint32_t pad = L.pad_index();
        for (uint32_t p = 0; p < P; ++p) {
            const uint64_t end = p + 1 < P ? L.ptr_v[(size_t) (p + 1) * nseg] : L.padded_nnz;
            for (uint64_t d = panel_real_end[p]; d < end; ++d) {
                if (d_idx16) d_idx16[d] = (uint16_t) pad; else d_idx[d] = pad;
                if (d_perm) d_perm[d] = ~0u;
                if (d_val) d_val[d] = 0.f;
            }
        }
    }

    // 4. head flags, ranks and per-word head prefix counts over the virtual segments
    const size_t nwords = L.padded_nnz / 32;
    L.flags32.assign(nwords + 16, 0);  // two tiles of spare words: the kernel reads metadata two tiles ahead
    L.rank_of_seg.assign(nv, -1);
    L.seg_of_rank.clear();
    L.seg_of_rank.reserve(std::min<size_t>(nv, (size_t) nnz + P));
    for (size_t v = 0; v < nv; ++v) {
        if (L.ptr_v[v + 1] > L.ptr_v[v]) {
            L.rank_of_seg[v] = (int32_t) L.seg_of_rank.size();
            L.seg_of_rank.push_back((uint32_t) (v % nseg));
            const uint64_t head = L.ptr_v[v];
            L.flags32[head >> 5] |= 1u << (head & 31);
        }
    }
    L.nne = (uint32_t) L.seg_of_rank.size();
    L.hpre.assign(nwords + 16, 0);
    uint32_t run = 0;
    for (size_t w = 0; w < nwords + 16; ++w) {
        L.hpre[w] = run;
        run += (uint32_t) __builtin_popcount(L.flags32[w]);
    }
    {   // most ranks one workgroup chunk touches: sizes the LDS per-segment window (or its fallback)
        const uint64_t chunk_words = chunk / 32;
        for (uint64_t w0 = 0; w0 < nwords; w0 += chunk_words) {
            const uint32_t lo = L.hpre[w0] > 0 ? L.hpre[w0] - 1 : 0, hi = L.hpre[std::min<uint64_t>(w0 + chunk_words, nwords)];
            L.max_wg_ranks = std::max(L.max_wg_ranks, hi - lo);
        }
    }
    // 5. workgroup -> panel (LDS panels: a workgroup stages exactly one slice)
    if (PR && L.lds) {
        const uint32_t nwg = L.nspans / L.spans_per_wg;
        L.wg_panel.assign(nwg, 0);
        uint32_t p = 0;
        for (uint32_t w = 0; w < nwg; ++w) {
            const uint64_t start = (uint64_t) w * chunk;
            while (p + 1 < P && start >= L.ptr_v[(size_t) (p + 1) * nseg]) ++p;
            L.wg_panel[w] = p;
        }
    }
}

}  // namespace mfx

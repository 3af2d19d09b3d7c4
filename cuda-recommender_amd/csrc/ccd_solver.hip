// ccd_solver.hip -- host orchestration of CCD++ on one GPU (or one shard of a multi-GPU run).
//
// Replaces ccdpp_NV (cuda_src/CCD_CUDA.cu:224-451).  Differences that are design, not accident:
//   * one HIP stream, no hipDeviceSynchronize between launches (the reference syncs after every
//     launch group, CCD_CUDA.cu:197-200,216-219); the host waits once per outer iteration;
//   * default schedule fuses subtract(t-1) + add-back(t) + the first v-/u-sweep of rank t into
//     two passes (one over the CSC copy, one over the CSR copy): 24 B/nnz per rank instead of
//     (48+16T) B/nnz as written.  Arithmetic per element is unchanged (same roundings, same
//     order of the two updates); only the order of the fp32 reduction differs from the CPU.
//   * mfx_params.schedule = 0 runs the kernel sequence exactly as written, for A/B and parity.
#include "ccd_solver.hpp"
#include "layout_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>

namespace mfx {

namespace {
// MFX_SETUP_TIMING=1 prints where the one-time layout build spends its time.
struct PhaseTimer {
    bool on = getenv("MFX_SETUP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[mfx setup] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
int SegStreamStore::build(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx,
                          const float* val, mfx_memspace space, const FlatLayoutOptions& opt, int build_mode, hipStream_t st) {
    MFX_REQUIRE(build_mode >= 0 && build_mode <= 2, "layout_build must be 0 (auto), 1 (host) or 2 (device)");
    MFX_REQUIRE(nnz == 0 || idx, "null index array with %llu non-zeros", (unsigned long long) nnz);
    if (opt.panel_rows && opt.lds)
        MFX_REQUIRE(opt.spans_per_wg == 4 || opt.spans_per_wg == 8 || opt.spans_per_wg == 16, "wg_waves must be 4, 8 or 16");
    if (build_mode != 1) {
        bool done = false;
        MFX_TRY(build_device(nseg, nnz, G, ptr, idx, val, space, opt, st, &done));
        if (done) { built_on_device_ = true; MFX_TRY(build_fuse_tables(st)); return build_owner_lists(st); }
        MFX_REQUIRE(build_mode != 2, "layout_build = 2: the pattern is not grouped (some segment visits a panel more than "
                                     "once; sort the indices inside every row / column) -- the device builder cannot take it");
    }
    MFX_REQUIRE(!opt.scatter, "the scatter layout is built by the device pipeline only (pattern not grouped, or layout_build = 1)");
    MFX_TRY(build_host(nseg, nnz, G, ptr, idx, val, space, opt, st));
    MFX_TRY(build_fuse_tables(st));
    return build_owner_lists(st);
}

// (r4) Owner lists of the segment-owner fused passes (k_seg_owner): plain layout of a small matrix only.  Long segments
// (one workgroup each) and the rest (one wavefront each), both longest first so that the big items start first.
int SegStreamStore::build_owner_lists(hipStream_t st) {
    if (view.panel_rows != 0 || view.scatter || view.nseg == 0 || !view.ptr || view.nnz >= 4000000ull) return MFX_OK;
    std::vector<uint32_t> ptr_h((size_t) view.nseg + 1);
    MFX_HIP(hipMemcpyAsync(ptr_h.data(), view.ptr, sizeof(uint32_t) * ptr_h.size(), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> longs, shorts;
    const uint32_t thr = seg_owner_long_threshold();
    for (uint32_t c = 0; c < view.nseg; ++c) (ptr_h[c + 1] - ptr_h[c] >= thr ? longs : shorts).push_back(c);
    auto by_len = [&](uint32_t x, uint32_t y) { return ptr_h[x + 1] - ptr_h[x] > ptr_h[y + 1] - ptr_h[y]; };
    std::stable_sort(longs.begin(), longs.end(), by_len);
    std::stable_sort(shorts.begin(), shorts.end(), by_len);
    auto quads = [&](const std::vector<uint32_t>& v) {
        std::vector<uint32_t> out(std::max<size_t>(1, v.size()) * 4, 0u);
        for (size_t i = 0; i < v.size(); ++i) { out[4 * i] = v[i]; out[4 * i + 1] = ptr_h[v[i]]; out[4 * i + 2] = ptr_h[v[i] + 1]; }
        return out;
    };
    const std::vector<uint32_t> lq = quads(longs), sq = quads(shorts);
    MFX_TRY(own_long_.alloc(lq.size())); MFX_TRY(own_long_.upload(lq.data(), lq.size(), MFX_HOST, st));
    MFX_TRY(own_short_.alloc(sq.size())); MFX_TRY(own_short_.upload(sq.data(), sq.size(), MFX_HOST, st));
    MFX_HIP(hipStreamSynchronize(st));
    view.own_long = own_long_.get(); view.own_short = own_short_.get();
    view.own_nlong = (uint32_t) longs.size(); view.own_nshort = (uint32_t) shorts.size();
    for (uint32_t c = 0; c < view.nseg; ++c) view.own_max_len = std::max(view.own_max_len, ptr_h[c + 1] - ptr_h[c]);
    return MFX_OK;
}

// Fused finalize: which segment groups each workgroup chunk contributes to, how many chunks a group waits for, and
// the dispatch order (ascending first segment).  One small kernel finds every chunk's first / last segment; the
// rest is host arithmetic over a few thousand chunks.
int SegStreamStore::build_fuse_tables(hipStream_t st) {
    // LDS panels with 16-span workgroups (1024 threads), or (r4) the plain layout: 256-thread workgroups of four spans
    const bool plain = view.panel_rows == 0;
    if (!(plain || (view.lds_panels && view.spans_per_wg == 16)) || view.scatter || view.nseg == 0 || view.nspans == 0 || !view.ptr_v || !view.rank_code) return MFX_OK;
    const uint32_t wg_spans = plain ? 4u : 16u, block = plain ? 256u : 1024u;
    const uint32_t nchunks = (view.nspans + wg_spans - 1) / wg_spans;
    if (panel_end_dev_.size() == 0 || plain) {
        std::vector<uint32_t> one{(uint32_t) view.nnz};
        const std::vector<uint32_t>& pe = plain ? one : built_on_device_ ? panel_end_host_ : layout_.panel_real_end;
        if (pe.size() != view.npanels) return MFX_OK;  // no panel ends at hand: the separate finalize kernel stays
        MFX_TRY(panel_end_dev_.alloc(pe.size()));
        MFX_TRY(panel_end_dev_.upload(pe.data(), pe.size(), MFX_HOST, st));
        MFX_HIP(hipStreamSynchronize(st));
    }
    DevBuf<uint32_t> d_first, d_last;
    MFX_TRY(d_first.alloc(nchunks)); MFX_TRY(d_last.alloc(nchunks));
    MFX_TRY(launch_chunk_seg_range(view, wg_spans, panel_end_dev_.get(), d_first.get(), d_last.get(), st));
    std::vector<uint32_t> first(nchunks), last(nchunks);
    MFX_HIP(hipMemcpyAsync(first.data(), d_first.get(), sizeof(uint32_t) * nchunks, hipMemcpyDeviceToHost, st));
    MFX_HIP(hipMemcpyAsync(last.data(), d_last.get(), sizeof(uint32_t) * nchunks, hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    const uint32_t gs = fused_group_size(view.npanels, block), ngroups = (view.nseg + gs - 1) / gs;
    std::vector<uint32_t> g0(nchunks), g1(nchunks), expected(ngroups, 0u), order(nchunks), orphans;
    uint32_t max_groups = 0;
    for (uint32_t w = 0; w < nchunks; ++w) {
        if (first[w] == 0xFFFFFFFFu) { g0[w] = 1; g1[w] = 0; continue; }  // padding only: contributes nowhere
        MFX_REQUIRE(first[w] <= last[w] && last[w] < view.nseg, "fused finalize: chunk %u covers segments %u..%u", w, first[w], last[w]);
        g0[w] = first[w] / gs; g1[w] = last[w] / gs;
        for (uint32_t g = g0[w]; g <= g1[w]; ++g) ++expected[g];
        max_groups = std::max(max_groups, g1[w] - g0[w] + 1);
    }
    for (uint32_t g = 0; g < ngroups; ++g) if (expected[g] == 0) orphans.push_back(g);
    for (uint32_t w = 0; w < nchunks; ++w) order[w] = w;
    // MFX_FUSE_FINALIZE=2: dispatch in ascending order of the first segment, so that a group's chunks run at about
    // the same time and groups complete all along the pass (1: stored, panel-major order -- groups complete while the
    // last panel is processed).  Measured slower still: every panel's slice is then live at once.
    const char* fz_env = std::getenv("MFX_FUSE_FINALIZE");
    if (fz_env && std::atoi(fz_env) == 2)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });  // padding-only chunks last
    MFX_TRY(fz_order_.alloc(nchunks)); MFX_TRY(fz_order_.upload(order.data(), nchunks, MFX_HOST, st));
    MFX_TRY(fz_g0_.alloc(nchunks)); MFX_TRY(fz_g0_.upload(g0.data(), nchunks, MFX_HOST, st));
    MFX_TRY(fz_g1_.alloc(nchunks)); MFX_TRY(fz_g1_.upload(g1.data(), nchunks, MFX_HOST, st));
    MFX_TRY(fz_expected_.alloc(ngroups)); MFX_TRY(fz_expected_.upload(expected.data(), ngroups, MFX_HOST, st));
    MFX_TRY(fz_arrived_.alloc_zero(ngroups, st));
    MFX_TRY(fz_orphans_.alloc(orphans.empty() ? 1 : orphans.size()));
    MFX_TRY(fz_orphans_.upload(orphans.data(), orphans.size(), MFX_HOST, st));
    MFX_HIP(hipStreamSynchronize(st));  // the host vectors behind the uploads
    view.fz_order = fz_order_.get(); view.fz_g0 = fz_g0_.get(); view.fz_g1 = fz_g1_.get(); view.fz_expected = fz_expected_.get();
    view.fz_arrived = fz_arrived_.get(); view.fz_orphans = fz_orphans_.get(); view.fz_norphans = (uint32_t) orphans.size();
    view.fz_ngroups = ngroups; view.fz_max_chunk_groups = max_groups;
    return MFX_OK;
}

// Device pipeline (layout_kernels.hpp).  Host-resident inputs are uploaded as they are (12 B per
// non-zero and orientation) and everything else happens in HBM.
int SegStreamStore::build_device(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx,
                                 const float* val, mfx_memspace space, const FlatLayoutOptions& opt, hipStream_t st, bool* done) {
    *done = false;
    PhaseTimer tm;
    DevBuf<uint32_t> in_ptr, in_idx;
    DevBuf<float> in_val;
    LayoutBuildIn in;
    in.nseg = nseg; in.nnz = nnz; in.G = G;
    MFX_TRY(ptr_.alloc((size_t) nseg + 1));
    MFX_TRY(ptr_.upload(ptr, (size_t) nseg + 1, space, st));
    in.ptr = ptr_.get();
    if (space == MFX_HOST) {
        MFX_TRY(in_idx.alloc(nnz ? nnz : 1)); MFX_TRY(in_idx.upload(idx, nnz, MFX_HOST, st));
        if (val) { MFX_TRY(in_val.alloc(nnz ? nnz : 1)); MFX_TRY(in_val.upload(val, nnz, MFX_HOST, st)); }
        in.idx = in_idx.get(); in.val = val ? in_val.get() : nullptr;
    } else {
        in.idx = idx; in.val = val;
    }
    tm.lap("dev: inputs in HBM");
    // the same derived quantities as build_flat_layout
    const bool lds = opt.panel_rows ? opt.lds : true;
    const uint32_t P = opt.panel_rows ? std::max(1u, (G + opt.panel_rows - 1) / opt.panel_rows) : 1u;
    const uint32_t spans_per_wg = (opt.panel_rows && opt.lds) ? std::max(1u, opt.spans_per_wg) : 1u;
    uint32_t tps = opt.tiles_per_span ? opt.tiles_per_span : pick_tiles_per_span(nnz, opt.panel_rows != 0 && opt.lds);
    if (tps & 1) ++tps;
    const uint64_t span = (uint64_t) tps * kTileElems, chunk = span * spans_per_wg;
    const size_t nv = (size_t) P * nseg;
    MFX_REQUIRE((uint64_t) P * nseg < 0x7FFFFFFFull, "panels x segments exceeds the 32-bit virtual-segment range");
    if (opt.panel_rows && lds) MFX_REQUIRE(opt.panel_rows <= 0xFFFFu, "panel_rows must be <= 65535");
    in.npanels = P; in.panel_rows = opt.panel_rows; in.local_idx = opt.panel_rows != 0 && lds; in.idx16 = in.local_idx;
    in.pad_index = opt.panel_rows ? (lds ? opt.panel_rows : G) : 0u;
    in.span_len = (uint32_t) span; in.chunk = chunk; in.transpose_tiles = opt.scatter;

    MFX_TRY(lk_check_ptr(in, st));
    DevBuf<uint32_t> cnt, S, scratch, dstart, ddelta, v_of_rank, dmax;
    MFX_TRY(first_q_dev_.alloc(nv ? nv : 1));
    MFX_TRY(cnt.alloc(nv ? nv : 1));
    bool grouped = true;
    MFX_TRY(lk_runs_and_counts(in, first_q_dev_.get(), cnt.get(), &grouped, st));
    if (!grouped) { first_q_dev_.release(); ptr_.release(); return MFX_OK; }
    tm.lap("dev: runs + counts");
    // panel-major exclusive scan; panel starts -> padded panel bases (host arithmetic over P values)
    MFX_TRY(S.alloc(nv + 1));
    MFX_TRY(scratch.alloc(scan_scratch_words(nv)));
    MFX_TRY(lk_exclusive_scan(cnt.get(), S.get(), nv, false, scratch.get(), st));
    MFX_TRY(dstart.alloc((size_t) P + 1));
    MFX_TRY(lk_panel_starts(S.get(), nseg, P, dstart.get(), st));
    std::vector<uint32_t> ustart((size_t) P + 1), delta(P), real_end(P);
    MFX_HIP(hipMemcpyAsync(ustart.data(), dstart.get(), sizeof(uint32_t) * ustart.size(), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> base((size_t) P + 1);
    uint64_t pos = 0;
    for (uint32_t p = 0; p < P; ++p) {
        base[p] = pos;
        pos += ustart[p + 1] - ustart[p];
        real_end[p] = (uint32_t) pos;
        pos = (pos + chunk - 1) / chunk * chunk;
    }
    if (pos == 0) pos = chunk;  // nnz == 0: keep one (all padding) chunk so that grids are never empty
    base[P] = pos;
    const uint64_t padded = pos;
    MFX_REQUIRE(padded < 0xFFFFFF00ull, "padded non-zero count exceeds the 32-bit position range");
    for (uint32_t p = 0; p < P; ++p) delta[p] = (uint32_t) base[p] - ustart[p];
    MFX_TRY(ddelta.alloc(P)); MFX_TRY(ddelta.upload(delta.data(), P, MFX_HOST, st));
    MFX_TRY(ptr_v_.alloc(nv + 1));
    MFX_TRY(lk_ptr_v(S.get(), ddelta.get(), nseg, nv, (uint32_t) padded, ptr_v_.get(), st));
    // head bits, their prefix counts, ranks
    const size_t nwords = padded / 32;
    MFX_TRY(flags32_.alloc_zero(nwords + 16, st));
    MFX_TRY(lk_heads(ptr_v_.get(), nv, flags32_.get(), st));
    MFX_TRY(hpre_.alloc(nwords + 17));
    if (scan_scratch_words(nwords + 16) > scratch.size()) MFX_TRY(scratch.alloc(scan_scratch_words(nwords + 16)));
    MFX_TRY(lk_exclusive_scan(flags32_.get(), hpre_.get(), nwords + 16, true, scratch.get(), st));
    uint32_t nne = 0;
    MFX_HIP(hipMemcpyAsync(&nne, hpre_.get() + nwords + 16, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    MFX_REQUIRE(nne < 0x80000000u, "too many virtual segments");
    MFX_TRY(rank_code_.alloc(nv ? nv : 1));
    MFX_TRY(seg_of_rank_.alloc(nne ? nne : 1));
    MFX_TRY(v_of_rank.alloc(nne ? nne : 1));
    MFX_TRY(lk_ranks(ptr_v_.get(), nv, nseg, (uint32_t) span, flags32_.get(), hpre_.get(), rank_code_.get(), seg_of_rank_.get(),
                     v_of_rank.get(), st));
    tm.lap("dev: scans, heads, ranks");
    // stored order
    if (in.idx16) MFX_TRY(idx16_.alloc(padded)); else MFX_TRY(idx_.alloc(padded));
    MFX_TRY(val_.alloc(padded));
    if (opt.scatter) MFX_TRY(segid_.alloc(padded));
    MFX_TRY(lk_place(in, padded, ptr_v_.get(), first_q_dev_.get(), cnt.get(), flags32_.get(), hpre_.get(), v_of_rank.get(),
                     in.idx16 ? static_cast<void*>(idx16_.get()) : static_cast<void*>(idx_.get()), val_.get(),
                     opt.scatter ? segid_.get() : nullptr, st));
    if (opt.scatter && !opt.scatter_ids32) {  // ids as one-byte steps + a base per tile (11 instead of 14 B per non-zero streamed)
        MFX_TRY(seg_delta_.alloc(padded));
        MFX_TRY(tile_base_.alloc(padded / kTileElems));
        bool fits = false;
        MFX_TRY(lk_delta_encode(segid_.get(), padded, seg_delta_.get(), tile_base_.get(), &fits, st));
        if (fits) segid_.release();
        else { seg_delta_.release(); tile_base_.release(); }  // a step above 255 somewhere: a long run of segments without an entry in some panel
    }
    MFX_TRY(dmax.alloc_zero(1, st));
    MFX_TRY(lk_max_wg_ranks(hpre_.get(), nwords, (size_t) (chunk / 32), dmax.get(), st));
    uint32_t max_wg_ranks = 0;
    MFX_HIP(hipMemcpyAsync(&max_wg_ranks, dmax.get(), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    // per-segment counts are the single-panel counts; with one panel that is `cnt` itself
    MFX_TRY(seg_cnt_.alloc(nseg));
    {
        LayoutBuildIn one = in;
        one.npanels = 1; one.nnz = 0;  // counts only: no index pass
        DevBuf<uint32_t> tmp_first;
        MFX_TRY(tmp_first.alloc(nseg));
        bool g2 = true;
        MFX_TRY(lk_runs_and_counts(one, tmp_first.get(), seg_cnt_.get(), &g2, st));
    }
    const uint32_t nspans = (uint32_t) (padded / span);
    std::vector<uint32_t> wg_panel;
    if (opt.panel_rows && lds) {  // workgroup -> panel (a workgroup stages exactly one slice)
        const uint32_t nwg = nspans / spans_per_wg;
        wg_panel.assign(nwg, 0);
        uint32_t p = 0;
        for (uint32_t w = 0; w < nwg; ++w) {
            const uint64_t start = (uint64_t) w * chunk;
            while (p + 1 < P && start >= base[p + 1]) ++p;
            wg_panel[w] = p;
        }
    }
    MFX_TRY(wg_panel_.alloc(wg_panel.empty() ? 1 : wg_panel.size()));
    MFX_TRY(wg_panel_.upload(wg_panel.data(), wg_panel.size(), MFX_HOST, st));
    MFX_TRY(panel_end_dev_.alloc(P)); MFX_TRY(panel_end_dev_.upload(real_end.data(), P, MFX_HOST, st));
    uint32_t scat_nwg = 0;
    if (opt.scatter) {  // persistent workgroups: chunk ranges, the slabs they write, first slab of every panel; no partials / carries
        MFX_REQUIRE(opt.panel_rows && lds && spans_per_wg == 16, "scatter layout needs LDS panels and 16-span workgroups");
        int dev = 0, cus = 0;
        MFX_HIP(hipGetDevice(&dev));
        MFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        // one workgroup per CU (a workgroup takes the whole LDS) unless the solver asks for fewer (CUs left to a concurrent
        // collective); MFX_SCATTER_WGS overrides for experiments and tests
        uint32_t want = opt.scatter_wgs ? opt.scatter_wgs : (cus > 0 ? (uint32_t) cus : 256u);
        if (const char* e = std::getenv("MFX_SCATTER_WGS")) { const int v = std::atoi(e); if (v > 0) want = (uint32_t) v; }
        // panel groups (consecutive panels; one launch each, SegStreamDev::scat_ngroups): every launch spreads ITS chunks
        // over the workgroups; the slabs are numbered across the groups, so a panel's slabs stay consecutive
        const uint32_t ngroups = std::max(1u, std::min(std::min(opt.scatter_groups, P), (uint32_t) SegStreamDev::kMaxScatterGroups));
        std::vector<uint32_t> chunk_lo, slab0, slab_lo((size_t) P + 1, 0u), slabs_of_panel(P, 0u);
        uint32_t nslabs = 0;
        for (uint32_t g = 0; g < ngroups; ++g) {
            const uint32_t p_lo = (uint32_t) ((uint64_t) P * g / ngroups), p_hi = (uint32_t) ((uint64_t) P * (g + 1) / ngroups);
            const uint32_t c_lo = (uint32_t) (base[p_lo] / chunk), c_hi = (uint32_t) (base[p_hi] / chunk);
            const uint32_t nwg = c_hi > c_lo ? std::max(1u, std::min(want, c_hi - c_lo)) : 0u;
            grp_nwg_[g] = nwg; grp_tab_[g] = (uint32_t) chunk_lo.size(); grp_wg0_[g] = (uint32_t) slab0.size();
            grp_lo_[g] = std::min<uint64_t>((uint64_t) p_lo * opt.panel_rows, G);
            scat_nwg = std::max(scat_nwg, nwg);
            if (nwg == 0) continue;
            const size_t t0 = chunk_lo.size();
            for (uint32_t w = 0; w <= nwg; ++w) chunk_lo.push_back(c_lo + (uint32_t) ((uint64_t) (c_hi - c_lo) * w / nwg));
            for (uint32_t w = 0; w < nwg; ++w) {
                slab0.push_back(nslabs);
                for (uint32_t c = chunk_lo[t0 + w]; c < chunk_lo[t0 + w + 1]; ++c)
                    if (c == chunk_lo[t0 + w] || wg_panel[c] != wg_panel[c - 1]) { ++nslabs; ++slabs_of_panel[wg_panel[c]]; }
            }
        }
        grp_lo_[ngroups] = G;
        ngroups_ = ngroups;
        // ranges and panels both ascend, so the slabs of a panel are consecutive in the numbering above
        for (uint32_t p = 0; p < P; ++p) slab_lo[p + 1] = slab_lo[p] + slabs_of_panel[p];
        MFX_REQUIRE(slab_lo[P] == nslabs, "scatter layout: slab bookkeeping is inconsistent (%u vs %u)", slab_lo[P], nslabs);
        MFX_TRY(slab_lo_.alloc(slab_lo.size())); MFX_TRY(slab_lo_.upload(slab_lo.data(), slab_lo.size(), MFX_HOST, st));
        MFX_TRY(scat_chunk_lo_.alloc(chunk_lo.size())); MFX_TRY(scat_chunk_lo_.upload(chunk_lo.data(), chunk_lo.size(), MFX_HOST, st));
        MFX_TRY(scat_slab0_.alloc(slab0.size())); MFX_TRY(scat_slab0_.upload(slab0.data(), slab0.size(), MFX_HOST, st));
        MFX_TRY(scat_slab_bad_.alloc_zero(std::max(1u, nslabs), st));
        MFX_TRY(wgacc_.alloc((size_t) std::max(1u, nslabs) * 2 * opt.panel_rows));
        MFX_HIP(hipStreamSynchronize(st));  // the host vectors behind the uploads
    } else {
        MFX_TRY(part_.alloc_zero(nne ? nne : 1, st));
        MFX_TRY(carry_.alloc_zero(nspans, st));
    }
    MFX_HIP(hipStreamSynchronize(st));  // host vectors behind the uploads, temporaries behind the kernels
    tm.lap("dev: placement + rest");
    if (opt.scatter) { flags32_.release(); hpre_.release(); rank_code_.release(); seg_of_rank_.release(); }

    layout_ = FlatLayoutHost();  // scalars only: the arrays live in HBM
    layout_.nseg = nseg; layout_.gather_len = G; layout_.npanels = P; layout_.panel_rows = opt.panel_rows; layout_.lds = lds;
    layout_.spans_per_wg = spans_per_wg; layout_.nne = nne; layout_.nspans = nspans; layout_.tiles_per_span = tps;
    layout_.nnz = nnz; layout_.padded_nnz = padded; layout_.max_wg_ranks = max_wg_ranks; layout_.perm_is_runs = true;
    view = SegStreamDev();
    view.nseg = nseg; view.nne = nne; view.nnz = nnz; view.padded_nnz = padded; view.nspans = nspans;
    view.tiles_per_span = tps; view.npanels = P; view.panel_rows = opt.panel_rows;
    view.lds_panels = opt.panel_rows != 0 && lds;
    view.spans_per_wg = spans_per_wg; view.gather_len = G;
    view.ptr = ptr_.get(); view.ptr_v = ptr_v_.get(); view.seg_cnt = seg_cnt_.get(); view.idx = idx_.get();
    view.idx16 = idx16_.get();
    view.val = val_.get(); view.flags32 = flags32_.get(); view.hpre = hpre_.get(); view.rank_code = rank_code_.get();
    view.seg_of_rank = seg_of_rank_.get(); view.max_wg_ranks = max_wg_ranks;
    view.wg_panel = wg_panel_.get(); view.perm = nullptr; view.part = part_.get();
    view.carry = carry_.get();
    view.scatter = opt.scatter; view.segid = segid_.get(); view.seg_delta = seg_delta_.get(); view.tile_base = tile_base_.get(); view.wgacc = wgacc_.get(); view.slab_lo = slab_lo_.get();
    view.scat_nwg = scat_nwg; view.scat_chunk_lo = scat_chunk_lo_.get(); view.scat_slab0 = scat_slab0_.get(); view.scat_slab_bad = scat_slab_bad_.get();
    if (opt.scatter) {
        view.scat_ngroups = ngroups_;
        for (uint32_t g = 0; g < ngroups_; ++g) { view.scat_grp_nwg[g] = grp_nwg_[g]; view.scat_grp_tab[g] = grp_tab_[g]; view.scat_grp_wg0[g] = grp_wg0_[g]; }
        for (uint32_t g = 0; g <= ngroups_; ++g) view.scat_grp_lo[g] = grp_lo_[g];
    }
    *done = true;
    return MFX_OK;
}

int SegStreamStore::build_host(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx,
                               const float* val, mfx_memspace space, const FlatLayoutOptions& opt, hipStream_t st) {
    // The host builder (one pass over the pattern, a few threads); device-resident inputs are brought
    // down once for it.  Takes every pattern, and is the checker of the device pipeline in the tests.
    PhaseTimer tm;
    std::vector<uint32_t> ptr_buf, idx_buf;
    std::vector<float> val_buf;
    const uint32_t* ptr_h = ptr;
    const uint32_t* idx_h = idx;
    const float* val_h = val;
    if (space == MFX_DEVICE) {
        ptr_buf.resize((size_t) nseg + 1);
        MFX_HIP(hipMemcpy(ptr_buf.data(), ptr, sizeof(uint32_t) * ptr_buf.size(), hipMemcpyDeviceToHost));
        ptr_h = ptr_buf.data();
        idx_buf.resize(nnz);
        if (nnz) MFX_HIP(hipMemcpy(idx_buf.data(), idx, sizeof(uint32_t) * nnz, hipMemcpyDeviceToHost));
        idx_h = idx_buf.data();
        if (val) {
            val_buf.resize(nnz);
            if (nnz) MFX_HIP(hipMemcpy(val_buf.data(), val, sizeof(float) * nnz, hipMemcpyDeviceToHost));
            val_h = val_buf.data();
        }
    }
    tm.lap("inputs to host");
    MFX_REQUIRE(ptr_h[0] == 0 && ptr_h[nseg] == nnz, "segment pointer array does not span [0, nnz]");
    for (uint32_t c = 0; c < nseg; ++c)
        MFX_REQUIRE(ptr_h[c] <= ptr_h[c + 1], "segment pointer array is not monotone at %u", c);
    {   // every gathered index must be in range: checked here, on the host, so that a bad input is an
        // error message and not a GPU fault
        struct Ctx { const uint32_t* idx; uint32_t G; std::atomic<uint64_t> bad; } cx{idx_h, G, {~0ull}};
        parallel_ranges_u64(nnz, [](uint64_t b, uint64_t e, void* p) {
            Ctx& c = *static_cast<Ctx*>(p);
            for (uint64_t q = b; q < e; ++q)
                if (c.idx[q] >= c.G) { uint64_t cur = c.bad.load(); while (q < cur && !c.bad.compare_exchange_weak(cur, q)) {} break; }
        }, &cx);
        const uint64_t bad = cx.bad.load();
        MFX_REQUIRE(bad == ~0ull, "index %u at position %llu is out of range [0, %u)", idx_h[bad == ~0ull ? 0 : bad],
                    (unsigned long long) bad, G);
    }
    tm.lap("validation");
    FlatLayoutOptions bopt = opt;  // one pass places indices (16-bit for LDS panels), provenance and values
    bopt.emit_idx16 = opt.panel_rows != 0 && opt.lds;
    bopt.emit_val = true;
    bopt.compact_perm = true;  // ascending indices (the usual input): no 4 B/nnz provenance array at all
    bopt.val = val_h;
    if (bopt.emit_idx16) MFX_REQUIRE(opt.panel_rows <= 0xFFFFu, "panel_rows must be <= 65535");
    build_flat_layout(ptr_h, idx_h, nseg, nnz, G, bopt, &layout_);
    tm.lap("build_flat_layout");
    FlatLayoutHost& L = layout_;
    MFX_REQUIRE(L.padded_nnz < 0xFFFFFF00ull, "padded non-zero count exceeds the 32-bit position range");
    MFX_REQUIRE((uint64_t) L.npanels * nseg < 0x7FFFFFFFull, "panels x segments exceeds the 32-bit virtual-segment range");

    const size_t nv = (size_t) L.npanels * nseg;
    MFX_TRY(ptr_.alloc((size_t) nseg + 1));
    MFX_TRY(ptr_.upload(ptr_h, (size_t) nseg + 1, MFX_HOST, st));
    MFX_TRY(ptr_v_.alloc(nv + 1));
    MFX_TRY(ptr_v_.upload(L.ptr_v.data(), nv + 1, MFX_HOST, st));
    MFX_TRY(seg_cnt_.alloc(nseg));
    MFX_TRY(seg_cnt_.upload(L.seg_cnt.data(), nseg, MFX_HOST, st));
    if (L.panel_rows && L.lds) {  // panel-local indices (and the zero slot, index panel_rows) fit 16 bits
        MFX_TRY(idx16_.alloc(L.padded_nnz));
        MFX_TRY(idx16_.upload(L.idx16.data(), L.padded_nnz, MFX_HOST, st));
    } else {
        MFX_TRY(idx_.alloc(L.padded_nnz));
        MFX_TRY(idx_.upload(L.idx_local.data(), L.padded_nnz, MFX_HOST, st));
    }
    if (L.perm_is_runs) {
        first_q_host_.assign(L.first_q.begin(), L.first_q.end());
        panel_end_host_.assign(L.panel_real_end.begin(), L.panel_real_end.end());
    } else {
        MFX_TRY(perm_.alloc(L.padded_nnz));
        MFX_TRY(perm_.upload(L.perm.data(), L.padded_nnz, MFX_HOST, st));
    }
    MFX_TRY(val_.alloc(L.padded_nnz));
    MFX_TRY(val_.upload(L.val_st.data(), L.padded_nnz, MFX_HOST, st));
    MFX_TRY(flags32_.alloc(L.flags32.size()));
    MFX_TRY(flags32_.upload(L.flags32.data(), L.flags32.size(), MFX_HOST, st));
    MFX_TRY(hpre_.alloc(L.hpre.size()));
    MFX_TRY(hpre_.upload(L.hpre.data(), L.hpre.size(), MFX_HOST, st));
    // rank per virtual segment + "runs into later spans" bit: the finalize then only reads a segment's
    // pointers when it has carries to add (rare), 12 instead of 20 bytes per virtual segment
    MFX_REQUIRE(L.nne < 0x80000000u, "too many virtual segments");
    std::vector<uint32_t> rank_code(nv);
    {
        struct Ctx { uint32_t* dst; const int32_t* rank; const uint32_t* ptr_v; uint32_t span; } cx{rank_code.data(), L.rank_of_seg.data(), L.ptr_v.data(), L.span_len()};
        parallel_ranges_u64(nv, [](uint64_t b, uint64_t e, void* p) {
            Ctx& c = *static_cast<Ctx*>(p);
            for (uint64_t v = b; v < e; ++v) {
                if (c.rank[v] < 0) { c.dst[v] = 0xFFFFFFFFu; continue; }
                const uint32_t lo = c.ptr_v[v], hi = c.ptr_v[v + 1];
                c.dst[v] = (uint32_t) c.rank[v] | ((lo / c.span != (hi - 1) / c.span) ? 0x80000000u : 0u);
            }
        }, &cx);
    }
    MFX_TRY(rank_code_.alloc(nv ? nv : 1));
    MFX_TRY(rank_code_.upload(rank_code.data(), nv, MFX_HOST, st));
    MFX_TRY(seg_of_rank_.alloc(L.nne ? L.nne : 1));
    MFX_TRY(seg_of_rank_.upload(L.seg_of_rank.data(), L.nne, MFX_HOST, st));
    MFX_TRY(wg_panel_.alloc(L.wg_panel.empty() ? 1 : L.wg_panel.size()));
    MFX_TRY(wg_panel_.upload(L.wg_panel.data(), L.wg_panel.size(), MFX_HOST, st));
    MFX_TRY(part_.alloc_zero(L.nne ? L.nne : 1, st));
    MFX_TRY(carry_.alloc_zero(L.nspans, st));
    // the host vectors behind the async uploads must outlive the copies
    MFX_HIP(hipStreamSynchronize(st));
    tm.lap("allocations + uploads");

    view.nseg = nseg; view.nne = L.nne; view.nnz = nnz; view.padded_nnz = L.padded_nnz; view.nspans = L.nspans;
    view.tiles_per_span = L.tiles_per_span; view.npanels = L.npanels; view.panel_rows = L.panel_rows;
    view.lds_panels = L.panel_rows != 0 && L.lds;
    view.spans_per_wg = L.spans_per_wg; view.gather_len = G;
    view.ptr = ptr_.get(); view.ptr_v = ptr_v_.get(); view.seg_cnt = seg_cnt_.get(); view.idx = idx_.get();
    view.idx16 = idx16_.get();
    view.val = val_.get(); view.flags32 = flags32_.get(); view.hpre = hpre_.get(); view.rank_code = rank_code_.get();
    view.seg_of_rank = seg_of_rank_.get(); view.max_wg_ranks = L.max_wg_ranks;
    view.wg_panel = wg_panel_.get(); view.perm = perm_.get(); view.part = part_.get();
    view.carry = carry_.get();
    // the big host-side vectors are no longer needed
    for (auto* v : {&L.idx_local, &L.perm, &L.first_q}) { v->clear(); v->shrink_to_fit(); }
    L.idx16.clear(); L.idx16.shrink_to_fit();
    L.val_st.clear(); L.val_st.shrink_to_fit();
    for (auto* v : {&L.flags32, &L.hpre}) { v->clear(); v->shrink_to_fit(); }
    return MFX_OK;
}

int SegStreamStore::unpermute(float* out, hipStream_t st) {
    if (view.perm) return launch_unpermute(view, out, st);
    if (!built_on_device_ && first_q_dev_.size() == 0 && !first_q_host_.empty()) {
        MFX_TRY(first_q_dev_.alloc(first_q_host_.size()));
        MFX_TRY(first_q_dev_.upload(first_q_host_.data(), first_q_host_.size(), MFX_HOST, st));
        MFX_TRY(panel_end_dev_.alloc(panel_end_host_.size()));
        MFX_TRY(panel_end_dev_.upload(panel_end_host_.data(), panel_end_host_.size(), MFX_HOST, st));
        MFX_HIP(hipStreamSynchronize(st));
    }
    return launch_unpermute_runs(view, first_q_dev_.get(), panel_end_dev_.get(), out, st);
}

// LDS panels pay off when the gathered vector is too big for L1 yet cutting it into LDS-sized
// panels leaves virtual segments long enough to amortise the per-segment bookkeeping.
// LDS bytes per gathered index of the CSR copy's slice: (v_prev_new, v_t_old, v_t_new) -- ccd_kernels.hip, ModeTraits<FM_FCSR>::S
constexpr uint32_t kCsrSliceBytes = 12;

FlatLayoutOptions choose_layout(const mfx_params& p, uint32_t nseg, uint64_t nnz, uint32_t G, uint32_t elem_bytes,
                                bool need_plain) {
    FlatLayoutOptions o;
    o.tiles_per_span = p.tiles_per_span > 0 ? (uint32_t) p.tiles_per_span : 0;
    // 16 waves x 2 workgroups (64 KB of LDS each) = 32 resident waves per CU
    o.spans_per_wg = p.wg_waves > 0 ? (uint32_t) p.wg_waves : 16;
    o.panel_rows = 0;
    if (need_plain || p.panel_rows == -1) return o;
    if (p.panel_rows < -1) {  // explicit cache panels of -panel_rows entries
        o.panel_rows = std::min<uint32_t>((uint32_t) -p.panel_rows, G ? G : 1u);
        o.lds = false;
        return o;
    }
    // Small matrices: a pass is a few microseconds, the operand vectors live in L1/L2, and a 56 KB
    // slice load per workgroup would cost more than the streaming it serves (measured, ML-1M shape
    // k = 40: 0.86 ms per outer iteration plain vs 1.57 ms with panels; ML-10M shape: panels win 1.65x).
    if (p.panel_rows == 0 && nnz < 4000000ull) return o;
    // 64 KB of LDS per workgroup (two 1024-thread workgroups per CU): 8 KB for the staged per-segment
    // operands, 56 KB for the slice.  Measured on the Netflix shape (round-1 sweep with a temporary env knob): slices of
    // 40/48/56/64/72 KB give 30.1/29.9/28.0/29.1/28.4 ms per outer iteration.
    // (round 2, 12-byte CSR slice entries, CSC / CSR pass per launch: 48 KB 199 / 230 us, 56 KB 183 / 191, 64 KB 187 / 231,
    // 71 KB 187 / 192 -- profiles/r02_exp_slice.txt)
    uint32_t slice_kb = 56;
    if (const char* e = std::getenv("MFX_SLICE_KB")) { const int v = std::atoi(e); if (v >= 8 && v <= 150) slice_kb = (uint32_t) v; }  // (A/B)
    uint32_t pr = p.panel_rows > 0 ? (uint32_t) p.panel_rows : (slice_kb * 1024u) / elem_bytes - 1;
    if (pr >= G) pr = G;  // the whole gathered vector fits: one panel
    const uint64_t npanels = (G + pr - 1) / pr;
    // (r4) EQUAL panels: the same count, but ceil(G / npanels) entries each instead of full slices and a remainder.  A short last
    // panel is a load imbalance (its workgroups finish early), and the smaller slices can let a third workgroup onto a CU (ML-10M
    // shape, CSR copy: 4777 / 4777 / 1123 columns -> 3 x 3559: 51 instead of 64 KB of LDS per workgroup): fused CSR pass 37.9 -> 26.5 us,
    // outer iteration at k = 40 2.78 -> 2.32 ms; ML-20M shape 5.07 -> 4.47 ms; 300 000 x 40 000 with 3e7 ratings 6.36 -> 5.67 ms;
    // Netflix shape 25.0 -> 24.9 ms (profiles/r04_exp_midsize.txt).  MFX_EQUAL_PANELS=0: full slices + remainder (A/B).
    if (p.panel_rows == 0 && npanels > 1) {
        const char* e = std::getenv("MFX_EQUAL_PANELS");
        if (!(e && std::atoi(e) == 0)) pr = (uint32_t) ((G + npanels - 1) / npanels);
    }
    const double mean_vseg = (double) nnz / ((double) npanels * (double) (nseg ? nseg : 1));
    // ... unless the matrix is small enough that the per-(panel, segment) bookkeeping stays cheap: up to 8 M virtual
    // segments the LDS panels still win (700 000 x 40 000 with 3e7 / 1.5e7 / 8e6 ratings, 7.1 / 3.6 / 1.9 entries per
    // pair: 7.7 / 5.4 / 3.9 ms per outer iteration at k = 32 against 10.1 / 7.3 / 5.1 ms in the scatter layout; at 2.8e7
    // virtual segments -- 2 M x 100 k, 1 M x 200 k -- the scatter layout wins by 1.4x, at 1.75e8 -- config 5's shard --
    // by 2.4x: k_finalize's cost grows with the virtual segments, the scatter pass's with the panels).
    const bool small_dims = npanels * (uint64_t) (nseg ? nseg : 1) <= 8000000ull && mean_vseg >= 1.5;
    if (p.panel_rows == 0 && npanels > 1 && mean_vseg < 8.0 && !small_dims) {
        // LDS-sized panels would shred the segments (hyper-sparse shard: < 8 entries per (panel, segment)
        // pair).  Cut at L2 granularity instead: 2 MB slices keep every gather an L2 hit (measured,
        // tools/ubench_gather.hip sweep: 100 M 8-byte gathers take 0.60 ms against a 2 MB table, 1.01 ms
        // against 8 MB), with whole segments of >= 8 entries per slice or not at all.
        const uint32_t cpr = (2u << 20) / (elem_bytes == kCsrSliceBytes ? 16u : elem_bytes);  // cache panels gather the float4 itself
        if (cpr >= G) return o;
        const uint64_t cpanels = (G + cpr - 1) / cpr;
        if ((double) nnz / ((double) cpanels * (double) (nseg ? nseg : 1)) < 8.0) return o;
        o.panel_rows = cpr;
        o.lds = false;
        return o;
    }
    o.panel_rows = pr;
    // A workgroup stages the per-segment operands of the ranks it touches in an LDS window of 1024 entries
    // (ccd_kernels.hip); ranks beyond it fall back to two dependent global loads each.  Keep a workgroup's
    // chunk (spans_per_wg spans) at ~0.85 x 1024 virtual segments of mean length: found at Z = 9.9e8
    // (4.8 M x 17 770, 41 entries per (panel, row) pair), where 16-tile spans put 1 600 ranks in every
    // chunk and the CSR pass fell from 0.79 to 0.57 of the HBM roofline.
    if (o.tiles_per_span == 0 && npanels >= 1) {
        const double chunk_cap = 0.85 * 1024.0 * std::max(1.0, mean_vseg);
        uint32_t cap = (uint32_t) std::min<double>(16.0, chunk_cap / ((double) o.spans_per_wg * kTileElems));
        cap &= ~1u;  // spans are consumed in tile pairs
        const uint32_t want = pick_tiles_per_span(nnz, true);
        o.tiles_per_span = std::max<uint32_t>(2u, std::min<uint32_t>(want, std::max<uint32_t>(2u, cap)));
    }
    return o;
}

// ------------------------------------------------------------------------------------------------
static const char* kKernelNames[KernelProfiler::K_COUNT] = {
    "ccd_fused_csc_pass", "ccd_fused_csr_pass", "ccd_flat_sweep", "ccd_flat_resid", "ccd_finalize",
    "ccd_combine_dense", "ccd_pack", "test_rmse", "rccl_allreduce", "ccd_wave_sweep", "ccd_wave_resid",
    "ccd_scatter_v_pass", "ccd_scatter_u_pass", "ccd_scatter_sweep", "ccd_scatter_resid", "ccd_scatter_combine",
    "ccd_ref_order_sweep", "host_enqueue_outer_iteration"};

const char* KernelProfiler::name(int id) { return id >= 0 && id < K_COUNT ? kKernelNames[id] : "?"; }

KernelProfiler::~KernelProfiler() {
    for (hipEvent_t e : pool_) (void) hipEventDestroy(e);
}

int KernelProfiler::take(hipEvent_t* e) {
    if (used_ == pool_.size()) {
        hipEvent_t n;
        MFX_HIP(hipEventCreate(&n));
        pool_.push_back(n);
    }
    *e = pool_[used_++];
    return MFX_OK;
}

int KernelProfiler::begin(int id, hipStream_t st) {
    if (!on_) return MFX_OK;
    Rec r;
    r.id = id;
    MFX_TRY(take(&r.a));
    MFX_TRY(take(&r.b));
    MFX_HIP(hipEventRecord(r.a, st));
    recs_.push_back(r);
    return MFX_OK;
}

int KernelProfiler::end(hipStream_t st) {
    if (!on_) return MFX_OK;
    MFX_HIP(hipEventRecord(recs_.back().b, st));
    return MFX_OK;
}

int KernelProfiler::collect() {
    for (const Rec& r : recs_) {
        float ms = 0.f;
        MFX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        seconds[r.id] += ms * 1e-3;
        launches[r.id] += 1;
    }
    recs_.clear();
    used_ = 0;
    return MFX_OK;
}

void KernelProfiler::reset_totals() {
    for (int i = 0; i < K_COUNT; ++i) { seconds[i] = 0; launches[i] = 0; }
}

#define PROFS(id, stream, call)               \
    do {                                      \
        MFX_TRY(prof_.begin((id), (stream))); \
        MFX_TRY(call);                        \
        MFX_TRY(prof_.end((stream)));         \
    } while (0)
#define PROF(id, call) PROFS(id, st_, call)

// ------------------------------------------------------------------------------------------------
int CcdSolver::create(CcdSolver** out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                      mfx_memspace space, const mfx_shard* shard) {
    MFX_REQUIRE(out && R && p, "mfx_ccd_create: null argument");
    std::unique_ptr<CcdSolver> s(new CcdSolver());
    MFX_TRY(s->init(R, T, p, space, shard));
    *out = s.release();
    return MFX_OK;
}

CcdSolver::~CcdSolver() {
    (void) hipSetDevice(device_);
    if (st_) (void) hipStreamSynchronize(st_);
    if (graph_exec_) (void) hipGraphExecDestroy(graph_exec_);
    if (graph_) (void) hipGraphDestroy(graph_);
    for (hipEvent_t& e : ev_)
        if (e) (void) hipEventDestroy(e);
    for (hipEvent_t& e : ev_rank_)
        if (e) (void) hipEventDestroy(e);
    for (hipEvent_t& e : ev_grp_)
        if (e) (void) hipEventDestroy(e);
    if (ev_join_) (void) hipEventDestroy(ev_join_);
    if (st2_) {
        (void) hipStreamSynchronize(st2_);
        (void) hipStreamDestroy(st2_);
    }
    if (ref_streams_.side) {
        (void) hipStreamSynchronize(ref_streams_.side);
        (void) hipStreamDestroy(ref_streams_.side);
    }
    if (ref_streams_.fork) (void) hipEventDestroy(ref_streams_.fork);
    if (ref_streams_.join) (void) hipEventDestroy(ref_streams_.join);
    if (st_) {
        (void) hipStreamSynchronize(st_);
        (void) hipStreamDestroy(st_);
    }
}

int CcdSolver::init(const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space,
                    const mfx_shard* shard) {
    MFX_REQUIRE(R->rows > 0 && R->cols > 0 && R->nnz >= 0, "bad matrix shape %lld x %lld, nnz %lld",
                (long long) R->rows, (long long) R->cols, (long long) R->nnz);
    MFX_REQUIRE(R->rows < (int64_t) 0xFFFFFFFFll && R->cols < (int64_t) 0xFFFFFFFFll &&
                    R->nnz < (int64_t) 0xFFFF0000ll, "matrix exceeds 32-bit index range");
    MFX_REQUIRE(p->k >= 1, "k must be >= 1");
    MFX_REQUIRE(p->maxinneriter >= 1, "maxinneriter must be >= 1");
    MFX_REQUIRE(R->nnz == 0 || (R->csc_col_ptr && R->csc_row_idx && R->csc_val && R->csr_row_ptr &&
                                R->csr_col_idx && R->csr_val), "null CSR/CSC array");
    p_ = *p;
    device_ = p->device;
    MFX_TRY(use_device(device_));
    MFX_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    for (hipEvent_t& e : ev_) MFX_HIP(hipEventCreate(&e));
    m_ = (uint32_t) R->rows; n_ = (uint32_t) R->cols; k_ = p->k; nnz_ = (uint64_t) R->nnz;
    prof_.enable(p->profile != 0 || p->schedule == 0);
    if (const char* e = std::getenv("MFX_FUSE_FINALIZE")) fuse_finalize_ = std::atoi(e) != 0 ? 1 : 0;  // opt-in: see fuse_finalize_

    // Layouts.  Hyper-sparse shapes (LDS-sized panels would leave < 8 entries per (panel, segment) pair on
    // either side, so that side would fall back to L2 "cache panels") take the scatter layout on BOTH sides
    // (ccd_scatter.hip); kernel_variant = 2 forces it, panel_rows != 0 or kernel_variant = 0 rule it out.
    MFX_REQUIRE(p->kernel_variant >= -1 && p->kernel_variant <= 3, "kernel_variant must be -1 ... 3");
    MFX_REQUIRE(p->kernel_variant != -1 || p->schedule == 0, "kernel_variant = -1 (reference-order sweeps) goes with schedule = 0");
    MFX_REQUIRE(p->kernel_variant != -1 || !(shard && shard->comm), "kernel_variant = -1 is a single-GPU parity mode (a sharded sum has no reference order)");
    ref_order_ = p->kernel_variant == -1;
    const bool need_plain = p->schedule == 0 && p->kernel_variant <= 0;
    bool want_scatter = p->kernel_variant == 2 || p->kernel_variant == 3;  // 3: scatter with explicit 32-bit ids
    if (!want_scatter && !need_plain && p->panel_rows == 0 && p->layout_build != 1) {
        const FlatLayoutOptions a = choose_layout(*p, m_, nnz_, n_, kCsrSliceBytes, false), b = choose_layout(*p, n_, nnz_, m_, sizeof(float2), false);
        want_scatter = (a.panel_rows && !a.lds) || (b.panel_rows && !b.lds);
    }
    // (r4) sharded solve in the scatter layout: the column pass runs panel group by panel group, and group j's sums are
    // combined, all-reduced and finalized on a second stream while group j + 1 is streamed -- only the last group's
    // exchange stays exposed.  RCCL's kernels need CUs of their own for that: one block of rcclGenericKernel holds 19.7 KB
    // of LDS and ~280 registers per lane, and neither fits next to a scatter workgroup (150 KB of LDS, 4 x 112 registers per
    // SIMD lane), so the column pass leaves `reserve` CUs free.  MFX_OVERLAP_GROUPS (1 = off) / MFX_COMM_RESERVE_CUS.
    if (shard && shard->comm) {
        // Off by default: two groups cost ~60-85 us per rank-one update on one GPU; whether the half of the
        // all-reduce they hide is longer than that can only be measured on more than one GPU -- bench.py's config5_strong leg
        // does, with 1 and with 2 groups, and keeps the better.
        int groups = 1, reserve = -1;
        if (const char* e = std::getenv("MFX_OVERLAP_GROUPS")) groups = std::atoi(e);
        if (const char* e = std::getenv("MFX_COMM_RESERVE_CUS")) reserve = std::atoi(e);
        overlap_groups_ = (uint32_t) std::max(1, std::min(groups, (int) SegStreamDev::kMaxScatterGroups));
        if (reserve < 0) reserve = overlap_groups_ > 1 ? 16 : 0;
        comm_reserve_cus_ = overlap_groups_ > 1 ? (uint32_t) reserve : 0u;
    }
    int rc = build_stores(R, p, space, want_scatter);
    if (rc != MFX_OK && want_scatter && p->kernel_variant != 2 && p->kernel_variant != 3) {  // e.g. unsorted indices: the ordinary layouts take anything
        csr_ = SegStreamStore(); csc_ = SegStreamStore();
        rc = build_stores(R, p, space, false);
    }
    MFX_TRY(rc);
    if (scatter_) {  // bound of a fixed-point sum: the fullest local index of each store = the OTHER store's longest segment
        for (int side = 0; side < 2; ++side) {
            SegStreamStore& other = side == 0 ? csc_ : csr_;
            std::vector<uint32_t> cnt(other.view.nseg);
            if (!cnt.empty()) MFX_HIP(hipMemcpy(cnt.data(), other.view.seg_cnt, sizeof(uint32_t) * cnt.size(), hipMemcpyDeviceToHost));
            uint32_t mx = 1;
            for (uint32_t c : cnt) mx = std::max(mx, c);
            (side == 0 ? csr_ : csc_).view.scat_max_local_cnt = mx;  // csr_'s local dimension is the columns
        }
    }
    {   // (r4) small matrices (plain layout on both sides, single GPU, no extensions): segment-owner fused passes
        const char* e = std::getenv("MFX_OWNER_PASSES");
        // ... as long as there are few enough segments: one wavefront per segment means (rows + columns) waves per rank, and a wave
        // of a 20-entry row fills 8 % of its lanes.  Measured under graph replay (profiles/r04_exp_small.txt, owner vs flat, ms per
        // outer iteration): 6040 x 3706 0.54 / 0.79, 20 000 x 8 000 0.89 / 1.04, 30 000 x 10 000 (3.9 M ratings) 1.38 / 1.87 -- but
        // 70 000 x 2 000 2.00 / 1.79, 2 000 x 70 000 1.95 / 1.77, 200 000 x 100 000 2.07 / 1.17.  MFX_OWNER_PASSES=1 forces, 0 forbids.
        // ... and no segment so long that its one owner becomes the whole pass (a 256-thread workgroup walks 2048 entries per round)
        const bool few_segments = std::max(m_, n_) <= 40000u && std::max(csc_.view.own_max_len, csr_.view.own_max_len) <= 65536u;
        owner_mode_ = !(e && std::atoi(e) == 0) && (few_segments || (e && std::atoi(e) == 1)) && p->schedule == 1 && p->kernel_variant == 1 && !scatter_ &&
                      !(shard && shard->comm) && csc_.view.own_short && csr_.view.own_short && !(p->do_nmf || p->eps > 0.f || p->rank_trace);
    }
    if (!scatter_) overlap_groups_ = 1;
    overlap_groups_ = std::min(overlap_groups_, csr_.view.scat_ngroups);
    if (overlap_groups_ > 1) {
        MFX_HIP(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking));
        for (uint32_t g = 0; g < overlap_groups_; ++g) MFX_HIP(hipEventCreateWithFlags(&ev_grp_[g], hipEventDisableTiming));
        MFX_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
    if (ref_order_) {  // dispatch order of the reference-order sweeps: longest segment first (ccd_reforder.hip)
        const char* e = std::getenv("MFX_REF_FUSED");
        ref_fused_ = !(e && std::atoi(e) == 0);
        for (int side = 0; side < 2; ++side) {
            const SegStreamDev& v = side == 0 ? csc_.view : csr_.view;
            std::vector<uint32_t> ptr_h((size_t) v.nseg + 1), order;
            MFX_HIP(hipMemcpyAsync(ptr_h.data(), v.ptr, sizeof(uint32_t) * ptr_h.size(), hipMemcpyDeviceToHost, st_));
            MFX_HIP(hipStreamSynchronize(st_));
            (side == 0 ? ref_nlong_csc_ : ref_nlong_csr_) = ref_sweep_order(ptr_h.data(), v.nseg, &order, ref_fused_);
            DevBuf<uint32_t>& dst = side == 0 ? ref_order_csc_ : ref_order_csr_;
            MFX_TRY(dst.alloc(order.size()));
            MFX_TRY(dst.upload(order.data(), order.size(), MFX_HOST, st_));
            MFX_HIP(hipStreamSynchronize(st_));
        }
        if (ref_fused_) {
            // (the owner passes have no sweep / update split to report: launch events only when profiling is asked for)
            prof_.enable(p->profile != 0);
            ref_streams_.main = st_;
            MFX_HIP(hipStreamCreateWithFlags(&ref_streams_.side, hipStreamNonBlocking));
            MFX_HIP(hipEventCreateWithFlags(&ref_streams_.fork, hipEventDisableTiming));
            MFX_HIP(hipEventCreateWithFlags(&ref_streams_.join, hipEventDisableTiming));
            MFX_TRY(ref_zero_cols_.alloc_zero(n_, st_));
        }
    }

    MFX_TRY(W_.alloc_zero((size_t) k_ * m_, st_));
    MFX_TRY(H_.alloc_zero((size_t) k_ * n_, st_));
    MFX_TRY(packA_.alloc_zero(m_, st_));
    MFX_TRY(packB_.alloc_zero(n_, st_));
    MFX_TRY(packC_.alloc_zero(n_, st_));
    MFX_TRY(gh_cols_.alloc_zero((size_t) 2 * n_, st_));
    MFX_TRY(gh_rows_.alloc_zero((size_t) 2 * m_, st_));

    if (shard && shard->comm) {
        MFX_REQUIRE(shard->global_col_nnz, "sharded solve needs global_col_nnz");
        comm_ = shard->comm;
        MFX_TRY(global_col_nnz_.alloc(n_));
        MFX_TRY(global_col_nnz_.upload(shard->global_col_nnz, n_, space, st_));
        global_test_nnz_ = shard->global_test_nnz;
        // No collective in here: a rank whose setup fails must not leave the others inside one.  The
        // ranks meet in mfx_comm_agree() after create (which also takes RCCL's lazy connection setup),
        // and the first iterate() call warms the data-path all-reduce up outside its timed span.
    }
    nnz_test_ = T ? T->nnz : 0;
    if (!comm_) global_test_nnz_ = nnz_test_;
    if (nnz_test_ > 0) {
        MFX_REQUIRE(T->row && T->col && T->val, "null test array");
        MFX_TRY(t_row_.alloc(nnz_test_)); MFX_TRY(t_row_.upload(T->row, nnz_test_, space, st_));
        MFX_TRY(t_col_.alloc(nnz_test_)); MFX_TRY(t_col_.upload(T->col, nnz_test_, space, st_));
        MFX_TRY(t_val_.alloc(nnz_test_)); MFX_TRY(t_val_.upload(T->val, nnz_test_, space, st_));
        MFX_TRY(check_index_range(t_row_.get(), (uint64_t) nnz_test_, m_, "test-set row", st_));
        MFX_TRY(check_index_range(t_col_.get(), (uint64_t) nnz_test_, n_, "test-set column", st_));
    }
    MFX_TRY(rmse_partials_.alloc_zero(kRmseBlocks, st_));
    // opt-in extensions
    MFX_REQUIRE(p->eps >= 0.f && p->eps < 1.f, "eps must be in [0, 1)");
    ext_on_ = p->do_nmf != 0 || p->eps > 0.f || p->rank_trace != 0;
    if (p->eps > 0.f || p->rank_trace) MFX_REQUIRE(!comm_, "eps / rank_trace are not available in a sharded solve");
    MFX_REQUIRE(!ref_order_ || !(p->do_nmf || p->eps > 0.f), "do_nmf / eps are not available with kernel_variant = -1");
    if (p->eps > 0.f) {
        MFX_TRY(fundec_seg_.alloc_zero(std::max(m_, n_), st_));
        MFX_TRY(fundec_sum_.alloc_zero(1, st_));
    }
    if (p->rank_trace) {
        MFX_TRY(old_w_.alloc_zero(m_, st_));
        MFX_TRY(old_h_.alloc_zero(n_, st_));
        MFX_TRY(test_resid_.alloc_zero(nnz_test_ > 0 ? (size_t) nnz_test_ : 1, st_));
        MFX_TRY(r1_sum_.alloc_zero(1, st_));
        for (hipEvent_t& e : ev_rank_) MFX_HIP(hipEventCreate(&e));
    }
    MFX_TRY(rmse_sum_.alloc_zero(1, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    return MFX_OK;
}

// Both orientations, side by side (the serial stretches of one overlap the parallel passes / transfers of
// the other).  Error text is thread-local, so it is carried across.
int CcdSolver::build_stores(const mfx_csx* R, const mfx_params* p, mfx_memspace space, bool scatter) {
    const bool need_plain = p->schedule == 0 && p->kernel_variant <= 0;
    auto options = [&](uint32_t nseg, uint32_t G, uint32_t elem_bytes, bool is_csr) {
        if (!scatter) return choose_layout(*p, nseg, nnz_, G, elem_bytes, need_plain);
        // scatter: 24 B of LDS per local index (operand pair + two 64-bit accumulators) and the whole 160 KB of a CU
        // for one workgroup: 6816 + 1 padding slot = 163 608 B.  Fewer, larger panels = fewer re-reads of the
        // streamed operand, which is what bounds the pass (6144 -> 6816: 10 % fewer line fills).
        constexpr uint32_t kScatterPanel = 6816;
        FlatLayoutOptions o;
        o.scatter = true; o.lds = true; o.spans_per_wg = 16;
        o.scatter_ids32 = p->kernel_variant == 3;
        o.panel_rows = std::min<uint32_t>(p->panel_rows > 0 ? (uint32_t) p->panel_rows : kScatterPanel, std::max<uint32_t>(G, 1u));
        // the row-major store is the one the COLUMN pass streams: it alone is launched by panel groups / on fewer CUs
        const uint32_t groups = is_csr ? overlap_groups_ : 1u;
        o.scatter_groups = groups;
        // (r4) PHASE ALIGNMENT.  The persistent workgroups own equal, contiguous ranges of the panel-major stream, and inside
        // a panel the streamed operand is read in ascending order: workgroup w starts at phase frac(w * P / nwg) of "its"
        // panel and all of them advance at the same rate.  With P = 146.7 or 183.4 panels for 256 workgroups those phases are
        // 256 different ones -- at any moment the chip reads 256 places spread over the whole 10-12 MB operand, nothing of
        // it stays in a 4 MB L2 until the next panel comes by, and every operand line is an L2 MISS (round 3's counters:
        // TCC_MISS = all reads).  With EQUAL panels and P a multiple of nwg / 8 the phase of workgroup w depends on w mod 8
        // only -- and so does its XCD (workgroup b is dispatched to XCD b % 8): every XCD's workgroups walk the operand IN
        // STEP, one window of it is live per L2, and the operand lines of all but the first panel are L2 hits (TCC_MISS
        // 23.1 M -> 12.6 M per u-pass: the streams alone; u-pass 448 -> 354 us, v-pass 358 -> 307 us on the config-5 shard,
        // profiles/r04_exp_scatter_alignment.txt).  Two to four phases per XCD measure within 3 % of one.  Equal nnz per
        // range is kept, so a skewed matrix only blurs the windows.  With panel groups every LAUNCH is aligned on its own:
        // the panels of a group against the workgroups of its launch.  MFX_SCATTER_PANEL_MULT=0: round 3's panel count.
        int dev = 0, cus = 0;
        const char* mult_env = std::getenv("MFX_SCATTER_PANEL_MULT");
        const bool align = !(mult_env && std::atoi(mult_env) == 0);
        if (p->panel_rows == 0 && hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 16) {
            uint32_t nwg = (uint32_t) cus;
            if (is_csr && comm_reserve_cus_ > 0 && comm_reserve_cus_ + 8 <= nwg) { nwg = (nwg - comm_reserve_cus_) / 8 * 8; o.scatter_wgs = nwg; }
            const uint32_t q = nwg / 8, p0 = (G + kScatterPanel - 1) / kScatterPanel;
            auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
            uint32_t P = 0;
            if (align && p0 >= q / 2 * groups) {  // enough panels for the operand to outgrow an L2 window
                for (uint32_t c = (p0 + groups - 1) / groups * groups; c <= 2 * p0 + q * groups; c += groups)
                    if (q / gcd(q, c / groups) <= 2) { P = c; break; }
            }
            if (P == 0 && groups > 1 && p0 >= groups) P = (p0 + groups - 1) / groups * groups;  // equal groups at least
            if (P == 0 && align && p0 > 1) P = p0;  // few panels: the same count, equal sizes (no short last panel)
            if (P) o.panel_rows = (G + P - 1) / P;
        }
        // spans of 4 tiles when the matrix is large: with persistent workgroups (ccd_scatter.hip) the span length only
        // sets the granularity of their chunk ranges and the padding at every panel's end -- 2 ... 14 tiles measure
        // within 3 % of each other on the config-5 shard (profiles/r03_sweep_shard_persistent.txt), 28 is 12 % slower.
        o.tiles_per_span = p->tiles_per_span > 0 ? (uint32_t) p->tiles_per_span : (nnz_ >= (32u << 20) ? 4u : 0u);
        return o;
    };
    scatter_ = scatter;
    int rc_csr = MFX_OK;
    std::string err_csr;
    // (host allocations of several GB can fail: no exception may leave a thread or cross the C ABI;
    // ThreadGang runs the job inline when no thread can be started and joins in its destructor)
    ThreadGang csr_gang;
    csr_gang.run([&] {
        try {
            rc_csr = use_device(device_);
            if (rc_csr == MFX_OK)
                rc_csr = csr_.build(m_, nnz_, n_, R->csr_row_ptr, R->csr_col_idx, R->csr_val, space, options(m_, n_, kCsrSliceBytes, true),
                                    scatter ? 2 : p->layout_build, st_);
        } catch (const std::exception& ex) {
            rc_csr = fail(MFX_ERR_ALLOC, "building the CSR copy failed: %s", ex.what());
        }
        if (rc_csr != MFX_OK) err_csr = last_error();
    });
    int rc_csc;
    try {
        rc_csc = csc_.build(n_, nnz_, m_, R->csc_col_ptr, R->csc_row_idx, R->csc_val, space, options(n_, m_, sizeof(float2), false),
                            scatter ? 2 : p->layout_build, st_);
    } catch (const std::exception& ex) {
        rc_csc = fail(MFX_ERR_ALLOC, "building the CSC copy failed: %s", ex.what());
    }
    csr_gang.wait();
    if (rc_csc != MFX_OK) return rc_csc;
    if (rc_csr != MFX_OK) { last_error() = err_csr; return rc_csr; }
    return MFX_OK;
}

int CcdSolver::set_factors(const float* W, const float* H, mfx_memspace space) {
    MFX_REQUIRE(W, "mfx_ccd_set_factors: W is required");
    MFX_REQUIRE(oiter_ == 0, "factors can only be set before the first iteration");
    // CCD++ starts from H = 0 with the residual equal to R (src/CCD.cpp:55-60, CCD_CUDA.cu:287);
    // a non-zero H would need the residual rebuilt, which the reference never does either.
    MFX_REQUIRE(H == nullptr, "mfx_ccd_set_factors: H must be NULL (CCD++ starts from H = 0)");
    MFX_TRY(use_device(device_));
    MFX_TRY(W_.upload(W, (size_t) k_ * m_, space, st_));
    MFX_HIP(hipMemsetAsync(H_.get(), 0, sizeof(float) * (size_t) k_ * n_, st_));
    MFX_TRY(launch_pack2(m_, nullptr, Wt(0), packA_.get(), st_));
    MFX_TRY(launch_pack2(n_, nullptr, Ht(0), packB_.get(), st_));
    MFX_HIP(hipStreamSynchronize(st_));
    pending_sub_ = -1;
    factors_set_ = true;
    test_resid_valid_ = false;  // rank_trace: the test residual is rebuilt from the new factors
    return MFX_OK;
}

// (g,h) of the column side are partial sums over this shard's rows: all-reduce before dividing.
int CcdSolver::finalize_cols(const FinalizeArgs& base) {
    FinalizeArgs f = base;
    if (comm_) {
        PROF(KernelProfiler::K_COMBINE, launch_combine_dense(csc_.view, gh_cols_.get(), st_));
        PROF(KernelProfiler::K_ALLREDUCE, comm_allreduce_f32(comm_, gh_cols_.get(), (size_t) 2 * n_, st_));
        f.gh_dense = gh_cols_.get();
        f.cnt_override = global_col_nnz_.get();
    }
    PROF(KernelProfiler::K_FINALIZE, launch_finalize(csc_.view, f, st_));
    return MFX_OK;
}

// Slabs of the scatter pass that just ran -> dense (g, h) -> all-reduce over the shards (column side) ->
// the ordinary finalize from a dense buffer.  `cols`: the sums are over columns (they came from csr_).
int CcdSolver::scatter_finalize(bool cols, const FinalizeArgs& base, int group, hipStream_t st) {
    if (!st) st = st_;
    SegStreamStore& src = cols ? csr_ : csc_;  // the store that was streamed; results are per its local dimension
    float* gh = cols ? gh_cols_.get() : gh_rows_.get();
    const SegStreamDev& v = src.view;
    SegStreamDev out_view;  // finalize from a dense buffer only needs the length
    out_view.nseg = cols ? n_ : m_;
    for (uint32_t g = group < 0 ? 0u : (uint32_t) group; g < (group < 0 ? v.scat_ngroups : (uint32_t) group + 1); ++g) {
        const uint32_t lo = v.scat_grp_lo[g], hi = v.scat_grp_lo[g + 1];
        if (hi <= lo) continue;
        FinalizeArgs f = base;
        f.seg_base = lo; f.gh_len = hi - lo;
        f.cnt_override = (cols ? csc_ : csr_).view.seg_cnt;  // |Omega| of the reduced dimension = the OTHER store's segment counts
        if (cols && comm_) {  // the sums leave the GPU in between: slabs -> dense (g, h) -> all-reduce -> finalize from the reduced buffer
            PROFS(KernelProfiler::K_SCAT_COMBINE, st, launch_scatter_combine(v, gh, st, (int) g));
            f.gh_dense = gh + 2 * (size_t) lo;  // the group's block: g of [lo, hi), then h
            PROFS(KernelProfiler::K_ALLREDUCE, st, comm_allreduce_f32(comm_, gh + 2 * (size_t) lo, (size_t) 2 * (hi - lo), st));
            f.cnt_override = global_col_nnz_.get();
        } else {              // (r4) nothing in between: the finalize adds the slabs itself (one launch, the same bits)
            f.slab_src = &v;
        }
        out_view.seg_cnt = f.cnt_override;
        PROFS(KernelProfiler::K_FINALIZE, st, launch_finalize(out_view, f, st));
    }
    return MFX_OK;
}

FinalizeArgs CcdSolver::fin_base() const {
    FinalizeArgs f;
    f.lambda = p_.lambda;
    f.nmf = p_.do_nmf != 0;
    if (p_.eps > 0.f) { f.fundec_seg = fundec_seg_.get(); f.fundec_sum = fundec_sum_.get(); }
    return f;
}

// eps > 0: the function decrease of the inner iteration that just ran (its v- and u-update added it up on the
// device) against eps * the running maximum of this outer iteration -- LIBPMF 1.41's rule, as restated in
// oracle/mf_oracle.cpp (orc_ccdr1_ext).  One host round trip per inner iteration: this mode is opt-in.
int CcdSolver::inner_stop(uint32_t t, int it, bool* stop) {
    *stop = false;
    if (!(p_.eps > 0.f)) return MFX_OK;
    double cur = 0.0;
    MFX_HIP(hipMemcpyAsync(&cur, fundec_sum_.get(), sizeof(double), hipMemcpyDeviceToHost, st_));
    MFX_HIP(hipMemsetAsync(fundec_sum_.get(), 0, sizeof(double), st_));
    MFX_HIP(hipStreamSynchronize(st_));
    if (cur < fundec_max_ * (double) p_.eps) {
        if (it == 1) ++early_stop_;
        *stop = true;
        return MFX_OK;
    }
    if (!(cur_oiter_ == 1 && t == 0 && it == 1)) fundec_max_ = std::max(fundec_max_, cur);
    return MFX_OK;
}

// rank_trace: calrmse_r1 (src/tools.cpp:261-270) after every rank, as the commented block src/CCD.cpp:141-148 would
// print it under verbose && do_predict.  The test residual starts as the test values minus the model's prediction.
int CcdSolver::trace_begin(uint32_t t) {
    if (!test_resid_valid_ && nnz_test_ > 0) {
        MFX_TRY(flush_pending());
        MFX_TRY(launch_test_resid_init(nnz_test_, t_row_.get(), t_col_.get(), t_val_.get(), W_.get(), H_.get(), m_, n_, k_,
                                       test_resid_.get(), st_));
        test_resid_valid_ = true;
    }
    MFX_HIP(hipMemcpyAsync(old_w_.get(), Wt(t), sizeof(float) * m_, hipMemcpyDeviceToDevice, st_));
    MFX_HIP(hipMemcpyAsync(old_h_.get(), Ht(t), sizeof(float) * n_, hipMemcpyDeviceToDevice, st_));
    MFX_HIP(hipEventRecord(ev_rank_[0], st_));
    return MFX_OK;
}

int CcdSolver::trace_end(uint32_t t) {
    MFX_HIP(hipEventRecord(ev_rank_[1], st_));
    double sum = 0.0;
    if (nnz_test_ > 0) {
        MFX_TRY(launch_test_r1(nnz_test_, t_row_.get(), t_col_.get(), test_resid_.get(), Wt(t), Ht(t), old_w_.get(), old_h_.get(),
                               rmse_partials_.get(), kRmseBlocks, r1_sum_.get(), st_));
        MFX_HIP(hipMemcpyAsync(&sum, r1_sum_.get(), sizeof(double), hipMemcpyDeviceToHost, st_));
    }
    MFX_HIP(hipStreamSynchronize(st_));
    float ms = 0.f;
    MFX_HIP(hipEventElapsedTime(&ms, ev_rank_[0], ev_rank_[1]));
    const size_t slot = trace_done_.size() * k_ + t;
    if (trace_rmse_.size() <= slot) { trace_rmse_.resize(slot + 1, std::nan("")); trace_secs_.resize(slot + 1, 0.0); }
    trace_rmse_[slot] = nnz_test_ > 0 ? std::sqrt(sum / (double) nnz_test_) : 0.0;
    trace_secs_[slot] = ms * 1e-3;
    if (p_.verbose) {  // the line of the commented block src/CCD.cpp:141-148
        printf("iter %d rank %d time %f rmse %f\n", (int) cur_oiter_, (int) t + 1, trace_secs_[slot], trace_rmse_[slot]);
        fflush(stdout);
    }
    return MFX_OK;
}

int CcdSolver::rank_trace(int cap, double* rmse, double* seconds, int iters_cap, int32_t* ranks_done) const {
    const size_t n = trace_done_.size() * k_;
    for (size_t i = 0; i < n && (int) i < cap; ++i) {
        if (rmse) rmse[i] = i < trace_rmse_.size() ? trace_rmse_[i] : std::nan("");
        if (seconds) seconds[i] = i < trace_secs_.size() ? trace_secs_[i] : 0.0;
    }
    for (size_t i = 0; i < trace_done_.size() && (int) i < iters_cap; ++i)
        if (ranks_done) ranks_done[i] = trace_done_[i];
    return (int) trace_done_.size();
}

int CcdSolver::rank_fused_scatter(uint32_t t) {
    const uint32_t next = (t + 1) % k_;
    // same invariant as rank_fused: packA = (u_prev_new | 0, W[t] old), packB = (v_prev_new | 0, H[t] old).
    // v-update: stream the ROW-major copy; columns are local (slice packB, accumulators), rows stream (packA)
    FinalizeArgs fv = fin_base();
    fv.lambda = p_.lambda; fv.out_vec = Ht(t); fv.pack2 = packB_.get(); fv.next_vec = Ht(next); fv.pack4 = packC_.get();
    fv.pack4_as3 = true;  // 12-byte triples: a quarter fewer line fills of the streamed operand in the u-pass
    if (overlap_groups_ > 1) {
        // panel group g: pass on st_; its combine -> all-reduce -> finalize on st2_, under the pass of group g + 1.  A group's
        // finalize writes H[t], packB and the triples for ITS columns only; the launches still to come read packB for
        // theirs.  st_ joins st2_ before the u-pass, which needs every column's new value.
        for (uint32_t g = 0; g < overlap_groups_; ++g) {
            PROF(KernelProfiler::K_SCAT_V, launch_scatter(SM_V, csr_.view, packB_.get(), packA_.get(), 0, st_, (int) g));
            MFX_HIP(hipEventRecord(ev_grp_[g], st_));
            MFX_HIP(hipStreamWaitEvent(st2_, ev_grp_[g], 0));
            MFX_TRY(scatter_finalize(true, fv, (int) g, st2_));
        }
        MFX_HIP(hipEventRecord(ev_join_, st2_));
        MFX_HIP(hipStreamWaitEvent(st_, ev_join_, 0));
    } else {
        PROF(KernelProfiler::K_SCAT_V, launch_scatter(SM_V, csr_.view, packB_.get(), packA_.get(), 0, st_));
        MFX_TRY(scatter_finalize(true, fv));
    }
    // u-update: stream the COLUMN-major copy; rows are local (slice packA), columns stream (packC, triples)
    PROF(KernelProfiler::K_SCAT_U, launch_scatter(SM_U, csc_.view, packA_.get(), packC_.get(), 0, st_));
    FinalizeArgs fu = fin_base();
    fu.lambda = p_.lambda; fu.out_vec = Wt(t); fu.pack2 = packA_.get(); fu.next_vec = Wt(next);
    MFX_TRY(scatter_finalize(false, fu));
    bool stop = false;
    MFX_TRY(inner_stop(t, 1, &stop));
    for (int it = 2; it <= p_.maxinneriter && !stop; ++it) {  // remaining inner iterations: read-only sweeps
        MFX_TRY(sweep(csc_, Wt(t), Ht(t), true));
        MFX_TRY(sweep(csr_, Ht(t), Wt(t), false));
        MFX_TRY(inner_stop(t, it, &stop));
    }
    if (p_.maxinneriter > 1) {
        PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(t), Wt(next), packA_.get(), st_));
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(t), Ht(next), packB_.get(), st_));
    }
    pending_sub_ = (int32_t) t;
    return MFX_OK;
}

// (r4) Small matrices: two launches per rank -- every segment has one owner, which finalizes it inside the pass (k_seg_owner).
int CcdSolver::rank_fused_owner(uint32_t t) {
    const uint32_t next = (t + 1) % k_;
    // same invariant as rank_fused: packA = (u_prev_new | 0, W[t] old), packB = (v_prev_new | 0, H[t] old)
    FinalizeArgs fv = fin_base();
    fv.lambda = p_.lambda; fv.out_vec = Ht(t); fv.pack2 = packB_.get(); fv.next_vec = Ht(next); fv.pack4 = packC_.get();
    PROF(KernelProfiler::K_FCSC, launch_seg_owner(FM_FCSC, csc_.view, packA_.get(), packB_.get(), fv, st_));
    FinalizeArgs fu = fin_base();
    fu.lambda = p_.lambda; fu.out_vec = Wt(t); fu.pack2 = packA_.get(); fu.next_vec = Wt(next);
    PROF(KernelProfiler::K_FCSR, launch_seg_owner(FM_FCSR, csr_.view, packC_.get(), packA_.get(), fu, st_));
    for (int it = 2; it <= p_.maxinneriter; ++it) {  // remaining inner iterations: read-only sweeps, finalized by their owners too
        FinalizeArgs f2 = fin_base(); f2.out_vec = Ht(t);
        PROF(KernelProfiler::K_SWEEP, launch_seg_owner(FM_SWEEP, csc_.view, Wt(t), nullptr, f2, st_));
        FinalizeArgs f3 = fin_base(); f3.out_vec = Wt(t);
        PROF(KernelProfiler::K_SWEEP, launch_seg_owner(FM_SWEEP, csr_.view, Ht(t), nullptr, f3, st_));
    }
    if (p_.maxinneriter > 1) {  // the packs must carry the FINAL (u_t, v_t)
        PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(t), Wt(next), packA_.get(), st_));
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(t), Ht(next), packB_.get(), st_));
    }
    pending_sub_ = (int32_t) t;
    return MFX_OK;
}

// (r4) Reference-order mode on the schedule of rank_fused_owner: every sum in the reference's order (launch_ref_owner), the
// element update the reference's own -- two launches per rank, no separate residual pass; bit for bit the as-written sequence.
int CcdSolver::rank_ref_fused(uint32_t t, bool add_back) {
    const uint32_t next = (t + 1) % k_;
    // invariant on entry, as in rank_fused: packA = (u_prev_new | 0, W[t] old), packB = (v_prev_new | 0, H[t] old)
    if (!add_back)  // first outer iteration: the reference does not add rank t back (src/CCD.cpp:100-104; H starts at 0, so the term
                    // is u * 0 anyway) -- the column side's add-back operand reads 0 whatever H holds
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, pending_sub_ >= 0 ? Ht((uint32_t) pending_sub_) : nullptr, ref_zero_cols_.get(), packB_.get(), st_));
    FinalizeArgs fv = fin_base();
    fv.lambda = p_.lambda; fv.out_vec = Ht(t); fv.pack2 = packB_.get(); fv.next_vec = Ht(next); fv.pack4 = packC_.get();
    PROF(KernelProfiler::K_SWEEP_REF, launch_ref_owner(FM_FCSC, csc_.view, ref_order_csc_.get(), ref_nlong_csc_, packA_.get(), packB_.get(), fv, ref_streams_));
    FinalizeArgs fu = fin_base();
    fu.lambda = p_.lambda; fu.out_vec = Wt(t); fu.pack2 = packA_.get(); fu.next_vec = Wt(next);
    PROF(KernelProfiler::K_SWEEP_REF, launch_ref_owner(FM_FCSR, csr_.view, ref_order_csr_.get(), ref_nlong_csr_, packC_.get(), packA_.get(), fu, ref_streams_));
    bool stop = false;
    MFX_TRY(inner_stop(t, 1, &stop));
    for (int it = 2; it <= p_.maxinneriter && !stop; ++it) {  // remaining inner iterations: read-only sweeps
        FinalizeArgs f2 = fin_base(); f2.out_vec = Ht(t);
        PROF(KernelProfiler::K_SWEEP_REF, launch_ref_owner(FM_SWEEP, csc_.view, ref_order_csc_.get(), ref_nlong_csc_, Wt(t), nullptr, f2, ref_streams_));
        FinalizeArgs f3 = fin_base(); f3.out_vec = Wt(t);
        PROF(KernelProfiler::K_SWEEP_REF, launch_ref_owner(FM_SWEEP, csr_.view, ref_order_csr_.get(), ref_nlong_csr_, Ht(t), nullptr, f3, ref_streams_));
        MFX_TRY(inner_stop(t, it, &stop));
    }
    if (p_.maxinneriter > 1) {  // the packs must carry the FINAL (u_t, v_t)
        PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(t), Wt(next), packA_.get(), st_));
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(t), Ht(next), packB_.get(), st_));
    }
    pending_sub_ = (int32_t) t;
    return MFX_OK;
}

int CcdSolver::rank_fused(uint32_t t) {
    if (scatter_) return rank_fused_scatter(t);
    if (owner_mode_) return rank_fused_owner(t);
    const uint32_t next = (t + 1) % k_;
    // invariant on entry: packA = (u_prev_new | 0, W[t] old), packB = (v_prev_new | 0, H[t] old)
    // With MFX_FUSE_FINALIZE=1 the finalize of each pass runs INSIDE the pass (fused_finalize, ccd_kernels.hip);
    // the column side of a sharded solve keeps the separate kernel: its sums go through the all-reduce first.
    // Off by default: bit-identical, but measured SLOWER than pass + k_finalize (see the comment there).
    FinalizeArgs fv = fin_base();
    fv.lambda = p_.lambda; fv.out_vec = Ht(t); fv.pack2 = packB_.get(); fv.next_vec = Ht(next);
    fv.pack4 = packC_.get();
    if (use_fused(csc_) && !comm_) {
        PROF(KernelProfiler::K_FCSC, launch_flat_fused(FM_FCSC, csc_.view, packA_.get(), packB_.get(), fv, st_));
    } else {
        PROF(KernelProfiler::K_FCSC, launch_flat(FM_FCSC, csc_.view, packA_.get(), packB_.get(), 0, st_));
        MFX_TRY(finalize_cols(fv));
    }
    // per-row scalars of the CSR pass are exactly packA (u_prev_new, u_t_old), indexed by row
    FinalizeArgs fu = fin_base();
    fu.lambda = p_.lambda; fu.out_vec = Wt(t); fu.pack2 = packA_.get(); fu.next_vec = Wt(next);
    if (use_fused(csr_)) {
        PROF(KernelProfiler::K_FCSR, launch_flat_fused(FM_FCSR, csr_.view, packC_.get(), packA_.get(), fu, st_));
    } else {
        PROF(KernelProfiler::K_FCSR, launch_flat(FM_FCSR, csr_.view, packC_.get(), packA_.get(), 0, st_));
        PROF(KernelProfiler::K_FINALIZE, launch_finalize(csr_.view, fu, st_));
    }

    bool stop = false;
    MFX_TRY(inner_stop(t, 1, &stop));
    for (int it = 2; it <= p_.maxinneriter && !stop; ++it) {  // remaining inner iterations: read-only sweeps
        FinalizeArgs f2 = fin_base(); f2.out_vec = Ht(t);
        if (use_fused(csc_) && !comm_ && csc_.view.panel_rows == 0) {
            PROF(KernelProfiler::K_SWEEP, launch_flat_fused(FM_SWEEP, csc_.view, Wt(t), nullptr, f2, st_));
        } else {
            PROF(KernelProfiler::K_SWEEP, launch_flat(FM_SWEEP, csc_.view, Wt(t), nullptr, 0, st_));
            MFX_TRY(finalize_cols(f2));
        }
        FinalizeArgs f3 = fin_base(); f3.out_vec = Wt(t);
        if (use_fused(csr_) && csr_.view.panel_rows == 0) {
            PROF(KernelProfiler::K_SWEEP, launch_flat_fused(FM_SWEEP, csr_.view, Ht(t), nullptr, f3, st_));
        } else {
            PROF(KernelProfiler::K_SWEEP, launch_flat(FM_SWEEP, csr_.view, Ht(t), nullptr, 0, st_));
            PROF(KernelProfiler::K_FINALIZE, launch_finalize(csr_.view, f3, st_));
        }
        MFX_TRY(inner_stop(t, it, &stop));
    }
    if (p_.maxinneriter > 1) {  // the packs must carry the FINAL (u_t, v_t)
        PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(t), Wt(next), packA_.get(), st_));
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(t), Ht(next), packB_.get(), st_));
    }
    pending_sub_ = (int32_t) t;
    return MFX_OK;
}

int CcdSolver::flush_pending() {
    if (pending_sub_ < 0) return MFX_OK;
    const uint32_t t = (uint32_t) pending_sub_, next = (t + 1) % k_;
    MFX_TRY(resid(csc_, Wt(t), Ht(t), 0));
    MFX_TRY(resid(csr_, Ht(t), Wt(t), 0));
    if (ref_order_ && !ref_fused_) { pending_sub_ = -1; return MFX_OK; }  // (the operand packs below belong to the fused schedules)
    PROF(KernelProfiler::K_PACK, launch_pack2(m_, nullptr, Wt(next), packA_.get(), st_));
    PROF(KernelProfiler::K_PACK, launch_pack2(n_, nullptr, Ht(next), packB_.get(), st_));
    pending_sub_ = -1;
    return MFX_OK;
}

int CcdSolver::sweep(SegStreamStore& s, const float* vec, float* out, bool is_col_side) {
    FinalizeArgs f = fin_base();
    f.lambda = p_.lambda;
    f.out_vec = out;
    if (scatter_) {  // sums over columns stream the row-major store (vec = u, by row), and vice versa
        SegStreamStore& src = is_col_side ? csr_ : csc_;
        PROF(KernelProfiler::K_SCAT_SWEEP, launch_scatter(SM_SWEEP, src.view, nullptr, vec, 0, st_));
        return scatter_finalize(is_col_side, f);
    }
    if (ref_order_) {  // g, h and the division in one kernel, in the reference's order; no separate finalize
        PROF(KernelProfiler::K_SWEEP_REF, launch_sweep_ref(s.view, (is_col_side ? ref_order_csc_ : ref_order_csr_).get(), is_col_side ? ref_nlong_csc_ : ref_nlong_csr_, vec, p_.lambda, out, st_));
        return MFX_OK;
    }
    if (p_.kernel_variant == 0) {
        float* gh = is_col_side ? gh_cols_.get() : gh_rows_.get();
        PROF(KernelProfiler::K_SWEEP_WAVE, launch_sweep_wave(s.view, vec, gh, gh + s.view.nseg, st_));
        f.gh_dense = gh;
        if (is_col_side && comm_) {
            PROF(KernelProfiler::K_ALLREDUCE, comm_allreduce_f32(comm_, gh, (size_t) 2 * n_, st_));
            f.cnt_override = global_col_nnz_.get();
        }
        PROF(KernelProfiler::K_FINALIZE, launch_finalize(s.view, f, st_));
    } else {
        PROF(KernelProfiler::K_SWEEP, launch_flat(FM_SWEEP, s.view, vec, nullptr, 0, st_));
        if (is_col_side) MFX_TRY(finalize_cols(f));
        else PROF(KernelProfiler::K_FINALIZE, launch_finalize(s.view, f, st_));
    }
    return MFX_OK;
}

int CcdSolver::resid(SegStreamStore& s, const float* gathered, const float* per_seg, int add) {
    if (scatter_) {  // the store's own copy: local index = its gathered dimension (slice), segment id streams
        PROF(KernelProfiler::K_SCAT_RESID, launch_scatter(SM_RESID, s.view, gathered, per_seg, add, st_));
        return MFX_OK;
    }
    if (p_.kernel_variant == 0)  // (the reference-order mode, -1, takes the flat kernel over its plain layout: the update is
                                 // elementwise -- bit-exact in every kernel -- and the flat one is twice as fast)
        PROF(KernelProfiler::K_RESID_WAVE, launch_resid_wave(s.view, gathered, per_seg, add, st_));
    else
        PROF(KernelProfiler::K_RESID, launch_flat(FM_RESID, s.view, gathered, per_seg, add, st_));
    return MFX_OK;
}

// The reference's kernel sequence, one launch per reference kernel (cuda_src/CCD_CUDA.cu:349-377).
int CcdSolver::rank_as_written(uint32_t t, bool add_back) {
    float* u = Wt(t);
    float* v = Ht(t);
    if (ref_order_ && ref_fused_) return rank_ref_fused(t, add_back);
    if (ref_order_) {
        // (r4) reference-order mode: the subtraction of rank t - 1 and the add-back of rank t are two consecutive elementwise passes
        // over the same copy (src/CCD.cpp:100-134) -- applied in ONE pass per copy, in the same order and with the same two roundings
        // per element (element_op<FM_FCSC>: (r - a b) + c d, unfused), so every stored residual keeps the reference's bits while the
        // mode streams each copy once per rank instead of twice.  The subtraction of a rank stays pending until the next rank (or a
        // reader of the residual: flush_pending) applies it.
        if (pending_sub_ >= 0) {
            const uint32_t prev = (uint32_t) pending_sub_;
            if (add_back) {
                PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(prev), u, packA_.get(), st_));
                PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(prev), v, packB_.get(), st_));
                PROF(KernelProfiler::K_RESID, launch_flat(FM_FCSC, csc_.view, packA_.get(), packB_.get(), 0, st_));  // (its sums are not used)
                PROF(KernelProfiler::K_RESID, launch_flat(FM_FCSC, csr_.view, packB_.get(), packA_.get(), 0, st_));
            } else {
                MFX_TRY(resid(csc_, Wt(prev), Ht(prev), 0));
                MFX_TRY(resid(csr_, Ht(prev), Wt(prev), 0));
            }
            pending_sub_ = -1;
        } else if (add_back) {
            MFX_TRY(resid(csc_, u, v, 1));
            MFX_TRY(resid(csr_, v, u, 1));
        }
        bool stop = false;
        for (int it = 1; it <= p_.maxinneriter && !stop; ++it) {
            MFX_TRY(sweep(csc_, u, v, true));
            MFX_TRY(sweep(csr_, v, u, false));
            MFX_TRY(inner_stop(t, it, &stop));
        }
        pending_sub_ = (int32_t) t;
        return MFX_OK;
    }
    if (add_back) {
        MFX_TRY(resid(csc_, u, v, 1));
        MFX_TRY(resid(csr_, v, u, 1));
    }
    bool stop = false;
    for (int it = 1; it <= p_.maxinneriter && !stop; ++it) {
        MFX_TRY(sweep(csc_, u, v, true));   // v <- rank-one over columns, using u
        MFX_TRY(sweep(csr_, v, u, false));  // u <- rank-one over rows, using the new v
        MFX_TRY(inner_stop(t, it, &stop));
    }
    MFX_TRY(resid(csc_, u, v, 0));
    MFX_TRY(resid(csr_, v, u, 0));
    return MFX_OK;
}

int CcdSolver::test_rmse(double* rmse_out) {
    *rmse_out = 0.0;
    if (global_test_nnz_ <= 0) return MFX_OK;
    if (nnz_test_ > 0) {
        PROF(KernelProfiler::K_RMSE,
             launch_test_sqerr(nnz_test_, t_row_.get(), t_col_.get(), t_val_.get(), W_.get(), H_.get(), m_, n_, k_,
                               0, rmse_partials_.get(), kRmseBlocks, rmse_sum_.get(), st_));
    } else {
        MFX_HIP(hipMemsetAsync(rmse_sum_.get(), 0, sizeof(double), st_));
    }
    if (comm_) PROF(KernelProfiler::K_ALLREDUCE, comm_allreduce_f64(comm_, rmse_sum_.get(), 1, st_));
    double sum = 0.0;
    MFX_HIP(hipMemcpyAsync(&sum, rmse_sum_.get(), sizeof(double), hipMemcpyDeviceToHost, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    *rmse_out = std::sqrt(sum / (double) global_test_nnz_);
    return MFX_OK;
}

// Every launch of a fused outer iteration has the same arguments in every iteration (rank t's
// slices and the operand packs), so after one eager iteration the whole k x 4 launch sequence is
// captured once and replayed: one host call per outer iteration instead of 4k.  Matters when the
// kernels are a few microseconds long (ML-100K / ML-1M sized inputs); irrelevant at Netflix size.
int CcdSolver::enqueue_outer_iteration(int64_t oiter) {
    const bool graphable = p_.schedule == 1 && p_.graph >= 0 && !comm_ && !prof_.enabled() && !graph_failed_ && !ext_on_;
    if (graphable && graph_exec_) {
        MFX_HIP(hipGraphLaunch(graph_exec_, st_));
        pending_sub_ = (int32_t) k_ - 1;
        return MFX_OK;
    }
    // first iteration runs eagerly (it also sets the kernels' LDS attributes); capture from the second
    const bool capture = graphable && oiter >= 2;
    if (capture && hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void) hipGetLastError();
        graph_failed_ = true;
        return enqueue_outer_iteration(oiter);
    }
    int rc = MFX_OK;
    cur_oiter_ = oiter;
    fundec_max_ = 0.0;
    early_stop_ = 0;
    uint32_t done = 0;
    for (uint32_t t = 0; t < k_ && rc == MFX_OK; ++t) {
        if (p_.eps > 0.f && early_stop_ >= 5) break;  // LIBPMF: five ranks stopped in their first inner iteration
        if (p_.rank_trace) rc = trace_begin(t);
        if (rc == MFX_OK) rc = p_.schedule == 0 ? rank_as_written(t, oiter > 1) : rank_fused(t);
        if (rc == MFX_OK && p_.rank_trace) rc = trace_end(t);
        ++done;
    }
    if (rc == MFX_OK && done < k_ && p_.schedule == 1 && pending_sub_ >= 0) {
        // the next outer iteration starts at rank 0, not at the rank the packs were primed for
        const uint32_t last = (uint32_t) pending_sub_;
        PROF(KernelProfiler::K_PACK, launch_pack2(m_, Wt(last), Wt(0), packA_.get(), st_));
        PROF(KernelProfiler::K_PACK, launch_pack2(n_, Ht(last), Ht(0), packB_.get(), st_));
    }
    if (ext_on_) {
        trace_done_.push_back((int32_t) done);
        if (p_.rank_trace) { trace_rmse_.resize(trace_done_.size() * k_, std::nan("")); trace_secs_.resize(trace_done_.size() * k_, 0.0); }
    }
    if (!capture) return rc;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st_, &g);
    if (rc != MFX_OK || e != hipSuccess || !g || hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0) != hipSuccess) {
        (void) hipGetLastError();
        if (g) (void) hipGraphDestroy(g);
        graph_exec_ = nullptr;
        graph_failed_ = true;  // fall back to eager launches for good
        return enqueue_outer_iteration(oiter);
    }
    graph_ = g;
    MFX_HIP(hipGraphLaunch(graph_exec_, st_));  // the capture recorded the work without running it
    return MFX_OK;
}

int CcdSolver::iterate(int n_outer, int with_rmse, mfx_iter_report* reports) {
    MFX_REQUIRE(n_outer >= 0, "n_outer must be >= 0");
    MFX_REQUIRE(factors_set_, "mfx_ccd_iterate: call mfx_ccd_set_factors first");
    MFX_TRY(use_device(device_));
    trace_rmse_.clear(); trace_secs_.clear(); trace_done_.clear();
    if (comm_ && !comm_warm_ && n_outer > 0) {
        // RCCL builds its rings / connections lazily, inside the first collective of a given size class:
        // take that hit before the first timed iteration, with an all-reduce of the (still zero) column
        // buffer.  Every rank reaches this point with the same n_outer, so the collective is matched.
        MFX_TRY(comm_allreduce_f32(comm_, gh_cols_.get(), (size_t) 2 * n_, st_));
        if (overlap_groups_ > 1) {  // ... and the panel-group sized ones on the stream they will run on
            const SegStreamDev& v = csr_.view;
            MFX_HIP(hipStreamSynchronize(st_));
            for (uint32_t g = 0; g < v.scat_ngroups; ++g)
                if (v.scat_grp_lo[g + 1] > v.scat_grp_lo[g])
                    MFX_TRY(comm_allreduce_f32(comm_, gh_cols_.get() + 2 * (size_t) v.scat_grp_lo[g], (size_t) 2 * (v.scat_grp_lo[g + 1] - v.scat_grp_lo[g]), st2_));
            MFX_HIP(hipStreamSynchronize(st2_));
        }
        MFX_HIP(hipStreamSynchronize(st_));
        comm_warm_ = true;
    }
    for (int it = 0; it < n_outer; ++it) {
        const int64_t oiter = oiter_ + 1;
        double before[KernelProfiler::K_COUNT];
        for (int i = 0; i < KernelProfiler::K_COUNT; ++i) before[i] = prof_.seconds[i];
        MFX_HIP(hipEventRecord(ev_[0], st_));
        const auto host_t0 = std::chrono::steady_clock::now();
        MFX_TRY(enqueue_outer_iteration(oiter));
        if (prof_.enabled()) {  // host time to ENQUEUE the iteration's launches (and collectives): hidden as long as it stays below the GPU time
            prof_.seconds[KernelProfiler::K_HOST_ENQUEUE] += std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count();
            prof_.launches[KernelProfiler::K_HOST_ENQUEUE] += 1;
        }
        MFX_HIP(hipEventRecord(ev_[1], st_));
        double rmse = 0.0;
        MFX_HIP(hipEventRecord(ev_[2], st_));
        if (with_rmse) MFX_TRY(test_rmse(&rmse));
        MFX_HIP(hipEventRecord(ev_[3], st_));
        MFX_HIP(hipStreamSynchronize(st_));
        MFX_TRY(prof_.collect());
        float ms_iter = 0.f, ms_rmse = 0.f;
        MFX_HIP(hipEventElapsedTime(&ms_iter, ev_[0], ev_[1]));
        MFX_HIP(hipEventElapsedTime(&ms_rmse, ev_[2], ev_[3]));
        mfx_iter_report rep;
        if (p_.schedule == 0) {  // the reference's split: sweeps vs residual updates
            auto d = [&](int id) { return prof_.seconds[id] - before[id]; };
            rep.update_time = d(KernelProfiler::K_RESID) + d(KernelProfiler::K_RESID_WAVE);
            rep.rank_time = ms_iter * 1e-3 - rep.update_time;
        } else {  // fused passes do both at once: everything is booked as rank_time
            rep.rank_time = ms_iter * 1e-3;
            rep.update_time = 0.0;
        }
        rep.rmse = rmse;
        rep.rmse_time = ms_rmse * 1e-3;
        rank_acc_ += rep.rank_time;
        update_acc_ += rep.update_time;
        oiter_ = oiter;
        if (reports) reports[it] = rep;
        if (p_.verbose && (!comm_ || comm_->rank == 0)) {
            // log line format of cuda_src/CCD_CUDA.cu:405-406
            printf("[-INFO-] iteration num %d \trank_time %.4lf|%.4lf s \tupdate_time %.4lf|%.4lfs \tRMSE=%lf time:%fs\n",
                   (int) oiter, rep.rank_time, rank_acc_, rep.update_time, update_acc_, rep.rmse, rep.rmse_time);
            fflush(stdout);
        }
    }
    if (!p_.profile) prof_.reset_totals();
    return MFX_OK;
}

int CcdSolver::set_profile(bool on) {
    // (schedule 0 keeps its launch events on regardless: its rank/update split is derived from them)
    p_.profile = on ? 1 : 0;
    prof_.enable(on || (p_.schedule == 0 && !ref_fused_));
    prof_.reset_totals();
    return MFX_OK;
}

int CcdSolver::get_factors(float* W, float* H, mfx_memspace space) {
    MFX_TRY(use_device(device_));
    const hipMemcpyKind kind = space == MFX_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (W) MFX_HIP(hipMemcpyAsync(W, W_.get(), sizeof(float) * (size_t) k_ * m_, kind, st_));
    if (H) MFX_HIP(hipMemcpyAsync(H, H_.get(), sizeof(float) * (size_t) k_ * n_, kind, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    return MFX_OK;
}

int CcdSolver::get_residual(float* csc_val, float* csr_val) {
    MFX_TRY(use_device(device_));
    MFX_TRY(flush_pending());
    DevBuf<float> tmp;  // residuals are stored panel-major: put them back in input order
    MFX_TRY(tmp.alloc(nnz_ ? nnz_ : 1));
    for (int side = 0; side < 2; ++side) {
        float* out = side == 0 ? csc_val : csr_val;
        if (!out || !nnz_) continue;
        MFX_TRY((side == 0 ? csc_ : csr_).unpermute(tmp.get(), st_));
        MFX_HIP(hipMemcpyAsync(out, tmp.get(), sizeof(float) * nnz_, hipMemcpyDeviceToHost, st_));
        MFX_HIP(hipStreamSynchronize(st_));
    }
    MFX_HIP(hipStreamSynchronize(st_));  // (an empty matrix copies nothing above, but flush_pending may have recorded launch events)
    MFX_TRY(prof_.collect());
    return MFX_OK;
}

}  // namespace mfx

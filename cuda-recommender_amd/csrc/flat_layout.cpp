#include "flat_layout.hpp"

#include <algorithm>

namespace mfx {

static uint32_t pick_tiles_per_span(uint64_t nnz) {
    // Aim for ~48k spans (256 CUs x 32 resident waves x ~6 rounds) but keep a span between
    // 2 and 16 tiles: shorter spans waste the per-span prologue, longer ones leave the tail of
    // the grid unbalanced and lengthen nothing useful.
    const uint64_t target_spans = 49152;
    uint64_t t = (nnz / kTileElems + target_spans - 1) / target_spans;
    t = (t + 1) & ~uint64_t(1);  // the kernel consumes tiles in pairs
    return (uint32_t) std::min<uint64_t>(16, std::max<uint64_t>(2, t));
}

void build_flat_layout(const uint32_t* ptr, uint32_t nseg, uint64_t nnz, uint32_t tiles_per_span,
                       FlatLayoutHost* out) {
    FlatLayoutHost& L = *out;
    L = FlatLayoutHost();
    L.nseg = nseg;
    L.nnz = nnz;
    if (tiles_per_span == 0) tiles_per_span = pick_tiles_per_span(nnz);
    if (tiles_per_span & 1) ++tiles_per_span;
    L.tiles_per_span = tiles_per_span;
    const uint64_t span = (uint64_t) tiles_per_span * kTileElems;
    L.nspans = (uint32_t) std::max<uint64_t>(1, (nnz + span - 1) / span);
    L.padded_nnz = (uint64_t) L.nspans * span;
    L.flags.assign(L.padded_nnz / 64, 0);
    L.rank_of_seg.assign(nseg, -1);
    L.seg_of_rank.clear();
    L.seg_of_rank.reserve(nseg);
    for (uint32_t c = 0; c < nseg; ++c) {
        if (ptr[c + 1] > ptr[c]) {
            L.rank_of_seg[c] = (int32_t) L.seg_of_rank.size();
            L.seg_of_rank.push_back(c);
            const uint64_t head = ptr[c];
            L.flags[head >> 6] |= uint64_t(1) << (head & 63);
        }
    }
    L.nne = (uint32_t) L.seg_of_rank.size();
    L.span_rank_base.assign(L.nspans, 0);
    uint32_t r = 0;
    for (uint32_t s = 0; s < L.nspans; ++s) {
        const uint64_t start = (uint64_t) s * span;
        while (r < L.nne && ptr[L.seg_of_rank[r]] < start) ++r;
        L.span_rank_base[s] = r;  // heads strictly before the span's first element
    }
}

}  // namespace mfx

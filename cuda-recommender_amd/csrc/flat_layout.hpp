// flat_layout.hpp -- static metadata that lets one kernel stream a CSR/CSC orientation as a flat,
// nnz-balanced array while still reducing per segment (row or column).
//
// The sparsity pattern never changes during a solve, so everything that depends only on `ptr`
// is computed once, on the host, at solver creation:
//
//   span            a fixed run of L = 256 * tiles_per_span consecutive non-zeros, owned by one
//                   wavefront (64 lanes x 4 consecutive non-zeros per tile, 16-byte loads)
//   head flag       1 bit per non-zero: "first entry of its segment" (0.125 B/nnz of extra reads,
//                   against 4 B/nnz for an explicit per-nnz segment id)
//   rank            index of a segment among the NON-EMPTY segments; element -> rank is a popcount
//   span_rank_base  number of heads before the span's first element
//
// Reduction contract (kernels in ccd_kernels.hip): the span that contains a segment's head stores
// that segment's partial sum to part[rank] (exactly one writer, plain store); a span whose first
// element is NOT a head stores the sum of its leading run to carry[span].  finalize adds, in span
// order, the carries of spans s with  ptr[c]/L < s <= (ptr[c+1]-1)/L  -- deterministic, no atomics.
#pragma once

#include <cstdint>
#include <vector>

namespace mfx {

constexpr uint32_t kTileElems = 256;  // 64 lanes x 4 elements

struct FlatLayoutHost {
    uint32_t nseg = 0;            // segments (columns for CSC, rows for CSR)
    uint32_t nne = 0;             // non-empty segments
    uint32_t nspans = 0;          // wave spans
    uint32_t tiles_per_span = 0;  // span length / 256
    uint64_t nnz = 0;
    uint64_t padded_nnz = 0;      // nspans * span length (idx/val/flags are allocated to this)
    std::vector<uint64_t> flags;           // [padded_nnz / 64]
    std::vector<int32_t> rank_of_seg;      // [nseg], -1 for an empty segment
    std::vector<uint32_t> seg_of_rank;     // [nne]
    std::vector<uint32_t> span_rank_base;  // [nspans]
    uint32_t span_len() const { return tiles_per_span * kTileElems; }
};

// tiles_per_span = 0 picks one from nnz (enough spans to fill 256 CUs several times over).
void build_flat_layout(const uint32_t* ptr, uint32_t nseg, uint64_t nnz, uint32_t tiles_per_span,
                       FlatLayoutHost* out);

}  // namespace mfx

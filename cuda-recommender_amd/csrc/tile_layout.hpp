// tile_layout.hpp -- 2-D tile order for hyper-sparse orientations (static, built once on the host).
//
// When a (gather panel, segment) pair holds less than one entry on average -- config 5's shards:
// 1.25 M x 1 M with 125 M ratings -- the panel layouts of flat_layout.hpp either shred the segments
// (LDS panels: one partial sum per pair, more pairs than ratings) or keep the gather in L2 (cache
// panels: bound by the texture path's ~145 G gathers/s).  The tile order keeps BOTH operands in LDS:
//
//   segment block b   QB consecutive segments.  The workgroup that owns a block keeps their per-
//                     segment operands AND their (g, h) accumulators in LDS for the whole pass, so a
//                     segment's partial sums never leave the CU until the pass is over.
//   gather panel p    SR consecutive gathered indices; the workgroup walks the panels of its strip
//                     in order, staging one slice of the operand pack at a time (double-buffered).
//   tile (b, p)       the entries of block b whose gathered index lies in panel p, ordered by
//                     segment, then input order; stored as sub-tiles of kSubTile slots.  A slot is
//                     (segment - b*QB) << 16 | (index - p*SR) plus the value.  A run (the entries of
//                     one segment inside one tile) never crosses a sub-tile boundary: the builder
//                     pads, so a wavefront reduces whole runs and every (tile, segment) pair has
//                     exactly one writer -- plain LDS read-modify-write, no atomics, fixed order.
//   strip r           the panels are cut into R strips; workgroup (b, r) walks strip r of block b and
//                     leaves its accumulators in gh_part[r][segment]; the finalize adds the R
//                     strips in order.  R only exists to fill the chip (blocks x R workgroups).
//
// Padding slots carry segment QB and index SR: both are zero slots in LDS, so they contribute exact
// zeros.  Stream order: block-major, panel-minor -- every workgroup reads one contiguous range.
#pragma once

#include <cstdint>
#include <vector>

namespace mfx {

constexpr uint32_t kSubTile = 128;  // slots per sub-tile: 64 lanes x 2 slots

struct TileLayoutHost {
    uint32_t nseg = 0, gather_len = 0;
    uint32_t QB = 0, SR = 0, nB = 0, nP = 0;
    uint64_t nnz = 0, padded = 0;          // real entries / stored slots (multiple of kSubTile)
    std::vector<uint32_t> tile_sub;        // [nB*nP + 1] first sub-tile of tile b*nP + p
    std::vector<uint32_t> code;            // [padded] (segment_local << 16) | index_local
    std::vector<uint32_t> perm;            // [padded] input position, ~0u for padding
    std::vector<uint32_t> seg_cnt;         // [nseg]
    uint32_t pad_code() const { return (QB << 16) | SR; }
};

// ptr/idx: input orientation (host).  Returns false -- and leaves *out unspecified -- when the tile
// order does not suit the pattern: some run is longer than a sub-tile, or alignment padding would
// exceed `max_pad_frac` of the stored slots.  QB, SR <= 65535.
bool build_tile_layout(const uint32_t* ptr, const uint32_t* idx, uint32_t nseg, uint64_t nnz, uint32_t G,
                       uint32_t QB, uint32_t SR, double max_pad_frac, TileLayoutHost* out);

}  // namespace mfx

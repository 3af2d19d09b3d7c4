// ccd_kernels.hpp -- launchers for the CCD++ device kernels (gfx950, wave64).
//
// Two families implement the reference's three CCD kernels (cuda_src/CCD_CUDA.cu:24-104):
//   wave-per-segment  (variant 0)  one wavefront walks one row/column; simple, used as the
//                                  in-library cross-check and for the as-written schedule
//   flat-stream       (variant 1)  nnz-balanced spans over the flat idx/val arrays with head-flag
//                                  segmented reduction (flat_layout.hpp); also carries the two
//                                  FUSED passes that the default schedule is built from
#pragma once

#include "common.hpp"

namespace mfx {

// Device view of one orientation of the residual matrix (CSC: segments = columns, gathered
// index = row; CSR: the transpose) together with its flat-stream metadata.
struct SegStreamDev {
    uint32_t nseg = 0;         // real segments
    uint32_t nne = 0;          // non-empty virtual segments
    uint64_t nnz = 0;          // real non-zeros
    uint64_t padded_nnz = 0;   // stored elements
    uint32_t nspans = 0;
    uint32_t tiles_per_span = 0;
    uint32_t npanels = 1;
    uint32_t panel_rows = 0;   // 0: plain layout, gather from global memory
    bool lds_panels = false;   // panel_rows != 0: slices staged in LDS (16-bit local indices) or cache panels (global gather)
    uint32_t spans_per_wg = 1;
    uint32_t gather_len = 0;
    const uint32_t* ptr = nullptr;             // [nseg+1] input-order pointers (plain layout only)
    const uint32_t* ptr_v = nullptr;           // [npanels*nseg+1] virtual-segment pointers (stored coords)
    const uint32_t* seg_cnt = nullptr;         // [nseg]
    const uint32_t* idx = nullptr;             // [padded nnz] gathered index (plain layout, cache panels)
    const uint16_t* idx16 = nullptr;           // [padded nnz] panel-local gathered index (LDS panels)
    float* val = nullptr;                      // [padded nnz] residual copy, updated in place
    const uint32_t* flags32 = nullptr;         // [padded nnz / 32 + 16] head bits
    const uint32_t* hpre = nullptr;            // [padded nnz / 32 + 16] heads before each word
    uint32_t max_wg_ranks = 0;                 // most ranks any workgroup chunk touches
    const uint32_t* rank_code = nullptr;       // [npanels*nseg] 0xFFFFFFFF = empty, else rank | (bit 31: runs into later spans)
    const uint32_t* seg_of_rank = nullptr;     // [nne] real segment id
    const uint32_t* wg_panel = nullptr;        // [nspans/spans_per_wg]
    const uint32_t* perm = nullptr;            // [padded nnz] input position, ~0u for padding; nullptr: SegStreamStore::unpermute knows
    // scatter layout (ccd_scatter.hip; hyper-sparse orientations): the panel-major order of the LDS-panel
    // layout, read the other way round -- idx16 is the local index of the REDUCED dimension, segid the id of
    // the streamed one; no head flags / ranks / partials
    bool scatter = false;
    const uint32_t* segid = nullptr;           // [padded nnz] segment of every stored element (pad: 0); nullptr when the byte steps below fit
    const uint8_t* seg_delta = nullptr;        // [padded nnz] step from the previous entry of the tile's sorted order
    const uint32_t* tile_base = nullptr;       // [padded nnz / 256] segment of a tile's first sorted entry
    unsigned long long* wgacc = nullptr;       // [slabs][2 * panel_rows] fixed-point (g, h) slabs
    uint32_t scat_nwg = 0;                     // persistent workgroups of the scatter pass (one per CU, or fewer chunks): of the widest launch
    const uint32_t* scat_chunk_lo = nullptr;   // [sum over groups (nwg + 1)] chunk range of every persistent workgroup, group after group
    const uint32_t* scat_slab0 = nullptr;      // [sum over groups nwg] first slab of a workgroup (it writes one per panel it visits)
    // (r4) PANEL GROUPS: the pass can be launched group by group (consecutive panels each), so that the sums of group j
    // are final -- and can be combined, all-reduced and finalized on a second stream -- while group j + 1 is streamed.
    // One group (the whole store) unless the solver overlaps a collective with the pass (sharded column side).
    static constexpr uint32_t kMaxScatterGroups = 16;
    uint32_t scat_ngroups = 1;
    uint32_t scat_grp_nwg[kMaxScatterGroups] = {};       // workgroups of group j's launch
    uint32_t scat_grp_tab[kMaxScatterGroups] = {};       // offset of its chunk ranges in scat_chunk_lo
    uint32_t scat_grp_wg0[kMaxScatterGroups] = {};       // offset of its workgroups in scat_slab0
    uint32_t scat_grp_lo[kMaxScatterGroups + 1] = {};    // first local index (of the gathered dimension) of group j; [ngroups] = gather_len
    const uint32_t* slab_lo = nullptr;         // [npanels + 1] first slab of every panel (a panel's slabs are consecutive)
    uint32_t* scat_slab_bad = nullptr;         // [slabs] a term of the slab was not representable in the fixed-point sums
    uint32_t scat_max_local_cnt = 0;           // most stored entries any index of the gathered (local) dimension has: bounds a fixed-point sum (0: unknown, treated as 1)
    // fused finalize (LDS panels, 16-span workgroups): see fused_finalize in ccd_kernels.hip; nullptr = not available
    const uint32_t* fz_order = nullptr;        // [workgroups] dispatch slot -> chunk
    const uint32_t* fz_g0 = nullptr;           // [workgroups] first / last segment group of a chunk
    const uint32_t* fz_g1 = nullptr;
    const uint32_t* fz_expected = nullptr;     // [groups]
    uint32_t* fz_arrived = nullptr;            // [groups], zero between launches
    const uint32_t* fz_orphans = nullptr;      // [fz_norphans]
    uint32_t fz_norphans = 0, fz_ngroups = 0, fz_max_chunk_groups = 0;
    // (r4) segment-owner fused passes of small matrices (plain layout; k_seg_owner in ccd_kernels.hip): nullptr = not available
    const uint32_t* own_long = nullptr;        // [own_nlong][4] (segment, first entry, end, 0) of the segments of >= seg_owner_long_threshold() entries: one workgroup each
    const uint32_t* own_short = nullptr;       // [own_nshort][4] all other segments, longest first: one wavefront each
    uint32_t own_nlong = 0, own_nshort = 0;
    uint32_t own_max_len = 0;                  // the longest segment (the owner form gives balance up: the solver declines it beyond 65536 entries)
    // reduction scratch written by the flat kernels
    float2* part = nullptr;    // [nne] (g, h) per non-empty virtual segment
    float2* carry = nullptr;   // [nspans] (g, h) of a span's leading run
};

enum FlatMode : int {
    FM_SWEEP = 0,  // g += x*val, h += x*x                      x = gather_f32[idx]
    FM_RESID = 1,  // val (+/-)= gather_f32[idx] * perseg_f32[seg]
    FM_FCSC = 2,   // float2 A = gather[idx], float2 B = perseg[seg]:
                   //   val = (val - A.x*B.x) + A.y*B.y ; g += A.y*val ; h += A.y^2
    FM_FCSR = 3,   // float4 C = gather[idx], float2 D = perseg[seg]:
                   //   val = (val - C.x*D.x) + C.y*D.y ; g += C.z*val ; h += C.z^2
};

// Flat-stream pass over `s`.  gather/perseg element types depend on `mode` (see FlatMode).
int launch_flat(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg,
                int add, hipStream_t st);

// Scatter passes (ccd_scatter.hip).  `s` is the store of the orientation being STREAMED; results are per
// index of its gathered ("local") dimension.  slice_src: operands of the local dimension; global_op:
// operands of the streamed dimension (indexed by segment id).
enum ScatterMode : int {
    SM_V = 0,      // slice float2 B, streamed float2 A: val = (val - A.x*B.x) + A.y*B.y ; g += A.y*val ; h += A.y^2
    SM_U = 1,      // slice float2 D, streamed float4 C: val = (val - C.x*D.x) + C.y*D.y ; g += C.z*val ; h += C.z^2
    SM_SWEEP = 2,  // streamed float x: g += x*val ; h += x*x
    SM_RESID = 3,  // slice float y, streamed float x: val (+/-)= y*x
};
// group: panel group to launch (SegStreamDev::scat_ngroups), -1 = all of them, one launch after the other
int launch_scatter(ScatterMode mode, const SegStreamDev& s, const void* slice_src, const void* global_op, int add, hipStream_t st, int group = -1);
// slabs of a scatter pass -> dense (g, h) over the local dimension, GROUP-MAJOR: group j with local indices [lo, hi)
// owns gh[2 lo .. 2 hi): g of its indices first, then h -- one contiguous all-reduce buffer per group; with one group
// that is gh[0..G) = g, gh[G..2G) = h (G = s.gather_len)
int launch_scatter_combine(const SegStreamDev& s, float* gh, hipStream_t st, int group = -1);

// Wave-per-segment kernels (plain layout only: they walk the input-order arrays).
int launch_sweep_wave(const SegStreamDev& s, const float* vec, float* g_dense, float* h_dense,
                      hipStream_t st);
int launch_resid_wave(const SegStreamDev& s, const float* gathered, const float* per_seg, int add,
                      hipStream_t st);

// Reference-order sweep (ccd_reforder.hip; plain layout only): out[c] = g / h with g and h accumulated strictly left to
// right in unfused fp32 from (0, lambda * count) -- RankOneUpdate_Original_float (src/CCD.cpp:6-16) bit for bit.
// order: dispatch order of the segments (ref_sweep_order: longest first) or nullptr (ascending).
// nlong: the leading `order` entries that take the two-wave plain-add form (ref_sweep_order's return value; 0 with order == nullptr)
int launch_sweep_ref(const SegStreamDev& s, const uint32_t* order, uint32_t nlong, const float* vec, float lambda, float* out, hipStream_t st);
// owner_form: the count of long segments for launch_ref_owner (>= 4096 entries) instead of launch_sweep_ref's (>= 32768)
uint32_t ref_sweep_order(const uint32_t* ptr_host, uint32_t nseg, std::vector<uint32_t>* order, bool owner_form = false);

// partials + carries of a flat pass -> dense gh[0..nseg) = g, gh[nseg..2nseg) = h (0 for empty).
int launch_combine_dense(const SegStreamDev& s, float* gh, hipStream_t st);

// Finalize:  x = cnt ? g / (lambda*cnt + h) : 0  for every segment, plus the operand packs the
// next fused passes read.  Source of (g, h): the flat partials of `s` (gh_dense == nullptr) or a
// dense, already all-reduced buffer gh_dense[2*nseg].
struct FinalizeArgs {
    const float* gh_dense = nullptr;    // [2*nseg] or nullptr
    // dense source restricted to the segments [seg_base, seg_base + gh_len) (a panel group of the scatter pass): gh_dense
    // then points at that group's block -- gh_len sums g, then gh_len sums h; gh_len = 0 means all nseg segments
    uint32_t seg_base = 0, gh_len = 0;
    // (r4) ... or the slabs of the scatter pass that just streamed `slab_src` (its gathered dimension = these segments): combine and
    // finalize in one launch, bit-identical to launch_scatter_combine + the dense form (gh_dense must be nullptr)
    const struct SegStreamDev* slab_src = nullptr;
    const uint32_t* cnt_override = nullptr;  // global |Omega_c| (multi-GPU) or nullptr = local count
    float lambda = 0.f;
    float* out_vec = nullptr;           // W[t] / H[t] slice, [nseg]
    float2* pack2 = nullptr;            // in: (prev_new, cur_old); out: (cur_new, next_old)
    const float* next_vec = nullptr;    // W[t+1] / H[t+1] slice (old values)
    float4* pack4 = nullptr;            // out: (prev_new, cur_old, cur_new, 0), may be nullptr
    // Opt-in extensions (mfx_params.do_nmf / eps; the reference parses these flags and ignores them):
    bool nmf = false;                   // clamp the new value at 0
    double* fundec_seg = nullptr;       // [nseg] scratch: per-segment function decrease h (old - new)^2 (clamped: -2 g old + h old^2)
    double* fundec_sum = nullptr;       // *fundec_sum += sum of fundec_seg, in a fixed order (set together with fundec_seg)
    bool pack4_as3 = false;             // ... stored as 12-byte triples in the same buffer (scatter u-pass: the streamed operand's line fills are what bounds it)
};
int launch_finalize(const SegStreamDev& s, const FinalizeArgs& a, hipStream_t st);
// (r4) Reference-order OWNER pass (ccd_reforder.hip): launch_seg_owner's contract -- pass + division + operand packs in one launch, FM_FCSC /
// FM_FCSR (perseg must be f.pack2) or the read-only FM_SWEEP -- with every sum in the reference's order and the division g / h from
// h = lambda * count + ... (src/CCD.cpp:6-16): bit for bit the reference's values.  order / nlong: ref_sweep_order's.  Items holding a
// long segment run as a second kernel on rs.side when it is set (joined into rs.main before returning), else behind each other on rs.main.
struct RefStreams {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
int launch_ref_owner(FlatMode mode, const SegStreamDev& s, const uint32_t* order, uint32_t nlong, const void* gather, const void* perseg,
                     const FinalizeArgs& f, const RefStreams& rs);

// calrmse_r1 (src/tools.cpp:261-270): resid[q] -= Wt[row] * Ht[col] - oldWt[row] * oldHt[col]; *sum_out = sum resid^2
// resid[q] = val[q] - sum_t W[t][row] * H[t][col] (fp32, rank order): where a rank trace starts from
int launch_test_resid_init(int64_t nnz_test, const uint32_t* row, const uint32_t* col, const float* val, const float* W, const float* H,
                           int64_t rows, int64_t cols, int64_t k, float* resid, hipStream_t st);
int launch_test_r1(int64_t nnz_test, const uint32_t* row, const uint32_t* col, float* resid, const float* Wt, const float* Ht,
                   const float* oldWt, const float* oldHt, double* block_partials, uint32_t nblocks, double* sum_out, hipStream_t st);
// segments per group of the fused finalize = 1024 / panel lanes of the finalize (a function of the panel count)
uint32_t fused_group_size(uint32_t npanels, uint32_t block);
// FM_FCSC / FM_FCSR pass + the finalize of its sums inside the same launch (s.fz_* must be set; f.gh_dense and
// f.cnt_override are not supported: sharded column sums go through the all-reduce and the separate kernel)
int launch_flat_fused(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, const FinalizeArgs& f, hipStream_t st);
// (r4) pass + finalize of one rank-one half-step in ONE launch, every segment owned by one wavefront / workgroup (small matrices,
// plain layout): FM_FCSC / FM_FCSR (perseg must be f.pack2) or the read-only FM_SWEEP (perseg nullptr, only f.out_vec is written)
int launch_seg_owner(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, const FinalizeArgs& f, hipStream_t st);
uint32_t seg_owner_long_threshold();
// first / last segment of every workgroup chunk's real entries (0xFFFFFFFF / 0 for a chunk of padding only)
// wg_spans: spans per workgroup chunk (16 for LDS panels, 4 for the plain layout's 256-thread workgroups)
int launch_chunk_seg_range(const SegStreamDev& s, uint32_t wg_spans, const uint32_t* panel_end, uint32_t* seg_first, uint32_t* seg_last, hipStream_t st);

// out[perm[e]] = val[e] for every stored, non-padding element: residual back in input order.
int launch_unpermute(const SegStreamDev& s, float* out, hipStream_t st);

// Same for a layout without a perm array (flat_layout.hpp, perm_is_runs).
int launch_unpermute_runs(const SegStreamDev& s, const uint32_t* first_q, const uint32_t* panel_end, float* out, hipStream_t st);

// pack[i] = (x ? x[i] : 0, y[i])
int launch_pack2(uint32_t n, const float* x, const float* y, float2* pack, hipStream_t st);

// Test RMSE: sum over the test set of (pred - val)^2 in fp64 -> *sum_out (device double).
// reference: GPU_rmse (cuda_src/CUDA_AUX.cu:3-27) + host sum (CCD_CUDA.cu:394-401);
// arithmetic follows calrmse/dot (src/tools.cpp:184-198,235-248): fp32 products, fp64 sums.
int launch_test_sqerr(int64_t nnz_test, const uint32_t* row, const uint32_t* col, const float* val,
                      const float* W, const float* H, int64_t rows, int64_t cols, int64_t k,
                      int ifALS, double* block_partials, uint32_t nblocks, double* sum_out,
                      hipStream_t st);
constexpr uint32_t kRmseBlocks = 1024;

// Every idx[q] (device array, q < n) must be < bound: MFX_ERR_INVALID naming the first offender
// otherwise ("%s index ... out of range").  Synchronises `st`.  Used on every index array that a
// kernel will use as an address without further checks (ALS gather indices, test-set rows/columns),
// so that a malformed input is an error message and not a GPU fault.
int check_index_range(const uint32_t* d_idx, uint64_t n, uint32_t bound, const char* what, hipStream_t st);

}  // namespace mfx

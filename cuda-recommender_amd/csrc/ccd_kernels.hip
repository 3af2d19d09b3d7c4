// ccd_kernels.hip -- CCD++ device kernels for gfx950 (CDNA4, wave64).
//
// Every kernel here is HBM-bound integer/fp32 streaming work (0.25-0.5 flop/byte): no MFMA, no
// GEMM reshaping.  What matters is (1) 16-byte coalesced walks over idx/val, (2) keeping the
// gathered factor vectors cache-resident (non-temporal hints on the streams), (3) nnz-balanced
// work so one 230k-entry column costs the same per wave as 230k entries of short rows, and
// (4) no atomics: every output has exactly one writer, so results are bitwise reproducible.
#include "ccd_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "flat_layout.hpp"

namespace mfx {
namespace {

constexpr int kBlock = 256;  // 4 wavefronts
constexpr uint32_t kPerSegLdsCap = 1024;  // per-segment operands staged in LDS per workgroup

// native vector types: the non-temporal builtins reject HIP's struct-based uint4/float4
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;

// Cross-lane moves on the VALU's DPP path (a few cycles) instead of ds_bpermute (an LDS round trip
// per step): rows are 16 lanes; row_shr:n shifts inside a row, row_bcast15 / row_bcast31 carry
// lane 15 / lane 31 into the following row(s), wave_shr:1 shifts the whole wavefront by one lane.
// Lanes without a source read 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, true));
}
constexpr int kDppRowShr = 0x110, kDppWaveShr1 = 0x138, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

// Sum over the 64 lanes, returned wave-uniform (fixed order: reproducible).
__device__ __forceinline__ float wave_sum(float x) {
    const uint32_t lane = threadIdx.x & 63;
    x += dpp_mov<kDppRowShr + 1>(x);
    x += dpp_mov<kDppRowShr + 2>(x);
    x += dpp_mov<kDppRowShr + 4>(x);
    x += dpp_mov<kDppRowShr + 8>(x);               // lane 15 of each row = row total
    const float r15 = dpp_mov<kDppBcast15, 0xA>(x);  // rows 1, 3 <- lane 15 of rows 0, 2
    if (lane & 16) x += r15;
    const float r31 = dpp_mov<kDppBcast31, 0xC>(x);  // rows 2, 3 <- lane 31
    if (lane & 32) x += r31;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// Inclusive segmented scan over lanes: lane l ends with the sum of x over lanes [h(l), l] where
// h(l) is the nearest lane <= l that starts a run (bit set in `heads`); dist = l - h(l), or >= 64
// when no such lane exists.  Unsegmented prefix when heads == 0.
__device__ __forceinline__ void seg_scan2(float& xg, float& xh, uint32_t lane, uint32_t dist) {
    const uint32_t in_row = lane & 15;
#define MFX_SCAN_STEP(D)                                                            \
    {                                                                               \
        const float yg = dpp_mov<kDppRowShr + D>(xg), yh = dpp_mov<kDppRowShr + D>(xh); \
        if (in_row >= D && dist >= D) { xg += yg; xh += yh; }                     \
    }
    MFX_SCAN_STEP(1) MFX_SCAN_STEP(2) MFX_SCAN_STEP(4) MFX_SCAN_STEP(8)
#undef MFX_SCAN_STEP
    {   // rows 1, 3 take lane 15 of the previous row if no run starts in [row start, l]
        const float yg = dpp_mov<kDppBcast15, 0xA>(xg), yh = dpp_mov<kDppBcast15, 0xA>(xh);
        if ((lane & 16) && dist > in_row) { xg += yg; xh += yh; }
    }
    {   // rows 2, 3 take lane 31 if no run starts in [32, l]
        const float yg = dpp_mov<kDppBcast31, 0xC>(xg), yh = dpp_mov<kDppBcast31, 0xC>(xh);
        if ((lane & 32) && dist > lane - 32) { xg += yg; xh += yh; }
    }
}

// Unfused multiply-then-add/sub, like the reference CPU build (no FMA contraction), so that the
// residual update is bit-identical to src/CCD.cpp:25,36 given identical operands.
// (HIP's __fmul_rn/__fadd_rn are plain `*`/`+` and get contracted into v_fma under hipcc's default
// -ffp-contract=fast; the pragma strips the `contract` flag from these operations so they cannot.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// ---------------------------------------------------------------------------------------------
// Variant 0: one wavefront per segment.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sweep_wave(uint32_t nseg, const uint32_t* __restrict__ ptr,
                                                       const uint32_t* __restrict__ idx,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ vec,
                                                       float* __restrict__ g_out, float* __restrict__ h_out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    for (uint32_t c = wave; c < nseg; c += nwaves) {
        const uint32_t lo = ptr[c], hi = ptr[c + 1];
        float g = 0.f, h = 0.f;
        for (uint32_t p = lo + lane; p < hi; p += 64) {
            const float x = vec[idx[p]];
            g += x * val[p];
            h += x * x;
        }
        g = wave_sum(g);
        h = wave_sum(h);
        if (lane == 0) {
            g_out[c] = g;
            h_out[c] = h;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_resid_wave(uint32_t nseg, const uint32_t* __restrict__ ptr,
                                                       const uint32_t* __restrict__ idx,
                                                       float* __restrict__ val,
                                                       const float* __restrict__ gathered,
                                                       const float* __restrict__ per_seg, int add) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    for (uint32_t c = wave; c < nseg; c += nwaves) {
        const uint32_t lo = ptr[c], hi = ptr[c + 1];
        const float b = per_seg[c];
        for (uint32_t p = lo + lane; p < hi; p += 64) {
            const float prod = mul_rn(gathered[idx[p]], b);
            val[p] = add ? add_rn(val[p], prod) : sub_rn(val[p], prod);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 1: flat-stream kernel.  One wavefront owns one span of tiles_per_span * 256 consecutive
// stored non-zeros; lane l of a tile owns elements 4l..4l+3 (one 16-byte load each of idx and val).
//
// LDS = true (panel layout): a workgroup of BLOCK/64 spans works inside ONE panel.  It first copies
// that panel's slice of the operand pack into LDS (coalesced, served by L2), then every per-nonzero
// gather is a ds_read instead of a global load.  Measured on MI355X (profiles/r01_ubench_gather.txt):
// the same pass runs at 1.8-2.3 TB/s with the gather going to L2 (one 64/128-byte L2 request per
// 8 useful bytes: request-rate bound) and at 5.5-5.9 TB/s -- the pure streaming rate -- from LDS.
// ---------------------------------------------------------------------------------------------
struct FlatArgs {
    const void* idx;  // uint32 (plain layout) or uint16 panel-local (LDS panels)
    float* val;
    const uint32_t* flags32;
    const uint32_t* hpre;
    const uint32_t* seg_of_rank;
    const uint32_t* wg_panel;
    uint32_t nspans;
    uint32_t tiles_per_span;
    uint32_t panel_rows;
    uint32_t gather_len;
    uint32_t nne;
    uint64_t nnz;  // plain layout: elements at or beyond nnz are padding (cache panels: = padded nnz)
    const void* gather;
    const void* perseg;
    float2* part;   // [nne] (g, h) of every non-empty virtual segment, written by the span holding its head
    float2* carry;  // [nspans] (g, h) of a span's leading run that continues an earlier span's segment
    int add;
    // ---- fused finalize (k_flat<..., FUSE = true>): the workgroup whose arrival completes a group of segments
    // turns that group's partial sums into the new factor entries inside the pass (see fused_finalize below)
    const uint32_t* wg_order;   // [workgroups] dispatch slot -> chunk, ascending first segment
    const uint32_t* wg_g0;      // [workgroups] first / last segment group a chunk contributes to (g0 > g1: none)
    const uint32_t* wg_g1;
    const uint32_t* expected;   // [groups] chunks contributing to a group
    uint32_t* arrived;          // [groups] zero between launches: the completing workgroup resets its word
    const uint32_t* orphans;    // [norphans] groups nobody contributes to (all segments empty): slot 0 finalizes them
    uint32_t norphans, ngroups, panel_lanes;
    // what k_finalize takes
    uint32_t nseg, npanels;
    const uint32_t* ptr_v;
    const uint32_t* rank_code;
    const uint32_t* seg_cnt;
    float lambda;
    float* out_vec;
    float2* pack2;
    const float* next_vec;
    float4* pack4;
};

// G: element of the gathered operand in global memory; S: the same inside the LDS slice (the fused CSR pass keeps
// 12 of its 16 bytes: 4 panels instead of 5 at the Netflix shape, and 4 VGPRs fewer per tile); P: per-segment operand
struct F3 { float x, y, z; };
template <typename S, typename G> __device__ __forceinline__ S to_slice(const G& g) { return g; }
template <> __device__ __forceinline__ F3 to_slice<F3, float4>(const float4& g) { return F3{g.x, g.y, g.z}; }
template <typename G, typename S> __device__ __forceinline__ G from_slice(const S& s) { return s; }
template <> __device__ __forceinline__ float4 from_slice<float4, F3>(const F3& s) { return make_float4(s.x, s.y, s.z, 0.f); }
template <int MODE> struct ModeTraits;
template <> struct ModeTraits<FM_SWEEP> { using G = float;  using S = float;  using P = float;  static constexpr bool kPerSeg = false, kWrite = false, kDot = true; };
template <> struct ModeTraits<FM_RESID> { using G = float;  using S = float;  using P = float;  static constexpr bool kPerSeg = true,  kWrite = true,  kDot = false; };
template <> struct ModeTraits<FM_FCSC>  { using G = float2; using S = float2; using P = float2; static constexpr bool kPerSeg = true,  kWrite = true,  kDot = true; };
template <> struct ModeTraits<FM_FCSR>  { using G = float4; using S = F3;     using P = float2; static constexpr bool kPerSeg = true,  kWrite = true,  kDot = true; };

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ float2 zero_of(float2) { return make_float2(0.f, 0.f); }
__device__ __forceinline__ float4 zero_of(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }

// One element: new residual value and its (g, h) contribution.
template <int MODE>
__device__ __forceinline__ void element_op(float v, const typename ModeTraits<MODE>::G& ga,
                                           const typename ModeTraits<MODE>::P& ps, int add,
                                           float& v_out, float& gc, float& hc) {
    if constexpr (MODE == FM_SWEEP) {
        v_out = v;
        gc = ga * v;
        hc = ga * ga;
    } else if constexpr (MODE == FM_RESID) {
        const float prod = mul_rn(ga, ps);
        v_out = add ? add_rn(v, prod) : sub_rn(v, prod);
        gc = 0.f;
        hc = 0.f;
    } else if constexpr (MODE == FM_FCSC) {
        v_out = add_rn(sub_rn(v, mul_rn(ga.x, ps.x)), mul_rn(ga.y, ps.y));
        gc = ga.y * v_out;
        hc = ga.y * ga.y;
    } else {
        v_out = add_rn(sub_rn(v, mul_rn(ga.x, ps.x)), mul_rn(ga.y, ps.y));
        gc = ga.z * v_out;
        hc = ga.z * ga.z;
    }
}

// PSCHK: some workgroup touches more ranks than the LDS per-segment window holds, so fetches
// must check the window and fall back to global memory (decided on the host from the layout).
template <int BLOCK> __device__ void fused_finalize(const FlatArgs& a, uint32_t chunk, unsigned char* lds_raw);

// agent-scope, write-through (sc1) 8-byte store / load of a partial sum: what another workgroup of the SAME launch
// may read needs no release fence on this side and no acquire on the other (cdna_hip_programming.md, guideline 16)
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) uint32_t;
template <bool FUSE>
__device__ __forceinline__ void store_partial(float2* p, float g, float h) {
    if constexpr (FUSE) {
        const unsigned long long bits = (unsigned long long) __builtin_bit_cast(uint32_t, g) | ((unsigned long long) __builtin_bit_cast(uint32_t, h) << 32);
        __hip_atomic_store((gu64*) p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = make_float2(g, h);
    }
}
__device__ __forceinline__ float2 load_partial_sc1(const float2* p) {
    const unsigned long long bits = __hip_atomic_load((gu64*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (uint32_t) bits), __builtin_bit_cast(float, (uint32_t) (bits >> 32)));
}

template <int MODE, bool LDS, int BLOCK, bool PSCHK, bool FUSE = false>
__global__ __launch_bounds__(BLOCK) void k_flat(FlatArgs a) {
    static_assert(!FUSE || ModeTraits<MODE>::kDot, "fused finalize: passes that produce sums");
    using TR = ModeTraits<MODE>;
    // FUSE: dispatch slot -> chunk through wg_order (ascending first segment, so that the chunks of one segment
    // group run at about the same time and groups complete all along the pass, not at its end)
    const uint32_t chunk = FUSE ? a.wg_order[blockIdx.x] : blockIdx.x;
    using G = typename TR::G;
    using P = typename TR::P;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    using S = typename TR::S;
    S* __restrict__ slice = reinterpret_cast<S*>(lds_raw);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t span = __builtin_amdgcn_readfirstlane(chunk * (BLOCK / 64) + (threadIdx.x >> 6));
    const G* __restrict__ gather = static_cast<const G*>(a.gather);
    const P* __restrict__ perseg = static_cast<const P*>(a.perseg);
    const uint32_t span_words = a.tiles_per_span * (kTileElems / 32);
    const uint64_t start = (uint64_t) span * a.tiles_per_span * kTileElems;
    // panel-local indices are 16-bit: 2 B/nnz instead of 4 (10 B/nnz per fused pass instead of 12)
    using IdxVec = typename std::conditional<LDS, u16x4, u32x4>::type;
    using IdxElem = typename std::conditional<LDS, uint16_t, uint32_t>::type;
    const IdxVec* __restrict__ idx4 = reinterpret_cast<const IdxVec*>(static_cast<const IdxElem*>(a.idx) + start) + lane;
    f32x4* __restrict__ val4 = reinterpret_cast<f32x4*>(a.val + start) + lane;
    // per-lane view of the head metadata: 8 lanes share one 32-element word
    const uint32_t* __restrict__ flw = a.flags32 + (size_t) span * span_words + (lane >> 3);
    const uint32_t* __restrict__ hpw = a.hpre + (size_t) span * span_words + (lane >> 3);
    const uint32_t sh = (lane & 7) * 4;
    IdxVec id_n;
    f32x4 v_n;
    uint32_t fl_n, hp_n, hp_nn;
    if constexpr (LDS) {
        // The first tile's streams go out BEFORE the slice is staged: a CU holds two of these workgroups, and while
        // one of them stages (two to three dependent round trips before its barrier) half the CU would otherwise
        // have nothing in flight.  (LDS panels: every chunk holds BLOCK / 64 whole spans, so the span exists.)
        id_n = __builtin_nontemporal_load(idx4);
        v_n = __builtin_nontemporal_load(val4);
        fl_n = flw[0]; hp_n = hpw[0]; hp_nn = hpw[8];
        // stage this workgroup's panel slice; slot panel_rows is the zero entry padding points at
        const uint32_t panel = a.wg_panel[chunk];
        const uint32_t gbase = panel * a.panel_rows;
        const uint32_t cnt = a.gather_len - gbase < a.panel_rows ? a.gather_len - gbase : a.panel_rows;
        for (uint32_t i = threadIdx.x; i < cnt; i += BLOCK) slice[i] = to_slice<S, G>(gather[gbase + i]);
        if (threadIdx.x == 0) slice[a.panel_rows] = S{};
    }
    // The workgroup touches a contiguous window of ranks; stage their per-segment operands next to
    // the slice so that segmented tiles read LDS instead of chasing seg_of_rank -> perseg through L2.
    P* __restrict__ ps_lds = reinterpret_cast<P*>(lds_raw + (((size_t) a.panel_rows + 1) * sizeof(S) + 15) / 16 * 16);
    uint32_t win_base = 0;
    if constexpr (LDS && TR::kPerSeg) {
        const uint32_t first = chunk * (BLOCK / 64);
        const uint32_t rb0 = a.hpre[(size_t) first * span_words];
        win_base = rb0 > 0 ? rb0 - 1 : 0;
        const uint32_t win_end = a.hpre[(size_t) (first + BLOCK / 64) * span_words];  // heads before the next chunk
        uint32_t cnt = win_end - win_base;
        if (cnt > kPerSegLdsCap) cnt = kPerSegLdsCap;
        for (uint32_t j = threadIdx.x; j < cnt; j += BLOCK) ps_lds[j] = perseg[a.seg_of_rank[win_base + j]];
    }
    if constexpr (LDS) __syncthreads();
    // a wave without work (plain layout: a span beyond the last one, or one that holds only padding) skips the stream loop;
    // with the fused finalize it still joins the epilogue's barriers
    bool live = span < a.nspans;  // (LDS panels: always true, every chunk holds BLOCK / 64 spans)
    if constexpr (!FUSE) { if (!live) return; }
    auto fetch_ps = [&](uint32_t r) -> P {  // r: rank, always valid where this is called
        if constexpr (LDS) {
            const uint32_t rl = r - win_base;
            if constexpr (!PSCHK) return ps_lds[rl];
            else if (rl < kPerSegLdsCap) return ps_lds[rl];
        }
        return perseg[a.seg_of_rank[r]];
    };
    uint32_t ntiles = a.tiles_per_span;
    if constexpr (!LDS) {  // plain layout: tiles that start at or beyond nnz hold only padding
        const uint64_t left = a.nnz > start ? a.nnz - start : 0;
        const uint32_t nlive = (uint32_t) ((left + kTileElems - 1) / kTileElems);
        if (nlive < ntiles) ntiles = nlive;
        if (live && ntiles == 0) {  // nothing stored here (plain layout, nnz == 0): still own the carry slot
            if (TR::kDot && lane == 0) store_partial<FUSE>(a.carry + span, 0.f, 0.f);
            live = false;
        }
        if constexpr (!FUSE) { if (!live) return; }
        if (live) {
            id_n = __builtin_nontemporal_load(idx4);
            v_n = __builtin_nontemporal_load(val4);
            fl_n = flw[0]; hp_n = hpw[0]; hp_nn = hpw[8];
        }
    }
    if (live) {
    const uint32_t rank_base = __builtin_amdgcn_readfirstlane(hp_n);  // heads before this span
    uint32_t cur1 = rank_base;  // (rank of the segment open at the current position) + 1, wave-uniform
    P pcur{};
    if constexpr (TR::kPerSeg) {
        if (cur1 > 0) pcur = fetch_ps(cur1 - 1);
    }
    float og = 0.f, oh = 0.f;  // per-lane sums of the open segment since its last head
    bool open_spread = false;  // wave-uniform: og/oh hold per-lane partials (else only lane 0 is non-zero)

    for (uint32_t tile = 0; tile < ntiles; ++tile) {
        const IdxVec id = id_n;
        const f32x4 v = v_n;
        const uint32_t fl = fl_n, hp = hp_n;
        // heads before the NEXT tile decide right now whether this tile is segmented, so that count
        // is fetched two tiles ahead (it arrived during the previous iteration: no wait here);
        // streams and head bits run one tile ahead.  The metadata arrays carry 16 spare words, so
        // reading two tiles past the last span stays in bounds (and yields heads_total).
        const uint32_t cur1_next = __builtin_amdgcn_readfirstlane(hp_nn);
        hp_n = hp_nn;
        if (tile + 1 < ntiles) {
            id_n = __builtin_nontemporal_load(idx4 + (tile + 1) * 64);
            v_n = __builtin_nontemporal_load(val4 + (tile + 1) * 64);
        }
        fl_n = flw[(tile + 1) * 8];
        hp_nn = hpw[(tile + 2) * 8];
        const uint64_t base = start + (uint64_t) tile * kTileElems;
        // plain layout only: the tile that straddles nnz needs its padding masked; with LDS panels
        // padding gathers the zero slot and contributes exact zeros
        const bool partial = !LDS && base + kTileElems > a.nnz;
        const uint64_t e0 = base + (uint64_t) lane * 4;

        const uint32_t ids[4] = {(uint32_t) id.x, (uint32_t) id.y, (uint32_t) id.z, (uint32_t) id.w};
        const float vs[4] = {v.x, v.y, v.z, v.w};
        G ga[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (LDS) ga[e] = from_slice<G, S>(slice[ids[e]]);
            else ga[e] = ids[e] < a.gather_len ? gather[ids[e]] : zero_of(G{});  // cache panels pad with index G
        }
        float vo[4], gc[4], hc[4];

        if (cur1_next == cur1) {
            // ---- no segment starts in this tile: everything belongs to the open segment ----
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                element_op<MODE>(vs[e], ga[e], pcur, a.add, vo[e], gc[e], hc[e]);
                if (partial && e0 + e >= a.nnz) { gc[e] = 0.f; hc[e] = 0.f; }
                if constexpr (TR::kDot) { og += gc[e]; oh += hc[e]; }
            }
            open_spread = true;
        } else {
            // ---- segmented tile ----
            const uint32_t nib = (fl >> sh) & 0xFu;
            // (rank open at this lane's first element) + 1 = heads before it, globally
            const uint32_t r1 = hp + (uint32_t) __popc(fl & ((1u << sh) - 1u));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                P ps{};
                if constexpr (TR::kPerSeg) {
                    // element e's rank + 1 = r1 + heads among elements 0..e of this lane (>= 1 always:
                    // the very first stored element is a head)
                    const uint32_t re1 = r1 + (uint32_t) __popc(nib & ((2u << e) - 1u));
                    ps = fetch_ps(re1 - 1);
                }
                element_op<MODE>(vs[e], ga[e], ps, a.add, vo[e], gc[e], hc[e]);
                if (partial && e0 + e >= a.nnz) { gc[e] = 0.f; hc[e] = 0.f; }
            }
            if constexpr (TR::kDot) {
                // carry-in: the open segment's sum so far, as a wave-uniform value
                float cin_g, cin_h;
                if (open_spread) {
                    cin_g = wave_sum(og);
                    cin_h = wave_sum(oh);
                } else {  // after a segmented tile the open sum sits in lane 0
                    cin_g = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, og)));
                    cin_h = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oh)));
                }
                open_spread = false;
                // lane-serial pass: fg/fh = run before the lane's first head, ag/ah = run after its
                // last head; runs between two heads of the same lane are complete segments.
                float ag = 0.f, ah = 0.f, fg = 0.f, fh = 0.f;
                bool seen = false;
                uint32_t close1 = r1;  // (rank closed by the lane's next head) + 1
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((nib >> e) & 1u) {
                        if (!seen) {
                            fg = ag; fh = ah; seen = true;
                        } else {  // head..head inside one lane: started in this span by construction
                            store_partial<FUSE>(a.part + (close1 - 1), ag, ah);
                        }
                        ++close1;
                        ag = 0.f; ah = 0.f;
                    }
                    ag += gc[e];
                    ah += hc[e];
                }
                if (!seen) { fg = ag; fh = ah; }
                // segmented inclusive scan over lanes of the right-propagating value
                const uint64_t M = __ballot(seen);
                const uint64_t upto = M & ((uint64_t(2) << lane) - 1);  // head lanes <= this lane
                const uint32_t dist = upto ? lane - (63u - (uint32_t) __clzll((long long) upto)) : 64u;
                float xg = ag, xh = ah;
                seg_scan2(xg, xh, lane, dist);
                float eg = dpp_mov<kDppWaveShr1>(xg), eh = dpp_mov<kDppWaveShr1>(xh);  // exclusive: lane l-1
                if ((M & ((uint64_t(1) << lane) - 1)) == 0) { eg += cin_g; eh += cin_h; }
                if (seen) {  // this lane's first head closes the segment of rank r1 - 1
                    const float tg = eg + fg, th = eh + fh;
                    if (r1 > rank_base) {  // it started inside this span: we own its slot
                        store_partial<FUSE>(a.part + (r1 - 1), tg, th);
                    } else {  // it started in an earlier span: this is the span's head carry
                        store_partial<FUSE>(a.carry + span, tg, th);
                    }
                }
                // new open segment: everything after the tile's last head
                const float ng = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xg), 63));
                const float nh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xh), 63));
                og = lane == 0 ? ng : 0.f;
                oh = lane == 0 ? nh : 0.f;
            }
            cur1 = cur1_next;
            if constexpr (TR::kPerSeg) pcur = fetch_ps(cur1 - 1);
        }
        if constexpr (TR::kWrite) {
            if (!partial) {
__builtin_nontemporal_store(f32x4{vo[0], vo[1], vo[2], vo[3]}, val4 + tile * 64);  // (write-through sc1 stores instead: passes +4 %, finalize +4 us)
            } else {
                float* vp = a.val + e0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e0 + e < a.nnz) vp[e] = vo[e];
            }
        }
    }
    if constexpr (TR::kDot) {
        const float tg = open_spread ? wave_sum(og) : og, th = open_spread ? wave_sum(oh) : oh;
        if (lane == 0) {
            if (cur1 > rank_base) {  // the open segment's head lies in this span: we own its slot
                store_partial<FUSE>(a.part + (cur1 - 1), tg, th);
            } else {  // the whole span is interior to one segment
                store_partial<FUSE>(a.carry + span, tg, th);
            }
        }
    }
    }  // live
    if constexpr (FUSE) fused_finalize<BLOCK>(a, chunk, lds_raw);
}


// ---------------------------------------------------------------------------------------------
// (r4) Segment-owner fused pass for SMALL matrices (plain layout, a few million ratings at most): pass AND finalize in
// one launch, without any cross-workgroup step.  Small shapes are launch-bound -- under hipGraph replay a dependent
// node costs ~4.8 us whether it streams 1e5 or 1e6 ratings (profiles/r04_kernel_stats_ml1m.csv) -- so an outer
// iteration is (nodes per rank) x k x ~5 us, and the flat-stream form needs four nodes per rank: its nnz-balanced
// spans cut segments across workgroups, so the sums must meet in a second kernel (k_finalize), and meeting inside
// the pass ("last workgroup finalizes") costs more than the node it saves (ccd_solver.hpp, fuse_finalize_).  Here a
// segment has ONE owner instead: a wavefront for a short segment, a whole workgroup for a long one (the lists are
// made once per store, longest first).  The owner has the complete (g, h), divides, and writes the factor entry and
// the operand packs of the next passes itself: two nodes per rank.  Measured (profiles/r04_exp_small.txt), ML-1M shape
// k = 40: 0.795 -> 0.538 ms per outer iteration, ML-100K shape k = 10: 0.188 -> 0.115 ms; the kernel itself takes what
// the flat pass takes (8.7 vs 8.9 us per eager launch) -- the finalize nodes are simply gone.
// What mattered on the way (same file): 256-thread workgroups, not 1024 (0.75 -> 0.60 ms: a 16-wave workgroup waits for a
// CU with that many free slots); one 256-entry tile per wave and round, not two (fewer registers: 0.60 -> 0.54); a long-
// segment threshold of 1024 entries (256 ... 16384 swept).  What did not: 16-byte against 4-byte accesses, deeper unrolling,
// the length of the owner's dependent load chain.
// Per element the arithmetic is element_op's (the reference's unfused update); a sum is added lane by lane over the
// owner's quads in order, then over the wave on the DPP path, then over the waves in order: fixed, hence bitwise reproducible.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSegOwnerLong = 1024;  // entries from which a segment gets a whole workgroup (a wave walks up to four 256-entry tiles; MFX_OWNER_LONG overrides, A/B)
#ifndef MFX_OWNER_UL
#define MFX_OWNER_UL 2
#define MFX_OWNER_US 1
#endif
#ifndef MFX_SEG_BLOCK
#define MFX_SEG_BLOCK 256
#endif
constexpr int kSegBlock = MFX_SEG_BLOCK;
struct SegOwnerArgs {
    const uint32_t* ptr;   // [nseg + 1] input-order pointers (plain layout)
    const uint32_t* idx;
    float* val;
    const void* gather;
    const void* perseg;
    const uint4* long_list;      // [nlong] (segment, first entry, end, -) one workgroup each
    const uint4* short_list;     // [nshort] one wavefront each, longest first.  The owner's whole address chain hangs on this ONE
                                 // load (whose address is known at launch): entry -> indices -> gathered operand, three round trips
    uint32_t nlong, nshort;
    float lambda;
    float* out_vec;
    float2* pack2;
    const float* next_vec;
    float4* pack4;
};

template <int MODE>
__device__ __forceinline__ void seg_owner_finish(const SegOwnerArgs& a, uint32_t c, uint32_t cnt, float g, float h,
                                                 const typename ModeTraits<MODE>::P& ps, float next_old) {
    // reference: g / (lambda * |Omega| + sum u^2), 0 for an empty segment (src/CCD.cpp:6-16,112)
    const float x = cnt ? g / add_rn(mul_rn(a.lambda, (float) cnt), h) : 0.f;
    a.out_vec[c] = x;
    if constexpr (MODE == FM_FCSC || MODE == FM_FCSR) {  // perseg IS pack2: (prev_new, cur_old) of this segment, read by its owner only
        if (a.pack4) a.pack4[c] = make_float4(ps.x, ps.y, x, 0.f);
        a.pack2[c] = make_float2(x, a.next_vec == a.out_vec ? x : next_old);
    }
}

template <int MODE>
__global__ __launch_bounds__(kSegBlock) void k_seg_owner(SegOwnerArgs a) {
    using TR = ModeTraits<MODE>;
    using G = typename TR::G;
    using P = typename TR::P;
    const G* __restrict__ gather = static_cast<const G*>(a.gather);
    const P* __restrict__ perseg = static_cast<const P*>(a.perseg);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float sg[kSegBlock / 64], sh[kSegBlock / 64];
    // Lane l of a tile owns FOUR CONSECUTIVE entries (one 16-byte load each of indices and values, one 16-byte store), like
    // the flat kernel: the texture addresser takes ~16 cycles per wave instruction whatever its width, and a first version
    // with one dword per lane and instruction (40 memory instructions per 512 entries instead of 14) spent the kernel there
    // (ML-1M shape: 11 us per launch).  A tile starts at a multiple of four entries (16-byte alignment); positions outside
    // [lo, hi) -- the neighbours' entries in the first / last quad, clamped re-reads past the end -- are masked out of the
    // sums and never stored (a partial quad is stored entry by entry).  U tiles per round: all index / value loads of a
    // round go out together, then its 4 U gathers.
    auto walk = [&](uint32_t lo, uint32_t hi, uint32_t first, uint32_t stride, const P& ps, float& g, float& h, auto unroll) {
        constexpr int U = decltype(unroll)::value;
        g = 0.f; h = 0.f;
        if (hi == lo) return;
        const uint32_t base = lo & ~3u;
        uint32_t pq[U];
        u32x4 id4[U];
        f32x4 v4[U];
        auto streams = [&](uint32_t p0, uint32_t (&pp)[U], u32x4 (&ii)[U], f32x4 (&vals)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pp[u] = p0 + (uint32_t) u * 4u * stride;
                const uint32_t at = pp[u] < hi ? pp[u] : base;  // (clamped: masked below through pp)
                ii[u] = *reinterpret_cast<const u32x4*>(a.idx + at);
                vals[u] = *reinterpret_cast<const f32x4*>(a.val + at);
            }
        };
        uint32_t p = base + 4u * first;
        if (p < hi) streams(p, pq, id4, v4);
        while (p < hi) {
            G ga[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) ga[u][e] = gather[id4[u][e]];
            // the next round's streams go out behind this round's gathers: a steady-state round is ONE round trip, not two
            const uint32_t pn = p + 4u * stride * U;
            uint32_t pqn[U];
            u32x4 id4n[U];
            f32x4 v4n[U];
            if (pn < hi) streams(pn, pqn, id4n, v4n);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 o;
                bool all_live = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t pos = pq[u] + (uint32_t) e;
                    const bool live = pos >= lo && pos < hi;
                    all_live &= live;
                    float vo, gc, hc;
                    element_op<MODE>(v4[u][e], ga[u][e], ps, 0, vo, gc, hc);
                    o[e] = vo;
                    g += live ? gc : 0.f; h += live ? hc : 0.f;
                }
                if constexpr (TR::kWrite) {
                    if (all_live) {
                        *reinterpret_cast<f32x4*>(a.val + pq[u]) = o;
                    } else if (pq[u] < hi) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const uint32_t pos = pq[u] + (uint32_t) e; if (pos >= lo && pos < hi) a.val[pos] = o[e]; }
                    }
                }
            }
            p = pn;
            if (p < hi) {
#pragma unroll
                for (int u = 0; u < U; ++u) { pq[u] = pqn[u]; id4[u] = id4n[u]; v4[u] = v4n[u]; }
            }
        }
    };
    if (blockIdx.x < a.nlong) {  // ---- a long segment: the whole workgroup
        const uint4 e = a.long_list[blockIdx.x];
        const uint32_t c = e.x, lo = e.y, hi = e.z;
        P ps{};
        if constexpr (TR::kPerSeg) ps = perseg[c];
        float g, h;
        walk(lo, hi, threadIdx.x, kSegBlock, ps, g, h, std::integral_constant<int, MFX_OWNER_UL>{});
        g = wave_sum(g); h = wave_sum(h);
        if (lane == 0) { sg[wave] = g; sh[wave] = h; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float tg = 0.f, th = 0.f;
            for (int w = 0; w < kSegBlock / 64; ++w) { tg += sg[w]; th += sh[w]; }
            seg_owner_finish<MODE>(a, c, hi - lo, tg, th, ps, a.next_vec ? a.next_vec[c] : 0.f);
        }
        return;
    }
    // ---- short segments: one per wavefront, grid-strided over the list
    const uint32_t nwaves = (gridDim.x - a.nlong) * (kSegBlock / 64);
    for (uint32_t q = (blockIdx.x - a.nlong) * (kSegBlock / 64) + wave; q < a.nshort; q += nwaves) {
        const uint4 e = a.short_list[q];
        const uint32_t c = e.x, lo = e.y, hi = e.z;
        P ps{};
        if constexpr (TR::kPerSeg) ps = perseg[c];
        const float next_old = a.next_vec ? a.next_vec[c] : 0.f;  // (needed last: fetched first)
        float g, h;
        walk(lo, hi, lane, 64, ps, g, h, std::integral_constant<int, MFX_OWNER_US>{});
        g = wave_sum(g); h = wave_sum(h);
        if (lane == 0) seg_owner_finish<MODE>(a, c, hi - lo, g, h, ps, next_old);
    }
}

// Adds, in span order, the carries that belong to the stored range [lo, hi).  SC1: the partials were written by
// other workgroups of the SAME launch (fused finalize) and are read past the caches.
template <bool SC1>
__device__ __forceinline__ float2 load_partial(const float2* p) {
    if constexpr (SC1) return load_partial_sc1(p);
    else return *p;
}
template <bool SC1 = false>
__device__ __forceinline__ void add_carries(uint32_t lo, uint32_t hi, uint32_t span_len, const float2* __restrict__ carry,
                                            float& g, float& h) {
    uint32_t s = lo / span_len + 1;
    const uint32_t s_end = (hi - 1) / span_len;  // inclusive
    for (; s + 3 <= s_end; s += 4) {  // 4 independent loads in flight, sequential adds
        const float2 c0 = load_partial<SC1>(carry + s), c1 = load_partial<SC1>(carry + s + 1), c2 = load_partial<SC1>(carry + s + 2),
                     c3 = load_partial<SC1>(carry + s + 3);
        g += c0.x; g += c1.x; g += c2.x; g += c3.x;
        h += c0.y; h += c1.y; h += c2.y; h += c3.y;
    }
    for (; s <= s_end; ++s) { const float2 c = load_partial<SC1>(carry + s); g += c.x; h += c.y; }
}

struct GatherPartsArgs {
    uint32_t nseg, npanels, span_len;
    const uint32_t* ptr_v;
    const uint32_t* rank_code;  // [npanels*nseg] kNoRank for an empty virtual segment, else rank | kCarryBit
    const float2* part;
    const float2* carry;
};
constexpr uint32_t kNoRank = 0xFFFFFFFFu, kCarryBit = 0x80000000u;

// (g, h) of real segment c over panels p0, p0 + stride, ...: each virtual segment = part + carries.
// A lookup is rank -> part (two dependent loads; the carry bit says whether the segment runs into
// later spans -- rare -- and only then are its pointers read), issued for kBatch panels at a time.
// The additions stay in panel order.
template <bool SC1 = false>
__device__ __forceinline__ void segment_sums(const GatherPartsArgs& a, uint32_t c, uint32_t p0, uint32_t stride,
                                             float& g, float& h) {
    constexpr int kBatch = 5;
    g = 0.f;
    h = 0.f;
    for (uint32_t pb = p0; pb < a.npanels; pb += stride * kBatch) {
        uint32_t r[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const uint32_t p = pb + q * stride;
            r[q] = p < a.npanels ? a.rank_code[(size_t) p * a.nseg + c] : kNoRank;
        }
        float2 pp[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; ++q) pp[q] = r[q] != kNoRank ? load_partial<SC1>(a.part + (r[q] & ~kCarryBit)) : make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            if (r[q] != kNoRank) {
                if (r[q] & kCarryBit) {
                    const size_t v = (size_t) (pb + q * stride) * a.nseg + c;
                    add_carries<SC1>(a.ptr_v[v], a.ptr_v[v + 1], a.span_len, a.carry, pp[q].x, pp[q].y);
                }
                g += pp[q].x;
                h += pp[q].y;
            }
        }
    }
}

// PL "panel lanes" cooperate on one segment (each walks every PL-th panel: the per-panel lookups
// are a chain of three dependent loads, so 59 panels walked by one thread cost ~40 us); their
// partial sums are added in lane order through LDS -- a fixed order, so still reproducible.
template <int PL>
__device__ __forceinline__ bool block_segment_sums(const GatherPartsArgs& a, uint32_t& c, float& g, float& h) {
    constexpr int SEGS = kBlock / PL;
    __shared__ float sg[kBlock], sh[kBlock];
    const uint32_t sl = threadIdx.x % SEGS, pl = threadIdx.x / SEGS;
    c = blockIdx.x * SEGS + sl;
    g = 0.f;
    h = 0.f;
    if (c < a.nseg) segment_sums(a, c, pl, PL, g, h);
    if constexpr (PL > 1) {
        sg[threadIdx.x] = g;
        sh[threadIdx.x] = h;
        __syncthreads();
        if (pl == 0) {
            for (int q = 1; q < PL; ++q) { g += sg[q * SEGS + sl]; h += sh[q * SEGS + sl]; }
        }
    }
    return pl == 0 && c < a.nseg;
}

template <int PL>
__global__ __launch_bounds__(kBlock) void k_combine_dense(GatherPartsArgs a, float* __restrict__ gh) {
    uint32_t c;
    float g, h;
    if (!block_segment_sums<PL>(a, c, g, h)) return;
    gh[c] = g;
    gh[a.nseg + c] = h;
}

struct FinKernelArgs {
    GatherPartsArgs parts;
    const uint32_t* seg_cnt;
    const float* gh_dense;
    uint32_t seg_base, gh_len;  // dense source: segments [seg_base, seg_base + gh_len), g then h in gh_dense (a panel group)
    // ... or (r4) the fixed-point slabs of a scatter pass themselves: k_scatter_combine's sum and conversion done here, no dense
    // buffer and no combine launch in between (whenever no all-reduce sits between the pass and its finalize)
    const unsigned long long* slab_acc;
    const uint32_t* slab_lo;
    const uint32_t* slab_bad;
    uint32_t slab_pr;
    const uint32_t* cnt_override;
    float lambda;
    float* out_vec;
    float2* pack2;
    const float* next_vec;
    float4* pack4;
    bool pack4_as3;
    bool nmf;
    double* fundec_seg;
};

template <int PL>
__global__ __launch_bounds__(kBlock) void k_finalize(FinKernelArgs a) {
    // The lane that will own segment c is known up front: fetch what it needs at the very end (count,
    // old pack entry, next vector entry) before the partial sums are chased, so that those loads do
    // not queue up behind the rank -> part -> (barrier) chain.
    constexpr int SEGS = kBlock / PL;
    const bool flat = !a.gh_dense && !a.slab_acc;
    const uint32_t dl = blockIdx.x * kBlock + threadIdx.x;  // dense source: index inside the group's block
    const uint32_t c0 = flat ? blockIdx.x * SEGS + threadIdx.x % SEGS : a.seg_base + dl;
    const bool owner = flat ? (c0 < a.parts.nseg && threadIdx.x / SEGS == 0) : dl < a.gh_len;
    uint32_t cnt = 0;
    float2 old = make_float2(0.f, 0.f);
    float next = 0.f;
    if (owner) {
        cnt = a.cnt_override ? a.cnt_override[c0] : a.seg_cnt[c0];
        if (a.pack2) { old = a.pack2[c0]; next = a.next_vec[c0]; }
    }
    uint32_t c;
    float g, h;
    if (a.slab_acc) {  // the slabs of the panel of c, added as integers (any order: the same bits), exactly as k_scatter_combine does
        c = c0;
        if (dl >= a.gh_len) return;
        const uint32_t p = c / a.slab_pr, l = c - p * a.slab_pr;
        unsigned long long ig = 0, ih = 0;
        uint32_t bad = 0;
        for (uint32_t w = a.slab_lo[p]; w < a.slab_lo[p + 1]; ++w) {
            const unsigned long long* sl = a.slab_acc + (size_t) w * 2 * a.slab_pr + 2 * l;
            ig += sl[0]; ih += sl[1];
            bad |= a.slab_bad[w];
        }
        constexpr double inv = 1.0 / 68719476736.0;  // 2^-36 (ccd_scatter.hip, to_fixed)
        g = bad ? __builtin_nanf("") : (float) ((double) (long long) ig * inv);
        h = bad ? __builtin_nanf("") : (float) ((double) (long long) ih * inv);
    } else if (a.gh_dense) {  // PL == 1 by construction
        c = c0;
        if (dl >= a.gh_len) return;
        g = a.gh_dense[dl];
        h = a.gh_dense[a.gh_len + dl];
    } else if (!block_segment_sums<PL>(a.parts, c, g, h)) {
        return;
    }
    // reference: g / (lambda * |Omega| + sum u^2), 0 for an empty segment (src/CCD.cpp:6-16,112)
    const float den = add_rn(mul_rn(a.lambda, (float) cnt), h);
    float x = cnt ? g / den : 0.f;
    if (a.nmf || a.fundec_seg) {  // opt-in extensions (LIBPMF meaning of -N / -e; see FinalizeArgs)
        const float was = a.pack2 ? old.y : a.out_vec[c];
        double fd = 0.0;
        if (cnt) {
            if (a.nmf && x < 0.f) {
                x = 0.f;
                fd = -2.0 * (double) g * (double) was + (double) den * (double) was * (double) was;
            } else {
                const double delta = (double) was - (double) x;
                fd = (double) den * delta * delta;
            }
        }
        if (a.fundec_seg) a.fundec_seg[c] = fd;
    }
    a.out_vec[c] = x;
    if (a.pack2) {
        if (a.pack4) {
            if (a.pack4_as3) {
                float* p3 = reinterpret_cast<float*>(a.pack4) + 3 * (size_t) c;
                p3[0] = old.x; p3[1] = old.y; p3[2] = x;
            } else {
                a.pack4[c] = make_float4(old.x, old.y, x, 0.f);
            }
        }
        // k = 1: the "next" rank is this one, so its old value is the x just written (out_vec aliases next_vec)
        a.pack2[c] = make_float2(x, a.next_vec == a.out_vec ? x : next);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused finalize (verdict r1, item 8) -- OPT-IN (MFX_FUSE_FINALIZE=1), because it measured slower.
// k_finalize is two dependent HBM round trips behind a kernel boundary: 14-16 us, twice per rank, 8.6 % of
// the Netflix-shape step.  Fused: segments are cut into groups of GS = BLOCK / panel_lanes; the layout
// knows how many chunks contribute to a group (`expected`); a workgroup that has finished its chunk adds
// 1 to the counter of every group it contributes to, and the workgroup whose add completes a group
// finalizes it on the spot -- same lookups, same panel-lane split, same order of additions as
// k_finalize, hence the same bits (tests: test_fused_finalize_equals_the_separate_kernel, and at the
// Netflix shape test_fullsize_fused_finalize_is_bit_identical).
// Visibility (cdna_hip_programming.md guideline 16, the fan-in row of MI355X_MICROARCH.md's table):
// every partial is an sc1 (write-through) store; every wave drains its stores (vmcnt(0)) and the
// workgroup meets before ONE agent-scope atomic add per group; the workgroup whose add returned
// expected - 1 reads the partials with sc1 loads, after a barrier behind that add.  No wave waits for
// another workgroup: nothing can hang.  The counters return to zero by themselves (the completing
// workgroup resets its word; the next launch is a kernel boundary away).
// Measured, Netflix shape, per launch (tools/exp_fuse.sh, profiles/r02_exp_fuse.txt):
//   separate kernels                      CSC 184 + 17 us   CSR 189 + 17 us     25.7 ms per outer iteration
//   fused, stored (panel-major) order     256               243                 31.8
//   fused, dispatch by first segment      302               316                 39.1   (all 67 slices live at once)
//   fused kernel, nobody completes + k_finalize   211 + 17  212 + 17            28.9   (plain stores: 207 / 207)
// i.e. the arrival alone -- drain the stores, meet, one returning atomic -- costs ~20 us per pass: with 64 KB
// of LDS per workgroup a CU holds two workgroups, each of the ~6 it runs in a row idles half the CU for
// the 2-3 us of that round trip; and in stored order every group completes while the LAST panel is being
// processed, so its ~45 workgroups do the whole finalize (6 serial group steps each).  A kernel boundary
// is cheaper than both.
template <int BLOCK>
__device__ __forceinline__ void finalize_group(const FlatArgs& a, const GatherPartsArgs& ga, uint32_t group, float* sg, float* sh) {
    const uint32_t PL = a.panel_lanes, SEGS = BLOCK / PL;
    const uint32_t sl = threadIdx.x % SEGS, pl = threadIdx.x / SEGS;
    const uint32_t c = group * SEGS + sl;
    const bool owner = pl == 0 && c < a.nseg;
    uint32_t cnt = 0;
    float2 old = make_float2(0.f, 0.f);
    float next = 0.f;
    if (owner) {
        cnt = a.seg_cnt[c];
        if (a.pack2) { old = a.pack2[c]; next = a.next_vec[c]; }
    }
    float g = 0.f, h = 0.f;
    if (c < a.nseg) segment_sums<true>(ga, c, pl, PL, g, h);
    if (PL > 1) {
        sg[threadIdx.x] = g;
        sh[threadIdx.x] = h;
        __syncthreads();
        if (pl == 0)
            for (uint32_t q = 1; q < PL; ++q) { g += sg[q * SEGS + sl]; h += sh[q * SEGS + sl]; }
        __syncthreads();
    }
    if (!owner) return;
    const float x = cnt ? g / add_rn(mul_rn(a.lambda, (float) cnt), h) : 0.f;
    a.out_vec[c] = x;
    if (a.pack2) {
        if (a.pack4) a.pack4[c] = make_float4(old.x, old.y, x, 0.f);
        a.pack2[c] = make_float2(x, a.next_vec == a.out_vec ? x : next);
    }
}

template <int BLOCK>
__device__ void fused_finalize(const FlatArgs& a, uint32_t chunk, unsigned char* lds_raw) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its partials have left the CU
    __syncthreads();
    float* sg = reinterpret_cast<float*>(lds_raw);     // the slice and the operand window are dead now
    float* sh = sg + BLOCK;
    uint32_t* todo = reinterpret_cast<uint32_t*>(sh + BLOCK);  // [0] = count, then the groups this workgroup completed
    if (threadIdx.x == 0) todo[0] = 0;
    __syncthreads();
    const uint32_t g0 = a.wg_g0[chunk], g1 = a.wg_g1[chunk];
    if (g0 <= g1) {
        for (uint32_t g = g0 + threadIdx.x; g <= g1; g += BLOCK) {
            const uint32_t before = __hip_atomic_fetch_add((gu32*) (a.arrived + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == a.expected[g]) {
                __hip_atomic_store((gu32*) (a.arrived + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                todo[1 + atomicAdd(&todo[0], 1u)] = g;
            }
        }
    }
    __syncthreads();  // behind the adds: the sc1 loads below come after the add that completed their group
    GatherPartsArgs ga;
    ga.nseg = a.nseg; ga.npanels = a.npanels; ga.span_len = a.tiles_per_span * kTileElems; ga.ptr_v = a.ptr_v;
    ga.rank_code = a.rank_code; ga.part = a.part; ga.carry = a.carry;
    const uint32_t ntodo = todo[0];
    for (uint32_t i = 0; i < ntodo; ++i) finalize_group<BLOCK>(a, ga, todo[1 + i], sg, sh);
    if (blockIdx.x == 0)
        for (uint32_t i = 0; i < a.norphans; ++i) finalize_group<BLOCK>(a, ga, a.orphans[i], sg, sh);
}

__global__ __launch_bounds__(kBlock) void k_unpermute(uint64_t n, const uint32_t* __restrict__ perm,
                                                      const float* __restrict__ val, float* __restrict__ out) {
    const uint64_t e = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    const uint32_t q = perm[e];
    if (q != ~0u) out[q] = val[e];
}

// The same when every virtual segment is one run of consecutive input positions (flat_layout.hpp,
// perm_is_runs): one wavefront per virtual segment, out[first_q[v] + k] = val[ptr_v[v] + k].
template <bool TR>  // TR: tiles stored transposed (scatter layout): position d of the panel-major order sits at tile + 4 (d & 63) + ((d >> 6) & 3)
__global__ __launch_bounds__(kBlock) void k_unpermute_runs(uint32_t nv, uint32_t nseg, const uint32_t* __restrict__ ptr_v,
                                                           const uint32_t* __restrict__ first_q,
                                                           const uint32_t* __restrict__ panel_end,
                                                           const float* __restrict__ val, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
    for (uint32_t v = wave; v < nv; v += nwaves) {
        const uint32_t lo = ptr_v[v], pe = panel_end[v / nseg];
        uint32_t hi = ptr_v[v + 1];
        if (hi > pe) hi = pe;  // a panel's last virtual segment also spans the padding
        if (hi <= lo) continue;
        const uint32_t q0 = first_q[v];
        for (uint32_t k = lane; k < hi - lo; k += 64) {
            const uint32_t d = lo + k;
            out[q0 + k] = val[TR ? ((d & ~255u) | ((d & 63u) << 2) | ((d >> 6) & 3u)) : d];
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_pack2(uint32_t n, const float* __restrict__ x,
                                                  const float* __restrict__ y, float2* __restrict__ pack) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) pack[i] = make_float2(x ? x[i] : 0.f, y[i]);
}

// ---------------------------------------------------------------------------------------------
// Test RMSE
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_test_sqerr(int64_t nnz, const uint32_t* __restrict__ row,
                                                       const uint32_t* __restrict__ col,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ W, const float* __restrict__ H,
                                                       int64_t rows, int64_t cols, int64_t k, int ifALS,
                                                       double* __restrict__ partials) {
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    for (int64_t q = (int64_t) blockIdx.x * kBlock + threadIdx.x; q < nnz; q += (int64_t) gridDim.x * kBlock) {
        const int64_t i = row[q], j = col[q];
        double pred = 0.0;
        if (ifALS) {
            const float* w = W + i * k;
            const float* h = H + j * k;
            for (int64_t t = 0; t < k; ++t) pred += (double) mul_rn(w[t], h[t]);
        } else {
            for (int64_t t = 0; t < k; ++t) pred += (double) mul_rn(W[t * rows + i], H[t * cols + j]);
        }
        const double err = pred - (double) val[q];
        acc += err * err;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

// *out += sum of x[0..n), one wave, fixed order
__global__ void k_sum_add(uint32_t n, const double* __restrict__ x, double* __restrict__ out) {
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) acc += x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) *out += acc;
}

__global__ __launch_bounds__(kBlock) void k_test_resid_init(int64_t nnz, const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                            const float* __restrict__ val, const float* __restrict__ W,
                                                            const float* __restrict__ H, int64_t rows, int64_t cols, int64_t k,
                                                            float* __restrict__ resid) {
    for (int64_t q = (int64_t) blockIdx.x * kBlock + threadIdx.x; q < nnz; q += (int64_t) gridDim.x * kBlock) {
        const int64_t i = row[q], j = col[q];
        float r = val[q];
        for (int64_t t = 0; t < k; ++t) r = sub_rn(r, mul_rn(W[t * rows + i], H[t * cols + j]));
        resid[q] = r;
    }
}

// calrmse_r1, src/tools.cpp:261-270: the test residual follows one rank's change; per-block sums of its squares
__global__ __launch_bounds__(kBlock) void k_test_r1(int64_t nnz, const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                    float* __restrict__ resid, const float* __restrict__ Wt, const float* __restrict__ Ht,
                                                    const float* __restrict__ oldWt, const float* __restrict__ oldHt,
                                                    double* __restrict__ partials) {
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    for (int64_t q = (int64_t) blockIdx.x * kBlock + threadIdx.x; q < nnz; q += (int64_t) gridDim.x * kBlock) {
        const uint32_t i = row[q], j = col[q];
        const float r = sub_rn(resid[q], sub_rn(mul_rn(Wt[i], Ht[j]), mul_rn(oldWt[i], oldHt[j])));
        resid[q] = r;
        acc += (double) mul_rn(r, r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

__global__ void k_sum_partials(uint32_t n, const double* __restrict__ partials, double* __restrict__ out) {
    // one wave, fixed order: reproducible
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) acc += partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) *out = acc;
}

__global__ __launch_bounds__(kBlock) void k_check_range(uint64_t n, const uint32_t* __restrict__ idx, uint32_t bound,
                                                         unsigned long long* __restrict__ first_bad) {
    unsigned long long bad = ~0ull;
    for (uint64_t q = (uint64_t) blockIdx.x * kBlock + threadIdx.x; q < n; q += (uint64_t) gridDim.x * kBlock)
        if (idx[q] >= bound && q < bad) bad = q;
    if (bad != ~0ull) atomicMin(first_bad, bad);
}

// First / last segment among the REAL entries of every workgroup chunk (the fused finalize's group tables are
// derived from these on the host): the virtual segment holding stored position q is the last v with
// ptr_v[v] <= q (empty ones share their successor's start, so that one is never empty).
__global__ __launch_bounds__(kBlock) void k_chunk_seg_range(uint32_t nchunks, uint64_t chunk_len, const uint32_t* __restrict__ ptr_v,
                                                             size_t nv, uint32_t nseg, const uint32_t* __restrict__ wg_panel,
                                                             const uint32_t* __restrict__ panel_end, uint32_t* __restrict__ seg_first,
                                                             uint32_t* __restrict__ seg_last) {
    const uint32_t w = blockIdx.x * kBlock + threadIdx.x;
    if (w >= nchunks) return;
    const uint64_t lo = (uint64_t) w * chunk_len;
    uint64_t hi = lo + chunk_len;
    const uint32_t pe = panel_end[wg_panel ? wg_panel[w] : 0u];  // (plain layout: one panel)
    if (hi > pe) hi = pe;
    if (hi <= lo) { seg_first[w] = 0xFFFFFFFFu; seg_last[w] = 0; return; }
    auto holder = [&](uint32_t q) {  // last v in [0, nv) with ptr_v[v] <= q
        size_t a = 0, b = nv;        // invariant: ptr_v[a] <= q (ptr_v[0] = 0), answer in [a, b)
        while (b - a > 1) {
            const size_t mid = a + (b - a) / 2;
            if (ptr_v[mid] <= q) a = mid; else b = mid;
        }
        return (uint32_t) (a % nseg);
    };
    seg_first[w] = holder((uint32_t) lo);
    seg_last[w] = holder((uint32_t) (hi - 1));
}

uint32_t seg_grid(uint32_t nseg) {
    // wave-per-segment kernels grid-stride; 8 blocks per CU is plenty
    const uint32_t want = (nseg + (kBlock / 64) - 1) / (kBlock / 64);
    return want < 1 ? 1 : (want > 256 * 8 ? 256 * 8 : want);
}

}  // namespace

#define MFX_LAUNCH_CHECK() MFX_HIP(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel, and one process may
// drive several devices from several threads (mfx_train -nGPUs): remember what was set per
// (instantiation, device), under a lock.
namespace {
struct LdsAttrCache {
    std::mutex m;
    size_t bytes[64] = {};
};
int ensure_dynamic_lds(const void* kernel, LdsAttrCache& c, size_t lds_bytes) {
    int dev = 0;
    MFX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(c.m);
    if (dev >= 0 && dev < 64 && lds_bytes <= c.bytes[dev]) return MFX_OK;
    MFX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
    if (dev >= 0 && dev < 64) c.bytes[dev] = lds_bytes;
    return MFX_OK;
}
}  // namespace

template <int MODE, bool LDS, int BLOCK, bool PSCHK, bool FUSE = false>
int launch_flat_t(const FlatArgs& a, uint32_t grid, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 48 * 1024) {
        static LdsAttrCache cache;  // per instantiation
        MFX_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(k_flat<MODE, LDS, BLOCK, PSCHK, FUSE>), cache, lds_bytes));
    }
    hipLaunchKernelGGL((k_flat<MODE, LDS, BLOCK, PSCHK, FUSE>), dim3(grid), dim3(BLOCK), lds_bytes, st, a);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

template <int MODE, int BLOCK>
int launch_flat_lds(const SegStreamDev& s, const FlatArgs& a, uint32_t grid, size_t lds_bytes, hipStream_t st) {
    if (ModeTraits<MODE>::kPerSeg && s.max_wg_ranks > kPerSegLdsCap)
        return launch_flat_t<MODE, true, BLOCK, true>(a, grid, lds_bytes, st);
    return launch_flat_t<MODE, true, BLOCK, false>(a, grid, lds_bytes, st);
}

template <int MODE>
int launch_flat_mode(const SegStreamDev& s, const FlatArgs& a, hipStream_t st) {
    if (!s.lds_panels)
        return launch_flat_t<MODE, false, kBlock, false>(a, (s.nspans + (kBlock / 64) - 1) / (kBlock / 64), 0, st);
    size_t lds_bytes = (((size_t) s.panel_rows + 1) * sizeof(typename ModeTraits<MODE>::S) + 15) / 16 * 16;
    if (ModeTraits<MODE>::kPerSeg) lds_bytes += (size_t) kPerSegLdsCap * sizeof(typename ModeTraits<MODE>::P);
    const uint32_t grid = s.nspans / s.spans_per_wg;
    switch (s.spans_per_wg) {
        case 4: return launch_flat_lds<MODE, 256>(s, a, grid, lds_bytes, st);
        case 8: return launch_flat_lds<MODE, 512>(s, a, grid, lds_bytes, st);
        case 16: return launch_flat_lds<MODE, 1024>(s, a, grid, lds_bytes, st);
        default: return fail(MFX_ERR_INVALID, "panel layout: spans_per_wg must be 4, 8 or 16 (got %u)", s.spans_per_wg);
    }
}

static FlatArgs flat_args_of(const SegStreamDev& s, const void* gather, const void* perseg, int add) {
    FlatArgs a = {};
    a.idx = s.lds_panels ? static_cast<const void*>(s.idx16) : static_cast<const void*>(s.idx);
    a.val = s.val; a.flags32 = s.flags32; a.hpre = s.hpre; a.seg_of_rank = s.seg_of_rank;
    a.wg_panel = s.wg_panel; a.nspans = s.nspans;
    a.tiles_per_span = s.tiles_per_span; a.panel_rows = s.panel_rows; a.gather_len = s.gather_len; a.nne = s.nne;
    // cache panels: padding sits at every panel's end and gathers a zero, so no tile is masked by position
    a.nnz = (s.panel_rows && !s.lds_panels) ? s.padded_nnz : s.nnz; a.gather = gather; a.perseg = perseg; a.part = s.part;
    a.carry = s.carry; a.add = add;
    return a;
}

static int panel_lanes(const SegStreamDev& s);
uint32_t fused_group_size(uint32_t npanels, uint32_t block) { return block / (npanels >= 8 ? 16u : npanels >= 2 ? 4u : 1u); }

template <int MODE>
int launch_flat_fused_mode(const SegStreamDev& s, const FlatArgs& a, hipStream_t st) {
    if (!s.panel_rows) {  // plain layout: 256-thread workgroups of four spans, no slice; the LDS is the epilogue's scratch only
        const size_t epilogue = 2 * kBlock * sizeof(float) + ((size_t) s.fz_max_chunk_groups + 2) * sizeof(uint32_t);
        return launch_flat_t<MODE, false, kBlock, false, true>(a, (s.nspans + (kBlock / 64) - 1) / (kBlock / 64), epilogue, st);
    }
    size_t lds_bytes = (((size_t) s.panel_rows + 1) * sizeof(typename ModeTraits<MODE>::S) + 15) / 16 * 16 + (size_t) kPerSegLdsCap * sizeof(typename ModeTraits<MODE>::P);
    const size_t epilogue = 2 * 1024 * sizeof(float) + ((size_t) s.fz_max_chunk_groups + 2) * sizeof(uint32_t);
    if (lds_bytes < epilogue) lds_bytes = epilogue;
    const uint32_t grid = s.nspans / s.spans_per_wg;
    if (s.max_wg_ranks > kPerSegLdsCap) return launch_flat_t<MODE, true, 1024, true, true>(a, grid, lds_bytes, st);
    return launch_flat_t<MODE, true, 1024, false, true>(a, grid, lds_bytes, st);
}

int launch_flat_fused(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, const FinalizeArgs& f, hipStream_t st) {
    MFX_REQUIRE(((s.lds_panels && s.spans_per_wg == 16) || !s.panel_rows) && s.fz_order && s.fz_g0 && s.fz_g1 && s.fz_expected && s.fz_arrived,
                "launch_flat_fused: the layout carries no fused-finalize tables");
    MFX_REQUIRE(!f.gh_dense && !f.cnt_override && !f.pack4_as3 && !f.nmf && !f.fundec_seg, "launch_flat_fused: dense / overridden / extended finalize inputs are not fused");
    MFX_REQUIRE(mode == FM_FCSC || mode == FM_FCSR || mode == FM_SWEEP, "launch_flat_fused: bad mode %d", (int) mode);
    FlatArgs a = flat_args_of(s, gather, perseg, 0);
    a.wg_order = s.fz_order; a.wg_g0 = s.fz_g0; a.wg_g1 = s.fz_g1; a.expected = s.fz_expected; a.arrived = s.fz_arrived;
    a.orphans = s.fz_orphans; a.norphans = s.fz_norphans; a.ngroups = s.fz_ngroups; a.panel_lanes = (uint32_t) panel_lanes(s);
    a.nseg = s.nseg; a.npanels = s.npanels; a.ptr_v = s.ptr_v; a.rank_code = s.rank_code; a.seg_cnt = s.seg_cnt;
    a.lambda = f.lambda; a.out_vec = f.out_vec; a.pack2 = f.pack2; a.next_vec = f.next_vec; a.pack4 = f.pack4;
    if (mode == FM_SWEEP) {
        MFX_REQUIRE(!s.panel_rows, "launch_flat_fused: the fused read-only sweep exists for the plain layout only");
        return launch_flat_fused_mode<FM_SWEEP>(s, a, st);
    }
    return mode == FM_FCSC ? launch_flat_fused_mode<FM_FCSC>(s, a, st) : launch_flat_fused_mode<FM_FCSR>(s, a, st);
}

int launch_flat(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, int add,
                hipStream_t st) {
    const FlatArgs a = flat_args_of(s, gather, perseg, add);
    switch (mode) {
        case FM_SWEEP: return launch_flat_mode<FM_SWEEP>(s, a, st);
        case FM_RESID: return launch_flat_mode<FM_RESID>(s, a, st);
        case FM_FCSC: return launch_flat_mode<FM_FCSC>(s, a, st);
        case FM_FCSR: return launch_flat_mode<FM_FCSR>(s, a, st);
        default: return fail(MFX_ERR_INVALID, "launch_flat: bad mode %d", (int) mode);
    }
}

int launch_chunk_seg_range(const SegStreamDev& s, uint32_t wg_spans, const uint32_t* panel_end, uint32_t* seg_first, uint32_t* seg_last, hipStream_t st) {
    const uint32_t nchunks = (s.nspans + wg_spans - 1) / wg_spans;
    if (nchunks == 0) return MFX_OK;
    hipLaunchKernelGGL(k_chunk_seg_range, dim3((nchunks + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nchunks,
                       (uint64_t) s.tiles_per_span * kTileElems * wg_spans, s.ptr_v, (size_t) s.npanels * s.nseg, s.nseg, s.panel_rows ? s.wg_panel : nullptr,
                       panel_end, seg_first, seg_last);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int check_index_range(const uint32_t* d_idx, uint64_t n, uint32_t bound, const char* what, hipStream_t st) {
    if (n == 0) return MFX_OK;
    DevBuf<unsigned long long> flag;
    MFX_TRY(flag.alloc(1));
    MFX_HIP(hipMemsetAsync(flag.get(), 0xFF, sizeof(unsigned long long), st));
    const uint32_t grid = (uint32_t) std::min<uint64_t>((n + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL(k_check_range, dim3(grid), dim3(kBlock), 0, st, n, d_idx, bound, flag.get());
    MFX_LAUNCH_CHECK();
    unsigned long long bad = ~0ull;
    MFX_HIP(hipMemcpyAsync(&bad, flag.get(), sizeof(bad), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    if (bad == ~0ull) return MFX_OK;
    uint32_t v = 0;
    MFX_HIP(hipMemcpy(&v, d_idx + bad, sizeof(v), hipMemcpyDeviceToHost));
    return fail(MFX_ERR_INVALID, "%s %u at position %llu is out of range [0, %u)", what, v, bad, bound);
}

int launch_sweep_wave(const SegStreamDev& s, const float* vec, float* g_dense, float* h_dense, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    MFX_REQUIRE(s.panel_rows == 0, "wave-per-segment kernels need the plain layout");
    hipLaunchKernelGGL(k_sweep_wave, dim3(seg_grid(s.nseg)), dim3(kBlock), 0, st, s.nseg, s.ptr, s.idx, s.val,
                       vec, g_dense, h_dense);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_resid_wave(const SegStreamDev& s, const float* gathered, const float* per_seg, int add, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    MFX_REQUIRE(s.panel_rows == 0, "wave-per-segment kernels need the plain layout");
    hipLaunchKernelGGL(k_resid_wave, dim3(seg_grid(s.nseg)), dim3(kBlock), 0, st, s.nseg, s.ptr, s.idx, s.val,
                       gathered, per_seg, add);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

static GatherPartsArgs parts_of(const SegStreamDev& s) {
    GatherPartsArgs g;
    g.nseg = s.nseg; g.npanels = s.npanels; g.span_len = s.tiles_per_span * kTileElems; g.ptr_v = s.ptr_v;
    g.rank_code = s.rank_code; g.part = s.part; g.carry = s.carry;
    return g;
}

static int panel_lanes(const SegStreamDev& s) { return s.npanels >= 8 ? 16 : s.npanels >= 2 ? 4 : 1; }  // = 1024 / fused_group_size(npanels)

int launch_combine_dense(const SegStreamDev& s, float* gh, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    const GatherPartsArgs a = parts_of(s);
    const int pl = panel_lanes(s);
    const dim3 grid((s.nseg + kBlock / pl - 1) / (kBlock / pl)), block(kBlock);
    if (pl == 16) hipLaunchKernelGGL(k_combine_dense<16>, grid, block, 0, st, a, gh);
    else if (pl == 4) hipLaunchKernelGGL(k_combine_dense<4>, grid, block, 0, st, a, gh);
    else hipLaunchKernelGGL(k_combine_dense<1>, grid, block, 0, st, a, gh);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_finalize(const SegStreamDev& s, const FinalizeArgs& f, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    FinKernelArgs a;
    a.parts = parts_of(s); a.seg_cnt = s.seg_cnt; a.gh_dense = f.gh_dense; a.cnt_override = f.cnt_override;
    a.seg_base = f.gh_len ? f.seg_base : 0u; a.gh_len = f.gh_len ? f.gh_len : s.nseg;
    a.slab_acc = nullptr; a.slab_lo = nullptr; a.slab_bad = nullptr; a.slab_pr = 1;
    if (f.slab_src) {
        MFX_REQUIRE(!f.gh_dense && f.slab_src->scatter && f.slab_src->wgacc && f.slab_src->slab_lo && f.slab_src->scat_slab_bad && f.slab_src->gather_len == s.nseg,
                    "launch_finalize: bad slab source");
        a.slab_acc = f.slab_src->wgacc; a.slab_lo = f.slab_src->slab_lo; a.slab_bad = f.slab_src->scat_slab_bad; a.slab_pr = f.slab_src->panel_rows;
    }
    MFX_REQUIRE(!f.gh_len || ((f.gh_dense || f.slab_src) && (uint64_t) f.seg_base + f.gh_len <= s.nseg), "launch_finalize: bad segment range %u + %u of %u", f.seg_base, f.gh_len, s.nseg);
    a.lambda = f.lambda; a.out_vec = f.out_vec; a.pack2 = f.pack2; a.next_vec = f.next_vec; a.pack4 = f.pack4; a.pack4_as3 = f.pack4_as3; a.nmf = f.nmf; a.fundec_seg = f.fundec_seg;
    const bool dense = f.gh_dense || f.slab_src;
    const int pl = dense ? 1 : panel_lanes(s);
    const uint32_t nfin = dense ? a.gh_len : s.nseg;
    if (nfin == 0) return MFX_OK;
    const dim3 grid((nfin + kBlock / pl - 1) / (kBlock / pl)), block(kBlock);
    if (pl == 16) hipLaunchKernelGGL(k_finalize<16>, grid, block, 0, st, a);
    else if (pl == 4) hipLaunchKernelGGL(k_finalize<4>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_finalize<1>, grid, block, 0, st, a);
    MFX_LAUNCH_CHECK();
    if (f.fundec_seg && f.fundec_sum) {
        hipLaunchKernelGGL(k_sum_add, dim3(1), dim3(64), 0, st, s.nseg, f.fundec_seg, f.fundec_sum);
        MFX_LAUNCH_CHECK();
    }
    return MFX_OK;
}

int launch_unpermute(const SegStreamDev& s, float* out, hipStream_t st) {
    if (s.padded_nnz == 0) return MFX_OK;
    hipLaunchKernelGGL(k_unpermute, dim3((uint32_t) ((s.padded_nnz + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       s.padded_nnz, s.perm, s.val, out);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_unpermute_runs(const SegStreamDev& s, const uint32_t* first_q, const uint32_t* panel_end, float* out, hipStream_t st) {
    const uint32_t nv = s.npanels * s.nseg;
    if (nv == 0) return MFX_OK;
    const uint32_t blocks = std::min<uint32_t>((nv + kBlock / 64 - 1) / (kBlock / 64), 65536u);
    if (s.scatter) hipLaunchKernelGGL(k_unpermute_runs<true>, dim3(blocks), dim3(kBlock), 0, st, nv, s.nseg, s.ptr_v, first_q, panel_end, s.val, out);
    else hipLaunchKernelGGL(k_unpermute_runs<false>, dim3(blocks), dim3(kBlock), 0, st, nv, s.nseg, s.ptr_v, first_q, panel_end, s.val, out);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_pack2(uint32_t n, const float* x, const float* y, float2* pack, hipStream_t st) {
    if (n == 0) return MFX_OK;
    hipLaunchKernelGGL(k_pack2, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, x, y, pack);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_test_sqerr(int64_t nnz_test, const uint32_t* row, const uint32_t* col, const float* val,
                      const float* W, const float* H, int64_t rows, int64_t cols, int64_t k, int ifALS,
                      double* block_partials, uint32_t nblocks, double* sum_out, hipStream_t st) {
    hipLaunchKernelGGL(k_test_sqerr, dim3(nblocks), dim3(kBlock), 0, st, nnz_test, row, col, val, W, H, rows,
                       cols, k, ifALS, block_partials);
    MFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, nblocks, block_partials, sum_out);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_test_resid_init(int64_t nnz_test, const uint32_t* row, const uint32_t* col, const float* val, const float* W, const float* H,
                           int64_t rows, int64_t cols, int64_t k, float* resid, hipStream_t st) {
    if (nnz_test <= 0) return MFX_OK;
    const uint32_t grid = (uint32_t) std::min<int64_t>((nnz_test + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL(k_test_resid_init, dim3(grid), dim3(kBlock), 0, st, nnz_test, row, col, val, W, H, rows, cols, k, resid);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_test_r1(int64_t nnz_test, const uint32_t* row, const uint32_t* col, float* resid, const float* Wt, const float* Ht,
                   const float* oldWt, const float* oldHt, double* block_partials, uint32_t nblocks, double* sum_out, hipStream_t st) {
    hipLaunchKernelGGL(k_test_r1, dim3(nblocks), dim3(kBlock), 0, st, nnz_test, row, col, resid, Wt, Ht, oldWt, oldHt, block_partials);
    MFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, nblocks, block_partials, sum_out);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_seg_owner(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, const FinalizeArgs& f, hipStream_t st) {
    MFX_REQUIRE(s.panel_rows == 0 && !s.scatter && s.ptr && s.idx && s.own_short && s.own_long, "launch_seg_owner: needs the plain layout with its owner lists");
    MFX_REQUIRE(!f.gh_dense && !f.cnt_override && !f.pack4_as3 && !f.nmf && !f.fundec_seg, "launch_seg_owner: dense / overridden / extended finalize inputs are not supported");
    MFX_REQUIRE(mode == FM_SWEEP || ((mode == FM_FCSC || mode == FM_FCSR) && f.pack2 && f.next_vec && f.pack2 == perseg), "launch_seg_owner: bad mode / packs");
    if (s.nseg == 0) return MFX_OK;
    SegOwnerArgs a;
    a.ptr = s.ptr; a.idx = s.idx; a.val = s.val; a.gather = gather; a.perseg = perseg;
    a.long_list = reinterpret_cast<const uint4*>(s.own_long); a.short_list = reinterpret_cast<const uint4*>(s.own_short); a.nlong = s.own_nlong; a.nshort = s.own_nshort;
    a.lambda = f.lambda; a.out_vec = f.out_vec; a.pack2 = f.pack2; a.next_vec = f.next_vec; a.pack4 = f.pack4;
    const uint32_t short_wgs = std::min<uint32_t>((s.own_nshort + kSegBlock / 64 - 1) / (kSegBlock / 64), 4096u);
    const dim3 grid(s.own_nlong + short_wgs), block(kSegBlock);
    if (grid.x == 0) return MFX_OK;
    switch (mode) {
        case FM_SWEEP: hipLaunchKernelGGL(k_seg_owner<FM_SWEEP>, grid, block, 0, st, a); break;
        case FM_FCSC: hipLaunchKernelGGL(k_seg_owner<FM_FCSC>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(k_seg_owner<FM_FCSR>, grid, block, 0, st, a); break;
    }
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}
uint32_t seg_owner_long_threshold() {
    if (const char* e = std::getenv("MFX_OWNER_LONG")) { const int v = std::atoi(e); if (v > 0) return (uint32_t) v; }  // (A/B)
    return kSegOwnerLong;
}

}  // namespace mfx

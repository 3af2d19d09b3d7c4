// ccd_kernels.hip -- CCD++ device kernels for gfx950 (CDNA4, wave64).
//
// Every kernel here is HBM-bound integer/fp32 streaming work (0.25-0.5 flop/byte): no MFMA, no
// GEMM reshaping.  What matters is (1) 16-byte coalesced walks over idx/val, (2) keeping the
// gathered factor vectors cache-resident (non-temporal hints on the streams), (3) nnz-balanced
// work so one 230k-entry column costs the same per wave as 230k entries of short rows, and
// (4) no atomics: every output has exactly one writer, so results are bitwise reproducible.
#include "ccd_kernels.hpp"

#include "flat_layout.hpp"

namespace mfx {
namespace {

constexpr int kBlock = 256;  // 4 wavefronts

// native vector types: the non-temporal builtins reject HIP's struct-based uint4/float4
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Unfused multiply-then-add/sub, like the reference CPU build (no FMA contraction), so that the
// residual update is bit-identical to src/CCD.cpp:25,36 given identical operands.
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

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
// non-zeros; lane l of a tile owns elements 4l..4l+3 (one 16-byte load each of idx and val).
// ---------------------------------------------------------------------------------------------
struct FlatArgs {
    const uint32_t* idx;
    float* val;
    const uint64_t* flags;
    const uint32_t* seg_of_rank;
    const uint32_t* span_rank_base;
    uint32_t nspans;
    uint32_t tiles_per_span;
    uint64_t nnz;
    const void* gather;
    const void* perseg;
    float* gpart;
    float* hpart;
    float* carry_g;
    float* carry_h;
    int add;
};

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<FM_SWEEP> { using G = float;  using P = float;  static constexpr bool kPerSeg = false, kWrite = false, kDot = true; };
template <> struct ModeTraits<FM_RESID> { using G = float;  using P = float;  static constexpr bool kPerSeg = true,  kWrite = true,  kDot = false; };
template <> struct ModeTraits<FM_FCSC>  { using G = float2; using P = float2; static constexpr bool kPerSeg = true,  kWrite = true,  kDot = true; };
template <> struct ModeTraits<FM_FCSR>  { using G = float4; using P = float2; static constexpr bool kPerSeg = true,  kWrite = true,  kDot = true; };

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

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_flat(FlatArgs a) {
    using TR = ModeTraits<MODE>;
    using G = typename TR::G;
    using P = typename TR::P;
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform values are forced into SGPRs so that flag words / span metadata become scalar loads
    const uint32_t span = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    if (span >= a.nspans) return;
    const G* __restrict__ gather = static_cast<const G*>(a.gather);
    const P* __restrict__ perseg = static_cast<const P*>(a.perseg);
    const uint64_t start = (uint64_t) span * a.tiles_per_span * kTileElems;
    const int32_t rank_base = (int32_t) a.span_rank_base[span];
    int32_t cur = rank_base - 1;  // rank of the segment that is open at the current position
    P pcur{};
    if constexpr (TR::kPerSeg) {
        if (cur >= 0) pcur = perseg[a.seg_of_rank[cur]];
    }
    float og = 0.f, oh = 0.f;  // per-lane sums of the open segment since its last head

    const u32x4* __restrict__ idx4 = reinterpret_cast<const u32x4*>(a.idx + start) + lane;
    f32x4* __restrict__ val4 = reinterpret_cast<f32x4*>(a.val + start) + lane;
    uint32_t ntiles = a.tiles_per_span;
    {   // tiles that start at or beyond nnz hold only padding
        const uint64_t left = a.nnz > start ? a.nnz - start : 0;
        const uint32_t live = (uint32_t) ((left + kTileElems - 1) / kTileElems);
        if (live < ntiles) ntiles = live;
    }
    u32x4 id_n = {0, 0, 0, 0};
    f32x4 v_n = {0.f, 0.f, 0.f, 0.f};
    if (ntiles) {
        id_n = __builtin_nontemporal_load(idx4);
        v_n = __builtin_nontemporal_load(val4);
    }
    for (uint32_t tile = 0; tile < ntiles; ++tile) {
        const u32x4 id = id_n;
        const f32x4 v = v_n;
        if (tile + 1 < ntiles) {  // prefetch the next tile's streams (one tile ahead)
            id_n = __builtin_nontemporal_load(idx4 + (tile + 1) * 64);
            v_n = __builtin_nontemporal_load(val4 + (tile + 1) * 64);
        }
        const uint64_t base = start + (uint64_t) tile * kTileElems;
        const uint64_t* fw = a.flags + (base >> 6);
        const uint64_t w0 = fw[0], w1 = fw[1], w2 = fw[2], w3 = fw[3];
        const bool partial = base + kTileElems > a.nnz;  // wave-uniform: tile straddles the end
        const uint64_t e0 = base + (uint64_t) lane * 4;

        const uint32_t ids[4] = {id.x, id.y, id.z, id.w};
        const float vs[4] = {v.x, v.y, v.z, v.w};
        G ga[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ga[e] = gather[ids[e]];
        float vo[4], gc[4], hc[4];

        if ((w0 | w1 | w2 | w3) == 0) {
            // ---- no segment starts in this tile: everything belongs to the open segment ----
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                element_op<MODE>(vs[e], ga[e], pcur, a.add, vo[e], gc[e], hc[e]);
                if (partial && e0 + e >= a.nnz) { gc[e] = 0.f; hc[e] = 0.f; }
                if constexpr (TR::kDot) { og += gc[e]; oh += hc[e]; }
            }
        } else {
            // ---- segmented tile ----
            const uint32_t wsel = lane >> 4;
            const uint64_t myw = wsel == 0 ? w0 : wsel == 1 ? w1 : wsel == 2 ? w2 : w3;
            const uint32_t c0 = __popcll(w0), c1 = c0 + __popcll(w1), c2 = c1 + __popcll(w2);
            const uint32_t ctot = c2 + __popcll(w3);
            const uint32_t below = wsel == 0 ? 0u : wsel == 1 ? c0 : wsel == 2 ? c1 : c2;
            const uint32_t sh = (lane & 15) * 4;
            const uint32_t nib = (uint32_t) (myw >> sh) & 0xFu;
            // heads in the tile before this lane's first element
            const uint32_t before = below + __popcll(myw & ((uint64_t(1) << sh) - 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                P ps{};
                if constexpr (TR::kPerSeg) {
                    const int32_t r = cur + (int32_t) before + (int32_t) __popc(nib & ((2u << e) - 1u));
                    ps = perseg[a.seg_of_rank[r]];
                }
                element_op<MODE>(vs[e], ga[e], ps, a.add, vo[e], gc[e], hc[e]);
                if (partial && e0 + e >= a.nnz) { gc[e] = 0.f; hc[e] = 0.f; }
            }
            if constexpr (TR::kDot) {
                // carry-in: the open segment's sum so far, as a wave-uniform value
                const float cin_g = wave_sum(og), cin_h = wave_sum(oh);
                // lane-serial pass: fg/fh = run before the lane's first head, ag/ah = run after its
                // last head; runs between two heads of the same lane are complete segments.
                float ag = 0.f, ah = 0.f, fg = 0.f, fh = 0.f;
                bool seen = false;
                int32_t r_close = cur + (int32_t) before;  // rank closed by the lane's next head
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((nib >> e) & 1u) {
                        if (!seen) {
                            fg = ag; fh = ah; seen = true;
                        } else {  // head..head inside one lane: started in this span by construction
                            a.gpart[r_close] = ag;
                            a.hpart[r_close] = ah;
                        }
                        ++r_close;
                        ag = 0.f; ah = 0.f;
                    }
                    ag += gc[e];
                    ah += hc[e];
                }
                if (!seen) { fg = ag; fh = ah; }
                // segmented inclusive scan over lanes of the right-propagating value
                const uint64_t M = __ballot(seen);
                float xg = ag, xh = ah;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const float yg = __shfl_up(xg, d, 64), yh = __shfl_up(xh, d, 64);
                    // add lanes [l-2d+1, l-d] only if no head lane lies in (l-d, l]
                    const uint32_t lo_bit = lane >= (uint32_t) d ? lane - d + 1 : 0u;
                    const bool ok = lane >= (uint32_t) d && ((M >> lo_bit) & ((uint64_t(1) << d) - 1)) == 0;
                    if (ok) { xg += yg; xh += yh; }
                }
                float eg = __shfl_up(xg, 1, 64), eh = __shfl_up(xh, 1, 64);
                if (lane == 0) { eg = 0.f; eh = 0.f; }
                if ((M & ((uint64_t(1) << lane) - 1)) == 0) { eg += cin_g; eh += cin_h; }
                if (seen) {  // this lane's first head closes the segment of rank cur+before
                    const int32_t rc = cur + (int32_t) before;
                    const float tg = eg + fg, th = eh + fh;
                    if (rc >= rank_base) {
                        a.gpart[rc] = tg;
                        a.hpart[rc] = th;
                    } else {  // it started in an earlier span: this is the span's head carry
                        a.carry_g[span] = tg;
                        a.carry_h[span] = th;
                    }
                }
                // new open segment: everything after the tile's last head
                const float ng = __shfl(xg, 63, 64), nh = __shfl(xh, 63, 64);
                og = lane == 0 ? ng : 0.f;
                oh = lane == 0 ? nh : 0.f;
            }
            cur += (int32_t) ctot;
            if constexpr (TR::kPerSeg) pcur = perseg[a.seg_of_rank[cur]];
        }
        if constexpr (TR::kWrite) {
            if (!partial) {
                __builtin_nontemporal_store(f32x4{vo[0], vo[1], vo[2], vo[3]}, val4 + tile * 64);
            } else {
                float* vp = a.val + e0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e0 + e < a.nnz) vp[e] = vo[e];
            }
        }
    }
    if constexpr (TR::kDot) {
        const float tg = wave_sum(og), th = wave_sum(oh);
        if (lane == 0) {
            if (cur >= rank_base) {  // the open segment's head lies in this span: we own its slot
                a.gpart[cur] = tg;
                a.hpart[cur] = th;
            } else {  // the whole span is interior to one segment
                a.carry_g[span] = tg;
                a.carry_h[span] = th;
            }
        }
    }
}

// Adds, in span order, the carries that belong to segment [lo, hi).
__device__ __forceinline__ void add_carries(uint32_t lo, uint32_t hi, uint32_t span_len,
                                            const float* __restrict__ cg, const float* __restrict__ ch,
                                            float& g, float& h) {
    uint32_t s = lo / span_len + 1;
    const uint32_t s_end = (hi - 1) / span_len;  // inclusive
    for (; s + 3 <= s_end; s += 4) {  // 4 independent loads in flight, sequential adds
        const float g0 = cg[s], g1 = cg[s + 1], g2 = cg[s + 2], g3 = cg[s + 3];
        const float h0 = ch[s], h1 = ch[s + 1], h2 = ch[s + 2], h3 = ch[s + 3];
        g += g0; g += g1; g += g2; g += g3;
        h += h0; h += h1; h += h2; h += h3;
    }
    for (; s <= s_end; ++s) { g += cg[s]; h += ch[s]; }
}

__global__ __launch_bounds__(kBlock) void k_combine_dense(uint32_t nseg, uint32_t span_len,
                                                          const uint32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ rank_of_seg,
                                                          const float* __restrict__ gpart,
                                                          const float* __restrict__ hpart,
                                                          const float* __restrict__ cg,
                                                          const float* __restrict__ ch,
                                                          float* __restrict__ gh) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= nseg) return;
    float g = 0.f, h = 0.f;
    const int32_t r = rank_of_seg[c];
    if (r >= 0) {
        g = gpart[r];
        h = hpart[r];
        add_carries(ptr[c], ptr[c + 1], span_len, cg, ch, g, h);
    }
    gh[c] = g;
    gh[nseg + c] = h;
}

struct FinKernelArgs {
    uint32_t nseg, span_len;
    const uint32_t* ptr;
    const int32_t* rank_of_seg;
    const float *gpart, *hpart, *cg, *ch;
    const float* gh_dense;
    const uint32_t* cnt_override;
    float lambda;
    float* out_vec;
    float2* pack2;
    const float* next_vec;
    float4* pack4;
};

__global__ __launch_bounds__(kBlock) void k_finalize(FinKernelArgs a) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= a.nseg) return;
    const uint32_t lo = a.ptr[c], hi = a.ptr[c + 1];
    float g = 0.f, h = 0.f;
    if (a.gh_dense) {
        g = a.gh_dense[c];
        h = a.gh_dense[a.nseg + c];
    } else {
        const int32_t r = a.rank_of_seg[c];
        if (r >= 0) {
            g = a.gpart[r];
            h = a.hpart[r];
            add_carries(lo, hi, a.span_len, a.cg, a.ch, g, h);
        }
    }
    const uint32_t cnt = a.cnt_override ? a.cnt_override[c] : hi - lo;
    // reference: g / (lambda * |Omega| + sum u^2), 0 for an empty segment (src/CCD.cpp:6-16,112)
    const float x = cnt ? g / add_rn(mul_rn(a.lambda, (float) cnt), h) : 0.f;
    a.out_vec[c] = x;
    if (a.pack2) {
        const float2 old = a.pack2[c];
        if (a.pack4) a.pack4[c] = make_float4(old.x, old.y, x, 0.f);
        a.pack2[c] = make_float2(x, a.next_vec[c]);
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

__global__ void k_sum_partials(uint32_t n, const double* __restrict__ partials, double* __restrict__ out) {
    // one wave, fixed order: reproducible
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) acc += partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) *out = acc;
}

uint32_t seg_grid(uint32_t nseg) {
    // wave-per-segment kernels grid-stride; 8 blocks per CU is plenty
    const uint32_t want = (nseg + (kBlock / 64) - 1) / (kBlock / 64);
    return want < 1 ? 1 : (want > 256 * 8 ? 256 * 8 : want);
}

}  // namespace

#define MFX_LAUNCH_CHECK() MFX_HIP(hipGetLastError())

int launch_flat(FlatMode mode, const SegStreamDev& s, const void* gather, const void* perseg, int add,
                hipStream_t st) {
    FlatArgs a;
    a.idx = s.idx; a.val = s.val; a.flags = s.flags; a.seg_of_rank = s.seg_of_rank;
    a.span_rank_base = s.span_rank_base; a.nspans = s.nspans; a.tiles_per_span = s.tiles_per_span;
    a.nnz = s.nnz; a.gather = gather; a.perseg = perseg; a.gpart = s.gpart; a.hpart = s.hpart;
    a.carry_g = s.carry_g; a.carry_h = s.carry_h; a.add = add;
    const dim3 grid((s.nspans + (kBlock / 64) - 1) / (kBlock / 64)), block(kBlock);
    switch (mode) {
        case FM_SWEEP: hipLaunchKernelGGL(k_flat<FM_SWEEP>, grid, block, 0, st, a); break;
        case FM_RESID: hipLaunchKernelGGL(k_flat<FM_RESID>, grid, block, 0, st, a); break;
        case FM_FCSC: hipLaunchKernelGGL(k_flat<FM_FCSC>, grid, block, 0, st, a); break;
        case FM_FCSR: hipLaunchKernelGGL(k_flat<FM_FCSR>, grid, block, 0, st, a); break;
        default: return fail(MFX_ERR_INVALID, "launch_flat: bad mode %d", (int) mode);
    }
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_sweep_wave(const SegStreamDev& s, const float* vec, float* g_dense, float* h_dense, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    hipLaunchKernelGGL(k_sweep_wave, dim3(seg_grid(s.nseg)), dim3(kBlock), 0, st, s.nseg, s.ptr, s.idx, s.val,
                       vec, g_dense, h_dense);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_resid_wave(const SegStreamDev& s, const float* gathered, const float* per_seg, int add, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    hipLaunchKernelGGL(k_resid_wave, dim3(seg_grid(s.nseg)), dim3(kBlock), 0, st, s.nseg, s.ptr, s.idx, s.val,
                       gathered, per_seg, add);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_combine_dense(const SegStreamDev& s, float* gh, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    hipLaunchKernelGGL(k_combine_dense, dim3((s.nseg + kBlock - 1) / kBlock), dim3(kBlock), 0, st, s.nseg,
                       s.tiles_per_span * kTileElems, s.ptr, s.rank_of_seg, s.gpart, s.hpart, s.carry_g,
                       s.carry_h, gh);
    MFX_LAUNCH_CHECK();
    return MFX_OK;
}

int launch_finalize(const SegStreamDev& s, const FinalizeArgs& f, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    FinKernelArgs a;
    a.nseg = s.nseg; a.span_len = s.tiles_per_span * kTileElems; a.ptr = s.ptr; a.rank_of_seg = s.rank_of_seg;
    a.gpart = s.gpart; a.hpart = s.hpart; a.cg = s.carry_g; a.ch = s.carry_h; a.gh_dense = f.gh_dense;
    a.cnt_override = f.cnt_override; a.lambda = f.lambda; a.out_vec = f.out_vec; a.pack2 = f.pack2;
    a.next_vec = f.next_vec; a.pack4 = f.pack4;
    hipLaunchKernelGGL(k_finalize, dim3((s.nseg + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a);
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

}  // namespace mfx

// ccd_reforder.hip -- the rank-one sweep of CCD++ in the REFERENCE'S OWN SUMMATION ORDER
// (mfx_params.schedule = 0, kernel_variant = -1; mfx_rank_one_sweep(..., variant = -1)).
//
// The reference adds a column's terms strictly left to right in fp32, on the CPU
// (RankOneUpdate_Original_float, src/CCD.cpp:6-16) and on the GPU (RankOneUpdate_dev, one thread per
// column, cuda_src/CCD_CUDA.cu:3-22):
//     g = 0; h = lambda * |Omega_c|;  for p in column c, ascending:  g += u[i_p] * r_p;  h += u[i_p] * u[i_p];
//     v_c = g / h                      (0 for an empty column)
// Every other sweep kernel of this library reduces in a tree (k_flat), lane-strided (k_sweep_wave) or in
// fixed point (k_scatter): same value up to the summation order, never the same bits.  This file is the
// parity mode that IS the same bits -- the CCD++ counterpart of als_exact.hip -- so that "the product
// path deviates from the reference only by summation order" can be shown against the oracle instead of
// argued through float64 (tests/test_gpu_fullsize.py).
//
// Shape: one wavefront per segment.  The 64 lanes load 64 consecutive entries (coalesced), form the two
// products in parallel -- each is a single rounded fp32 multiply in the reference as well -- and the running
// sums then travel lane to lane on the DPP path: step j computes, in every lane l,
//     s_l <- s_{l-1} + x_l            (one v_add_f32_dpp wave_shr:1; the incoming carry rides in lane 0's term)
// and after l + 1 steps lane l holds ((carry + x_0) + x_1) + ... + x_l, the reference's own chain.  The chain
// is inherently serial (fp32 addition is not associative): one dependent add per entry and sum, ~12 clocks
// per entry for g and h together, i.e. ~1.2 ms for a 237 k-entry column.  Segments are dispatched longest
// first so that such a column starts at once and the short ones fill in behind it.  A parity mode, not a
// fast path: no LDS panels, the gather goes to L2.
#include "ccd_kernels.hpp"

#include <algorithm>
#include <numeric>

namespace mfx {
namespace {

constexpr int kRefBlock = 256;                 // 4 wavefronts, one segment each at a time
constexpr int kDppWaveShr1 = 0x138;
constexpr uint32_t kStage = 256;               // entries per pipeline stage: 4 per lane

__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// lane l <- value of lane l - 1; lane 0 <- 0.  (bound_ctrl:0 with a zero `old`: the form the DPP combiner folds into
// the consuming add -- one v_add_f32_dpp per step; with `old` = carry every step paid a v_mov and a hazard nop more.)
__device__ __forceinline__ float shr1(float s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), kDppWaveShr1, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float x, uint32_t lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// n (wave-uniform, 1..64) leading lanes hold terms: on return (cg, ch) are the running sums behind term n - 1.
// The carry enters through lane 0's term: q_0 = carry + x_0 (rounded once, as in the reference), q_l = x_l; a step is
// s_l <- s_{l-1} + q_l with s_{-1} = 0, so lane 0 holds 0 + q_0 = q_0 from the first step on (a running sum that
// started at +0 or at lambda * count >= 0 is never -0, the one value 0 + q would not reproduce).
template <bool FULL>
__device__ __forceinline__ void chain64(float pg, float ph, uint32_t n, uint32_t lane, float& cg, float& ch) {
    const float qg = lane == 0 ? add_rn(cg, pg) : pg;
    const float qh = lane == 0 ? add_rn(ch, ph) : ph;
    float sg = qg, sh = qh;
    if constexpr (FULL) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            sg = add_rn(shr1(sg), qg);
            sh = add_rn(shr1(sh), qh);
        }
        cg = lane_value(sg, 63);
        ch = lane_value(sh, 63);
    } else {
        for (uint32_t j = 0; j < n; ++j) {
            sg = add_rn(shr1(sg), qg);
            sh = add_rn(shr1(sh), qh);
        }
        cg = lane_value(sg, n - 1);
        ch = lane_value(sh, n - 1);
    }
}

struct Stage {
    uint32_t i[4];
    float r[4];
};

__device__ __forceinline__ void load_stage(Stage& s, const uint32_t* __restrict__ idx, const float* __restrict__ val,
                                           uint32_t base, uint32_t hi, uint32_t lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p = base + 64u * j + lane;
        const bool ok = p < hi && p >= base;  // (p >= base: no wrap at the top of the 32-bit range)
        s.i[j] = ok ? idx[p] : 0u;
        s.r[j] = ok ? val[p] : 0.f;
    }
}

__global__ __launch_bounds__(kRefBlock) void k_sweep_ref(uint32_t nseg, const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx,
                                                         const float* __restrict__ val, const float* __restrict__ vec,
                                                         float lambda, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kRefBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kRefBlock) >> 6;
    for (uint32_t w = wave; w < nseg; w += nwaves) {
        const uint32_t c = order ? order[w] : w;
        const uint32_t lo = __builtin_amdgcn_readfirstlane(ptr[c]), hi = __builtin_amdgcn_readfirstlane(ptr[c + 1]);
        if (lo >= hi) {  // empty: 0 (src/CCD.cpp:8)
            if (lane == 0) out[c] = 0.f;
            continue;
        }
        float cg = 0.f;
        float ch = mul_rn(lambda, (float) (hi - lo));  // float * unsigned (src/CCD.cpp:112,120)
        // three stages in flight: index / value loads of stage t + 2, the gather of stage t + 1, the chain of stage t
        Stage a, b;
        float xa[4], xb[4];
        load_stage(a, idx, val, lo, hi, lane);
        load_stage(b, idx, val, lo + kStage, hi, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = vec[a.i[j]];
        for (uint32_t base = lo; base < hi && base >= lo; base += kStage) {
            Stage c2;
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[j] = vec[b.i[j]];
            load_stage(c2, idx, val, base + 2 * kStage, hi, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t first = base + 64u * j;
                if (first >= hi || first < base) break;
                const uint32_t n = hi - first;
                const float pg = mul_rn(xa[j], a.r[j]);
                const float ph = mul_rn(xa[j], xa[j]);
                if (n >= 64) chain64<true>(pg, ph, 64, lane, cg, ch);
                else chain64<false>(pg, ph, n, lane, cg, ch);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { xa[j] = xb[j]; a.r[j] = b.r[j]; }
            b = c2;
        }
        if (lane == 0) out[c] = cg / ch;  // correctly rounded fp32 division, as on the host
    }
}

}  // namespace

// Segments by descending length (ties: ascending id): the dispatch order of k_sweep_ref.
void ref_sweep_order(const uint32_t* ptr_host, uint32_t nseg, std::vector<uint32_t>* order) {
    order->resize(nseg);
    std::iota(order->begin(), order->end(), 0u);
    std::stable_sort(order->begin(), order->end(), [&](uint32_t x, uint32_t y) {
        return ptr_host[x + 1] - ptr_host[x] > ptr_host[y + 1] - ptr_host[y];
    });
}

int launch_sweep_ref(const SegStreamDev& s, const uint32_t* order, const float* vec, float lambda, float* out, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    MFX_REQUIRE(s.panel_rows == 0 && s.ptr && (s.nnz == 0 || (s.idx && s.val)), "the reference-order sweep needs the plain layout");
    const uint32_t waves_per_block = kRefBlock / 64;
    const uint32_t grid = std::min<uint32_t>((s.nseg + waves_per_block - 1) / waves_per_block, 256u * 8u);
    hipLaunchKernelGGL(k_sweep_ref, dim3(grid), dim3(kRefBlock), 0, st, s.nseg, order, s.ptr, s.idx, s.val, vec, lambda, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MFX_ERR_HIP, "k_sweep_ref launch failed: %s", hipGetErrorString(e));
    return MFX_OK;
}

}  // namespace mfx

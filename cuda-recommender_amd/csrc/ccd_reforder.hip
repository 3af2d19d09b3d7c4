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
#include <cstdlib>
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

// One segment, one wavefront: the lane-to-lane DPP chain.
__device__ __forceinline__ void ref_segment_dpp(uint32_t c, const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx,
                                                const float* __restrict__ val, const float* __restrict__ vec, float lambda,
                                                float* __restrict__ out, uint32_t lane) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(ptr[c]), hi = __builtin_amdgcn_readfirstlane(ptr[c + 1]);
    if (lo >= hi) {  // empty: 0 (src/CCD.cpp:8)
        if (lane == 0) out[c] = 0.f;
        return;
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

__global__ __launch_bounds__(kRefBlock) void k_sweep_ref(uint32_t nseg, const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx,
                                                         const float* __restrict__ val, const float* __restrict__ vec,
                                                         float lambda, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kRefBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kRefBlock) >> 6;
    for (uint32_t w = wave; w < nseg; w += nwaves) ref_segment_dpp(order ? order[w] : w, ptr, idx, val, vec, lambda, out, lane);
}

// ---------------------------------------------------------------------------------------------
// (r4) The same chain at ~5 clocks per entry instead of ~12.  tools/ubench_chain.hip: a dependent v_add_f32_dpp wave_shr:1 -- the
// lane-to-lane step above -- has a 12.4-clock dependent-issue latency, and a second interleaved chain rides in it for free (g only:
// 1227 us, g and h: 1276 us per sweep at the Netflix shape), whereas a PLAIN dependent v_add_f32 issues every 4.6 clocks.  So the
// running sum stays in ONE register per lane and the terms come to it: a wave writes the 256 products of a stage to LDS in entry
// order and reads them back as broadcast ds_read_b128 (all lanes the same four terms), every lane carrying the same sum
// redundantly.  Two plain chains in one wave would be issue-bound at 8.5 clocks per entry, so g and h get a WAVE EACH (a 128-thread
// workgroup per segment; both waves load and gather for themselves, they only meet for the division).  Terms past a segment's end
// are +0: s + 0 == s bit for bit (a running sum that started at +0 or lambda * count >= 0 is never -0), so there are no tail masks.
// Measured (tools/exp_ref_single.sh, exp_ref.sh): one 240 000-entry column 1845 -> 1186 us (in the kernel the broadcast reads cost
// issue slots too: 0.64 of the lane-to-lane form, not the microbenchmark's 0.37); v-sweep of the Netflix shape 1.93 -> 1.32 ms.
// ---------------------------------------------------------------------------------------------
constexpr int kRef2Block = 128;
constexpr uint32_t kRefLong = 32768;           // entries from which a segment takes the two-wave plain-add form (MFX_REF_LONG overrides, A/B):
                                               // Netflix shape, v-sweep 1.32 ms with 32768 against 1.39-1.49 ms with 8192 ... 1 and 1.93 ms without
constexpr uint32_t kStage2 = 1024;             // entries per pipeline stage of k_sweep_ref2: 16 per lane (a stage's chain takes ~2 us:
                                               // two stages of loads in flight cover the memory latency; with 256-entry stages they did not)
using f32x4r = __attribute__((ext_vector_type(4))) float;
struct Stage2 {
    uint32_t i[16];
    float r[16];
};
// branch-free: positions past the end are clamped re-reads of the segment's last entry (their terms are replaced by +0 below)
__device__ __forceinline__ void load_stage2(Stage2& s, const uint32_t* __restrict__ idx, const float* __restrict__ val,
                                            uint32_t base, uint32_t hi, uint32_t lane) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t p = base + 64u * j + lane;
        p = (p < hi && p >= base) ? p : hi - 1;
        s.i[j] = idx[p];
        s.r[j] = val[p];
    }
}

// `order` lists the segments longest first; the first nlong of them (>= kRefLong entries) get a workgroup each and the plain-add
// chains, the others a wavefront each and the lane-to-lane chain above (for a 200-entry row the two-wave form costs more than it
// saves: two barriers, a 1024-entry stage mostly of padding -- u-sweep of the Netflix shape 610 -> 1250 us when it took all rows).
__global__ __launch_bounds__(kRef2Block) void k_sweep_ref2(uint32_t nseg, uint32_t nlong, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx,
                                                           const float* __restrict__ val, const float* __restrict__ vec,
                                                           float lambda, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float terms[2][2][kStage2];  // [wave][buffer][entry of the stage]
    __shared__ float g_slot;
    const uint32_t lane = threadIdx.x & 63, role = threadIdx.x >> 6;  // role 0: the g chain, role 1: the h chain
    if (blockIdx.x >= nlong) {  // ---- short segments: one per wavefront, grid-strided
        const uint32_t nwaves = (gridDim.x - nlong) * (kRef2Block / 64);
        for (uint32_t w = nlong + (blockIdx.x - nlong) * (kRef2Block / 64) + role; w < nseg; w += nwaves)
            ref_segment_dpp(order[w], ptr, idx, val, vec, lambda, out, lane);
        return;
    }
    for (uint32_t w = blockIdx.x; w < nlong; w += gridDim.x) {  // (one iteration: the grid holds a workgroup per long segment)
        const uint32_t c = order[w];
        const uint32_t lo = __builtin_amdgcn_readfirstlane(ptr[c]), hi = __builtin_amdgcn_readfirstlane(ptr[c + 1]);
        float s = role == 0 ? 0.f : mul_rn(lambda, (float) (hi - lo));  // float * unsigned (src/CCD.cpp:112,120)
        if (hi > lo) {
            // stages in flight: index / value loads of stage t + 2, the gather of stage t + 1, products + chain of stage t
            Stage2 a, b;
            float xa[16], xb[16];
            load_stage2(a, idx, val, lo, hi, lane);
            if (hi - lo > kStage2) load_stage2(b, idx, val, lo + kStage2, hi, lane);
#pragma unroll
            for (int j = 0; j < 16; ++j) xa[j] = vec[a.i[j]];
            uint32_t buf = 0;
            for (uint32_t base = lo; base < hi && base >= lo; base += kStage2, buf ^= 1u) {
                const uint32_t left = hi - base;  // entries from this stage on (wave-uniform)
                Stage2 c2;
                if (left > kStage2) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) xb[j] = vec[b.i[j]];
                    if (left > 2 * kStage2) load_stage2(c2, idx, val, base + 2 * kStage2, hi, lane);
                }
                float* t = terms[role][buf];
                const uint32_t nslots = left >= kStage2 ? 16u : (left + 63) / 64;  // 64-entry rows of the stage that hold entries
#pragma unroll
                for (int j = 0; j < 16; ++j)  // entry base + 64 j + lane; past the end: +0
                    if ((uint32_t) j < nslots) t[64 * j + lane] = (64u * j + lane < left) ? (role == 0 ? mul_rn(xa[j], a.r[j]) : mul_rn(xa[j], xa[j])) : 0.f;
                const f32x4r* tq = reinterpret_cast<const f32x4r*>(t);
                // (the wave's own ds_writes are ordered before its ds_reads: LDS is in order per wave)
                // 16 terms per round; the NEXT round's four broadcast reads are issued before this round's sixteen dependent adds (an
                // LDS round trip is ~100 clocks).  Rows of 64 slots are written whole (+0 past the end), so whole rounds are safe.
                auto round16 = [&](const f32x4r& x0, const f32x4r& x1, const f32x4r& x2, const f32x4r& x3) {
                    s = add_rn(s, x0[0]); s = add_rn(s, x0[1]); s = add_rn(s, x0[2]); s = add_rn(s, x0[3]);
                    s = add_rn(s, x1[0]); s = add_rn(s, x1[1]); s = add_rn(s, x1[2]); s = add_rn(s, x1[3]);
                    s = add_rn(s, x2[0]); s = add_rn(s, x2[1]); s = add_rn(s, x2[2]); s = add_rn(s, x2[3]);
                    s = add_rn(s, x3[0]); s = add_rn(s, x3[1]); s = add_rn(s, x3[2]); s = add_rn(s, x3[3]);
                };
                const uint32_t nr = left >= kStage2 ? kStage2 / 16 : (left + 15) / 16;  // rounds of 16 terms
                // two rounds of reads in flight (y: next round, z: the one after)
                f32x4r y0 = tq[0], y1 = tq[1], y2 = tq[2], y3 = tq[3];
                f32x4r z0 = tq[4], z1 = tq[5], z2 = tq[6], z3 = tq[7];
                uint32_t r = 0;
                for (; r + 4 <= nr; r += 4) {  // four rounds straight-line (64 terms)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4r x0 = y0, x1 = y1, x2 = y2, x3 = y3;
                        y0 = z0; y1 = z1; y2 = z2; y3 = z3;
                        const uint32_t qn = (4 * (r + u + 2)) & (kStage2 / 4 - 1);  // (wraps to the first quads after the last rounds: harmless re-reads)
                        z0 = tq[qn]; z1 = tq[qn + 1]; z2 = tq[qn + 2]; z3 = tq[qn + 3];
                        __builtin_amdgcn_sched_barrier(0);
                        round16(x0, x1, x2, x3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                for (; r < nr; ++r) {
                    const f32x4r x0 = y0, x1 = y1, x2 = y2, x3 = y3;
                    y0 = z0; y1 = z1; y2 = z2; y3 = z3;
                    const uint32_t qn = (4 * (r + 2)) & (kStage2 / 4 - 1);
                    z0 = tq[qn]; z1 = tq[qn + 1]; z2 = tq[qn + 2]; z3 = tq[qn + 3];
                    round16(x0, x1, x2, x3);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) { xa[j] = xb[j]; a.r[j] = b.r[j]; }
                b = c2;
            }
        }
        if (role == 0 && lane == 0) g_slot = s;
        __syncthreads();
        if (role == 1 && lane == 0) out[c] = hi > lo ? g_slot / s : 0.f;  // correctly rounded fp32 division, as on the host; empty: 0 (src/CCD.cpp:8)
        __syncthreads();  // g_slot is free again
    }
}

}  // namespace

// Segments by descending length (ties: ascending id): the dispatch order of k_sweep_ref.
uint32_t ref_sweep_order(const uint32_t* ptr_host, uint32_t nseg, std::vector<uint32_t>* order) {
    order->resize(nseg);
    std::iota(order->begin(), order->end(), 0u);
    std::stable_sort(order->begin(), order->end(), [&](uint32_t x, uint32_t y) {
        return ptr_host[x + 1] - ptr_host[x] > ptr_host[y + 1] - ptr_host[y];
    });
    static const uint32_t thr = [] { const char* e = std::getenv("MFX_REF_LONG"); const int v = e ? std::atoi(e) : 0; return v > 0 ? (uint32_t) v : kRefLong; }();
    uint32_t nlong = 0;
    while (nlong < nseg && nlong < 8192u && ptr_host[(*order)[nlong] + 1] - ptr_host[(*order)[nlong]] >= thr) ++nlong;
    return nlong;
}

int launch_sweep_ref(const SegStreamDev& s, const uint32_t* order, uint32_t nlong, const float* vec, float lambda, float* out, hipStream_t st) {
    if (s.nseg == 0) return MFX_OK;
    MFX_REQUIRE(s.panel_rows == 0 && s.ptr && (s.nnz == 0 || (s.idx && s.val)), "the reference-order sweep needs the plain layout");
    static const bool dpp_form = [] { const char* e = std::getenv("MFX_REF_SWEEP_DPP"); return e && std::atoi(e) != 0; }();  // (A/B: the r3 kernel alone)
    if (dpp_form || !order || nlong == 0) {  // (no long segment -- e.g. the row side of the Netflix shape: the r3 launch shape, 0.61 against 0.79 ms)
        const uint32_t waves_per_block = kRefBlock / 64;
        const uint32_t grid = std::min<uint32_t>((s.nseg + waves_per_block - 1) / waves_per_block, 256u * 8u);
        hipLaunchKernelGGL(k_sweep_ref, dim3(grid), dim3(kRefBlock), 0, st, s.nseg, order, s.ptr, s.idx, s.val, vec, lambda, out);
    } else {
        const uint32_t nshort = s.nseg - nlong;
        const uint32_t grid = nlong + std::min<uint32_t>((nshort + kRef2Block / 64 - 1) / (kRef2Block / 64), 256u * 16u);
        hipLaunchKernelGGL(k_sweep_ref2, dim3(std::max(grid, 1u)), dim3(kRef2Block), 0, st, s.nseg, nlong, order, s.ptr, s.idx, s.val, vec, lambda, out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MFX_ERR_HIP, "k_sweep_ref launch failed: %s", hipGetErrorString(e));
    return MFX_OK;
}

}  // namespace mfx

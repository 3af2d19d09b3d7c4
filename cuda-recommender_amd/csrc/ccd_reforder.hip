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
// first so that such a column starts at once and the short ones fill in behind it.
// This first form (k_sweep_ref, and k_sweep_ref2 below it) is the as-written sequence of the mode, one launch per
// reference kernel (MFX_REF_FUSED=0); the default since the second half of round 4 are the owner passes at the end
// of this file: quad-row chains at 5.75 clocks per entry on the default path's schedule.
#include "ccd_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>
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
constexpr uint32_t kRefLong = 32768;           // k_sweep_ref2 (the as-written sequence, MFX_REF_FUSED=0): entries from which a segment takes the two-wave plain-add form:
                                               // Netflix shape, v-sweep 1.32 ms with 32768 against 1.39-1.49 ms with 8192 ... 1 and 1.93 ms without
constexpr uint32_t kRefSplit = 2048;           // launch_ref_owner: entries from which a segment goes to k_ref_split (MFX_REF_LONG overrides both, A/B)
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


// ---------------------------------------------------------------------------------------------
// (r4) QUAD-ROW CHAINS: four segments per wavefront, and the fused pass of the default schedule in the reference's summation order.
//
// tools/ubench_chain.hip, second half: with the running sum as the PLAIN operand and the term as the DPP operand,
//     v_add_f32_dpp s, x, s row_newbcast:j          (lane j of every row of 16 lanes, to the whole row)
// a dependent add issues every 5.75 clocks (4.63 for a plain VGPR term, 7.3 with ds_read_b128 broadcasts, 12.4 for the lane-to-lane
// wave_shr:1 step), PROVIDED the two s_nop the compiler puts between DPP instructions that read a register the previous VALU
// instruction wrote are left out -- the hardware's hazard concerns the DPP (permuted) operand, which here was loaded long before;
// the sums of the microbenchmark are bit-identical with and without them.  Hence inline asm for the 64 adds of a chunk.
// The form: every ROW of 16 lanes carries its own segment's running sums (all 16 lanes the same value, redundantly); lane j of a
// row holds entries 4 j ... 4 j + 3 of the row's current 64-entry chunk (ONE 16-byte load each of indices and values per lane and
// chunk, like k_seg_owner), and the 64 adds of a chunk walk j = 0 ... 15, e = 0 ... 3: entry order.  A wavefront advances FOUR
// independent chains per instruction where the lane-to-lane form advanced one -- rows of 200 entries (the u-sweep) cost a quarter
// of the vector instructions -- and there is no LDS traffic in the chain at all.  Terms outside [lo, hi) are +0: s + 0 == s bit
// for bit (a sum that started at +0 or lambda * count >= 0 is never -0), so chunks are aligned to 16 bytes and rows of different
// lengths share a wave without masks in the chain.
//
// On top of it the reference-order mode gets the SCHEDULE of the default path (ccd_solver.hip, rank_fused_owner): the pending
// subtraction of rank t - 1, the add-back of rank t and the first sweep of rank t in one pass per copy -- r' = (r - a b) + c d with
// the reference's own unfused roundings (the element update of every other path), the term from r' -- and the division and the
// operand packs of the next pass written by the segment's owner: two launches per rank instead of four sweeps / passes, no separate
// residual pass (128 x 0.57 ms per outer iteration at the Netflix shape).  Every stored residual, every sum and every factor
// entry keeps the reference's bits (tests/test_gpu_ccd_reforder.py, test_gpu_fullsize.py: unchanged).
//
// Two kernels per launch_ref_owner:
//   k_ref_quad   one wavefront per item (4 segments, longest first), both chains (g, h) interleaved: issue-bound; loads two chunks
//                ahead, the gather one chunk ahead, many waves per SIMD.  A throughput form: a wave's own progress on a long segment is
//                one memory round trip per chunk (a 150 000-entry column takes it 2 ms).
//   k_ref_split  items whose longest segment has >= kRefSplit = 2048 entries: a 512-thread workgroup per item.  Wave 0 runs the four g
//                chains and wave 1 the four h chains -- ONE dependent add per entry and nothing else: their terms come from LDS, 64
//                per ds_read_b128, put there a batch of 36 chunks ahead by the six loader waves (loads, gather, element update,
//                write-back of r', products), one LDS-only barrier per batch.  What it took to make the chains the bound
//                (237 k-entry column of the Netflix shape: 705 us per pass = 6.3 clocks per entry; tools/exp_ref_owner.sh,
//                profiles/r04_exp_reforder_owner.txt):
//                  * loaders that never wait for a load younger than a batch (explicit s_waitcnt: the compiler's counters, made
//                    conservative by the conditional stores between a load and its use, waited for the loads just issued), and a
//                    barrier that does not drain vmcnt (__syncthreads does): 1187 -> ~960 us, and no further --
//                  * because k_ref_quad's waves on the SAME CU queue their gathers in front of the loaders' (with 1 quad workgroup per
//                    2 CUs the split kernel took 705 us, with 2 per CU 1010 us; s_setprio changes nothing, it is the memory
//                    pipeline of the CU).  A split workgroup therefore takes its CU for itself: 8 waves x 256 registers = the CU's
//                    register file (the `asm volatile("" ::: "v255")`), and it is launched FIRST, on the main stream, with k_ref_quad
//                    behind it on the side stream, so that its workgroups are placed before the first quad wave.
//                  * with that, the split form is also the efficient one for every segment that keeps a CU busy (a CU serves four
//                    chains at full speed): the threshold went from 32768 to 2048 entries (per outer iteration, with the LDS table
//                    below: 102 ms at 8192, 92 at 6144, 80 at 4096, 71.5 at 3072, 70.4 at 2048, 69-70 at 1536 ... 512; before the
//                    table 147 ms at 16384, 120 / 143 / 177 ms at 65536 / 100 000 / 150 000).
// Netflix shape, k = 64: v-pass 0.59 ms (the 237 k-entry chain), u-pass 0.47 ms (k_ref_quad with the LDS table; 0.72 ms without, bound by
// the L2 gather like the plain flat pass), the mode 201 -> 66-68 ms per outer iteration (profiles/r04_bench_reforder.json).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
using u32x4r = __attribute__((ext_vector_type(4))) uint32_t;

struct RefOwnerArgs {
    const uint32_t* ptr;      // [nseg + 1] input-order pointers (plain layout)
    const uint32_t* idx;
    float* val;
    const void* gather;
    const void* perseg;
    const uint32_t* order;    // [nseg] segments, longest first
    uint32_t nseg;
    uint32_t item0, nitems;   // k_ref_quad: items [item0, nitems); k_ref_split: item = blockIdx.x (< item0)
    float lambda;
    float* out_vec;
    float2* pack2;            // in: (prev_new, cur_old) = perseg; out: (cur_new, next_old)
    const float* next_vec;
    float4* pack4;            // out: (prev_new, cur_old, cur_new, 0), may be nullptr
};

// G: element of the gathered operand; S: the same inside k_ref_quad's LDS table (12 of the 16 bytes of a float4 operand); P: per-segment operand
struct F3r { float x, y, z; };
template <int MODE> struct RefTraits;
template <> struct RefTraits<FM_SWEEP> { using G = float;  using S = float;  using P = float;  static constexpr bool kWrite = false, kPerSeg = false; };
template <> struct RefTraits<FM_FCSC>  { using G = float2; using S = float2; using P = float2; static constexpr bool kWrite = true,  kPerSeg = true; };
template <> struct RefTraits<FM_FCSR>  { using G = float4; using S = F3r;    using P = float2; static constexpr bool kWrite = true,  kPerSeg = true; };
__device__ __forceinline__ float to_tab(float g) { return g; }
__device__ __forceinline__ float2 to_tab(const float2& g) { return g; }
__device__ __forceinline__ F3r to_tab(const float4& g) { return F3r{g.x, g.y, g.z}; }
__device__ __forceinline__ float from_tab(float s) { return s; }
__device__ __forceinline__ float2 from_tab(const float2& s) { return s; }
__device__ __forceinline__ float4 from_tab(const F3r& s) { return make_float4(s.x, s.y, s.z, 0.f); }

// One element: the new residual value (the reference's unfused update, src/CCD.cpp:25,36 -- element_op of ccd_kernels.hip) and the
// operand of its two products.
template <int MODE>
__device__ __forceinline__ void ref_element(float v, const typename RefTraits<MODE>::G& ga, const typename RefTraits<MODE>::P& ps,
                                            float& v_out, float& op) {
    if constexpr (MODE == FM_SWEEP) {
        v_out = v; op = ga;
    } else if constexpr (MODE == FM_FCSC) {
        v_out = add_rn(sub_rn(v, mul_rn(ga.x, ps.x)), mul_rn(ga.y, ps.y)); op = ga.y;
    } else {
        v_out = add_rn(sub_rn(v, mul_rn(ga.x, ps.x)), mul_rn(ga.y, ps.y)); op = ga.z;
    }
}

// 64 dependent adds: s <- s + x[e] of lane j of the row, j = 0 ... 15, e = 0 ... 3 (see above for the missing s_nops)
#define MFX_BC1(S, X, J) "v_add_f32_dpp " S ", " X ", " S " row_newbcast:" #J " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define MFX_BC4(J) MFX_BC1("%0", "%1", J) MFX_BC1("%0", "%2", J) MFX_BC1("%0", "%3", J) MFX_BC1("%0", "%4", J)
#define MFX_BC8(J) MFX_BC1("%0", "%2", J) MFX_BC1("%1", "%6", J) MFX_BC1("%0", "%3", J) MFX_BC1("%1", "%7", J) \
                   MFX_BC1("%0", "%4", J) MFX_BC1("%1", "%8", J) MFX_BC1("%0", "%5", J) MFX_BC1("%1", "%9", J)
#define MFX_ALL16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
__device__ __forceinline__ void chain_one(float& s, const f32x4r& x) {
    asm volatile("s_nop 1\n" MFX_ALL16(MFX_BC4) : "+v"(s) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}
__device__ __forceinline__ void chain_two(float& sg, float& sh, const f32x4r& xg, const f32x4r& xh) {
    asm volatile("s_nop 1\n" MFX_ALL16(MFX_BC8)
                 : "+v"(sg), "+v"(sh)
                 : "v"(xg[0]), "v"(xg[1]), "v"(xg[2]), "v"(xg[3]), "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]));
}

// The segment of this lane's row: item q holds the segments order[4 q ... 4 q + 3].
struct RowSeg {
    uint32_t c, lo, hi, b0, nch;  // b0: first chunk's first entry (lo rounded down to 16 bytes); nch: 64-entry chunks
    bool have;
};
__device__ __forceinline__ RowSeg row_segment(const RefOwnerArgs& a, uint32_t item, uint32_t lane) {
    RowSeg r;
    const uint32_t q = 4u * item + (lane >> 4);
    r.have = q < a.nseg;
    r.c = r.have ? a.order[q] : 0u;
    r.lo = r.have ? a.ptr[r.c] : 0u;
    r.hi = r.have ? a.ptr[r.c + 1] : 0u;
    r.b0 = r.lo & ~3u;
    r.nch = r.hi > r.lo ? (r.hi - r.b0 + 63u) / 64u : 0u;
    return r;
}
__device__ __forceinline__ uint32_t rows_max(uint32_t x) {  // wave-uniform maximum over the four rows
    const uint32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32),
                   d = __builtin_amdgcn_readlane(x, 48);
    return max(max(a, b), max(c, d));
}
// quad of chunk t: first entry, and where to load it from (a quad with no entry of the segment re-reads entries 0 ... 3: every
// store holds at least one padded tile)
__device__ __forceinline__ uint32_t quad_pos(const RowSeg& r, uint32_t t, uint32_t lane) { return r.b0 + 64u * t + 4u * (lane & 15u); }
__device__ __forceinline__ uint32_t quad_at(const RowSeg& r, uint32_t p) { return p < r.hi ? p : 0u; }

// TAB: the operands of the indices below tab_n sit in the workgroup's LDS table (k_ref_quad); only the others go to L2
template <int MODE, bool TAB = false>
__device__ __forceinline__ void gather_quad(const RefOwnerArgs& a, const RowSeg& r, uint32_t p, const u32x4r& id,
                                            typename RefTraits<MODE>::G (&g)[4], const typename RefTraits<MODE>::S* tab = nullptr, uint32_t tab_n = 0) {
    using G = typename RefTraits<MODE>::G;
    const G* __restrict__ gather = static_cast<const G*>(a.gather);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t pos = p + (uint32_t) e;
        const uint32_t ii = (pos >= r.lo && pos < r.hi) ? id[e] : 0u;  // (padding and the neighbours' entries: not this gather's indices)
        if constexpr (TAB) {
            // both loads, unconditionally (the global one of a table-served lane re-reads operand 0: one request for all of them): a load
            // inside a branch makes every s_waitcnt the compiler derives afterwards allow for it NOT having been issued -- too strict
            const G from_l2 = gather[ii < tab_n ? 0u : ii];
            const G from_lds = from_tab(tab[ii < tab_n ? ii : 0u]);
            g[e] = ii < tab_n ? from_lds : from_l2;
        } else {
            g[e] = gather[ii];
        }
    }
}
// element update + write-back of a quad; the two products of every entry of [lo, hi), +0 elsewhere
template <int MODE, bool STORE, bool WANT_G, bool WANT_H>
__device__ __forceinline__ void quad_terms(const RefOwnerArgs& a, const RowSeg& r, uint32_t p, const f32x4r& v,
                                           const typename RefTraits<MODE>::G (&g)[4], const typename RefTraits<MODE>::P& ps,
                                           f32x4r& xg, f32x4r& xh) {
    f32x4r o;
    bool all_live = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t pos = p + (uint32_t) e;
        const bool live = pos >= r.lo && pos < r.hi;
        all_live &= live;
        float vo, op;
        ref_element<MODE>(v[e], g[e], ps, vo, op);
        o[e] = vo;
        if constexpr (WANT_G) xg[e] = live ? mul_rn(op, vo) : 0.f;
        if constexpr (WANT_H) xh[e] = live ? mul_rn(op, op) : 0.f;
    }
    if constexpr (STORE && RefTraits<MODE>::kWrite) {
        if (all_live) {
            *reinterpret_cast<f32x4r*>(a.val + p) = o;
        } else if (p < r.hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const uint32_t pos = p + (uint32_t) e; if (pos >= r.lo && pos < r.hi) a.val[pos] = o[e]; }
        }
    }
}
// the owner's last step: the reference's division (g / h with h started at lambda * count; 0 for an empty segment, src/CCD.cpp:6-16)
// and the operand packs of the next passes (seg_owner_finish of ccd_kernels.hip)
template <int MODE>
__device__ __forceinline__ void ref_finish(const RefOwnerArgs& a, const RowSeg& r, float g, float h, const typename RefTraits<MODE>::P& ps, float next_old) {
    const float x = r.hi > r.lo ? g / h : 0.f;  // correctly rounded fp32 division, as on the host
    a.out_vec[r.c] = x;
    if constexpr (MODE != FM_SWEEP) {
        if (a.pack4) a.pack4[r.c] = make_float4(ps.x, ps.y, x, 0.f);
        a.pack2[r.c] = make_float2(x, a.next_vec == a.out_vec ? x : next_old);
    }
}

// TAB: the workgroup copies the first tab_n operands of the gather table into LDS once and serves those indices from there.  The
// plain form's u-pass is bound by its gathers -- every 16-byte operand is a 64-byte sector from L2 (0.72 ms at the Netflix shape, the
// plain flat pass's 0.57 ms plus the chains) -- and the column operands of that pass are a 17 770-entry table: 13 600 of them fit a CU's
// LDS as 12-byte triples.  One 1024-thread workgroup per CU then (the table is per workgroup), grid-strided over the items.
constexpr int kQuadBlock = 128, kQuadTabBlock = 1024;
constexpr size_t kQuadTabBytes = 160 * 1024 - 512;
template <int MODE, bool TAB>
__global__ __launch_bounds__(TAB ? kQuadTabBlock : kQuadBlock) void k_ref_quad(RefOwnerArgs a, uint32_t tab_n) {
    using TR = RefTraits<MODE>;
    using G = typename TR::G;
    using S = typename TR::S;
    using P = typename TR::P;
    extern __shared__ __attribute__((aligned(16))) unsigned char quad_lds[];
    const S* tab = reinterpret_cast<const S*>(quad_lds);
    if constexpr (TAB) {
        S* w = reinterpret_cast<S*>(quad_lds);
        const G* __restrict__ gather = static_cast<const G*>(a.gather);
        for (uint32_t i = threadIdx.x; i < tab_n; i += blockDim.x) w[i] = to_tab(gather[i]);
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    // An item's start-up is a chain of dependent loads: order -> pointers (+ per-segment operands) -> indices -> gather.  The first two
    // links are taken off it: while item i runs, the pointers and operands of item i + 1 (whose segment ids arrived during item i - 1)
    // and the segment ids of item i + 2 are in flight.  All of these loads are unconditional (an absent segment re-reads segment 0 and
    // is voided afterwards): see gather_quad on conditional loads and the compiler's wait counts.
    struct Meta { uint32_t c, lo, hi; P ps; float next_old; };
    auto seg_id = [&](uint32_t it, bool& have) {  // (it may lie past the last item: wave-uniformly voided)
        const uint32_t q = 4u * it + (lane >> 4);
        have = it < a.nitems && q < a.nseg;
        return a.order[have ? q : 0u];
    };
    auto fetch_meta = [&](uint32_t c, bool have) {
        Meta m;
        m.c = have ? c : 0u;
        m.lo = a.ptr[m.c];
        m.hi = a.ptr[m.c + 1];
        m.ps = P{};
        if constexpr (TR::kPerSeg) m.ps = static_cast<const P*>(a.perseg)[m.c];
        m.next_old = 0.f;
        if constexpr (MODE != FM_SWEEP) m.next_old = a.next_vec[m.c];
        if (!have) m.lo = m.hi = 0u;
        return m;
    };
    auto row_of = [&](const Meta& m, bool have) {
        RowSeg r;
        r.have = have; r.c = m.c; r.lo = m.lo; r.hi = m.hi;
        r.b0 = r.lo & ~3u;
        r.nch = r.hi > r.lo ? (r.hi - r.b0 + 63u) / 64u : 0u;
        return r;
    };
    uint32_t item = a.item0 + wave;
    bool have_cur, have_nxt, have_n2;
    const uint32_t c0 = seg_id(item, have_cur);
    uint32_t c_nxt = seg_id(item + nwaves, have_nxt);
    Meta m_cur = fetch_meta(c0, have_cur);
    for (; item < a.nitems; item += nwaves) {
        const Meta m_nxt = fetch_meta(c_nxt, have_nxt);             // item + nwaves: used after this item
        const uint32_t c_n2 = seg_id(item + 2u * nwaves, have_n2);  // item + 2 nwaves: its pointers are fetched during the next item
        const RowSeg r = row_of(m_cur, have_cur);
        const P ps = m_cur.ps;
        const float next_old = m_cur.next_old;
        float sg = 0.f, sh = mul_rn(a.lambda, (float) (r.hi - r.lo));  // float * unsigned (src/CCD.cpp:112,120)
        const uint32_t nmax = rows_max(r.nch);
        if (nmax) {
            // A ring of four chunks: the indices and values of chunks t + 1 ... t + 3 are in flight (or here) while chunk t runs its
            // chains; the gather of chunk t + 1 goes out one chunk ahead (two operand sets, alternating).  Rows of up to 256 entries
            // -- the typical user row -- have ALL their loads issued at once (a first version kept two chunks in flight and paid
            // one memory round trip per chunk).  The loop is unrolled by the ring size: every register index is static.
            u32x4r I[4];
            f32x4r V[4];
            G ga[4], gb[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t at = quad_at(r, quad_pos(r, (uint32_t) d, lane));
                I[d] = *reinterpret_cast<const u32x4r*>(a.idx + at);
                V[d] = *reinterpret_cast<const f32x4r*>(a.val + at);
            }
            gather_quad<MODE, TAB>(a, r, quad_pos(r, 0, lane), I[0], ga, tab, tab_n);
            for (uint32_t t0 = 0; t0 < nmax; t0 += 4) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t t = t0 + (uint32_t) d;
                    if (t >= nmax) break;
                    G (&g_cur)[4] = (d & 1) ? gb : ga;
                    G (&g_next)[4] = (d & 1) ? ga : gb;
                    gather_quad<MODE, TAB>(a, r, quad_pos(r, t + 1, lane), I[(d + 1) & 3], g_next, tab, tab_n);
                    f32x4r xg, xh;
                    quad_terms<MODE, true, true, true>(a, r, quad_pos(r, t, lane), V[d], g_cur, ps, xg, xh);
                    const uint32_t at = quad_at(r, quad_pos(r, t + 4, lane));  // (this set's indices went into the gather a chunk ago)
                    I[d] = *reinterpret_cast<const u32x4r*>(a.idx + at);
                    V[d] = *reinterpret_cast<const f32x4r*>(a.val + at);
                    chain_two(sg, sh, xg, xh);
                }
            }
        }
        if (r.have && (lane & 15u) == 0) ref_finish<MODE>(a, r, sg, sh, ps, next_old);
        m_cur = m_nxt; have_cur = have_nxt;
        c_nxt = c_n2; have_nxt = have_n2;
    }
}

constexpr int kSplitLoaders = 6;                           // loader waves of a split workgroup (waves 2 ... 7): eight waves = two per SIMD,
constexpr int kSplitBlock = 64 * (2 + kSplitLoaders);      // 256 registers each (at ten waves the allocator moved in-flight gathers about)
constexpr uint32_t kSplitBatchMax = 36;                    // chunks per batch: 6 per loader wave (4 with 16-byte gather operands)
constexpr uint32_t kSplitChunkFloats = 2 * 256;            // a chunk's terms: g of the four rows, then h
constexpr size_t kSplitLds = (size_t) 2 * kSplitBatchMax * kSplitChunkFloats * sizeof(float) + 16;  // two batches + the four g results
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence drains
// EVERY outstanding memory operation of the wave (s_waitcnt vmcnt(0)) -- including the loads a loader wave has issued for the batches
// to come, i.e. one fully exposed memory round trip per batch.  The waves of a split workgroup exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(kSplitBlock) void k_ref_split(RefOwnerArgs a) {
    using TR = RefTraits<MODE>;
    using G = typename TR::G;
    using P = typename TR::P;
    extern __shared__ __attribute__((aligned(16))) float split_lds[];
    float* g_slot = split_lds + 2 * kSplitBatchMax * kSplitChunkFloats;
    constexpr int LG = sizeof(G) > 8 ? 4 : 6;              // chunks per loader wave and batch
    constexpr uint32_t kSplitBatch = LG * kSplitLoaders;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowSeg r = row_segment(a, blockIdx.x, lane);
    P ps{};
    if constexpr (TR::kPerSeg) { if (r.have) ps = static_cast<const P*>(a.perseg)[r.c]; }
    const uint32_t nmax = rows_max(r.nch);
    const uint32_t nbatch = (nmax + kSplitBatch - 1) / kSplitBatch;
    // Loader wave w (= wave - 2) owns the chunks t = w (mod kSplitLoaders) of every batch: kSplitGroup chunks, handled together.
    // Nothing a loader waits for was issued less than a batch (the chains' ~5 us) earlier: while it does the arithmetic of batch n,
    // the gathers of batch n + 1 and the index / value loads of batches n + 2 / n + 1 are in flight (two register sets that swap
    // roles every batch -- the batch loop is unrolled by two, there are no copies of registers a load is still to write).  Under
    // k_ref_quad's traffic a round trip takes ~3.4 us: a first version that waited for the loads, then for the gathers, twice per
    // batch, took 1187 us for the 237 k-entry column (the r3 kernel's time), one exposed round trip per batch 850-920 us.
    u32x4r id[LG];
    f32x4r va[LG], vb[LG];
    G ga[LG][4], gb[LG][4];
    auto chunk_of = [&](uint32_t n, int u) { return n * kSplitBatch + (uint32_t) u * kSplitLoaders + (wave - 2u); };
    auto load_idx = [&](uint32_t n) {
#pragma unroll
        for (int u = 0; u < LG; ++u) {
            const uint32_t t = chunk_of(n, u), p = quad_pos(r, t, lane);
            id[u] = *reinterpret_cast<const u32x4r*>(a.idx + (t < nmax ? quad_at(r, p) : 0u));
        }
    };
    auto load_val = [&](uint32_t n, f32x4r (&v)[LG]) {
#pragma unroll
        for (int u = 0; u < LG; ++u) {
            const uint32_t t = chunk_of(n, u), p = quad_pos(r, t, lane);
            v[u] = *reinterpret_cast<const f32x4r*>(a.val + (t < nmax ? quad_at(r, p) : 0u));
        }
    };
    auto gather_batch = [&](uint32_t n, G (&g)[LG][4]) {  // from id = the indices of batch n
#pragma unroll
        for (int u = 0; u < LG; ++u) {
            const uint32_t t = chunk_of(n, u);
            RowSeg rr = r;
            if (t >= nmax) rr.hi = rr.lo;  // (a chunk past the end: nothing is live, nothing is gathered or stored)
            gather_quad<MODE>(a, rr, quad_pos(r, t, lane), id[u], g[u]);
        }
    };
    // the terms of batch n into buffer n & 1; on entry (v_cur, g_cur) belong to batch n and id holds the indices of batch n + 1
    auto produce = [&](uint32_t n, const f32x4r (&v_cur)[LG], f32x4r (&v_next)[LG], const G (&g_cur)[LG][4], G (&g_next)[LG][4]) {
        // ONE wait, for everything the previous batch issued (a batch-time ago), then all of this batch's loads, then the arithmetic.
        // Left to the compiler's counters the arithmetic waited for loads issued a few instructions earlier: its model allows for the
        // conditional stores (and, in a first version, conditional loads) between a load and its use NOT having been issued, so
        // every vmcnt it derives is too strict by that many -- one exposed round trip per batch, the very thing the pipeline removes.
        // Loads of batches past the end re-read entries 0 ... 3 (unconditional: same reason).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        gather_batch(n + 1, g_next);
        load_val(n + 1, v_next);
        load_idx(n + 2);
#pragma unroll
        for (int u = 0; u < LG; ++u) {
            const uint32_t t = chunk_of(n, u);
            if (t < nmax) {
                f32x4r xg, xh;
                quad_terms<MODE, true, true, true>(a, r, quad_pos(r, t, lane), v_cur[u], g_cur[u], ps, xg, xh);
                float* dst = split_lds + ((n & 1u) * kSplitBatch + (t - n * kSplitBatch)) * kSplitChunkFloats + 4u * lane;
                *reinterpret_cast<f32x4r*>(dst) = xg;
                *reinterpret_cast<f32x4r*>(dst + 256) = xh;
            }
        }
    };
    auto consume = [&](uint32_t b, float& s) {
        const uint32_t cnt = min(kSplitBatch, nmax - b * kSplitBatch);
        const float* src = split_lds + (b & 1u) * kSplitBatch * kSplitChunkFloats + wave * 256u + 4u * lane;
        f32x4r x = *reinterpret_cast<const f32x4r*>(src);
        for (uint32_t c = 0; c < cnt; ++c) {
            const f32x4r xn = *reinterpret_cast<const f32x4r*>(src + min(c + 1, cnt - 1) * kSplitChunkFloats);
            chain_one(s, x);
            x = xn;
        }
    };
    float s = wave == 0 ? 0.f : mul_rn(a.lambda, (float) (r.hi - r.lo));  // wave 0: g, wave 1: h (float * unsigned, src/CCD.cpp:112,120)
    const bool loader = wave >= 2;
    asm volatile("" ::: "v255");  // (256 registers per wave: see above -- the workgroup's eight waves fill the CU's register file)
    if (loader && nbatch) {
        load_idx(0);
        load_val(0, va);
        gather_batch(0, ga);
        load_idx(1);
        produce(0, va, vb, ga, gb);
    }
    lds_barrier();
    for (uint32_t b = 0; b < nbatch; b += 2) {
        // even batch b is consumed while the loaders make batch b + 1 (its inputs in the b sets), then the other way round
        if (wave < 2) consume(b, s);
        else if (loader && b + 1 < nbatch) produce(b + 1, vb, va, gb, ga);
        lds_barrier();
        if (b + 1 >= nbatch) break;
        if (wave < 2) consume(b + 1, s);
        else if (loader && b + 2 < nbatch) produce(b + 2, va, vb, ga, gb);
        lds_barrier();
    }
    if (wave == 0 && (lane & 15u) == 0) g_slot[lane >> 4] = s;
    lds_barrier();
    if (wave == 1 && r.have && (lane & 15u) == 0) {
        float next_old = 0.f;
        if constexpr (MODE != FM_SWEEP) { if (a.next_vec) next_old = a.next_vec[r.c]; }
        ref_finish<MODE>(a, r, g_slot[lane >> 4], s, ps, next_old);
    }
}

}  // namespace

// Items (four segments of `order` each) that take the split form: those holding one of the nlong long segments.
uint32_t ref_split_items(uint32_t nlong) { return (nlong + 3u) / 4u; }

int launch_ref_owner(FlatMode mode, const SegStreamDev& s, const uint32_t* order, uint32_t nlong, const void* gather, const void* perseg,
                     const FinalizeArgs& f, const RefStreams& rs) {
    if (s.nseg == 0) return MFX_OK;
    MFX_REQUIRE(s.panel_rows == 0 && !s.scatter && s.ptr && s.idx && s.val && order, "launch_ref_owner: needs the plain layout and the dispatch order");
    MFX_REQUIRE(!f.gh_dense && !f.cnt_override && !f.pack4_as3 && !f.nmf && !f.fundec_seg && f.out_vec, "launch_ref_owner: dense / overridden / extended finalize inputs are not supported");
    MFX_REQUIRE(mode == FM_SWEEP || ((mode == FM_FCSC || mode == FM_FCSR) && f.pack2 && f.next_vec && f.pack2 == perseg), "launch_ref_owner: bad mode / packs");
    RefOwnerArgs a;
    a.ptr = s.ptr; a.idx = s.idx; a.val = s.val; a.gather = gather; a.perseg = perseg; a.order = order; a.nseg = s.nseg;
    a.nitems = (s.nseg + 3u) / 4u;
    a.item0 = std::min(ref_split_items(nlong), a.nitems);
    a.lambda = f.lambda; a.out_vec = f.out_vec; a.pack2 = f.pack2; a.next_vec = f.next_vec; a.pack4 = f.pack4;
    const bool side = a.item0 > 0 && rs.side && rs.fork && rs.join && a.item0 < a.nitems;
    // the split kernel is enqueued FIRST, on the main stream; the quad kernel goes to the side stream behind an event recorded on the main
    // stream just before (so both wait for the same earlier work and then run side by side, the split workgroups placed first)
    hipStream_t st_split = rs.main, st_quad = side ? rs.side : rs.main;
    if (a.item0 > 0) {
        static std::mutex m;
        static bool attr_set[3][64] = {};
        int dev = 0;
        MFX_HIP(hipGetDevice(&dev));
        const int mi = mode == FM_SWEEP ? 0 : (mode == FM_FCSC ? 1 : 2);
        {
            std::lock_guard<std::mutex> lk(m);
            if (dev < 0 || dev >= 64 || !attr_set[mi][dev]) {
                const void* fn = mode == FM_SWEEP ? reinterpret_cast<const void*>(k_ref_split<FM_SWEEP>)
                                 : mode == FM_FCSC ? reinterpret_cast<const void*>(k_ref_split<FM_FCSC>) : reinterpret_cast<const void*>(k_ref_split<FM_FCSR>);
                MFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kSplitLds));
                if (dev >= 0 && dev < 64) attr_set[mi][dev] = true;
            }
        }
        if (side) {
            MFX_HIP(hipEventRecord(rs.fork, rs.main));
            MFX_HIP(hipStreamWaitEvent(rs.side, rs.fork, 0));
        }
        switch (mode) {
            case FM_SWEEP: hipLaunchKernelGGL(k_ref_split<FM_SWEEP>, dim3(a.item0), dim3(kSplitBlock), kSplitLds, st_split, a); break;
            case FM_FCSC: hipLaunchKernelGGL(k_ref_split<FM_FCSC>, dim3(a.item0), dim3(kSplitBlock), kSplitLds, st_split, a); break;
            default: hipLaunchKernelGGL(k_ref_split<FM_FCSR>, dim3(a.item0), dim3(kSplitBlock), kSplitLds, st_split, a); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MFX_ERR_HIP, "k_ref_split launch failed: %s", hipGetErrorString(e));
    }
    if (a.item0 < a.nitems) {
        const uint32_t nq = a.nitems - a.item0;
        // the LDS table form when a CU's LDS covers at least a quarter of the operand table (MFX_REF_QUAD_TAB=0: never; = n > 1: a table
        // of n bytes whatever it covers -- tests and the fuzzer exercise partly covered tables on small matrices with it)
        const char* tab_env = std::getenv("MFX_REF_QUAD_TAB");
        const long tab_set = tab_env ? std::atol(tab_env) : -1;
        const size_t elem = mode == FM_SWEEP ? sizeof(float) : mode == FM_FCSC ? sizeof(float2) : sizeof(F3r);
        const char* cap_env = std::getenv("MFX_REF_QUAD_TAB_BYTES");  // (A/B: a smaller table under the default rule -- two workgroups per CU from 80 KB down:
                                                                      // Netflix shape 73.4 ms per outer iteration with 80 KB, 71.7 with 106 KB, 75.2 with 53 KB, 66-68 with the full 160 KB)
        const size_t tab_cap = cap_env && std::atol(cap_env) > 0 ? std::min<size_t>((size_t) std::atol(cap_env), kQuadTabBytes) : kQuadTabBytes;
        const size_t tab_bytes = tab_set > 1 ? std::min<size_t>((size_t) tab_set, kQuadTabBytes) : tab_cap;
        const uint32_t tab_n = (uint32_t) std::min<size_t>(s.gather_len, tab_bytes / elem);
        const bool tab = tab_set != 0 && tab_n > 0 && (tab_set > 1 || ((uint64_t) tab_n * 4u >= s.gather_len && nq >= 64u));
        if (tab) {
            const size_t lds = (size_t) tab_n * elem;
            const void* fn = mode == FM_SWEEP ? reinterpret_cast<const void*>(k_ref_quad<FM_SWEEP, true>)
                             : mode == FM_FCSC ? reinterpret_cast<const void*>(k_ref_quad<FM_FCSC, true>) : reinterpret_cast<const void*>(k_ref_quad<FM_FCSR, true>);
            if (lds > 48 * 1024) {
                static std::mutex m;
                static size_t set_bytes[3][64] = {};
                int dev = 0;
                MFX_HIP(hipGetDevice(&dev));
                const int mi = mode == FM_SWEEP ? 0 : (mode == FM_FCSC ? 1 : 2);
                std::lock_guard<std::mutex> lk(m);
                if (dev < 0 || dev >= 64 || set_bytes[mi][dev] < lds) {
                    MFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kQuadTabBytes));
                    if (dev >= 0 && dev < 64) set_bytes[mi][dev] = kQuadTabBytes;
                }
            }
            const uint32_t waves_per_block = kQuadTabBlock / 64;
            const uint32_t per_cu = (uint32_t) std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / std::max<size_t>(lds, 1)));
            const uint32_t grid = std::min<uint32_t>((nq + waves_per_block - 1) / waves_per_block, 256u * per_cu);
            switch (mode) {
                case FM_SWEEP: hipLaunchKernelGGL((k_ref_quad<FM_SWEEP, true>), dim3(grid), dim3(kQuadTabBlock), lds, st_quad, a, tab_n); break;
                case FM_FCSC: hipLaunchKernelGGL((k_ref_quad<FM_FCSC, true>), dim3(grid), dim3(kQuadTabBlock), lds, st_quad, a, tab_n); break;
                default: hipLaunchKernelGGL((k_ref_quad<FM_FCSR, true>), dim3(grid), dim3(kQuadTabBlock), lds, st_quad, a, tab_n); break;
            }
        } else {
            const uint32_t waves_per_block = kQuadBlock / 64;
            static const uint32_t cap = [] { const char* e = std::getenv("MFX_REF_QUAD_WGS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? (uint32_t) v : 256u * 16u; }();
            const uint32_t grid = std::min<uint32_t>((nq + waves_per_block - 1) / waves_per_block, cap);
            switch (mode) {
                case FM_SWEEP: hipLaunchKernelGGL((k_ref_quad<FM_SWEEP, false>), dim3(grid), dim3(kQuadBlock), 0, st_quad, a, 0u); break;
                case FM_FCSC: hipLaunchKernelGGL((k_ref_quad<FM_FCSC, false>), dim3(grid), dim3(kQuadBlock), 0, st_quad, a, 0u); break;
                default: hipLaunchKernelGGL((k_ref_quad<FM_FCSR, false>), dim3(grid), dim3(kQuadBlock), 0, st_quad, a, 0u); break;
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MFX_ERR_HIP, "k_ref_quad launch failed: %s", hipGetErrorString(e));
    }
    if (side) {
        MFX_HIP(hipEventRecord(rs.join, rs.side));
        MFX_HIP(hipStreamWaitEvent(rs.main, rs.join, 0));
    }
    return MFX_OK;
}

// Segments by descending length (ties: ascending id): the dispatch order of k_sweep_ref.
uint32_t ref_sweep_order(const uint32_t* ptr_host, uint32_t nseg, std::vector<uint32_t>* order, bool owner_form) {
    order->resize(nseg);
    std::iota(order->begin(), order->end(), 0u);
    std::stable_sort(order->begin(), order->end(), [&](uint32_t x, uint32_t y) {
        return ptr_host[x + 1] - ptr_host[x] > ptr_host[y + 1] - ptr_host[y];
    });
    const uint32_t env_thr = [] { const char* e = std::getenv("MFX_REF_LONG"); const int v = e ? std::atoi(e) : 0; return v > 0 ? (uint32_t) v : 0u; }();  // (A/B, tests)
    const uint32_t thr = env_thr ? env_thr : (owner_form ? kRefSplit : kRefLong);
    uint32_t nlong = 0;
    while (nlong < nseg && nlong < 8192u && ptr_host[(*order)[nlong] + 1] - ptr_host[(*order)[nlong]] >= thr) ++nlong;
    // owner form: the split kernel (and the second stream it costs) only where it has work -- a segment of >= 8192 entries (long for a
    // k_ref_quad wave, which advances a chunk per memory round trip) or at least 64 segments over the threshold (Netflix shape: 67-70 ms
    // per outer iteration with the row side's 1240 rows of >= 2048 entries on the split kernel, 71-72 ms without)
    if (owner_form && !env_thr && nseg && ptr_host[(*order)[0] + 1] - ptr_host[(*order)[0]] < 8192u && nlong < 64u) nlong = 0;
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

// als_solver.hip -- ALS kernels (gfx950) and host orchestration.
//
// Kernel shape.  One wavefront per work item; a work item is a whole segment, or a chunk of a long
// one (AlsHalf::build).  The gathered factor rows go straight from global memory into an MFMA
// operand layout -- no LDS staging, no cross-lane traffic:
//   * 32 < k <= 64, k % 4 == 0: k_als_gram16 -- v_mfma_f32_16x16x4_f32 on "column sets", rows fetched
//     with 16-byte loads, four whole rows per wave instruction (see the comment at the kernel);
//   * any other k <= 128: k_als_gram<NT> -- v_mfma_f32_32x32x2_f32, lane l supplies A[i = l&31][kk = l>>5]
//     and B[kk][j = l&31], so lane l loads X[row(q0 + (l>>5))][32*I + (l&31)]; the same register is
//     the A operand of tile (I, J) and the B operand of tile (J', I).
// Only the upper (block) triangle is accumulated.  fp32 MFMA is an exact k-ordered fmaf chain, so
// results are reproducible run to run.
//
// Tail (per segment, still one wave): accumulators -> LDS (lower triangle only, rows packed and
// 16-B aligned), + lambda on the diagonal (plain lambda, src/ALS.cpp:120-122), left-looking Cholesky
// (the reference's row-by-row scheme, src/ALS.cpp:6-23; dot products fused, one scale by 1/sqrt(pivot) per column),
// then L z = b and L^T y = z instead of the reference's explicit inverse (same solution up to
// rounding; tolerance in the tests).
#include "als_solver.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <type_traits>

#include "ccd_kernels.hpp"

namespace mfx {
namespace {

// unfused multiply / subtract (HIP's __fmul_rn is a plain `*` and would be contracted into v_fma)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f32

// 1 / x: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_nr(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}

// Blocked MFMA Cholesky for k > 32 (chol_blocked); false = the round-1 row-by-row forms, kept for A/B runs.
#ifndef MFX_ALS_BLOCKED
#define MFX_ALS_BLOCKED 1
#endif
constexpr bool kBlockedCholesky = MFX_ALS_BLOCKED != 0;

struct AlsArgs {
    const AlsItem* items;
    const AlsReduce* reduces;
    uint32_t count;  // items (gram kernel) or reduces (reduce kernel)
    const uint32_t* idx;
    const float* val;
    const float* X;    // [x_rows + 1][k]: row x_rows is all zeros (gather target of positions past a segment's end)
    uint32_t x_rows;
    uint32_t sentinel;  // idx[sentinel] = x_rows, val[sentinel] = 0: what a position past a segment's end loads
    float* Y;
    uint32_t k;
    float lambda;
    float* ws;
    uint32_t* spd_fail;
    float* gram_out;  // != nullptr: dump the k x k Gramian (no lambda) of item 0 and stop
    unsigned long long* phases;  // != nullptr (MFX_ALS_PHASES=1): s_memtime clocks per phase, summed over the waves:
                                 // [0] Gramian loop, [1] staging into LDS, [2] factorisation, [3] triangular solves, [4] systems
};
constexpr uint32_t kPhaseCopies = 1024;
__device__ __forceinline__ void phase_mark(const AlsArgs& a, int slot, unsigned long long& t) {
    if (a.phases && t) {  // (t == 0: a caller that does not take part, e.g. the reducers of split segments)
        const unsigned long long now = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) atomicAdd(a.phases + (blockIdx.x % kPhaseCopies) * 8 + slot, now - t);  // (spread: 480 k waves on one line serialise)
        t = now;
    }
}

template <int NT> struct Tiles { static constexpr int kCount = NT * (NT + 1) / 2; };

template <int NT>
__device__ __forceinline__ size_t slot_floats() { return (size_t) Tiles<NT>::kCount * 1024 + (size_t) NT * 64; }

// LDS image of one system: the lower triangle only, rows packed back to back with every row start
// rounded up to 4 floats (16-B aligned for ds_read_b128): 8.7 KB for k = 64 instead of 17.4 KB for
// the square, which is what lets 16 instead of 9 single-wave workgroups share a CU.  The rhs follows.
constexpr int roff_host(int r) { return 4 * ((r >> 2) + 1) * (2 * (r >> 2) + (r & 3)); }
__device__ __forceinline__ int roff(int r) { const int g = r >> 2, m = r & 3; return 4 * (g + 1) * (2 * g + m); }

// 32x32x2 accumulators (upper block triangle) + rhs -> LDS image.
template <int NT>
__device__ __forceinline__ void stage_tiles32(f32x16 (&acc)[Tiles<NT>::kCount], float (&bacc)[NT], float* lds) {
    constexpr int KP = 32 * NT;
    const uint32_t lane = threadIdx.x & 63, c31 = lane & 31, h = lane >> 5;
    float* L = lds;
    float* bv = lds + roff(KP);
    int ti = 0;
#pragma unroll
    for (int I = 0; I < NT; ++I) {
#pragma unroll
        for (int J = I; J < NT; ++J, ++ti) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = I * 32 + (r & 3) + 8 * (r >> 2) + 4 * (int) h;
                const int col = J * 32 + (int) c31;
                const float x = acc[ti][r];
                if (row >= col) L[roff(row) + col] = x;  // diagonal tiles hold both (r,c) and (c,r): same value
                else L[roff(col) + row] = x;
            }
        }
    }
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        const float t = bacc[I] + __shfl_xor(bacc[I], 32, 64);
        if (h == 0) bv[I * 32 + c31] = t;
    }
}


// ---- Blocked Cholesky on the matrix cores, for KP = 32 * NT with NT >= 2 -----------------------------------
// The packed lower-triangular LDS image is factored block column by block column (32 x 32 blocks):
//   update   T_IJ = A_IJ - sum_{K<J} L_IK L_JK^T   v_mfma_f32_32x32x2_f32: lane (r, h) feeds row r of both blocks,
//            four consecutive columns per ds_read_b128 (columns 8t+4h+e in MFMA step 4t+e: the two halves of the
//            wave cover the K dimension between them)
//   panel    columns of block column J, one at a time, for 64 rows per pass: every lane keeps its row of the
//            panel in registers (the k <= 64 scheme: pivot row by LDS broadcast, packed fp32 math) and computes
//            L[row][i] = (T[row][i] - sum_{q<i} L[i][q] L[row][q]) / p_i.  First pass: lanes 0..31 hold the
//            diagonal block (they produce the pivots), lanes 32..63 the block below it -- its triangular solve
//            is the very same update; further passes take two more blocks each, with the pivots read back.
// Against the row-by-row form this takes the O(k^3) part off the VALU / LDS path (which bounded k > 64: every
// product needed two LDS rows, 200 ms per iteration at k = 128).  Rows / columns k .. KP-1 of the image are an
// identity block (set by the caller).
template <bool DIAG>
__device__ __forceinline__ void chol_panel_pass(float* __restrict__ L, int J, int blk_lo, int blk_hi, bool& spd_ok) {
    const int lane = (int) (threadIdx.x & 63), r31 = lane & 31, h = lane >> 5;
    const bool stores = h == 0 || blk_hi >= 0;                    // lanes 32..63 without a block of their own shadow blk_lo
    const int row = ((h && blk_hi >= 0) ? blk_hi : blk_lo) * 32 + r31;
    float* blk = L + roff(row) + J * 32;
    f32x2 r2[16];
#pragma unroll
    for (int q = 0; q < 32; q += 4) {  // (diagonal block: reads past the diagonal stay inside the image, never used)
        const f32x4 x = *reinterpret_cast<const f32x4*>(blk + q);
        r2[q / 2] = x.lo;
        r2[q / 2 + 1] = x.hi;
    }
    auto rl = [](float x, int src_lane) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
    };
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float* pivrow = L + roff(J * 32 + i) + J * 32;  // row i of L_JJ: the same address in every lane
        f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q + 4 <= i; q += 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(pivrow + q);
            s01 = fma2(x.lo, r2[q / 2], s01);
            s23 = fma2(x.hi, r2[q / 2 + 1], s23);
        }
#pragma unroll
        for (int q = i & ~3; q < i; ++q) s01.x = __builtin_fmaf(pivrow[q], r2[q / 2][q & 1], s01.x);
        const f32x2 s4 = s01 + s23;  // one v_pk_add_f32, then the two halves
        const float sum = r2[i / 2][i & 1] - (s4.x + s4.y);
        float lji;
        if constexpr (DIAG) {
            const float piv = rl(sum, i);  // lane i < 32 owns the diagonal entry
            spd_ok = spd_ok && piv > 0.f;
            lji = sum * __builtin_amdgcn_rsqf(piv);  // lane i: pivot / sqrt(pivot) = the diagonal entry (see factor_solve, k <= 64)
            if (stores && (h || lane >= i)) blk[i] = lji;
        } else {
            lji = sum * __builtin_amdgcn_rcpf(pivrow[i]);  // (1 ulp, like the rsq of the diagonal pass)
            if (stores) blk[i] = lji;
        }
        r2[i / 2][i & 1] = lji;  // (diagonal block, lanes above the pivot: a slot they never read)
    }
}

// (r3) The diagonal pass RIGHT-LOOKING and entirely in registers.  In the left-looking form above step i reads row i of
// the diagonal block from LDS -- a row whose entries the previous steps have only just written there: an LDS write ->
// read round trip inside every one of the 32 dependent steps of a pass, with one wave per SIMD and nothing to hide it
// behind.  Here every lane keeps its row of the block column in registers, and once column i is scaled its rank-one
// update is applied to the columns still to come, a_c -= l_i * L[c][i], with L[c][i] taken from lane c by v_readlane:
// 496 readlane + fma pairs per pass instead of 120 ds_read_b128 and 260 v_pk_fma, but the dependent chain of a step is
// readlane(pivot) -> rsq -> scale -> readlane -> fma (~50 clocks) and the LDS only sees the 32 column stores.  Lanes
// 32..63 (the block below the diagonal one) run the very same updates on their rows.
#ifndef MFX_ALS_RL
#define MFX_ALS_RL 1
#endif
__device__ __forceinline__ void chol_diag_pass_rl(float* __restrict__ L, int J, int blk_hi, bool& spd_ok) {
    const int lane = (int) (threadIdx.x & 63), r31 = lane & 31, h = lane >> 5;
    const bool stores = h == 0 || blk_hi >= 0;                    // lanes 32..63 without a block of their own shadow the diagonal block
    const int row = ((h && blk_hi >= 0) ? blk_hi : J) * 32 + r31;
    float* blk = L + roff(row) + J * 32;
    f32x2 a2[16];  // columns (2 q, 2 q + 1) of the lane's row: the rank-one updates run as v_pk_fma_f32 on aligned pairs
#pragma unroll
    for (int q = 0; q < 32; q += 4) {  // (diagonal block: reads past the diagonal stay inside the image, never used)
        const f32x4 x = *reinterpret_cast<const f32x4*>(blk + q);
        a2[q / 2] = x.lo;
        a2[q / 2 + 1] = x.hi;
    }
    auto rl = [](float x, int src_lane) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
    };
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float piv = rl(a2[i / 2][i & 1], i);  // lane i < 32 owns the diagonal entry, all earlier rank-one updates applied
        spd_ok = spd_ok && piv > 0.f;
        const float l = a2[i / 2][i & 1] * __builtin_amdgcn_rsqf(piv);  // lane i: pivot / sqrt(pivot) = the diagonal entry
        if (stores && (h || lane >= i)) blk[i] = l;
        const f32x2 nl = {-l, -l};
        if ((i & 1) == 0) a2[i / 2][1] = __builtin_fmaf(-l, rl(l, i + 1), a2[i / 2][1]);  // the odd partner of an even column
#pragma unroll
        for (int c = (i | 1) + 1; c < 32; c += 2)  // (rows above the diagonal: slots they never store)
            a2[c / 2] = fma2(nl, f32x2{rl(l, c), rl(l, c + 1)}, a2[c / 2]);
    }
}

template <int NT>
__device__ void chol_blocked(float* __restrict__ L, bool& spd_ok, const AlsArgs& a) {
    const int lane = (int) (threadIdx.x & 63), r31 = lane & 31, h = lane >> 5;
    unsigned long long tsub = a.phases ? __builtin_readcyclecounter() : 0ull;  // [5] MFMA updates, [6] diagonal passes, [7] passes below
#pragma unroll 1
    for (int J = 0; J < NT; ++J) {
        if (J > 0) {
#pragma unroll 1
            for (int I = J; I < NT; ++I) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* rowI = L + roff(I * 32 + r31) + 4 * h;
                const float* rowJ = L + roff(J * 32 + r31) + 4 * h;
#pragma unroll 1
                for (int K = 0; K < J; ++K) {
                    // all eight operand reads of the K-block go out before its first MFMA: one LDS latency per block
                    // instead of four (the wave is alone on its SIMD: nothing else hides them)
                    f32x4 av[4], bv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        av[t] = *reinterpret_cast<const f32x4*>(rowI + K * 32 + 8 * t);
                        bv[t] = *reinterpret_cast<const f32x4*>(rowJ + K * 32 + 8 * t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], bv[t][e], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // accumulator register r of lane l: row (r&3) + 8 (r>>2) + 4 h, column r31
                    const int row = I * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = J * 32 + r31;
                    if (row >= col) L[roff(row) + col] = sub_rn(L[roff(row) + col], acc[r]);
                }
            }
            __syncthreads();
            phase_mark(a, 5, tsub);
        }
        if constexpr (MFX_ALS_RL != 0) chol_diag_pass_rl(L, J, J + 1 < NT ? J + 1 : -1, spd_ok);
        else chol_panel_pass<true>(L, J, J, J + 1 < NT ? J + 1 : -1, spd_ok);
        __syncthreads();
        phase_mark(a, 6, tsub);
#pragma unroll 1
        for (int I0 = J + 2; I0 < NT; I0 += 2) chol_panel_pass<false>(L, J, I0, I0 + 1 < NT ? I0 + 1 : -1, spd_ok);
        __syncthreads();
        phase_mark(a, 7, tsub);
    }
}

// (r3) Triangular solves for 64 < k <= 128 in 32-column blocks, every operand of the 2 x KP dependent steps in
// registers.  The row-by-row form below reads L[lane][i] (forward) / L[i][lane] (backward) from LDS inside each step:
// with one or two waves per SIMD nothing hides that read, and MFX_ALS_PHASES measured 90 000 clocks per system for the
// solves at k = 128 -- more than the Gramian (85 000) or the factorisation (81 000).  Here a block's operands are
// loaded up front -- forward: 32 consecutive entries of the lane's own rows (b128 reads); backward: element `lane` of 32
// consecutive rows (lane-contiguous, conflict-free) -- and a step is scale, v_readlane, masked fma on the UNSCALED
// unknowns (lane i carries z_i * L[i][i] until the end, as in the k <= 64 path).  Lane l owns rows l and l + 64; rows
// k .. KP-1 are identity rows with a zero right-hand side, so no step needs a bound on k.
template <int NT>
__device__ __forceinline__ void solve_blocked(const float* __restrict__ L, const float* __restrict__ bv, const AlsArgs& a, uint32_t seg, int k) {
    constexpr int KP = 32 * NT;
    const int lane = (int) (threadIdx.x & 63);
    const bool has1 = lane + 64 < KP;                 // (KP = 96: lanes 32..63 own no second row)
    const int r1 = has1 ? lane + 64 : KP - 1;         // ... they shadow the last row and never store
    float z0 = lane < k ? bv[lane] : 0.f;
    float z1 = (has1 && lane + 64 < k) ? bv[lane + 64] : 0.f;
    const float rp0 = rcp_nr(L[roff(lane) + lane]);
    const float rp1 = rcp_nr(L[roff(r1) + r1]);
    float lanef = (float) lane;
    asm volatile("" : "+v"(lanef));  // (opaque: keeps the masks float compares, see factor_solve k <= 64)
    auto rl = [](float x, int src_lane) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
    };
    // ---- forward: L z = b
#pragma unroll
    for (int B = 0; B < NT; ++B) {
        float a0[32], a1[32];
#pragma unroll
        for (int q = 0; q < 32; q += 4) {  // entries past a row's diagonal: inside the image, masked below
            const f32x4 x = *reinterpret_cast<const f32x4*>(L + roff(lane) + 32 * B + q);
            const f32x4 y = *reinterpret_cast<const f32x4*>(L + roff(r1) + 32 * B + q);
            a0[q] = x[0]; a0[q + 1] = x[1]; a0[q + 2] = x[2]; a0[q + 3] = x[3];
            a1[q] = y[0]; a1[q + 1] = y[1]; a1[q + 2] = y[2]; a1[q + 3] = y[3];
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int i = 32 * B + t;
            if (i < 64) {
                const float zi = rl(z0 * rp0, i);
                z0 = lanef > (float) i ? __builtin_fmaf(-a0[t], zi, z0) : z0;
                z1 = __builtin_fmaf(-a1[t], zi, z1);                      // rows 64 .. are all below row i
            } else {
                const float zi = rl(z1 * rp1, i - 64);
                z1 = lanef > (float) (i - 64) ? __builtin_fmaf(-a1[t], zi, z1) : z1;
            }
        }
    }
    z0 *= rp0;
    z1 *= rp1;
    // ---- backward: L^T y = z
#pragma unroll
    for (int B = NT - 1; B >= 0; --B) {
        float a0[32], a1[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {  // element `lane` (and lane + 64) of rows 32 B + t: contiguous over the lanes
            const float* row = L + roff(32 * B + t);
            a0[t] = row[lane];
            a1[t] = row[r1];
        }
#pragma unroll
        for (int t = 31; t >= 0; --t) {
            const int i = 32 * B + t;
            if (i >= 64) {
                const float yi = rl(z1 * rp1, i - 64);
                z1 = lanef < (float) (i - 64) ? __builtin_fmaf(-a1[t], yi, z1) : z1;
                z0 = __builtin_fmaf(-a0[t], yi, z0);                      // rows 0 .. 63 are all above row i
            } else {
                const float yi = rl(z0 * rp0, i);
                z0 = lanef < (float) i ? __builtin_fmaf(-a0[t], yi, z0) : z0;
            }
        }
    }
    z0 *= rp0;
    z1 *= rp1;
    float* y = a.Y + (size_t) seg * k;
    if (lane < k) y[lane] = z0;
    if (has1 && lane + 64 < k) y[lane + 64] = z1;
}

// LDS image -> + lambda, Cholesky, two triangular solves, Y[seg] <- solution.
// FULL: k == KP known at compile time (k = 64: no per-column `i < k` branches, no `lane < k` masks; user half at the
// Netflix shape 8.24 -> 7.89 ms)
template <int NT, bool FULL = false>
__device__ void factor_solve(float* lds, const AlsArgs& a, uint32_t seg, unsigned long long tmark = 0) {
    constexpr int KP = 32 * NT;
    const uint32_t lane = threadIdx.x & 63;
    const int k = FULL ? KP : (int) a.k;
    float* L = lds;
    float* bv = lds + roff(KP);
    __syncthreads();
    if (a.gram_out) {
        for (int e = (int) lane; e < k * k; e += 64) {
            int r = e / k, c = e % k;
            if (FULL) { r = 16 * (r & 3) + (r >> 2); c = 16 * (c & 3) + (c >> 2); }  // the permuted image of stage_tiles16_perm
            a.gram_out[e] = r >= c ? L[roff(r) + c] : L[roff(c) + r];
        }
        return;
    }
    for (int i = (int) lane; i < KP; i += 64) L[roff(i) + i] = i < k ? add_rn(L[roff(i) + i], a.lambda) : 1.0f;  // rows k.. : identity
    __syncthreads();

    // Left-looking Cholesky on the lower triangle, row i at a time (the reference's choldc1 loop,
    // src/ALS.cpp:6-23):  sum = A[i][j] - sum_q L[i][q] * L[j][q];  j == i: p = sqrt(sum);  else
    // L[j][i] = sum / p.  The dot product over q runs in four independent partial sums (the
    // reference's single accumulator would be a 64-deep dependent chain per row).  k <= 64 and the blocked form use
    // fused multiply-adds and one 1/sqrt(pivot) scale per column; the legacy k > 64 row-by-row form keeps the
    // unfused arithmetic of round 1.
    phase_mark(a, 1, tmark);  // staging (+ lambda, barriers)
    if constexpr (NT >= 3 && kBlockedCholesky) {  // (measured at k = 64: 16.9 ms per iteration blocked vs 15.9 in registers)
        bool spd_ok = true;
        chol_blocked<NT>(L, spd_ok, a);
        if (lane == 0 && !spd_ok) atomicAdd(a.spd_fail, 1u);
        phase_mark(a, 2, tmark);
    } else if constexpr (NT <= 2) {
        // k <= 64: lane j keeps its own row j in registers (static indices after full unrolling), so
        // only row i -- the same for every lane -- is read from LDS, as broadcast ds_read_b128 of
        // whole 4-column groups; the up to three columns past the last whole group come from lane
        // i's registers by v_readlane.  Each finished column goes back to LDS with one ds_write_b32,
        // so later rows find it there and the image is complete for the triangular solves.  Against
        // reading both rows from LDS this halves the LDS traffic (which bounded the user half-sweep,
        // 480 k systems) and drops the per-lane address arithmetic.
        // Register pairs and explicit 2-wide fused products: v_pk_fma_f32 on naturally aligned pairs (left to
        // itself the SLP vectoriser pairs non-adjacent columns and pays for it in v_mov).
        f32x2 r2[KP / 2];
        const int row = (int) lane < KP ? (int) lane : KP - 1;  // KP = 32: the upper half-wave mirrors row 31, never stores
#pragma unroll
        for (int q = 0; q < KP; q += 4) {  // reads past the end of a short row stay inside L; those slots are never used
            const f32x4 x = *reinterpret_cast<const f32x4*>(&L[roff(row) + q]);
            r2[q / 2] = x.lo;
            r2[q / 2 + 1] = x.hi;
        }
        auto rl = [](float x, int src_lane) {
            return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
        };
        // Per column i: dot products as v_pk_fma_f32 (two columns per instruction, fused: one rounding less than the
        // reference's multiply-then-add), then ONE scale for the whole column: L[j][i] = sum_j * rs with
        // rs = 1/sqrt(pivot) (v_rsq_f32 + a Newton step, computed redundantly by every lane from the broadcast pivot).
        // Lane i's own product pivot * rs is the diagonal entry sqrt(pivot) to within an ulp, so there is no select
        // between "diagonal" and "below", no correctly rounded sqrt and no IEEE division in the loop (they were 21 of
        // the ~45 VALU instructions of a step; the solves below take 1 / L[i][i] once, in parallel).
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            if (i < k) {  // wave-uniform
                f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q + 4 <= i; q += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(&L[roff(i) + q]);
                    s01 = fma2(x.lo, r2[q / 2], s01);
                    s23 = fma2(x.hi, r2[q / 2 + 1], s23);
                }
#pragma unroll
                for (int q = i & ~3; q < i; ++q) s01.x = __builtin_fmaf(rl(r2[q / 2][q & 1], i), r2[q / 2][q & 1], s01.x);
                const f32x2 s4 = s01 + s23;  // one v_pk_add_f32, then the two halves
        const float sum = r2[i / 2][i & 1] - (s4.x + s4.y);
                const float piv = rl(sum, i);
                // v_rsq_f32 as it comes (1 ulp): a column of L scaled by (1 + 1e-7) perturbs L L^T like one more fp32
                // rounding of A's entries; a Newton step here costs four more VALU instructions per column
                const float lji = sum * __builtin_amdgcn_rsqf(piv);
                r2[i / 2][i & 1] = lji;  // lanes j < i: a register slot (column i > j) they never read
                if ((int) lane >= i && (int) lane < KP) L[roff((int) lane) + i] = lji;
            }
        }
        __syncthreads();
        phase_mark(a, 2, tmark);
        // Triangular solves on the UNSCALED unknowns: lane i carries z_i * L[i][i] until the very end, so a step is
        // scale (one multiply for all lanes), broadcast (v_readlane), update (one masked fma) -- no per-step select of
        // the finished component.  The forward pass takes L[lane][i] from the lane's registers, the backward pass reads
        // row i of L from LDS at a compile-time offset (lane-strided, conflict-free).  Rows k .. KP-1 are identity rows
        // with a zero rhs: their updates add exact zeros.
        // (the lane masks of the solves are FLOAT compares on purpose: as integer compares they are the store masks
        // of the factorisation loop above, get computed there, and 128 of them are kept alive across it in VGPR lanes)
        float z = lane < (uint32_t) k ? bv[lane] : 0.f;
        const float rp = lane < (uint32_t) k ? rcp_nr(L[roff((int) lane) + lane]) : 0.f;
        float lanef = (float) lane;
        asm volatile("" : "+v"(lanef));  // (opaque: otherwise the compare is folded back to the integer one)
#pragma unroll
        for (int i = 0; i < KP; ++i) {  // forward: L z = b
            if (i < k) {
                const float zi = rl(z * rp, i);
                z = lanef > (float) i ? __builtin_fmaf(-r2[i / 2][i & 1], zi, z) : z;
            }
        }
        z *= rp;  // = the solution of L z = b; the backward pass carries y_i * L[i][i] the same way
#pragma unroll
        for (int i = KP - 1; i >= 0; --i) {  // backward: L^T y = z
            if (i < k) {
                const float yi = rl(z * rp, i);
                const float lij = L[roff(i) + (int) lane];  // lanes >= i read past the row's diagonal: masked below
                z = lanef < (float) i ? __builtin_fmaf(-lij, yi, z) : z;
            }
        }
        z *= rp;
        // FULL: the system was factored under the column permutation of stage_tiles16_perm (unknown 16 e + c is column 4 c + e)
        if ((int) lane < k) a.Y[(size_t) seg * k + (FULL ? 4 * (lane & 15) + (lane >> 4) : lane)] = z;
        // "a is not positive definite" (src/ALS.cpp:12): a pivot <= 0 or NaN makes its rsq inf / NaN, which reaches every
        // later column and the solution -- one test of the result instead of one compare per pivot; one count per system
        // (the k > 64 form counts pivots)
        const bool broken = (int) lane < k && !(__builtin_fabsf(z) <= 3.0e38f);
        if (__ballot(broken) != 0 && lane == 0) atomicAdd(a.spd_fail, 1u);
        phase_mark(a, 3, tmark);
        return;
    } else {
        // k > 64: rows do not fit the register file next to the accumulators; row i is a broadcast
        // ds_read_b128, row j lane-strided and conflict-free.
        for (int i = 0; i < k; ++i) {
            float p = 0.f;
            for (int j0 = i; j0 < k; j0 += 64) {
                const int j = j0 + (int) lane;
                float sum = 0.f;
                if (j < k) {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    int q = 0;
                    for (; q + 4 <= i; q += 4) {
                        const float4 x = *reinterpret_cast<const float4*>(&L[roff(i) + q]);
                        const float4 y4 = *reinterpret_cast<const float4*>(&L[roff(j) + q]);
                        s0 = add_rn(s0, mul_rn(x.x, y4.x));
                        s1 = add_rn(s1, mul_rn(x.y, y4.y));
                        s2 = add_rn(s2, mul_rn(x.z, y4.z));
                        s3 = add_rn(s3, mul_rn(x.w, y4.w));
                    }
                    for (; q < i; ++q) s0 = add_rn(s0, mul_rn(L[roff(i) + q], L[roff(j) + q]));
                    sum = sub_rn(L[roff(j) + i], add_rn(add_rn(s0, s1), add_rn(s2, s3)));
                }
                if (j0 == i) {  // lane 0 holds the pivot of this row
                    const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sum)));
                    if (lane == 0 && !(piv > 0.f)) atomicAdd(a.spd_fail, 1u);
                    p = sqrtf(piv);
                }
                if (j < k) L[roff(j) + i] = (j == i) ? p : sum / p;
            }
            __syncthreads();
        }
    }
    if constexpr (NT >= 3 && kBlockedCholesky) {
        __syncthreads();
        solve_blocked<NT>(L, bv, a, seg, k);
        phase_mark(a, 3, tmark);
        return;
    }
    // Triangular solves (column oriented; lane r owns row r, two rows per lane for k > 64).  The
    // pivots' reciprocals are taken once, in parallel, so that each of the 2k sequential steps is a
    // broadcast (v_readlane, uniform index), one multiply and one fused update.
    float z0 = lane < (uint32_t) k ? bv[lane] : 0.f;
    float z1 = (NT > 2 && lane + 64 < (uint32_t) k) ? bv[lane + 64] : 0.f;
    const float rp0 = lane < (uint32_t) k ? 1.0f / L[roff((int) lane) + lane] : 0.f;
    const float rp1 = (NT > 2 && lane + 64 < (uint32_t) k) ? 1.0f / L[roff((int) lane + 64) + lane + 64] : 0.f;
    auto bcast = [](float x, int src_lane) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
    };
    for (int i = 0; i < k; ++i) {  // forward: L z = b
        const bool hi = NT > 2 && i >= 64;
        const float zi = bcast(hi ? z1 : z0, i & 63) * bcast(hi ? rp1 : rp0, i & 63);
        if ((int) lane == i) z0 = zi;
        if (NT > 2 && (int) lane + 64 == i) z1 = zi;
        if ((int) lane > i && (int) lane < k) z0 = sub_rn(z0, mul_rn(L[roff((int) lane) + i], zi));
        if (NT > 2 && (int) lane + 64 > i && (int) lane + 64 < k) z1 = sub_rn(z1, mul_rn(L[roff((int) lane + 64) + i], zi));
    }
    for (int i = k - 1; i >= 0; --i) {  // backward: L^T y = z
        const bool hi = NT > 2 && i >= 64;
        const float yi = bcast(hi ? z1 : z0, i & 63) * bcast(hi ? rp1 : rp0, i & 63);
        if ((int) lane == i) z0 = yi;
        if (NT > 2 && (int) lane + 64 == i) z1 = yi;
        if ((int) lane < i) z0 = sub_rn(z0, mul_rn(L[roff(i) + lane], yi));
        if (NT > 2 && (int) lane + 64 < i) z1 = sub_rn(z1, mul_rn(L[roff(i) + lane + 64], yi));
    }
    float* y = a.Y + (size_t) seg * k;
    if ((int) lane < k) y[lane] = z0;
    if (NT > 2 && (int) lane + 64 < k) y[lane + 64] = z1;
    phase_mark(a, 3, tmark);
}



// Waves per SIMD the LDS image allows anyway (k = 128: 34 KB per system -> one wave per SIMD; k = 96: two), stated
// so that the register allocator does not trade the Gramian loop's pipelining for an occupancy it cannot get.
constexpr int als_waves(int NT) { return NT >= 4 ? 1 : NT == 3 ? 2 : NT == 2 ? 3 : 6; }

template <int NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(als_waves(NT), als_waves(NT)))) void k_als_gram(AlsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x & 63, c31 = lane & 31, h = lane >> 5;
    const uint32_t item = blockIdx.x;
    if (item >= a.count) return;
    const AlsItem it = a.items[item];
    const uint32_t k = a.k;
    if (it.hi == it.lo) {  // empty segment: zero vector (src/ALS.cpp:151-157)
        for (uint32_t c = lane; c < k; c += 64) a.Y[(size_t) it.seg * k + c] = 0.f;
        return;
    }
    f32x16 acc[Tiles<NT>::kCount];
    float bacc[NT];
    unsigned long long tmark = a.phases ? __builtin_readcyclecounter() : 0ull;
#pragma unroll
    for (int t = 0; t < Tiles<NT>::kCount; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int I = 0; I < NT; ++I) bacc[I] = 0.f;

    constexpr int U = NT >= 3 ? 4 : 8;  // gathered row pairs per step
    if constexpr (NT >= 3) {
        // The pipeline of k_als_gram16 in its generic form (64-bit addressing: this kernel also takes the gather tables
        // that one cannot): two register sets used alternately by consecutive steps of 2 U entries, no copies -- MFMAs of
        // step s on one set while the factor rows (and ratings) of step s + 1 load into the other and the indices of step
        // s + 2 load behind them.  Positions past the segment's end load the SENTINEL entry of the index / value arrays
        // (zero row, rating 0); columns past k gather the zero row.  (As a compiler-scheduled loop with "next" and
        // "current" arrays the copies between the two forced `s_waitcnt vmcnt(0)` on the loads just issued in front of
        // every MFMA block: at one wave per SIMD, k = 128, the whole gather latency of every step lay open.)
        const uint32_t zrow = (uint32_t) __builtin_amdgcn_readfirstlane((int) a.x_rows);
        uint32_t ix[2][U];
        float rv[2][U];
        float av[2][U][NT];
        auto load_idx = [&](auto S, uint32_t q0) {
            constexpr int s = decltype(S)::value;
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + 2 * u + h;
                ix[s][u] = a.idx[q < it.hi ? q : a.sentinel];
            }
        };
        auto load_rows = [&](auto S, uint32_t q0) {
            constexpr int s = decltype(S)::value;
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t row = ix[s][u];
                asm volatile("" : "+v"(row));  // opaque use: keeps the index load where it was issued (see g16_load_rows)
                const uint32_t q = q0 + 2 * u + h;
                rv[s][u] = a.val[q < it.hi ? q : a.sentinel];
    #pragma unroll
                for (int I = 0; I < NT; ++I) {
                    const uint32_t col = I * 32 + c31;
                    const bool in = col < k;
                    av[s][u][I] = a.X[(size_t) (in ? row : zrow) * k + (in ? col : 0u)];
                }
            }
        };
        auto mfmas = [&](auto S) {
            constexpr int s = decltype(S)::value;
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                int ti = 0;
    #pragma unroll
                for (int I = 0; I < NT; ++I) {
                    bacc[I] += rv[s][u] * av[s][u][I];
    #pragma unroll
                    for (int J = I; J < NT; ++J, ++ti)
                        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][u][I], av[s][u][J], acc[ti], 0, 0, 0);
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        load_idx(S0{}, it.lo);
        load_idx(S1{}, it.lo + 2 * U);
        load_rows(S0{}, it.lo);
        for (uint32_t q0 = it.lo;;) {
            load_rows(S1{}, q0 + 2 * U);
            load_idx(S0{}, q0 + 4 * U);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(S0{});
            __builtin_amdgcn_sched_barrier(0);
            q0 += 2 * U;
            if (q0 >= it.hi) break;
            load_rows(S0{}, q0 + 2 * U);
            load_idx(S1{}, q0 + 4 * U);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(S1{});
            __builtin_amdgcn_sched_barrier(0);
            q0 += 2 * U;
            if (q0 >= it.hi) break;
        }
    } else {
        // k <= 64 (3 to 6 waves per SIMD): the compiler-scheduled form of the same pipeline, which is the faster
        // one there (k = 32: 7.9 ms per iteration against 9.0 ms for the explicit two-set form)
        // Same pipeline as k_als_gram16: indices / ratings two batches ahead, factor rows one batch ahead,
        // every load unconditional -- positions past the segment's end and columns past k gather from the
        // all-zero row X[x_rows].
        // A position past the segment's end loads the SENTINEL entry of the index / value arrays (the zero row, rating 0):
        // a select between two positions of one array.  Written as `q < hi ? idx[q] : x_rows` the conditional is folded
        // into a select between two ADDRESSES (the index array, the slot holding x_rows) feeding one flat_load -- which
        // counts on lgkmcnt as well as vmcnt, so every batch began with `s_waitcnt vmcnt(0) lgkmcnt(0)`: all loads in
        // flight drained, the one issued two instructions earlier included (58 % of the MFMA rate).
        const uint32_t zrow = (uint32_t) __builtin_amdgcn_readfirstlane((int) a.x_rows);
        uint32_t row_n[U];
        float rv_n[U], rv_c[U];
        float av_n[U][NT];
        auto load_idx = [&](uint32_t q0) {
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + 2 * u + h;
                const uint32_t qe = q < it.hi ? q : a.sentinel;
                row_n[u] = a.idx[qe];
                rv_n[u] = a.val[qe];
            }
        };
        auto load_rows = [&]() {
    #pragma unroll
            for (int u = 0; u < U; ++u) {
    #pragma unroll
                for (int I = 0; I < NT; ++I) {
                    const uint32_t col = I * 32 + c31;
                    const bool in = col < k;
                    av_n[u][I] = a.X[(size_t) (in ? row_n[u] : zrow) * k + (in ? col : 0u)];
                }
                rv_c[u] = rv_n[u];
            }
        };
        load_idx(it.lo);
        load_rows();
        load_idx(it.lo + 2 * U);
        for (uint32_t q0 = it.lo; q0 < it.hi; q0 += 2 * U) {
            float av[U][NT], rv[U];
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                rv[u] = rv_c[u];
    #pragma unroll
                for (int I = 0; I < NT; ++I) av[u][I] = av_n[u][I];
            }
            load_rows();
            load_idx(q0 + 4 * U);
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                int ti = 0;
    #pragma unroll
                for (int I = 0; I < NT; ++I) {
                    bacc[I] += rv[u] * av[u][I];
    #pragma unroll
                    for (int J = I; J < NT; ++J, ++ti)
                        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][I], av[u][J], acc[ti], 0, 0, 0);
                }
            }
        }
    }
    if (it.slot >= 0) {  // chunk of a long segment: park the raw accumulators, the reducer finishes
        float* w = a.ws + (size_t) it.slot * slot_floats<NT>();
#pragma unroll
        for (int t = 0; t < Tiles<NT>::kCount; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) w[t * 1024 + r * 64 + lane] = acc[t][r];
#pragma unroll
        for (int I = 0; I < NT; ++I) w[Tiles<NT>::kCount * 1024 + I * 64 + lane] = bacc[I];
        return;
    }
    phase_mark(a, 0, tmark);
    if (a.phases && lane == 0) atomicAdd(a.phases + (blockIdx.x % kPhaseCopies) * 8 + 4, 1ull);
    stage_tiles32<NT>(acc, bacc, lds);
    factor_solve<NT>(lds, a, it.seg, tmark);
}

template <int NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(als_waves(NT), als_waves(NT)))) void k_als_reduce(AlsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x & 63;
    if (blockIdx.x >= a.count) return;
    const AlsReduce rd = a.reduces[blockIdx.x];
    f32x16 acc[Tiles<NT>::kCount];
    float bacc[NT];
#pragma unroll
    for (int t = 0; t < Tiles<NT>::kCount; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int I = 0; I < NT; ++I) bacc[I] = 0.f;
    for (uint32_t s = 0; s < rd.nslots; ++s) {  // chunk order: deterministic
        const float* w = a.ws + (size_t) (rd.slot0 + s) * slot_floats<NT>();
#pragma unroll
        for (int t = 0; t < Tiles<NT>::kCount; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += w[t * 1024 + r * 64 + lane];
#pragma unroll
        for (int I = 0; I < NT; ++I) bacc[I] += w[Tiles<NT>::kCount * 1024 + I * 64 + lane];
    }
    stage_tiles32<NT>(acc, bacc, lds);
    factor_solve<NT>(lds, a, rd.seg);
}

// ---- 32 < k <= 64, k % 4 == 0: 16x16x4 tiles fed by 16-byte gathers ---------------------------------
// Lane l = (g = l >> 4, c = l & 15) loads X[row(q0 + g)][4c .. 4c+3] with ONE global_load_dwordx4: a
// wave instruction fetches four whole factor rows (1 KB) instead of two half rows (256 B), which is
// what the L2 / Infinity Cache gather rate wants.  Register e of that float4 is column 4c + e; taken
// as the A (and B) operand of v_mfma_f32_16x16x4_f32 (lane supplies A[i = c][kk = g]) it is "column
// set e", so tile (e, e') accumulates G[4c + e][4c' + e'] -- the Gramian under a fixed column
// permutation that stage_tiles16 undoes on the way to LDS.  10 of 16 tiles (upper triangle of the
// 4 x 4 set grid) = 320 MFMA cycles per 4 rows, against 384 for 3 of 4 32x32 tiles.
constexpr int kSets = 4, kTiles16 = kSets * (kSets + 1) / 2;

__device__ __forceinline__ void stage_tiles16(f32x4 (&acc)[kTiles16], float (&bacc)[kSets], float* lds) {
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* L = lds;
    float* bv = lds + roff(64);
    int ti = 0;
#pragma unroll
    for (int e = 0; e < kSets; ++e) {
#pragma unroll
        for (int f = e; f < kSets; ++f, ++ti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // accumulator register r of lane l is C[i = 4g + r][j = c]
                const int row = 4 * (4 * (int) g + r) + e;
                const int col = 4 * (int) c + f;
                const float x = acc[ti][r];
                if (row >= col) L[roff(row) + col] = x;
                else L[roff(col) + row] = x;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < kSets; ++e) {  // rhs: the four row groups hold partial sums of the same column
        float t = bacc[e] + __shfl_xor(bacc[e], 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (g == 0) bv[4 * c + e] = t;
    }
}

// k = 64: the same image under the symmetric permutation "column 4 c + e -> 16 e + c", i.e. in the order the MFMA tiles
// hold it.  Tile (e, f), e <= f, of lane (c, g) holds G'[16 e + 4 g + r][16 f + c], r = 0..3: mirrored, four CONSECUTIVE
// entries of row 16 f + c of the packed lower triangle -- one ds_write_b128 per tile, ten in all, where the natural order
// needs forty ds_write_b32 with a row-or-column select each (a Cholesky factorisation is as good under one symmetric
// permutation as under another; the solution is written back through the inverse permutation).  Diagonal tiles: the
// lanes whose four entries lie (at least partly) on or below the diagonal of their row write, g <= c >> 2; what reaches
// past the diagonal lands in the row's alignment padding; the entries of the other lanes are the mirror images of those.
__device__ __forceinline__ void stage_tiles16_perm(f32x4 (&acc)[kTiles16], float (&bacc)[kSets], float* lds) {
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* L = lds;
    float* bv = lds + roff(64);
    int ti = 0;
#pragma unroll
    for (int e = 0; e < kSets; ++e) {
#pragma unroll
        for (int f = e; f < kSets; ++f, ++ti) {
            float* dst = L + roff(16 * f + (int) c) + 16 * e + 4 * (int) g;
            if (f > e || g <= (c >> 2)) *reinterpret_cast<f32x4*>(dst) = acc[ti];
        }
    }
#pragma unroll
    for (int e = 0; e < kSets; ++e) {  // rhs: the four row groups hold partial sums of the same column
        float t = bacc[e] + __shfl_xor(bacc[e], 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (g == 0) bv[16 * e + c] = t;
    }
}

// Pipeline state of k_als_gram16: D register sets, used in turn by consecutive 16-row steps (no copies).
// (r4) two 4-row groups per step (8 gathered rows, 2 KB per set) instead of four: the loop's register sets halve, and a tail-dominated
// launch then fits FOUR waves per SIMD (128 VGPRs, 44 bytes of scratch per lane) -- user half of the Netflix shape 7.67 -> 7.26 ms, iteration
// 12.80 -> 12.30 ms; (waves, groups) = (3, 2) 12.80, (4, 4) 12.46, (5, 2) 21.3 (264 bytes of scratch), (4, 1) 13.3 (tools/exp_als_libs.sh)
#ifndef MFX_G16_U
#define MFX_G16_U 2
#endif
constexpr int kU16 = MFX_G16_U;  // 4-row MFMA groups per step
constexpr uint32_t kRows16 = 4u * kU16;
template <int D>
struct Gram16Regs {
    uint32_t ix[D][kU16];  // gathered row indices           (stage 0: loaded D steps ahead of their MFMAs)
    f32x4 av[D][kU16];     // gathered factor-row quarters   (stage 1: D - 1 steps ahead)
    float rv[D][kU16];     // ratings                        (stage 1)
    f32x4 acc[kTiles16];
    f32x2 bacc[2];         // rhs partial sums of column sets (0, 1) and (2, 3)
};

// The three stages of one 16-row step s of a work item (entries lo + 16 s + 4 u + g, u = 0..3), each on register
// set S.  Every load is unconditional and has a wave-uniform base (SGPR pair: the item's first entry, advanced by
// the scalar unit) plus a lane-constant 32-bit offset plus an immediate -- no per-load address arithmetic on the
// vector unit.  Positions past the item's end read on into the next segment's entries (or the arrays' zero padding,
// AlsHalf::build); their ROW OFFSET is replaced by the all-zero row X[x_rows], so they add exact zeros (a rating
// read from past the end multiplies that zero row).
template <int D, int S>
__device__ __forceinline__ void g16_load_idx(Gram16Regs<D>& r, const uint32_t* __restrict__ ibase, uint32_t s, uint32_t g) {
#pragma unroll
    for (int u = 0; u < kU16; ++u) r.ix[S][u] = ibase[s * kRows16 + 4 * u + g];
}
template <int D, int S>
__device__ __forceinline__ void g16_load_rows(Gram16Regs<D>& r, const char* __restrict__ Xb, const float* __restrict__ vbase,
                                              uint32_t s, uint32_t g, uint32_t len, bool col_ok, uint32_t rowbytes,
                                              uint32_t lane_off, uint32_t zero_off) {
#pragma unroll
    for (int u = 0; u < kU16; ++u) {
        const bool ok = col_ok && s * kRows16 + 4 * u + g < len;
        uint32_t ix = r.ix[S][u];
        // opaque use: otherwise the index load (only consumed when `ok`) is sunk out of the previous step into a
        // divergent branch right here, with a full wait behind it
        asm volatile("" : "+v"(ix));
        const uint32_t off = ok ? __umul24(ix, rowbytes) + lane_off : zero_off;  // x_rows < 2^24, table < 4 GB (launch_half)
        r.av[S][u] = *reinterpret_cast<const f32x4*>(Xb + off);
        r.rv[S][u] = vbase[s * kRows16 + 4 * u + g];
    }
}
template <int D, int S>
__device__ __forceinline__ void g16_mfma(Gram16Regs<D>& r) {
#pragma unroll
    for (int u = 0; u < kU16; ++u) {
        // rhs: two v_pk_fma_f32 with the rating duplicated into a register pair BY HAND.  The compiler's own form reads
        // the rating through op_sel from (rating, whatever sits in the odd partner register) -- which the allocator
        // fills with a destination of the loads just issued, and the waitcnt pass then drains every load in flight
        // (s_waitcnt vmcnt(0)) in front of the MFMA block of every second step.
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            0, __builtin_bit_cast(int, r.rv[S][u]), 0xE4 /* quad_perm [0,1,2,3]: a plain copy the optimiser cannot fold */, 0xF, 0xF, false));
        const f32x2 rr = {r.rv[S][u], hi};
        r.bacc[0] = fma2(rr, r.av[S][u].lo, r.bacc[0]);
        r.bacc[1] = fma2(rr, r.av[S][u].hi, r.bacc[1]);
        int ti = 0;
#pragma unroll
        for (int e = 0; e < kSets; ++e)
#pragma unroll
            for (int f = e; f < kSets; ++f, ++ti)
                r.acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.av[S][u][e], r.av[S][u][f], r.acc[ti], 0, 0, 0);
    }
}
struct Gram16Ctx {  // loop-invariant operands of the stages
    const uint32_t* ibase; const float* vbase; const char* Xb;
    uint32_t g, len, rowbytes, lane_off, zero_off;
    bool col_ok;
};
// Steps s, s + 1, ... on sets U, U + 1, ... D - 1: MFMAs of step s on set U, the factor rows of step s + D - 1 into the
// set the previous step has just released, the indices of step s + D into this step's own (already consumed) slots.
// True when the item is finished.  (sched_barrier: left to itself the scheduler sinks the loads of a step down to
// their first use, D - 1 steps later.)
template <int D, int U>
__device__ __forceinline__ bool g16_steps(Gram16Regs<D>& r, const Gram16Ctx& c, uint32_t& s) {
    if constexpr (U < D) {
        g16_load_rows<D, (U + D - 1) % D>(r, c.Xb, c.vbase, s + D - 1, c.g, c.len, c.col_ok, c.rowbytes, c.lane_off, c.zero_off);
        g16_load_idx<D, U>(r, c.ibase, s + D, c.g);
        __builtin_amdgcn_sched_barrier(0);
        g16_mfma<D, U>(r);
        __builtin_amdgcn_sched_barrier(0);
        if (++s * kRows16 >= c.len) return true;
        return g16_steps<D, U + 1>(r, c, s);
    } else {
        return false;
    }
}
template <int D, int U>
__device__ __forceinline__ void g16_prologue(Gram16Regs<D>& r, const Gram16Ctx& c) {
    if constexpr (U < D) {
        g16_load_idx<D, U>(r, c.ibase, U, c.g);
        g16_prologue<D, U + 1>(r, c);
        if constexpr (U + 1 < D)  // (after ALL index loads are in flight)
            g16_load_rows<D, U>(r, c.Xb, c.vbase, U, c.g, c.len, c.col_ok, c.rowbytes, c.lane_off, c.zero_off);
    }
}

// Why the loop looks the way it does: with ~100 vector instructions around the 40 MFMAs of a step (64-bit address
// arithmetic per gathered row, clamps and selects per index, register copies between "next" and "current" sets, a
// flat_load born from a select between two addresses) the loop was replaced by this explicit two-set pipeline:
// ~45 vector instructions per step.  tools/ubench_mfma32.hip replays it on L1-resident data: 130 TF of the 157 TF
// fp32 matrix peak at 2.4 GHz; inside the solver (2.2 GHz under load) the Gramian alone runs at 62 % of the matrix
// rate whether the gather is served from HBM, L2 or L1 -- the rest of a half-sweep is the per-system tail.
// WAVES per SIMD: a launch whose items are long (the item half: 2048-row chunks, hardly any tails) is fastest with
// TWO waves per SIMD (5.42 -> 4.91 ms at the Netflix shape), one dominated by per-system tails (the user half, 206
// entries per system) wants the latency hiding of three or four (7.9 ms; 9.2 ms at two; r4: four, see kU16).  Three = 168 VGPRs, no
// spills; four = 128 VGPRs + 18 spilled dwords, same time.
template <int WAVES, int D, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_als_gram16(AlsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const uint32_t item = blockIdx.x;
    if (item >= a.count) return;
    const AlsItem it = a.items[item];
    const uint32_t k = a.k;
    if (it.hi == it.lo) {  // empty segment: zero vector (src/ALS.cpp:151-157)
        for (uint32_t cc = lane; cc < k; cc += 64) a.Y[(size_t) it.seg * k + cc] = 0.f;
        return;
    }
    Gram16Regs<D> r;
    unsigned long long tmark = a.phases ? __builtin_readcyclecounter() : 0ull;
#pragma unroll
    for (int t = 0; t < kTiles16; ++t) r.acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.bacc[0] = r.bacc[1] = f32x2{0.f, 0.f};

    Gram16Ctx cx;
    cx.len = it.hi - it.lo;
    cx.ibase = a.idx + it.lo;
    cx.vbase = a.val + it.lo;
    cx.Xb = reinterpret_cast<const char*>(a.X);
    cx.rowbytes = 4 * k;
    cx.col_ok = 4 * c < k;                 // lanes past column k gather the zero row
    cx.lane_off = 16 * c;
    cx.zero_off = a.x_rows * cx.rowbytes;
    cx.g = g;
    g16_prologue<D, 0>(r, cx);
    for (uint32_t s = 0;;)
        if (g16_steps<D, 0>(r, cx, s)) break;
    if (it.slot >= 0) {  // chunk of a long segment: park the raw accumulators, the reducer finishes
        float* w = a.ws + (size_t) it.slot * slot_floats<2>();
#pragma unroll
        for (int t = 0; t < kTiles16; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[t * 256 + q * 64 + lane] = r.acc[t][q];
#pragma unroll
        for (int e = 0; e < kSets; ++e) w[kTiles16 * 256 + e * 64 + lane] = r.bacc[e >> 1][e & 1];
        return;
    }
    phase_mark(a, 0, tmark);
    if (a.phases && lane == 0) atomicAdd(a.phases + (blockIdx.x % kPhaseCopies) * 8 + 4, 1ull);
    float bacc[kSets] = {r.bacc[0].x, r.bacc[0].y, r.bacc[1].x, r.bacc[1].y};
    if constexpr (FULL) stage_tiles16_perm(r.acc, bacc, lds);
    else stage_tiles16(r.acc, bacc, lds);
    factor_solve<2, FULL>(lds, a, it.seg, tmark);
}

__global__ __launch_bounds__(64) void k_als_reduce16(AlsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x & 63;
    if (blockIdx.x >= a.count) return;
    const AlsReduce rd = a.reduces[blockIdx.x];
    f32x4 acc[kTiles16];
    float bacc[kSets];
#pragma unroll
    for (int t = 0; t < kTiles16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < kSets; ++e) bacc[e] = 0.f;
    for (uint32_t s = 0; s < rd.nslots; ++s) {  // chunk order: deterministic
        const float* w = a.ws + (size_t) (rd.slot0 + s) * slot_floats<2>();
#pragma unroll
        for (int t = 0; t < kTiles16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += w[t * 256 + r * 64 + lane];
#pragma unroll
        for (int e = 0; e < kSets; ++e) bacc[e] += w[kTiles16 * 256 + e * 64 + lane];
    }
    stage_tiles16(acc, bacc, lds);
    factor_solve<2>(lds, a, rd.seg);
}

// (waves per SIMD, pipeline depth) of k_als_gram16 for launches of long items / of tail-dominated items.  Depth:
// measured at the Netflix shape, item half: (2 waves, depth 2) 4.90 ms, (2, 4) 4.93, (2, 6) 4.94, (3, 4) 5.09 -- the
// gather of 25 GB of 256-byte rows from a 123 MB table runs at 5.2 TB/s either way.
#ifndef MFX_G16_WL
#define MFX_G16_WL 2
#define MFX_G16_DL 2
#define MFX_G16_WS 4
#define MFX_G16_DS 2
#endif
int launch_half_16(const AlsArgs& base, uint32_t nitems, uint32_t nreduces, uint64_t nnz, hipStream_t st) {
    const size_t lds_bytes = ((size_t) roff_host(64) + 64) * sizeof(float);
    AlsArgs a = base;
    if (nitems) {
        a.count = nitems;
        // mean entries per work item: long items -> two waves per SIMD, tail-dominated launches -> three
        const bool longs = nnz / nitems >= 1024, full = a.k == 64;
        if (longs && full) hipLaunchKernelGGL((k_als_gram16<MFX_G16_WL, MFX_G16_DL, true>), dim3(nitems), dim3(64), lds_bytes, st, a);
        else if (longs) hipLaunchKernelGGL((k_als_gram16<MFX_G16_WL, MFX_G16_DL, false>), dim3(nitems), dim3(64), lds_bytes, st, a);
        else if (full) hipLaunchKernelGGL((k_als_gram16<MFX_G16_WS, MFX_G16_DS, true>), dim3(nitems), dim3(64), lds_bytes, st, a);
        else hipLaunchKernelGGL((k_als_gram16<MFX_G16_WS, MFX_G16_DS, false>), dim3(nitems), dim3(64), lds_bytes, st, a);
        MFX_HIP(hipGetLastError());
    }
    if (nreduces) {
        a.count = nreduces;
        hipLaunchKernelGGL(k_als_reduce16, dim3(nreduces), dim3(64), lds_bytes, st, a);
        MFX_HIP(hipGetLastError());
    }
    return MFX_OK;
}

template <int NT>
int launch_half_nt(const AlsArgs& base, uint32_t nitems, uint32_t nreduces, hipStream_t st) {
    constexpr int KP = 32 * NT;
    // packed lower triangle (rows rounded up to 4 floats) + rhs: see solve_tail
    const size_t lds_bytes = ((size_t) 4 * (KP / 4 + 1) * (2 * (KP / 4)) + KP) * sizeof(float);
    if (lds_bytes > 48 * 1024) {  // a per-device attribute; setting it again is cheap next to a half-sweep
        MFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_als_gram<NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
        MFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_als_reduce<NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
    }
    AlsArgs a = base;
    if (nitems) {
        a.count = nitems;
        hipLaunchKernelGGL(k_als_gram<NT>, dim3(nitems), dim3(64), lds_bytes, st, a);
        MFX_HIP(hipGetLastError());
    }
    if (nreduces) {
        a.count = nreduces;
        hipLaunchKernelGGL(k_als_reduce<NT>, dim3(nreduces), dim3(64), lds_bytes, st, a);
        MFX_HIP(hipGetLastError());
    }
    return MFX_OK;
}

int launch_half(const AlsArgs& a, uint32_t nitems, uint32_t nreduces, uint64_t nnz, hipStream_t st) {
    const uint32_t nt = (a.k + 31) / 32;
    // 16-byte aligned factor rows; k_als_gram16 forms 32-bit byte offsets into X with a 24-bit multiply
    if (a.k > 32 && a.k <= 64 && a.k % 4 == 0 && a.x_rows < (1u << 24) && ((uint64_t) a.x_rows + 1) * a.k * 4 < (1ull << 32))
        return launch_half_16(a, nitems, nreduces, nnz, st);
    switch (nt) {
        case 1: return launch_half_nt<1>(a, nitems, nreduces, st);
        case 2: return launch_half_nt<2>(a, nitems, nreduces, st);
        case 3: return launch_half_nt<3>(a, nitems, nreduces, st);
        case 4: return launch_half_nt<4>(a, nitems, nreduces, st);
        default: return fail(MFX_ERR_INVALID, "ALS: rank k = %u not supported (1 <= k <= 128)", a.k);
    }
}

constexpr uint32_t kAlsChunk = 2048;  // gathered rows per wavefront before a segment is split
constexpr uint32_t kAlsPad = 128;     // entries behind the index / value arrays (see AlsHalf::build)
// k_als_gram16 at pipeline depth D loads the indices of step s + D while it works on step s: at most 16 (D + 1) + 15
// entries past an item's end
static_assert(kAlsPad >= 16 * ((MFX_G16_DL > MFX_G16_DS ? MFX_G16_DL : MFX_G16_DS) + 2), "index / value padding too short for the pipeline depth");

}  // namespace

size_t als_ws_floats(uint32_t nslots, uint32_t k) {
    const size_t nt = (k + 31) / 32;
    return (size_t) nslots * (nt * (nt + 1) / 2 * 1024 + nt * 64);
}

int AlsHalf::build(uint32_t nseg_, uint64_t nnz_, uint32_t G, const uint32_t* ptr_in, const uint32_t* idx_in,
                   const float* val_in, mfx_memspace space, uint32_t chunk, hipStream_t st) {
    MFX_REQUIRE(nnz_ == 0 || (idx_in && val_in), "null index / value array with %llu non-zeros", (unsigned long long) nnz_);
    nseg = nseg_;
    nnz = nnz_;
    std::vector<uint32_t> hp((size_t) nseg + 1);
    if (space == MFX_DEVICE) MFX_HIP(hipMemcpy(hp.data(), ptr_in, sizeof(uint32_t) * hp.size(), hipMemcpyDeviceToHost));
    else memcpy(hp.data(), ptr_in, sizeof(uint32_t) * hp.size());
    MFX_REQUIRE(hp[0] == 0 && hp[nseg] == nnz, "segment pointer array does not span [0, nnz]");
    std::vector<AlsItem> it;
    std::vector<AlsReduce> rd;
    it.reserve((size_t) nseg + nnz / chunk + 1);
    uint32_t slots = 0;
    for (uint32_t s = 0; s < nseg; ++s) {
        MFX_REQUIRE(hp[s] <= hp[s + 1], "segment pointer array is not monotone at %u", s);
        const uint32_t lo = hp[s], hi = hp[s + 1];
        if (hi - lo <= chunk) {
            it.push_back(AlsItem{s, lo, hi, -1});
        } else {
            const uint32_t pieces = (hi - lo + chunk - 1) / chunk;
            rd.push_back(AlsReduce{s, slots, pieces});
            for (uint32_t c = 0; c < pieces; ++c)
                it.push_back(AlsItem{s, lo + c * chunk, std::min(hi, lo + (c + 1) * chunk), (int32_t) (slots + c)});
            slots += pieces;
        }
    }
    nitems = (uint32_t) it.size();
    nreduces = (uint32_t) rd.size();
    nslots = slots;
    MFX_TRY(ptr.alloc(hp.size())); MFX_TRY(ptr.upload(hp.data(), hp.size(), MFX_HOST, st));
    // kAlsPad extra entries each: entry nnz is (G, 0) = "the all-zero row of X, rating 0", the stand-in of k_als_gram<NT>
    // for positions past a segment's end; the rest is zero padding that k_als_gram16 may read (and ignore) past the
    // last segment
    MFX_TRY(idx.alloc(nnz + kAlsPad)); MFX_TRY(idx.upload(idx_in, nnz, space, st));
    MFX_TRY(val.alloc(nnz + kAlsPad)); MFX_TRY(val.upload(val_in, nnz, space, st));
    MFX_HIP(hipMemsetAsync(idx.get() + nnz, 0, sizeof(uint32_t) * kAlsPad, st));
    MFX_HIP(hipMemsetAsync(val.get() + nnz, 0, sizeof(float) * kAlsPad, st));
    MFX_HIP(hipMemcpyAsync(idx.get() + nnz, &G, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    MFX_TRY(items.alloc(nitems ? nitems : 1)); MFX_TRY(items.upload(it.data(), nitems, MFX_HOST, st));
    MFX_TRY(reduces.alloc(nreduces ? nreduces : 1)); MFX_TRY(reduces.upload(rd.data(), nreduces, MFX_HOST, st));
    MFX_HIP(hipStreamSynchronize(st));
    // the Gramian kernels use idx[q] as a row of X without further checks
    MFX_TRY(check_index_range(idx.get(), nnz, G, "ALS gather index", st));
    return MFX_OK;
}

int als_half_launch(const AlsHalf& h, const float* X, uint32_t x_rows, float* Y, uint32_t k, float lambda, float* ws,
                    uint32_t* spd_fail, hipStream_t st, unsigned long long* phases) {
    AlsArgs a{};
    a.phases = phases;  // MFX_ALS_PHASES=1 (diagnostic): per-phase clocks of the half-sweep kernels, printed by AlsSolver::iterate
    a.items = h.items.get(); a.reduces = h.reduces.get(); a.idx = h.idx.get(); a.val = h.val.get();
    a.X = X; a.x_rows = x_rows; a.sentinel = (uint32_t) h.nnz; a.Y = Y; a.k = k; a.lambda = lambda; a.ws = ws; a.spd_fail = spd_fail; a.gram_out = nullptr;
    return launch_half(a, h.nitems, h.nreduces, h.nnz, st);
}

// ------------------------------------------------------------------------------------------------
int AlsSolver::create(AlsSolver** out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space,
                      const mfx_als_shard* shard) {
    MFX_REQUIRE(out && R && p, "mfx_als_create: null argument");
    std::unique_ptr<AlsSolver> s(new AlsSolver());
    MFX_TRY(s->init(R, T, p, space, shard));
    *out = s.release();
    return MFX_OK;
}

AlsSolver::~AlsSolver() {
    (void) hipSetDevice(device_);
    for (hipEvent_t& e : ev_)
        if (e) (void) hipEventDestroy(e);
    if (st_) {
        (void) hipStreamSynchronize(st_);
        (void) hipStreamDestroy(st_);
    }
}

// Block boundaries of every rank, gathered through the communicator itself: each rank contributes
// its own (lo, hi) into a zeroed vector and a sum all-reduce fills in the rest.
static int gather_bounds(mfx_comm_s* c, int64_t lo, int64_t hi, std::vector<int64_t>* bounds, hipStream_t st) {
    DevBuf<double> d;
    std::vector<double> h((size_t) c->nranks * 2, 0.0);
    h[(size_t) c->rank * 2] = (double) lo;
    h[(size_t) c->rank * 2 + 1] = (double) hi;
    MFX_TRY(d.alloc(h.size()));
    MFX_TRY(d.upload(h.data(), h.size(), MFX_HOST, st));
    MFX_TRY(comm_allreduce_f64(c, d.get(), h.size(), st));
    MFX_HIP(hipMemcpyAsync(h.data(), d.get(), sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    bounds->assign((size_t) c->nranks + 1, 0);
    for (int r = 0; r < c->nranks; ++r) {
        MFX_REQUIRE((int64_t) h[(size_t) r * 2] == (*bounds)[r], "ALS shards are not contiguous in rank order");
        (*bounds)[(size_t) r + 1] = (int64_t) h[(size_t) r * 2 + 1];
    }
    return MFX_OK;
}

int AlsSolver::init(const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space, const mfx_als_shard* shard) {
    MFX_REQUIRE(R->rows > 0 && R->cols > 0 && R->nnz >= 0, "bad matrix shape");
    MFX_REQUIRE(R->rows < (int64_t) 0xFFFFFFFFll && R->cols < (int64_t) 0xFFFFFFFFll &&
                    R->nnz < (int64_t) 0xFFFF0000ll, "matrix exceeds 32-bit index range");
    MFX_REQUIRE(p->k >= 1 && p->k <= 128, "ALS: rank k = %u not supported (1 <= k <= 128)", p->k);
    MFX_REQUIRE(R->csc_col_ptr && R->csr_row_ptr, "null CSR/CSC pointer array");
    p_ = *p;
    device_ = p->device;
    MFX_TRY(use_device(device_));
    MFX_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    for (hipEvent_t& e : ev_) MFX_HIP(hipEventCreate(&e));
    m_ = (uint32_t) R->rows; n_ = (uint32_t) R->cols; k_ = p->k;
    uint32_t lrows = m_, lcols = n_;
    uint64_t nnz_rows = (uint64_t) R->nnz, nnz_cols = (uint64_t) R->nnz;
    if (shard && shard->comm) {
        MFX_REQUIRE(space == MFX_HOST, "sharded ALS takes host pointers");
        MFX_REQUIRE(0 <= shard->row_lo && shard->row_lo <= shard->row_hi && shard->row_hi <= R->rows &&
                        0 <= shard->col_lo && shard->col_lo <= shard->col_hi && shard->col_hi <= R->cols,
                    "bad ALS shard ranges");
        comm_ = shard->comm;
        row_lo_ = (uint32_t) shard->row_lo; col_lo_ = (uint32_t) shard->col_lo;
        lrows = (uint32_t) (shard->row_hi - shard->row_lo); lcols = (uint32_t) (shard->col_hi - shard->col_lo);
        nnz_rows = R->csr_row_ptr[lrows]; nnz_cols = R->csc_col_ptr[lcols];
        global_test_nnz_ = shard->global_test_nnz;
        row_hi_ = (uint32_t) shard->row_hi; col_hi_ = (uint32_t) shard->col_hi;
        // No collective in here (mfx.h, mfx_comm_agree): a rank that fails any check of its own setup must not leave
        // the others inside one.  The block boundaries of the other ranks are gathered by the first iterate() call,
        // which every rank reaches only after mfx_comm_agree reported that everybody's setup succeeded.
    }
    // W-half walks CSR rows with csr_val (src/ALS.cpp:132), H-half walks CSC columns
    MFX_TRY(rows_.build(lrows, nnz_rows, n_, R->csr_row_ptr, R->csr_col_idx, R->csr_val, space, kAlsChunk, st_));
    MFX_TRY(cols_.build(lcols, nnz_cols, m_, R->csc_col_ptr, R->csc_row_idx, R->csc_val, space, kAlsChunk, st_));
    // one extra, all-zero row each: the Gramian kernel gathers it for positions past a segment's end
    MFX_TRY(W_.alloc_zero(((size_t) m_ + 1) * k_, st_));
    MFX_TRY(H_.alloc_zero(((size_t) n_ + 1) * k_, st_));
    MFX_TRY(ws_.alloc(std::max<size_t>(1, als_ws_floats(std::max(rows_.nslots, cols_.nslots), k_))));
    MFX_TRY(spd_fail_.alloc_zero(1, st_));
    if (std::getenv("MFX_ALS_PHASES")) { MFX_TRY(phases_.alloc_zero((size_t) kPhaseCopies * 8, st_)); }
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
    MFX_TRY(rmse_sum_.alloc_zero(1, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    return MFX_OK;
}

// After a half-sweep every rank holds only its own block of the factor it just solved: ONE grouped exchange
// (every owner broadcasts its block inside a single ncclGroupStart / End) makes the replica whole again.
int AlsSolver::exchange(float* X, const std::vector<int64_t>& bounds) {
    MFX_REQUIRE(shards_met_ && bounds.size() == (size_t) comm_->nranks + 1, "ALS exchange without validated shard boundaries");
    std::vector<int64_t> elems(bounds.size());
    for (size_t r = 0; r < bounds.size(); ++r) elems[r] = bounds[r] * (int64_t) k_;
    return comm_allgather_blocks_f32(comm_, X, elems.data(), st_);
}

// First iterate() of a sharded solve: everyone's block boundaries.  Every rank sees the same gathered vector, so a
// partition that is not contiguous in rank order or does not cover the matrix fails on ALL ranks alike.
// The boundaries are gathered into LOCAL vectors and become the solver's only after every check has passed: a failed
// first iterate() (not contiguous / does not cover) must leave the solver in the state "not met" -- round 3 keyed on
// row_bounds_.empty(), which gather_bounds had already filled, so a second iterate() went on to exchange() with
// unvalidated (or, for the columns, missing) boundaries.
int AlsSolver::meet_shards() {
    shards_met_ = false;
    std::vector<int64_t> rb, cb;
    MFX_TRY(gather_bounds(comm_, row_lo_, row_hi_, &rb, st_));
    MFX_TRY(gather_bounds(comm_, col_lo_, col_hi_, &cb, st_));
    const size_t want = (size_t) comm_->nranks + 1;
    MFX_REQUIRE(rb.size() == want && cb.size() == want, "ALS shards: gathered %zu / %zu boundaries for %d ranks", rb.size(), cb.size(), comm_->nranks);
    MFX_REQUIRE(rb.back() == (int64_t) m_ && cb.back() == (int64_t) n_, "ALS shards do not cover the matrix");
    row_bounds_.swap(rb);
    col_bounds_.swap(cb);
    shards_met_ = true;
    return MFX_OK;
}

int AlsSolver::print_phases(const char* what) {
    unsigned long long h[8] = {};
    std::vector<unsigned long long> all((size_t) kPhaseCopies * 8);
    MFX_HIP(hipStreamSynchronize(st_));
    MFX_HIP(hipMemcpy(all.data(), phases_.get(), sizeof(unsigned long long) * all.size(), hipMemcpyDeviceToHost));
    MFX_HIP(hipMemset(phases_.get(), 0, sizeof(unsigned long long) * all.size()));
    for (size_t c = 0; c < kPhaseCopies; ++c)
        for (int q = 0; q < 8; ++q) h[q] += all[c * 8 + q];
    const double n = h[4] ? (double) h[4] : 1.0;
    fprintf(stderr, "[mfx als phases] %-22s systems %llu; s_memtime clocks per system: gramian %.0f, staging %.0f, factorisation %.0f (k > 64: MFMA updates %.0f, "
            "diagonal passes %.0f, passes below %.0f), solves %.0f\n", what, h[4], h[0] / n, h[1] / n, h[2] / n, h[5] / n, h[6] / n, h[7] / n, h[3] / n);
    return MFX_OK;
}

int AlsSolver::set_factors(const float* W, const float* H, mfx_memspace space) {
    // W's initial content is irrelevant (overwritten before its first read, src/ALS.cpp:98-158)
    MFX_REQUIRE(H, "mfx_als_set_factors: H is required");
    MFX_TRY(use_device(device_));
    if (W) MFX_TRY(W_.upload(W, (size_t) m_ * k_, space, st_));
    MFX_TRY(H_.upload(H, (size_t) n_ * k_, space, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    factors_set_ = true;
    return MFX_OK;
}

int AlsSolver::iterate(int n_iter, int with_rmse, mfx_iter_report* reports) {
    MFX_REQUIRE(n_iter >= 0, "n_iter must be >= 0");
    MFX_REQUIRE(factors_set_, "mfx_als_iterate: call mfx_als_set_factors first");
    MFX_TRY(use_device(device_));
    if (comm_ && !shards_met_ && n_iter > 0) MFX_TRY(meet_shards());
    for (int it = 0; it < n_iter; ++it) {
        MFX_HIP(hipMemsetAsync(spd_fail_.get(), 0, sizeof(uint32_t), st_));
        MFX_HIP(hipEventRecord(ev_[0], st_));
        if (p_.schedule == 0)  // as written: the reference's arithmetic, bit for bit (als_exact.hip)
            MFX_TRY(als_half_exact_launch(rows_, H_.get(), W_.get() + (size_t) row_lo_ * k_, k_, p_.lambda, spd_fail_.get(), st_));
        else
            MFX_TRY(als_half_launch(rows_, H_.get(), n_, W_.get() + (size_t) row_lo_ * k_, k_, p_.lambda, ws_.get(),
                                    spd_fail_.get(), st_, phases_.get()));
        if (comm_) MFX_TRY(exchange(W_.get(), row_bounds_));
        MFX_HIP(hipEventRecord(ev_[1], st_));
        if (phases_.size()) MFX_TRY(print_phases("user half (W over H)"));
        if (p_.schedule == 0)
            MFX_TRY(als_half_exact_launch(cols_, W_.get(), H_.get() + (size_t) col_lo_ * k_, k_, p_.lambda, spd_fail_.get(), st_));
        else
            MFX_TRY(als_half_launch(cols_, W_.get(), m_, H_.get() + (size_t) col_lo_ * k_, k_, p_.lambda, ws_.get(),
                                    spd_fail_.get(), st_, phases_.get()));
        if (comm_) MFX_TRY(exchange(H_.get(), col_bounds_));
        MFX_HIP(hipEventRecord(ev_[2], st_));
        if (phases_.size()) MFX_TRY(print_phases("item half (H over W)"));
        double rmse = 0.0, sum = 0.0;
        if (with_rmse && global_test_nnz_ > 0) {
            if (nnz_test_ > 0)
                MFX_TRY(launch_test_sqerr(nnz_test_, t_row_.get(), t_col_.get(), t_val_.get(), W_.get(), H_.get(), m_, n_,
                                          k_, 1, rmse_partials_.get(), kRmseBlocks, rmse_sum_.get(), st_));
            else
                MFX_HIP(hipMemsetAsync(rmse_sum_.get(), 0, sizeof(double), st_));
            if (comm_) MFX_TRY(comm_allreduce_f64(comm_, rmse_sum_.get(), 1, st_));
            MFX_HIP(hipMemcpyAsync(&sum, rmse_sum_.get(), sizeof(double), hipMemcpyDeviceToHost, st_));
        }
        MFX_HIP(hipEventRecord(ev_[3], st_));
        uint32_t bad = 0;
        MFX_HIP(hipMemcpyAsync(&bad, spd_fail_.get(), sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
        MFX_HIP(hipStreamSynchronize(st_));
        if (with_rmse && global_test_nnz_ > 0) rmse = std::sqrt(sum / (double) global_test_nnz_);
        float ms_w = 0.f, ms_h = 0.f, ms_r = 0.f;
        MFX_HIP(hipEventElapsedTime(&ms_w, ev_[0], ev_[1]));
        MFX_HIP(hipEventElapsedTime(&ms_h, ev_[1], ev_[2]));
        MFX_HIP(hipEventElapsedTime(&ms_r, ev_[2], ev_[3]));
        t_half_[0] += ms_w * 1e-3; t_half_[1] += ms_h * 1e-3; n_half_[0]++; n_half_[1]++;
        mfx_iter_report rep;
        rep.rank_time = 0.0;
        rep.update_time = (ms_w + ms_h) * 1e-3;
        rep.rmse = rmse;
        rep.rmse_time = ms_r * 1e-3;
        update_acc_ += rep.update_time;
        ++iter_;
        if (reports) reports[it] = rep;
        // the reference prints this from inside the kernel for every failing pivot (ALS_CUDA.cu:11-13)
        if (bad && p_.verbose && (!comm_ || comm_->rank == 0)) printf(" a is not positive definite! (%u systems or pivots)\n", bad);
        if (p_.verbose && (!comm_ || comm_->rank == 0)) {
            // log line format of cuda_src/ALS_CUDA.cu:360-361
            printf("[-INFO-] iteration num %d \tupdate_time %.4lf|%.4lfs \tRMSE=%lf time:%fs\n", (int) iter_,
                   rep.update_time, update_acc_, rep.rmse, rep.rmse_time);
            fflush(stdout);
        }
    }
    return MFX_OK;
}

int AlsSolver::get_factors(float* W, float* H, mfx_memspace space) {
    MFX_TRY(use_device(device_));
    const hipMemcpyKind kind = space == MFX_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (W) MFX_HIP(hipMemcpyAsync(W, W_.get(), sizeof(float) * (size_t) m_ * k_, kind, st_));
    if (H) MFX_HIP(hipMemcpyAsync(H, H_.get(), sizeof(float) * (size_t) n_ * k_, kind, st_));
    MFX_HIP(hipStreamSynchronize(st_));
    return MFX_OK;
}

int AlsSolver::kernel_times(int cap, const char** names, double* seconds, int64_t* launches) {
    static const char* nm[2] = {"als_half_rows(W over H)", "als_half_cols(H over W)"};
    int n = 0;
    for (int i = 0; i < 2 && n < cap; ++i) {
        if (!n_half_[i]) continue;
        if (names) names[n] = nm[i];
        if (seconds) seconds[n] = t_half_[i];
        if (launches) launches[n] = n_half_[i];
        ++n;
    }
    t_half_[0] = t_half_[1] = 0; n_half_[0] = n_half_[1] = 0;
    return n;
}

// ------------------------------------------------------------------------------------------------
namespace {
struct OpStream {
    hipStream_t st = nullptr;
    ~OpStream() { if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); } }
};
}  // namespace

int als_gramian_op(int64_t cnt, const uint32_t* idx, int64_t nrows_x, const float* X, int64_t k, float* A, int device) {
    MFX_REQUIRE(k <= 128, "ALS: rank k = %lld not supported (1 <= k <= 128)", (long long) k);
    MFX_TRY(use_device(device));
    if (cnt == 0) { memset(A, 0, sizeof(float) * k * k); return MFX_OK; }
    MFX_REQUIRE(cnt <= kAlsChunk, "mfx_als_gramian: at most %u gathered rows per call", kAlsChunk);
    OpStream os;
    MFX_HIP(hipStreamCreateWithFlags(&os.st, hipStreamNonBlocking));
    DevBuf<uint32_t> didx, fail_cnt; DevBuf<float> dval, dX, dY, dA; DevBuf<AlsItem> ditem;
    const uint32_t zrow = (uint32_t) nrows_x;
    MFX_TRY(didx.alloc_zero(cnt + kAlsPad, os.st)); MFX_TRY(didx.upload(idx, cnt, MFX_HOST, os.st));
    MFX_HIP(hipMemcpyAsync(didx.get() + cnt, &zrow, sizeof(uint32_t), hipMemcpyHostToDevice, os.st));
    MFX_TRY(check_index_range(didx.get(), (uint64_t) cnt, (uint32_t) nrows_x, "ALS gather index", os.st));
    MFX_TRY(dval.alloc_zero(cnt + kAlsPad, os.st));
    MFX_TRY(dX.alloc_zero(((size_t) nrows_x + 1) * k, os.st)); MFX_TRY(dX.upload(X, (size_t) nrows_x * k, MFX_HOST, os.st));
    MFX_TRY(dY.alloc_zero(k, os.st)); MFX_TRY(dA.alloc_zero((size_t) k * k, os.st));
    MFX_TRY(fail_cnt.alloc_zero(1, os.st));
    AlsItem it{0, 0, (uint32_t) cnt, -1};
    MFX_TRY(ditem.alloc(1)); MFX_TRY(ditem.upload(&it, 1, MFX_HOST, os.st));
    AlsArgs a{};
    a.items = ditem.get(); a.idx = didx.get(); a.val = dval.get(); a.X = dX.get(); a.x_rows = (uint32_t) nrows_x; a.sentinel = (uint32_t) cnt; a.Y = dY.get();
    a.k = (uint32_t) k; a.lambda = 0.f; a.spd_fail = fail_cnt.get(); a.gram_out = dA.get();
    MFX_TRY(launch_half(a, 1, 0, (uint64_t) cnt, os.st));
    MFX_HIP(hipMemcpyAsync(A, dA.get(), sizeof(float) * k * k, hipMemcpyDeviceToHost, os.st));
    MFX_HIP(hipStreamSynchronize(os.st));
    return MFX_OK;
}

int als_half_op(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx, const float* val,
                int64_t nrows_x, const float* X, float* Y, int64_t k, float lambda, int variant, int device) {
    MFX_REQUIRE(k <= 128, "ALS: rank k = %lld not supported (1 <= k <= 128)", (long long) k);
    MFX_TRY(use_device(device));
    OpStream os;
    MFX_HIP(hipStreamCreateWithFlags(&os.st, hipStreamNonBlocking));
    AlsHalf h;
    MFX_TRY(h.build((uint32_t) nseg, (uint64_t) nnz, (uint32_t) nrows_x, ptr, idx, val, MFX_HOST, kAlsChunk, os.st));
    DevBuf<float> dX, dY, ws; DevBuf<uint32_t> fail_cnt;
    MFX_TRY(dX.alloc_zero(((size_t) nrows_x + 1) * k, os.st)); MFX_TRY(dX.upload(X, (size_t) nrows_x * k, MFX_HOST, os.st));
    MFX_TRY(dY.alloc_zero((size_t) nseg * k, os.st));
    MFX_TRY(ws.alloc(std::max<size_t>(1, als_ws_floats(h.nslots, (uint32_t) k))));
    MFX_TRY(fail_cnt.alloc_zero(1, os.st));
    if (variant == 0) MFX_TRY(als_half_exact_launch(h, dX.get(), dY.get(), (uint32_t) k, lambda, fail_cnt.get(), os.st));
    else MFX_TRY(als_half_launch(h, dX.get(), (uint32_t) nrows_x, dY.get(), (uint32_t) k, lambda, ws.get(), fail_cnt.get(), os.st));
    MFX_HIP(hipMemcpyAsync(Y, dY.get(), sizeof(float) * (size_t) nseg * k, hipMemcpyDeviceToHost, os.st));
    MFX_HIP(hipStreamSynchronize(os.st));
    return MFX_OK;
}

}  // namespace mfx

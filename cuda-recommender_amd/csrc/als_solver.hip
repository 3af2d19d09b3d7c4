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
// (the reference's row-by-row scheme, src/ALS.cpp:6-23, inner dot product in four partial sums),
// then L z = b and L^T y = z instead of the reference's explicit inverse (same solution up to
// rounding; tolerance in the tests).
#include "als_solver.hpp"

#include <algorithm>
#include <cmath>
#include <memory>

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

// Correctly rounded sqrt for normal-range inputs: v_sqrt_f32 (<= 1 ulp) and the two residual checks of
// the compiler's own expansion, without its rescaling of inputs below 2^-96 and its zero / inf /
// NaN pass-through -- a pivot is lambda + a sum of squares; anything else is flagged as an SPD failure
// by the caller.  Same result as sqrtf() for every input the solver can meet, 9 instead of 20 VALU
// instructions, 64 times per system.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float s_up = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.f ? s_dn : s;
    r = r_up > 0.f ? s_up : r;
    return r;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 mul2_rn(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f32x2 add2_rn(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a + b;
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
    float* Y;
    uint32_t k;
    float lambda;
    float* ws;
    uint32_t* spd_fail;
    float* gram_out;  // != nullptr: dump the k x k Gramian (no lambda) of item 0 and stop
};

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
            s01 = add2_rn(s01, mul2_rn(x.lo, r2[q / 2]));
            s23 = add2_rn(s23, mul2_rn(x.hi, r2[q / 2 + 1]));
        }
#pragma unroll
        for (int q = i & ~3; q < i; ++q) s01.x = add_rn(s01.x, mul_rn(pivrow[q], r2[q / 2][q & 1]));
        const float sum = sub_rn(r2[i / 2][i & 1], add_rn(add_rn(s01.x, s01.y), add_rn(s23.x, s23.y)));
        float lji;
        if constexpr (DIAG) {
            const float piv = rl(sum, i);  // lane i < 32 owns the diagonal entry
            spd_ok = spd_ok && piv > 0.f;
            const float pv = sqrt_rn_normal(piv);
            lji = lane == i ? pv : sum / pv;
            if (stores && (h || lane >= i)) blk[i] = lji;
        } else {
            lji = sum / pivrow[i];
            if (stores) blk[i] = lji;
        }
        r2[i / 2][i & 1] = lji;  // (diagonal block, lanes above the pivot: a slot they never read)
    }
}

template <int NT>
__device__ void chol_blocked(float* __restrict__ L, bool& spd_ok) {
    const int lane = (int) (threadIdx.x & 63), r31 = lane & 31, h = lane >> 5;
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
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(rowI + K * 32 + 8 * t);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(rowJ + K * 32 + 8 * t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // accumulator register r of lane l: row (r&3) + 8 (r>>2) + 4 h, column r31
                    const int row = I * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = J * 32 + r31;
                    if (row >= col) L[roff(row) + col] = sub_rn(L[roff(row) + col], acc[r]);
                }
            }
            __syncthreads();
        }
        chol_panel_pass<true>(L, J, J, J + 1 < NT ? J + 1 : -1, spd_ok);
        __syncthreads();
#pragma unroll 1
        for (int I0 = J + 2; I0 < NT; I0 += 2) chol_panel_pass<false>(L, J, I0, I0 + 1 < NT ? I0 + 1 : -1, spd_ok);
        __syncthreads();
    }
}

// LDS image -> + lambda, Cholesky, two triangular solves, Y[seg] <- solution.
template <int NT>
__device__ void factor_solve(float* lds, const AlsArgs& a, uint32_t seg) {
    constexpr int KP = 32 * NT;
    const uint32_t lane = threadIdx.x & 63;
    const int k = (int) a.k;
    float* L = lds;
    float* bv = lds + roff(KP);
    __syncthreads();
    if (a.gram_out) {
        for (int e = (int) lane; e < k * k; e += 64) {
            const int r = e / k, c = e % k;
            a.gram_out[e] = r >= c ? L[roff(r) + c] : L[roff(c) + r];
        }
        return;
    }
    for (int i = (int) lane; i < KP; i += 64) L[roff(i) + i] = i < k ? add_rn(L[roff(i) + i], a.lambda) : 1.0f;  // rows k.. : identity
    __syncthreads();

    // Left-looking Cholesky on the lower triangle, row i at a time (the reference's choldc1 loop,
    // src/ALS.cpp:6-23):  sum = A[i][j] - sum_q L[i][q] * L[j][q];  j == i: p = sqrt(sum);  else
    // L[j][i] = sum / p.  The dot product over q runs in four independent partial sums (the
    // reference's single accumulator would be a 64-deep dependent chain per row); products and sums
    // stay unfused.  Same arithmetic in both variants below, so they agree bit for bit.
    if constexpr (NT >= 3 && kBlockedCholesky) {  // (measured at k = 64: 16.9 ms per iteration blocked vs 15.9 in registers)
        bool spd_ok = true;
        chol_blocked<NT>(L, spd_ok);
        if (lane == 0 && !spd_ok) atomicAdd(a.spd_fail, 1u);
    } else if constexpr (NT <= 2) {
        // k <= 64: lane j keeps its own row j in registers (static indices after full unrolling), so
        // only row i -- the same for every lane -- is read from LDS, as broadcast ds_read_b128 of
        // whole 4-column groups; the up to three columns past the last whole group come from lane
        // i's registers by v_readlane.  Each finished column goes back to LDS with one ds_write_b32,
        // so later rows find it there and the image is complete for the triangular solves.  Against
        // reading both rows from LDS this halves the LDS traffic (which bounded the user half-sweep,
        // 480 k systems) and drops the per-lane address arithmetic.
        // Register pairs and explicit 2-wide products / sums: v_pk_mul_f32 + v_pk_add_f32 on naturally
        // aligned pairs (left to itself the SLP vectoriser pairs non-adjacent columns and pays for it in v_mov).
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
        bool spd_ok = true;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            if (i < k) {  // wave-uniform
                f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q + 4 <= i; q += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(&L[roff(i) + q]);
                    s01 = add2_rn(s01, mul2_rn(x.lo, r2[q / 2]));
                    s23 = add2_rn(s23, mul2_rn(x.hi, r2[q / 2 + 1]));
                }
#pragma unroll
                for (int q = i & ~3; q < i; ++q) s01.x = add_rn(s01.x, mul_rn(rl(r2[q / 2][q & 1], i), r2[q / 2][q & 1]));
                const float sum = sub_rn(r2[i / 2][i & 1], add_rn(add_rn(s01.x, s01.y), add_rn(s23.x, s23.y)));
                const float piv = rl(sum, i);
                spd_ok = spd_ok && piv > 0.f;  // wave-uniform; NaN fails the comparison as well
                const float p = sqrt_rn_normal(piv);
                const float lji = (int) lane == i ? p : sum / p;
                r2[i / 2][i & 1] = lji;  // lanes j < i: a register slot (column i > j) they never read
                if ((int) lane >= i && (int) lane < KP) L[roff((int) lane) + i] = lji;
            }
        }
        if (lane == 0 && !spd_ok) atomicAdd(a.spd_fail, 1u);  // one count per system (the k > 64 form counts pivots)
        __syncthreads();
        // Triangular solves, same arithmetic as the generic loops below but unrolled: the forward pass
        // takes L[lane][i] from the lane's registers, the backward pass reads row i of L from LDS at a
        // compile-time offset (lane-strided, conflict-free); no per-step address arithmetic.
        float z = lane < (uint32_t) k ? bv[lane] : 0.f;
        const float rp = lane < (uint32_t) k ? 1.0f / L[roff((int) lane) + lane] : 0.f;
#pragma unroll
        for (int i = 0; i < KP; ++i) {  // forward: L z = b
            if (i < k) {
                const float zi = rl(z, i) * rl(rp, i);
                const float upd = sub_rn(z, mul_rn(r2[i / 2][i & 1], zi));
                z = (int) lane == i ? zi : ((int) lane > i && (int) lane < k ? upd : z);
            }
        }
#pragma unroll
        for (int i = KP - 1; i >= 0; --i) {  // backward: L^T y = z
            if (i < k) {
                const float yi = rl(z, i) * rl(rp, i);
                const float lij = L[roff(i) + ((int) lane < i ? (int) lane : 0)];
                const float upd = sub_rn(z, mul_rn(lij, yi));
                z = (int) lane == i ? yi : ((int) lane < i ? upd : z);
            }
        }
        if ((int) lane < k) a.Y[(size_t) seg * k + lane] = z;
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
}



template <int NT>
__global__ __launch_bounds__(64) void k_als_gram(AlsArgs a) {
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
#pragma unroll
    for (int t = 0; t < Tiles<NT>::kCount; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int I = 0; I < NT; ++I) bacc[I] = 0.f;

    constexpr int U = NT >= 3 ? 4 : 8;  // gathered row pairs per batch (two batches in flight)
    // Same pipeline as k_als_gram16: indices / ratings two batches ahead, factor rows one batch ahead,
    // every load unconditional -- positions past the segment's end and columns past k gather from the
    // all-zero row X[x_rows].
    const uint32_t last = it.hi - 1;
    uint32_t row_n[U];
    float rv_n[U], rv_c[U];
    float av_n[U][NT];
    auto load_idx = [&](uint32_t q0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 2 * u + h;
            const uint32_t r = a.idx[q < last ? q : last];
            row_n[u] = q < it.hi ? r : a.x_rows;
            rv_n[u] = a.val[q < last ? q : last];
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int I = 0; I < NT; ++I) {
                const uint32_t col = I * 32 + c31;
                const bool in = col < k;
                av_n[u][I] = a.X[(size_t) (in ? row_n[u] : a.x_rows) * k + (in ? col : 0u)];
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
    stage_tiles32<NT>(acc, bacc, lds);
    factor_solve<NT>(lds, a, it.seg);
}

template <int NT>
__global__ __launch_bounds__(64) void k_als_reduce(AlsArgs a) {
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

// 4 waves per SIMD: 125 VGPRs + 16 spilled dwords instead of 132 VGPRs and 3 waves (measured: 16.3 -> 15.6 ms per
// iteration at the Netflix shape; 5 waves = 96 VGPRs + 162 spills: 21.1 ms)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_als_gram16(AlsArgs a) {
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
    f32x4 acc[kTiles16];
    float bacc[kSets];
#pragma unroll
    for (int t = 0; t < kTiles16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < kSets; ++e) bacc[e] = 0.f;

    constexpr int U = 4;  // 4-row steps per batch: 16 gathered rows (4 KB) in flight per wave, twice over
    // Two dependent round trips per batch (index -> factor row), both taken off the critical path:
    // indices and ratings are fetched two batches ahead, factor rows one batch ahead.  Every load is
    // UNCONDITIONAL and nothing is masked afterwards: positions past the segment's end (clamped to its
    // last entry for the index / rating loads) and lanes past column k gather from the all-zero row
    // X[x_rows], so they add exact zeros.  With the loads inside divergent branches -- or selects on
    // the loaded rows -- the compiler either drains the memory queue (s_waitcnt vmcnt(0)) in front of
    // the MFMA block or moves the consumption up to the loads: either way load latency and matrix work
    // serialise within a wave.
    const uint32_t last = it.hi - 1;
    const uint32_t cc = 4 * c < k ? c : 0u;
    const bool col_ok = 4 * c < k;
    uint32_t row_n[U];
    float rv_n[U], rv_c[U];
    f32x4 av_n[U];
    auto load_idx = [&](uint32_t q0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 4 * u + g;
            const uint32_t r = a.idx[q < last ? q : last];
            row_n[u] = (col_ok && q < it.hi) ? r : a.x_rows;
            rv_n[u] = a.val[q < last ? q : last];  // multiplies a zero row when q is past the end
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av_n[u] = *(reinterpret_cast<const f32x4*>(a.X + (size_t) row_n[u] * k) + cc);
            rv_c[u] = rv_n[u];
        }
    };
    load_idx(it.lo);
    load_rows();
    load_idx(it.lo + 4 * U);
    for (uint32_t q0 = it.lo; q0 < it.hi; q0 += 4 * U) {
        f32x4 av[U];
        float rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { av[u] = av_n[u]; rv[u] = rv_c[u]; }
        load_rows();                // rows of the next batch (their indices arrived during the last one)
        load_idx(q0 + 8 * U);       // indices of the batch after that
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ti = 0;
#pragma unroll
            for (int e = 0; e < kSets; ++e) {
                bacc[e] += rv[u] * av[u][e];
#pragma unroll
                for (int f = e; f < kSets; ++f, ++ti)
                    acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], av[u][f], acc[ti], 0, 0, 0);
            }
        }
    }
    if (it.slot >= 0) {  // chunk of a long segment: park the raw accumulators, the reducer finishes
        float* w = a.ws + (size_t) it.slot * slot_floats<2>();
#pragma unroll
        for (int t = 0; t < kTiles16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) w[t * 256 + r * 64 + lane] = acc[t][r];
#pragma unroll
        for (int e = 0; e < kSets; ++e) w[kTiles16 * 256 + e * 64 + lane] = bacc[e];
        return;
    }
    stage_tiles16(acc, bacc, lds);
    factor_solve<2>(lds, a, it.seg);
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

int launch_half_16(const AlsArgs& base, uint32_t nitems, uint32_t nreduces, hipStream_t st) {
    const size_t lds_bytes = ((size_t) roff_host(64) + 64) * sizeof(float);
    AlsArgs a = base;
    if (nitems) {
        a.count = nitems;
        hipLaunchKernelGGL(k_als_gram16, dim3(nitems), dim3(64), lds_bytes, st, a);
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

int launch_half(const AlsArgs& a, uint32_t nitems, uint32_t nreduces, hipStream_t st) {
    const uint32_t nt = (a.k + 31) / 32;
    if (a.k > 32 && a.k <= 64 && a.k % 4 == 0) return launch_half_16(a, nitems, nreduces, st);  // 16-byte aligned factor rows
    switch (nt) {
        case 1: return launch_half_nt<1>(a, nitems, nreduces, st);
        case 2: return launch_half_nt<2>(a, nitems, nreduces, st);
        case 3: return launch_half_nt<3>(a, nitems, nreduces, st);
        case 4: return launch_half_nt<4>(a, nitems, nreduces, st);
        default: return fail(MFX_ERR_INVALID, "ALS: rank k = %u not supported (1 <= k <= 128)", a.k);
    }
}

constexpr uint32_t kAlsChunk = 1024;  // gathered rows per wavefront before a segment is split

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
    MFX_TRY(idx.alloc(nnz ? nnz : 1)); MFX_TRY(idx.upload(idx_in, nnz, space, st));
    MFX_TRY(val.alloc(nnz ? nnz : 1)); MFX_TRY(val.upload(val_in, nnz, space, st));
    MFX_TRY(items.alloc(nitems ? nitems : 1)); MFX_TRY(items.upload(it.data(), nitems, MFX_HOST, st));
    MFX_TRY(reduces.alloc(nreduces ? nreduces : 1)); MFX_TRY(reduces.upload(rd.data(), nreduces, MFX_HOST, st));
    MFX_HIP(hipStreamSynchronize(st));
    // the Gramian kernels use idx[q] as a row of X without further checks
    MFX_TRY(check_index_range(idx.get(), nnz, G, "ALS gather index", st));
    return MFX_OK;
}

int als_half_launch(const AlsHalf& h, const float* X, uint32_t x_rows, float* Y, uint32_t k, float lambda, float* ws,
                    uint32_t* spd_fail, hipStream_t st) {
    AlsArgs a{};
    a.items = h.items.get(); a.reduces = h.reduces.get(); a.idx = h.idx.get(); a.val = h.val.get();
    a.X = X; a.x_rows = x_rows; a.Y = Y; a.k = k; a.lambda = lambda; a.ws = ws; a.spd_fail = spd_fail; a.gram_out = nullptr;
    return launch_half(a, h.nitems, h.nreduces, st);
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
        MFX_TRY(gather_bounds(comm_, shard->row_lo, shard->row_hi, &row_bounds_, st_));
        MFX_TRY(gather_bounds(comm_, shard->col_lo, shard->col_hi, &col_bounds_, st_));
        MFX_REQUIRE(row_bounds_.back() == R->rows && col_bounds_.back() == R->cols, "ALS shards do not cover the matrix");
    }
    // W-half walks CSR rows with csr_val (src/ALS.cpp:132), H-half walks CSC columns
    MFX_TRY(rows_.build(lrows, nnz_rows, n_, R->csr_row_ptr, R->csr_col_idx, R->csr_val, space, kAlsChunk, st_));
    MFX_TRY(cols_.build(lcols, nnz_cols, m_, R->csc_col_ptr, R->csc_row_idx, R->csc_val, space, kAlsChunk, st_));
    // one extra, all-zero row each: the Gramian kernel gathers it for positions past a segment's end
    MFX_TRY(W_.alloc_zero(((size_t) m_ + 1) * k_, st_));
    MFX_TRY(H_.alloc_zero(((size_t) n_ + 1) * k_, st_));
    MFX_TRY(ws_.alloc(std::max<size_t>(1, als_ws_floats(std::max(rows_.nslots, cols_.nslots), k_))));
    MFX_TRY(spd_fail_.alloc_zero(1, st_));
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

// After a half-sweep every rank holds only its own block of the factor it just solved: one
// broadcast per rank (root = owner) makes the replica whole again.
int AlsSolver::exchange(float* X, const std::vector<int64_t>& bounds) {
    for (int r = 0; r < comm_->nranks; ++r) {
        const int64_t lo = bounds[r], hi = bounds[(size_t) r + 1];
        MFX_TRY(comm_broadcast_f32(comm_, X + (size_t) lo * k_, (size_t) (hi - lo) * k_, r, st_));
    }
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
    for (int it = 0; it < n_iter; ++it) {
        MFX_HIP(hipMemsetAsync(spd_fail_.get(), 0, sizeof(uint32_t), st_));
        MFX_HIP(hipEventRecord(ev_[0], st_));
        if (p_.schedule == 0)  // as written: the reference's arithmetic, bit for bit (als_exact.hip)
            MFX_TRY(als_half_exact_launch(rows_, H_.get(), W_.get() + (size_t) row_lo_ * k_, k_, p_.lambda, spd_fail_.get(), st_));
        else
            MFX_TRY(als_half_launch(rows_, H_.get(), n_, W_.get() + (size_t) row_lo_ * k_, k_, p_.lambda, ws_.get(),
                                    spd_fail_.get(), st_));
        if (comm_) MFX_TRY(exchange(W_.get(), row_bounds_));
        MFX_HIP(hipEventRecord(ev_[1], st_));
        if (p_.schedule == 0)
            MFX_TRY(als_half_exact_launch(cols_, W_.get(), H_.get() + (size_t) col_lo_ * k_, k_, p_.lambda, spd_fail_.get(), st_));
        else
            MFX_TRY(als_half_launch(cols_, W_.get(), m_, H_.get() + (size_t) col_lo_ * k_, k_, p_.lambda, ws_.get(),
                                    spd_fail_.get(), st_));
        if (comm_) MFX_TRY(exchange(H_.get(), col_bounds_));
        MFX_HIP(hipEventRecord(ev_[2], st_));
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
    MFX_TRY(didx.alloc(cnt)); MFX_TRY(didx.upload(idx, cnt, MFX_HOST, os.st));
    MFX_TRY(check_index_range(didx.get(), (uint64_t) cnt, (uint32_t) nrows_x, "ALS gather index", os.st));
    MFX_TRY(dval.alloc_zero(cnt, os.st));
    MFX_TRY(dX.alloc_zero(((size_t) nrows_x + 1) * k, os.st)); MFX_TRY(dX.upload(X, (size_t) nrows_x * k, MFX_HOST, os.st));
    MFX_TRY(dY.alloc_zero(k, os.st)); MFX_TRY(dA.alloc_zero((size_t) k * k, os.st));
    MFX_TRY(fail_cnt.alloc_zero(1, os.st));
    AlsItem it{0, 0, (uint32_t) cnt, -1};
    MFX_TRY(ditem.alloc(1)); MFX_TRY(ditem.upload(&it, 1, MFX_HOST, os.st));
    AlsArgs a{};
    a.items = ditem.get(); a.idx = didx.get(); a.val = dval.get(); a.X = dX.get(); a.x_rows = (uint32_t) nrows_x; a.Y = dY.get();
    a.k = (uint32_t) k; a.lambda = 0.f; a.spd_fail = fail_cnt.get(); a.gram_out = dA.get();
    MFX_TRY(launch_half(a, 1, 0, os.st));
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

// als_exact.hip -- ALS half-sweep "as written" (mfx_params.schedule = 0): the reference's arithmetic,
// operation for operation and in its order, so that the result equals the CPU reference BIT FOR BIT
// (tests/test_gpu_als.py compares with the golden fixtures that way).  It is the parity mode of the ALS
// path, the counterpart of CCD++'s schedule 0 -- not the fast path (als_solver.hip: MFMA Gramian,
// Cholesky solve) and not meant to be: one wavefront per system, everything in LDS, lanes only spread
// over outputs that the reference computes independently of each other.
//
// Restated, per user row / item column (src/ALS.cpp:98-158 / :161-219):
//   Mt_byM_multiply                 (:66-79)   SUM += M[K][I] * M[K][J], K ascending, unfused
//   + lambda on the diagonal        (:120-122)
//   choldc1                         (:6-23)    sum = a[i][j]; for k = i-1 .. 0: sum -= a[i][k] * a[j][k]
//   choldcsl                        (:25-39)   double sum; sum -= A[j][k] * A[k][i]; A[j][i] = (float) sum / p[j]
//   inverseMatrix_CholeskyMethod    (:41-64)   A^-1 = L^-T L^-1, accumulated in place, then mirrored
//   subVector                       (:129-134) += val * H[col][c]
//   Wr[c]                           (:137-142) += subVector[s] * subMatrix[c][s]
// fp32 multiplies and adds stay unfused (the reference fixtures are built with -ffp-contract=off); sqrtf
// and the fp32 division are correctly rounded on both sides; the double accumulator of choldcsl is
// reproduced in fp64.
#include "als_solver.hpp"

namespace mfx {
namespace {

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
__device__ __forceinline__ double sub_rn(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

struct ExactArgs {
    const uint32_t* ptr;
    const uint32_t* idx;
    const float* val;
    const float* X;
    float* Y;
    uint32_t nseg, k;
    float lambda;
    uint32_t* spd_fail;
    const float* A_in;   // != nullptr: inverse-only mode (one system: A_in already carries + lambda)
    float* inv_out;
};

// In place on A (k x k, LDS): choldc1 + choldcsl + the product and the mirror of
// inverseMatrix_CholeskyMethod.  N is a second k x k LDS buffer, p holds the pivots.
// Called by every thread of the workgroup (its barriers are workgroup barriers); threads beyond the first 64 only meet at them.
// A is k x k with a row stride of ld = k + 1 floats: column walks (A[j][q] over lanes j) would otherwise hit one LDS bank 64 ways.
// N = L^-1 is lower triangular and stored packed, row q at q (q + 1) / 2: 25 KB of LDS per system at k = 64 instead of 34, six systems per CU
// instead of four (the inverses' dependent chains are what the user half is made of).
__device__ __forceinline__ int tri(int q) { return (q * (q + 1)) >> 1; }
template <int NC>  // column groups of 64 per lane: 1 for k <= 64 (the second group's chains would be computed and thrown away: choldcsl, 58 % of an
                   // inverse, runs its selects for every group it is given -- k = 64: 226 -> see profiles/r04_als_exact_time.txt)
__device__ void exact_inverse(float* __restrict__ A, float* __restrict__ N, float* __restrict__ p, int k, int ld, uint32_t* spd_fail) {
    const int lane = (int) threadIdx.x;
    const bool act = lane < 64;
    // choldc1 and choldcsl are ONE wave's work (64 lanes = the independent outputs of a step), so they need no workgroup barrier at
    // all: a wave's LDS operations execute in order, what its lanes wrote in one step its lanes read in the next.  The first version
    // met the other three waves twice per row and once per column -- ~190 barriers per system, a fifth of an inverse.  The pivot goes
    // from lane 0 to the others by v_readlane (the same float, so the same sqrtf).
    if (act) {
        // ---- choldc1: row i at a time; the j's of one i are independent
        for (int i = 0; i < k; ++i) {
            float sums[NC] = {};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = i + lane + 64 * c;
                if (j < k) {
                    float sum = A[i * ld + j];
#pragma unroll 8
                    for (int q = i - 1; q >= 0; --q) sum = sub_rn(sum, mul_rn(A[i * ld + q], A[j * ld + q]));  // (unrolled: eight pairs of LDS reads in flight)
                    sums[c] = sum;
                }
            }
            const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sums[0]), 0));
            const float pi = sqrtf(piv);
            if (lane == 0) {
                if (piv <= 0.f) atomicAdd(spd_fail, 1u);  // the reference prints " a is not positive definite!" and carries on
                p[i] = pi;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = i + lane + 64 * c;
                if (j > i && j < k) A[j * ld + i] = sums[c] / pi;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (compiler ordering of the LDS stores above against the loads below)
        }
        // ---- choldcsl, second loop: column i of L^-1 (lane = column; the reference works in place, column by
        // column, and column i only ever reads columns >= i of L, which it has not touched yet: reading L from A
        // and writing L^-1 to N is the same computation)
        for (int c = 0; c < NC; ++c) {
            const int i = lane + 64 * c;
            if (i < k) N[tri(i) + i] = 1.0f / p[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int j = 1; j < k; ++j) {
            double sum[NC] = {};
            // q < i contributes nothing in the reference (its loop starts at q = i): a subtracted +0.0 leaves every double as it is, -0.0
            // included, so the loop can be uniform over the lanes and unrolled (its LDS reads in flight eight at a time) -- entries of N
            // above the diagonal are never written: whatever they hold is discarded by the select, not multiplied
#pragma unroll 8
            for (int q = 0; q < j; ++q) {
                const float ljq = A[j * ld + q];  // broadcast
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = lane + 64 * c;
                    const float nqi = N[tri(q) + min(i, q)];  // (i > q is not an entry of the packed triangle: an unconditional read of the diagonal instead; the term is discarded below)
                    const double term = (i < j && q >= i) ? (double) mul_rn(ljq, nqi) : 0.0;
                    sum[c] = sub_rn(sum[c], term);
                }
            }
            const float pj = p[j];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = lane + 64 * c;
                if (i < j) N[tri(j) + i] = (float) sum[c] / pj;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    __syncthreads();
    // ---- A^-1 = L^-T L^-1: entry (i, j >= i) = sum over q >= j of N[q][i] * N[q][j]; the diagonal starts
    // from N[i][i]^2, the others from 0.0f + the first product.  Every entry is independent of the others (N is only read, A only
    // written): the rows i are dealt round-robin to ALL waves of the workgroup
    const int nwv = (int) (blockDim.x >> 6), wv = lane >> 6, l64 = lane & 63;
    for (int i = wv; i < k; i += nwv) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = i + l64 + 64 * c;
            if (j < k) {
                float acc;
                if (j == i) {
                    acc = mul_rn(N[tri(i) + i], N[tri(i) + i]);
#pragma unroll 8
                    for (int q = i + 1; q < k; ++q) acc = add_rn(acc, mul_rn(N[tri(q) + i], N[tri(q) + i]));
                } else {
                    acc = 0.0f;
#pragma unroll 8
                    for (int q = j; q < k; ++q) acc = add_rn(acc, mul_rn(N[tri(q) + i], N[tri(q) + j]));
                }
                A[i * ld + j] = acc;
                A[j * ld + i] = acc;
            }
        }
    }
    __syncthreads();
}

// (r4) Mt_byM_multiply and subVector with the ENTRIES in the outer loop.  Every output (I, J) of the Gramian is its own chain
//     SUM = 0; for K ascending: SUM += M[K][I] * M[K][J]          (unfused, src/ALS.cpp:66-79)
// and the chains are independent of each other, so the loop nest can be turned inside out without touching a single rounding: a
// gathered factor row is read ONCE (lane J takes M[K][J], the slice's M[K][I] come from the lanes that hold them: v_readlane) and advances every
// chain by one term.  The first version ran the reference's own nest -- per (I, J) a pass over all entries, three loads per term:
// 5.9 s per item half-sweep at the Netflix shape, 7.07 s per iteration.  The I range is cut into kExactWaves slices, one per wave of
// the workgroup (the 237 k-entry column is one system: its chains are its duration); (I, J) and (J, I) are the same sequence of
// products (a * b == b * a bit for bit), so the full square is written and no mirror is needed.  The right-hand side rides in wave 0.
constexpr int kExactWaves = 4;
constexpr int kExactBlock = 8;  // entries per block: their indices, ratings and factor rows are in flight while the block before is added up
template <int S, int NC>  // S: rows I per wave (k <= S * kExactWaves); NC: column groups of 64 per lane
__device__ __forceinline__ void exact_gramian(const ExactArgs& a, uint32_t lo, uint32_t hi, float* __restrict__ A, int ld, float* __restrict__ sv) {
    constexpr int B = kExactBlock;
    const int k = (int) a.k, lane = (int) (threadIdx.x & 63), wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), I0 = wave * S;
    const float* __restrict__ X = a.X;
    const uint32_t* __restrict__ idx = a.idx;
    const float* __restrict__ val = a.val;
    float acc[S][NC];
    float rhs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        rhs[c] = 0.0f;
#pragma unroll
        for (int t = 0; t < S; ++t) acc[t][c] = 0.0f;
    }
    if (I0 < k) {
        auto rl = [](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
        // block n: entries lo + B n ... ; lane l < B holds the index / rating of entry l of the block (clamped to the last entry)
        auto load_meta = [&](uint32_t q0, uint32_t& ids, float& vals) {
            const uint32_t q = min(q0 + (uint32_t) (lane & (B - 1)), hi - 1);
            ids = idx[q];
            vals = val[q];
        };
        auto load_rows = [&](uint32_t ids, float (&x)[B][NC]) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint32_t r = (uint32_t) __builtin_amdgcn_readlane((int) ids, b);
                const float* __restrict__ row = X + (size_t) r * k;
#pragma unroll
                for (int c = 0; c < NC; ++c) { const int J = lane + 64 * c; x[b][c] = J < k ? row[J] : 0.0f; }
            }
        };
        auto add_block = [&](uint32_t q0, const float (&x)[B][NC], float vals) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (q0 + (uint32_t) b >= hi) break;  // (wave-uniform)
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    // M[K][I]: lane I % 64 of column group I / 64 holds it (a readlane with a wave-uniform lane index; rows past k read 0)
                    const int I = I0 + t;
                    float xi;
                    if constexpr (NC == 1) {
                        xi = rl(x[b][0], I & 63);
                    } else {
                        const float x0 = rl(x[b][0], I & 63), x1 = rl(x[b][NC - 1], I & 63);
                        xi = I < 64 ? x0 : x1;
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[t][c] = add_rn(acc[t][c], mul_rn(xi, x[b][c]));
                }
                if (wave == 0) {
                    const float v = rl(vals, b);
#pragma unroll
                    for (int c = 0; c < NC; ++c) rhs[c] = add_rn(rhs[c], mul_rn(v, x[b][c]));
                }
            }
        };
        // two sets that swap roles: the rows of block n + 1 and the indices of block n + 2 are in flight under the adds of block n
        float xa[B][NC], xb[B][NC];
        uint32_t ids_a, ids_b;
        float vals_a, vals_b;
        load_meta(lo, ids_a, vals_a);
        load_meta(lo + B, ids_b, vals_b);
        load_rows(ids_a, xa);
        for (uint32_t q0 = lo; q0 < hi; q0 += 2 * B) {
            load_rows(ids_b, xb);                            // block n + 1
            const float va = vals_a;
            load_meta(q0 + 2 * B, ids_a, vals_a);            // block n + 2
            add_block(q0, xa, va);
            if (q0 + B >= hi) break;
            load_rows(ids_a, xa);                            // block n + 2
            const float vb = vals_b;
            load_meta(q0 + 3 * B, ids_b, vals_b);            // block n + 3
            add_block(q0 + B, xb, vb);
        }
#pragma unroll
        for (int t = 0; t < S; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int I = I0 + t, J = lane + 64 * c;
                if (I < k && J < k) A[I * ld + J] = acc[t][c];
            }
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) { const int J = lane + 64 * c; if (J < k) sv[J] = rhs[c]; }
        }
    }
}

template <bool BIG>  // BIG: 64 < k <= 128 (32 rows per wave, two column groups per lane: its registers must not cap the occupancy of the k <= 64 kernel)
__global__ __launch_bounds__(64 * kExactWaves) void k_als_exact(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = (int) a.k, lane = (int) threadIdx.x, ld = k + 1;
    float* A = lds;
    float* N = lds + (size_t) k * ld;
    float* p = N + (size_t) tri(k);
    float* sv = p + k;
    if (a.A_in) {
        for (int e = lane; e < k * k; e += (int) blockDim.x) A[(e / k) * ld + e % k] = a.A_in[e];
        __syncthreads();
        exact_inverse<BIG ? 2 : 1>(A, N, p, k, ld, a.spd_fail);
        for (int e = lane; e < k * k; e += (int) blockDim.x) a.inv_out[e] = A[(e / k) * ld + e % k];
        return;
    }
    const uint32_t seg = blockIdx.x;
    const uint32_t lo = a.ptr[seg], hi = a.ptr[seg + 1];
    float* y = a.Y + (size_t) seg * k;
    if (hi == lo) {  // src/ALS.cpp:151-157
        for (int c = lane; c < k; c += (int) blockDim.x) y[c] = 0.0f;
        return;
    }
    // Mt_byM_multiply + subVector (see exact_gramian)
    if constexpr (!BIG) exact_gramian<16, 1>(a, lo, hi, A, ld, sv);
    else exact_gramian<32, 2>(a, lo, hi, A, ld, sv);
    __syncthreads();
    for (int c = lane; c < k; c += (int) blockDim.x) A[c * ld + c] = add_rn(A[c * ld + c], a.lambda);
    __syncthreads();
    // the right-hand side was accumulated next to the Gramian; the inverse overwrites A, not sv
    exact_inverse<BIG ? 2 : 1>(A, N, p, k, ld, a.spd_fail);
    for (int c = lane; c < k; c += (int) blockDim.x) {
        float w = 0.0f;
#pragma unroll 8
        for (int s2 = 0; s2 < k; ++s2) w = add_rn(w, mul_rn(sv[s2], A[c * ld + s2]));
        y[c] = w;
    }
}

int launch_exact(const ExactArgs& a, uint32_t grid, hipStream_t st) {
    const size_t lds_bytes = ((size_t) a.k * (a.k + 1) + (size_t) a.k * (a.k + 1) / 2 + 2 * a.k) * sizeof(float);  // (A with rows padded by one float, N packed: see exact_inverse)
    MFX_REQUIRE(lds_bytes <= 160 * 1024, "ALS as written: rank k = %u does not fit LDS", a.k);
    const bool big = a.k > 16 * kExactWaves;
    if (lds_bytes > 48 * 1024)
        MFX_HIP(hipFuncSetAttribute(big ? reinterpret_cast<const void*>(k_als_exact<true>) : reinterpret_cast<const void*>(k_als_exact<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
    if (big) hipLaunchKernelGGL(k_als_exact<true>, dim3(grid), dim3(64 * kExactWaves), lds_bytes, st, a);
    else hipLaunchKernelGGL(k_als_exact<false>, dim3(grid), dim3(64 * kExactWaves), lds_bytes, st, a);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

}  // namespace

int als_half_exact_launch(const AlsHalf& h, const float* X, float* Y, uint32_t k, float lambda, uint32_t* spd_fail, hipStream_t st) {
    if (h.nseg == 0) return MFX_OK;
    ExactArgs a{};
    a.ptr = h.ptr.get(); a.idx = h.idx.get(); a.val = h.val.get(); a.X = X; a.Y = Y; a.nseg = h.nseg; a.k = k; a.lambda = lambda;
    a.spd_fail = spd_fail;
    return launch_exact(a, h.nseg, st);
}

int als_inverse_op(int64_t k, const float* A, float* Ainv, int device) {
    MFX_REQUIRE(k >= 1 && k <= 128, "ALS: rank k = %lld not supported (1 <= k <= 128)", (long long) k);
    MFX_TRY(use_device(device));
    DevBuf<float> dA, dO;
    DevBuf<uint32_t> bad;
    MFX_TRY(dA.alloc((size_t) k * k)); MFX_TRY(dO.alloc((size_t) k * k));
    MFX_TRY(bad.alloc_zero(1, nullptr));
    MFX_HIP(hipMemcpy(dA.get(), A, sizeof(float) * k * k, hipMemcpyHostToDevice));
    ExactArgs a{};
    a.k = (uint32_t) k; a.A_in = dA.get(); a.inv_out = dO.get(); a.spd_fail = bad.get();
    MFX_TRY(launch_exact(a, 1, nullptr));
    MFX_HIP(hipStreamSynchronize(nullptr));
    MFX_HIP(hipMemcpy(Ainv, dO.get(), sizeof(float) * k * k, hipMemcpyDeviceToHost));
    return MFX_OK;
}

}  // namespace mfx

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
__device__ void exact_inverse(float* __restrict__ A, float* __restrict__ N, float* __restrict__ p, int k, uint32_t* spd_fail) {
    const int lane = (int) threadIdx.x;
    // ---- choldc1: row i at a time; the j's of one i are independent
    for (int i = 0; i < k; ++i) {
        float sums[2] = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = i + lane + 64 * c;
            if (j < k) {
                float sum = A[i * k + j];
                for (int q = i - 1; q >= 0; --q) sum = sub_rn(sum, mul_rn(A[i * k + q], A[j * k + q]));
                sums[c] = sum;
            }
        }
        if (lane == 0) {
            if (sums[0] <= 0.f) atomicAdd(spd_fail, 1u);  // the reference prints " a is not positive definite!" and carries on
            p[i] = sqrtf(sums[0]);
        }
        __syncthreads();
        const float pi = p[i];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = i + lane + 64 * c;
            if (j > i && j < k) A[j * k + i] = sums[c] / pi;
        }
        __syncthreads();
    }
    // ---- choldcsl, second loop: column i of L^-1 (lane = column; the reference works in place, column by
    // column, and column i only ever reads columns >= i of L, which it has not touched yet: reading L from A
    // and writing L^-1 to N is the same computation)
    for (int c = 0; c < 2; ++c) {
        const int i = lane + 64 * c;
        if (i < k) N[i * k + i] = 1.0f / p[i];
    }
    __syncthreads();
    for (int j = 1; j < k; ++j) {
        double sum[2] = {0.0, 0.0};
        for (int q = 0; q < j; ++q) {
            const float ljq = A[j * k + q];  // broadcast
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = lane + 64 * c;
                if (i < j && q >= i) sum[c] = sub_rn(sum[c], (double) mul_rn(ljq, N[q * k + i]));
            }
        }
        const float pj = p[j];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = lane + 64 * c;
            if (i < j) N[j * k + i] = (float) sum[c] / pj;
        }
        __syncthreads();
    }
    // ---- A^-1 = L^-T L^-1: entry (i, j >= i) = sum over q >= j of N[q][i] * N[q][j]; the diagonal starts
    // from N[i][i]^2, the others from 0.0f + the first product
    for (int i = 0; i < k; ++i) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = i + lane + 64 * c;
            if (j < k) {
                float acc;
                if (j == i) {
                    acc = mul_rn(N[i * k + i], N[i * k + i]);
                    for (int q = i + 1; q < k; ++q) acc = add_rn(acc, mul_rn(N[q * k + i], N[q * k + i]));
                } else {
                    acc = 0.0f;
                    for (int q = j; q < k; ++q) acc = add_rn(acc, mul_rn(N[q * k + i], N[q * k + j]));
                }
                A[i * k + j] = acc;
                A[j * k + i] = acc;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_als_exact(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = (int) a.k, lane = (int) threadIdx.x;
    float* A = lds;
    float* N = lds + (size_t) k * k;
    float* p = N + (size_t) k * k;
    float* sv = p + k;
    if (a.A_in) {
        for (int e = lane; e < k * k; e += 64) A[e] = a.A_in[e];
        __syncthreads();
        exact_inverse(A, N, p, k, a.spd_fail);
        for (int e = lane; e < k * k; e += 64) a.inv_out[e] = A[e];
        return;
    }
    const uint32_t seg = blockIdx.x;
    const uint32_t lo = a.ptr[seg], hi = a.ptr[seg + 1];
    float* y = a.Y + (size_t) seg * k;
    if (hi == lo) {  // src/ALS.cpp:151-157
        for (int c = lane; c < k; c += 64) y[c] = 0.0f;
        return;
    }
    // Mt_byM_multiply: upper triangle, mirrored
    for (int I = 0; I < k; ++I) {
        for (int J = I + lane; J < k; J += 64) {
            float sum = 0.0f;
            for (uint32_t q = lo; q < hi; ++q) {
                const float* row = a.X + (size_t) a.idx[q] * k;
                sum = add_rn(sum, mul_rn(row[I], row[J]));
            }
            A[J * k + I] = sum;
            A[I * k + J] = sum;
        }
    }
    __syncthreads();
    for (int c = lane; c < k; c += 64) A[c * k + c] = add_rn(A[c * k + c], a.lambda);
    __syncthreads();
    exact_inverse(A, N, p, k, a.spd_fail);
    for (int c = lane; c < k; c += 64) {
        float s = 0.0f;
        for (uint32_t q = lo; q < hi; ++q) s = add_rn(s, mul_rn(a.val[q], a.X[(size_t) a.idx[q] * k + c]));
        sv[c] = s;
    }
    __syncthreads();
    for (int c = lane; c < k; c += 64) {
        float w = 0.0f;
        for (int s = 0; s < k; ++s) w = add_rn(w, mul_rn(sv[s], A[c * k + s]));
        y[c] = w;
    }
}

int launch_exact(const ExactArgs& a, uint32_t grid, hipStream_t st) {
    const size_t lds_bytes = ((size_t) 2 * a.k * a.k + 2 * a.k) * sizeof(float);
    MFX_REQUIRE(lds_bytes <= 160 * 1024, "ALS as written: rank k = %u does not fit LDS", a.k);
    if (lds_bytes > 48 * 1024)
        MFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_als_exact), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
    hipLaunchKernelGGL(k_als_exact, dim3(grid), dim3(64), lds_bytes, st, a);
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

// (r4) Positive control for "do fp32 MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?" (DESIGN section 5).
// One workgroup per CU, W waves per SIMD; a wave is either an MFMA wave (independent v_mfma_f32_16x16x4_f32 accumulators,
// the ALS Gramian's instruction) or a VALU wave (independent v_fma_f32 chains, optionally v_pk_fma_f32).  Timed: MFMA waves
// alone, VALU waves alone, both kinds together on every SIMD.  Overlap <=> t(both) ~ max, no overlap <=> t(both) ~ sum.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coexec.hip -o tools/build/ubench_coexec ; run: ubench_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int KIND>  // 0: MFMA, 1: VALU scalar fma, 2: VALU packed fma
__device__ __forceinline__ void body(int iters, float* out) {
    const int lane = threadIdx.x & 63;
    if constexpr (KIND == 0) {
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float a = 1.0f + lane * 1e-3f, b = 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (KIND == 1) {
        float x[8];
        for (int t = 0; t < 8; ++t) x[t] = lane * 0.01f + t;
        const float m = 0.999f, c = 0.001f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)  // 8 x 8 = 64 VALU instructions per iteration: 256 SIMD cycles, the MFMA body's 8 x 32
#pragma unroll
                for (int t = 0; t < 8; ++t) x[t] = __builtin_fmaf(x[t], m, c);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) s += x[t];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x2 x[8];
        for (int t = 0; t < 8; ++t) x[t] = f32x2{lane * 0.01f + t, 1.f};
        const f32x2 m = {0.999f, 0.998f}, c = {0.001f, 0.002f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) x[t] = __builtin_elementwise_fma(x[t], m, c);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) s += x[t].x + x[t].y;
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

// waves 0 .. 4 nm - 1 are MFMA waves, the following 4 nv VALU waves (wave w sits on SIMD w % 4)
template <int VK>
__global__ __launch_bounds__(1024) void k_mix(int nm, int nv, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4 * nm) body<0>(iters, out);
    else if (wave < 4 * (nm + nv)) body<VK>(iters, out);
}

static float run(int vk, int nm, int nv, int iters, float* out, int cus) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int threads = 256 * (nm + nv);
    auto launch = [&] {
        if (vk == 1) hipLaunchKernelGGL(k_mix<1>, dim3(cus), dim3(threads), 0, 0, nm, nv, iters, out);
        else hipLaunchKernelGGL(k_mix<2>, dim3(cus), dim3(threads), 0, 0, nm, nv, iters, out);
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 1024);
    const int iters = 20000;
    for (int vk = 1; vk <= 2; ++vk) {
        printf("== VALU kind %s; %d iterations of 8 MFMAs (256 SIMD cycles) / 64 VALU instructions (256 SIMD cycles) per wave, one workgroup on each of %d CUs\n",
               vk == 1 ? "v_fma_f32" : "v_pk_fma_f32", iters, cus);
        for (int w = 1; w <= 2; ++w) {
            const float tm = run(vk, w, 0, iters, out, cus), tv = run(vk, 0, w, iters, out, cus), tb = run(vk, w, w, iters, out, cus);
            printf("%d MFMA wave(s) per SIMD alone %.3f ms | %d VALU wave(s) per SIMD alone %.3f ms | both kinds together %.3f ms  (max %.3f, sum %.3f)\n",
                   w, tm, w, tv, tb, tm > tv ? tm : tv, tm + tv);
        }
    }
    return 0;
}

// (r4) How fast can a strictly sequential fp32 sum go on gfx950?  Dependent-issue latency of the candidate chain steps, one wave
// per workgroup, one workgroup: (a) v_add_f32_dpp wave_shr:1 (the lane-to-lane chain of ccd_reforder.hip), (b) plain v_add_f32
// with a VGPR term (terms broadcast from LDS), (c) the same with two independent chains interleaved, (d) v_add_f32 with the term
// read from LDS four at a time (ds_read_b128 broadcast) inside the loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/build/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float shr1(float s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x138, 0xF, 0xF, true));
}
template <int KIND>
__global__ __launch_bounds__(64) void k_chain(int iters, const float* __restrict__ in, float* out, long long* clk) {
    __shared__ __attribute__((aligned(16))) float terms[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) terms[i] = in[i];
    __syncthreads();
    float s = in[threadIdx.x], t = s * 0.5f;
    const float q = in[64 + threadIdx.x];
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) s = add_rn(shr1(s), q);
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 64; ++j) s = add_rn(s, q);
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 64; ++j) { s = add_rn(s, q); t = add_rn(t, q); }
        } else {
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&terms[(i * 64 + j) & 1020]);
                s = add_rn(s, x[0]); s = add_rn(s, x[1]); s = add_rn(s, x[2]); s = add_rn(s, x[3]);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = s + t;
    if (threadIdx.x == 0) *clk = t1 - t0;
}
int main() {
    float *in, *out; long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&clk, 8);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1e-3f * (i % 7);
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    const int iters = 2000;
    const char* names[4] = {"v_add_f32_dpp wave_shr:1 (one chain)", "v_add_f32 VGPR term (one chain)", "v_add_f32, two chains interleaved (per pair)", "v_add_f32, terms by ds_read_b128 broadcast"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-48s %.2f counter ticks per dependent step (s_memtime-class counter: 100 MHz x ... see ratio between rows)\n", names[kind], (double) c / (iters * 64.0));
    }
    return 0;
}

// ubench_mfma32.hip -- what keeps v_mfma_f32_16x16x4_f32 from its issue rate in a Gramian-shaped loop (gfx950).
// Question behind it: k_als_gram16 (als_solver.hip) runs its 40 MFMAs per 16 gathered rows at ~62 % of the matrix
// rate whether the gather is served from HBM, L2 or L1 and whether the loop holds 100 or 45 other vector
// instructions.  Variants (all: one 64-thread workgroup per wave slot, W waves per SIMD, 10 accumulators updated
// from 4 operand registers per row group exactly like the kernel):
//   0  MFMAs only, operands fixed
//   1  + the rhs update (2 v_pk_fma_f32 + 1 dpp mov per row group)
//   2  + 12 L1-resident loads per step, consumed one step later (the kernel's pipeline, no gather arithmetic)
//   4  + a private (scratch) segment     5  + the 11 KB accumulator park of a long segment's chunk
//   3  the kernel's load phase: row offsets formed (24-bit multiply, select) from indices loaded two steps earlier
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma32 tools/ubench_mfma32.hip ; run: ubench_mfma32 [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int VAR, int W>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) void k(const float* __restrict__ tab, int steps, float* out, int idxmul) {
    const unsigned lane = threadIdx.x;
    float priv[8];
    if (VAR >= 4) {  // force a private (scratch) segment, like the real kernel's spills
#pragma unroll
        for (int i = 0; i < 8; ++i) priv[i] = tab[i + lane];
        asm volatile("" ::: "memory");
    }
    f32x4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 b0 = {0.f, 0.f}, b1 = {0.f, 0.f};
    f32x4 av[2][4];
    float rv[2][4];
    const f32x4* t4 = reinterpret_cast<const f32x4*>(tab);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[s][u] = t4[(lane + 64 * u + 256 * s) & 1023]; rv[s][u] = tab[(lane + u) & 1023]; }
    unsigned ix[2][4];
    const unsigned* itab = reinterpret_cast<const unsigned*>(tab) + 8192;  // small values: row numbers
    const char* Xb = reinterpret_cast<const char*>(tab);
    const unsigned g = lane >> 4, lane_off = 16 * (lane & 15);
    const unsigned len = (unsigned) steps * 16 - 5, rowbytes = (unsigned) idxmul;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) ix[s][u] = itab[(16 * s + 4 * u + g) & 1023];
    auto loads = [&](auto S, int step) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (VAR >= 3) {  // the kernel's load phase: row offset from an index loaded two steps earlier
                const bool ok = (unsigned) step * 16 + 4 * u + g < len;
                unsigned i = ix[s][u];
                asm volatile("" : "+v"(i));
                const unsigned off = ok ? __umul24(i, rowbytes) + lane_off : 0u;
                av[s][u] = *reinterpret_cast<const f32x4*>(Xb + off);
                rv[s][u] = tab[((unsigned) step * 16 + 4 * u + g) & 4095];
            } else {
                av[s][u] = t4[(lane + 64 * u + 16 * step) & 1023];   // 16 KB table: L1 resident
                rv[s][u] = tab[(lane + 4 * u + step) & 4095];
            }
        }
    };
    auto load_idx = [&](auto S, int step) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) ix[s][u] = itab[((unsigned) step * 16 + 4 * u + g) & 1023];
    };
    auto mfmas = [&](auto S) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (VAR >= 1) {
                const float hi = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rv[s][u]), 0xE4, 0xF, 0xF, false));
                const f32x2 rr = {rv[s][u], hi};
                b0 = __builtin_elementwise_fma(rr, av[s][u].lo, b0);
                b1 = __builtin_elementwise_fma(rr, av[s][u].hi, b1);
            }
            int ti = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int f = e; f < 4; ++f, ++ti)
                    acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][u][e], av[s][u][f], acc[ti], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    for (int st = 0; st < steps; st += 2) {
        if (VAR >= 2) loads(I1{}, st + 1);
        if (VAR >= 3) load_idx(I0{}, st + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I0{});
        __builtin_amdgcn_sched_barrier(0);
        if (VAR >= 2) loads(I0{}, st + 2);
        if (VAR >= 3) load_idx(I1{}, st + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I1{});
        __builtin_amdgcn_sched_barrier(0);
    }
    if (VAR >= 5) {  // park the accumulators like a chunk of a long segment does (11 KB per wave)
        float* w = out + 65536 * 64 + (size_t) (blockIdx.x & 8191) * 44 * 64;
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[(t * 4 + q) * 64 + lane] = acc[t][q];
        w[40 * 64 + lane] = b0.x; w[41 * 64 + lane] = b0.y; w[42 * 64 + lane] = b1.x; w[43 * 64 + lane] = b1.y;
    }
    float t = b0.x + b0.y + b1.x + b1.y;
    if (VAR >= 4) t += priv[(steps + lane) & 7];
#pragma unroll
    for (int i = 0; i < 10; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(blockIdx.x & 4095) * 64 + lane] = t;
}

template <int VAR, int W>
void run(const float* tab, float* out, int steps) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * 4 * W;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<VAR, W>), dim3(grid), dim3(64), 0, 0, tab, 64, out, 256);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<VAR, W>), dim3(grid), dim3(64), 0, 0, tab, steps, out, 256);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double) grid * steps * 40;
    const double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at a nominal 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / ((double) W * steps * 40);
    printf("var %d  waves/SIMD %d  %8.3f ms  %6.1f TF  %5.1f cycles(2.4 GHz) per MFMA per SIMD\n", VAR, W, ms, tf, cyc);
}

// the item half of the Netflix-shaped ALS iteration as launched by the solver: 106 900 one-wave workgroups of 64 steps,
// `lds` bytes of dynamic LDS each (the Cholesky image the real kernel reserves)
template <int W, int VAR = 3>
void run_short(const float* tab, float* out, int grid, int steps, int lds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<VAR, W>), dim3(1024), dim3(64), lds, 0, tab, 64, out, 256);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<VAR, W>), dim3(grid), dim3(64), lds, 0, tab, steps, out, 256);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double) grid * steps * 40;
    printf("var %d short workgroups: grid %d x %d steps, %d B LDS, waves/SIMD <= %d: %8.3f ms  %6.1f TF  %5.1f cycles(2.4 GHz) per MFMA per SIMD\n",
           VAR, grid, steps, lds, W, ms, mfma * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfma / 1024));
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4096;
    float* tab; float* out;
    CK(hipMalloc(&tab, 4096 * sizeof(float) * 4)); CK(hipMemset(tab, 0, 4096 * sizeof(float) * 4));
    {   // words 8192 .. 9215: row numbers 0 .. 7 (the gather stays inside the first 2 KB: L1 resident)
        unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (unsigned) (i * 7) & 7u;
        CK(hipMemcpy(reinterpret_cast<unsigned*>(tab) + 8192, h, sizeof(h), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, sizeof(float) * 64 * (65536 + 8192 * 44)));
    run<0, 1>(tab, out, steps); run<0, 2>(tab, out, steps); run<0, 4>(tab, out, steps);
    run<1, 1>(tab, out, steps); run<1, 4>(tab, out, steps);
    run<2, 1>(tab, out, steps); run<2, 2>(tab, out, steps); run<2, 4>(tab, out, steps);
    run<3, 1>(tab, out, steps); run<3, 2>(tab, out, steps); run<3, 4>(tab, out, steps);
    run_short<4>(tab, out, 106900, 64, 0);
    run_short<4>(tab, out, 106900, 64, 8960);
    run_short<4>(tab, out, 480189, 14, 8960);
    run_short<4>(tab, out, 13362, 512, 8960);
    run_short<4, 4>(tab, out, 106900, 64, 8960);
    run_short<4, 5>(tab, out, 106900, 64, 8960);
    run_short<4, 4>(tab, out, 480189, 14, 8960);
    return 0;
}

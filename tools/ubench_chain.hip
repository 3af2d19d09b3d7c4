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
template <int J> __device__ __forceinline__ float bcast16(float x) {  // lane J of every row of 16 lanes, to the whole row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + J, 0xF, 0xF, true));
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
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&terms[(i * 64 + j) & 1020]);
                s = add_rn(s, x[0]); s = add_rn(s, x[1]); s = add_rn(s, x[2]); s = add_rn(s, x[3]);
            }
        } else if constexpr (KIND == 4) {
            // 64 terms in ONE ds_read_b128: lane l reads terms 4 (l & 15) .. + 3, so every row of 16 lanes holds the same 64 terms and
            // term 4 j + e reaches all lanes as the DPP operand `x[e] row_newbcast:j` of the add itself (the running sum is the PLAIN operand)
            const f32x4 x = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 4 * (threadIdx.x & 15)]);
#define BC4(J) s = add_rn(bcast16<J>(x[0]), s); s = add_rn(bcast16<J>(x[1]), s); s = add_rn(bcast16<J>(x[2]), s); s = add_rn(bcast16<J>(x[3]), s);
            BC4(0) BC4(1) BC4(2) BC4(3) BC4(4) BC4(5) BC4(6) BC4(7) BC4(8) BC4(9) BC4(10) BC4(11) BC4(12) BC4(13) BC4(14) BC4(15)
        } else if constexpr (KIND == 7) {
            // the same as 4 without the compiler's s_nop 1 between the adds (its DPP hazard rule looks at every VGPR a DPP instruction
            // reads; the hardware's concerns the DPP operand, and the running sum is the plain one) -- the printed sums say whether that holds
            const f32x4 x = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 4 * (threadIdx.x & 15)]);
            float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
#define A4(J) "v_add_f32_dpp %0, %1, %0 row_newbcast:" #J " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %0, %2, %0 row_newbcast:" #J " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %0, %3, %0 row_newbcast:" #J " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %0, %4, %0 row_newbcast:" #J " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            asm volatile("s_nop 1\n" A4(0) A4(1) A4(2) A4(3) A4(4) A4(5) A4(6) A4(7) A4(8) A4(9) A4(10) A4(11) A4(12) A4(13) A4(14) A4(15)
                         : "+v"(s) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        } else if constexpr (KIND == 8) {
            // sixteen chains per wave: every QUAD of lanes carries one sum, lane k of the quad holds entries 16 m + 4 k ... + 3 of the quad's
            // 64-entry chunk in x_m (m = 0 ... 3); the add takes lane k of the quad through quad_perm:[k,k,k,k]
            const f32x4 xa = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 4 * (threadIdx.x & 3)]);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 16 + 4 * (threadIdx.x & 3)]);
            const f32x4 xc = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 32 + 4 * (threadIdx.x & 3)]);
            const f32x4 xd = *reinterpret_cast<const f32x4*>(&terms[((i * 64) & 960) + 48 + 4 * (threadIdx.x & 3)]);
#define Q1(X, K) "v_add_f32_dpp %0, " X ", %0 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define Q4(A, B, C, D, K) Q1(A, K) Q1(B, K) Q1(C, K) Q1(D, K)
#define Q16(A, B, C, D) Q4(A, B, C, D, 0) Q4(A, B, C, D, 1) Q4(A, B, C, D, 2) Q4(A, B, C, D, 3)
            asm volatile("s_nop 1\n" Q16("%1", "%2", "%3", "%4") Q16("%5", "%6", "%7", "%8") Q16("%9", "%10", "%11", "%12") Q16("%13", "%14", "%15", "%16")
                         : "+v"(s)
                         : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]),
                           "v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(xd[0]), "v"(xd[1]), "v"(xd[2]), "v"(xd[3]));
        } else if constexpr (KIND == 5) {
            // terms from global memory through the scalar cache (wave-uniform address): v_add_f32 v, s, v
            const float* __restrict__ tp = in + ((i * 64) & 960);
#pragma unroll
            for (int j = 0; j < 64; ++j) s = add_rn(s, tp[j]);
        } else {
            // one coalesced dword per lane, v_readlane per term
            const float x = terms[((i * 64) & 960) + threadIdx.x];
#pragma unroll
            for (int j = 0; j < 64; ++j) s = add_rn(s, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j)));
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
    const char* names[9] = {"v_add_f32_dpp wave_shr:1 (one chain)", "v_add_f32 VGPR term (one chain)", "v_add_f32, two chains interleaved (per pair)", "v_add_f32, terms by ds_read_b128 broadcast", "v_add_f32_dpp row_newbcast (64 terms per ds_read_b128)", "v_add_f32 with SGPR term (s_load)", "v_readlane + v_add_f32", "v_add_f32_dpp row_newbcast, no s_nop between the adds", "v_add_f32_dpp quad_perm broadcast, no s_nop (16 chains per wave)"};
    for (int kind = 0; kind < 9; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 4) hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 5) hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 6) hipLaunchKernelGGL(k_chain<6>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 7) hipLaunchKernelGGL(k_chain<7>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            if (kind == 8) hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(64), 0, 0, iters, in, out, clk);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        float o0; hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost);
        printf("%-48s %.2f counter ticks per dependent step (s_memtime-class counter: 100 MHz x ... see ratio between rows)   lane 0's sum %.9g\n", names[kind], (double) c / (iters * 64.0), o0);
    }
    return 0;
}

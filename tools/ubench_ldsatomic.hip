// ubench_ldsatomic.hip -- LDS atomic throughput on one CU-filling workgroup (1024 threads), gfx950.
// Question behind it: the scatter pass (ccd_scatter.hip) issues two ds_add_u64 per non-zero; what does the LDS
// sustain for 64-bit adds, 32-bit adds and native f32 adds, with random (bank-conflicting) and with
// conflict-free addresses?  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o ubench_ldsatomic tools/ubench_ldsatomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kSlots = 6144;

template <int KIND, bool RANDOM>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    unsigned long long* a64 = reinterpret_cast<unsigned long long*>(raw);
    unsigned* a32 = reinterpret_cast<unsigned*>(raw);
    float* af = reinterpret_cast<float*>(raw);
    for (int i = threadIdx.x; i < 2 * kSlots; i += 1024) a64[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        unsigned slot;
        if (RANDOM) { x = x * 1664525u + 1013904223u; slot = (x >> 8) % kSlots; }
        else slot = (threadIdx.x + it * 64) % kSlots;
        if (KIND == 0) { atomicAdd(&a64[2 * slot], (unsigned long long) x); atomicAdd(&a64[2 * slot + 1], (unsigned long long) it); }
        if (KIND == 1) { atomicAdd(&a32[4 * slot], x); atomicAdd(&a32[4 * slot + 2], (unsigned) it); }
        if (KIND == 2) { atomicAdd(&af[4 * slot], (float) it); atomicAdd(&af[4 * slot + 2], 1.0f); }
        if (KIND == 3) { atomicAdd(&a32[4 * slot], x); atomicAdd(&a32[4 * slot + 1], x >> 3); atomicAdd(&a32[4 * slot + 2], (unsigned) it); atomicAdd(&a32[4 * slot + 3], 1u); }
        if (KIND == 4) { atomicAdd(&a64[slot], (unsigned long long) x); }   // one 64-bit add per element, dense slots
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < 2 * kSlots; i += 1024) s += a64[i];
    if (s == 0x1234567ull) out[blockIdx.x] = s;
}

template <int KIND, bool RANDOM>
void run(const char* name, int per_elem) {
    const int iters = 4096, blocks = 256;
    unsigned long long* out;
    CK(hipMalloc(&out, blocks * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, RANDOM>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlots * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, RANDOM>), dim3(blocks), dim3(1024), 2 * kSlots * 8, 0, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, RANDOM>), dim3(blocks), dim3(1024), 2 * kSlots * 8, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double elems = (double) iters * 1024;            // per workgroup (= per CU when blocks == CUs)
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-44s %7.3f ms  %6.2f clk per element per CU  (%d atomics/element -> %5.2f lane-atomics/clk/CU)\n", name, ms, clk / elems,
           per_elem, per_elem * elems / clk);
    CK(hipFree(out));
}

int main() {
    run<0, true>("2 x ds_add_u64, random slots", 2);
    run<0, false>("2 x ds_add_u64, conflict-free slots", 2);
    run<4, true>("1 x ds_add_u64, random slots", 1);
    run<1, true>("2 x ds_add_u32, random slots", 2);
    run<1, false>("2 x ds_add_u32, conflict-free slots", 2);
    run<3, true>("4 x ds_add_u32, random slots", 4);
    run<2, true>("2 x ds_add_f32 (native), random slots", 2);
    run<2, false>("2 x ds_add_f32 (native), conflict-free", 2);
    return 0;
}

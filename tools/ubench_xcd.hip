// ubench_xcd.hip -- which XCD does workgroup b land on?  (1024-thread workgroups, one per CU because of their LDS)
//   build: hipcc --offload-arch=gfx950 -O3 -o ubench_xcd tools/ubench_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned spin) {
    extern __shared__ unsigned lds[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = xcc;
    __syncthreads();
    unsigned x = lds[(threadIdx.x + 1) & 1023];
    for (unsigned i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;  // keep the CU busy for a while
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) | ((x & 1) << 31 >> 31 << 8 & 0);
}
int main() {
    const int nb = 2048;
    unsigned* d; CK(hipMalloc(&d, nb * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 147 * 1024));
    for (unsigned spin : {0u, 20000u}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 147 * 1024, 0, d, spin);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(nb); CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost));
        printf("spin %u: first 64 blocks: ", spin);
        for (int b = 0; b < 64; ++b) printf("%u", h[b] & 0xF);
        int match = 0; for (int b = 0; b < nb; ++b) match += (h[b] & 0xF) == (h[b % 8] & 0xF);
        printf("\n  blocks whose XCD equals that of block (b %% 8): %d of %d\n", match, nb);
        printf("  blocks 256..319: "); for (int b = 256; b < 320; ++b) printf("%u", h[b] & 0xF); printf("\n");
        printf("  blocks 1024..1087: "); for (int b = 1024; b < 1088; ++b) printf("%u", h[b] & 0xF); printf("\n");
    }
    return 0;
}

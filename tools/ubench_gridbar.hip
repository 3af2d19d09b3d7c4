// (r4) What does a chip-wide barrier INSIDE a kernel cost next to a dependent hipGraph node (~4.8 us, profiles/r04_kernel_stats_ml1m.csv)?
// Small matrices are launch-bound: 2 k dependent nodes per outer iteration.  A persistent kernel would replace every node
// boundary by a grid barrier -- but the 8 XCDs' L2s are not coherent with each other, so what a workgroup publishes has to leave its
// L2 and what it reads afterwards must not come from a stale line.  Two ways, both measured here with P phases in one launch; in every
// phase a workgroup (one per CU) dirties D bytes of private data (the residual it owns), publishes its slice of a shared table (the
// factor column / operand pack), passes the barrier, and reads the WHOLE table (the gather) -- checked against the expected values:
//   mode 0  "fences":  plain stores; agent-scope RELEASE fence (buffer_wbl2 sc1: the L2's dirty lines, private ones included) ->
//                      relaxed add -> relaxed spin -> agent-scope ACQUIRE fence (buffer_inv sc1) -> plain loads
//   mode 1  "write-through": the table is stored with agent-scope relaxed atomic stores (sc1 write-through, nothing else leaves the
//                      L2), s_waitcnt vmcnt(0) -> relaxed add -> relaxed spin -> ACQUIRE fence -> plain loads
//   arrival: flat (every workgroup adds to ONE counter) or by XCD (a counter per XCC_ID; the XCD's last arrival adds to the chip's)
// Every spin is bounded (kSpinLimit polls, then the kernel flags an error and every wave leaves): nothing can hang.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gridbar.hip -o tools/build/ubench_gridbar ; run: ubench_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using gu32 = __attribute__((address_space(1))) unsigned;
constexpr unsigned kSpinLimit = 1u << 22;
constexpr int kBlock = 256;

struct Bar {
    unsigned* chip;      // [0]: arrivals (flat) or XCD completions (by XCD); monotone over the phases
    unsigned* xcd;       // [8 * 32]: arrivals per XCD (one 128-byte line each)
    unsigned* xcd_size;  // [8]: workgroups per XCD (counted in phase 0)
    unsigned* err;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u; }  // HW_REG_XCC_ID[3:0]

// thread 0 of every workgroup; `target` = value the counter reaches when everyone has arrived at this barrier
__device__ __forceinline__ bool spin_until(unsigned* ctr, unsigned target, unsigned* err) {
    for (unsigned i = 0; i < kSpinLimit; ++i) {
        if ((int) (__hip_atomic_load((gu32*) ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
        if ((i & 255u) == 255u && __hip_atomic_load((gu32*) err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store((gu32*) err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

template <bool FENCES, bool BYXCD>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned phase /* 1-based */, unsigned nblocks, unsigned my_xcd, unsigned my_xcd_size, unsigned nxcd_live) {
    __shared__ int ok;
    if constexpr (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool good;
        if constexpr (BYXCD) {
            const unsigned before = __hip_atomic_fetch_add((gu32*) (b.xcd + 32 * my_xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == phase * my_xcd_size) __hip_atomic_fetch_add((gu32*) b.chip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            good = spin_until(b.chip, phase * nxcd_live, b.err);
        } else {
            __hip_atomic_fetch_add((gu32*) b.chip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            good = spin_until(b.chip, phase * nblocks, b.err);
        }
        ok = good;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

// table: [2][nblocks * slice] floats (double-buffered by phase parity so that a fast workgroup's next slice cannot race a slow reader)
template <bool FENCES, bool BYXCD>
__global__ __launch_bounds__(kBlock) void k_phases(Bar b, unsigned phases, unsigned slice, float* table, float* priv, unsigned dirty_floats,
                                                  unsigned* mismatches, float* sink, int skip_barrier) {
    const unsigned nblocks = gridDim.x, blk = blockIdx.x;
    const unsigned my_xcd = xcc_id();
    // phase 0 (flat): count the workgroups of every XCD
    if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*) (b.xcd_size + my_xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ unsigned s_size, s_live;
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add((gu32*) (b.chip + 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool good = spin_until(b.chip + 32, nblocks, b.err);
            unsigned live = 0, mine = 0;
            for (unsigned x = 0; x < 8; ++x) {
                const unsigned c = __hip_atomic_load((gu32*) (b.xcd_size + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                live += c != 0;
                if (x == my_xcd) mine = c;
            }
            s_size = good ? mine : 0; s_live = live;
        }
        __syncthreads();
        if (s_size == 0) return;
    }
    const unsigned my_size = s_size, live = s_live;
    const unsigned total = nblocks * slice;
    float acc = 0.f;
    unsigned bad = 0;
    for (unsigned p = 1; p <= phases; ++p) {
        float* tab = table + (size_t) (p & 1u) * total;
        // private work: dirty lines in this XCD's L2
        float* mine = priv + (size_t) blk * dirty_floats;
        for (unsigned i = threadIdx.x; i < dirty_floats; i += kBlock) mine[i] = mine[i] + 1.f;
        // publish the slice
        for (unsigned i = threadIdx.x; i < slice; i += kBlock) {
            const float v = (float) (p * 7u + blk + i);
            if constexpr (FENCES) tab[blk * slice + i] = v;
            else __hip_atomic_store((gu32*) (tab + blk * slice + i), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!skip_barrier && !grid_barrier<FENCES, BYXCD>(b, p, nblocks, my_xcd, my_size, live)) return;
        // the gather: the whole table, plain 16-byte loads
        const float4* tab4 = reinterpret_cast<const float4*>(tab);
        for (unsigned i = threadIdx.x; i < total / 4; i += kBlock) {
            const float4 v = tab4[i];
            const unsigned e = 4 * i, s0 = e / slice, r0 = e % slice;  // (slice is a multiple of 4)
            const float want = (float) (p * 7u + s0 + r0);
            bad += (v.x != want) + (v.y != want + 1.f) + (v.z != want + 2.f) + (v.w != want + 3.f);
            acc += v.x + v.w;
        }
    }
    if (bad && !skip_barrier) atomicAdd(mismatches, bad);
    sink[blk * kBlock + threadIdx.x] = acc;
}

__global__ void k_node(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f; }

int main(int argc, char** argv) {
    const unsigned phases = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned ncu = prop.multiProcessorCount;
    printf("device: %s, %u CUs; %u phases per launch; one %d-thread workgroup per CU unless stated\n", prop.gcnArchName, ncu, phases, kBlock);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned* ctr;
    CK(hipMalloc(&ctr, 4096 * sizeof(unsigned)));
    float *table, *priv, *sink;
    unsigned* mism;
    const unsigned max_slice = 256, max_dirty = 16384;  // floats
    CK(hipMalloc(&table, 2ull * 2 * ncu * max_slice * sizeof(float)));
    CK(hipMalloc(&priv, (size_t) 2 * ncu * max_dirty * sizeof(float)));
    CK(hipMalloc(&sink, 2ull * ncu * kBlock * sizeof(float)));
    CK(hipMalloc(&mism, sizeof(unsigned)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto run = [&](int mode, bool byxcd, unsigned nblocks, unsigned slice, unsigned dirty, int skip = 0) {
        CK(hipMemsetAsync(ctr, 0, 4096 * sizeof(unsigned), st));
        CK(hipMemsetAsync(mism, 0, sizeof(unsigned), st));
        CK(hipMemsetAsync(priv, 0, (size_t) 2 * ncu * max_dirty * sizeof(float), st));
        Bar b{ctr, ctr + 64, ctr + 64 + 8 * 32, ctr + 1024};
        CK(hipEventRecord(e0, st));
        if (mode == 0 && !byxcd) hipLaunchKernelGGL((k_phases<true, false>), dim3(nblocks), dim3(kBlock), 0, st, b, phases, slice, table, priv, dirty, mism, sink, skip);
        if (mode == 0 && byxcd) hipLaunchKernelGGL((k_phases<true, true>), dim3(nblocks), dim3(kBlock), 0, st, b, phases, slice, table, priv, dirty, mism, sink, skip);
        if (mode == 1 && !byxcd) hipLaunchKernelGGL((k_phases<false, false>), dim3(nblocks), dim3(kBlock), 0, st, b, phases, slice, table, priv, dirty, mism, sink, skip);
        if (mode == 1 && byxcd) hipLaunchKernelGGL((k_phases<false, true>), dim3(nblocks), dim3(kBlock), 0, st, b, phases, slice, table, priv, dirty, mism, sink, skip);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err = 0, h_mism = 0, sizes[8];
        CK(hipMemcpy(&h_err, ctr + 1024, sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h_mism, mism, sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(sizes, ctr + 64 + 8 * 32, sizeof(sizes), hipMemcpyDeviceToHost));
        if (skip) { printf("no barrier (same work)      wgs %4u  table %6.1f KB  private %7.1f KB/wg : %7.3f us per phase\n", nblocks, nblocks * slice * 4 / 1024.0, dirty * 4 / 1024.0, ms * 1e3 / phases); return true; }
        printf("%-13s %-7s wgs %4u  table %6.1f KB  private %7.1f KB/wg : %7.3f us per phase   %s%s  [wgs per XCD:", mode == 0 ? "fences" : "write-through",
               byxcd ? "by-XCD" : "flat", nblocks, nblocks * slice * 4 / 1024.0, dirty * 4 / 1024.0, ms * 1e3 / phases, h_err ? "TIMED OUT " : "",
               h_mism ? "STALE READS" : "values ok");
        for (int x = 0; x < 8; ++x) printf(" %u", sizes[x]);
        printf("]\n");
        fflush(stdout);
        return !h_err;
    };

    // baseline: a chain of dependent graph nodes
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_node, dim3(ncu), dim3(kBlock), 0, st, sink);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipGraph replay, 200 dependent near-empty kernels (%u wgs): %.3f us per node\n", ncu, ms * 1e3 / 2000);
    }
    bool ok = true;
    run(1, false, ncu, 64, 0, 1); run(1, false, ncu, 64, 1024, 1); run(1, false, ncu, 64, 12288, 1); run(1, false, ncu, 16, 0, 1); run(1, false, 2 * ncu, 32, 6144, 1);
    for (int mode = 0; mode < 2 && ok; ++mode)
        for (int byxcd = 0; byxcd < 2 && ok; ++byxcd) {
            ok = ok && run(mode, byxcd, ncu, 64, 0);         // barrier + a 64 KB table
            ok = ok && run(mode, byxcd, ncu, 64, 1024);      // + 4 KB of private dirty data per workgroup (ML-1M: 12 MB / 256)
            ok = ok && run(mode, byxcd, ncu, 64, 12288);     // + 48 KB
            ok = ok && run(mode, byxcd, ncu, 16, 0);         // a 16 KB table
            ok = ok && run(mode, byxcd, 2 * ncu, 32, 6144);  // two workgroups per CU
        }
    return ok ? 0 : 2;
}

// ubench_gather.hip -- standalone microbenchmark (NOT part of libmfx): what limits a
// "stream idx/val + gather a small vector + write val" pass on MI355X?
//   build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip
//   run  : ./ubench_gather [nnz_millions]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// GATHER: 0 none, 1 global float2, 2 LDS float2 (slice of `vlen` entries staged per block)
// NT: non-temporal stream hints.  DEPTH: tiles of 256 elements prefetched ahead (1 or 2).
template <int GATHER, bool NT, int DEPTH, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pass(const uint32_t* __restrict__ idx, float* __restrict__ val,
                                                const float2* __restrict__ vec, uint32_t vlen, uint32_t tiles_per_wave,
                                                uint64_t nnz, float* __restrict__ out) {
    extern __shared__ float2 lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (GATHER == 2) {
        for (uint32_t i = threadIdx.x; i < vlen; i += BLOCK) lds[i] = vec[i];
        __syncthreads();
    }
    const uint64_t start = (uint64_t) wave * tiles_per_wave * 256;
    if (start >= nnz) return;
    const u32x4* i4 = reinterpret_cast<const u32x4*>(idx + start) + lane;
    f32x4* v4 = reinterpret_cast<f32x4*>(val + start) + lane;
    float g = 0.f, h = 0.f;
    u32x4 idn[DEPTH];
    f32x4 vn[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        idn[d] = NT ? __builtin_nontemporal_load(i4 + d * 64) : i4[d * 64];
        vn[d] = NT ? __builtin_nontemporal_load(v4 + d * 64) : v4[d * 64];
    }
    for (uint32_t t = 0; t < tiles_per_wave; t += DEPTH) {
        u32x4 id[DEPTH];
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { id[d] = idn[d]; v[d] = vn[d]; }
        if (t + DEPTH < tiles_per_wave) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                idn[d] = NT ? __builtin_nontemporal_load(i4 + (t + DEPTH + d) * 64) : i4[(t + DEPTH + d) * 64];
                vn[d] = NT ? __builtin_nontemporal_load(v4 + (t + DEPTH + d) * 64) : v4[(t + DEPTH + d) * 64];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float2 a;
                if (GATHER == 0) a = make_float2(1.0f + (float) (id[d][e] & 1), 0.5f);
                else if (GATHER == 1) a = vec[id[d][e]];
                else a = lds[id[d][e]];
                const float nv = (v[d][e] - a.x * 0.25f) + a.y * 0.125f;
                o[e] = nv;
                g += a.y * nv;
                h += a.y * a.y;
            }
            if (NT) __builtin_nontemporal_store(o, v4 + (t + d) * 64);
            else v4[(t + d) * 64] = o;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); h += __shfl_xor(h, o, 64); }
    if (lane == 0) { out[2 * wave] = g; out[2 * wave + 1] = h; }
}


using u16x8 = __attribute__((ext_vector_type(8))) uint16_t;
using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;

// EPL = elements per lane: 4 (u16x4 idx + one f32x4) or 8 (u16x8 idx + two f32x4 at 32-B lane stride)
template <int EPL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pass16(const uint16_t* __restrict__ idx, float* __restrict__ val,
                                                  const float2* __restrict__ vec, uint32_t vlen, uint32_t tiles_per_wave,
                                                  uint64_t nnz, float* __restrict__ out) {
    extern __shared__ float2 lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < vlen; i += BLOCK) lds[i] = vec[i];
    __syncthreads();
    constexpr int TILE = 64 * EPL;
    const uint64_t start = (uint64_t) wave * tiles_per_wave * TILE;
    if (start >= nnz) return;
    float g = 0.f, h = 0.f;
    if constexpr (EPL == 4) {
        const u16x4* i4 = reinterpret_cast<const u16x4*>(idx + start) + lane;
        f32x4* v4 = reinterpret_cast<f32x4*>(val + start) + lane;
        u16x4 idn = __builtin_nontemporal_load(i4);
        f32x4 vn = __builtin_nontemporal_load(v4);
        for (uint32_t t = 0; t < tiles_per_wave; ++t) {
            const u16x4 id = idn; const f32x4 v = vn;
            if (t + 1 < tiles_per_wave) { idn = __builtin_nontemporal_load(i4 + (t + 1) * 64); vn = __builtin_nontemporal_load(v4 + (t + 1) * 64); }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 a = lds[id[e]];
                const float nv = (v[e] - a.x * 0.25f) + a.y * 0.125f;
                o[e] = nv; g += a.y * nv; h += a.y * a.y;
            }
            __builtin_nontemporal_store(o, v4 + t * 64);
        }
    } else {
        const u16x8* i8 = reinterpret_cast<const u16x8*>(idx + start) + lane;
        f32x4* v4 = reinterpret_cast<f32x4*>(val + start) + 2 * lane;
        u16x8 idn = __builtin_nontemporal_load(i8);
        f32x4 vn0 = __builtin_nontemporal_load(v4), vn1 = __builtin_nontemporal_load(v4 + 1);
        for (uint32_t t = 0; t < tiles_per_wave; ++t) {
            const u16x8 id = idn; const f32x4 v0 = vn0, v1 = vn1;
            if (t + 1 < tiles_per_wave) {
                idn = __builtin_nontemporal_load(i8 + (t + 1) * 64);
                vn0 = __builtin_nontemporal_load(v4 + (t + 1) * 128); vn1 = __builtin_nontemporal_load(v4 + (t + 1) * 128 + 1);
            }
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 a = lds[id[e]];
                const float vv = e < 4 ? v0[e & 3] : v1[e & 3];
                const float nv = (vv - a.x * 0.25f) + a.y * 0.125f;
                if (e < 4) o0[e & 3] = nv; else o1[e & 3] = nv;
                g += a.y * nv; h += a.y * a.y;
            }
            __builtin_nontemporal_store(o0, v4 + t * 128);
            __builtin_nontemporal_store(o1, v4 + t * 128 + 1);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); h += __shfl_xor(h, o, 64); }
    if (lane == 0) { out[2 * wave] = g; out[2 * wave + 1] = h; }
}

template <int EPL, int BLOCK>
void run16(const char* name, const uint16_t* idx, float* val, const float2* vec, uint32_t vlen, uint32_t tiles, uint64_t nnz, float* out) {
    const uint64_t per_wave = (uint64_t) tiles * 64 * EPL;
    const uint64_t waves = (nnz + per_wave - 1) / per_wave;
    const uint32_t grid = (uint32_t) ((waves + BLOCK / 64 - 1) / (BLOCK / 64));
    const size_t lds = (size_t) vlen * sizeof(float2);
    if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pass16<EPL, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_pass16<EPL, BLOCK>), dim3(grid), dim3(BLOCK), lds, 0, idx, val, vec, vlen, tiles, nnz, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-44s vlen=%8u tiles=%3u block=%4d  %8.1f us  %7.1f GB/s (10 B/nnz)\n", name, vlen, tiles, BLOCK, best * 1e3, 10.0 * nnz / (best * 1e-3) / 1e9);
}

template <int GATHER, bool NT, int DEPTH, int BLOCK>
void run(const char* name, const uint32_t* idx, float* val, const float2* vec, uint32_t vlen, uint32_t tiles,
         uint64_t nnz, float* out) {
    const uint64_t waves = (nnz + (uint64_t) tiles * 256 - 1) / ((uint64_t) tiles * 256);
    const uint32_t grid = (uint32_t) ((waves + BLOCK / 64 - 1) / (BLOCK / 64));
    const size_t lds = GATHER == 2 ? (size_t) vlen * sizeof(float2) : 0;
    if (lds > 48 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pass<GATHER, NT, DEPTH, BLOCK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_pass<GATHER, NT, DEPTH, BLOCK>), dim3(grid), dim3(BLOCK), lds, 0, idx, val, vec, vlen, tiles, nnz, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-44s vlen=%8u tiles=%3u block=%4d  %8.1f us  %7.1f GB/s (12 B/nnz)\n", name, vlen, tiles, BLOCK, best * 1e3,
           12.0 * nnz / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const uint64_t nnz = (uint64_t) (argc > 1 ? atoi(argv[1]) : 96) * 1024 * 1024;
    uint32_t* idx; float* val; float2* vec; float* out;
    const uint32_t vmax = 1 << 20;
    CK(hipMalloc(&idx, nnz * 4)); CK(hipMalloc(&val, nnz * 4)); CK(hipMalloc(&vec, vmax * 8)); CK(hipMalloc(&out, 64 << 20));
    std::vector<float2> hv(vmax, make_float2(0.5f, 0.25f));
    CK(hipMemcpy(vec, hv.data(), vmax * 8, hipMemcpyHostToDevice));
    CK(hipMemset(val, 0, nnz * 4));

    if (argc > 2 && std::string(argv[2]) == "sweep") {
        // global (L2 / Infinity Cache served) 8-byte gathers against the table size: what a panel cut
        // at L2 granularity would buy for matrices whose segments are too short for LDS panels
        std::vector<uint32_t> hs(nnz);
        for (uint32_t vlen : {16384u, 65536u, 131072u, 262144u, 524288u, 1048576u}) {
            uint64_t s = 1234567;
            for (uint64_t i = 0; i < nnz; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hs[i] = (uint32_t) (s % vlen); }
            CK(hipMemcpy(idx, hs.data(), nnz * 4, hipMemcpyHostToDevice));
            printf("---- vlen %u = %u KB table (uniform random) ----\n", vlen, vlen * 8 / 1024);
            run<1, true, 2, 256>("global gather f2, nt, depth2", idx, val, vec, vlen, 12, nnz, out);
        }
        return 0;
    }
    {   // 16-bit indices: 4 vs 8 elements per lane
        const uint32_t vlen = 7168;
        std::vector<uint16_t> h16(nnz);
        uint64_t s2 = 424242;
        for (uint64_t i = 0; i < nnz; ++i) { s2 ^= s2 << 13; s2 ^= s2 >> 7; s2 ^= s2 << 17; h16[i] = (uint16_t) (s2 % vlen); }
        uint16_t* d16; CK(hipMalloc(&d16, nnz * 2)); CK(hipMemcpy(d16, h16.data(), nnz * 2, hipMemcpyHostToDevice));
        printf("---- u16 indices, vlen %u ----\n", vlen);
        run16<4, 1024>("u16 LDS gather, 4 el/lane, 1024thr, t8", d16, val, vec, vlen, 8, nnz, out);
        run16<8, 1024>("u16 LDS gather, 8 el/lane, 1024thr, t4", d16, val, vec, vlen, 4, nnz, out);
        run16<8, 1024>("u16 LDS gather, 8 el/lane, 1024thr, t8", d16, val, vec, vlen, 8, nnz, out);
        run16<4, 512>("u16 LDS gather, 4 el/lane, 512thr, t8", d16, val, vec, vlen, 8, nnz, out);
        run16<8, 512>("u16 LDS gather, 8 el/lane, 512thr, t4", d16, val, vec, vlen, 4, nnz, out);
        return 0;
    }
    std::vector<uint32_t> hi(nnz);
    for (uint32_t vlen : {480189u, 17770u, 8192u}) {
        // sorted-ish within runs like a real column: random walk with random restarts
        uint64_t s = 88172645463325252ull;
        uint32_t cur = 0;
        for (uint64_t i = 0; i < nnz; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const uint32_t step = (uint32_t) (s % (vlen / 64 + 2));
            cur += 1 + step;
            if (cur >= vlen) cur = (uint32_t) ((s >> 32) % 64);
            hi[i] = cur;
        }
        CK(hipMemcpy(idx, hi.data(), nnz * 4, hipMemcpyHostToDevice));
        printf("---- vlen %u (ascending runs, mean stride %u) ----\n", vlen, vlen / 128 + 1);
        run<0, true, 1, 256>("stream only, nt, depth1", idx, val, vec, vlen, 12, nnz, out);
        run<0, false, 1, 256>("stream only, plain, depth1", idx, val, vec, vlen, 12, nnz, out);
        run<0, true, 2, 256>("stream only, nt, depth2", idx, val, vec, vlen, 12, nnz, out);
        run<1, true, 1, 256>("global gather f2, nt, depth1", idx, val, vec, vlen, 12, nnz, out);
        run<1, false, 1, 256>("global gather f2, plain, depth1", idx, val, vec, vlen, 12, nnz, out);
        run<1, true, 2, 256>("global gather f2, nt, depth2", idx, val, vec, vlen, 12, nnz, out);
        if (vlen <= 17770) {
            run<2, true, 1, 512>("LDS gather f2, nt, depth1, 512thr", idx, val, vec, vlen, 16, nnz, out);
            run<2, true, 2, 512>("LDS gather f2, nt, depth2, 512thr", idx, val, vec, vlen, 16, nnz, out);
            run<2, true, 2, 1024>("LDS gather f2, nt, depth2, 1024thr", idx, val, vec, vlen, 16, nnz, out);
            run<2, true, 2, 1024>("LDS gather f2, nt, depth2, 1024thr, t32", idx, val, vec, vlen, 32, nnz, out);
            run<2, true, 2, 256>("LDS gather f2, nt, depth2, 256thr t32", idx, val, vec, vlen, 32, nnz, out);
        }
    }
    // fully random indices
    for (uint32_t vlen : {480189u, 8192u}) {
        uint64_t s = 1234567;
        for (uint64_t i = 0; i < nnz; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hi[i] = (uint32_t) (s % vlen); }
        CK(hipMemcpy(idx, hi.data(), nnz * 4, hipMemcpyHostToDevice));
        printf("---- vlen %u (uniform random) ----\n", vlen);
        run<1, true, 2, 256>("global gather f2, nt, depth2", idx, val, vec, vlen, 12, nnz, out);
        if (vlen <= 17770) run<2, true, 2, 1024>("LDS gather f2, nt, depth2, 1024thr", idx, val, vec, vlen, 16, nnz, out);
    }
    return 0;
}

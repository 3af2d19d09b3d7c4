// ubench_scatter.hip -- standalone microbenchmark (NOT part of libmfx): can a hyper-sparse pass avoid
// random global gathers altogether?
//
// Shape under test: config 5's shard (1.25 M x 1 M, 125 M ratings per GPU).  Idea: store the residual
// copy panel-major over the REDUCED dimension (panels of <= 8192 columns: their (g, h) accumulators and
// their operands sit in LDS, 16-bit local column index per entry), entries inside a panel sorted by the
// other index (row): the row operand is then an ascending, nearly sequential global read (a wave's 256
// entries span ~320 rows = ~20 cache lines instead of 256), and the per-column reduction becomes a
// scatter-add into LDS -- made order-independent, hence bitwise reproducible, by accumulating 64-bit
// fixed point with ds_add_u64.
//
//   build: hipcc --offload-arch=gfx950 -O3 -o ubench_scatter tools/ubench_scatter.hip
//   run  : ./ubench_scatter [nnz_millions] [rows] [panel_cols]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr float kScale = 68719476736.0f;  // 2^36

__device__ __forceinline__ unsigned long long to_fixed(float x) {
    // exact: |x| * 2^36 < 2^63; two's complement through the double
    const double d = (double) x * (double) kScale;
    return (unsigned long long) (long long) d;
}

// ROWG: 0 = no row operand (constant), 1 = global gather pack[row] (ascending rows), 2 = the same with random rows
// ACC : 0 = none (register sums), 1 = LDS fp32 atomics (non-deterministic), 2 = LDS 64-bit fixed point atomics
template <int ROWG, int ACC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_scatter(const uint32_t* __restrict__ row, const uint16_t* __restrict__ col,
                                                   float* __restrict__ val, const float2* __restrict__ rowpack,
                                                   const float2* __restrict__ colpack, uint32_t pcols, uint32_t tiles_per_wave,
                                                   uint64_t nnz, unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2* cslice = reinterpret_cast<float2*>(lds_raw);                                        // [pcols] per-column operands
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds_raw + (size_t) pcols * 8);  // [pcols][2] (g, h)
    float* accf = reinterpret_cast<float*>(acc);
    for (uint32_t i = threadIdx.x; i < pcols; i += BLOCK) cslice[i] = colpack[i];
    for (uint32_t i = threadIdx.x; i < 2 * pcols; i += BLOCK) acc[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t start = (uint64_t) wave * tiles_per_wave * 256;
    float g = 0.f, h = 0.f;
    if (start < nnz) {
        const u32x4* r4 = reinterpret_cast<const u32x4*>(row + start) + lane;
        const u16x4* c4 = reinterpret_cast<const u16x4*>(col + start) + lane;
        f32x4* v4 = reinterpret_cast<f32x4*>(val + start) + lane;
        u32x4 rn = __builtin_nontemporal_load(r4);
        u16x4 cn = __builtin_nontemporal_load(c4);
        f32x4 vn = __builtin_nontemporal_load(v4);
        for (uint32_t t = 0; t < tiles_per_wave; ++t) {
            const u32x4 r = rn; const u16x4 c = cn; const f32x4 v = vn;
            if (t + 1 < tiles_per_wave) {
                rn = __builtin_nontemporal_load(r4 + (t + 1) * 64);
                cn = __builtin_nontemporal_load(c4 + (t + 1) * 64);
                vn = __builtin_nontemporal_load(v4 + (t + 1) * 64);
            }
            float2 rp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rp[e] = ROWG ? rowpack[r[e]] : make_float2(0.5f + (float) (r[e] & 1), 0.25f);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 cp = cslice[c[e]];
                const float nv = (v[e] - rp[e].x * cp.x) + rp[e].y * cp.y;
                o[e] = nv;
                const float gc = rp[e].y * nv, hc = rp[e].y * rp[e].y;
                if (ACC == 0) { g += gc; h += hc; }
                else if (ACC == 1) { atomicAdd(&accf[4 * c[e]], gc); atomicAdd(&accf[4 * c[e] + 2], hc); }
                else { atomicAdd(&acc[2 * c[e]], to_fixed(gc)); atomicAdd(&acc[2 * c[e] + 1], to_fixed(hc)); }
            }
            __builtin_nontemporal_store(o, v4 + t * 64);
        }
    }
    __syncthreads();
    // flush: per-workgroup partial accumulators (integer sums: any combination order gives the same bits)
    unsigned long long* dst = out + (size_t) blockIdx.x * 2 * pcols;
    for (uint32_t i = threadIdx.x; i < 2 * pcols; i += BLOCK) dst[i] = acc[i] + (ACC == 0 ? (unsigned long long) (g + h) : 0ull);
}

template <int ROWG, int ACC, int BLOCK>
void run(const char* name, const uint32_t* row, const uint16_t* col, float* val, const float2* rowpack, const float2* colpack,
         uint32_t pcols, uint32_t tiles, uint64_t nnz, unsigned long long* out) {
    const uint64_t waves = (nnz + (uint64_t) tiles * 256 - 1) / ((uint64_t) tiles * 256);
    const uint32_t grid = (uint32_t) ((waves + BLOCK / 64 - 1) / (BLOCK / 64));
    const size_t lds = (size_t) pcols * 8 + (size_t) pcols * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<ROWG, ACC, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_scatter<ROWG, ACC, BLOCK>), dim3(grid), dim3(BLOCK), lds, 0, row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-58s pcols=%5u tiles=%3u block=%4d grid=%6u  %8.1f us  %6.1f Gnnz/s  %7.1f GB/s (12 B/nnz contract)\n", name, pcols, tiles, BLOCK,
           grid, best * 1e3, nnz / (best * 1e-3) / 1e9, 12.0 * nnz / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const uint64_t nnz = (uint64_t) (argc > 1 ? atoi(argv[1]) : 120) * 1024 * 1024;
    const uint32_t rows = argc > 2 ? (uint32_t) atoi(argv[2]) : 1250000u;
    const uint32_t pcols = argc > 3 ? (uint32_t) atoi(argv[3]) : 6144u;
    uint32_t *row, *rowr; uint16_t* col; float* val; float2 *rowpack, *colpack; unsigned long long* out;
    CK(hipMalloc(&row, nnz * 4)); CK(hipMalloc(&rowr, nnz * 4)); CK(hipMalloc(&col, nnz * 2)); CK(hipMalloc(&val, nnz * 4));
    CK(hipMalloc(&rowpack, (size_t) rows * 8)); CK(hipMalloc(&colpack, (size_t) pcols * 8));
    CK(hipMalloc(&out, (size_t) 1 << 30));
    std::vector<float2> hp(rows, make_float2(0.5f, 0.25f));
    CK(hipMemcpy(rowpack, hp.data(), (size_t) rows * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(colpack, hp.data(), (size_t) pcols * 8, hipMemcpyHostToDevice));
    CK(hipMemset(val, 0, nnz * 4));
    // a panel = pcols columns of a 1M-column matrix at 100 entries per row: entries per (panel, row) ~ Poisson(100 * pcols / 1e6);
    // the stream is the concatenation of panels, rows ascending inside each
    std::vector<uint32_t> hr(nnz), hrr(nnz);
    std::vector<uint16_t> hc(nnz);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const double mean_per_row = 100.0 * pcols / 1.0e6;
    uint32_t cur = 0;
    for (uint64_t i = 0; i < nnz; ++i) {
        // geometric gap with mean 1 / mean_per_row rows between consecutive entries
        const double u = (double) (rnd() >> 11) / 9007199254740992.0;
        const uint32_t gap = (uint32_t) (-__builtin_log(1.0 - u) / mean_per_row);
        cur += gap;
        if (cur >= rows) cur = (uint32_t) (rnd() % 8);  // next panel
        hr[i] = cur;
        hrr[i] = (uint32_t) (rnd() % rows);
        hc[i] = (uint16_t) (rnd() % pcols);
    }
    if (argc > 4) {
        // "real" mode: draw nnz uniform (i, j) pairs of a rows x 1M matrix and lay them out exactly as the solver does --
        // panel-major over the columns, row-major inside a panel (stable counting sort by panel of the row-sorted pairs)
        const uint32_t cols = 1000000u, P = (cols + pcols - 1) / pcols;
        std::vector<uint64_t> key(nnz);
        for (uint64_t i = 0; i < nnz; ++i) key[i] = ((uint64_t) (rnd() % rows) << 32) | (uint32_t) (rnd() % cols);
        std::sort(key.begin(), key.end());  // row-major (CSR order)
        std::vector<uint64_t> cnt(P + 1, 0);
        for (uint64_t i = 0; i < nnz; ++i) ++cnt[(uint32_t) key[i] / pcols + 1];
        for (uint32_t p = 0; p < P; ++p) cnt[p + 1] += cnt[p];
        for (uint64_t i = 0; i < nnz; ++i) {
            const uint32_t j = (uint32_t) key[i], p = j / pcols;
            const uint64_t d = cnt[p]++;
            hr[d] = (uint32_t) (key[i] >> 32);
            hc[d] = (uint16_t) (j - p * pcols);
        }
        printf("REAL layout: %u panels, rows ascending inside each\n", P);
    }
    CK(hipMemcpy(row, hr.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rowr, hrr.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(col, hc.data(), nnz * 2, hipMemcpyHostToDevice));
    printf("nnz %llu, rows %u, panel of %u columns (%.2f entries per (panel,row)), 14 B/nnz streamed\n", (unsigned long long) nnz, rows, pcols, mean_per_row);
    for (uint32_t tiles : {16u, 64u}) {
        run<0, 0, 1024>("stream + LDS col gather, register sums", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<1, 0, 1024>("+ ascending global row operand", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<2, 0, 1024>("+ RANDOM global row operand (today's cache-panel bound)", rowr, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<0, 1, 1024>("stream + LDS fp32 atomics", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<0, 2, 1024>("stream + LDS u64 fixed-point atomics", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<1, 2, 1024>("ascending row operand + LDS u64 atomics (the design)", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<1, 2, 512>("ascending row operand + LDS u64 atomics, 512 thr", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
        run<1, 1, 1024>("ascending row operand + LDS fp32 atomics", row, col, val, rowpack, colpack, pcols, tiles, nnz, out);
    }
    return 0;
}

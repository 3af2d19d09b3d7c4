// layout_kernels.hip -- device pipeline behind SegStreamStore::build (see layout_kernels.hpp).
// All of it is HBM-bound integer streaming: coalesced 16-byte walks, one binary search per thread
// (not per element), integer atomics only where two threads may meet on a word (run bounds, head bits).
#include "layout_kernels.hpp"

#include <algorithm>

namespace mfx {
namespace {

constexpr int kLB = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;

#define LK_LAUNCH_CHECK() MFX_HIP(hipGetLastError())

inline uint32_t grid_for(uint64_t n, uint32_t per_block) { return (uint32_t) std::max<uint64_t>(1, (n + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLB) void k_check_ptr(uint32_t nseg, uint32_t nnz, const uint32_t* __restrict__ ptr,
                                                   uint32_t* __restrict__ bad) {
    const uint32_t c = blockIdx.x * kLB + threadIdx.x;
    if (c == 0 && (ptr[0] != 0 || ptr[nseg] != nnz)) atomicMin(bad, nseg);  // "does not span [0, nnz]"
    if (c < nseg && ptr[c] > ptr[c + 1]) atomicMin(bad, c);
}

// Run bounds of every (panel, segment) pair.  A thread owns 4 consecutive input positions; it finds
// the segment of the first by binary search over ptr (L2-resident) and walks on from there.
__global__ __launch_bounds__(kLB) void k_runs(LayoutBuildIn in, uint32_t* __restrict__ first_q, uint32_t* __restrict__ last_q,
                                              unsigned long long* __restrict__ first_bad) {
    const uint64_t q0 = ((uint64_t) blockIdx.x * kLB + threadIdx.x) * 4;
    if (q0 >= in.nnz) return;
    const uint32_t* __restrict__ ptr = in.ptr;
    const uint32_t* __restrict__ idx = in.idx;
    uint32_t lo = 0, hi = in.nseg;  // first c with ptr[c + 1] > q0
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (ptr[mid + 1] > (uint32_t) q0) hi = mid; else lo = mid + 1;
    }
    uint32_t c = lo;
    uint32_t seg_end = ptr[c + 1], seg_beg = ptr[c];
    const uint32_t n = in.nnz - q0 < 4 ? (uint32_t) (in.nnz - q0) : 4u;
    uint32_t id[6];  // idx[q0 - 1 .. q0 + 4], where they exist
    id[0] = q0 > 0 ? idx[q0 - 1] : 0u;
#pragma unroll
    for (int e = 0; e < 5; ++e) id[e + 1] = q0 + e < in.nnz ? idx[q0 + e] : 0u;
    const uint32_t PR = in.panel_rows;
    uint32_t pn[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) pn[e] = id[e] / PR;
    unsigned long long bad = ~0ull;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if ((uint32_t) e >= n) break;
        const uint32_t q = (uint32_t) q0 + e;
        while (q >= seg_end) { ++c; seg_beg = seg_end; seg_end = ptr[c + 1]; }
        if (id[e + 1] >= in.G) { if (q < bad) bad = q; continue; }
        const uint32_t p = pn[e + 1];
        const size_t v = (size_t) p * in.nseg + c;
        if (q == seg_beg || pn[e] != p) atomicMin(&first_q[v], q);
        if (q + 1 == seg_end || pn[e + 2] != p) atomicMax(&last_q[v], q);
    }
    if (bad != ~0ull) atomicMin(first_bad, bad);
}

// cnt[v] <- last_q[v] - first_q[v] + 1 (in place over last_q), 0 for an untouched pair; total in 64 bits
__global__ __launch_bounds__(kLB) void k_counts(size_t nv, const uint32_t* __restrict__ first_q, uint32_t* __restrict__ cnt,
                                                unsigned long long* __restrict__ total) {
    __shared__ unsigned long long red[kLB / 64];
    unsigned long long acc = 0;
    for (size_t v = (size_t) blockIdx.x * kLB + threadIdx.x; v < nv; v += (size_t) gridDim.x * kLB) {
        const uint32_t f = first_q[v];
        const uint32_t c = f == kNone ? 0u : cnt[v] - f + 1u;
        cnt[v] = c;
        acc += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < kLB / 64; ++w) s += red[w];
        if (s) atomicAdd(total, s);
    }
}

// one panel: a pair is a whole segment
__global__ __launch_bounds__(kLB) void k_counts_single(uint32_t nseg, const uint32_t* __restrict__ ptr, uint32_t* __restrict__ first_q,
                                                       uint32_t* __restrict__ cnt) {
    const uint32_t c = blockIdx.x * kLB + threadIdx.x;
    if (c >= nseg) return;
    first_q[c] = ptr[c];
    cnt[c] = ptr[c + 1] - ptr[c];
}

__global__ __launch_bounds__(kLB) void k_check_range(uint64_t n, const uint32_t* __restrict__ idx, uint32_t bound,
                                                     unsigned long long* __restrict__ first_bad) {
    unsigned long long bad = ~0ull;
    for (uint64_t q = (uint64_t) blockIdx.x * kLB + threadIdx.x; q < n; q += (uint64_t) gridDim.x * kLB)
        if (idx[q] >= bound && q < bad) bad = q;
    if (bad != ~0ull) atomicMin(first_bad, bad);
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan, three launches: per-block sums, scan of the block sums (one workgroup), per-block
// scan + offset.  A block owns kScanItems consecutive inputs, a thread 16 of them.
constexpr int kScanPer = 16, kScanItems = kLB * kScanPer;

template <bool POPC> __device__ __forceinline__ uint32_t scan_f(uint32_t x) { return POPC ? (uint32_t) __popc(x) : x; }

__device__ __forceinline__ uint32_t block_exclusive(uint32_t x, uint32_t* wave_tot /* [kLB/64] */, uint32_t& block_total) {
    // inclusive scan inside the wave, then across the 4 waves through LDS
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t) o) inc += y;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kLB / 64; ++w) {
        if ((uint32_t) w < wave) base += wave_tot[w];
        tot += wave_tot[w];
    }
    block_total = tot;
    __syncthreads();
    return base + inc - x;
}

template <bool POPC>
__global__ __launch_bounds__(kLB) void k_scan_reduce(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wt[kLB / 64];
    const size_t base = (size_t) blockIdx.x * kScanItems + (size_t) threadIdx.x * kScanPer;
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e)
        if (base + e < n) s += scan_f<POPC>(in[base + e]);
    uint32_t tot;
    (void) block_exclusive(s, wt, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// in place over block_sums[0 .. nb); block_sums[nb] <- grand total.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_scan_blocksums(uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t b = threadIdx.x * per, e = b + per < nb ? b + per : nb;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += block_sums[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {  // 1024 serial adds: negligible next to the passes over the data
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t t = part[i]; part[i] = run; run += t; }
        block_sums[nb] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = b; i < e; ++i) { const uint32_t t = block_sums[i]; block_sums[i] = run; run += t; }
}

template <bool POPC>
__global__ __launch_bounds__(kLB) void k_scan_final(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                    const uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t wt[kLB / 64];
    const size_t base = (size_t) blockIdx.x * kScanItems + (size_t) threadIdx.x * kScanPer;
    uint32_t x[kScanPer];
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        x[e] = base + e < n ? scan_f<POPC>(in[base + e]) : 0u;
        s += x[e];
    }
    uint32_t tot;
    uint32_t run = block_exclusive(s, wt, tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        if (base + e < n) out[base + e] = run;
        run += x[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nb];
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLB) void k_ptr_v(const uint32_t* __restrict__ S, const uint32_t* __restrict__ delta, uint32_t nseg,
                                               size_t nv, uint32_t padded, uint32_t* __restrict__ ptr_v) {
    const size_t v = (size_t) blockIdx.x * kLB + threadIdx.x;
    if (v > nv) return;
    ptr_v[v] = v < nv ? S[v] + delta[v / nseg] : padded;
}

__global__ void k_panel_starts(const uint32_t* __restrict__ S, uint32_t nseg, uint32_t P, uint32_t* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= P) out[p] = S[(size_t) p * nseg];
}

__global__ __launch_bounds__(kLB) void k_heads(const uint32_t* __restrict__ ptr_v, size_t nv, uint32_t* __restrict__ flags32) {
    const size_t v = (size_t) blockIdx.x * kLB + threadIdx.x;
    if (v >= nv) return;
    const uint32_t lo = ptr_v[v], hi = ptr_v[v + 1];
    if (hi > lo) atomicOr(&flags32[lo >> 5], 1u << (lo & 31));
}

__global__ __launch_bounds__(kLB) void k_ranks(const uint32_t* __restrict__ ptr_v, size_t nv, uint32_t nseg, uint32_t span_len,
                                               const uint32_t* __restrict__ flags32, const uint32_t* __restrict__ hpre,
                                               uint32_t* __restrict__ rank_code, uint32_t* __restrict__ seg_of_rank,
                                               uint32_t* __restrict__ v_of_rank) {
    const size_t v = (size_t) blockIdx.x * kLB + threadIdx.x;
    if (v >= nv) return;
    const uint32_t lo = ptr_v[v], hi = ptr_v[v + 1];
    if (hi <= lo) { rank_code[v] = kNone; return; }
    const uint32_t w = lo >> 5;
    const uint32_t rank = hpre[w] + (uint32_t) __popc(flags32[w] & ((1u << (lo & 31)) - 1u));
    rank_code[v] = rank | ((lo / span_len != (hi - 1) / span_len) ? 0x80000000u : 0u);
    seg_of_rank[rank] = (uint32_t) (v % nseg);
    v_of_rank[rank] = (uint32_t) v;
}

// One pass over the stored positions, 4 per thread.  rank -> virtual segment -> source run: no search.
// TR: transposed tiles (scatter layout) -- the thread's four stored positions tile + 4l + e hold the entries
// tile + 64e + l of the panel-major order, so each needs its own head-count lookup.
template <bool IDX16, bool TR>
__global__ __launch_bounds__(kLB) void k_place(LayoutBuildIn in, uint64_t padded, const uint32_t* __restrict__ ptr_v,
                                               const uint32_t* __restrict__ first_q, const uint32_t* __restrict__ cnt,
                                               const uint32_t* __restrict__ flags32, const uint32_t* __restrict__ hpre,
                                               const uint32_t* __restrict__ v_of_rank, void* __restrict__ idx_out,
                                               float* __restrict__ val_out, uint32_t* __restrict__ seg_out) {
    const uint64_t d0 = ((uint64_t) blockIdx.x * kLB + threadIdx.x) * 4;
    if (d0 >= padded) return;
    uint32_t cur = kNone, pv = 0, cn = 0, fq = 0, pbase = 0, seg = 0;
    uint32_t oi[4], os[4];
    float ov[4];
    uint32_t r1 = 0, fl = 0, sh = 0;
    if constexpr (!TR) {
        const uint32_t w = (uint32_t) (d0 >> 5);
        sh = (uint32_t) (d0 & 31);
        fl = flags32[w];
        r1 = hpre[w] + (uint32_t) __popc(fl & ((1u << sh) - 1u));  // heads before d0
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint32_t sd;    // position in the panel-major order
        uint32_t rank;  // >= 0: the first stored position is a head
        if constexpr (TR) {
            sd = (uint32_t) (d0 & ~255ull) + 64u * e + (uint32_t) ((d0 & 255) >> 2);
            const uint32_t w = sd >> 5, s2 = sd & 31;
            const uint32_t f2 = flags32[w];
            rank = hpre[w] + (uint32_t) __popc(f2 & ((2u << s2) - 1u)) - 1u;  // heads at or before sd, minus one
        } else {
            sd = (uint32_t) d0 + e;
            if ((fl >> (sh + e)) & 1u) ++r1;
            rank = r1 - 1;
        }
        if (rank != cur) {
            cur = rank;
            const uint32_t v = v_of_rank[rank];
            pv = ptr_v[v]; cn = cnt[v]; fq = first_q[v];
            pbase = in.local_idx ? (v / in.nseg) * in.panel_rows : 0u;
            seg = v % in.nseg;
        }
        const uint32_t off = sd - pv;
        if (off < cn) {
            const uint32_t q = fq + off;
            oi[e] = in.idx[q] - pbase;
            ov[e] = in.val ? in.val[q] : 0.f;
            os[e] = seg;
        } else {  // padding folded into the panel's last virtual segment
            oi[e] = in.pad_index;
            ov[e] = 0.f;
            os[e] = 0u;
        }
    }
    if constexpr (IDX16) {
        uint2 pk;
        pk.x = (oi[0] & 0xFFFFu) | (oi[1] << 16);
        pk.y = (oi[2] & 0xFFFFu) | (oi[3] << 16);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(idx_out) + d0) = pk;
    } else {
        *reinterpret_cast<uint4*>(static_cast<uint32_t*>(idx_out) + d0) = make_uint4(oi[0], oi[1], oi[2], oi[3]);
    }
    *reinterpret_cast<float4*>(val_out + d0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (seg_out) *reinterpret_cast<uint4*>(seg_out + d0) = make_uint4(os[0], os[1], os[2], os[3]);
}

__global__ __launch_bounds__(kLB) void k_max_wg_ranks(const uint32_t* __restrict__ hpre, size_t nwords, size_t chunk_words,
                                                      size_t nchunks, uint32_t* __restrict__ out) {
    const size_t c = (size_t) blockIdx.x * kLB + threadIdx.x;
    if (c >= nchunks) return;
    const size_t w0 = c * chunk_words;
    const size_t w1 = w0 + chunk_words < nwords ? w0 + chunk_words : nwords;
    const uint32_t lo = hpre[w0] > 0 ? hpre[w0] - 1 : 0;
    atomicMax(out, hpre[w1] - lo);
}


// Scatter layout: the per-element segment ids of a tile (ascending in the tile's sorted order; padding carries 0)
// as one base per tile + one byte per element = the step from the previous sorted entry.  One wavefront per
// tile; tiles are stored transposed (sorted entry 64 e + l sits at tile + 4 l + e), so lane l's four bytes are
// one 32-bit store.  Padding repeats the last real id (running maximum).  *overflow is set when a step does not
// fit a byte; the caller then keeps the 32-bit ids.
__global__ __launch_bounds__(kLB) void k_delta_encode(uint64_t ntiles, const uint32_t* __restrict__ seg, uint32_t* __restrict__ delta4,
                                                      uint32_t* __restrict__ tile_base, uint32_t* __restrict__ overflow) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t tile = ((uint64_t) blockIdx.x * kLB + threadIdx.x) >> 6;
    if (tile >= ntiles) return;
    const uint4 s = reinterpret_cast<const uint4*>(seg + tile * 256)[lane];
    uint32_t m[4] = {s.x, s.y, s.z, s.w};
    uint32_t carry = 0, word = 0;
    bool over = false;
    for (int e = 0; e < 4; ++e) {
        uint32_t x = m[e];                              // running maximum over the sorted order, group e
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (lane >= (uint32_t) off && y > x) x = y;
        }
        if (carry > x) x = carry;
        uint32_t prev = __shfl_up(x, 1, 64);            // the sorted predecessor: lane - 1, or the previous group's last lane
        if (lane == 0) prev = e == 0 ? x : carry;
        carry = __shfl(x, 63, 64);
        const uint32_t d = x - prev;
        over |= d > 255u;
        word |= (d & 255u) << (8 * e);
        if (e == 0 && lane == 0) tile_base[tile] = x;
    }
    delta4[tile * 64 + lane] = word;
    if (over) atomicOr(overflow, 1u);
}

// small device words with a host mirror
template <typename T>
int read_word(const T* d, T* h, hipStream_t st) {
    MFX_HIP(hipMemcpyAsync(h, d, sizeof(T), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    return MFX_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
int lk_check_ptr(const LayoutBuildIn& in, hipStream_t st) {
    DevBuf<uint32_t> bad;
    MFX_TRY(bad.alloc(1));
    MFX_HIP(hipMemsetAsync(bad.get(), 0xFF, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_check_ptr, dim3(grid_for((uint64_t) in.nseg + 1, kLB)), dim3(kLB), 0, st, in.nseg, (uint32_t) in.nnz, in.ptr,
                       bad.get());
    LK_LAUNCH_CHECK();
    uint32_t b = kNone;
    MFX_TRY(read_word(bad.get(), &b, st));
    if (b == kNone) return MFX_OK;
    if (b == in.nseg) return fail(MFX_ERR_INVALID, "segment pointer array does not span [0, nnz]");
    return fail(MFX_ERR_INVALID, "segment pointer array is not monotone at %u", b);
}

int lk_runs_and_counts(const LayoutBuildIn& in, uint32_t* first_q, uint32_t* cnt, bool* grouped, hipStream_t st) {
    *grouped = true;
    const size_t nv = (size_t) in.npanels * in.nseg;
    DevBuf<unsigned long long> words;  // [0] first bad position, [1] sum of counts
    MFX_TRY(words.alloc(2));
    const unsigned long long init[2] = {~0ull, 0ull};
    MFX_HIP(hipMemcpyAsync(words.get(), init, sizeof(init), hipMemcpyHostToDevice, st));
    if (in.npanels <= 1) {
        hipLaunchKernelGGL(k_counts_single, dim3(grid_for(in.nseg, kLB)), dim3(kLB), 0, st, in.nseg, in.ptr, first_q, cnt);
        LK_LAUNCH_CHECK();
        if (in.nnz) {
            hipLaunchKernelGGL(k_check_range, dim3((uint32_t) std::min<uint64_t>(grid_for(in.nnz, kLB), 256 * 16)), dim3(kLB), 0, st, in.nnz,
                               in.idx, in.G, words.get());
            LK_LAUNCH_CHECK();
        }
    } else {
        MFX_HIP(hipMemsetAsync(first_q, 0xFF, sizeof(uint32_t) * nv, st));
        MFX_HIP(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * nv, st));
        if (in.nnz) {
            hipLaunchKernelGGL(k_runs, dim3(grid_for(in.nnz, kLB * 4)), dim3(kLB), 0, st, in, first_q, cnt, words.get());
            LK_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_counts, dim3((uint32_t) std::min<uint64_t>(grid_for(nv, kLB), 256 * 16)), dim3(kLB), 0, st, nv, first_q, cnt,
                           words.get() + 1);
        LK_LAUNCH_CHECK();
    }
    unsigned long long h[2] = {~0ull, 0ull};
    MFX_HIP(hipMemcpyAsync(h, words.get(), sizeof(h), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    if (h[0] != ~0ull) {
        uint32_t v = 0;
        MFX_HIP(hipMemcpy(&v, in.idx + h[0], sizeof(v), hipMemcpyDeviceToHost));
        return fail(MFX_ERR_INVALID, "index %u at position %llu is out of range [0, %u)", v, h[0], in.G);
    }
    if (in.npanels > 1 && h[1] != in.nnz) *grouped = false;  // some pair spans foreign entries: visited more than once
    return MFX_OK;
}

size_t scan_scratch_words(size_t n) { return (n + kScanItems - 1) / kScanItems + 2; }

int lk_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, bool popcount, uint32_t* scratch, hipStream_t st) {
    const uint32_t nb = (uint32_t) std::max<size_t>(1, (n + kScanItems - 1) / kScanItems);
    if (popcount) hipLaunchKernelGGL(k_scan_reduce<true>, dim3(nb), dim3(kLB), 0, st, in, n, scratch);
    else hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nb), dim3(kLB), 0, st, in, n, scratch);
    LK_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, st, scratch, nb);
    LK_LAUNCH_CHECK();
    if (popcount) hipLaunchKernelGGL(k_scan_final<true>, dim3(nb), dim3(kLB), 0, st, in, out, n, scratch, nb);
    else hipLaunchKernelGGL(k_scan_final<false>, dim3(nb), dim3(kLB), 0, st, in, out, n, scratch, nb);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_ptr_v(const uint32_t* S, const uint32_t* delta, uint32_t nseg, size_t nv, uint32_t padded, uint32_t* ptr_v, hipStream_t st) {
    hipLaunchKernelGGL(k_ptr_v, dim3(grid_for(nv + 1, kLB)), dim3(kLB), 0, st, S, delta, nseg, nv, padded, ptr_v);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_panel_starts(const uint32_t* S, uint32_t nseg, uint32_t P, uint32_t* out, hipStream_t st) {
    hipLaunchKernelGGL(k_panel_starts, dim3(grid_for((uint64_t) P + 1, 64)), dim3(64), 0, st, S, nseg, P, out);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_heads(const uint32_t* ptr_v, size_t nv, uint32_t* flags32, hipStream_t st) {
    if (nv == 0) return MFX_OK;
    hipLaunchKernelGGL(k_heads, dim3(grid_for(nv, kLB)), dim3(kLB), 0, st, ptr_v, nv, flags32);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_ranks(const uint32_t* ptr_v, size_t nv, uint32_t nseg, uint32_t span_len, const uint32_t* flags32, const uint32_t* hpre,
             uint32_t* rank_code, uint32_t* seg_of_rank, uint32_t* v_of_rank, hipStream_t st) {
    if (nv == 0) return MFX_OK;
    hipLaunchKernelGGL(k_ranks, dim3(grid_for(nv, kLB)), dim3(kLB), 0, st, ptr_v, nv, nseg, span_len, flags32, hpre, rank_code,
                       seg_of_rank, v_of_rank);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_place(const LayoutBuildIn& in, uint64_t padded, const uint32_t* ptr_v, const uint32_t* first_q, const uint32_t* cnt,
             const uint32_t* flags32, const uint32_t* hpre, const uint32_t* v_of_rank, void* idx_out, float* val_out,
             uint32_t* seg_out, hipStream_t st) {
    const dim3 grid(grid_for(padded, kLB * 4)), block(kLB);
    if (in.transpose_tiles) {
        MFX_REQUIRE(in.idx16, "transposed tiles come with 16-bit local indices");
        hipLaunchKernelGGL((k_place<true, true>), grid, block, 0, st, in, padded, ptr_v, first_q, cnt, flags32, hpre, v_of_rank, idx_out, val_out, seg_out);
    } else if (in.idx16) {
        hipLaunchKernelGGL((k_place<true, false>), grid, block, 0, st, in, padded, ptr_v, first_q, cnt, flags32, hpre, v_of_rank, idx_out, val_out, seg_out);
    } else {
        hipLaunchKernelGGL((k_place<false, false>), grid, block, 0, st, in, padded, ptr_v, first_q, cnt, flags32, hpre, v_of_rank, idx_out, val_out, seg_out);
    }
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_max_wg_ranks(const uint32_t* hpre, size_t nwords, size_t chunk_words, uint32_t* out, hipStream_t st) {
    const size_t nchunks = (nwords + chunk_words - 1) / chunk_words;
    if (nchunks == 0) return MFX_OK;
    hipLaunchKernelGGL(k_max_wg_ranks, dim3(grid_for(nchunks, kLB)), dim3(kLB), 0, st, hpre, nwords, chunk_words, nchunks, out);
    LK_LAUNCH_CHECK();
    return MFX_OK;
}

int lk_delta_encode(const uint32_t* seg, uint64_t padded, uint8_t* delta, uint32_t* tile_base, bool* fits, hipStream_t st) {
    *fits = true;
    const uint64_t ntiles = padded / 256;
    if (ntiles == 0) return MFX_OK;
    DevBuf<uint32_t> over;
    MFX_TRY(over.alloc_zero(1, st));
    hipLaunchKernelGGL(k_delta_encode, dim3(grid_for(ntiles * 64, kLB)), dim3(kLB), 0, st, ntiles, seg, reinterpret_cast<uint32_t*>(delta),
                       tile_base, over.get());
    LK_LAUNCH_CHECK();
    uint32_t h = 0;
    MFX_TRY(read_word(over.get(), &h, st));
    *fits = h == 0;
    return MFX_OK;
}

}  // namespace mfx

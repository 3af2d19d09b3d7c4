// ccd_scatter.hip -- CCD++ passes for HYPER-SPARSE orientations (config 5's shard: 1.25 M x 1 M with
// 125 M ratings, ~100 entries per row / column, density 1e-4).  Measured numbers and their history: DESIGN.md section 4.2.
//
// Why another kernel.  The flat-stream kernel (ccd_kernels.hip) reduces per segment and gathers the
// other operand; its gather is served from LDS only if the gathered dimension is cut into LDS-sized
// panels, and at this density a (panel, segment) pair holds 0.6 entries: the segments are shredded.
// Round 1 therefore fell back to L2-sized "cache panels" with per-lane global gathers, which the
// texture path serves at ~85 G gathers/s -- 0.2 of the HBM roofline (profiles/r02_ubench_scatter.txt).
//
// What this kernel does instead: SWAP THE ROLES.  A pass that needs the sums over COLUMNS streams the
// ROW-major copy (and vice versa), stored panel-major over the columns:
//   * the panel's columns are local (16-bit index): their operands AND their (g, h) accumulators sit in
//     LDS (24 B per column, up to 6816 columns per workgroup: all of a CU's 160 KB);
//   * inside a panel the entries keep the copy's row-major order, so the row operand is an ASCENDING,
//     nearly sequential global read (a wave's 256 entries span ~400 rows = ~25 cache lines instead of
//     256 random ones); the row id of an entry is one byte -- its step from the previous entry of the tile's
//     sorted order -- plus one base per tile (11 B per non-zero streamed; explicit 32-bit ids, 14 B, where a step overflows);
//   * the reduction is a scatter-add into the LDS accumulators.  It is made ORDER-INDEPENDENT, hence
//     bitwise reproducible, by accumulating the fp32 contributions in 64-bit fixed point (ds_add_u64,
//     scale 2^36: exact for |sum| < 1.3e8, resolution 1.5e-11; an fp32 sum of n such terms is off by
//     ~6e-8 * |sum| * sqrt(n), the fixed-point one by ~1e-11 * sqrt(n)).  LDS fp32 atomics would be both
//     non-deterministic and 4x slower (profiles/r02_ubench_ldsatomic.txt).
// A workgroup flushes its accumulators to a slab when it leaves a panel; k_scatter_combine adds the slabs of a
// panel (integers: any order gives the same bits) and converts to the dense fp32 (g, h) that the ordinary
// finalize / all-reduce path takes.
//
// (r3) PERSISTENT WORKGROUPS.  The grid is one workgroup per CU (the LDS allows no more); workgroup w owns a
// contiguous chunk range of the panel-major stream and keeps a panel's slice and accumulators in LDS across all of
// its chunks of that panel -- it reloads / flushes only where the panel changes inside its range (~256 + npanels slabs
// per pass instead of one per 57 k-entry chunk), and a wave's tile pipeline runs on across chunk boundaries.
//
// (r4) WHAT BOUNDED THE PASS, AND THE FIX.  Round 3 read its counters as "the CU's read path from L2, one 128-byte request
// per ~10.8 clocks".  The same counters say more: TCC_MISS equalled ALL reads -- every line of the streamed operand, which
// each panel re-reads in full (npanels x 10-12 MB per pass), missed the L2 and came through the fabric, and streams plus
// operand together moved ~6.2-6.7 TB/s of L2-miss traffic: the pass was bound by the memory side, on bytes it did not
// need to fetch from there.  The operand missed because the 256 persistent workgroups were at 256 different PHASES of
// their panels (range length = 0.57 / 0.72 panel), so at any moment the chip read 256 places spread over the whole
// operand.  With EQUAL panels and a panel count that is a multiple of (workgroups / 8) -- the layout rule in
// CcdSolver::build_stores -- the phase of workgroup w depends on w mod 8 only, which is also its XCD: every XCD's
// workgroups walk the operand in step, one window of it is live per L2, and the re-reads are L2 hits (TCC_MISS 23.1 M ->
// 12.6 M per row-sum pass = the streams alone).  Config-5 shard: column-sum pass 358 -> 307 us, row-sum pass 448 -> 354 us
// (profiles/r04_exp_scatter_alignment.txt).  What is left, from knock-out builds in the same file: the streams alone
// (gathers served from L1) take 236-268 us = 5.2-5.8 TB/s of the 11 B per non-zero; the operand now costs its bytes at
// ~18 TB/s, the guide's L2-served rate, ON TOP (+48 us for 1.6 GB of pairs, +129 us for 2.3 GB of 12-byte triples; a
// float4 operand: +167 us) -- the L2 -> L1 fills and the HBM streams share the CU's one vector-memory path and add.
// Neither the LDS atomics (knocked out: -2 %), nor the fixed-point conversion (-1 %), nor latency (streams three tiles
// ahead, gathers one: +-0) bound it.
//
// (r4) PANEL GROUPS.  A store can be launched group by group (consecutive panels each; SegStreamDev::scat_ngroups), each
// launch aligned on its own, so that in a sharded solve group j's sums are combined, all-reduced and finalized on a
// second stream while group j + 1 is streamed (CcdSolver::rank_fused_scatter).  The sums are integers: any grouping
// gives the same bits (test_scatter_overlap_groups_are_bit_identical).
//
// The per-element arithmetic is the reference's (unfused multiply, subtract, multiply, add) as in the flat
// kernel, so both residual copies keep holding bit-identical values whichever kernel updates them.
#include "ccd_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace mfx {
namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kScatBlock = 1024;  // 16 wavefronts, one workgroup per CU (the LDS is the limit)

__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// fp32 -> 64-bit fixed point (two's complement), scale 2^36: every SUM must stay below 2^27 (kFixedLimit; enforced per
// term against kFixedLimit / (entries of the fullest local index), see `bad` and ScatterArgs::term_limit), resolution 2^-36 (terms are truncated toward zero at that resolution; an fp32 with |x| >= 2^-12
// converts exactly).
// (r2: a 3-instruction conversion -- fma onto 1.5 * 2^52, integer in the low mantissa bits -- instead of this
// 9-instruction f64 -> i64 sequence changed nothing: 339 vs 345 us on the Netflix shape, 531 vs 510 us on the
// shard.  Neither the VALU nor the LDS atomics bound the pass: tools/ubench_ldsatomic.hip measures 0.49 clk per
// element per CU for two random ds_add_u64, a quarter of the pass's time per element.)
__device__ __forceinline__ unsigned long long to_fixed(float x) {
    return (unsigned long long) (long long) ((double) x * 68719476736.0);
}
constexpr float kFixedLimit = 134217728.f;  // 2^27

struct ScatterArgs {
    const uint16_t* lidx;     // [padded] local index inside the panel (pad: panel_rows)
    const uint32_t* segid;    // [padded] id of the streamed dimension (row of a row-major copy), ascending inside a panel (IDS32)
    const uint8_t* seg_delta; // [padded] ... or its step from the previous entry of the tile's sorted order
    const uint32_t* tile_base;// [padded / 256] ... and the id of the tile's first sorted entry
    float* val;               // [padded] residual copy
    const uint32_t* wg_panel; // [workgroups]
    uint32_t tiles_per_span, panel_rows, local_len;
    const void* slice_src;    // operands of the local dimension, [local_len]
    const void* global_op;    // operands of the streamed dimension, indexed by segid
    unsigned long long* wgacc;  // [slabs][2 * panel_rows]
    const uint32_t* chunk_lo; // [workgroups + 1] chunk range of every persistent workgroup
    const uint32_t* slab0;    // [workgroups] first slab a workgroup writes (one per panel it visits, ascending)
    uint32_t* slab_bad;       // [slabs] != 0: some term of the slab was NaN / Inf / beyond the fixed-point range
    float term_limit;         // a term must stay below this: kFixedLimit / (most entries any local index has), so that no SUM can wrap
    int add;
};

struct F3 { float x, y, z; };  // (prev_new, cur_old, cur_new) of the streamed dimension, 12 bytes apart
template <int MODE> struct ScatTraits;
// the two fused passes: slice (prev_new, cur_old) of the local dimension; streamed operand float2 (V) or a 12-byte triple (U)
template <> struct ScatTraits<SM_V>     { using S = float2; using G = float2; static constexpr bool kSlice = true,  kAcc = true,  kWrite = true; };
template <> struct ScatTraits<SM_U>     { using S = float2; using G = F3;     static constexpr bool kSlice = true,  kAcc = true,  kWrite = true; };
template <> struct ScatTraits<SM_SWEEP> { using S = float;  using G = float;  static constexpr bool kSlice = false, kAcc = true,  kWrite = false; };
template <> struct ScatTraits<SM_RESID> { using S = float;  using G = float;  static constexpr bool kSlice = true,  kAcc = false, kWrite = true; };

__host__ __device__ constexpr size_t align16(size_t x) { return (x + 15) / 16 * 16; }
template <int MODE>
__host__ __device__ size_t scat_slice_bytes(uint32_t pr) { return ScatTraits<MODE>::kSlice ? align16(((size_t) pr + 1) * sizeof(typename ScatTraits<MODE>::S)) : 0; }
template <int MODE>
__host__ __device__ size_t scat_lds_bytes(uint32_t pr) { return scat_slice_bytes<MODE>(pr) + (ScatTraits<MODE>::kAcc ? ((size_t) pr + 1) * 16 : 0); }

// Inclusive prefix sums over the 64 lanes of two 16-bit fields at once (every field total stays below 2^16), on
// the DPP path; lanes without a source read 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t src) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) src, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_prefix_2x16(uint32_t x) {
    x += dpp_u32<0x111>(x);        // row_shr:1
    x += dpp_u32<0x112>(x);        // row_shr:2
    x += dpp_u32<0x114>(x);        // row_shr:4
    x += dpp_u32<0x118>(x);        // row_shr:8 -> prefix inside each row of 16
    x += dpp_u32<0x142, 0xA>(x);   // row_bcast15: rows 1, 3 += total of rows 0, 2
    x += dpp_u32<0x143, 0xC>(x);   // row_bcast31: rows 2, 3 += total of rows 0-1
    return x;
}

template <int MODE, bool IDS32>
__global__ __launch_bounds__(kScatBlock) void k_scatter(ScatterArgs a) {
    using TR = ScatTraits<MODE>;
    using S = typename TR::S;
    using G = typename TR::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    S* __restrict__ slice = reinterpret_cast<S*>(lds_raw);
    unsigned long long* __restrict__ acc = reinterpret_cast<unsigned long long*>(lds_raw + scat_slice_bytes<MODE>(a.panel_rows));
    __shared__ uint32_t bad_any;
    const uint32_t pr = a.panel_rows;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nt = a.tiles_per_span;
    const G* __restrict__ gop = static_cast<const G*>(a.global_op);
    // Inside every 256-entry tile the builder stores the row-sorted entries TRANSPOSED: lane l's four elements are
    // sorted entries l, 64 + l, 128 + l, 192 + l of the tile.  The streams keep their 16-byte-per-lane loads, and
    // gather instruction e covers 64 CONSECUTIVE sorted entries -- ~100 rows, 7 cache lines -- so the four gathers
    // of a tile touch disjoint quarters of its row window, each line exactly once.  (With the natural order every
    // one of the four instructions touched all ~26 lines of the window and relied on the 32 KB L1 to hold 16 waves'
    // windows between them: it does not, and the pass ran at 0.55-0.65 ms instead of 0.4.)
    // Schedule: two statically named register sets (no register is rotated through a move: a move of a register
    // that is still in flight is a wait); per tile the four gathers go out back to back, then the stream loads of
    // the next tile, then the tile is computed.  Straight-line body, no branch (a branch merges two wait counts
    // into the smaller one); the loads past the run's end are clamped re-reads.  The sched_barriers pin the issue
    // order -- left to itself the compiler sinks every gather to its use behind the LDS atomics of the previous
    // element and waits vmcnt(0) four times per tile.
    // Segment ids: IDS32 streams them (4 B per entry); otherwise a tile carries one byte per entry -- the step from
    // the previous entry of its sorted order -- and one base id: lane l's dword holds the steps of sorted entries
    // l, 64 + l, 128 + l, 192 + l, so two packed 16-bit prefix scans over the lanes plus the group totals give the
    // four ids (a tile spans at most 255 * 256 ids).  ~30 VALU instructions per tile for 3 B per entry less.
    struct Tile { u16x4 l; u32x4 s; uint32_t d, base; f32x4 v; uint32_t tile; };
    // tile index (in units of 256 stored entries) of this wave's q-th tile of a run of chunks that starts at chunk
    // `c0`: wave i owns span i of every chunk, i.e. tiles ((c0 + q / nt) * 16 + i) * nt + q % nt
    auto stream = [&](uint32_t tile) {
        Tile x;
        x.tile = tile;
        const uint64_t e0 = (uint64_t) tile * 256;
        // issue order pinned: ids first, values last.  The wait before a tile's gathers then never has to
        // cover the value load -- nor, on the loop's back edge, the previous tile's store behind it.
        if constexpr (IDS32) {
            x.s = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.segid + e0) + lane);
        } else {
            x.d = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.seg_delta + e0) + lane);
            x.base = a.tile_base[tile];
        }
        x.l = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(a.lidx + e0) + lane);
        __builtin_amdgcn_sched_barrier(0);
        x.v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.val + e0) + lane);
        return x;
    };
    auto ids = [&](const Tile& x) {
        if constexpr (IDS32) {
            return x.s;
        } else {
            const uint32_t p01 = wave_prefix_2x16((x.d & 0xFFu) | ((x.d & 0xFF00u) << 8));
            const uint32_t p23 = wave_prefix_2x16(((x.d >> 16) & 0xFFu) | ((x.d >> 24) << 16));
            const uint32_t t01 = (uint32_t) __builtin_amdgcn_readlane((int) p01, 63);
            const uint32_t t23 = (uint32_t) __builtin_amdgcn_readlane((int) p23, 63);
            const uint32_t b1 = x.base + (t01 & 0xFFFFu), b2 = b1 + (t01 >> 16), b3 = b2 + (t23 & 0xFFFFu);
            u32x4 r;
            r[0] = x.base + (p01 & 0xFFFFu);
            r[1] = b1 + (p01 >> 16);
            r[2] = b2 + (p23 & 0xFFFFu);
            r[3] = b3 + (p23 >> 16);
            return r;
        }
    };
    struct Gath { G g[4]; };
    auto gather = [&](const Tile& x) {
        Gath r;
        const u32x4 id = ids(x);
#pragma unroll
        for (int e = 0; e < 4; ++e) r.g[e] = gop[id[e]];  // ascending inside a panel: a few cache lines per wave
        return r;
    };
    bool bad = false;  // a term that the fixed-point accumulators cannot hold (NaN, Inf, |x| >= 2^27)
    auto compute = [&](const Tile& x, const Gath& gp) {
        // three phases, so that the four slice reads go out together and the eight atomics follow without an LDS read
        // between them (element by element every slice read also waited for the previous element's atomics: lgkmcnt(0))
        f32x4 o;
        float gc[4], hc[4];
        S sp[4];
        if constexpr (TR::kSlice) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[e] = slice[x.l[e]];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (MODE == SM_V) {
                o[e] = add_rn(sub_rn(x.v[e], mul_rn(gp.g[e].x, sp[e].x)), mul_rn(gp.g[e].y, sp[e].y));
                gc[e] = gp.g[e].y * o[e];
                hc[e] = gp.g[e].y * gp.g[e].y;
            } else if constexpr (MODE == SM_U) {
                o[e] = add_rn(sub_rn(x.v[e], mul_rn(gp.g[e].x, sp[e].x)), mul_rn(gp.g[e].y, sp[e].y));
                gc[e] = gp.g[e].z * o[e];
                hc[e] = gp.g[e].z * gp.g[e].z;
            } else if constexpr (MODE == SM_SWEEP) {
                o[e] = x.v[e];
                gc[e] = gp.g[e] * x.v[e];
                hc[e] = gp.g[e] * gp.g[e];
            } else {
                const float prod = mul_rn(sp[e], gp.g[e]);
                o[e] = a.add ? add_rn(x.v[e], prod) : sub_rn(x.v[e], prod);
                gc[e] = 0.f; hc[e] = 0.f;
            }
        }
        if constexpr (TR::kWrite) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(a.val + (uint64_t) x.tile * 256) + lane);
        if constexpr (TR::kAcc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t l = x.l[e];
                bad |= !(__builtin_fabsf(gc[e]) < a.term_limit) | !(hc[e] < a.term_limit);  // (NaN compares false)
                atomicAdd(&acc[2 * l], to_fixed(gc[e]));
                atomicAdd(&acc[2 * l + 1], to_fixed(hc[e]));
            }
        }
    };

    const uint32_t c_end = a.chunk_lo[blockIdx.x + 1];
    uint32_t chunk = a.chunk_lo[blockIdx.x];
    uint32_t slab = a.slab0[blockIdx.x];
    if (threadIdx.x == 0) bad_any = 0;
    while (chunk < c_end) {
        // ---- a run: the chunks of ONE panel inside this workgroup's range
        const uint32_t panel = a.wg_panel[chunk];
        uint32_t run_end = chunk + 1;
        while (run_end < c_end && a.wg_panel[run_end] == panel) ++run_end;
        const uint32_t gbase = panel * pr;
        const uint32_t cnt = a.local_len - gbase < pr ? a.local_len - gbase : pr;
        if constexpr (TR::kSlice) {
            const S* __restrict__ src = static_cast<const S*>(a.slice_src);
            // slots cnt .. pr (the tail of a short last panel and the padding slot) hold zeros: padding entries
            // then keep their stored 0 and contribute exact zeros
            for (uint32_t i = threadIdx.x; i <= pr; i += kScatBlock) slice[i] = i < cnt ? src[gbase + i] : S{};
        }
        if constexpr (TR::kAcc)
            for (uint32_t i = threadIdx.x; i < 2 * (pr + 1); i += kScatBlock) acc[i] = 0ull;
        __syncthreads();

        // this wave's tiles of the run, in order: span `wave` of chunk, chunk + 1, ...; a cursor on the scalar unit
        // (tile index in units of 256 stored entries).  Past the end it stays on the last tile: a clamped re-read
        // that is never computed.
        const uint32_t Q = (run_end - chunk) * nt;  // even: nt is
        uint32_t cur = (chunk * (kScatBlock / 64) + wave) * nt, t_in = 0, left = Q;
        auto next_tile = [&]() {
            const uint32_t r = cur;
            if (left > 1) {
                --left; ++cur;
                if (++t_in == nt) { t_in = 0; cur += (kScatBlock / 64 - 1) * nt; }
            }
            return r;
        };
        // (r4) Software pipeline: the streams run THREE tiles ahead of the tile being computed, its gathers ONE tile ahead.
        // With the workgroups phase-aligned (whole panels each, see SegStreamStore::build_device) the streamed operand is
        // served by L2, and what bounded the pass next was the memory in flight per CU: one workgroup of 16 waves with one
        // tile of streams (1.8 KB read) outstanding each cannot cover the loaded HBM latency (29 KB per CU in flight ->
        // 4.9 TB/s of streams with the gathers knocked out); and every tile exposed the L2 round trip of its own gathers
        // (issued, then waited for).  Four statically named stream sets and two gather sets, rotated by unrolling.
        Tile S0 = stream(next_tile()), S1 = stream(next_tile()), S2 = stream(next_tile()), S3;
        Gath G0 = gather(S0), G1;
        uint32_t q = 0;
        for (; q + 4 <= Q; q += 4) {
            G1 = gather(S1); S3 = stream(next_tile());
            __builtin_amdgcn_sched_barrier(0);
            compute(S0, G0);
            __builtin_amdgcn_sched_barrier(0);
            G0 = gather(S2); S0 = stream(next_tile());
            __builtin_amdgcn_sched_barrier(0);
            compute(S1, G1);
            __builtin_amdgcn_sched_barrier(0);
            G1 = gather(S3); S1 = stream(next_tile());
            __builtin_amdgcn_sched_barrier(0);
            compute(S2, G0);
            __builtin_amdgcn_sched_barrier(0);
            G0 = gather(S0); S2 = stream(next_tile());
            __builtin_amdgcn_sched_barrier(0);
            compute(S3, G1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < Q) {  // Q is even: two tiles left, S0 (gathered) and S1
            G1 = gather(S1);
            __builtin_amdgcn_sched_barrier(0);
            compute(S0, G0);
            compute(S1, G1);
        }
        if constexpr (TR::kAcc) {
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) bad_any = 1;
            bad = false;
        }
        __syncthreads();  // every wave has left the panel: its sums are complete, its slice is dead
        if constexpr (TR::kAcc) {
            unsigned long long* __restrict__ dst = a.wgacc + (size_t) slab * 2 * pr;
            for (uint32_t i = threadIdx.x; i < 2 * pr; i += kScatBlock) dst[i] = acc[i];
            if (threadIdx.x == 0) { a.slab_bad[slab] = bad_any; bad_any = 0; }
            ++slab;
            // (the same thread zeroes exactly the words it just copied, at the top of the next run: no barrier needed
            // between the two; the one above orders the slice reload behind every wave's last read)
        }
        chunk = run_end;
    }
}

// gh[c] = g, gh[G + c] = h of local-dimension index c: the slabs of its panel added as integers.  A slab that met a
// term its fixed-point accumulators cannot hold (NaN, Inf, |x| >= 2^27: a diverging solve) poisons the panel's sums
// with NaN -- the flat path and the reference propagate non-finite values the same way -- instead of delivering
// finite but wrong numbers.
__global__ __launch_bounds__(256) void k_scatter_combine(uint32_t c_lo, uint32_t c_hi, uint32_t pr, const uint32_t* __restrict__ slab_lo,
                                                         const unsigned long long* __restrict__ wgacc,
                                                         const uint32_t* __restrict__ slab_bad, float* __restrict__ gh) {
    // local indices [c_lo, c_hi) = one panel group; its block of gh starts at 2 c_lo: g of the group, then h
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, c = c_lo + i, len = c_hi - c_lo;
    if (c >= c_hi) return;
    const uint32_t p = c / pr, l = c - p * pr;
    unsigned long long g = 0, h = 0;
    uint32_t bad = 0;
    for (uint32_t w = slab_lo[p]; w < slab_lo[p + 1]; ++w) {
        const unsigned long long* s = wgacc + (size_t) w * 2 * pr + 2 * l;
        g += s[0];
        h += s[1];
        bad |= slab_bad[w];
    }
    constexpr double inv = 1.0 / 68719476736.0;
    float* out = gh + 2 * (size_t) c_lo;
    out[i] = bad ? __builtin_nanf("") : (float) ((double) (long long) g * inv);
    out[len + i] = bad ? __builtin_nanf("") : (float) ((double) (long long) h * inv);
}

template <int MODE, bool IDS32>
int launch_scatter_t(const SegStreamDev& s, const ScatterArgs& a, uint32_t nwg, hipStream_t st) {
    const size_t lds = scat_lds_bytes<MODE>(s.panel_rows);
    MFX_REQUIRE(lds + 64 <= 160 * 1024, "scatter layout: %u local entries do not fit LDS", s.panel_rows);  // (+ the static bad_any word)
    if (lds > 48 * 1024) {  // a per-device attribute of the kernel: set once per (instantiation, device)
        static std::mutex m;
        static size_t set_bytes[64] = {};
        int dev = 0;
        MFX_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(m);
        if (dev < 0 || dev >= 64 || lds > set_bytes[dev]) {
            MFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<MODE, IDS32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            if (dev >= 0 && dev < 64) set_bytes[dev] = lds;
        }
    }
    hipLaunchKernelGGL((k_scatter<MODE, IDS32>), dim3(nwg), dim3(kScatBlock), lds, st, a);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

}  // namespace

int launch_scatter(ScatterMode mode, const SegStreamDev& s, const void* slice_src, const void* global_op, int add, hipStream_t st, int group) {
    MFX_REQUIRE(s.scatter && s.spans_per_wg == kScatBlock / 64 && s.tiles_per_span % 2 == 0, "launch_scatter: not a scatter layout");
    MFX_REQUIRE(s.scat_nwg > 0 && s.scat_chunk_lo && s.scat_slab0 && s.slab_lo && s.scat_slab_bad, "launch_scatter: the layout carries no workgroup ranges");
    MFX_REQUIRE(s.segid || (s.seg_delta && s.tile_base), "launch_scatter: the layout carries no segment ids");
    MFX_REQUIRE(s.scat_ngroups >= 1 && s.scat_ngroups <= SegStreamDev::kMaxScatterGroups && group < (int) s.scat_ngroups, "launch_scatter: bad panel group %d of %u", group, s.scat_ngroups);
    const bool ids32 = s.seg_delta == nullptr;
    for (uint32_t g = group < 0 ? 0u : (uint32_t) group; g < (group < 0 ? s.scat_ngroups : (uint32_t) group + 1); ++g) {
        const uint32_t nwg = s.scat_grp_nwg[g];
        if (nwg == 0) continue;  // (a group without chunks: fewer panels than groups)
        ScatterArgs a;
        a.lidx = s.idx16; a.segid = s.segid; a.seg_delta = s.seg_delta; a.tile_base = s.tile_base; a.val = s.val; a.wg_panel = s.wg_panel; a.tiles_per_span = s.tiles_per_span;
        a.panel_rows = s.panel_rows; a.local_len = s.gather_len; a.slice_src = slice_src; a.global_op = global_op; a.wgacc = s.wgacc;
        a.chunk_lo = s.scat_chunk_lo + s.scat_grp_tab[g]; a.slab0 = s.scat_slab0 + s.scat_grp_wg0[g]; a.slab_bad = s.scat_slab_bad;
        // (r4) the accumulators hold |x| < 2^27; a per-term bound alone lets a SUM of in-range terms wrap to a finite
        // wrong value.  With at most scat_max_local_cnt entries per local index, terms below 2^27 / that count cannot.
        a.term_limit = kFixedLimit / (float) (s.scat_max_local_cnt ? s.scat_max_local_cnt : 1u);
        a.add = add;
        int rc;
        switch (mode) {
            case SM_V: rc = ids32 ? launch_scatter_t<SM_V, true>(s, a, nwg, st) : launch_scatter_t<SM_V, false>(s, a, nwg, st); break;
            case SM_U: rc = ids32 ? launch_scatter_t<SM_U, true>(s, a, nwg, st) : launch_scatter_t<SM_U, false>(s, a, nwg, st); break;
            case SM_SWEEP: rc = ids32 ? launch_scatter_t<SM_SWEEP, true>(s, a, nwg, st) : launch_scatter_t<SM_SWEEP, false>(s, a, nwg, st); break;
            case SM_RESID: rc = ids32 ? launch_scatter_t<SM_RESID, true>(s, a, nwg, st) : launch_scatter_t<SM_RESID, false>(s, a, nwg, st); break;
            default: return fail(MFX_ERR_INVALID, "launch_scatter: bad mode %d", (int) mode);
        }
        MFX_TRY(rc);
    }
    return MFX_OK;
}

int launch_scatter_combine(const SegStreamDev& s, float* gh, hipStream_t st, int group) {
    if (s.gather_len == 0) return MFX_OK;
    MFX_REQUIRE(s.scat_ngroups >= 1 && s.scat_ngroups <= SegStreamDev::kMaxScatterGroups && group < (int) s.scat_ngroups, "launch_scatter_combine: bad panel group %d of %u", group, s.scat_ngroups);
    for (uint32_t g = group < 0 ? 0u : (uint32_t) group; g < (group < 0 ? s.scat_ngroups : (uint32_t) group + 1); ++g) {
        const uint32_t lo = s.scat_grp_lo[g], hi = s.scat_grp_lo[g + 1];
        if (hi <= lo) continue;
        hipLaunchKernelGGL(k_scatter_combine, dim3((hi - lo + 255) / 256), dim3(256), 0, st, lo, hi, s.panel_rows, s.slab_lo, s.wgacc, s.scat_slab_bad, gh);
        MFX_HIP(hipGetLastError());
    }
    return MFX_OK;
}

}  // namespace mfx

// tile_emulator.cpp -- CPU check of the 2-D tile order (tile_layout.hpp): builds layouts for random
// patterns, checks the storage invariants the GPU kernel relies on (bijection, codes, runs inside one
// sub-tile, tiles sorted by segment) and replays the kernel's accumulation (per workgroup = block x
// strip, tiles in panel order, one read-modify-write per run) against exact integer sums.
// Built and run by tests/test_layout_cpu.py (no GPU needed).
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "tile_layout.hpp"

using namespace mfx;

static int check(uint32_t nseg, uint32_t G, const std::vector<uint32_t>& lens, uint32_t QB, uint32_t SR, uint32_t R,
                 bool sorted_idx, std::mt19937& rng, bool* built) {
    std::vector<uint32_t> ptr(nseg + 1, 0);
    for (uint32_t c = 0; c < nseg; ++c) ptr[c + 1] = ptr[c] + lens[c];
    const uint64_t nnz = ptr[nseg];
    std::vector<uint32_t> idx(nnz);
    for (auto& x : idx) x = rng() % G;
    if (sorted_idx) for (uint32_t c = 0; c < nseg; ++c) std::sort(idx.begin() + ptr[c], idx.begin() + ptr[c + 1]);
    std::vector<long long> val(nnz), vec(G);
    for (auto& x : val) x = (long long) (rng() % 7) - 3;
    for (auto& x : vec) x = (long long) (rng() % 5) - 2;
    TileLayoutHost L;
    *built = build_tile_layout(ptr.data(), idx.data(), nseg, nnz, G, QB, SR, 1.0, &L);
    if (!*built) {  // legal only if some (segment, panel) run is longer than a sub-tile
        for (uint32_t c = 0; c < nseg; ++c) {
            std::vector<uint32_t> cnt((G + SR - 1) / SR, 0);
            for (uint32_t q = ptr[c]; q < ptr[c + 1]; ++q) if (++cnt[idx[q] / SR] > kSubTile) return 0;
        }
        printf("builder refused a pattern without long runs\n");
        return 1;
    }
    if (L.padded % kSubTile || L.code.size() != L.padded || L.perm.size() != L.padded) { printf("sizes\n"); return 1; }
    if (L.tile_sub.size() != (size_t) L.nB * L.nP + 1) { printf("tile_sub size\n"); return 1; }
    std::vector<char> seen(nnz, 0);
    std::vector<long long> part((size_t) R * nseg, 0);
    std::vector<int> writes((size_t) R * nseg, 0);
    for (uint32_t b = 0; b < L.nB; ++b) {
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t p_lo = (uint32_t) ((uint64_t) r * L.nP / R), p_hi = (uint32_t) ((uint64_t) (r + 1) * L.nP / R);
            for (uint32_t p = p_lo; p < p_hi; ++p) {
                const size_t t = (size_t) b * L.nP + p;
                if (L.tile_sub[t + 1] < L.tile_sub[t]) { printf("tile_sub not monotone\n"); return 1; }
                int last_seg = -1;
                for (uint32_t j = L.tile_sub[t]; j < L.tile_sub[t + 1]; ++j) {
                    int run_seg = -1; long long acc = 0;
                    bool saw_pad = false;
                    for (uint32_t e = 0; e < kSubTile; ++e) {
                        const uint64_t at = (uint64_t) j * kSubTile + e;
                        const uint32_t code = L.code[at], q = L.perm[at];
                        const uint32_t sl = code >> 16, gl = code & 0xFFFF;
                        if (q == ~0u) { if (code != L.pad_code()) { printf("pad code\n"); return 1; } saw_pad = true; continue; }
                        if (saw_pad) { printf("padding in the middle of a sub-tile\n"); return 1; }
                        if (q >= nnz || seen[q]) { printf("perm not a bijection\n"); return 1; }
                        seen[q] = 1;
                        const uint32_t s = b * L.QB + sl;
                        if (sl >= L.QB || s >= nseg || q < ptr[s] || q >= ptr[s + 1]) { printf("segment code\n"); return 1; }
                        if (gl >= L.SR || p * L.SR + gl != idx[q]) { printf("index code\n"); return 1; }
                        if ((int) s < last_seg) { printf("tile not sorted by segment\n"); return 1; }
                        if ((int) s != run_seg) {
                            if (run_seg >= 0) { part[(size_t) r * nseg + run_seg] += acc; ++writes[(size_t) r * nseg + run_seg]; }
                            if ((int) s == last_seg && run_seg != (int) s && e == 0 && last_seg >= 0) {
                                // the same segment continuing in a new sub-tile = a run crossing the boundary
                                printf("run crosses a sub-tile boundary\n"); return 1;
                            }
                            run_seg = (int) s; acc = 0;
                        }
                        last_seg = (int) s;
                        acc += vec[idx[q]] * val[q];
                    }
                    if (run_seg >= 0) { part[(size_t) r * nseg + run_seg] += acc; ++writes[(size_t) r * nseg + run_seg]; }
                }
            }
        }
    }
    for (uint64_t q = 0; q < nnz; ++q) if (!seen[q]) { printf("missing element\n"); return 1; }
    for (uint32_t c = 0; c < nseg; ++c) {
        long long ref = 0, got = 0;
        for (uint32_t q = ptr[c]; q < ptr[c + 1]; ++q) ref += vec[idx[q]] * val[q];
        for (uint32_t r = 0; r < R; ++r) got += part[(size_t) r * nseg + c];
        if (got != ref || L.seg_cnt[c] != lens[c]) { printf("segment %u: got %lld want %lld\n", c, got, ref); return 1; }
    }
    return 0;
}

int main() {
    std::mt19937 rng(777);
    int cases = 0, refused = 0;
    for (int trial = 0; trial < 40; ++trial) {
        const uint32_t nseg = 1 + rng() % 3000, G = 1 + rng() % 5000;
        std::vector<uint32_t> lens(nseg);
        for (auto& l : lens) {
            const uint32_t k = rng() % 10;
            l = k < 3 ? 0 : k < 8 ? rng() % 6 : k < 9 ? rng() % 60 : rng() % 700;
        }
        if (trial % 9 == 0) for (auto& l : lens) l = 0;
        for (uint32_t QB : {1u, 7u, 256u, 4608u}) {
            for (uint32_t SR : {1u, 50u, 1000u, 4608u}) {
                if ((uint64_t) ((nseg + QB - 1) / QB) * ((G + SR - 1) / SR) > 400000) continue;
                for (uint32_t R : {1u, 3u}) {
                    bool built = false;
                    if (check(nseg, G, lens, QB, SR, R, (trial & 1) != 0, rng, &built)) {
                        printf("FAILED trial %d QB %u SR %u R %u\n", trial, QB, SR, R);
                        return 1;
                    }
                    ++cases;
                    refused += built ? 0 : 1;
                }
            }
        }
    }
    printf("tile emulator: %d cases ok (%d refused for long runs)\n", cases, refused);
    return 0;
}

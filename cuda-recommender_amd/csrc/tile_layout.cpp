// tile_layout.cpp -- host-side builder of the 2-D tile order (see tile_layout.hpp).
#include "tile_layout.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

namespace mfx {
namespace {

// fn(b) for every block, blocks handed out dynamically to up to 16 host threads (plain
// std::thread: libmfx must not drag a second OpenMP runtime into a process that hosts torch's).
template <typename F>
void for_each_block(uint32_t nB, F fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = std::max(1u, std::min({16u, hw ? hw : 1u, nB}));
    if (nt == 1) { for (uint32_t b = 0; b < nB; ++b) fn(b); return; }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&] { for (uint32_t b; (b = next.fetch_add(1)) < nB;) fn(b); });
    for (auto& x : th) x.join();
}

// Walks block b in stored order and calls place(panel, first_slot, q_list, count) for every run.
// cur[p] = slot cursor of tile (b, p), relative to the tile's start.  Returns false when a run does
// not fit a sub-tile.
struct RunWalker {
    const uint32_t* ptr;
    const uint32_t* idx;
    uint32_t SR, nP;
    std::vector<uint32_t> cur;        // [nP]
    std::vector<uint32_t> order;      // scratch: positions of one segment sorted by panel (stable)

    template <typename Place>
    bool walk(uint32_t s_lo, uint32_t s_hi, Place place) {
        cur.assign(nP, 0);
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const uint32_t lo = ptr[s], hi = ptr[s + 1];
            if (lo == hi) continue;
            // positions of this segment grouped by panel, input order inside a panel; the usual
            // input (indices ascending) is already grouped
            bool grouped = true;
            for (uint32_t q = lo + 1; q < hi && grouped; ++q) grouped = idx[q] / SR >= idx[q - 1] / SR;
            const uint32_t n = hi - lo;
            order.resize(n);
            for (uint32_t i = 0; i < n; ++i) order[i] = lo + i;
            if (!grouped)
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return idx[x] / SR < idx[y] / SR; });
            for (uint32_t i = 0; i < n;) {
                const uint32_t p = idx[order[i]] / SR;
                uint32_t j = i + 1;
                while (j < n && idx[order[j]] / SR == p) ++j;
                const uint32_t c = j - i;
                if (c > kSubTile) return false;
                uint32_t at = cur[p];
                if (at % kSubTile + c > kSubTile) at = (at / kSubTile + 1) * kSubTile;  // keep the run inside one sub-tile
                place(p, at, &order[i], c, s - s_lo);
                cur[p] = at + c;
                i = j;
            }
        }
        return true;
    }
};

}  // namespace

bool build_tile_layout(const uint32_t* ptr, const uint32_t* idx, uint32_t nseg, uint64_t nnz, uint32_t G,
                       uint32_t QB, uint32_t SR, double max_pad_frac, TileLayoutHost* out) {
    TileLayoutHost& L = *out;
    L = TileLayoutHost();
    if (QB == 0 || SR == 0 || QB > 0xFFFFu || SR > 0xFFFFu || nseg == 0 || G == 0) return false;
    L.nseg = nseg; L.gather_len = G; L.QB = QB; L.SR = SR; L.nnz = nnz;
    L.nB = (nseg + QB - 1) / QB;
    L.nP = (G + SR - 1) / SR;
    const uint32_t nB = L.nB, nP = L.nP;
    if ((uint64_t) nB * nP > (1ull << 28)) return false;
    L.seg_cnt.resize(nseg);
    for (uint32_t s = 0; s < nseg; ++s) L.seg_cnt[s] = ptr[s + 1] - ptr[s];

    // pass A: sub-tiles per tile
    std::vector<uint32_t> subs((size_t) nB * nP, 0);
    std::atomic<bool> ok{true};
    for_each_block(nB, [&](uint32_t b) {
        if (!ok.load(std::memory_order_relaxed)) return;
        RunWalker w{ptr, idx, SR, nP, {}, {}};
        const uint32_t s_lo = b * QB, s_hi = std::min(nseg, s_lo + QB);
        if (!w.walk(s_lo, s_hi, [](uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t) {})) { ok = false; return; }
        for (uint32_t p = 0; p < nP; ++p) subs[(size_t) b * nP + p] = (w.cur[p] + kSubTile - 1) / kSubTile;
    });
    if (!ok) return false;
    L.tile_sub.resize((size_t) nB * nP + 1);
    uint64_t run = 0;
    for (size_t t = 0; t < (size_t) nB * nP; ++t) { L.tile_sub[t] = (uint32_t) run; run += subs[t]; }
    if (run * kSubTile >= 0xFFFFFF00ull) return false;
    L.tile_sub[(size_t) nB * nP] = (uint32_t) run;
    L.padded = std::max<uint64_t>(run, 1) * kSubTile;  // nnz == 0: one all-padding sub-tile keeps the buffers non-empty
    if (nnz > 0 && (double) (L.padded - nnz) > max_pad_frac * (double) L.padded) return false;

    // pass B: placement
    L.code.assign(L.padded, L.pad_code());
    L.perm.assign(L.padded, ~0u);
    for_each_block(nB, [&](uint32_t b) {
        RunWalker w{ptr, idx, SR, nP, {}, {}};
        const uint32_t s_lo = b * QB, s_hi = std::min(nseg, s_lo + QB);
        w.walk(s_lo, s_hi, [&](uint32_t p, uint32_t at, const uint32_t* qs, uint32_t c, uint32_t s_local) {
            const uint64_t base = (uint64_t) L.tile_sub[(size_t) b * nP + p] * kSubTile + at;
            for (uint32_t i = 0; i < c; ++i) {
                const uint32_t q = qs[i];
                L.code[base + i] = (s_local << 16) | (idx[q] - p * SR);
                L.perm[base + i] = q;
            }
        });
    });
    return true;
}

}  // namespace mfx

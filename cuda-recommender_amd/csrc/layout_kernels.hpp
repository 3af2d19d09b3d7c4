// layout_kernels.hpp -- device pipeline that builds the panel-major flat layout of flat_layout.hpp
// from an input-order CSR / CSC orientation that already sits in HBM.
//
// The host builder (flat_layout.cpp) remains the reference implementation and the fallback; this
// pipeline produces bit-identical arrays for every pattern it accepts, namely the GROUPED ones: inside
// every segment the entries of one panel are consecutive in the input (true whenever the gathered
// indices ascend inside a segment, which is what every CSR / CSC converter emits).  A virtual segment
// (panel, segment) is then one run [first_q, first_q + cnt) of input positions and the whole build is
// streaming work:
//
//   runs        one pass over idx: run starts / ends -> first_q[v] (atomicMin), last_q[v] (atomicMax);
//               range check of every gathered index in the same pass
//   counts      cnt[v] = last_q - first_q + 1; the pattern is grouped  <=>  sum(cnt) == nnz
//   scan        exclusive prefix of cnt in panel-major order (+ per-panel padding to whole chunks)
//   heads       one bit per non-empty virtual segment, hpre = exclusive prefix of the words' popcounts
//   ranks       rank_code / seg_of_rank / v_of_rank from (hpre, flags)
//   place       one pass over the STORED positions: each looks its source position up through its rank
//               (no search), writes the 16- or 32-bit index and the value, or the padding
//
// ~30 B/nnz of traffic in all: a few milliseconds for 1e8 non-zeros, against ~0.1 s for the host
// builder plus its transfers.
#pragma once

#include "common.hpp"

namespace mfx {

struct LayoutBuildIn {
    uint32_t nseg = 0;
    uint64_t nnz = 0;
    uint32_t G = 0;            // gathered dimension: every idx must be < G
    uint32_t npanels = 1;      // P
    uint32_t panel_rows = 0;   // PR (0: plain layout, one panel)
    bool local_idx = false;    // LDS panels: stored index = idx - panel * PR
    bool idx16 = false;        // ... stored as uint16
    uint32_t pad_index = 0;
    bool transpose_tiles = false;  // scatter layout: stored position tile + 4l + e holds entry tile + 64e + l of the panel-major order
    uint32_t span_len = 0;     // tiles_per_span * 256
    uint64_t chunk = 0;        // span_len * spans_per_wg: panels are padded to whole chunks
    const uint32_t* ptr = nullptr;  // device, [nseg + 1]
    const uint32_t* idx = nullptr;  // device, [nnz]
    const float* val = nullptr;     // device, [nnz] or nullptr (zeros)
};

// ptr[0] == 0, ptr[nseg] == nnz, monotone.  MFX_ERR_INVALID with the offending segment otherwise.
int lk_check_ptr(const LayoutBuildIn& in, hipStream_t st);

// first_q / cnt [P * nseg] (device, caller-allocated).  *grouped = false when some (panel, segment)
// pair is visited more than once (the caller then falls back to the host builder).  Also range-checks
// every index (MFX_ERR_INVALID naming the first offender).  Synchronises `st`.
int lk_runs_and_counts(const LayoutBuildIn& in, uint32_t* first_q, uint32_t* cnt, bool* grouped, hipStream_t st);

// out[i] = sum_{j < i} f(in[j]) for i in [0, n]  (n + 1 outputs); f = identity or popcount.  `scratch`
// must hold scan_scratch_words(n) words.  out may alias in when f is the identity... it may not: keep
// them distinct.
size_t scan_scratch_words(size_t n);
int lk_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, bool popcount, uint32_t* scratch, hipStream_t st);

// ptr_v[v] = S[v] + delta[v / nseg] for v < nv, ptr_v[nv] = padded.  (delta: device, [P])
int lk_ptr_v(const uint32_t* S, const uint32_t* delta, uint32_t nseg, size_t nv, uint32_t padded, uint32_t* ptr_v, hipStream_t st);
// gather S[p * nseg] for p in [0, P] into out[P + 1] (device)
int lk_panel_starts(const uint32_t* S, uint32_t nseg, uint32_t P, uint32_t* out, hipStream_t st);

// flags32 (zeroed by the caller): bit ptr_v[v] for every v with ptr_v[v + 1] > ptr_v[v]
int lk_heads(const uint32_t* ptr_v, size_t nv, uint32_t* flags32, hipStream_t st);

// rank_code[v] (0xFFFFFFFF = empty, else rank | bit 31 when the virtual segment runs into later spans),
// seg_of_rank[rank] = v % nseg, v_of_rank[rank] = v
int lk_ranks(const uint32_t* ptr_v, size_t nv, uint32_t nseg, uint32_t span_len, const uint32_t* flags32,
             const uint32_t* hpre, uint32_t* rank_code, uint32_t* seg_of_rank, uint32_t* v_of_rank, hipStream_t st);

// stored arrays: idx_out is uint16[padded] (in.idx16) or uint32[padded]; val_out float[padded];
// seg_out (may be nullptr) uint32[padded]: the segment of every stored element, 0 for padding
int lk_place(const LayoutBuildIn& in, uint64_t padded, const uint32_t* ptr_v, const uint32_t* first_q, const uint32_t* cnt,
             const uint32_t* flags32, const uint32_t* hpre, const uint32_t* v_of_rank, void* idx_out, float* val_out,
             uint32_t* seg_out, hipStream_t st);

// Scatter layout: seg (the uint32 ids lk_place wrote, transposed tiles) -> delta[padded] (one byte per element: the
// step from the previous entry of the tile's sorted order) + tile_base[padded / 256] (id of the tile's first
// entry).  *fits = false when some step exceeds 255 (delta / tile_base are then unusable).  Synchronises `st`.
int lk_delta_encode(const uint32_t* seg, uint64_t padded, uint8_t* delta, uint32_t* tile_base, bool* fits, hipStream_t st);

// max over workgroup chunks of the ranks a chunk touches (incl. the one open at its start) -> *out (device word, zeroed by the caller)
int lk_max_wg_ranks(const uint32_t* hpre, size_t nwords, size_t chunk_words, uint32_t* out, hipStream_t st);

}  // namespace mfx

// flat_layout.hpp -- static metadata that lets one kernel stream a CSR/CSC orientation as a flat,
// nnz-balanced array while still reducing per segment (row or column), with the gathered factor
// entries served from LDS.
//
// The sparsity pattern never changes during a solve, so everything that depends only on the
// pattern is computed once, on the host, at solver creation:
//
//   panel           the gathered index space [0, G) is cut into panels of `panel_rows` entries so
//                   that one panel's slice of the operand pack fits in LDS.  Non-zeros are stored
//                   PANEL-MAJOR: all entries whose gathered index falls in panel 0 (ordered by
//                   segment, then in the input's order), then panel 1, ...  Stored indices are
//                   panel-local.  npanels == 1 and panel_rows == 0 is the plain layout (global
//                   gather, input order).
//   cache panel     (lds == false, panel_rows != 0) the same panel-major order, but with panels
//                   sized for the L2 instead of LDS: indices stay global (32-bit), the kernel
//                   gathers from global memory and only ever touches one ~2 MB slice of the
//                   operand pack at a time.  For matrices whose segments are too short to be cut
//                   into LDS-sized panels (hyper-sparse shards: every (panel, segment) pair would
//                   hold < 1 entry).  Padding carries index G and gathers an exact zero.
//   virtual segment (panel p, segment c) -> v = p * nseg + c; ptr_v[v] .. ptr_v[v+1] in the padded
//                   panel-major coordinates.  Every panel is padded to a whole number of
//                   workgroup chunks; the padding is folded into the panel's last virtual segment
//                   and points at a zero slot (LDS index panel_rows), so it contributes nothing.
//   span            L = 256 * tiles_per_span consecutive stored non-zeros, owned by one wavefront
//                   (64 lanes x 4 consecutive non-zeros per tile, 16-byte loads); a workgroup
//                   chunk is spans_per_wg consecutive spans of ONE panel.
//   head flag       1 bit per stored non-zero: "first entry of its virtual segment", kept as one
//                   32-bit word per 32 stored non-zeros (flags32)
//   rank            index of a virtual segment among the NON-EMPTY ones.  hpre[w] = number of heads
//                   before word w, so element -> rank is hpre[w] + popcount(flags32[w] & below) - 1:
//                   two small per-lane loads per tile (0.25 B/nnz) instead of a per-nnz segment id.
//
// Reduction contract (ccd_kernels.hip): the span that contains a virtual segment's head stores that
// segment's partial sum to part[rank] (exactly one writer, plain store); a span whose first element
// is NOT a head stores the sum of its leading run to carry[span].  finalize adds, for segment c,
// over panels p in order: part[rank(p,c)] + carries of spans s with
// ptr_v[v]/L < s <= (ptr_v[v+1]-1)/L -- deterministic, no atomics.
#pragma once

#include <algorithm>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace mfx {

constexpr uint32_t kTileElems = 256;  // 64 lanes x 4 elements

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big stored
// arrays (hundreds of MB) are written exactly once, by the threads that place the non-zeros, instead of
// being zero-filled -- and page-faulted in -- by one thread first.
template <typename T>
struct NoInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAllocator<U>; };
    template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using HostVec = std::vector<T, NoInitAllocator<T>>;

// A few host worker threads that can neither leak nor take the process down: a thread that cannot be
// started (std::system_error: thread limit of the container) makes the job run inline instead; the
// destructor joins whatever was started (a joinable std::thread's destructor is std::terminate); an
// exception thrown by a job (std::bad_alloc) is carried to wait(), which rethrows it on the caller's
// thread, where the C ABI turns it into an error code.
class ThreadGang {
public:
    ThreadGang() = default;
    ThreadGang(const ThreadGang&) = delete;
    ThreadGang& operator=(const ThreadGang&) = delete;
    ~ThreadGang() { join_all(); }
    template <typename F>
    void run(F job) {
        auto guarded = [this, job]() mutable {
            try { job(); } catch (...) { std::lock_guard<std::mutex> lk(m_); if (!err_) err_ = std::current_exception(); }
        };
        try {
            th_.emplace_back(guarded);
        } catch (const std::system_error&) {
            guarded();
        }
    }
    void wait() {
        join_all();
        if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }
    static unsigned width() {
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(16u, hw ? hw : 1u));
    }

private:
    void join_all() {
        for (std::thread& t : th_) if (t.joinable()) t.join();
        th_.clear();
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::exception_ptr err_;
};

struct FlatLayoutHost {
    uint32_t nseg = 0;            // real segments (columns for CSC, rows for CSR)
    uint32_t gather_len = 0;      // G: length of the gathered index space
    uint32_t npanels = 1;
    uint32_t panel_rows = 0;      // 0: plain layout (no LDS staging)
    bool lds = true;              // panel_rows != 0: LDS panels (panel-local indices) or cache panels (global indices)
    uint32_t spans_per_wg = 1;
    uint32_t nne = 0;             // non-empty virtual segments
    uint32_t nspans = 0;
    uint32_t tiles_per_span = 0;
    uint64_t nnz = 0;             // real non-zeros
    uint64_t padded_nnz = 0;      // stored elements (nspans * span length)
    uint32_t max_wg_ranks = 0;    // most ranks any workgroup chunk touches (incl. the one open at its start)
    std::vector<uint32_t> ptr_v;           // [npanels*nseg + 1]
    std::vector<uint32_t> seg_cnt;         // [nseg] real entries per segment
    std::vector<uint32_t> flags32;         // [padded_nnz / 32 + 16] head bits (tail words zero)
    std::vector<uint32_t> hpre;            // [padded_nnz / 32 + 16] heads before each word (tail = nne)
    std::vector<int32_t> rank_of_seg;      // [npanels*nseg], -1 for an empty virtual segment
    std::vector<uint32_t> seg_of_rank;     // [nne] REAL segment id of each rank
    std::vector<uint32_t> wg_panel;        // [nspans / spans_per_wg] (panel layout only)
    HostVec<uint32_t> idx_local;           // [padded_nnz] panel-local gathered index (pad: zero slot); empty if idx16 was asked for
    HostVec<uint16_t> idx16;               // [padded_nnz] the same as 16-bit values (FlatLayoutOptions::emit_idx16)
    HostVec<uint32_t> perm;                // [padded_nnz] input position of each stored element, ~0u for pad
    HostVec<float> val_st;                 // [padded_nnz] values in stored order, 0 for pad (FlatLayoutOptions::emit_val)
    // FlatLayoutOptions::compact_perm and every segment's entries grouped by panel in input order (true
    // for ascending indices): `perm` stays empty; virtual segment v holds input positions first_q[v],
    // first_q[v] + 1, ... over its real entries [ptr_v[v], min(ptr_v[v+1], panel_real_end[panel]))
    bool perm_is_runs = false;
    HostVec<uint32_t> first_q;             // [npanels*nseg] (only meaningful for non-empty virtual segments)
    std::vector<uint32_t> panel_real_end;  // [npanels] end of each panel's real entries (before its padding)
    uint32_t span_len() const { return tiles_per_span * kTileElems; }
    uint32_t pad_index() const { return panel_rows ? (lds ? panel_rows : gather_len) : 0u; }
};

struct FlatLayoutOptions {
    uint32_t tiles_per_span = 0;  // 0: choose
    uint32_t panel_rows = 0;      // 0: plain layout
    bool lds = true;              // false: cache panels (global indices, gather from L2)
    uint32_t spans_per_wg = 1;    // waves per workgroup in the LDS-panel kernel
    // Optional outputs produced in the same pass that places the non-zeros (saves two more passes over
    // hundreds of MB): 16-bit indices instead of idx_local (LDS panels only), values in stored order.
    bool emit_idx16 = false;
    bool emit_val = false;
    bool compact_perm = false;    // see FlatLayoutHost::perm_is_runs
    // scatter layout (ccd_scatter.hip): LDS panels + an explicit per-element segment id, no flags / ranks /
    // partials; built by the device pipeline only
    bool scatter = false;
    bool scatter_ids32 = false;   // keep the 4-byte ids even when one-byte steps would do (A/B, tests)
    uint32_t scatter_groups = 1;  // panel groups the pass can be launched by (ccd_kernels.hpp, SegStreamDev::scat_ngroups)
    uint32_t scatter_wgs = 0;     // persistent workgroups per launch; 0 = one per CU
    const float* val = nullptr;   // input-order values for emit_val; nullptr = zeros
};

// ptr/idx are the input orientation (host pointers); G is the gathered dimension.
// Runs fn(begin, end) over [0, n) on up to 16 host threads (plain std::thread: libmfx must not drag a
// second OpenMP runtime into a process that already hosts torch's).
void parallel_ranges_u64(uint64_t n, void (*fn)(uint64_t, uint64_t, void*), void* ctx);

// span length (in 256-element tiles) chosen for `nnz` stored non-zeros; `panels`: LDS-panel layout
uint32_t pick_tiles_per_span(uint64_t nnz, bool panels);

void build_flat_layout(const uint32_t* ptr, const uint32_t* idx, uint32_t nseg, uint64_t nnz, uint32_t G,
                       const FlatLayoutOptions& opt, FlatLayoutHost* out);

}  // namespace mfx

// ccd_solver.hpp -- resident CCD++ solver: device state + the outer/rank/inner loop of
// ccdpp_NV (cuda_src/CCD_CUDA.cu:224-451) re-designed for MI355X: everything stream-ordered,
// no per-launch host sync, fused passes by default (see DESIGN.md "CCD++ schedule").
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "ccd_kernels.hpp"
#include "comm.hpp"
#include "flat_layout.hpp"

namespace mfx {

// Owns one orientation's device arrays; `view` is what the kernels see.
class SegStreamStore {
public:
    // ptr/idx/val live in `space` (input order).  val may be nullptr (zeros).  G = length of the
    // gathered index space.  opt.panel_rows != 0 stores the non-zeros panel-major (flat_layout.hpp).
    // build_mode = mfx_params.layout_build: 0 device pipeline when the pattern is grouped, host builder
    // otherwise; 1 host builder; 2 device pipeline or MFX_ERR_INVALID.
    int build(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx, const float* val,
              mfx_memspace space, const FlatLayoutOptions& opt, int build_mode, hipStream_t st);
    bool built_on_device() const { return built_on_device_; }
    bool can_fuse_finalize() const { return view.fz_order != nullptr; }
    SegStreamDev view;
    const FlatLayoutHost& layout() const { return layout_; }
    // out[input position] = stored value, for every real entry (test / debug path: mfx_ccd_get_residual)
    int unpermute(float* out, hipStream_t st);

private:
    int build_host(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx, const float* val,
                   mfx_memspace space, const FlatLayoutOptions& opt, hipStream_t st);
    // *done = false (and MFX_OK) when the pattern is not grouped: nothing was built
    int build_device(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx, const float* val,
                     mfx_memspace space, const FlatLayoutOptions& opt, hipStream_t st, bool* done);
    // group tables of the fused finalize (LDS panels with 16-span workgroups; ccd_kernels.hip, fused_finalize)
    int build_fuse_tables(hipStream_t st);
    int build_owner_lists(hipStream_t st);  // (r4) segment-owner fused passes of small matrices (plain layout)
    DevBuf<uint32_t> own_long_, own_short_;
    bool built_on_device_ = false;
    FlatLayoutHost layout_;
    DevBuf<uint32_t> fz_order_, fz_g0_, fz_g1_, fz_expected_, fz_arrived_, fz_orphans_;
    DevBuf<uint32_t> ptr_, ptr_v_, seg_cnt_, idx_, seg_of_rank_, flags32_, hpre_, wg_panel_, perm_;
    DevBuf<float> val_;
    DevBuf<float2> part_, carry_;
    DevBuf<uint32_t> rank_code_;
    DevBuf<uint16_t> idx16_;
    DevBuf<uint32_t> segid_, slab_lo_, tile_base_, scat_chunk_lo_, scat_slab0_, scat_slab_bad_;  // scatter layout
    DevBuf<uint8_t> seg_delta_;
    DevBuf<unsigned long long> wgacc_;
    uint32_t ngroups_ = 1, grp_nwg_[SegStreamDev::kMaxScatterGroups] = {}, grp_tab_[SegStreamDev::kMaxScatterGroups] = {},
             grp_wg0_[SegStreamDev::kMaxScatterGroups] = {}, grp_lo_[SegStreamDev::kMaxScatterGroups + 1] = {};  // scatter panel groups
    // run-compressed provenance (perm_is_runs): kept on the host, uploaded on the first unpermute()
    std::vector<uint32_t> first_q_host_, panel_end_host_;
    DevBuf<uint32_t> first_q_dev_, panel_end_dev_;
};

// Panel size for an orientation whose largest LDS-staged pack element is `elem_bytes` wide:
// 0 (plain layout) when panels are off or would shred the segments; see ccd_solver.hip.
FlatLayoutOptions choose_layout(const mfx_params& p, uint32_t nseg, uint64_t nnz, uint32_t G, uint32_t elem_bytes,
                                bool need_plain);

// Optional per-launch HIP-event bracketing (mfx_params.profile).  Events are recorded on the
// solver's own stream -- the stream the kernels run on.
class KernelProfiler {
public:
    ~KernelProfiler();
    void enable(bool on) { on_ = on; }
    bool enabled() const { return on_; }
    int begin(int name_id, hipStream_t st);
    int end(hipStream_t st);
    int collect();  // after a stream sync: fold elapsed times into the per-name totals
    void reset_totals();
    static const char* name(int id);
    enum { K_FCSC = 0, K_FCSR, K_SWEEP, K_RESID, K_FINALIZE, K_COMBINE, K_PACK, K_RMSE, K_ALLREDUCE,
           K_SWEEP_WAVE, K_RESID_WAVE, K_SCAT_V, K_SCAT_U, K_SCAT_SWEEP, K_SCAT_RESID, K_SCAT_COMBINE, K_SWEEP_REF, K_HOST_ENQUEUE, K_COUNT };
    double seconds[K_COUNT] = {};
    int64_t launches[K_COUNT] = {};

private:
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs_;
    std::vector<hipEvent_t> pool_;
    size_t used_ = 0;
    bool on_ = false;
    int take(hipEvent_t* e);
};

class CcdSolver {
public:
    static int create(CcdSolver** out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                      mfx_memspace space, const mfx_shard* shard);
    ~CcdSolver();
    int set_factors(const float* W, const float* H, mfx_memspace space);
    int iterate(int n_outer, int with_rmse, mfx_iter_report* reports);
    int get_factors(float* W, float* H, mfx_memspace space);
    int get_residual(float* csc_val, float* csr_val);
    KernelProfiler& profiler() { return prof_; }
    int set_profile(bool on);
    // per-rank trace of the last iterate() call (rank_trace): [outer iterations][k] RMSE of calrmse_r1 and seconds
    // (NaN / 0 for ranks the eps rule skipped), ranks updated per outer iteration
    int rank_trace(int cap, double* rmse, double* seconds, int iters_cap, int32_t* ranks_done) const;
    void layout_info(int side, int32_t out[4]) const {
        const SegStreamDev& v = side == 0 ? csc_.view : csr_.view;
        out[0] = (int32_t) v.npanels; out[1] = (int32_t) v.panel_rows; out[2] = v.scatter ? (v.seg_delta ? 2 : 3) : v.lds_panels ? 1 : 0; out[3] = (int32_t) v.tiles_per_span;
    }

private:
    CcdSolver() = default;
    int init(const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space,
             const mfx_shard* shard);
    // ---- opt-in extensions: the flags the reference parses and ignores (mfx_params.do_nmf / eps / rank_trace) ----
    bool ext_on_ = false;
    FinalizeArgs fin_base() const;     // lambda + the extension fields every finalize of this solver carries
    int inner_stop(uint32_t t, int it, bool* stop);   // eps: LIBPMF's rule after one inner iteration
    int trace_begin(uint32_t t);
    int trace_end(uint32_t t);
    DevBuf<double> fundec_seg_, fundec_sum_, r1_sum_;
    DevBuf<float> old_w_, old_h_, test_resid_;
    bool test_resid_valid_ = false;
    int64_t cur_oiter_ = 0;
    double fundec_max_ = 0.0;
    int early_stop_ = 0;
    hipEvent_t ev_rank_[2] = {nullptr, nullptr};
    std::vector<double> trace_rmse_, trace_secs_;
    std::vector<int32_t> trace_done_;
    // finalize inside the fused passes ("last workgroup to arrive finalizes", bit-identical to the separate kernel): OPT-IN,
    // MFX_FUSE_FINALIZE=1 / 2 -- it measured slower than a kernel boundary on LDS panels (r2, ccd_kernels.hip) AND, generalised
    // to the plain layout of small matrices in r4 (one graph node per pass instead of two), there as well: ML-1M shape k = 40
    // under hipGraph replay 0.786 ms per outer iteration with 160 nodes of ~4.9 us, 1.115 ms with 80 fused nodes of ~13.9 us
    // (drain of the write-through partials, one returning agent-scope atomic, then the completing workgroup's chain of
    // dependent past-the-cache loads: ~9 us of tail against a ~5 us node boundary; profiles/r04_exp_small.txt).
    int fuse_finalize_ = 0;
    bool use_fused(const SegStreamStore& s) const { return fuse_finalize_ == 1 && !ext_on_ && s.can_fuse_finalize(); }
    int rank_fused(uint32_t t);
    int rank_fused_owner(uint32_t t);  // (r4) small matrices: k_seg_owner, two launches per rank
    bool owner_mode_ = false;
    int rank_as_written(uint32_t t, bool add_back);
    int flush_pending();
    int sweep(SegStreamStore& s, const float* vec, float* out, bool is_col_side);
    int resid(SegStreamStore& s, const float* gathered, const float* per_seg, int add);
    // scatter mode (hyper-sparse: both stores in the scatter layout, roles swapped -- see ccd_scatter.hip):
    // sums over COLUMNS come from the row-major store csr_, sums over ROWS from csc_
    bool scatter_ = false;
    // kernel_variant = -1: sweeps in the reference's summation order (ccd_reforder.hip), bit-identical to src/CCD.cpp
    bool ref_order_ = false;
    DevBuf<uint32_t> ref_order_csc_, ref_order_csr_;  // segments, longest first
    uint32_t ref_nlong_csc_ = 0, ref_nlong_csr_ = 0;  // ... of which this many are long (>= 32768 entries: ccd_reforder.hip)
    // (r4) the mode on the default path's schedule: subtraction of rank t - 1, add-back of rank t, first sweep and division in one
    // reference-order owner pass per copy (launch_ref_owner), the long segments' workgroups on a second stream.  MFX_REF_FUSED=0: the
    // as-written sequence (separate residual passes, k_sweep_ref / k_sweep_ref2), kept for A/B.
    bool ref_fused_ = false;
    RefStreams ref_streams_;
    DevBuf<float> ref_zero_cols_;      // [n] zeros: the add-back operand of the first outer iteration (the reference skips that add-back)
    int rank_ref_fused(uint32_t t, bool add_back);
    // slabs -> dense (g,h) -> [all-reduce] -> finalize, for one panel group of the streamed store (-1: all), on `st` (nullptr: st_)
    int scatter_finalize(bool cols, const FinalizeArgs& base, int group = -1, hipStream_t st = nullptr);
    // (r4) overlap of the column-side exchange with the column pass (sharded solve, scatter layout): see init()
    uint32_t overlap_groups_ = 1, comm_reserve_cus_ = 0;
    hipStream_t st2_ = nullptr;
    hipEvent_t ev_grp_[SegStreamDev::kMaxScatterGroups] = {}, ev_join_ = nullptr;
    int rank_fused_scatter(uint32_t t);
    int finalize_cols(const FinalizeArgs& base);  // CSC side: all-reduce across shards if sharded
    int test_rmse(double* rmse_out);
    float* Wt(uint32_t t) { return W_.get() + (size_t) t * m_; }
    float* Ht(uint32_t t) { return H_.get() + (size_t) t * n_; }

    int device_ = 0;
    hipStream_t st_ = nullptr;
    mfx_params p_{};
    uint32_t m_ = 0, n_ = 0, k_ = 0;
    uint64_t nnz_ = 0;
    SegStreamStore csc_, csr_;
    DevBuf<float> W_, H_;
    DevBuf<float2> packA_, packB_;  // [m], [n]
    DevBuf<float4> packC_;          // [n]
    DevBuf<float> gh_cols_, gh_rows_;  // dense (g,h): [2n], [2m]
    // multi-GPU
    mfx_comm_s* comm_ = nullptr;
    DevBuf<uint32_t> global_col_nnz_;
    int64_t global_test_nnz_ = 0;
    // test set
    int64_t nnz_test_ = 0;
    DevBuf<uint32_t> t_row_, t_col_;
    DevBuf<float> t_val_;
    DevBuf<double> rmse_partials_, rmse_sum_;
    // state
    int64_t oiter_ = 0;        // outer iterations completed
    double rank_acc_ = 0, update_acc_ = 0;  // running totals for the reference's log line
    int32_t pending_sub_ = -1; // fused schedule: rank whose new (u,v) is not yet subtracted
    bool factors_set_ = false;
    bool comm_warm_ = false;   // the data-path all-reduce has run once (RCCL's lazy connection setup)
    KernelProfiler prof_;
    hipEvent_t ev_[6] = {};
    hipGraph_t graph_ = nullptr;          // one outer iteration of the fused schedule (k ranks x 4 launches)
    hipGraphExec_t graph_exec_ = nullptr;
    bool graph_failed_ = false;
    int enqueue_outer_iteration(int64_t oiter);
    int build_stores(const mfx_csx* R, const mfx_params* p, mfx_memspace space, bool scatter);
};

}  // namespace mfx

struct mfx_ccd_s {
    mfx::CcdSolver* impl;
};

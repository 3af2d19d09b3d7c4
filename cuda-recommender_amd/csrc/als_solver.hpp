// als_solver.hpp -- resident ALS solver (replaces als_NV, cuda_src/ALS_CUDA.cu:200-406).
//
// Per half-sweep and per segment (user row / item column): A = sum x x^T + lambda I on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32, exact fp32), b = sum r x, Cholesky A = L L^T in LDS and two
// triangular solves.  No device-side malloc, no per-thread k^2 scratch (the reference's
// updateW_overH_kernel does both, ALS_CUDA.cu:93-94).
#pragma once

#include <vector>

#include "comm.hpp"
#include "common.hpp"

namespace mfx {

struct AlsItem {      // one wavefront's work: entries [lo, hi) of segment `seg`
    uint32_t seg, lo, hi;
    int32_t slot;     // < 0: the whole segment, solve in place; >= 0: partial Gramian slot
};
struct AlsReduce {    // a segment whose Gramian was split over `nslots` consecutive slots
    uint32_t seg, slot0, nslots;
};

// One orientation (rows over H, or columns over W).
struct AlsHalf {
    uint32_t nseg = 0;
    uint64_t nnz = 0;
    DevBuf<uint32_t> ptr, idx;
    DevBuf<float> val;
    DevBuf<AlsItem> items;
    DevBuf<AlsReduce> reduces;
    uint32_t nitems = 0, nreduces = 0, nslots = 0;
    // G = number of factor rows the indices address: every idx must be < G (checked on the device)
    int build(uint32_t nseg, uint64_t nnz, uint32_t G, const uint32_t* ptr, const uint32_t* idx, const float* val,
              mfx_memspace space, uint32_t chunk, hipStream_t st);
};

class AlsSolver {
public:
    static int create(AlsSolver** out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                      mfx_memspace space, const mfx_als_shard* shard = nullptr);
    ~AlsSolver();
    int set_factors(const float* W, const float* H, mfx_memspace space);
    int iterate(int n_iter, int with_rmse, mfx_iter_report* reports);
    int get_factors(float* W, float* H, mfx_memspace space);
    int kernel_times(int cap, const char** names, double* seconds, int64_t* launches);

private:
    AlsSolver() = default;
    int init(const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space, const mfx_als_shard* shard);
    int exchange(float* X, const std::vector<int64_t>& bounds);  // every rank broadcasts its block of X (one grouped call)
    int meet_shards();  // first iterate() of a sharded solve: gathers everyone's block boundaries (never in create)
    int device_ = 0;
    hipStream_t st_ = nullptr;
    mfx_params p_{};
    uint32_t m_ = 0, n_ = 0, k_ = 0;
    // sharded solve: this rank's row / column block and everyone's block boundaries
    mfx_comm_s* comm_ = nullptr;
    uint32_t row_lo_ = 0, col_lo_ = 0, row_hi_ = 0, col_hi_ = 0;
    std::vector<int64_t> row_bounds_, col_bounds_;
    bool shards_met_ = false;  // both boundary vectors gathered AND validated (meet_shards)
    int64_t global_test_nnz_ = 0;
    AlsHalf rows_, cols_;
    DevBuf<float> W_, H_, ws_;
    DevBuf<uint32_t> spd_fail_;
    DevBuf<unsigned long long> phases_;  // MFX_ALS_PHASES=1: per-phase clocks of the half-sweep kernels (diagnostic)
    int print_phases(const char* what);
    int64_t nnz_test_ = 0;
    DevBuf<uint32_t> t_row_, t_col_;
    DevBuf<float> t_val_;
    DevBuf<double> rmse_partials_, rmse_sum_;
    int64_t iter_ = 0;
    double update_acc_ = 0;
    bool factors_set_ = false;
    hipEvent_t ev_[6] = {};
    double t_half_[2] = {0, 0};
    int64_t n_half_[2] = {0, 0};
};

// Launches one half-sweep: Y[seg] = argmin over segment `seg` given factor rows X[x_rows + 1][k],
// whose last row must be all zeros.
int als_half_launch(const AlsHalf& h, const float* X, uint32_t x_rows, float* Y, uint32_t k, float lambda, float* ws,
                    uint32_t* spd_fail, hipStream_t st, unsigned long long* phases = nullptr);
// floats of workspace needed for `nslots` partial slots at rank k
size_t als_ws_floats(uint32_t nslots, uint32_t k);

// The half-sweep "as written" (als_exact.hip): the reference's arithmetic in its order, bit for bit.
int als_half_exact_launch(const AlsHalf& h, const float* X, float* Y, uint32_t k, float lambda, uint32_t* spd_fail, hipStream_t st);
// inverseMatrix_CholeskyMethod on one k x k matrix (host pointers), same arithmetic
int als_inverse_op(int64_t k, const float* A, float* Ainv, int device);

int als_gramian_op(int64_t cnt, const uint32_t* idx, int64_t nrows_x, const float* X, int64_t k, float* A,
                   int device);
int als_half_op(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx, const float* val,
                int64_t nrows_x, const float* X, float* Y, int64_t k, float lambda, int variant, int device);

}  // namespace mfx

struct mfx_als_s {
    mfx::AlsSolver* impl;
};

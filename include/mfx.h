/*
 * mfx.h -- C ABI of libmfx.so: MI355X-native CCD++ / ALS matrix factorization.
 *
 * This is the drop-in boundary for the GPU path of Zialus/CUDA-Recommender: the two
 * solver entry points the reference's driver calls (src/main.cpp:11-17 -> runCUDA ->
 * kernel_wrapper_ccdpp_NV / kernel_wrapper_als_NV, cuda_src/CCD_CUDA.h:49,
 * cuda_src/ALS_CUDA.h:40), restated over plain pointers and sizes so that any host
 * language can bind them.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - All functions return 0 on success and a negative mfx_status on failure;
 *     mfx_last_error() returns a thread-local human readable message.  Nothing here
 *     calls exit()/abort() or resets the device (the reference's cudaDeviceReset(),
 *     cuda_src/CCD_CUDA.cu:167,177, is deliberately NOT reproduced).
 *   - Caller owns every buffer passed in; the library owns device memory behind handles.
 *   - Indices are 0-based uint32, values fp32 (reference: DTYPE float, src/pmf_util.h:26).
 *   - Factor layouts are the reference's:
 *       CCD++ : W flat [k][rows]  (W[t*rows+i]),  H flat [k][cols]   (cuda_src/CCD_CUDA.cu:255-261)
 *       ALS   : W flat [rows][k]  (W[i*k+c]),     H flat [cols][k]   (cuda_src/ALS_CUDA.cu:229-243)
 *   - There is no CPU fallback: every compute entry point fails with MFX_ERR_NO_DEVICE
 *     when no HIP device is usable.
 */
#ifndef MFX_H
#define MFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision: bumped whenever a struct layout or an entry point's argument list changes (2: mfx_params grew the
 * opt-in extension fields and mfx_als_half a `variant` argument; kernel_variant -1).  A binding must compare it
 * with mfx_version() before its first call -- mfx/_lib.py does. */
#define MFX_VERSION 2

typedef enum mfx_status {
    MFX_OK = 0,
    MFX_ERR_INVALID = -1,   /* bad argument / inconsistent sizes */
    MFX_ERR_NO_DEVICE = -2, /* no usable HIP device (never silently falls back to the CPU) */
    MFX_ERR_HIP = -3,       /* a HIP runtime call failed ("CCD FAILED: %s", CCD_CUDA.cu:174) */
    MFX_ERR_COMM = -4,      /* RCCL missing or a collective failed */
    MFX_ERR_ALLOC = -5
} mfx_status;

/* Where the arrays of an mfx_csx / mfx_coo / factor argument live. */
typedef enum mfx_memspace { MFX_HOST = 0, MFX_DEVICE = 1 } mfx_memspace;

/* Dual CSR+CSC rating matrix == the six raw-pointer getters of the reference's
 * SparseMatrix (src/pmf_util.h:83-105).  Both orientations must describe the same
 * matrix; within a row/column the order of entries is the summation order. */
typedef struct mfx_csx {
    int64_t rows, cols, nnz;
    const uint32_t* csc_col_ptr; /* [cols+1] */
    const uint32_t* csc_row_idx; /* [nnz]    */
    const float* csc_val;        /* [nnz]    */
    const uint32_t* csr_row_ptr; /* [rows+1] */
    const uint32_t* csr_col_idx; /* [nnz]    */
    const float* csr_val;        /* [nnz]    */
} mfx_csx;

/* COO test set == reference TestData getters (src/pmf_util.h:196-206).  nnz may be 0. */
typedef struct mfx_coo {
    int64_t nnz;
    const uint32_t* row;
    const uint32_t* col;
    const float* val;
} mfx_coo;

/* The fields of the reference's `parameter` (src/pmf.h:8-43) that the GPU path reads
 * (cuda_src/CCD_CUDA.cu:225-231), plus the knobs this implementation adds. */
typedef struct mfx_params {
    uint32_t k;                /* rank                         (-k, default 10)  */
    float lambda;              /* regularisation               (-l, default 0.1) */
    int32_t maxiter;           /* outer iterations             (-t, default 5)   */
    int32_t maxinneriter;      /* CCD++ inner iterations T     (-T, default 1)   */
    uint32_t nBlocks;          /* accepted and ignored: kernels pick their own geometry */
    uint32_t nThreadsPerBlock; /* accepted and ignored                                   */
    int32_t verbose;           /* 1: print the reference's "[-INFO-] iteration num" line */
    int32_t device;            /* HIP device ordinal (reference hard-codes 0)            */
    int32_t schedule;          /* CCD++ kernel schedule: 0 = as written (separate add-back,
                                  sweeps, subtract launches, one per reference kernel),
                                  1 = fused passes (default; same arithmetic, fewer bytes).
                                  ALS: 0 = as written (explicit Cholesky inverse in the reference's operation
                                  order, bit-identical to src/ALS.cpp), 1 = MFMA Gramian + Cholesky solve */
    int32_t kernel_variant;    /* -1 = REFERENCE-ORDER parity mode (schedule 0 only, single GPU): every rank-one sum is added strictly
                                  left to right in unfused fp32 exactly like src/CCD.cpp:6-16 -- W, H and both residual copies come out
                                  bit-identical to the reference's CPU solver (csrc/ccd_reforder.hip; the CCD++ counterpart of ALS
                                  schedule 0; the subtraction of a rank and the add-back of the next are applied in one pass, same roundings); ~8x slower than the default path;
                                  0 = wave-per-segment kernels (schedule 0 only), 1 = flat-stream kernels (default),
                                  2 = force the scatter layout (csrc/ccd_scatter.hip), which hyper-sparse shapes
                                  get on their own: < 8 entries per (LDS panel, row / column) pair;
                                  3 = the same with explicit 32-bit segment ids in the stream (what a layout
                                  falls back to when some one-byte step between consecutive ids overflows) */
    int32_t profile;           /* 1: bracket every launch with HIP events (mfx_*_kernel_times) */
    int32_t tiles_per_span;    /* flat-stream span length / 256; 0 = choose from nnz */
    int32_t panel_rows;        /* panels of the gathered index space. 0 = choose (LDS panels, 64 KB of LDS per
                                  workgroup, when segments stay long enough; else 2 MB cache panels served by
                                  L2; else none), -1 = off (plain layout, gather from L2 / Infinity Cache),
                                  > 0 = explicit LDS panel of that many entries, < -1 = explicit cache
                                  panel of -panel_rows entries */
    int32_t wg_waves;          /* wavefronts per workgroup of the panel kernel: 4, 8 or 16; 0 = 16 */
    int32_t graph;             /* 0 = replay each outer iteration of the fused schedule as one hipGraph (single
                                  GPU, no per-launch profiling): removes host launch cost when the kernels are
                                  only a few microseconds long; -1 = always launch eagerly */
    int32_t layout_build;      /* where the one-time panel-major layout is built: 0 = on the GPU when the pattern
                                  allows it (inside every segment the entries of one panel are consecutive, e.g.
                                  ascending indices -- what every CSR/CSC converter produces), else on the host;
                                  1 = host builder; 2 = GPU builder or MFX_ERR_INVALID */
    /* ---- opt-in extensions, all 0 by default.  The reference PARSES -N, -e, -p, -q (src/pmf.h:33-36) and reads none
     * of them in its solvers, so a drop-in must ignore them by default too; set these to give them the meaning they
     * have in LIBPMF 1.41's ccd-r1.cpp, of which the reference is a fork (CCD++ only, single GPU for eps / rank_trace).
     * do_nmf and eps are "parity unpinned" (no code or fixture in the reference tree; checked against the oracle's
     * restatement of the published algorithm); rank_trace is the reference's own calrmse_r1 (src/tools.cpp:261-270),
     * whose call site is commented out at src/CCD.cpp:141-148. */
    int32_t do_nmf;            /* -N: new coordinate values are clamped at 0 (non-negative factorisation) */
    float eps;                 /* -e: > 0 enables the function-decrease stopping rule of the inner iterations (and of the
                                  rank loop after five ranks that stop at once); costs one host sync per inner iteration */
    int32_t rank_trace;        /* -p with -q: test RMSE after every rank (mfx_ccd_rank_trace); with verbose also printed
                                  as the reference's commented line "iter %d rank %d time %f rmse %f" */
} mfx_params;

/* One outer iteration's numbers == the fields of the reference's log line
 * (cuda_src/CCD_CUDA.cu:405-406, cuda_src/ALS_CUDA.cu:360-361), in seconds. */
typedef struct mfx_iter_report {
    double rank_time;   /* CCD++: rank-one sweeps (GPU time, HIP events)            */
    double update_time; /* CCD++: residual updates; ALS: the whole iteration        */
    double rmse;        /* test RMSE after this iteration (0 if no test set)        */
    double rmse_time;
} mfx_iter_report;

const char* mfx_last_error(void);
int mfx_version(void);
/* Number of usable HIP devices (0 if none); never fails. */
int mfx_device_count(void);
void mfx_params_default(mfx_params* p); /* reference defaults, src/pmf.h:26-42 */

/* ------------------------------------------------------------------------------------
 * One-shot solvers: exactly what runCUDA() calls (src/main.cpp:11-17).
 * Replaces kernel_wrapper_ccdpp_NV (cuda_src/CCD_CUDA.cu:164-179) / ccdpp_NV (:224-451).
 *   W [k][rows] in: initial factors (initial_col, src/tools.cpp:165-173); out: result.
 *   H [k][cols] in: ignored -- CCD++ starts from H = 0 (CCD_CUDA.cu:287); out: result.
 *   reports: NULL or [maxiter].
 * ---------------------------------------------------------------------------------- */
int mfx_ccdpp_run(const mfx_csx* R, const mfx_coo* T, float* W, float* H, const mfx_params* p,
                  mfx_iter_report* reports);
/* Replaces kernel_wrapper_als_NV (cuda_src/ALS_CUDA.cu:183-198) / als_NV (:200-406).
 *   W [rows][k] in: ignored (overwritten before first read); H [cols][k] in: initial. */
int mfx_als_run(const mfx_csx* R, const mfx_coo* T, float* W, float* H, const mfx_params* p,
                mfx_iter_report* reports);

/* ------------------------------------------------------------------------------------
 * Resident solvers: the same loops split into create / iterate / fetch so that a caller
 * (bench, a service, a multi-GPU driver) can keep R in HBM across calls and time only
 * the iterations.  `space` says whether R, T and the factor pointers are host or device
 * pointers (device pointers must belong to p->device).
 * ---------------------------------------------------------------------------------- */
typedef struct mfx_comm_s* mfx_comm_t;
typedef struct mfx_ccd_s* mfx_ccd_t;
typedef struct mfx_als_s* mfx_als_t;

/* Multi-GPU description of one user-row-block shard.  R passed to
 * mfx_ccd_create is then the LOCAL sub-matrix: rows = this rank's rows, cols = all items,
 * csc_* = the local CSC over local row ids.  NULL means "single GPU". */
typedef struct mfx_shard {
    mfx_comm_t comm;                 /* communicator over all shards                      */
    const uint32_t* global_col_nnz;  /* [cols] |Omega_c| over ALL shards (lambda scaling) */
    int64_t global_test_nnz;         /* Zt over all shards (RMSE denominator)             */
} mfx_shard;

int mfx_ccd_create(mfx_ccd_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                   mfx_memspace space, const mfx_shard* shard);
/* W [k][rows] required; H [k][cols] may be NULL (= zeros, the reference's start). */
int mfx_ccd_set_factors(mfx_ccd_t s, const float* W, const float* H, mfx_memspace space);
/* Runs n_outer more outer iterations (the iteration counter persists, so the first ever
 * iteration skips the add-back exactly like oiter == 1 in src/CCD.cpp:100).  reports:
 * NULL or [n_outer].  with_rmse = 0 skips the per-iteration test RMSE. */
int mfx_ccd_iterate(mfx_ccd_t s, int n_outer, int with_rmse, mfx_iter_report* reports);
int mfx_ccd_get_factors(mfx_ccd_t s, float* W, float* H, mfx_memspace space);
/* Copies the two residual copies out (test hook: R-hat in CSC order and in CSR order). */
int mfx_ccd_get_residual(mfx_ccd_t s, float* csc_val, float* csr_val);
/* Per-kernel GPU time of the last mfx_ccd_iterate call, measured with HIP events on the
 * solver's stream: names[i] / seconds[i] / launches[i] for i < returned count (<= cap). */
int mfx_ccd_kernel_times(mfx_ccd_t s, int cap, const char** names, double* seconds,
                         int64_t* launches);
/* Per-rank trace of the last mfx_ccd_iterate call (mfx_params.rank_trace, and the ranks the eps rule let run):
 * rmse / seconds are [outer iterations of that call][k] (up to cap entries; NaN / 0 for skipped ranks), ranks_done
 * [iters_cap] the number of ranks updated per outer iteration.  Any pointer may be NULL.  Returns the number of
 * outer iterations recorded (0 when no extension was on). */
int mfx_ccd_rank_trace(mfx_ccd_t s, int cap, double* rmse, double* seconds, int iters_cap, int32_t* ranks_done);
/* Turns the per-launch event bracketing (mfx_params.profile) on or off between iterate calls. */
int mfx_ccd_set_profile(mfx_ccd_t s, int on);
/* Layout the solver chose for one residual copy (side 0 = CSC / column segments, 1 = CSR / row
 * segments): out[0] = panels, out[1] = entries per panel (0 = plain layout), out[2] = 2 scatter
 * layout (3: with 32-bit segment ids) / 1 LDS panels / 0 cache panels or plain, out[3] = tiles per span (tile order: segments per block).  For logs,
 * benchmarks and tests. */
int mfx_ccd_layout_info(mfx_ccd_t s, int side, int32_t out[4]);
int mfx_ccd_destroy(mfx_ccd_t s);

int mfx_als_create(mfx_als_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                   mfx_memspace space);
/* Multi-GPU ALS (SURVEY.md 8e "ALS", 8f N4): rank g owns user rows [row_lo,row_hi) for the W-half
 * and item columns [col_lo,col_hi) for the H-half; W and H are replicated and every half ends with
 * ONE grouped exchange in which every rank broadcasts its freshly solved block.  Create itself runs no collective:
 * the other ranks' block boundaries are gathered by the first mfx_als_iterate, i.e. after mfx_comm_agree.  R->rows / R->cols are the GLOBAL sizes;
 * csr_* describe the local rows (row_ptr rebased to 0, GLOBAL column indices), csc_* the local
 * columns (col_ptr rebased to 0, GLOBAL row indices); R->nnz is ignored (each orientation's count
 * is the last entry of its pointer array).  T holds the test ratings of the local rows with GLOBAL
 * indices.  Host pointers only. */
typedef struct mfx_als_shard {
    mfx_comm_t comm;
    int64_t row_lo, row_hi, col_lo, col_hi;
    int64_t global_test_nnz;
} mfx_als_shard;
int mfx_als_create_sharded(mfx_als_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                           const mfx_als_shard* shard);
int mfx_als_set_factors(mfx_als_t s, const float* W, const float* H, mfx_memspace space);
int mfx_als_iterate(mfx_als_t s, int n_iter, int with_rmse, mfx_iter_report* reports);
int mfx_als_get_factors(mfx_als_t s, float* W, float* H, mfx_memspace space);
int mfx_als_kernel_times(mfx_als_t s, int cap, const char** names, double* seconds,
                         int64_t* launches);
int mfx_als_destroy(mfx_als_t s);

/* ------------------------------------------------------------------------------------
 * Single operators (host pointers in, host pointers out): one call per reference
 * function on the path, used by the parity tests.
 * ---------------------------------------------------------------------------------- */
/* RankOneUpdate_v_kernel / _u_kernel (cuda_src/CCD_CUDA.cu:24-58) == the sweep of
 * src/CCD.cpp:110-113: out[c] = sum(vec[idx]*val) / (lambda*|Omega_c| + sum(vec[idx]^2)),
 * 0 for an empty segment.  variant: -1 = the reference's own summation order (sequential fp32, bit-identical to
 * the CPU reference), 0 = wave-per-segment kernel, 1 = flat-stream kernel gathering
 * from L2, 2 = flat-stream kernel with LDS panels (size chosen), >= 16 = LDS panels of `variant`
 * gathered entries, <= -16 = cache panels of `-variant` entries (test hooks: force many panels on
 * small inputs). */
int mfx_rank_one_sweep(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx,
                       const float* val, int64_t vec_len, const float* vec, float lambda,
                       float* out, int variant, int device);
/* UpdateRating_DUAL_kernel_NoLoss, one copy (cuda_src/CCD_CUDA.cu:60-82) ==
 * UpdateRating_Original_float (src/CCD.cpp:18-43): val[p] +=/-= gathered[idx[p]]*per_seg[c]. */
int mfx_update_rating(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx,
                      float* val, int64_t vec_len, const float* gathered, const float* per_seg,
                      int add, int variant, int device);
/* GPU_rmse + host reduction (cuda_src/CUDA_AUX.cu:3-27, CCD_CUDA.cu:383-401) ==
 * calrmse (src/tools.cpp:235-248): fp32 products, fp64 sums. */
int mfx_test_rmse(const mfx_coo* T, const float* W, const float* H, int64_t rows, int64_t cols,
                  int64_t k, int ifALS, double* rmse_out, int device);
/* Mt_byM_multiply_k (cuda_src/ALS_CUDA.cu:65-79): A[k][k] = sum over idx of x x^T (cnt <= 2048: one wavefront's
 * share of a segment; the half-sweep below splits longer segments and adds the pieces in a fixed order). */
int mfx_als_gramian(int64_t cnt, const uint32_t* idx, int64_t nrows_x, const float* X, int64_t k,
                    float* A, int device);
/* inverseMatrix_CholeskyMethod_k (cuda_src/ALS_CUDA.cu:41-62) == choldc1 / choldcsl /
 * inverseMatrix_CholeskyMethod (src/ALS.cpp:6-64) on one k x k matrix, in the reference's operation order
 * (bit-identical to the CPU reference): Ainv = A^-1 via Cholesky, mirrored. */
int mfx_als_inverse(int64_t k, const float* A, float* Ainv, int device);
/* One ALS half-sweep, updateW_overH_kernel / updateH_overW_kernel
 * (cuda_src/ALS_CUDA.cu:81-181): for every segment solve (X^T X + lambda I) y = X^T r.
 * variant 1 (default path): MFMA Gramian, Cholesky factorisation and two triangular solves; variant 0:
 * "as written" -- Gramian, explicit inverse and products in the reference's own operation order
 * (src/ALS.cpp:66-79, 6-64, 129-142), bit-identical to the CPU reference; the parity mode, also selected by
 * mfx_params.schedule = 0 in mfx_als_run / mfx_als_create. */
int mfx_als_half(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx,
                 const float* val, int64_t nrows_x, const float* X, float* Y, int64_t k,
                 float lambda, int variant, int device);

/* ------------------------------------------------------------------------------------
 * Communicator over RCCL (the reference has no distributed code).
 * One process per GPU: rank 0 calls mfx_comm_unique_id, ships the 128 bytes to the other
 * ranks by any means (bench.py uses torch.distributed), every rank calls mfx_comm_create.
 * ---------------------------------------------------------------------------------- */
#define MFX_COMM_ID_BYTES 128
int mfx_comm_unique_id(void* id_out /* MFX_COMM_ID_BYTES */);
int mfx_comm_create(mfx_comm_t* out, const void* id, int rank, int nranks, int device);
/* In-process loopback communicator: the `nranks` ranks of `group` are threads of ONE process (each
 * driving its own solver, on the same or on different devices); collectives are staged through
 * host memory and summed in rank order.  Slow by design -- it exists to run the sharded solver path
 * through the real kernels where RCCL cannot (two ranks on one GPU); production uses mfx_comm_create. */
int mfx_comm_create_local(mfx_comm_t* out, int group, int rank, int nranks, int device);
/* Collective over all ranks: *global_status = the worst (most negative) local_status.  Protocol for a
 * sharded solve: every rank runs its own setup (extract its shard, mfx_ccd_create / mfx_als_create_sharded,
 * set factors), then ALL ranks -- the ones whose setup failed too, with their error code -- call this once;
 * only if *global_status == MFX_OK does anyone go on to iterate.  No solver entry point runs a collective
 * before its first iterate call, so a failed rank can never leave the others waiting inside one.  (On RCCL
 * this is also where the lazy connection setup is paid.) */
int mfx_comm_agree(mfx_comm_t c, int local_status, int* global_status);
/* For a rank that fails AFTER the collectives started: releases the ranks waiting for it (loopback group:
 * every present and future collective returns MFX_ERR_COMM; RCCL: ncclCommAbort on every communicator of the
 * same unique id in this process).  The communicator can only be destroyed afterwards. */
int mfx_comm_abort(mfx_comm_t c);
int mfx_comm_rank(mfx_comm_t c);
int mfx_comm_size(mfx_comm_t c);
int mfx_comm_destroy(mfx_comm_t c);

/* ------------------------------------------------------------------------------------
 * Host-side helpers on the path (no GPU needed).
 * ---------------------------------------------------------------------------------- */
/* initial_col (src/tools.cpp:165-173): X flat [k][n], srand(0) then glibc rand() consumed
 * with i (0..n) outer and j (0..k) inner: X[j*n+i] = 0.1f*(float(rand())/RAND_MAX)+0.001f.
 * CCD++ calls it as (W, k, rows)/(H, k, cols); ALS as (W, rows, k)/(H, cols, k)
 * (src/main.cpp:86-98). */
void mfx_initial_col(float* X, int64_t k, int64_t n);
/* nnz-balanced contiguous row-block partition: bounds[g] .. bounds[g+1] are shard g's rows
 * (prefix sums of csr_row_ptr, SURVEY.md 8e "Partition").  bounds has nshards+1 entries. */
int mfx_partition_rows(int64_t rows, const uint32_t* csr_row_ptr, int nshards, int64_t* bounds);
/* Extracts shard [row_lo,row_hi) of R into caller-allocated arrays: local CSR (row_ptr
 * rebased to 0) and local CSC over local row ids (entries keep R's per-column order).
 * local_nnz = csr_row_ptr[row_hi]-csr_row_ptr[row_lo] sizes the idx/val arrays. */
int mfx_extract_shard(const mfx_csx* R, int64_t row_lo, int64_t row_hi, uint32_t* l_csr_row_ptr,
                      uint32_t* l_csr_col_idx, float* l_csr_val, uint32_t* l_csc_col_ptr,
                      uint32_t* l_csc_row_idx, float* l_csc_val);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H */

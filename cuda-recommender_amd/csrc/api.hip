// api.hip -- the extern "C" surface declared in include/mfx.h.
#include <algorithm>
#include <cmath>
#include <memory>
#include <new>
#include <stdexcept>

#include "als_solver.hpp"
#include "ccd_solver.hpp"

namespace mfx {

std::string& last_error() {
    static thread_local std::string e;
    return e;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MFX_ERR_NO_DEVICE, "no usable HIP device (%s); libmfx has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(MFX_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
    MFX_HIP(hipSetDevice(device));
    // hipGetLastError() is a sticky per-thread slot shared with every other HIP user of the process
    // (RCCL under torch.distributed leaves "invalid device ordinal" there while probing peers).  Every
    // libmfx launch reads the slot right after itself, so whatever sits in it on ENTRY to a libmfx call
    // is somebody else's: drop it here, once per call -- never between our own launches.
    (void) hipGetLastError();
    return MFX_OK;
}

}  // namespace mfx

using namespace mfx;

// No C++ exception may cross the C ABI (std::bad_alloc from a multi-GB host vector, std::system_error
// from a thread that could not be started, ...): every entry point that does work runs inside this.
template <typename F>
static int guarded(const char* what, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(MFX_ERR_ALLOC, "%s: out of host memory", what);
    } catch (const std::exception& ex) {
        return fail(MFX_ERR_INVALID, "%s: %s", what, ex.what());
    } catch (...) {
        return fail(MFX_ERR_INVALID, "%s: unknown C++ exception", what);
    }
}

extern "C" {

const char* mfx_last_error(void) { return last_error().c_str(); }
int mfx_version(void) { return MFX_VERSION; }

int mfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mfx_params_default(mfx_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->k = 10;            /* src/pmf.h:28 */
    p->lambda = 0.1f;     /* src/pmf.h:32 */
    p->maxiter = 5;       /* src/pmf.h:30 */
    p->maxinneriter = 1;  /* src/pmf.h:31 */
    p->nBlocks = 32;      /* src/pmf.h:39 */
    p->nThreadsPerBlock = 256;
    p->schedule = 1;
    p->kernel_variant = 1;
}

/* ------------------------------------------------------------------ resident CCD++ */
int mfx_ccd_create(mfx_ccd_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                   mfx_memspace space, const mfx_shard* shard) {
    return guarded("mfx_ccd_create", [&]() -> int {
        MFX_REQUIRE(out, "mfx_ccd_create: out is NULL");
        *out = nullptr;
        CcdSolver* s = nullptr;
        MFX_TRY(CcdSolver::create(&s, R, T, p, space, shard));
        *out = new mfx_ccd_s{s};
        return MFX_OK;
    });
}
int mfx_ccd_set_factors(mfx_ccd_t s, const float* W, const float* H, mfx_memspace space) {
    return guarded("mfx_ccd_set_factors", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->set_factors(W, H, space);
    });
}
int mfx_ccd_iterate(mfx_ccd_t s, int n_outer, int with_rmse, mfx_iter_report* reports) {
    return guarded("mfx_ccd_iterate", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->iterate(n_outer, with_rmse, reports);
    });
}
int mfx_ccd_get_factors(mfx_ccd_t s, float* W, float* H, mfx_memspace space) {
    return guarded("mfx_ccd_get_factors", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->get_factors(W, H, space);
    });
}
int mfx_ccd_get_residual(mfx_ccd_t s, float* csc_val, float* csr_val) {
    return guarded("mfx_ccd_get_residual", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->get_residual(csc_val, csr_val);
    });
}
int mfx_ccd_kernel_times(mfx_ccd_t s, int cap, const char** names, double* seconds, int64_t* launches) {
    return guarded("mfx_ccd_kernel_times", [&]() -> int {
        if (!s || !s->impl) return 0;
        KernelProfiler& pr = s->impl->profiler();
        int n = 0;
        for (int i = 0; i < KernelProfiler::K_COUNT && n < cap; ++i) {
            if (pr.launches[i] == 0) continue;
            if (names) names[n] = KernelProfiler::name(i);
            if (seconds) seconds[n] = pr.seconds[i];
            if (launches) launches[n] = pr.launches[i];
            ++n;
        }
        pr.reset_totals();
        return n;
    });
}
int mfx_ccd_set_profile(mfx_ccd_t s, int on) {
    return guarded("mfx_ccd_set_profile", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->set_profile(on != 0);
    });
}
int mfx_ccd_rank_trace(mfx_ccd_t s, int cap, double* rmse, double* seconds, int iters_cap, int32_t* ranks_done) {
    int n = 0;
    const int rc = guarded("mfx_ccd_rank_trace", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null argument");
        MFX_REQUIRE(cap >= 0 && iters_cap >= 0, "negative capacity");
        n = s->impl->rank_trace(cap, rmse, seconds, iters_cap, ranks_done);
        return MFX_OK;
    });
    return rc == MFX_OK ? n : rc;
}
int mfx_ccd_layout_info(mfx_ccd_t s, int side, int32_t out[4]) {
    return guarded("mfx_ccd_layout_info", [&]() -> int {
        MFX_REQUIRE(s && s->impl && out, "null argument");
        MFX_REQUIRE(side == 0 || side == 1, "side must be 0 (CSC) or 1 (CSR)");
        s->impl->layout_info(side, out);
        return MFX_OK;
    });
}
int mfx_ccd_destroy(mfx_ccd_t s) {
    return guarded("mfx_ccd_destroy", [&]() -> int {
        if (!s) return MFX_OK;
        delete s->impl;
        delete s;
        return MFX_OK;
    });
}

/* ------------------------------------------------------------------ one-shot CCD++ */
int mfx_ccdpp_run(const mfx_csx* R, const mfx_coo* T, float* W, float* H, const mfx_params* p,
                  mfx_iter_report* reports) {
    return guarded("mfx_ccdpp_run", [&]() -> int {
        MFX_REQUIRE(R && W && H && p, "mfx_ccdpp_run: null argument");
        mfx_ccd_t s = nullptr;
        int rc = mfx_ccd_create(&s, R, T, p, MFX_HOST, nullptr);
        if (rc == MFX_OK) rc = mfx_ccd_set_factors(s, W, nullptr, MFX_HOST);
        if (rc == MFX_OK) rc = mfx_ccd_iterate(s, p->maxiter, 1, reports);
        if (rc == MFX_OK) rc = mfx_ccd_get_factors(s, W, H, MFX_HOST);
        mfx_ccd_destroy(s);
        if (rc != MFX_OK) fprintf(stderr, "CCD FAILED: %s\n", mfx_last_error()); /* CCD_CUDA.cu:174-176 */
        return rc;
    });
}

/* ------------------------------------------------------------------ ALS */
int mfx_als_create(mfx_als_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p, mfx_memspace space) {
    return guarded("mfx_als_create", [&]() -> int {
        MFX_REQUIRE(out, "mfx_als_create: out is NULL");
        *out = nullptr;
        AlsSolver* s = nullptr;
        MFX_TRY(AlsSolver::create(&s, R, T, p, space));
        *out = new mfx_als_s{s};
        return MFX_OK;
    });
}
int mfx_als_create_sharded(mfx_als_t* out, const mfx_csx* R, const mfx_coo* T, const mfx_params* p,
                           const mfx_als_shard* shard) {
    return guarded("mfx_als_create_sharded", [&]() -> int {
        MFX_REQUIRE(out && shard && shard->comm, "mfx_als_create_sharded: null argument");
        *out = nullptr;
        AlsSolver* s = nullptr;
        MFX_TRY(AlsSolver::create(&s, R, T, p, MFX_HOST, shard));
        *out = new mfx_als_s{s};
        return MFX_OK;
    });
}
int mfx_als_set_factors(mfx_als_t s, const float* W, const float* H, mfx_memspace space) {
    return guarded("mfx_als_set_factors", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->set_factors(W, H, space);
    });
}
int mfx_als_iterate(mfx_als_t s, int n_iter, int with_rmse, mfx_iter_report* reports) {
    return guarded("mfx_als_iterate", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->iterate(n_iter, with_rmse, reports);
    });
}
int mfx_als_get_factors(mfx_als_t s, float* W, float* H, mfx_memspace space) {
    return guarded("mfx_als_get_factors", [&]() -> int {
        MFX_REQUIRE(s && s->impl, "null solver");
        return s->impl->get_factors(W, H, space);
    });
}
int mfx_als_kernel_times(mfx_als_t s, int cap, const char** names, double* seconds, int64_t* launches) {
    return guarded("mfx_als_kernel_times", [&]() -> int {
        if (!s || !s->impl) return 0;
        return s->impl->kernel_times(cap, names, seconds, launches);
    });
}
int mfx_als_destroy(mfx_als_t s) {
    return guarded("mfx_als_destroy", [&]() -> int {
        if (!s) return MFX_OK;
        delete s->impl;
        delete s;
        return MFX_OK;
    });
}
int mfx_als_run(const mfx_csx* R, const mfx_coo* T, float* W, float* H, const mfx_params* p,
                mfx_iter_report* reports) {
    return guarded("mfx_als_run", [&]() -> int {
        MFX_REQUIRE(R && W && H && p, "mfx_als_run: null argument");
        mfx_als_t s = nullptr;
        int rc = mfx_als_create(&s, R, T, p, MFX_HOST);
        if (rc == MFX_OK) rc = mfx_als_set_factors(s, nullptr, H, MFX_HOST);
        if (rc == MFX_OK) rc = mfx_als_iterate(s, p->maxiter, 1, reports);
        if (rc == MFX_OK) rc = mfx_als_get_factors(s, W, H, MFX_HOST);
        mfx_als_destroy(s);
        if (rc != MFX_OK) fprintf(stderr, "ALS FAILED: %s\n", mfx_last_error()); /* ALS_CUDA.cu:193-195 */
        return rc;
    });
}

/* ------------------------------------------------------------------ single operators */
namespace {
struct OpCtx {
    hipStream_t st = nullptr;
    ~OpCtx() { if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); } }
    int open(int device) {
        MFX_TRY(use_device(device));
        MFX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return MFX_OK;
    }
};
}  // namespace

// variant -> layout of the single-operator entry points (see mfx.h)
// variants of the single-operator entry points: -1 reference order / 0 wave per segment (plain layout), 1 flat (plain),
// 2 flat with the layout chosen for the shape, >= 16 explicit LDS panel, <= -16 explicit cache panel.  Anything else is an error.
static bool op_variant_ok(int variant) { return variant == -1 || variant == 0 || variant == 1 || variant == 2 || variant >= 16 || variant <= -16; }
static FlatLayoutOptions op_layout(int variant, int64_t nseg, int64_t nnz, int64_t vec_len) {
    mfx_params p;
    mfx_params_default(&p);
    p.panel_rows = (variant >= 16 || variant <= -16) ? variant : variant == 2 ? 0 : -1;
    return choose_layout(p, (uint32_t) nseg, (uint64_t) nnz, (uint32_t) vec_len, sizeof(float), variant == 0 || variant == -1);
}

int mfx_rank_one_sweep(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx, const float* val,
                       int64_t vec_len, const float* vec, float lambda, float* out, int variant, int device) {
    return guarded("mfx_rank_one_sweep", [&]() -> int {
        MFX_REQUIRE(nseg > 0 && nnz >= 0 && vec_len > 0 && ptr && vec && out, "mfx_rank_one_sweep: bad argument");
        MFX_REQUIRE(nnz == 0 || (idx && val), "mfx_rank_one_sweep: null idx / val with nnz > 0");
        MFX_REQUIRE(nseg < (int64_t) 0xFFFFFFFFll && vec_len < (int64_t) 0xFFFFFFFFll && nnz < (int64_t) 0xFFFF0000ll,
                    "mfx_rank_one_sweep: sizes exceed the 32-bit index range");
        MFX_REQUIRE(op_variant_ok(variant), "mfx_rank_one_sweep: variant must be -1, 0, 1, 2, >= 16 or <= -16 (got %d)", variant);
        OpCtx cx;
        MFX_TRY(cx.open(device));
        SegStreamStore s;
        MFX_TRY(s.build((uint32_t) nseg, (uint64_t) nnz, (uint32_t) vec_len, ptr, idx, val, MFX_HOST,
                        op_layout(variant, nseg, nnz, vec_len), 0, cx.st));
        DevBuf<float> dvec, dout, gh;
        MFX_TRY(dvec.alloc(vec_len)); MFX_TRY(dvec.upload(vec, vec_len, MFX_HOST, cx.st));
        MFX_TRY(dout.alloc(nseg));
        FinalizeArgs f; f.lambda = lambda; f.out_vec = dout.get();
        if (variant == -1) {  // the reference's summation order: sums and division in one kernel (ccd_reforder.hip)
            std::vector<uint32_t> order;
            const char* e_fused = std::getenv("MFX_REF_FUSED");
            const bool old_form = e_fused && std::atoi(e_fused) == 0;  // (A/B: the r3 / early-r4 kernels)
            const uint32_t nlong = ref_sweep_order(ptr, (uint32_t) nseg, &order, !old_form);
            DevBuf<uint32_t> dorder;
            MFX_TRY(dorder.alloc(order.size())); MFX_TRY(dorder.upload(order.data(), order.size(), MFX_HOST, cx.st));
            if (old_form) {
                MFX_TRY(launch_sweep_ref(s.view, dorder.get(), nlong, dvec.get(), lambda, dout.get(), cx.st));
            } else {
                RefStreams rs;
                rs.main = cx.st;  // (one stream: the long segments' kernel first, the others behind it)
                MFX_TRY(launch_ref_owner(FM_SWEEP, s.view, dorder.get(), nlong, dvec.get(), nullptr, f, rs));
            }
            MFX_HIP(hipMemcpyAsync(out, dout.get(), sizeof(float) * nseg, hipMemcpyDeviceToHost, cx.st));
            MFX_HIP(hipStreamSynchronize(cx.st));
            return MFX_OK;
        }
        if (variant == 0) {
            MFX_TRY(gh.alloc_zero((size_t) 2 * nseg, cx.st));
            MFX_TRY(launch_sweep_wave(s.view, dvec.get(), gh.get(), gh.get() + nseg, cx.st));
            f.gh_dense = gh.get();
        } else {
            MFX_TRY(launch_flat(FM_SWEEP, s.view, dvec.get(), nullptr, 0, cx.st));
        }
        MFX_TRY(launch_finalize(s.view, f, cx.st));
        MFX_HIP(hipMemcpyAsync(out, dout.get(), sizeof(float) * nseg, hipMemcpyDeviceToHost, cx.st));
        MFX_HIP(hipStreamSynchronize(cx.st));
        return MFX_OK;
    });
}

int mfx_update_rating(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx, float* val,
                      int64_t vec_len, const float* gathered, const float* per_seg, int add, int variant,
                      int device) {
    return guarded("mfx_update_rating", [&]() -> int {
        MFX_REQUIRE(nseg > 0 && nnz >= 0 && vec_len > 0 && ptr && gathered && per_seg, "mfx_update_rating: bad argument");
        MFX_REQUIRE(nnz == 0 || (idx && val), "mfx_update_rating: null idx / val with nnz > 0");
        MFX_REQUIRE(nseg < (int64_t) 0xFFFFFFFFll && vec_len < (int64_t) 0xFFFFFFFFll && nnz < (int64_t) 0xFFFF0000ll,
                    "mfx_update_rating: sizes exceed the 32-bit index range");
        MFX_REQUIRE(op_variant_ok(variant), "mfx_update_rating: variant must be -1, 0, 1, 2, >= 16 or <= -16 (got %d)", variant);
        OpCtx cx;
        MFX_TRY(cx.open(device));
        SegStreamStore s;
        MFX_TRY(s.build((uint32_t) nseg, (uint64_t) nnz, (uint32_t) vec_len, ptr, idx, val, MFX_HOST,
                        op_layout(variant, nseg, nnz, vec_len), 0, cx.st));
        DevBuf<float> dg, dp;
        MFX_TRY(dg.alloc(vec_len)); MFX_TRY(dg.upload(gathered, vec_len, MFX_HOST, cx.st));
        MFX_TRY(dp.alloc(nseg)); MFX_TRY(dp.upload(per_seg, nseg, MFX_HOST, cx.st));
        // 0: one wavefront per segment; -1 takes the flat kernel over its plain layout, like CcdSolver::resid (the update is
        // elementwise -- bit-identical in every kernel)
        if (variant == 0) MFX_TRY(launch_resid_wave(s.view, dg.get(), dp.get(), add, cx.st));
        else MFX_TRY(launch_flat(FM_RESID, s.view, dg.get(), dp.get(), add, cx.st));
        if (nnz) {
            DevBuf<float> tmp;
            MFX_TRY(tmp.alloc(nnz));
            MFX_TRY(s.unpermute(tmp.get(), cx.st));
            MFX_HIP(hipMemcpyAsync(val, tmp.get(), sizeof(float) * nnz, hipMemcpyDeviceToHost, cx.st));
            MFX_HIP(hipStreamSynchronize(cx.st));
        }
        MFX_HIP(hipStreamSynchronize(cx.st));
        return MFX_OK;
    });
}

int mfx_test_rmse(const mfx_coo* T, const float* W, const float* H, int64_t rows, int64_t cols, int64_t k,
                  int ifALS, double* rmse_out, int device) {
    return guarded("mfx_test_rmse", [&]() -> int {
        MFX_REQUIRE(T && W && H && rmse_out && rows > 0 && cols > 0 && k > 0, "mfx_test_rmse: bad argument");
        MFX_REQUIRE(rows < (int64_t) 0xFFFFFFFFll && cols < (int64_t) 0xFFFFFFFFll, "mfx_test_rmse: sizes exceed the 32-bit index range");
        MFX_REQUIRE(T->nnz <= 0 || (T->row && T->col && T->val), "mfx_test_rmse: null test array");
        *rmse_out = 0.0;
        if (T->nnz <= 0) return MFX_OK;
        OpCtx cx;
        MFX_TRY(cx.open(device));
        DevBuf<uint32_t> r, c; DevBuf<float> v, dW, dH; DevBuf<double> part, sum;
        MFX_TRY(r.alloc(T->nnz)); MFX_TRY(r.upload(T->row, T->nnz, MFX_HOST, cx.st));
        MFX_TRY(c.alloc(T->nnz)); MFX_TRY(c.upload(T->col, T->nnz, MFX_HOST, cx.st));
        MFX_TRY(v.alloc(T->nnz)); MFX_TRY(v.upload(T->val, T->nnz, MFX_HOST, cx.st));
        MFX_TRY(check_index_range(r.get(), (uint64_t) T->nnz, (uint32_t) rows, "test-set row", cx.st));
        MFX_TRY(check_index_range(c.get(), (uint64_t) T->nnz, (uint32_t) cols, "test-set column", cx.st));
        MFX_TRY(dW.alloc((size_t) rows * k)); MFX_TRY(dW.upload(W, (size_t) rows * k, MFX_HOST, cx.st));
        MFX_TRY(dH.alloc((size_t) cols * k)); MFX_TRY(dH.upload(H, (size_t) cols * k, MFX_HOST, cx.st));
        MFX_TRY(part.alloc_zero(kRmseBlocks, cx.st)); MFX_TRY(sum.alloc_zero(1, cx.st));
        MFX_TRY(launch_test_sqerr(T->nnz, r.get(), c.get(), v.get(), dW.get(), dH.get(), rows, cols, k, ifALS,
                                  part.get(), kRmseBlocks, sum.get(), cx.st));
        double s = 0;
        MFX_HIP(hipMemcpyAsync(&s, sum.get(), sizeof(double), hipMemcpyDeviceToHost, cx.st));
        MFX_HIP(hipStreamSynchronize(cx.st));
        *rmse_out = std::sqrt(s / (double) T->nnz);
        return MFX_OK;
    });
}

int mfx_als_gramian(int64_t cnt, const uint32_t* idx, int64_t nrows_x, const float* X, int64_t k, float* A,
                    int device) {
    return guarded("mfx_als_gramian", [&]() -> int {
        MFX_REQUIRE(cnt >= 0 && nrows_x > 0 && k > 0 && X && A && (cnt == 0 || idx), "mfx_als_gramian: bad argument");
        return als_gramian_op(cnt, idx, nrows_x, X, k, A, device);
    });
}

int mfx_als_inverse(int64_t k, const float* A, float* Ainv, int device) {
    return guarded("mfx_als_inverse", [&]() -> int {
        MFX_REQUIRE(A && Ainv, "mfx_als_inverse: null argument");
        return als_inverse_op(k, A, Ainv, device);
    });
}

int mfx_als_half(int64_t nseg, int64_t nnz, const uint32_t* ptr, const uint32_t* idx, const float* val,
                 int64_t nrows_x, const float* X, float* Y, int64_t k, float lambda, int variant, int device) {
    return guarded("mfx_als_half", [&]() -> int {
        MFX_REQUIRE(nseg > 0 && nnz >= 0 && ptr && X && Y && k > 0 && nrows_x > 0, "mfx_als_half: bad argument");
        MFX_REQUIRE(nnz == 0 || (idx && val), "mfx_als_half: null idx / val with nnz > 0");
        MFX_REQUIRE(nseg < (int64_t) 0xFFFFFFFFll && nrows_x < (int64_t) 0xFFFFFFFFll && nnz < (int64_t) 0xFFFF0000ll,
                    "mfx_als_half: sizes exceed the 32-bit index range");
        return als_half_op(nseg, nnz, ptr, idx, val, nrows_x, X, Y, k, lambda, variant, device);
    });
}

/* ------------------------------------------------------------------ communicator */
int mfx_comm_unique_id(void* id_out) {
    return guarded("mfx_comm_unique_id", [&]() -> int {
        MFX_REQUIRE(id_out, "null id buffer");
        return comm_unique_id(id_out);
    });
}
int mfx_comm_create(mfx_comm_t* out, const void* id, int rank, int nranks, int device) {
    return guarded("mfx_comm_create", [&]() -> int {
        return comm_create(out, id, rank, nranks, device);
    });
}
int mfx_comm_create_local(mfx_comm_t* out, int group, int rank, int nranks, int device) {
    return guarded("mfx_comm_create_local", [&]() -> int {
        return comm_create_local(out, group, rank, nranks, device);
    });
}
int mfx_comm_agree(mfx_comm_t c, int local_status, int* global_status) {
    return guarded("mfx_comm_agree", [&]() -> int { return comm_agree(c, local_status, global_status); });
}
int mfx_comm_abort(mfx_comm_t c) {
    return guarded("mfx_comm_abort", [&]() -> int { return comm_abort(c); });
}
int mfx_comm_rank(mfx_comm_t c) { return c ? c->rank : -1; }
int mfx_comm_size(mfx_comm_t c) { return c ? c->nranks : 0; }
int mfx_comm_destroy(mfx_comm_t c) { return comm_destroy(c); }

/* ------------------------------------------------------------------ host-side helpers */
void mfx_initial_col(float* X, int64_t k, int64_t n) {
    srand(0L);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < k; ++j) X[j * n + i] = 0.1f * (float(rand()) / (float) RAND_MAX) + 0.001f;
}

int mfx_partition_rows(int64_t rows, const uint32_t* csr_row_ptr, int nshards, int64_t* bounds) {
    return guarded("mfx_partition_rows", [&]() -> int {
        MFX_REQUIRE(rows > 0 && csr_row_ptr && nshards >= 1 && bounds, "mfx_partition_rows: bad argument");
        const uint64_t nnz = csr_row_ptr[rows];
        bounds[0] = 0;
        for (int g = 1; g < nshards; ++g) {
            // first row whose prefix reaches g/nshards of the non-zeros
            const uint64_t target = (nnz * (uint64_t) g + nshards - 1) / (uint64_t) nshards;
            const uint32_t* it = std::lower_bound(csr_row_ptr, csr_row_ptr + rows + 1, (uint32_t) target);
            int64_t r = it - csr_row_ptr;
            if (r < bounds[g - 1]) r = bounds[g - 1];
            if (r > rows) r = rows;
            bounds[g] = r;
        }
        bounds[nshards] = rows;
        return MFX_OK;
    });
}

int mfx_extract_shard(const mfx_csx* R, int64_t row_lo, int64_t row_hi, uint32_t* l_csr_row_ptr,
                      uint32_t* l_csr_col_idx, float* l_csr_val, uint32_t* l_csc_col_ptr,
                      uint32_t* l_csc_row_idx, float* l_csc_val) {
    return guarded("mfx_extract_shard", [&]() -> int {
        MFX_REQUIRE(R && row_lo >= 0 && row_lo <= row_hi && row_hi <= R->rows, "mfx_extract_shard: bad row range");
        MFX_REQUIRE(l_csr_row_ptr && l_csc_col_ptr, "mfx_extract_shard: null output");
        const uint32_t base = R->csr_row_ptr[row_lo];
        const uint32_t lnnz = R->csr_row_ptr[row_hi] - base;
        for (int64_t r = row_lo; r <= row_hi; ++r) l_csr_row_ptr[r - row_lo] = R->csr_row_ptr[r] - base;
        if (lnnz) {
            memcpy(l_csr_col_idx, R->csr_col_idx + base, sizeof(uint32_t) * lnnz);
            memcpy(l_csr_val, R->csr_val + base, sizeof(float) * lnnz);
        }
        // local CSC: keep every column's entries whose row falls in the block, in R's column order
        uint32_t w = 0;
        for (int64_t c = 0; c < R->cols; ++c) {
            l_csc_col_ptr[c] = w;
            for (uint32_t p = R->csc_col_ptr[c]; p < R->csc_col_ptr[c + 1]; ++p) {
                const uint32_t r = R->csc_row_idx[p];
                if (r >= (uint32_t) row_lo && r < (uint32_t) row_hi) {
                    l_csc_row_idx[w] = r - (uint32_t) row_lo;
                    l_csc_val[w] = R->csc_val[p];
                    ++w;
                }
            }
        }
        l_csc_col_ptr[R->cols] = w;
        MFX_REQUIRE(w == lnnz, "CSR and CSC disagree on the shard's nnz (%u vs %u)", lnnz, w);
        return MFX_OK;
    });
}

}  // extern "C"

/*
 * mf_oracle.cpp -- CPU oracle (restatement of the reference's OpenMP CCD++/ALS path).
 * TEST INFRASTRUCTURE ONLY: see mf_oracle.h for who may use it and how it is pinned.
 *
 * Written from the reference's algorithm, not from its text: flat arrays instead of
 * vector<vector<float>>, one segment-walk helper shared by both orientations.  What IS
 * kept deliberately identical is the floating-point evaluation order, because the
 * fixtures in tests/golden/ are compared bit for bit.
 */
#include "mf_oracle.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

/* OpenMP schedule of the reference's hot loops (src/CCD.cpp:4, src/ALS.cpp:4). */
#define ORC_SCHED schedule(dynamic, 500)

/* One column of the rank-one update: src/CCD.cpp:6-16.  Strictly left-to-right fp32. */
inline float rank_one_column(const unsigned* ptr, const unsigned* idx, const float* val,
                             long c, const float* vec, float lambda_scaled) {
    const unsigned lo = ptr[c], hi = ptr[c + 1];
    if (lo == hi) return 0.0f;
    float g = 0.0f, h = lambda_scaled;
    for (unsigned p = lo; p < hi; ++p) {
        const float x = vec[idx[p]];
        g += x * val[p];
        h += x * x;
    }
    return g / h;
}

inline void sweep(long ncols, const unsigned* ptr, const unsigned* idx, const float* val,
                  const float* vec, float lambda, float* out) {
#pragma omp parallel for ORC_SCHED
    for (long c = 0; c < ncols; ++c) {
        /* float * unsigned, as src/CCD.cpp:112 / :120 */
        out[c] = rank_one_column(ptr, idx, val, c, vec, lambda * (ptr[c + 1] - ptr[c]));
    }
}

inline float residual_update(long ncols, const unsigned* ptr, const unsigned* idx, float* val,
                             const float* gathered, const float* per_col, bool add) {
    float loss = 0.0f;
#pragma omp parallel for ORC_SCHED reduction(+ : loss)
    for (long c = 0; c < ncols; ++c) {
        const float hc = per_col[c];
        float inner = 0.0f;
        if (add) {
            for (unsigned p = ptr[c]; p < ptr[c + 1]; ++p) {
                val[p] += gathered[idx[p]] * hc;
                inner += val[p] * val[p];
            }
        } else {
            for (unsigned p = ptr[c]; p < ptr[c + 1]; ++p) {
                val[p] -= gathered[idx[p]] * hc;
                inner += val[p] * val[p];
            }
        }
        loss += inner;
    }
    return loss;
}

/* src/tools.cpp:184-198: fp32 products accumulated in double. */
inline double predict(const float* W, const float* H, long i, long j, long m, long n, long k,
                      bool als) {
    double acc = 0.0;
    if (als) {
        for (long t = 0; t < k; ++t) acc += W[i * k + t] * H[j * k + t];
    } else {
        for (long t = 0; t < k; ++t) acc += W[t * m + i] * H[t * n + j];
    }
    return acc;
}

/* src/ALS.cpp:6-23.  a is k*k row-major. */
int cholesky_factor(long n, float* a, float* p) {
    int bad = 0;
    for (long i = 0; i < n; ++i) {
        for (long j = i; j < n; ++j) {
            float sum = a[i * n + j];
            for (long q = i - 1; q >= 0; --q) sum -= a[i * n + q] * a[j * n + q];
            if (i == j) {
                if (sum <= 0) ++bad;
                p[i] = sqrtf(sum);
            } else {
                a[j * n + i] = sum / p[i];
            }
        }
    }
    return bad;
}

/* src/ALS.cpp:25-39: L^-1 in the lower triangle, with the reference's DOUBLE accumulator. */
int cholesky_lower_inverse(long n, float* a, float* p) {
    const int bad = cholesky_factor(n, a, p);
    for (long i = 0; i < n; ++i) {
        a[i * n + i] = 1 / p[i];
        for (long j = i + 1; j < n; ++j) {
            double sum = 0;
            for (long q = i; q < j; ++q) sum -= a[j * n + q] * a[q * n + i];
            a[j * n + i] = (float) sum / p[j];
        }
    }
    return bad;
}

/* src/ALS.cpp:41-64: A^-1 = L^-T L^-1, mirrored to the lower triangle. */
int spd_inverse(long n, float* a, float* scratch_p) {
    const int bad = cholesky_lower_inverse(n, a, scratch_p);
    for (long i = 0; i < n; ++i)
        for (long j = i + 1; j < n; ++j) a[i * n + j] = 0.0f;
    for (long i = 0; i < n; ++i) {
        a[i * n + i] *= a[i * n + i];
        for (long q = i + 1; q < n; ++q) a[i * n + i] += a[q * n + i] * a[q * n + i];
        for (long j = i + 1; j < n; ++j)
            for (long q = j; q < n; ++q) a[i * n + j] += a[q * n + i] * a[q * n + j];
    }
    for (long i = 0; i < n; ++i)
        for (long j = 0; j < i; ++j) a[i * n + j] = a[j * n + i];
    return bad;
}

/* src/ALS.cpp:66-79 with the gather of src/ALS.cpp:115-118 folded in. */
void gramian(long cnt, const unsigned* idx, const float* X, long k, float* A) {
    for (long I = 0; I < k; ++I) {
        for (long J = I; J < k; ++J) {
            float sum = 0.0f;
            for (long q = 0; q < cnt; ++q) {
                const float* row = X + (long) idx[q] * k;
                sum += row[I] * row[J];
            }
            A[J * k + I] = sum;
            A[I * k + J] = sum;
        }
    }
}

void als_half(long nseg, const unsigned* ptr, const unsigned* idx, const float* val,
              const float* X, float* Y, long k, float lambda) {
#pragma omp parallel
    {
        std::vector<float> A(k * k), b(k), p(k);
#pragma omp for ORC_SCHED
        for (long s = 0; s < nseg; ++s) {
            float* y = Y + s * k;
            const unsigned lo = ptr[s], hi = ptr[s + 1];
            if (hi == lo) { /* src/ALS.cpp:151-157 */
                for (long c = 0; c < k; ++c) y[c] = 0.0f;
                continue;
            }
            gramian(hi - lo, idx + lo, X, k, A.data());
            for (long c = 0; c < k; ++c) A[c * k + c] = A[c * k + c] + lambda; /* plain lambda */
            spd_inverse(k, A.data(), p.data());
            for (long c = 0; c < k; ++c) { /* src/ALS.cpp:129-134 */
                float acc = 0;
                for (unsigned q = lo; q < hi; ++q) acc += val[q] * X[(long) idx[q] * k + c];
                b[c] = acc;
            }
            for (long c = 0; c < k; ++c) { /* src/ALS.cpp:137-142 */
                float acc = 0;
                for (long d = 0; d < k; ++d) acc += b[d] * A[c * k + d];
                y[c] = acc;
            }
        }
    }
}


/* ---------------------------------------------------------------------------------------------
 * The flags the reference parses and never reads (src/pmf.h:33-36: eps, do_predict, verbose,
 * do_nmf).  What is PINNED here: calrmse_r1 -- real code in the reference, src/tools.cpp:261-270,
 * with its intended call site left in a comment at src/CCD.cpp:141-148 (verbose && do_predict:
 * per-rank "rmse" from an incrementally updated test residual).  What is NOT pinned ("parity
 * unpinned": no code, test or fixture for it anywhere under /root/reference): the meaning of eps
 * and do_nmf.  The reference is a fork of LIBPMF 1.41's ccd-r1.cpp, where they are
 *   do_nmf : the new coordinate value is clamped at 0;
 *   eps    : per update fundec = h * (old - new)^2 (clamped update: -2 g old + h old^2) with
 *            h = lambda |Omega| + sum u^2; an inner iteration's fundec is summed over its v- and
 *            u-update; the inner loop of a rank stops when that sum drops below eps * fundec_max
 *            (the running maximum over the outer iteration, the very first inner iteration of
 *            the first rank of the first outer iteration excepted), a rank that stops in its
 *            first inner iteration counts as an early stop, and the rank loop of an outer
 *            iteration ends after five of them.
 * Restated from that published algorithm; arithmetic in the reference's fp32 with the fundec
 * sums in double. */
inline float rank_one_column_ext(const unsigned* ptr, const unsigned* idx, const float* val, long c, const float* vec,
                                 float lambda_scaled, float old, int do_nmf, double* fundec) {
    const unsigned lo = ptr[c], hi = ptr[c + 1];
    if (lo == hi) return 0.0f;
    float g = 0.0f, h = lambda_scaled;
    for (unsigned p = lo; p < hi; ++p) {
        const float x = vec[idx[p]];
        g += x * val[p];
        h += x * x;
    }
    float x = g / h;
    if (do_nmf && x < 0.0f) {
        x = 0.0f;
        *fundec += -2.0 * (double) g * (double) old + (double) h * (double) old * (double) old;
    } else {
        const double delta = (double) old - (double) x;
        *fundec += (double) h * delta * delta;
    }
    return x;
}

inline double sweep_ext(long ncols, const unsigned* ptr, const unsigned* idx, const float* val, const float* vec,
                        float lambda, float* out, int do_nmf) {
    double fundec = 0.0;
#pragma omp parallel for ORC_SCHED reduction(+ : fundec)
    for (long c = 0; c < ncols; ++c)
        out[c] = rank_one_column_ext(ptr, idx, val, c, vec, lambda * (ptr[c + 1] - ptr[c]), out[c], do_nmf, &fundec);
    return fundec;
}

/* src/tools.cpp:261-270 */
inline double calrmse_r1(long nnz_test, const unsigned* row, const unsigned* col, float* resid, const float* Wt,
                         const float* Ht, const float* oldWt, const float* oldHt) {
    double rmse = 0;
#pragma omp parallel for reduction(+ : rmse)
    for (long q = 0; q < nnz_test; ++q) {
        resid[q] -= Wt[row[q]] * Ht[col[q]] - oldWt[row[q]] * oldHt[col[q]];
        rmse += resid[q] * resid[q];
    }
    return sqrt(rmse / nnz_test);
}

} // namespace

extern "C" {

int orc_max_threads(void) { return omp_get_max_threads(); }

void orc_initial_col(float* X, long k, long n) {
    srand(0L);
    for (long i = 0; i < n; ++i)
        for (long j = 0; j < k; ++j) X[j * n + i] = 0.1f * (float(rand()) / RAND_MAX) + 0.001f;
}

void orc_rank_one_sweep(long ncols, const unsigned* col_ptr, const unsigned* row_idx,
                        const float* val, const float* u, float lambda, float* out, int threads) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    sweep(ncols, col_ptr, row_idx, val, u, lambda, out);
}

float orc_update_rating(long ncols, const unsigned* col_ptr, const unsigned* row_idx, float* val,
                        const float* Wt, const float* Ht, int add, int threads) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    return residual_update(ncols, col_ptr, row_idx, val, Wt, Ht, add != 0);
}

double orc_calrmse(long nnz_test, const unsigned* test_row, const unsigned* test_col,
                   const float* test_val, const float* W, const float* H, long m, long n, long k,
                   int ifALS) {
    double acc = 0;
    for (long q = 0; q < nnz_test; ++q) {
        double err = -test_val[q];
        err += predict(W, H, test_row[q], test_col[q], m, n, k, ifALS != 0);
        acc += err * err;
    }
    return sqrt(acc / nnz_test);
}

void orc_ccdr1(long m, long n, long nnz, const unsigned* csc_col_ptr, const unsigned* csc_row_idx,
               float* csc_val, const unsigned* csr_row_ptr, const unsigned* csr_col_idx,
               float* csr_val, float* W, float* H, long k, float lambda, int maxiter,
               int maxinneriter, int threads, long nnz_test, const unsigned* test_row,
               const unsigned* test_col, const float* test_val, double* rmse_out,
               double* times_out) {
    (void) nnz;
    omp_set_num_threads(threads > 0 ? threads : 1);
    memset(H, 0, sizeof(float) * (size_t) k * (size_t) n); /* src/CCD.cpp:55-60 */
    std::vector<float> u(m), v(n);

    for (int oiter = 1; oiter <= maxiter; ++oiter) {
        double rank_time = 0, update_time = 0;
        for (long t = 0; t < k; ++t) {
            float* Wt = W + t * m;
            float* Ht = H + t * n;
            double t0 = omp_get_wtime();
            memcpy(u.data(), Wt, sizeof(float) * m);
            memcpy(v.data(), Ht, sizeof(float) * n);
            if (oiter > 1) { /* add the rank back: src/CCD.cpp:100-103 */
                residual_update(n, csc_col_ptr, csc_row_idx, csc_val, Wt, Ht, true);
                residual_update(m, csr_row_ptr, csr_col_idx, csr_val, Ht, Wt, true);
            }
            update_time += omp_get_wtime() - t0;

            t0 = omp_get_wtime();
            for (int it = 1; it <= maxinneriter; ++it) { /* src/CCD.cpp:107-123 */
                sweep(n, csc_col_ptr, csc_row_idx, csc_val, u.data(), lambda, v.data());
                sweep(m, csr_row_ptr, csr_col_idx, csr_val, v.data(), lambda, u.data());
            }
            rank_time += omp_get_wtime() - t0;

            t0 = omp_get_wtime();
            memcpy(Wt, u.data(), sizeof(float) * m);
            memcpy(Ht, v.data(), sizeof(float) * n);
            residual_update(n, csc_col_ptr, csc_row_idx, csc_val, u.data(), v.data(), false);
            residual_update(m, csr_row_ptr, csr_col_idx, csr_val, v.data(), u.data(), false);
            update_time += omp_get_wtime() - t0;
        }
        if (times_out) {
            times_out[2 * (oiter - 1)] = rank_time;
            times_out[2 * (oiter - 1) + 1] = update_time;
        }
        if (rmse_out)
            rmse_out[oiter - 1] =
                nnz_test > 0 ? orc_calrmse(nnz_test, test_row, test_col, test_val, W, H, m, n, k, 0)
                             : 0.0;
    }
}


void orc_ccdr1_ext(long m, long n, const unsigned* csc_col_ptr, const unsigned* csc_row_idx, float* csc_val,
                   const unsigned* csr_row_ptr, const unsigned* csr_col_idx, float* csr_val, float* W, float* H, long k,
                   float lambda, int maxiter, int maxinneriter, int threads, long nnz_test, const unsigned* test_row,
                   const unsigned* test_col, const float* test_val, int do_nmf, float eps, double* rmse_out,
                   double* rank_rmse_out, int* ranks_done_out) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    memset(H, 0, sizeof(float) * (size_t) k * (size_t) n);
    std::vector<float> u(m), v(n), oldu(m), oldv(n), tres(test_val, test_val + nnz_test);
    for (int oiter = 1; oiter <= maxiter; ++oiter) {
        double fundec_max = 0;
        int early_stop = 0, done = 0;
        for (long t = 0; t < k; ++t) {
            if (eps > 0 && early_stop >= 5) break;
            float* Wt = W + t * m;
            float* Ht = H + t * n;
            memcpy(u.data(), Wt, sizeof(float) * m);
            memcpy(v.data(), Ht, sizeof(float) * n);
            oldu = u;
            oldv = v; /* zero in the first outer iteration: H starts at 0 */
            if (oiter > 1) {
                residual_update(n, csc_col_ptr, csc_row_idx, csc_val, Wt, Ht, true);
                residual_update(m, csr_row_ptr, csr_col_idx, csr_val, Ht, Wt, true);
            }
            for (int it = 1; it <= maxinneriter; ++it) {
                double cur = sweep_ext(n, csc_col_ptr, csc_row_idx, csc_val, u.data(), lambda, v.data(), do_nmf);
                cur += sweep_ext(m, csr_row_ptr, csr_col_idx, csr_val, v.data(), lambda, u.data(), do_nmf);
                if (eps > 0) {
                    if (cur < fundec_max * (double) eps) {
                        if (it == 1) ++early_stop;
                        break;
                    }
                    if (!(oiter == 1 && t == 0 && it == 1)) fundec_max = cur > fundec_max ? cur : fundec_max;
                }
            }
            memcpy(Wt, u.data(), sizeof(float) * m);
            memcpy(Ht, v.data(), sizeof(float) * n);
            residual_update(n, csc_col_ptr, csc_row_idx, csc_val, u.data(), v.data(), false);
            residual_update(m, csr_row_ptr, csr_col_idx, csr_val, v.data(), u.data(), false);
            if (rank_rmse_out && nnz_test > 0)
                rank_rmse_out[(size_t) (oiter - 1) * k + t] =
                    calrmse_r1(nnz_test, test_row, test_col, tres.data(), u.data(), v.data(), oldu.data(), oldv.data());
            ++done;
        }
        if (ranks_done_out) ranks_done_out[oiter - 1] = done;
        if (rmse_out)
            rmse_out[oiter - 1] = nnz_test > 0 ? orc_calrmse(nnz_test, test_row, test_col, test_val, W, H, m, n, k, 0) : 0.0;
    }
}

void orc_gramian(long cnt, const unsigned* idx, const float* X, long k, float* A) {
    gramian(cnt, idx, X, k, A);
}

int orc_chol_inverse(long k, float* A) {
    std::vector<float> p(k);
    return spd_inverse(k, A, p.data());
}

void orc_als_half(long nseg, const unsigned* ptr, const unsigned* idx, const float* val,
                  const float* X, float* Y, long k, float lambda, int threads) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    als_half(nseg, ptr, idx, val, X, Y, k, lambda);
}

void orc_als(long m, long n, long nnz, const unsigned* csc_col_ptr, const unsigned* csc_row_idx,
             const float* csc_val, const unsigned* csr_row_ptr, const unsigned* csr_col_idx,
             const float* csr_val, float* W, float* H, long k, float lambda, int maxiter,
             int threads, long nnz_test, const unsigned* test_row, const unsigned* test_col,
             const float* test_val, double* rmse_out, double* times_out) {
    (void) nnz;
    omp_set_num_threads(threads > 0 ? threads : 1);
    for (int it = 0; it < maxiter; ++it) {
        double t0 = omp_get_wtime();
        als_half(m, csr_row_ptr, csr_col_idx, csr_val, H, W, k, lambda); /* W over H */
        als_half(n, csc_col_ptr, csc_row_idx, csc_val, W, H, k, lambda); /* H over W */
        if (times_out) times_out[it] = omp_get_wtime() - t0;
        if (rmse_out)
            rmse_out[it] =
                nnz_test > 0 ? orc_calrmse(nnz_test, test_row, test_col, test_val, W, H, m, n, k, 1)
                             : 0.0;
    }
}

} // extern "C"

// tools/als_sensitivity.cpp -- experiment (not product, not a test): how far does the ALS test RMSE of the
// golden fixtures move when ONLY the rounding changes on the CPU (FMA-contracted sums; Cholesky solve instead
// of the explicit inverse)?  Output committed as profiles/r01_als_sensitivity.txt; it is the reason the tiny
// fixture is compared at 3e-4 instead of 1e-4 in tests/test_gpu_als.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../oracle/mf_oracle.h"
extern "C" {
}
static void gram(long cnt, const unsigned* idx, const float* X, long k, float* A, bool fma) {
    for (long I = 0; I < k; ++I) for (long J = I; J < k; ++J) {
        float s = 0; for (long q = 0; q < cnt; ++q) { const float* r = X + (long) idx[q] * k; s = fma ? fmaf(r[I], r[J], s) : s + r[I] * r[J]; }
        A[J * k + I] = s; A[I * k + J] = s; }
}
static void chol_solve(long n, float* a, const float* b, float* y) {  // my GPU tail
    std::vector<float> p(n), z(n);
    for (long i = 0; i < n; ++i) for (long j = i; j < n; ++j) { float s = a[j * n + i]; for (long q = i - 1; q >= 0; --q) s -= a[i * n + q] * a[j * n + q];
        if (i == j) p[i] = sqrtf(s); else a[j * n + i] = s / p[i]; }
    for (long i = 0; i < n; ++i) z[i] = b[i];
    for (long i = 0; i < n; ++i) { z[i] = z[i] / p[i]; for (long r = i + 1; r < n; ++r) z[r] -= a[r * n + i] * z[i]; }
    for (long i = n - 1; i >= 0; --i) { z[i] = z[i] / p[i]; for (long q = 0; q < i; ++q) z[q] -= a[i * n + q] * z[i]; }
    for (long i = 0; i < n; ++i) y[i] = z[i];
}
static void half(long nseg, const unsigned* ptr, const unsigned* idx, const float* val, const float* X, float* Y, long k, float lam, bool fma, bool solve) {
    std::vector<float> A(k * k), b(k);
    for (long s = 0; s < nseg; ++s) {
        float* y = Y + s * k; unsigned lo = ptr[s], hi = ptr[s + 1];
        if (hi == lo) { for (long c = 0; c < k; ++c) y[c] = 0; continue; }
        gram(hi - lo, idx + lo, X, k, A.data(), fma);
        for (long c = 0; c < k; ++c) A[c * k + c] += lam;
        for (long c = 0; c < k; ++c) { float acc = 0; for (unsigned q = lo; q < hi; ++q) acc = fma ? fmaf(val[q], X[(long) idx[q] * k + c], acc) : acc + val[q] * X[(long) idx[q] * k + c]; b[c] = acc; }
        if (solve) chol_solve(k, A.data(), b.data(), y);
        else { orc_chol_inverse(k, A.data()); for (long c = 0; c < k; ++c) { float acc = 0; for (long d = 0; d < k; ++d) acc += b[d] * A[c * k + d]; y[c] = acc; } }
    }
}
int main(int argc, char** argv) {
    // reads a flat dump written by the python side
    FILE* f = fopen(argv[1], "rb"); long hdr[6]; fread(hdr, 8, 6, f); long m = hdr[0], n = hdr[1], nnz = hdr[2], k = hdr[3], nt = hdr[4];
    std::vector<unsigned> rp(m + 1), ci(nnz), cp(n + 1), ri(nnz), tr(nt), tc(nt); std::vector<float> rv(nnz), cv(nnz), tv(nt), H0(n * k);
    fread(rp.data(), 4, m + 1, f); fread(ci.data(), 4, nnz, f); fread(rv.data(), 4, nnz, f); fread(cp.data(), 4, n + 1, f); fread(ri.data(), 4, nnz, f); fread(cv.data(), 4, nnz, f);
    fread(tr.data(), 4, nt, f); fread(tc.data(), 4, nt, f); fread(tv.data(), 4, nt, f); fread(H0.data(), 4, n * k, f); fclose(f);
    float lam = atof(argv[2]);
    for (int fma = 0; fma < 2; ++fma) for (int solve = 0; solve < 2; ++solve) {
        std::vector<float> W(m * k), H = H0;
        printf("fma=%d solve=%d :", fma, solve);
        for (int it = 0; it < 3; ++it) {
            half(m, rp.data(), ci.data(), rv.data(), H.data(), W.data(), k, lam, fma, solve);
            half(n, cp.data(), ri.data(), cv.data(), W.data(), H.data(), k, lam, fma, solve);
            printf(" %.6f", orc_calrmse(nt, tr.data(), tc.data(), tv.data(), W.data(), H.data(), m, n, k, 1));
        }
        printf("\n");
    }
}

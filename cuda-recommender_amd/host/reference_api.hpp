// reference_api.hpp -- the two entry points the reference's driver calls (src/main.cpp:11-17),
// with the reference's exact signatures (cuda_src/CCD_CUDA.h:49, cuda_src/ALS_CUDA.h:40), implemented
// over libmfx's C ABI.  This header replaces cuda_src/{CCD,ALS}_CUDA.{h,cu} in a build of the
// reference's driver (INTEGRATION.md shows the three-line change).  Like the reference wrappers
// they return void; a failure has already been reported on stderr ("CCD FAILED: ...").
#pragma once

#include <vector>

#include "mfx.h"
#include "pmf.hpp"

namespace mfx_shim {

inline mfx_csx view(const SparseMatrix& R) {
    mfx_csx x;
    x.rows = R.rows; x.cols = R.cols; x.nnz = R.nnz;
    x.csc_col_ptr = R.get_csc_col_ptr(); x.csc_row_idx = R.get_csc_row_indx(); x.csc_val = R.get_csc_val();
    x.csr_row_ptr = R.get_csr_row_ptr(); x.csr_col_idx = R.get_csr_col_indx(); x.csr_val = R.get_csr_val();
    return x;
}
inline mfx_coo view(const TestData& T) {
    mfx_coo t;
    t.nnz = T.nnz; t.row = T.getTestRow(); t.col = T.getTestCol(); t.val = T.getTestVal();
    return t;
}
inline mfx_params params_of(const parameter& p) {
    mfx_params q;
    mfx_params_default(&q);
    q.k = p.k; q.lambda = p.lambda; q.maxiter = p.maxiter; q.maxinneriter = p.maxinneriter;
    q.nBlocks = p.nBlocks; q.nThreadsPerBlock = p.nThreadsPerBlock;
    q.verbose = 1;  // the reference wrappers always print the per-iteration line
    q.device = p.device; q.schedule = p.schedule; q.kernel_variant = p.kernel_variant; q.panel_rows = p.panel_rows;
    return q;
}
// MatData (vector of vectors) <-> the flat layouts of mfx.h; 64-bit indexing (the reference's
// `int indexPosition` overflows at rows*k >= 2^31, cuda_src/CCD_CUDA.cu:255-261).
inline std::vector<float> flatten(const MatData& M) {
    std::vector<float> f;
    size_t n = 0;
    for (const VecData& r : M) n += r.size();
    f.reserve(n);
    for (const VecData& r : M) f.insert(f.end(), r.begin(), r.end());
    return f;
}
inline void unflatten(const std::vector<float>& f, MatData& M) {
    size_t o = 0;
    for (VecData& r : M) {
        for (float& x : r) x = f[o++];
    }
}

}  // namespace mfx_shim

inline void kernel_wrapper_ccdpp_NV(SparseMatrix& R, TestData& T, MatData& W, MatData& H, parameter& parameters) {
    std::vector<float> w = mfx_shim::flatten(W), h = mfx_shim::flatten(H);
    const mfx_csx r = mfx_shim::view(R);
    const mfx_coo t = mfx_shim::view(T);
    const mfx_params p = mfx_shim::params_of(parameters);
    if (mfx_ccdpp_run(&r, &t, w.data(), h.data(), &p, nullptr) != MFX_OK) return;  // already reported
    mfx_shim::unflatten(w, W);
    mfx_shim::unflatten(h, H);
}

inline void kernel_wrapper_als_NV(SparseMatrix& R, TestData& T, MatData& W, MatData& H, parameter& parameters) {
    std::vector<float> w = mfx_shim::flatten(W), h = mfx_shim::flatten(H);
    const mfx_csx r = mfx_shim::view(R);
    const mfx_coo t = mfx_shim::view(T);
    const mfx_params p = mfx_shim::params_of(parameters);
    if (mfx_als_run(&r, &t, w.data(), h.data(), &p, nullptr) != MFX_OK) return;
    mfx_shim::unflatten(w, W);
    mfx_shim::unflatten(h, H);
}

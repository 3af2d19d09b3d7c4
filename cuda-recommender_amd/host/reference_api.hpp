// reference_api.hpp -- the two entry points the reference's driver calls (src/main.cpp:11-17),
// with the reference's exact signatures (cuda_src/CCD_CUDA.h:49, cuda_src/ALS_CUDA.h:40), implemented
// over libmfx's C ABI.  This header replaces cuda_src/{CCD,ALS}_CUDA.{h,cu} in a build of the
// reference's driver (INTEGRATION.md shows the three-line change).  Like the reference wrappers
// they return void; a failure has already been reported on stderr ("CCD FAILED: ...").
#pragma once

#include <vector>

#include "mfx.h"
// Inside the reference's own tree define MFX_SHIM_EXTERNAL_TYPES and include its "pmf.h" first: the
// shim then uses the reference's SparseMatrix / TestData / MatData / parameter as they are
// (tests/test_host.py::test_shim_compiles_against_reference_headers does exactly that).
#ifndef MFX_SHIM_EXTERNAL_TYPES
#include "pmf.hpp"
#endif

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
#ifndef MFX_SHIM_EXTERNAL_TYPES  // knobs only this repository's `parameter` has
    q.device = p.device; q.schedule = p.schedule; q.kernel_variant = p.kernel_variant; q.panel_rows = p.panel_rows; q.layout_build = p.layout_build;
    if (p.libpmf_flags) {  // opt-in: the reference reads none of these (src/pmf.h:33-36)
        q.do_nmf = p.do_nmf; q.eps = p.eps; q.rank_trace = (p.verbose && p.do_predict) ? 1 : 0;
    }
#endif
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

// ---------------------------------------------------------------------------------------------
// Multi-GPU CCD++ from one process: one host thread per shard, each driving its own device and its
// own resident solver; shards are nnz-balanced user-row blocks (mfx_partition_rows /
// mfx_extract_shard) and exchange one all-reduce of the (g, h) column partials per inner iteration.
// With fewer devices than shards the shards share device 0 through the in-process loopback
// communicator (a rehearsal mode: correct, not fast).  Same in/out contract as the 1-GPU wrapper.
// ---------------------------------------------------------------------------------------------
#include <thread>

inline void kernel_wrapper_ccdpp_multi(SparseMatrix& R, TestData& T, MatData& W, MatData& H, parameter& parameters,
                                       int n_shards) {
    const mfx_csx r = mfx_shim::view(R);
    const int ndev = mfx_device_count();
    const bool loopback = ndev < n_shards;
    if (ndev < 1) { fprintf(stderr, "CCD FAILED: no usable HIP device\n"); return; }
    if (n_shards < 1) { fprintf(stderr, "CCD FAILED: %d shards requested\n", n_shards); return; }
    std::vector<int64_t> bounds((size_t) n_shards + 1);
    if (mfx_partition_rows(r.rows, r.csr_row_ptr, n_shards, bounds.data()) != MFX_OK) { fprintf(stderr, "CCD FAILED: %s\n", mfx_last_error()); return; }
    // every shard must own at least one row (a solver rejects rows == 0, and a rank that fails while the
    // others are inside a collective would leave them there): checked BEFORE any thread starts
    for (int g = 0; g < n_shards; ++g)
        if (bounds[g + 1] <= bounds[g]) {
            fprintf(stderr, "CCD FAILED: cannot cut %lld rows into %d non-empty nnz-balanced shards (shard %d would be empty)\n",
                    (long long) r.rows, n_shards, g);
            return;
        }
    std::vector<uint32_t> col_nnz((size_t) r.cols);
    for (int64_t c = 0; c < r.cols; ++c) col_nnz[c] = r.csc_col_ptr[c + 1] - r.csc_col_ptr[c];
    unsigned char uid[MFX_COMM_ID_BYTES];
    if (!loopback && mfx_comm_unique_id(uid) != MFX_OK) { fprintf(stderr, "CCD FAILED: %s\n", mfx_last_error()); return; }
    if (loopback) printf("[info] %d shards on %d device(s): loopback communicator (rehearsal mode)\n", n_shards, ndev);
    const unsigned k = parameters.k;
    std::vector<std::vector<float>> Wl(n_shards), Hl(n_shards);
    std::vector<int> status(n_shards, MFX_OK);
    std::vector<std::string> errors(n_shards);
    auto worker = [&](int g) {
        const int64_t lo = bounds[g], hi = bounds[g + 1], nr = hi - lo;
        const uint32_t lnnz = r.csr_row_ptr[hi] - r.csr_row_ptr[lo];
        std::vector<uint32_t> rp(nr + 1), ci(lnnz), cp(r.cols + 1), ri(lnnz);
        std::vector<float> rv(lnnz), cv(lnnz);
        int rc = mfx_extract_shard(&r, lo, hi, rp.data(), ci.data(), rv.data(), cp.data(), ri.data(), cv.data());
        std::vector<uint32_t> tr, tc;
        std::vector<float> tv;
        for (long q = 0; q < T.nnz; ++q)
            if (T.getTestRow()[q] >= lo && T.getTestRow()[q] < hi) {
                tr.push_back(T.getTestRow()[q] - (uint32_t) lo); tc.push_back(T.getTestCol()[q]); tv.push_back(T.getTestVal()[q]);
            }
        mfx_csx lr = {nr, r.cols, (int64_t) lnnz, cp.data(), ri.data(), cv.data(), rp.data(), ci.data(), rv.data()};
        mfx_coo lt = {(int64_t) tv.size(), tr.data(), tc.data(), tv.data()};
        mfx_params p = mfx_shim::params_of(parameters);
        p.device = loopback ? 0 : g;
        mfx_comm_t comm = nullptr;
        mfx_ccd_t s = nullptr;
        if (rc == MFX_OK) rc = loopback ? mfx_comm_create_local(&comm, 4242, g, n_shards, 0) : mfx_comm_create(&comm, uid, g, n_shards, g);
        mfx_shard sh = {comm, col_nnz.data(), (int64_t) T.nnz};
        if (rc == MFX_OK) rc = mfx_ccd_create(&s, &lr, &lt, &p, MFX_HOST, &sh);
        Wl[g].resize((size_t) k * nr); Hl[g].resize((size_t) k * r.cols);
        for (unsigned t = 0; t < k; ++t) for (int64_t i = 0; i < nr; ++i) Wl[g][(size_t) t * nr + i] = W[t][lo + i];
        if (rc == MFX_OK) rc = mfx_ccd_set_factors(s, Wl[g].data(), nullptr, MFX_HOST);
        if (rc != MFX_OK) errors[g] = mfx_last_error();
        // setup is over on this rank, for better or worse: tell the others before anyone enters a collective
        if (comm) {
            int worst = rc;
            const int arc = mfx_comm_agree(comm, rc, &worst);
            if (rc == MFX_OK && arc != MFX_OK) { rc = arc; errors[g] = mfx_last_error(); }
            if (rc == MFX_OK && worst != MFX_OK) { rc = worst; errors[g] = "another shard failed during setup"; }
        }
        if (rc == MFX_OK) {
            rc = mfx_ccd_iterate(s, parameters.maxiter, 1, nullptr);
            if (rc == MFX_OK) rc = mfx_ccd_get_factors(s, Wl[g].data(), Hl[g].data(), MFX_HOST);
            if (rc != MFX_OK) {  // failed in mid-flight: release the ranks that wait for this one
                errors[g] = mfx_last_error();
                mfx_comm_abort(comm);
            }
        }
        mfx_ccd_destroy(s);
        mfx_comm_destroy(comm);
        status[g] = rc;
    };
    std::vector<std::thread> th;
    for (int g = 0; g < n_shards; ++g) th.emplace_back(worker, g);
    for (auto& x : th) x.join();
    for (int g = 0; g < n_shards; ++g)
        if (status[g] != MFX_OK) { fprintf(stderr, "CCD FAILED: shard %d: %s\n", g, errors[g].c_str()); return; }
    for (int g = 0; g < n_shards; ++g) {
        const int64_t lo = bounds[g], nr = bounds[g + 1] - lo;
        for (unsigned t = 0; t < k; ++t) for (int64_t i = 0; i < nr; ++i) W[t][lo + i] = Wl[g][(size_t) t * nr + i];
    }
    for (unsigned t = 0; t < k; ++t) for (int64_t j = 0; j < r.cols; ++j) H[t][j] = Hl[0][(size_t) t * r.cols + j];  // replicas agree
}

// pmf.hpp -- host data model of the mfx_train driver: the same types, member names and getters as
// the reference's host side (reference: src/pmf.h:6-43 `parameter`/`solvertype`,
// src/pmf_util.h:29-30 `VecData`/`MatData`, :34-149 `SparseMatrix`, :151-211 `TestData`) so that
// code written against the reference's driver API compiles against this header unchanged.
// Own implementation: std::vector storage, bounds/IO errors reported instead of aborting.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define DTYPE float

using VecData = std::vector<DTYPE>;
using MatData = std::vector<VecData>;

enum class solvertype { CCD, ALS };

class parameter {
public:
    solvertype solver_type = solvertype::CCD;
    unsigned k = 10;
    int threads = 4;
    int maxiter = 5;
    int maxinneriter = 1;  // help text says 5, the code's default is 1 (src/pmf.h:31)
    float lambda = 0.1f;
    float eps = 1e-3f;
    int do_predict = 0;
    int verbose = 0;
    int do_nmf = 0;
    bool enable_cuda = false;  // -CUDA: run the GPU path (here: HIP on MI355X)
    bool enable_omp = false;
    unsigned nBlocks = 32;            // accepted, ignored by the HIP kernels
    unsigned nThreadsPerBlock = 256;  // accepted, ignored
    char src_dir[1024];
    // additions of this implementation (extra flags of mfx_train, see main.cpp)
    int device = 0;
    int schedule = 1;
    int kernel_variant = 1;
    int panel_rows = 0;
    int layout_build = 0; // mfx_params.layout_build
    int n_gpus = 1;  // -nGPUs: user-row-block shards, one per GPU (CCD++ only)
    int libpmf_flags = 0;  // -libpmf_flags 1: -e / -N / -p / -q take their LIBPMF meaning (the reference ignores them; so does the default)
    parameter() { snprintf(src_dir, sizeof(src_dir), "../data/simple"); }
};

class SparseMatrix {
public:
    long rows = 0, cols = 0, nnz = 0, max_row_nnz_ = 0, max_col_nnz_ = 0;

    void initialize_matrix(long rows_, long cols_, long nnz_);
    // Reads the six binary files of one dataset directory; false + message on failure.
    bool read_binary_file(const std::string& csr_row_ptr, const std::string& csr_col_indx, const std::string& csr_val,
                          const std::string& csc_col_ptr, const std::string& csc_row_indx, const std::string& csc_val,
                          std::string* err);
    SparseMatrix get_shallow_transpose() const;  // a copy with the roles of CSR and CSC swapped

    unsigned* get_csc_col_ptr() const { return const_cast<unsigned*>(csc_col_ptr_.data()); }
    unsigned* get_csc_row_indx() const { return const_cast<unsigned*>(csc_row_indx_.data()); }
    DTYPE* get_csc_val() const { return const_cast<DTYPE*>(csc_val_.data()); }
    unsigned* get_csr_col_indx() const { return const_cast<unsigned*>(csr_col_indx_.data()); }
    unsigned* get_csr_row_ptr() const { return const_cast<unsigned*>(csr_row_ptr_.data()); }
    DTYPE* get_csr_val() const { return const_cast<DTYPE*>(csr_val_.data()); }

private:
    std::vector<unsigned> csc_col_ptr_, csr_row_ptr_, csc_row_indx_, csr_col_indx_;
    std::vector<DTYPE> csr_val_, csc_val_;
};

class TestData {
public:
    long rows = 0, cols = 0, nnz = 0;
    bool read_binary_file(long rows_, long cols_, long nnz_, const std::string& fname_data,
                          const std::string& fname_row, const std::string& fname_col, std::string* err);
    unsigned* getTestCol() const { return const_cast<unsigned*>(test_col.data()); }
    unsigned* getTestRow() const { return const_cast<unsigned*>(test_row.data()); }
    DTYPE* getTestVal() const { return const_cast<DTYPE*>(test_val.data()); }

private:
    std::vector<unsigned> test_row, test_col;
    std::vector<DTYPE> test_val;
};

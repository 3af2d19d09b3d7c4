// tools.cpp -- host-side helpers of mfx_train (see tools.hpp for the reference functions they mirror).
#include "tools.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "mfx.h"

namespace {

template <typename T>
bool read_array(const std::string& path, std::vector<T>& out, size_t count, std::string* err) {
    out.resize(count);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { *err = "can't open " + path; return false; }
    const size_t got = count ? fread(out.data(), sizeof(T), count, f) : 0;
    fclose(f);
    if (got != count) { *err = "short read on " + path; return false; }
    return true;
}

// ptr files hold int32 (reference reads them one int at a time, src/pmf_util.h:119-129)
bool read_ptr(const std::string& path, std::vector<unsigned>& out, size_t count, long* max_seg, std::string* err) {
    std::vector<int32_t> raw;
    if (!read_array(path, raw, count, err)) return false;
    out.resize(count);
    *max_seg = 0;
    for (size_t i = 0; i < count; ++i) {
        if (raw[i] < 0 || (i && raw[i] < raw[i - 1])) { *err = "pointer file " + path + " is not monotone"; return false; }
        out[i] = (unsigned) raw[i];
        if (i && (long) (raw[i] - raw[i - 1]) > *max_seg) *max_seg = raw[i] - raw[i - 1];
    }
    return true;
}

[[noreturn]] void die(const std::string& msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    exit(EXIT_FAILURE);
}

}  // namespace

void SparseMatrix::initialize_matrix(long rows_, long cols_, long nnz_) {
    rows = rows_; cols = cols_; nnz = nnz_;
}

bool SparseMatrix::read_binary_file(const std::string& csr_row_ptr, const std::string& csr_col_indx,
                                    const std::string& csr_val, const std::string& csc_col_ptr,
                                    const std::string& csc_row_indx, const std::string& csc_val, std::string* err) {
    if (!read_ptr(csr_row_ptr, csr_row_ptr_, (size_t) rows + 1, &max_row_nnz_, err)) return false;
    if (!read_array(csr_col_indx, csr_col_indx_, (size_t) nnz, err)) return false;
    if (!read_array(csr_val, csr_val_, (size_t) nnz, err)) return false;
    if (!read_ptr(csc_col_ptr, csc_col_ptr_, (size_t) cols + 1, &max_col_nnz_, err)) return false;
    if (!read_array(csc_row_indx, csc_row_indx_, (size_t) nnz, err)) return false;
    if (!read_array(csc_val, csc_val_, (size_t) nnz, err)) return false;
    if (csr_row_ptr_.back() != (unsigned) nnz || csc_col_ptr_.back() != (unsigned) nnz) {
        *err = "pointer files do not end at nnz";
        return false;
    }
    return true;
}

SparseMatrix SparseMatrix::get_shallow_transpose() const {
    SparseMatrix t;
    t.rows = cols; t.cols = rows; t.nnz = nnz;
    t.max_row_nnz_ = max_col_nnz_; t.max_col_nnz_ = max_row_nnz_;
    t.csc_col_ptr_ = csr_row_ptr_; t.csc_row_indx_ = csr_col_indx_; t.csc_val_ = csr_val_;
    t.csr_row_ptr_ = csc_col_ptr_; t.csr_col_indx_ = csc_row_indx_; t.csr_val_ = csc_val_;
    return t;
}

bool TestData::read_binary_file(long rows_, long cols_, long nnz_, const std::string& fname_data,
                                const std::string& fname_row, const std::string& fname_col, std::string* err) {
    rows = rows_; cols = cols_; nnz = nnz_;
    return read_array(fname_data, test_val, (size_t) nnz, err) && read_array(fname_row, test_row, (size_t) nnz, err) &&
           read_array(fname_col, test_col, (size_t) nnz, err);
}

void load(const char* srcdir, SparseMatrix& R, TestData& T) {
    const std::string dir(srcdir);
    std::ifstream meta(dir + "/meta_modified_all");
    if (!meta) { printf("Can't open meta input file.\n"); exit(EXIT_FAILURE); }  // same message as the reference
    long m, n, nnz;
    std::string coo[3], csr[3], csc[3], test[3];
    unsigned long nnz_test;
    if (!(meta >> m >> n >> nnz >> coo[0] >> coo[1] >> coo[2] >> csr[0] >> csr[1] >> csr[2] >> csc[0] >> csc[1] >> csc[2] >>
          nnz_test >> test[0] >> test[1] >> test[2]))
        die("meta_modified_all: expected 'm n nnz', 3+3+3 file names, 'nnz_test', 3 file names");
    auto t0 = std::chrono::high_resolution_clock::now();
    R.initialize_matrix(m, n, nnz);
    std::string err;
    if (!R.read_binary_file(dir + "/" + csr[0], dir + "/" + csr[1], dir + "/" + csr[2], dir + "/" + csc[0],
                            dir + "/" + csc[1], dir + "/" + csc[2], &err))
        die("load: " + err);
    auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "[info] Train TIMER: " << std::chrono::duration<double>(t1 - t0).count() << "s.\n";
    if (!T.read_binary_file(m, n, (long) nnz_test, dir + "/" + test[0], dir + "/" + test[1], dir + "/" + test[2], &err))
        die("load: " + err);
    auto t2 = std::chrono::high_resolution_clock::now();
    std::cout << "[info] Tests TIMER: " << std::chrono::duration<double>(t2 - t1).count() << "s.\n";
}

void initial_col(MatData& X, long k, long n) {
    std::vector<float> flat((size_t) k * n);
    mfx_initial_col(flat.data(), k, n);
    X.assign(k, VecData(n));
    for (long j = 0; j < k; ++j) memcpy(X[j].data(), flat.data() + (size_t) j * n, sizeof(float) * n);
}

double calculate_rmse_directly(MatData& W, MatData& H, TestData& T, int rank, bool ifALS) {
    auto t0 = std::chrono::high_resolution_clock::now();
    if (T.nnz == 0) exit(EXIT_FAILURE);  // the reference exits when there is nothing to score
    double acc = 0;
    for (long q = 0; q < T.nnz; ++q) {
        const long i = T.getTestRow()[q], j = T.getTestCol()[q];
        double pred = 0;
        for (int t = 0; t < rank; ++t) pred += ifALS ? W[i][t] * H[j][t] : W[t][i] * H[t][j];
        const double d = pred - (double) T.getTestVal()[q];
        acc += d * d;
    }
    const double rmse = std::sqrt(acc / (double) T.nnz);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("Test RMSE = %lf. Calculated in %lfs\n", rmse, std::chrono::duration<double>(t1 - t0).count());
    return rmse;
}

unsigned golden_compare(const MatData& W, const MatData& W_ref, unsigned k, unsigned m) {
    unsigned errors = 0;
    for (unsigned i = 0; i < k; ++i)
        for (unsigned j = 0; j < m; ++j)
            if (std::fabs((double) W[i][j] - (double) W_ref[i][j]) > 0.1 * std::fabs((double) W_ref[i][j])) ++errors;
    if (errors == 0) {
        std::cout << "Check... PASS!" << std::endl;
    } else {
        const unsigned entries = k * m;
        printf("Check... NO PASS! [%.4f%%] #Error = %u out of %u entries.\n", 100.0 * errors / entries, errors, entries);
    }
    return errors;
}

void exit_with_help() {
    printf("Usage: omp-pmf-train [options] data_dir [model_filename]\n"
           "options:\n"
           "    -k rank : set the rank (default 10)\n"
           "    -n threads : set the number of threads (default 4)\n"
           "    -l lambda : set the regularization parameter lambda (default 0.1)\n"
           "    -t max_iter: set the number of iterations (default 5)\n"
           "    -T max_inner_iter: set the number of inner iterations used in CCDR1 (default 5)\n"
           "    -e epsilon : set inner termination criterion epsilon of CCDR1 (default 1e-3)\n"
           "    -p do_predict: do prediction or not (default 0)\n"
           "    -q verbose: show information or not (default 0)\n"
           "    -N do_nmf: do nmf (default 0)\n"
           "    -CUDA: Flag to enable CUDA\n"
           "    -nBlocks: Number of blocks on CUDA (default 32)\n"
           "    -nThreadsPerBlock: Number of threads per block on CUDA (default 256)\n"
           "    -ALS: Flag to enable ALS algorithm, if not present CCD++ is used\n"
           "  additions of this build (the GPU path is HIP on MI355X; -CUDA keeps its name for drop-in use):\n"
           "    -device id : HIP device ordinal (default 0)\n"
           "    -schedule s : 1 fused passes (default), 0 one launch per reference kernel\n"
           "    -panel rows : LDS panel size, 0 auto, -1 off\n"
           "    -nGPUs n : CCD++ over n user-row-block shards, one GPU each (RCCL all-reduce per inner iteration)\n"
           "    -save file : write W then H in the reference's model format (save_mat_t); with -OMP alone: the reference-order leg's\n"
           "    -libpmf_flags 1 : honour -e, -N and -p/-q with their LIBPMF 1.41 meaning (CCD++, one GPU): function-decrease\n"
           "                      stopping rule, non-negative factors, per-rank test RMSE lines.  Default 0: parsed and\n"
           "                      ignored, like the reference does\n");
    exit(EXIT_FAILURE);
}

parameter parse_command_line(int argc, char** argv) {
    parameter param;
    int i;
    for (i = 1; i < argc; i++) {
        if (argv[i][0] != '-') break;
        if (++i >= argc) exit_with_help();  // every dashed token pre-consumes the next one
        const char* flag = argv[i - 1];
        if (!strcmp(flag, "-nBlocks")) param.nBlocks = atoi(argv[i]);
        else if (!strcmp(flag, "-nThreadsPerBlock")) param.nThreadsPerBlock = atoi(argv[i]);
        else if (!strcmp(flag, "-device")) param.device = atoi(argv[i]);
        else if (!strcmp(flag, "-schedule")) param.schedule = atoi(argv[i]);
        else if (!strcmp(flag, "-panel")) param.panel_rows = atoi(argv[i]);
        else if (!strcmp(flag, "-layout_build")) param.layout_build = atoi(argv[i]);
        else if (!strcmp(flag, "-nGPUs")) param.n_gpus = atoi(argv[i]);
        else if (!strcmp(flag, "-libpmf_flags")) param.libpmf_flags = atoi(argv[i]);
        else if (!strcmp(flag, "-save")) { /* handled by main (it rescans argv) */ }
        else if (!strcmp(flag, "-CUDA") || !strcmp(flag, "-HIP")) { param.enable_cuda = true; --i; }  // valueless: give it back
        else if (!strcmp(flag, "-OMP")) { param.enable_omp = true; --i; }
        else if (!strcmp(flag, "-ALS")) { param.solver_type = solvertype::ALS; --i; }
        else {
            switch (flag[1]) {
                case 'k': param.k = atoi(argv[i]); break;
                case 'n': param.threads = atoi(argv[i]); break;
                case 'l': param.lambda = (float) atof(argv[i]); break;
                case 't': param.maxiter = atoi(argv[i]); break;
                case 'T': param.maxinneriter = atoi(argv[i]); break;
                case 'e': param.eps = (float) atof(argv[i]); break;
                case 'p': param.do_predict = atoi(argv[i]); break;
                case 'q': param.verbose = atoi(argv[i]); break;
                case 'N': param.do_nmf = (atoi(argv[i]) == 1); break;
                default:
                    fprintf(stderr, "unknown option: -%c\n", flag[1]);
                    exit_with_help();
            }
        }
    }
    if (param.do_predict != 0) param.verbose = 1;
    if (i >= argc) exit_with_help();
    snprintf(param.src_dir, sizeof(param.src_dir), "%s", argv[i]);
    return param;
}

void save_mat_t(const MatData& A, FILE* fp, bool row_major) {
    if (!fp) die("output stream is not valid.");
    const long m = row_major ? (long) A.size() : (long) A[0].size();
    const long n = row_major ? (long) A[0].size() : (long) A.size();
    fwrite(&m, sizeof(long), 1, fp);
    fwrite(&n, sizeof(long), 1, fp);
    std::vector<float> buf((size_t) m * n);
    for (long i = 0; i < m; ++i)
        for (long j = 0; j < n; ++j) buf[(size_t) i * n + j] = row_major ? A[i][j] : A[j][i];
    fwrite(buf.data(), sizeof(float), buf.size(), fp);
}

MatData load_mat_t(FILE* fp, bool row_major) {
    if (!fp) die("input stream is not valid.");
    long m = 0, n = 0;
    if (fread(&m, sizeof(long), 1, fp) != 1 || fread(&n, sizeof(long), 1, fp) != 1 || m <= 0 || n <= 0)
        die("model file: bad header");
    std::vector<float> buf((size_t) m * n);
    if (fread(buf.data(), sizeof(float), buf.size(), fp) != buf.size()) die("model file: short read");
    MatData A = row_major ? MatData(m, VecData(n)) : MatData(n, VecData(m));
    for (long i = 0; i < m; ++i)
        for (long j = 0; j < n; ++j) (row_major ? A[i][j] : A[j][i]) = buf[(size_t) i * n + j];
    return A;
}

double calculate_rmse_from_file(FILE* model_fp, FILE* test_fp, FILE* output_fp) {
    auto t0 = std::chrono::high_resolution_clock::now();
    rewind(model_fp);
    MatData W = load_mat_t(model_fp, true);
    MatData H = load_mat_t(model_fp, true);
    const size_t rank = W[0].size();
    if (rank == 0 || H[0].size() != rank) die("Matrix is empty!");
    int i, j;
    double v, acc = 0;
    size_t n = 0;
    while (fscanf(test_fp, "%d %d %lf", &i, &j, &v) == 3) {
        if (i < 1 || j < 1 || (size_t) i > W.size() || (size_t) j > H.size()) die("test file: index out of range (indices are 1-based)");
        double pred = 0;
        for (size_t t = 0; t < rank; ++t) pred += W[i - 1][t] * H[j - 1][t];
        acc += (pred - v) * (pred - v);
        ++n;
        if (output_fp) fprintf(output_fp, "%lf\n", pred);
    }
    if (n == 0) exit(EXIT_FAILURE);
    const double rmse = std::sqrt(acc / (double) n);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("[FINAL INFO] Test RMSE = %f. Calculated in %lfs\n", rmse, std::chrono::duration<double>(t1 - t0).count());
    return rmse;
}

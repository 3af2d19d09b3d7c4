/*
 * ref_driver.cpp -- CUDA-free driver around the REFERENCE's own CPU sources.
 * TEST INFRASTRUCTURE ONLY; builds into oracle/_ref/ref_cpu (git-ignored).
 *
 * The reference's src/main.cpp cannot be compiled here (it pulls <cuda.h> in through
 * cuda_src/ALS_CUDA.h:7), so this file is the ~100-line replacement SURVEY.md §8(c) calls
 * for.  Nothing from the reference is copied: its CCD.cpp / ALS.cpp are #included from
 * where they lie under /root/reference/src (so their file-local inline functions can be
 * called for single-step vectors), tools.cpp / extras.cpp are compiled as separate objects
 * by oracle/Makefile.  Output: raw little-endian dumps that oracle/make_fixtures.py packs
 * into tests/golden/*.npz.
 *
 * usage: ref_cpu <ccd|als|steps> <data_dir> <out_dir> k lambda maxiter maxinner threads
 */
#include "CCD.cpp" /* reference: src/CCD.cpp (whole file, in place) */
#include "ALS.cpp" /* reference: src/ALS.cpp (whole file, in place) */

#include <string>

static void dump_f32(const std::string& path, const float* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    fwrite(p, sizeof(float), n, f);
    fclose(f);
}

/* The model file as the reference would write it: its OWN save_mat_t (src/tools.cpp:90-118), called the way its
 * driver's (commented-out) call site does -- save_mat_t(W, model_fp, ifALS); save_mat_t(H, model_fp, ifALS);
 * (src/main.cpp:146-147) -- on the factors the reference's solver just produced.  Pins the byte format of -save / -predict. */
static void dump_model(const std::string& path, const MatData& W, const MatData& H, bool ifALS) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    save_mat_t(W, f, ifALS);
    save_mat_t(H, f, ifALS);
    fclose(f);
}

static void dump_mat(const std::string& path, const MatData& M) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    for (const VecData& row : M) fwrite(row.data(), sizeof(float), row.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 9) {
        fprintf(stderr, "usage: %s <ccd|als|steps> data_dir out_dir k lambda maxiter maxinner threads\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1], out = std::string(argv[3]) + "/";
    parameter param;
    snprintf(param.src_dir, sizeof(param.src_dir), "%s", argv[2]);
    param.k = (unsigned) atoi(argv[4]);
    param.lambda = (float) atof(argv[5]);
    param.maxiter = atoi(argv[6]);
    param.maxinneriter = atoi(argv[7]);
    param.threads = atoi(argv[8]);

    SparseMatrix R;
    TestData T;
    load(param.src_dir, R, T);
    const unsigned k = param.k;
    MatData W, H;

    if (mode == "ccd") {
        initial_col(W, k, R.rows);
        initial_col(H, k, R.cols);
        dump_mat(out + "W0.bin", W);
        dump_mat(out + "H0.bin", H);
        ccdr1_OMP(R, W, H, T, param);
        dump_mat(out + "W.bin", W);
        dump_mat(out + "H.bin", H);
        dump_f32(out + "csc_val_final.bin", R.get_csc_val(), R.nnz);
        dump_f32(out + "csr_val_final.bin", R.get_csr_val(), R.nnz);
        dump_model(out + "model.bin", W, H, false);
        calculate_rmse_directly(W, H, T, k, false);
    } else if (mode == "als") {
        initial_col(W, R.rows, k);
        initial_col(H, R.cols, k);
        dump_mat(out + "W0.bin", W);
        dump_mat(out + "H0.bin", H);
        ALS_OMP(R, W, H, T, param);
        dump_mat(out + "W.bin", W);
        dump_mat(out + "H.bin", H);
        dump_model(out + "model.bin", W, H, true);
        calculate_rmse_directly(W, H, T, k, true);
    } else if (mode == "steps") {
        /* ---- CCD single steps, from the CCD-layout initial factors ---- */
        initial_col(W, k, R.rows);
        initial_col(H, k, R.cols);
        SparseMatrix Rt = R.get_shallow_transpose();
        VecData u = W[0], v(R.cols);
        for (long c = 0; c < R.cols; ++c)
            v[c] = RankOneUpdate_Original_float(R, c, u, param.lambda * (R.get_csc_col_ptr()[c + 1] - R.get_csc_col_ptr()[c]));
        dump_f32(out + "step_v1.bin", v.data(), v.size());
        for (long r = 0; r < Rt.cols; ++r)
            u[r] = RankOneUpdate_Original_float(Rt, r, v, param.lambda * (Rt.get_csc_col_ptr()[r + 1] - Rt.get_csc_col_ptr()[r]));
        dump_f32(out + "step_u1.bin", u.data(), u.size());
        UpdateRating_Original_float(R, u, v, false);
        UpdateRating_Original_float(Rt, v, u, false);
        dump_f32(out + "step_csc_sub.bin", R.get_csc_val(), R.nnz);
        dump_f32(out + "step_csr_sub.bin", R.get_csr_val(), R.nnz);
        UpdateRating_Original_float(R, W[1 % k], H[1 % k], true);
        UpdateRating_Original_float(Rt, H[1 % k], W[1 % k], true);
        dump_f32(out + "step_csc_add.bin", R.get_csc_val(), R.nnz);
        dump_f32(out + "step_csr_add.bin", R.get_csr_val(), R.nnz);
        double r0 = calrmse(T, W, H, false, true);
        FILE* f = fopen((out + "step_rmse_init_ccd.txt").c_str(), "w");
        fprintf(f, "%.17g\n", r0);
        fclose(f);

        /* ---- ALS single steps, from the ALS-layout initial factors ---- */
        MatData Ha;
        initial_col(Ha, R.cols, k);
        /* first row with at least one rating */
        long row = 0;
        while (row < R.rows && R.get_csr_row_ptr()[row + 1] == R.get_csr_row_ptr()[row]) ++row;
        const unsigned lo = R.get_csr_row_ptr()[row], hi = R.get_csr_row_ptr()[row + 1];
        std::vector<float*> gathered(hi - lo), A(k);
        std::vector<float> Abuf((size_t) k * k);
        for (unsigned q = lo; q < hi; ++q) gathered[q - lo] = Ha[R.get_csr_col_indx()[q]].data();
        for (unsigned i = 0; i < k; ++i) A[i] = &Abuf[(size_t) i * k];
        Mt_byM_multiply((int) (hi - lo), (int) k, gathered.data(), A.data());
        dump_f32(out + "step_gram.bin", Abuf.data(), Abuf.size());
        for (unsigned c = 0; c < k; ++c) A[c][c] = A[c][c] + param.lambda;
        inverseMatrix_CholeskyMethod((int) k, A.data());
        dump_f32(out + "step_inv.bin", Abuf.data(), Abuf.size());
        f = fopen((out + "step_als_row.txt").c_str(), "w");
        fprintf(f, "%ld\n", row);
        fclose(f);
    } else {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    return 0;
}

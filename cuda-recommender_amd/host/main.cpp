// main.cpp -- mfx_train: the reference's driver flow (reference: src/main.cpp:38-173) on top of
// libmfx.  Same flags, same log lines, same validation calls; the GPU path is this repository's
// HIP implementation behind kernel_wrapper_{ccdpp,als}_NV.  The reference's -OMP leg (its CPU
// solver) is not part of the product: the CPU restatement lives under oracle/ as a test oracle.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "reference_api.hpp"
#include "tools.hpp"

static void runCUDA(SparseMatrix& R, TestData& T, MatData& W, MatData& H, parameter& parameters, bool ALS) {
    if (ALS) kernel_wrapper_als_NV(R, T, W, H, parameters);
    else if (parameters.n_gpus > 1) kernel_wrapper_ccdpp_multi(R, T, W, H, parameters, parameters.n_gpus);
    else kernel_wrapper_ccdpp_NV(R, T, W, H, parameters);
}

// mfx_train -predict <model> <test.txt> <output>: the file-based scoring step the reference stubs
// out (src/main.cpp:146-149); the model must have been written row-major (rows x k), i.e. by an
// ALS run with -save, or converted.
static int predict_mode(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: mfx_train -predict model_file test_file output_file\n"); return EXIT_FAILURE; }
    FILE* model = fopen(argv[2], "rb");
    FILE* test = fopen(argv[3], "r");
    if (!model) { fprintf(stderr, "can't open model file %s\n", argv[2]); return EXIT_FAILURE; }
    if (!test) { fprintf(stderr, "can't open test file %s\n", argv[3]); return EXIT_FAILURE; }
    FILE* out = fopen(argv[4], "w");
    if (!out) { fprintf(stderr, "can't open output file %s\n", argv[4]); return EXIT_FAILURE; }
    calculate_rmse_from_file(model, test, out);
    fclose(model); fclose(test); fclose(out);
    return EXIT_SUCCESS;
}

int main(int argc, char* argv[]) {
    if (argc > 1 && !strcmp(argv[1], "-predict")) return predict_mode(argc, argv);
    auto t_start = std::chrono::high_resolution_clock::now();
    parameter param = parse_command_line(argc, argv);
    const char* save_path = nullptr;
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], "-save")) save_path = argv[i + 1];

    SparseMatrix R;
    TestData T;
    puts("------------------------------------------------------------");
    puts("[info] Loading R matrix...");
    auto t0 = std::chrono::high_resolution_clock::now();
    load(param.src_dir, R, T);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("[info] Loading rating data time: %lf s.\n", std::chrono::duration<double>(t1 - t0).count());
    puts("------------------------------------------------------------");

    const bool ifALS = param.solver_type == solvertype::ALS;
    puts(ifALS ? "[info] Picked Version: ALS!" : "[info] Picked Version: CCD!");
    MatData W_cuda, H_cuda, W_ref, H_ref;
    if (ifALS) {
        initial_col(W_cuda, R.rows, param.k); initial_col(H_cuda, R.cols, param.k);
        initial_col(W_ref, R.rows, param.k);  initial_col(H_ref, R.cols, param.k);
    } else {
        initial_col(W_cuda, param.k, R.rows); initial_col(H_cuda, param.k, R.cols);
        initial_col(W_ref, param.k, R.rows);  initial_col(H_ref, param.k, R.cols);
    }
    printf("[info] ThreadsPerBlock = %u | Blocks = %u | K = %u | InnerIter = %d | OuterIter = %d | Threads = %d | L = %.3f\n",
           param.nThreadsPerBlock, param.nBlocks, param.k, param.maxinneriter, param.maxiter, param.threads, param.lambda);

    if (param.enable_cuda) {
        puts("------------------------------------------------------------");
        puts("[INFO] Computing with CUDA...");
        auto t5 = std::chrono::high_resolution_clock::now();
        runCUDA(R, T, W_cuda, H_cuda, param, ifALS);
        auto t6 = std::chrono::high_resolution_clock::now();
        printf("[info] CUDA Training time: %lf s.\n", std::chrono::duration<double>(t6 - t5).count());
        puts("------------------------------------------------------------");
        calculate_rmse_directly(W_cuda, H_cuda, T, param.k, ifALS);
    }
    if (param.enable_omp) {
        puts("------------------------------------------------------------");
        puts("[info] -OMP: the CPU solver is not built into mfx_train (it exists as the test oracle under oracle/).");
    }
    std::cout << "[info] validate the results." << std::endl;
    if (ifALS) {
        golden_compare(W_cuda, W_ref, R.rows, param.k);
        golden_compare(H_cuda, H_ref, R.cols, param.k);
    } else {
        golden_compare(W_cuda, W_ref, param.k, R.rows);
        golden_compare(H_cuda, H_ref, param.k, R.cols);
    }
    if (save_path) {  // the step the reference stubs out (src/main.cpp:146-147)
        FILE* fp = fopen(save_path, "wb");
        if (!fp) { fprintf(stderr, "can't open model file %s\n", save_path); return EXIT_FAILURE; }
        save_mat_t(W_cuda, fp, ifALS);
        save_mat_t(H_cuda, fp, ifALS);
        fclose(fp);
        printf("[info] model written to %s\n", save_path);
    }
    puts("------------------------------------------------------------");
    auto t_end = std::chrono::high_resolution_clock::now();
    std::cout << "Total Time: " << std::chrono::duration<double>(t_end - t_start).count() << " s.\n";
    return EXIT_SUCCESS;
}

// main.cpp -- mfx_train, the command-line driver over libmfx.
//
// Speaks the reference driver's protocol (reference: src/main.cpp:38-173): the same flags, the same
// log lines in the same order, the same post-run checks -- so that scripts written against the
// reference (scripts/times.sh and friends) keep working -- with the GPU leg served by this repository's
// HIP implementation behind kernel_wrapper_{ccdpp,als}_NV.  Two steps the reference only stubs out are
// real here: -save <file> (model dump) and -predict (file-based scoring).  The reference's -OMP leg is
// its CPU solver, which is not part of the product (the CPU restatement lives under oracle/ as a test
// oracle): here -OMP runs the library's REFERENCE-ORDER parity modes on the GPU -- for CCD++ the sweeps that
// add every column strictly left to right in fp32 (csrc/ccd_reforder.hip), for ALS the reference's own
// operation order (csrc/als_exact.hip); both reproduce src/CCD.cpp / src/ALS.cpp bit for bit -- from the same
// initial factors, so that the driver's closing golden_compare checks the fast path against the reference's
// arithmetic exactly as the reference's -CUDA -OMP run checks its GPU path against its CPU path.  The leg says
// what it is in the log and does NOT print its time under the OMP name: a script that divides "OMP Training
// time" by "CUDA Training time" must not mistake a GPU/GPU ratio for a CPU speed-up.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "reference_api.hpp"
#include "tools.hpp"

namespace {

struct Stopwatch {
    std::chrono::high_resolution_clock::time_point origin = std::chrono::high_resolution_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - origin).count(); }
};

void rule() { puts("------------------------------------------------------------"); }

// Factor pair in the solver's layout: CCD++ keeps k rows of length rows/cols, ALS rows/cols rows of
// length k; both start from the reference's initial_col stream.
struct Factors {
    MatData W, H;
    Factors(bool als, unsigned k, long rows, long cols) {
        if (als) { initial_col(W, rows, k); initial_col(H, cols, k); }
        else { initial_col(W, k, rows); initial_col(H, k, cols); }
    }
};

void solve_on_gpu(SparseMatrix& R, TestData& T, Factors& f, parameter& prm, bool als) {
    if (als) kernel_wrapper_als_NV(R, T, f.W, f.H, prm);
    else if (prm.n_gpus > 1) kernel_wrapper_ccdpp_multi(R, T, f.W, f.H, prm, prm.n_gpus);
    else kernel_wrapper_ccdpp_NV(R, T, f.W, f.H, prm);
}

// -predict <model> <test.txt> <output>: file-based scoring; the model must be row-major (rows x k),
// i.e. written by an ALS run with -save, or converted.
int predict_from_files(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: mfx_train -predict model_file test_file output_file\n"); return EXIT_FAILURE; }
    const char* names[3] = {argv[2], argv[3], argv[4]};
    const char* modes[3] = {"rb", "r", "w"};
    const char* what[3] = {"model", "test", "output"};
    FILE* fp[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        fp[i] = fopen(names[i], modes[i]);
        if (!fp[i]) {
            fprintf(stderr, "can't open %s file %s\n", what[i], names[i]);
            for (int j = 0; j < i; ++j) fclose(fp[j]);
            return EXIT_FAILURE;
        }
    }
    calculate_rmse_from_file(fp[0], fp[1], fp[2]);
    for (FILE* f : fp) fclose(f);
    return EXIT_SUCCESS;
}

const char* option_value(int argc, char** argv, const char* flag) {
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], flag)) return argv[i + 1];
    return nullptr;
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc > 1 && !strcmp(argv[1], "-predict")) return predict_from_files(argc, argv);
    const Stopwatch whole_run;
    parameter prm = parse_command_line(argc, argv);
    const char* model_path = option_value(argc, argv, "-save");

    SparseMatrix R;
    TestData T;
    rule();
    puts("[info] Loading R matrix...");
    {
        const Stopwatch sw;
        load(prm.src_dir, R, T);
        printf("[info] Loading rating data time: %lf s.\n", sw.seconds());
    }
    rule();

    const bool als = prm.solver_type == solvertype::ALS;
    puts(als ? "[info] Picked Version: ALS!" : "[info] Picked Version: CCD!");
    Factors solved(als, prm.k, R.rows, R.cols), untouched(als, prm.k, R.rows, R.cols);
    printf("[info] ThreadsPerBlock = %u | Blocks = %u | K = %u | InnerIter = %d | OuterIter = %d | Threads = %d | L = %.3f\n",
           prm.nThreadsPerBlock, prm.nBlocks, prm.k, prm.maxinneriter, prm.maxiter, prm.threads, prm.lambda);

    if (prm.enable_cuda) {
        rule();
        puts("[INFO] Computing with CUDA...");
        const Stopwatch sw;
        solve_on_gpu(R, T, solved, prm, als);
        printf("[info] CUDA Training time: %lf s.\n", sw.seconds());
        rule();
        calculate_rmse_directly(solved.W, solved.H, T, prm.k, als);
    }
    if (prm.enable_omp) {
        rule();
        puts("[INFO] -OMP: this build has no CPU solver; computing the reference-order leg on the GPU "
             "(same arithmetic and summation order as the reference's CPU solver, bit-identical factors)...");
        parameter second = prm;
        second.schedule = 0;         // ALS: als_exact.hip; CCD++: one launch per reference kernel ...
        second.kernel_variant = -1;  // ... with the sweeps in the reference's summation order (ccd_reforder.hip)
        second.n_gpus = 1;           // a sharded sum has no reference order
        if (second.libpmf_flags) {   // -e / -N change the arithmetic: the reference ignores them, and so does its stand-in leg
            puts("[INFO] -OMP: the reference ignores -e / -N / -p / -q (src/pmf.h:33-36); the reference-order leg runs without -libpmf_flags.");
            second.libpmf_flags = 0;
        }
        const Stopwatch sw;
        solve_on_gpu(R, T, untouched, second, als);
        printf("[info] reference-order GPU leg training time: %lf s.\n", sw.seconds());
        rule();
        calculate_rmse_directly(untouched.W, untouched.H, T, prm.k, als);
    }

    // The reference compares the -CUDA factors with the -OMP leg's; a leg that did not run leaves its
    // factors at the initial values, and -- exactly like the reference run with one flag only -- the check
    // then reports a mismatch.
    std::cout << "[info] validate the results." << std::endl;
    const long w_outer = als ? R.rows : (long) prm.k, w_inner = als ? (long) prm.k : R.rows;
    const long h_outer = als ? R.cols : (long) prm.k, h_inner = als ? (long) prm.k : R.cols;
    golden_compare(solved.W, untouched.W, w_outer, w_inner);
    golden_compare(solved.H, untouched.H, h_outer, h_inner);

    if (model_path) {
        FILE* fp = fopen(model_path, "wb");
        if (!fp) { fprintf(stderr, "can't open model file %s\n", model_path); return EXIT_FAILURE; }
        // the -CUDA leg's factors (src/main.cpp:146-147 saves W_cuda / H_cuda); with -OMP alone, the reference-order leg's
        const Factors& out = prm.enable_cuda || !prm.enable_omp ? solved : untouched;
        save_mat_t(out.W, fp, als);
        save_mat_t(out.H, fp, als);
        fclose(fp);
        printf("[info] model written to %s\n", model_path);
    }
    rule();
    std::cout << "Total Time: " << whole_run.seconds() << " s.\n";
    return EXIT_SUCCESS;
}

// tools.hpp -- loader / init / reporting helpers of the driver (reference: src/tools.h, src/extras.h).
#pragma once

#include "pmf.hpp"

// reference: load (src/tools.cpp:3-85): reads <dir>/meta_modified_all and the nine binaries.
void load(const char* srcdir, SparseMatrix& R, TestData& T);
// reference: initial_col (src/tools.cpp:165-173), via the C ABI's mfx_initial_col.
void initial_col(MatData& X, long k, long n);
// reference: calculate_rmse_directly (src/extras.cpp:182-216); returns the RMSE it prints.
double calculate_rmse_directly(MatData& W, MatData& H, TestData& T, int rank, bool ifALS);
// reference: golden_compare (src/extras.cpp:218-238); returns the error count it prints.
unsigned golden_compare(const MatData& W, const MatData& W_ref, unsigned k, unsigned m);
// reference: parse_command_line / exit_with_help (src/extras.cpp:46-141)
parameter parse_command_line(int argc, char** argv);
void exit_with_help();
// reference: save_mat_t / load_mat_t model format (src/tools.cpp:90-153): [long m][long n][m*n f32 row-major]
void save_mat_t(const MatData& A, FILE* fp, bool row_major = true);
MatData load_mat_t(FILE* fp, bool row_major = true);
// reference: calculate_rmse_from_file (src/extras.cpp:143-180): model = W then H (row-major, rows x k),
// test file = text "i j v" with 1-BASED indices, one prediction per line written to output_fp.
double calculate_rmse_from_file(FILE* model_fp, FILE* test_fp, FILE* output_fp);

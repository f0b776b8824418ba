// matching.hpp -- maximum-product matching + scaling pre-permutation (host, part of the "initialize" phase).
#pragma once
#include <cstdint>
#include <vector>

namespace hipmf {

// true when some row's diagonal entry is missing, zero or smaller than threshold * (largest magnitude of the row)
// (pairs: rows / columns 2 k, 2 k + 1 are the real and imaginary parts of complex row / column k -- either part of the complex diagonal counts)
bool diagonal_is_weak(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, double threshold, bool pairs = false);

// n x n matrix in CSR.  On success (0): mrow[j] = row matched to column j (the row-permuted matrix B with
// B(j, :) = A(mrow[j], :) has the matched entries on its diagonal), and scalings dr (rows of A), dc (columns) such that
// |dr_i a_ij dc_j| <= 1 with equality on the matched entries.  -1: structurally singular (no perfect matching).
int32_t max_product_matching(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, std::vector<int32_t> &mrow,
                             std::vector<double> &dr, std::vector<double> &dc);

// The same for the real-equivalent form (order n = 2 x complex order) of a complex matrix: the matching runs on the moduli of the complex
// entries, a matched pair of rows moves together (mrow[2 k] even, mrow[2 k + 1] = mrow[2 k] + 1) and shares its scalings.
int32_t paired_matching(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, std::vector<int32_t> &mrow, std::vector<double> &dr,
                        std::vector<double> &dc);

} // namespace hipmf

// host_api.cpp -- implementation of russell_host.hpp and its flat C API (include/russell_host.h).
// Plain C++ (g++): this library loads without a GPU; the HIP backend (librussell_hipmf.so) is opened with
// dlopen when the first SolverHIPMF is allocated.  There is no CPU solver here: without the HIP library and
// a device, LinSolver::new fails with "HIPMF solver is not available" (cf. lin_solver.rs:125,132,137-138).
#include "russell_host.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <sstream>

#include "../../../include/russell_hipmf.h"

namespace russell {

// ---- enums ------------------------------------------------------------------------------------------
const char *genie_to_string(Genie g) {
    switch (g) {
    case Genie::Hipmf: return "hipmf";
    case Genie::Umfpack: return "umfpack";
    case Genie::Mumps: return "mumps";
    case Genie::Cudss: return "cudss";
    }
    return "hipmf";
}

Genie genie_from(const std::string &name) {
    std::string s = name;
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    if (s == "umfpack") return Genie::Umfpack;
    if (s == "mumps") return Genie::Mumps;
    if (s == "cudss") return Genie::Cudss;
    return Genie::Hipmf;
}

Sym genie_get_sym(Genie g, bool symmetric) {
    if (!symmetric) return Sym::No;
    return g == Genie::Umfpack ? Sym::YesFull : Sym::YesLower;
}

static const char *sym_name(Sym s) {
    switch (s) {
    case Sym::No: return "No";
    case Sym::YesFull: return "YesFull";
    case Sym::YesLower: return "YesLower";
    case Sym::YesUpper: return "YesUpper";
    }
    return "No";
}
static const char *ORDERING_NAMES[] = {"Amd", "Amf", "Auto", "Best", "BtfColamd", "Cholmod", "Colamd", "Metis", "No", "Pord", "Qamd", "Scotch"};
static const char *SCALING_NAMES[] = {"Auto", "Column", "Diagonal", "Max", "No", "RowCol", "RowColIter", "RowColRig", "Sum"};

// ---- COO ---------------------------------------------------------------------------------------------
StrError CooMatrix::create(CooMatrix &out, size_t nrow, size_t ncol, size_t max_nnz, Sym symmetric) {
    if (nrow < 1) return "nrow must be ≥ 1";
    if (ncol < 1) return "ncol must be ≥ 1";
    if (max_nnz < 1) return "max_nnz must be ≥ 1";
    if (symmetric != Sym::No && nrow != ncol) return "symmetric storage requires a square matrix";
    out.symmetric = symmetric;
    out.nrow = nrow, out.ncol = ncol, out.nnz = 0, out.max_nnz = max_nnz;
    out.indices_i.assign(max_nnz, 0);
    out.indices_j.assign(max_nnz, 0);
    out.values.assign(max_nnz, 0.0);
    return nullptr;
}

StrError CooMatrix::put(size_t i, size_t j, double aij) {
    if (i >= nrow) return "COO matrix: index of row is outside range";
    if (j >= ncol) return "COO matrix: index of column is outside range";
    if (nnz >= max_nnz) return "COO matrix: max number of items has been reached";
    if (symmetric == Sym::YesLower && j > i) return "COO matrix: j > i is incorrect for lower triangular storage";
    if (symmetric == Sym::YesUpper && j < i) return "COO matrix: j < i is incorrect for upper triangular storage";
    indices_i[nnz] = (int32_t)i;
    indices_j[nnz] = (int32_t)j;
    values[nnz] = aij;
    nnz++;
    return nullptr;
}

static inline bool triangular(Sym s) { return s == Sym::YesLower || s == Sym::YesUpper; }

StrError CooMatrix::mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const {
    if (u.size() < ncol) return "u.dim() must be ≥ the number of columns of the matrix";
    if (v.size() < nrow) return "v.dim() must be ≥ the number of rows of the matrix";
    std::fill(v.begin(), v.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t p = 0; p < nnz; p++) {
        size_t i = (size_t)indices_i[p], j = (size_t)indices_j[p];
        v[i] += alpha * values[p] * u[j];
        if (mirror && i != j) v[j] += alpha * values[p] * u[i];
    }
    return nullptr;
}

StrError CooMatrix::from(CooMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> row_indices, std::vector<int32_t> col_indices,
                         std::vector<double> values, Sym symmetric) {
    if (nrow < 1) return "nrow must be ≥ 1";
    if (ncol < 1) return "ncol must be ≥ 1";
    const size_t nnz = row_indices.size();
    if (nnz < 1) return "nnz must be ≥ 1";
    if (col_indices.size() != nnz) return "col_indices.len() must be = nnz";
    if (values.size() != nnz) return "values.len() must be = nnz";
    for (size_t k = 0; k < nnz; k++) {
        if (row_indices[k] < 0 || (size_t)row_indices[k] >= nrow) return "row index is out-of-range";
        if (col_indices[k] < 0 || (size_t)col_indices[k] >= ncol) return "col index is out-of-range";
    }
    out.symmetric = symmetric;
    out.nrow = nrow, out.ncol = ncol, out.nnz = nnz, out.max_nnz = nnz;
    out.indices_i = std::move(row_indices), out.indices_j = std::move(col_indices), out.values = std::move(values);
    return nullptr;
}

StrError CooMatrix::mat_vec_mul_update(std::vector<double> &v, double alpha, const std::vector<double> &u) const {
    if (u.size() < ncol) return "u.dim() must be ≥ the number of columns of the matrix";
    if (v.size() < nrow) return "v.dim() must be ≥ the number of rows of the matrix";
    const bool mirror = triangular(symmetric);
    for (size_t p = 0; p < nnz; p++) {
        size_t i = (size_t)indices_i[p], j = (size_t)indices_j[p];
        v[i] += alpha * values[p] * u[j];
        if (mirror && i != j) v[j] += alpha * values[p] * u[i];
    }
    return nullptr;
}

StrError CooMatrix::mat_t_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const {
    if (u.size() < nrow) return "u.dim() must be ≥ the number of rows of the matrix";
    if (v.size() < ncol) return "v.dim() must be ≥ the number of columns of the matrix";
    std::fill(v.begin(), v.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t p = 0; p < nnz; p++) {
        size_t j = (size_t)indices_i[p], i = (size_t)indices_j[p]; // transposed roles
        v[i] += alpha * values[p] * u[j];
        if (mirror && i != j) v[j] += alpha * values[p] * u[i];
    }
    return nullptr;
}

StrError CooMatrix::assign(double alpha, const CooMatrix &other) {
    if (other.nrow != nrow) return "matrices must have the same nrow";
    if (other.ncol != ncol) return "matrices must have the same ncol";
    if (other.symmetric != symmetric) return "matrices must have the same symmetric type";
    reset();
    for (size_t p = 0; p < other.nnz; p++)
        if (StrError e = put((size_t)other.indices_i[p], (size_t)other.indices_j[p], alpha * other.values[p])) return e;
    return nullptr;
}

StrError CooMatrix::add(double alpha, const CooMatrix &other) {
    if (other.nrow > nrow) return "other.nrow must be ≤ this.nrow";
    if (other.ncol > ncol) return "other.ncol must be ≤ this.ncol";
    if (other.symmetric != symmetric) return "matrices must have the same symmetric type";
    for (size_t p = 0; p < other.nnz; p++)
        if (StrError e = put((size_t)other.indices_i[p], (size_t)other.indices_j[p], alpha * other.values[p])) return e;
    return nullptr;
}

StrError CooMatrix::put_lagrange_block(const CooMatrix &bb) {
    if (bb.symmetric != Sym::No) return "the Lagrange block must not be symmetric";
    if (bb.ncol + bb.nrow > nrow) return "ncol(B) + nrow(B) must be ≤ nrow(A)";
    if (bb.ncol + bb.nrow > ncol) return "ncol(B) + nrow(B) must be ≤ ncol(A)";
    for (size_t p = 0; p < bb.nnz; p++) {
        const size_t i = (size_t)bb.indices_i[p], j = (size_t)bb.indices_j[p];
        const double x = bb.values[p];
        StrError e = nullptr;
        if (symmetric == Sym::YesLower) e = put(bb.ncol + i, j, x); // B only
        else if (symmetric == Sym::YesUpper) e = put(j, bb.ncol + i, x); // B^T only
        else {
            e = put(bb.ncol + i, j, x);
            if (!e) e = put(j, bb.ncol + i, x);
        }
        if (e) return e;
    }
    return nullptr;
}

StrError CooMatrix::to_dense(std::vector<double> &a) const {
    if (a.size() != nrow * ncol) return "wrong matrix dimensions";
    std::fill(a.begin(), a.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t p = 0; p < nnz; p++) {
        size_t i = (size_t)indices_i[p], j = (size_t)indices_j[p];
        a[i * ncol + j] += values[p];
        if (mirror && i != j) a[j * ncol + i] += values[p];
    }
    return nullptr;
}

size_t CooMatrix::get_actual_nnz() const {
    if (!triangular(symmetric)) return nnz;
    size_t actual = 0;
    for (size_t p = 0; p < nnz; p++) actual += indices_i[p] != indices_j[p] ? 2 : 1;
    return actual;
}

// ---- validated constructors and dense copies of the compressed forms (csc_matrix.rs:197-262,702-729; csr_matrix.rs:193-257,676-703) --
static StrError check_compressed(size_t nmajor, size_t nminor, size_t nrow, size_t ncol, const std::vector<int32_t> &ptr,
                                 const std::vector<int32_t> &idx, const std::vector<double> &values, Sym symmetric, bool csc) {
    if (nrow < 1) return "nrow must be ≥ 1";
    if (ncol < 1) return "ncol must be ≥ 1";
    if (ptr.size() != nmajor + 1) return csc ? "col_pointers.len() must be = ncol + 1" : "row_pointers.len() must be = nrow + 1";
    const int32_t nnz = ptr[nmajor];
    if (nnz < 1) return csc ? "nnz = col_pointers[ncol] must be ≥ 1" : "nnz = row_pointers[nrow] must be ≥ 1";
    if (idx.size() < (size_t)nnz) return csc ? "row_indices.len() must be ≥ nnz" : "col_indices.len() must be ≥ nnz";
    if (values.size() < (size_t)nnz) return "values.len() must be ≥ nnz";
    if (symmetric != Sym::No && nrow != ncol) return "symmetric storage requires a square matrix";
    for (size_t j = 0; j < nmajor; j++) {
        if (ptr[j] < 0) return csc ? "col pointers must be ≥ 0" : "row pointers must be ≥ 0";
        if (ptr[j] > ptr[j + 1]) return csc ? "col pointers must be sorted in ascending order" : "row pointers must be sorted in ascending order";
        for (int32_t p = ptr[j]; p < ptr[j + 1]; p++) {
            if (idx[(size_t)p] < 0) return csc ? "row indices must be ≥ 0" : "column indices must be ≥ 0";
            if ((size_t)idx[(size_t)p] >= nminor) return csc ? "row indices must be < nrow" : "column indices must be < ncol";
            if (p > ptr[j] && idx[(size_t)p - 1] > idx[(size_t)p])
                return csc ? "row indices must be sorted in ascending order (within their column)"
                           : "column indices must be sorted in ascending order (within their row)";
        }
    }
    return nullptr;
}

StrError CscMatrix::create(CscMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> col_pointers, std::vector<int32_t> row_indices,
                           std::vector<double> values, Sym symmetric) {
    if (StrError e = check_compressed(ncol, nrow, nrow, ncol, col_pointers, row_indices, values, symmetric, true)) return e;
    out = CscMatrix();
    out.symmetric = symmetric, out.nrow = nrow, out.ncol = ncol;
    out.col_pointers = std::move(col_pointers), out.row_indices = std::move(row_indices), out.values = std::move(values);
    return nullptr;
}

StrError CsrMatrix::create(CsrMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> row_pointers, std::vector<int32_t> col_indices,
                           std::vector<double> values, Sym symmetric) {
    if (StrError e = check_compressed(nrow, ncol, nrow, ncol, row_pointers, col_indices, values, symmetric, false)) return e;
    out = CsrMatrix();
    out.symmetric = symmetric, out.nrow = nrow, out.ncol = ncol;
    out.row_pointers = std::move(row_pointers), out.col_indices = std::move(col_indices), out.values = std::move(values);
    return nullptr;
}

StrError CscMatrix::to_dense(std::vector<double> &a) const {
    if (a.size() != nrow * ncol) return "wrong matrix dimensions";
    std::fill(a.begin(), a.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t j = 0; j < ncol; j++)
        for (int32_t p = col_pointers[j]; p < col_pointers[j + 1]; p++) {
            const size_t i = (size_t)row_indices[(size_t)p];
            a[i * ncol + j] += values[(size_t)p];
            if (mirror && i != j) a[j * ncol + i] += values[(size_t)p];
        }
    return nullptr;
}

StrError CsrMatrix::to_dense(std::vector<double> &a) const {
    if (a.size() != nrow * ncol) return "wrong matrix dimensions";
    std::fill(a.begin(), a.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t i = 0; i < nrow; i++)
        for (int32_t p = row_pointers[i]; p < row_pointers[i + 1]; p++) {
            const size_t j = (size_t)col_indices[(size_t)p];
            a[i * ncol + j] += values[(size_t)p];
            if (mirror && i != j) a[j * ncol + i] += values[(size_t)p];
        }
    return nullptr;
}

// ---- CSC <-> CSR (csc_matrix.rs:508-584, csr_matrix.rs:483-558): counting transposition, entries of a column (row) come out in
// ascending row (column) order because the source is swept row by row (column by column) ----------------------------------
StrError CscMatrix::from_csr(CscMatrix &out, const CsrMatrix &csr) {
    const size_t nnz = csr.nnz_final();
    out.symmetric = csr.symmetric;
    out.nrow = csr.nrow, out.ncol = csr.ncol;
    out.col_pointers.assign(csr.ncol + 1, 0);
    out.row_indices.assign(nnz, 0);
    out.values.assign(nnz, 0.0);
    out.temp_w.clear();
    for (size_t p = 0; p < nnz; p++) out.col_pointers[(size_t)csr.col_indices[p] + 1]++;
    for (size_t j = 0; j < csr.ncol; j++) out.col_pointers[j + 1] += out.col_pointers[j];
    std::vector<int32_t> next(out.col_pointers.begin(), out.col_pointers.end() - 1);
    for (size_t i = 0; i < csr.nrow; i++)
        for (int32_t p = csr.row_pointers[i]; p < csr.row_pointers[i + 1]; p++) {
            const int32_t dest = next[(size_t)csr.col_indices[p]]++;
            out.row_indices[(size_t)dest] = (int32_t)i;
            out.values[(size_t)dest] = csr.values[(size_t)p];
        }
    return nullptr;
}

StrError CsrMatrix::from_csc(CsrMatrix &out, const CscMatrix &csc) {
    const size_t nnz = csc.nnz_final();
    out.symmetric = csc.symmetric;
    out.nrow = csc.nrow, out.ncol = csc.ncol;
    out.row_pointers.assign(csc.nrow + 1, 0);
    out.col_indices.assign(nnz, 0);
    out.values.assign(nnz, 0.0);
    out.temp_w.clear();
    for (size_t p = 0; p < nnz; p++) out.row_pointers[(size_t)csc.row_indices[p] + 1]++;
    for (size_t i = 0; i < csc.nrow; i++) out.row_pointers[i + 1] += out.row_pointers[i];
    std::vector<int32_t> next(out.row_pointers.begin(), out.row_pointers.end() - 1);
    for (size_t j = 0; j < csc.ncol; j++)
        for (int32_t p = csc.col_pointers[j]; p < csc.col_pointers[j + 1]; p++) {
            const int32_t dest = next[(size_t)csc.row_indices[p]]++;
            out.col_indices[(size_t)dest] = (int32_t)j;
            out.values[(size_t)dest] = csc.values[(size_t)p];
        }
    return nullptr;
}

// ---- CSC (csc_matrix.rs:337-505) -------------------------------------------------------------------------
StrError CscMatrix::from_coo(CscMatrix &out, const CooMatrix &coo) {
    if (coo.nnz < 1) return "COO to CSC requires nnz > 0";
    out.symmetric = coo.symmetric;
    out.nrow = coo.nrow, out.ncol = coo.ncol;
    out.col_pointers.assign(coo.ncol + 1, 0);
    out.row_indices.assign(coo.nnz, 0);
    out.values.assign(coo.nnz, 0.0);
    out.temp_w.clear();
    return out.update_from_coo(coo);
}

StrError CscMatrix::update_from_coo(const CooMatrix &coo) {
    if (coo.symmetric != symmetric) return "coo.symmetric must be equal to csc.symmetric";
    if (coo.nrow != nrow) return "coo.nrow must be equal to csc.nrow";
    if (coo.ncol != ncol) return "coo.ncol must be equal to csc.ncol";
    if (coo.nnz != values.size()) return "coo.nnz must be equal to nnz(dup) = csc.row_indices.len() = csc.values.len()";
    const size_t nnz = coo.nnz, ndim = std::max(nrow, ncol);
    if (temp_w.empty()) {
        temp_rp.assign(nrow + 1, 0);
        temp_rj.assign(nnz, 0);
        temp_rx.assign(nnz, 0.0);
        temp_rc.assign(nrow, 0);
        temp_w.assign(ndim, 0);
    } else {
        std::fill(temp_w.begin(), temp_w.begin() + nrow, 0);
    }
    auto &rp = temp_rp;
    auto &rj = temp_rj;
    auto &rx = temp_rx;
    auto &rc = temp_rc;
    auto &w = temp_w;
    for (size_t k = 0; k < nnz; k++) w[coo.indices_i[k]]++;
    rp[0] = 0;
    for (size_t i = 0; i < nrow; i++) {
        rp[i + 1] = rp[i] + w[i];
        w[i] = rp[i];
    }
    for (size_t k = 0; k < nnz; k++) {
        size_t p = (size_t)w[coo.indices_i[k]]++;
        rj[p] = coo.indices_j[k];
        rx[p] = coo.values[k];
    }
    for (size_t j = 0; j < ncol; j++) w[j] = -1;
    for (size_t i = 0; i < nrow; i++) {
        size_t p1 = (size_t)rp[i], p2 = (size_t)rp[i + 1], dest = p1;
        for (size_t p = p1; p < p2; p++) {
            size_t j = (size_t)rj[p];
            if (w[j] >= (int32_t)p1) {
                rx[(size_t)w[j]] += rx[p];
            } else {
                w[j] = (int32_t)dest;
                if (dest != p) {
                    rj[dest] = (int32_t)j;
                    rx[dest] = rx[p];
                }
                dest++;
            }
        }
        rc[i] = dest - p1;
    }
    for (size_t j = 0; j < ncol; j++) w[j] = 0;
    for (size_t i = 0; i < nrow; i++)
        for (size_t p = (size_t)rp[i]; p < (size_t)rp[i] + rc[i]; p++) w[rj[p]]++;
    col_pointers[0] = 0;
    for (size_t j = 0; j < ncol; j++) col_pointers[j + 1] = col_pointers[j] + w[j];
    for (size_t j = 0; j < ncol; j++) w[j] = col_pointers[j];
    for (size_t i = 0; i < nrow; i++)
        for (size_t p = (size_t)rp[i]; p < (size_t)rp[i] + rc[i]; p++) {
            size_t cp = (size_t)w[rj[p]]++;
            row_indices[cp] = (int32_t)i;
            values[cp] = rx[p];
        }
    return nullptr;
}

StrError CscMatrix::mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const {
    if (u.size() < ncol) return "u.dim() must be ≥ the number of columns of the matrix";
    if (v.size() < nrow) return "v.dim() must be ≥ the number of rows of the matrix";
    std::fill(v.begin(), v.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t j = 0; j < ncol; j++)
        for (int32_t p = col_pointers[j]; p < col_pointers[j + 1]; p++) {
            size_t i = (size_t)row_indices[p];
            v[i] += alpha * values[p] * u[j];
            if (mirror && i != j) v[j] += alpha * values[p] * u[i];
        }
    return nullptr;
}

// ---- CSR (csr_matrix.rs:332-480) -------------------------------------------------------------------------
StrError CsrMatrix::from_coo(CsrMatrix &out, const CooMatrix &coo) {
    if (coo.nnz < 1) return "COO to CSR requires nnz > 0";
    out.symmetric = coo.symmetric;
    out.nrow = coo.nrow, out.ncol = coo.ncol;
    out.row_pointers.assign(coo.nrow + 1, 0);
    out.col_indices.assign(coo.nnz, 0);
    out.values.assign(coo.nnz, 0.0);
    out.temp_w.clear();
    return out.update_from_coo(coo);
}

StrError CsrMatrix::update_from_coo(const CooMatrix &coo) {
    if (coo.symmetric != symmetric) return "coo.symmetric must be equal to csr.symmetric";
    if (coo.nrow != nrow) return "coo.nrow must be equal to csr.nrow";
    if (coo.ncol != ncol) return "coo.ncol must be equal to csr.ncol";
    if (coo.nnz != values.size()) return "coo.nnz must be equal to nnz(dup) = self.col_indices.len() = csr.values.len()";
    const size_t nnz = coo.nnz, ndim = std::max(nrow, ncol);
    if (temp_w.empty()) {
        temp_rp.assign(nrow + 1, 0);
        temp_rjx.assign(nnz, std::make_pair(0, 0.0));
        temp_rc.assign(nrow, 0);
        temp_w.assign(ndim, 0);
    } else {
        std::fill(temp_w.begin(), temp_w.begin() + nrow, 0);
    }
    auto &rp = temp_rp;
    auto &rjx = temp_rjx;
    auto &rc = temp_rc;
    auto &w = temp_w;
    for (size_t k = 0; k < nnz; k++) w[coo.indices_i[k]]++;
    rp[0] = 0;
    for (size_t i = 0; i < nrow; i++) {
        rp[i + 1] = rp[i] + w[i];
        w[i] = rp[i];
    }
    for (size_t k = 0; k < nnz; k++) {
        size_t p = (size_t)w[coo.indices_i[k]]++;
        rjx[p].first = coo.indices_j[k];
        rjx[p].second = coo.values[k];
    }
    for (size_t j = 0; j < ncol; j++) w[j] = -1;
    for (size_t i = 0; i < nrow; i++) {
        size_t p1 = (size_t)rp[i], p2 = (size_t)rp[i + 1], dest = p1;
        for (size_t p = p1; p < p2; p++) {
            size_t j = (size_t)rjx[p].first;
            if (w[j] >= (int32_t)p1) {
                rjx[(size_t)w[j]].second += rjx[p].second;
            } else {
                w[j] = (int32_t)dest;
                if (dest != p) rjx[dest] = rjx[p];
                dest++;
            }
        }
        rc[i] = dest - p1;
    }
    row_pointers[0] = 0;
    for (size_t i = 0; i < nrow; i++) row_pointers[i + 1] = row_pointers[i] + (int32_t)rc[i];
    size_t k = 0;
    for (size_t i = 0; i < nrow; i++) {
        size_t p1 = (size_t)rp[i], p2 = p1 + rc[i];
        std::stable_sort(rjx.begin() + p1, rjx.begin() + p2, [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
        for (size_t p = p1; p < p2; p++) {
            col_indices[k] = rjx[p].first;
            values[k] = rjx[p].second;
            k++;
        }
    }
    return nullptr;
}

StrError CsrMatrix::mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const {
    if (u.size() < ncol) return "u.dim() must be ≥ the number of columns of the matrix";
    if (v.size() < nrow) return "v.dim() must be ≥ the number of rows of the matrix";
    std::fill(v.begin(), v.end(), 0.0);
    const bool mirror = triangular(symmetric);
    for (size_t i = 0; i < nrow; i++)
        for (int32_t p = row_pointers[i]; p < row_pointers[i + 1]; p++) {
            size_t j = (size_t)col_indices[p];
            v[i] += alpha * values[p] * u[j];
            if (mirror && i != j) v[j] += alpha * values[p] * u[i];
        }
    return nullptr;
}

// ---- verify (verify_lin_sys.rs:60-96) -------------------------------------------------------------------
StrError VerifyLinSys::from(VerifyLinSys &out, const CooMatrix &mat, const std::vector<double> &x, const std::vector<double> &rhs) {
    if (x.size() != mat.ncol) return "x.dim() must be equal to ncol";
    if (rhs.size() != mat.nrow) return "rhs.dim() must be equal to nrow";
    if (mat.nnz < 1) return "matrix is empty";
    double max_abs_a = 0.0;
    for (size_t p = 0; p < mat.nnz; p++) max_abs_a = std::max(max_abs_a, std::fabs(mat.values[p]));
    std::vector<double> ax(mat.nrow, 0.0);
    mat.mat_vec_mul(ax, 1.0, x);
    double max_abs_ax = 0.0, max_abs_diff = 0.0;
    for (size_t i = 0; i < mat.nrow; i++) {
        max_abs_ax = std::max(max_abs_ax, std::fabs(ax[i]));
        max_abs_diff = std::max(max_abs_diff, std::fabs(ax[i] - rhs[i]));
    }
    out.max_abs_a = max_abs_a;
    out.max_abs_ax = max_abs_ax;
    out.max_abs_diff = max_abs_diff;
    out.relative_error = max_abs_diff / (max_abs_a + 1.0);
    return nullptr;
}

StrError VerifyLinSys::from_complex(VerifyLinSys &out, const ComplexCooMatrix &mat, const std::vector<double> &x, const std::vector<double> &rhs) {
    if (x.size() != 2 * mat.ncol) return "x.dim() must be equal to ncol";
    if (rhs.size() != 2 * mat.nrow) return "rhs.dim() must be equal to nrow";
    if (mat.nnz < 1) return "matrix is empty";
    double max_abs_a = 0.0;
    for (size_t p = 0; p < mat.nnz; p++) max_abs_a = std::max(max_abs_a, std::hypot(mat.values[2 * p], mat.values[2 * p + 1]));
    std::vector<double> ax(2 * mat.nrow, 0.0);
    mat.mat_vec_mul(ax, 1.0, 0.0, x);
    double max_abs_ax = 0.0, max_abs_diff = 0.0;
    for (size_t i = 0; i < mat.nrow; i++) {
        max_abs_ax = std::max(max_abs_ax, std::hypot(ax[2 * i], ax[2 * i + 1]));
        max_abs_diff = std::max(max_abs_diff, std::hypot(ax[2 * i] - rhs[2 * i], ax[2 * i + 1] - rhs[2 * i + 1]));
    }
    out.max_abs_a = max_abs_a;
    out.max_abs_ax = max_abs_ax;
    out.max_abs_diff = max_abs_diff;
    out.relative_error = max_abs_diff / (max_abs_a + 1.0);
    return nullptr;
}

// ---- stats -------------------------------------------------------------------------------------------
// shortest decimal form that reads back to the same double (what Rust's `{}` prints for an f64)
static std::string shortest(double v) {
    char num[64];
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(num, sizeof num, "%.*g", prec, v);
        if (strtod(num, nullptr) == v) break;
    }
    std::string t(num);
    // %g may switch to exponent form; the values printed here (< 1000 of a unit, or seconds < 60) never need it,
    // except tiny fractions, which are re-printed in fixed form
    if (t.find('e') != std::string::npos) {
        snprintf(num, sizeof num, "%.12f", v);
        t = num;
        while (!t.empty() && t.back() == '0') t.pop_back();
        if (!t.empty() && t.back() == '.') t.pop_back();
    }
    return t;
}

static void format_below_one_second(std::string &buf, uint64_t value) {
    if (value < 1000ull) buf += std::to_string(value) + "ns";
    else if (value < 1000000ull) buf += shortest((double)value / 1e3) + "\xC2\xB5s"; // U+00B5 MICRO SIGN, as the reference prints
    else buf += shortest((double)value / 1e6) + "ms";
}

std::string format_nanoseconds(uint64_t nanoseconds) {
    if (nanoseconds == 0) return "0ns";
    const uint64_t second = 1000000000ull, minute = 60 * second, hour = 60 * minute;
    uint64_t value = nanoseconds;
    std::string buf;
    if (value < second) {
        format_below_one_second(buf, value);
        return buf;
    }
    if (value >= hour) {
        buf += std::to_string(value / hour) + "h";
        value %= hour;
    }
    if (value >= minute) {
        buf += std::to_string(value / minute) + "m";
        value %= minute;
    }
    if (value > 0) {
        if (value < second) format_below_one_second(buf, value);
        else buf += shortest((double)value / 1e9) + "s";
    }
    return buf;
}

bool is_memory_error(const char *e) {
    if (!e) return false;
    const std::string m(e);
    for (const char *key : {"MALLOC", "Not enough memory", "ALLOC_FAILED", "cudaMalloc", "memory is too small", "hipMalloc"})
        if (m.find(key) != std::string::npos) return true;
    return false;
}

void StatsLinSol::set_matrix_name_from_path(const std::string &filepath) {
    size_t slash = filepath.find_last_of("/\\");
    std::string base = slash == std::string::npos ? filepath : filepath.substr(slash + 1);
    size_t dot = base.find_last_of('.');
    if (dot != std::string::npos && dot > 0) base = base.substr(0, dot);
    matrix_name = base.empty() ? "Unknown" : base;
}

// coo_matrix.rs:872-887 (get_actual_nnz): off-diagonal entries of triangular storage count twice
template <typename Coo> static size_t actual_nnz(const Coo &coo) {
    if (coo.symmetric != Sym::YesLower && coo.symmetric != Sym::YesUpper) return coo.nnz;
    size_t actual = 0;
    for (size_t p = 0; p < coo.nnz; p++) actual += coo.indices_i[p] != coo.indices_j[p] ? 2 : 1;
    return actual;
}

void StatsLinSol::set_matrix_info_from_coo(const CooMatrix &coo) {
    nrow = coo.nrow, ncol = coo.ncol, nnz = coo.nnz, nnz_actual = actual_nnz(coo);
    complex = false;
    symmetric = sym_name(coo.symmetric);
}

void StatsLinSol::set_matrix_info_from_coo(const ComplexCooMatrix &coo) {
    nrow = coo.nrow, ncol = coo.ncol, nnz = coo.nnz, nnz_actual = actual_nnz(coo);
    complex = true;
    symmetric = sym_name(coo.symmetric);
}

static uint64_t avg(const std::vector<uint64_t> &v) {
    if (v.empty()) return 0;
    double s = 0;
    for (auto x : v) s += (double)x;
    return (uint64_t)(s / (double)v.size());
}
static std::string arr(const std::vector<uint64_t> &v) {
    std::ostringstream o;
    o << "[";
    for (size_t i = 0; i < v.size(); i++) o << (i ? "," : "") << v[i];
    o << "]";
    return o.str();
}
static std::string human_arr(const std::vector<uint64_t> &v) {
    std::ostringstream o;
    o << "[";
    for (size_t i = 0; i < v.size(); i++) o << (i ? "," : "") << "\"" << format_nanoseconds(v[i]) << "\"";
    o << "]";
    return o.str();
}
static std::string json_escape(const std::string &t) {
    std::string r;
    for (char c : t) {
        if (c == '"' || c == '\\') r += '\\';
        r += c;
    }
    return r;
}
// two-space indentation of a compact JSON text (what serde_json::to_string_pretty prints)
static std::string json_pretty(const std::string &compact) {
    std::string out;
    int depth = 0;
    bool in_string = false;
    auto newline = [&]() {
        out += '\n';
        out.append(2 * (size_t)depth, ' ');
    };
    for (size_t i = 0; i < compact.size(); i++) {
        const char c = compact[i];
        if (in_string) {
            out += c;
            if (c == '\\' && i + 1 < compact.size()) out += compact[++i];
            else if (c == '"') in_string = false;
            continue;
        }
        if (c == '"') {
            in_string = true;
            out += c;
        } else if (c == '{' || c == '[') {
            out += c;
            if (i + 1 < compact.size() && (compact[i + 1] == '}' || compact[i + 1] == ']')) {
                out += compact[++i]; // empty container stays on one line
            } else {
                depth++;
                newline();
            }
        } else if (c == '}' || c == ']') {
            depth--;
            newline();
            out += c;
        } else if (c == ',') {
            out += c;
            newline();
        } else if (c == ':') {
            out += ": ";
        } else
            out += c;
    }
    return out;
}

std::string StatsLinSol::to_json(bool pretty) const {
    std::vector<uint64_t> total;
    for (size_t i = 0; i < std::min(initialize_ns.size(), std::min(factorize_ns.size(), solve_ns.size())); i++)
        total.push_back(initialize_ns[i] + factorize_ns[i] + solve_ns[i]);
    char num[64];
    std::ostringstream o;
    auto f = [&](double v) {
        snprintf(num, sizeof num, "%.17g", v);
        std::string t(num);
        if (t.find_first_of(".enai") == std::string::npos) t += ".0"; // floats stay floats in the JSON text
        return t;
    };
    auto q = [&](const std::string &t) { return "\"" + json_escape(t) + "\""; };
    o << "{\"main\":{\"platform\":\"MI355X gfx950\",\"blas_lib\":\"none (hand-written HIP kernels)\",\"solver\":" << q(solver)
      << ",\"local_sparse\":false,\"out_of_memory\":" << (out_of_memory ? "true" : "false") << "},"
      << "\"matrix\":{\"name\":" << q(matrix_name) << ",\"nrow\":" << nrow << ",\"ncol\":" << ncol << ",\"nnz\":" << nnz
      << ",\"nnz_actual\":" << nnz_actual << ",\"complex\":" << (complex ? "true" : "false") << ",\"symmetric\":" << q(symmetric) << "},"
      << "\"requests\":{\"ordering\":" << q(ordering) << ",\"scaling\":" << q(scaling) << ",\"matching\":" << q(matching)
      << ",\"pivoting\":" << q(pivoting) << ",\"mumps_num_threads\":0"
      << ",\"positive_definite\":" << (positive_definite ? "true" : "false")
      << ",\"hybrid_memory_factor\":" << (has_hybrid_memory_factor ? f(hybrid_memory_factor) : std::string("null")) << "},"
      << "\"output\":{\"effective_ordering\":" << q(effective_ordering) << ",\"effective_scaling\":" << q(effective_scaling)
      << ",\"effective_matching\":" << q(effective_matching) << ",\"effective_pivoting\":" << q(effective_pivoting)
      << ",\"effective_mumps_num_threads\":0,\"openmp_num_threads\":0,\"umfpack_strategy\":" << q(umfpack_strategy)
      << ",\"umfpack_rcond_estimate\":" << f(rcond_estimate) << ",\"rcond_estimate\":" << f(rcond_estimate)
      << ",\"perturbed_pivots\":" << perturbed_pivots << "},"
      << "\"determinant\":{\"mantissa_real\":" << f(det_mantissa) << ",\"mantissa_imag\":" << f(det_mantissa_imag) << ",\"base\":" << f(det_base)
      << ",\"exponent\":" << f(det_exponent) << "},"
      << "\"verify\":{\"max_abs_a\":" << f(verify.max_abs_a) << ",\"max_abs_ax\":" << f(verify.max_abs_ax)
      << ",\"max_abs_diff\":" << f(verify.max_abs_diff) << ",\"relative_error\":" << f(verify.relative_error) << "},"
      << "\"time_human\":{\"read_matrix\":" << q(format_nanoseconds(read_matrix_ns)) << ",\"initialize_array\":" << human_arr(initialize_ns)
      << ",\"initialize\":" << q(format_nanoseconds(avg(initialize_ns))) << ",\"factorize_array\":" << human_arr(factorize_ns)
      << ",\"factorize\":" << q(format_nanoseconds(avg(factorize_ns))) << ",\"solve_array\":" << human_arr(solve_ns)
      << ",\"solve\":" << q(format_nanoseconds(avg(solve_ns))) << ",\"total_ifs_array\":" << human_arr(total)
      << ",\"total_ifs\":" << q(format_nanoseconds(avg(total))) << ",\"verify\":" << q(format_nanoseconds(verify_ns)) << "},"
      << "\"time_nanoseconds\":{\"read_matrix\":" << read_matrix_ns << ",\"initialize_array\":" << arr(initialize_ns)
      << ",\"initialize\":" << avg(initialize_ns) << ",\"factorize_array\":" << arr(factorize_ns) << ",\"factorize\":" << avg(factorize_ns)
      << ",\"solve_array\":" << arr(solve_ns) << ",\"solve\":" << avg(solve_ns) << ",\"total_ifs_array\":" << arr(total)
      << ",\"total_ifs\":" << avg(total) << ",\"verify\":" << verify_ns << "},"
      // (stats_lin_sol.rs:100-113: MUMPS's own error analysis; zeros for every other solver, as in the reference)
      << "\"mumps_stats\":{\"inf_norm_a\":0.0,\"inf_norm_x\":0.0,\"scaled_residual\":0.0,\"backward_error_omega1\":0.0,\"backward_error_omega2\":0.0,"
         "\"normalized_delta_x\":0.0,\"condition_number1\":0.0,\"condition_number2\":0.0}}";
    return pretty ? json_pretty(o.str()) : o.str();
}

// ---- the HIP backend, loaded at run time ---------------------------------------------------------------------
namespace {
struct Backend {
    void *dl = nullptr;
    decltype(&solver_hipmf_new) new_ = nullptr;
    decltype(&solver_hipmf_drop) drop = nullptr;
    decltype(&solver_hipmf_initialize) initialize = nullptr;
    decltype(&solver_hipmf_factorize) factorize = nullptr;
    decltype(&solver_hipmf_solve) solve = nullptr;
    decltype(&solver_hipmf_solve_many) solve_many = nullptr;
    decltype(&solver_hipmf_set_value_map) set_value_map = nullptr;
    decltype(&solver_hipmf_factorize_mapped) factorize_mapped = nullptr;
    decltype(&solver_hipmf_get_stats) get_stats = nullptr;
    decltype(&solver_hipmf_set_option) set_option = nullptr;
    decltype(&solver_hipmf_get_option) get_option = nullptr;
    decltype(&complex_solver_hipmf_new) znew = nullptr;
    decltype(&complex_solver_hipmf_drop) zdrop = nullptr;
    decltype(&complex_solver_hipmf_initialize) zinitialize = nullptr;
    decltype(&complex_solver_hipmf_factorize) zfactorize = nullptr;
    decltype(&complex_solver_hipmf_solve) zsolve = nullptr;
    decltype(&complex_solver_hipmf_set_value_map) zset_value_map = nullptr;
    decltype(&complex_solver_hipmf_factorize_mapped) zfactorize_mapped = nullptr;
    decltype(&complex_solver_hipmf_get_determinant) zget_determinant = nullptr;
    decltype(&complex_solver_hipmf_get_stats) zget_stats = nullptr;
    decltype(&solver_hipmf_get_counter) get_counter = nullptr;
    decltype(&complex_solver_hipmf_get_counter) zget_counter = nullptr;
    bool tried = false;
};
Backend g_backend;
std::string g_lib_path;

bool load_backend() {
    if (g_backend.tried) return g_backend.dl != nullptr;
    g_backend.tried = true;
    std::string path = g_lib_path;
    if (path.empty()) {
        const char *env = getenv("RUSSELL_HIPMF_LIB");
        if (env) path = env;
    }
    if (path.empty()) {
        Dl_info info;
        if (dladdr((void *)&load_backend, &info) && info.dli_fname) {
            std::string self = info.dli_fname;
            size_t slash = self.rfind('/');
            path = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/librussell_hipmf.so";
        }
    }
    void *dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) return false;
#define BIND(field, name)                                              \
    g_backend.field = (decltype(g_backend.field))dlsym(dl, name);      \
    if (!g_backend.field) {                                            \
        dlclose(dl);                                                   \
        return false;                                                  \
    }
    BIND(new_, "solver_hipmf_new")
    BIND(drop, "solver_hipmf_drop")
    BIND(initialize, "solver_hipmf_initialize")
    BIND(factorize, "solver_hipmf_factorize")
    BIND(solve, "solver_hipmf_solve")
    BIND(solve_many, "solver_hipmf_solve_many")
    BIND(set_value_map, "solver_hipmf_set_value_map")
    BIND(factorize_mapped, "solver_hipmf_factorize_mapped")
    BIND(get_stats, "solver_hipmf_get_stats")
    BIND(set_option, "solver_hipmf_set_option")
    BIND(get_option, "solver_hipmf_get_option")
    BIND(znew, "complex_solver_hipmf_new")
    BIND(zdrop, "complex_solver_hipmf_drop")
    BIND(zinitialize, "complex_solver_hipmf_initialize")
    BIND(zfactorize, "complex_solver_hipmf_factorize")
    BIND(zsolve, "complex_solver_hipmf_solve")
    BIND(zset_value_map, "complex_solver_hipmf_set_value_map")
    BIND(zfactorize_mapped, "complex_solver_hipmf_factorize_mapped")
    BIND(zget_determinant, "complex_solver_hipmf_get_determinant")
    BIND(zget_stats, "complex_solver_hipmf_get_stats")
    BIND(get_counter, "solver_hipmf_get_counter")
    BIND(zget_counter, "complex_solver_hipmf_get_counter")
#undef BIND
    g_backend.dl = dl;
    return true;
}

uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

void set_hipmf_library_path(const std::string &path) {
    g_lib_path = path;
    g_backend.tried = false;
}

// Ordering -> C-ABI constant, in the manner of umfpack_ordering (solver_umfpack.rs:457-472): the minimum-degree family (Amd, Amf, Qamd)
// selects the backend's approximate minimum degree, Best tries both orderings, No is the natural order, everything else the default
// (nested dissection)
static int32_t hipmf_ordering(Ordering o) {
    switch (o) {
    case Ordering::No: return HIPMF_ORDERING_NONE;
    case Ordering::Amd:
    case Ordering::Amf:
    case Ordering::Qamd: return HIPMF_ORDERING_AMD;
    case Ordering::Best: return HIPMF_ORDERING_BEST; // both, the sparser one (UMFPACK_ORDERING_BEST's meaning)
    default: return HIPMF_ORDERING_DEFAULT;
    }
}
// Scaling -> C-ABI constant, same mapping as umfpack_scaling (solver_umfpack.rs:475-487)
static int32_t hipmf_scaling(Scaling s) {
    switch (s) {
    case Scaling::Max: return HIPMF_SCALE_MAX;
    case Scaling::No: return HIPMF_SCALE_NONE;
    default: return HIPMF_SCALE_SUM;
    }
}

StrError handle_hipmf_error_code(int32_t err) {
    switch (err) {
    case HIPMF_WARNING_SINGULAR_MATRIX: return "Error(1): Matrix is singular";
    case ERROR_NULL_POINTER: return "Error: c-code returned null pointer (HIPMF)";
    case ERROR_MALLOC: return "Error: c-code failed to allocate memory (HIPMF)";
    case ERROR_VERSION: return "Error: c-code returned version error (HIPMF)";
    case ERROR_NOT_AVAILABLE: return "Error: c-code returned not available (HIPMF): a frontal matrix exceeds the LDS staging limit";
    case ERROR_NEED_INITIALIZATION: return "Error: c-code requires initialization (HIPMF)";
    case ERROR_NEED_FACTORIZATION: return "Error: c-code requires factorization (HIPMF)";
    case ERROR_ALREADY_INITIALIZED: return "Error: c-code requires no previous initialization (HIPMF)";
    case ERROR_HIP_MALLOC: return "hipMalloc failed in the C code (HIPMF): Not enough memory";
    case ERROR_HIP_MEMCPY: return "hipMemcpy failed in the C code (HIPMF)";
    case ERROR_HIP_SYNCHRONIZE: return "hipStreamSynchronize failed in the C code (HIPMF)";
    case ERROR_HIP_LAUNCH: return "a HIP kernel launch failed in the C code (HIPMF)";
    case ERROR_HIPMF_INVALID_MATRIX: return "HIPMF symbolic analysis failed: invalid CSR structure";
    case ERROR_HIPMF_SYMBOLIC: return "HIPMF symbolic analysis failed: internal error";
    case ERROR_HIPMF_INVALID_VALUE: return "HIPMF solve failed: invalid value";
    case ERROR_HIPMF_COMM: return "HIPMF: an RCCL call failed";
    case ERROR_HIPMF_NO_DEVICE: return "HIPMF: no HIP device is visible";
    default: return "Error: unknown error returned by c-code (HIPMF)";
    }
}

StrError SolverHIPMF::create(std::unique_ptr<SolverHIPMF> &out) {
    if (!load_backend()) return "HIPMF solver is not available";
    void *h = g_backend.new_();
    if (!h) return "c-code failed to allocate the HIPMF solver";
    out.reset(new SolverHIPMF());
    out->solver = h;
    return nullptr;
}

SolverHIPMF::~SolverHIPMF() {
    if (solver && g_backend.drop) g_backend.drop((InterfaceHIPMF *)solver);
}

// solver_cudss.rs:194-311 with this backend's symmetry rule and parameters
StrError SolverHIPMF::factorize(const CooMatrix &mat, const LinSolParams *params) {
    if (initialized) {
        if (mat.symmetric != initialized_sym) return "subsequent factorizations must use the same matrix (symmetric differs)";
        if (mat.nrow != initialized_ndim) return "subsequent factorizations must use the same matrix (ndim differs)";
        if (mat.nnz != initialized_nnz) return "subsequent factorizations must use the same matrix (nnz differs)";
        if (params) return "subsequent factorizations must not change LinSolParams";
        // The value map was built from the FIRST call's triplets.  The reference re-reads the indices on every call
        // (csr_matrix.rs:359-480), so the triplet order may change between calls: the map is only used while the indices are
        // the ones it was built from (O(nnz) compare); otherwise the values go through the host conversion, and a changed
        // pattern is refused.
        // (that compare reads 16 bytes per triplet -- 1.5 - 2 ms of a 8.6 ms repeat call at 5 M triplets -- and its answer is almost always
        //  "same": it runs on a host thread of its own BESIDE the mapped factorisation below, which is thrown away if the answer is "changed")
        if (!value_map_set) { // (the device refreshes the values through the map otherwise: no host conversion per call)
            const std::vector<int32_t> rp0 = csr.row_pointers, ci0 = csr.col_indices;
            StrError e = csr.update_from_coo(mat);
            if (e) return e;
            const size_t nz0 = (size_t)rp0[csr.nrow];
            if (csr.row_pointers != rp0 || std::memcmp(ci0.data(), csr.col_indices.data(), sizeof(int32_t) * nz0) != 0) {
                csr.row_pointers = rp0, csr.col_indices = ci0; // (the handle keeps the pattern it was initialised with: a later valid call is compared with THAT)
                return "subsequent factorizations must use the same matrix (sparsity pattern differs)";
            }
        }
    } else {
        if (mat.nrow != mat.ncol) return "the matrix must be square";
        if (mat.nnz < 1) return "the COO matrix must have at least one non-zero value";
        if (mat.symmetric == Sym::YesFull || mat.symmetric == Sym::YesUpper) return "HIPMF requires Sym::YesLower for symmetric matrices";
        initialized_sym = mat.symmetric;
        initialized_ndim = mat.nrow;
        initialized_nnz = mat.nnz;
        StrError e = CsrMatrix::from_coo(csr, mat);
        if (e) return e;
    }
    LinSolParams par = params ? *params : LinSolParams();
    const int32_t verbose = par.verbose ? 1 : 0;
    if (!initialized) {
        compute_determinant = par.compute_determinant;
        uint64_t t0 = now_ns();
        // the parameters the initialize signature does not carry (lin_sol_params.rs:13-16,39)
        // (round 6: Pivoting is a request, as for cuDSS -- solver_cudss.rs:233,298 hands it over and reads the EFFECTIVE strategy back.
        //  Every value is accepted; what runs is partial pivoting inside the pivot block = LocalBlock, with replaced pivots and the
        //  Krylov rescue for what that cannot fix; update_stats reports it)
        if (g_backend.set_option((InterfaceHIPMF *)solver, HIPMF_OPTION_PIVOTING, (double)(int32_t)par.pivoting) != SUCCESSFUL_EXIT) return "HIPMF: invalid pivoting option";
        const double mval = par.matching == Matching::None ? 0.0 : (par.matching == Matching::Auto ? 1.0 : 2.0);
        if (g_backend.set_option((InterfaceHIPMF *)solver, HIPMF_OPTION_MATCHING, mval) != SUCCESSFUL_EXIT) return "HIPMF: invalid matching option";
        if (par.has_hybrid_memory_factor) {
            if (!(par.hybrid_memory_factor >= 0.01 && par.hybrid_memory_factor <= 0.99)) return "hybrid_memory_factor must satisfy: 0.01 ≤ factor ≤ 0.99";
            (void)g_backend.set_option((InterfaceHIPMF *)solver, HIPMF_OPTION_HYBRID_MEMORY, par.hybrid_memory_factor);
        }
        int32_t status = g_backend.initialize((InterfaceHIPMF *)solver, hipmf_ordering(par.ordering), hipmf_scaling(par.scaling),
                                              par.has_pivot_epsilon ? par.pivot_epsilon : -1.0,
                                              par.has_refinement_nstep ? par.refinement_nstep : -1, verbose,
                                              mat.symmetric == Sym::YesLower ? 1 : 0,
                                              (par.positive_definite && mat.symmetric == Sym::YesLower) ? 1 : 0, (int32_t)csr.nrow,
                                              csr.row_pointers.data(), csr.col_indices.data(), csr.values.data());
        if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
        // value map for the repeat calls: CSR entry <- the COO triplets (duplicates in COO order) that sum into it
        {
            const size_t nz = (size_t)csr.row_pointers[csr.nrow];
            std::vector<int32_t> seg_ptr(nz + 1, 0), seg_idx(mat.nnz), pos(mat.nnz);
            for (size_t k = 0; k < mat.nnz; k++) {
                const int32_t i = mat.indices_i[k], j = mat.indices_j[k];
                const int32_t *b = csr.col_indices.data() + csr.row_pointers[i], *e = csr.col_indices.data() + csr.row_pointers[i + 1];
                pos[k] = (int32_t)(std::lower_bound(b, e, j) - csr.col_indices.data());
                seg_ptr[(size_t)pos[k] + 1]++;
            }
            for (size_t q = 0; q < nz; q++) seg_ptr[q + 1] += seg_ptr[q];
            std::vector<int32_t> w(seg_ptr.begin(), seg_ptr.end() - 1);
            for (size_t k = 0; k < mat.nnz; k++) seg_idx[(size_t)w[pos[k]]++] = (int32_t)k;
            value_map_set = g_backend.set_value_map((InterfaceHIPMF *)solver, (int32_t)mat.nnz, seg_ptr.data(), seg_idx.data()) == SUCCESSFUL_EXIT;
            if (value_map_set) {
                map_i.assign(mat.indices_i.begin(), mat.indices_i.begin() + (std::ptrdiff_t)mat.nnz);
                map_j.assign(mat.indices_j.begin(), mat.indices_j.begin() + (std::ptrdiff_t)mat.nnz);
            }
        }
        time_initialize_ns = now_ns() - t0;
        initialized = true;
        first_call = true;
    }
    uint64_t t0 = now_ns();
    int32_t status;
    if (value_map_set && !first_call) {
        std::future<bool> same = std::async(std::launch::async, [&]() {
            return std::memcmp(map_i.data(), mat.indices_i.data(), sizeof(int32_t) * mat.nnz) == 0 &&
                   std::memcmp(map_j.data(), mat.indices_j.data(), sizeof(int32_t) * mat.nnz) == 0;
        });
        status = g_backend.factorize_mapped((InterfaceHIPMF *)solver, &effective_ordering, &effective_scaling, &perturbed_pivots, &rcond_estimate,
                                            &determinant_coefficient, &determinant_exponent, compute_determinant ? 1 : 0, verbose,
                                            mat.values.data());
        if (!same.get()) {
            // the triplets came in another order than the map was built from: the factorisation just made used the wrong values.
            // Host conversion (which also refuses a changed pattern), then the plain factorisation; the map is not used again.
            // (the handle holds a factor of mis-mapped values from here until the redo below succeeds -- every early
            //  return of this branch must leave `factorized` false, or a later solve() would silently use that factor)
            factorized = false;
            value_map_set = false;
            map_i.clear(), map_j.clear();
            const std::vector<int32_t> rp0 = csr.row_pointers, ci0 = csr.col_indices;
            StrError e = csr.update_from_coo(mat);
            if (e) return e;
            const size_t nz0 = (size_t)rp0[csr.nrow];
            if (csr.row_pointers != rp0 || std::memcmp(ci0.data(), csr.col_indices.data(), sizeof(int32_t) * nz0) != 0) {
                csr.row_pointers = rp0, csr.col_indices = ci0; // (the handle keeps the pattern it was initialised with: a later valid call is compared with THAT)
                return "subsequent factorizations must use the same matrix (sparsity pattern differs)";
            }
            status = g_backend.factorize((InterfaceHIPMF *)solver, &effective_ordering, &effective_scaling, &perturbed_pivots, &rcond_estimate,
                                         &determinant_coefficient, &determinant_exponent, compute_determinant ? 1 : 0, verbose, csr.values.data());
        }
    } else
        status = g_backend.factorize((InterfaceHIPMF *)solver, &effective_ordering, &effective_scaling, &perturbed_pivots, &rcond_estimate,
                                         &determinant_coefficient, &determinant_exponent, compute_determinant ? 1 : 0, verbose, csr.values.data());
    first_call = false;
    if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
    time_factorize_ns = now_ns() - t0;
    factorized = true;
    int64_t istats[16];
    double dstats[16];
    if (g_backend.get_stats((InterfaceHIPMF *)solver, istats, dstats) == SUCCESSFUL_EXIT) effective_matching = istats[14] != 0;
    return nullptr;
}

int64_t SolverHIPMF::get_counter(int32_t which) const { return solver ? g_backend.get_counter((InterfaceHIPMF *)solver, which) : -1; }
int64_t ComplexSolverHIPMF::get_counter(int32_t which) const { return solver ? g_backend.zget_counter((InterfaceComplexHIPMF *)solver, which) : -1; }

StrError SolverHIPMF::solve(std::vector<double> &x, const std::vector<double> &rhs, bool verbose) {
    return solve_slices(x.data(), x.size(), rhs.data(), rhs.size(), verbose);
}

// the same on borrowed slices (&mut [f64], &[f64] in the reference): what the C glue of the Python driver calls, without copies
StrError SolverHIPMF::solve_slices(double *x, size_t nx, const double *rhs, size_t nr, bool verbose) {
    if (!factorized) return "the function factorize must be called before solve";
    if (nx != initialized_ndim) return "the dimension of the vector of unknown values x is incorrect";
    if (nr != initialized_ndim) return "the dimension of the right-hand side vector is incorrect";
    uint64_t t0 = now_ns();
    int32_t status = g_backend.solve((InterfaceHIPMF *)solver, x, rhs, verbose ? 1 : 0);
    if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
    time_solve_ns = now_ns() - t0;
    return nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
// complex twin through the real-equivalent system
StrError ComplexCooMatrix::create(ComplexCooMatrix &out, size_t nrow, size_t ncol, size_t max_nnz, Sym symmetric) {
    if (nrow < 1) return "nrow must be ≥ 1";
    if (ncol < 1) return "ncol must be ≥ 1";
    if (max_nnz < 1) return "max_nnz must be ≥ 1";
    out.symmetric = symmetric;
    out.nrow = nrow, out.ncol = ncol, out.nnz = 0, out.max_nnz = max_nnz;
    out.indices_i.assign(max_nnz, 0);
    out.indices_j.assign(max_nnz, 0);
    out.values.assign(2 * max_nnz, 0.0);
    return nullptr;
}

StrError ComplexCooMatrix::put(size_t i, size_t j, double re, double im) {
    // coo_matrix.rs:324-352 (same checks for the complex instantiation)
    if (i >= nrow) return "COO matrix: index of row is outside range";
    if (j >= ncol) return "COO matrix: index of column is outside range";
    if (nnz >= max_nnz) return "COO matrix: max number of items has been reached";
    if (symmetric == Sym::YesLower && j > i) return "COO matrix: j > i is incorrect for lower triangular storage";
    if (symmetric == Sym::YesUpper && j < i) return "COO matrix: j < i is incorrect for upper triangular storage";
    indices_i[nnz] = (int32_t)i, indices_j[nnz] = (int32_t)j;
    values[2 * nnz] = re, values[2 * nnz + 1] = im;
    nnz++;
    return nullptr;
}

StrError ComplexCooMatrix::mat_vec_mul(std::vector<double> &v, double ar, double ai, const std::vector<double> &u) const {
    if (u.size() != 2 * ncol) return "u.ndim must equal ncol";
    if (v.size() != 2 * nrow) return "v.ndim must equal nrow";
    std::fill(v.begin(), v.end(), 0.0);
    const bool mirror = symmetric == Sym::YesLower || symmetric == Sym::YesUpper;
    for (size_t k = 0; k < nnz; k++) {
        const size_t i = (size_t)indices_i[k], j = (size_t)indices_j[k];
        // alpha * a_ij
        const double cr = ar * values[2 * k] - ai * values[2 * k + 1], ci = ar * values[2 * k + 1] + ai * values[2 * k];
        v[2 * i] += cr * u[2 * j] - ci * u[2 * j + 1];
        v[2 * i + 1] += cr * u[2 * j + 1] + ci * u[2 * j];
        if (mirror && i != j) {
            v[2 * j] += cr * u[2 * i] - ci * u[2 * i + 1];
            v[2 * j + 1] += cr * u[2 * i + 1] + ci * u[2 * i];
        }
    }
    return nullptr;
}

StrError ComplexSolverHIPMF::create(std::unique_ptr<ComplexSolverHIPMF> &out) {
    if (!load_backend()) return "HIPMF solver is not available";
    void *h = g_backend.znew();
    if (!h) return "c-code failed to allocate the HIPMF solver";
    out.reset(new ComplexSolverHIPMF());
    out->solver = h;
    return nullptr;
}

ComplexSolverHIPMF::~ComplexSolverHIPMF() {
    if (solver && g_backend.zdrop) g_backend.zdrop((InterfaceComplexHIPMF *)solver);
}

// complex COO -> CSR with the duplicates summed in COO order within a row (the order the reference's conversion adds them,
// csr_matrix.rs:359-480 for NumCsrMatrix<Complex64>), columns ascending; also the triplet map (CSR entry <- its triplets)
StrError ComplexSolverHIPMF::to_csr(const ComplexCooMatrix &mat, bool pattern_too) {
    const size_t n = mat.nrow, nz = mat.nnz;
    if (pattern_too) {
        std::vector<int32_t> cnt(n + 1, 0);
        for (size_t k = 0; k < nz; k++) {
            if (mat.indices_i[k] < 0 || (size_t)mat.indices_i[k] >= n || mat.indices_j[k] < 0 || (size_t)mat.indices_j[k] >= n) return "COO matrix: index out of range";
            cnt[(size_t)mat.indices_i[k] + 1]++;
        }
        for (size_t i = 0; i < n; i++) cnt[i + 1] += cnt[i];
        std::vector<int32_t> byrow(nz), w(cnt.begin(), cnt.end() - 1);
        for (size_t k = 0; k < nz; k++) byrow[(size_t)w[(size_t)mat.indices_i[k]]++] = (int32_t)k; // (stable: COO order within a row)
        zrp.assign(n + 1, 0);
        zci.clear();
        seg_ptr.clear();
        seg_idx.clear();
        seg_idx.reserve(nz);
        std::vector<int32_t> row;
        for (size_t i = 0; i < n; i++) {
            row.assign(byrow.begin() + cnt[i], byrow.begin() + cnt[i + 1]);
            std::stable_sort(row.begin(), row.end(), [&](int32_t a, int32_t b) { return mat.indices_j[(size_t)a] < mat.indices_j[(size_t)b]; });
            for (size_t q = 0; q < row.size(); q++) {
                if (q == 0 || mat.indices_j[(size_t)row[q]] != mat.indices_j[(size_t)row[q - 1]]) { // a new CSR entry starts
                    seg_ptr.push_back((int32_t)seg_idx.size());
                    zci.push_back(mat.indices_j[(size_t)row[q]]);
                }
                seg_idx.push_back(row[q]);
            }
            zrp[i + 1] = (int32_t)zci.size();
        }
        seg_ptr.push_back((int32_t)seg_idx.size());
    }
    zvals.assign(2 * zci.size(), 0.0);
    for (size_t c = 0; c < zci.size(); c++)
        for (int32_t q = seg_ptr[c]; q < seg_ptr[c + 1]; q++) {
            zvals[2 * c] += mat.values[2 * (size_t)seg_idx[(size_t)q]];
            zvals[2 * c + 1] += mat.values[2 * (size_t)seg_idx[(size_t)q] + 1];
        }
    return nullptr;
}

StrError ComplexSolverHIPMF::factorize(const ComplexCooMatrix &mat, const LinSolParams *params) {
    LinSolParams par = params ? *params : LinSolParams();
    const int32_t verbose = par.verbose ? 1 : 0;
    if (initialized) {
        if (mat.symmetric != initialized_sym) return "subsequent factorizations must use the same matrix (symmetric differs)";
        if (mat.nrow != initialized_ndim) return "subsequent factorizations must use the same matrix (ndim differs)";
        if (mat.nnz != initialized_nnz) return "subsequent factorizations must use the same matrix (nnz differs)";
        if (params) return "subsequent factorizations must not change LinSolParams";
    } else {
        if (mat.nrow != mat.ncol) return "the matrix must be square";
        if (mat.nnz < 1) return "the COO matrix must have at least one non-zero value";
        if (mat.symmetric == Sym::YesFull || mat.symmetric == Sym::YesUpper) return "HIPMF requires Sym::YesLower for symmetric matrices";
        compute_determinant = par.compute_determinant;
        StrError e = to_csr(mat, true);
        if (e) return e;
        uint64_t t0 = now_ns();
        int32_t status = g_backend.zinitialize((InterfaceComplexHIPMF *)solver, hipmf_ordering(par.ordering), hipmf_scaling(par.scaling),
                                               par.has_pivot_epsilon ? par.pivot_epsilon : -1.0, par.has_refinement_nstep ? par.refinement_nstep : -1,
                                               verbose, mat.symmetric == Sym::YesLower ? 1 : 0, (int32_t)mat.nrow, zrp.data(), zci.data(), zvals.data());
        if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
        value_map_set = g_backend.zset_value_map((InterfaceComplexHIPMF *)solver, (int32_t)mat.nnz, seg_ptr.data(), seg_idx.data()) == SUCCESSFUL_EXIT;
        if (value_map_set) {
            map_i.assign(mat.indices_i.begin(), mat.indices_i.begin() + (std::ptrdiff_t)mat.nnz);
            map_j.assign(mat.indices_j.begin(), mat.indices_j.begin() + (std::ptrdiff_t)mat.nnz);
        }
        time_initialize_ns = now_ns() - t0;
        initialized_sym = mat.symmetric;
        initialized_ndim = mat.nrow;
        initialized_nnz = mat.nnz;
        initialized = true;
    }
    // the value map belongs to the triplet order it was built from (the reference re-reads the indices on every call)
    if (value_map_set && (std::memcmp(map_i.data(), mat.indices_i.data(), sizeof(int32_t) * mat.nnz) != 0 ||
                          std::memcmp(map_j.data(), mat.indices_j.data(), sizeof(int32_t) * mat.nnz) != 0)) {
        const std::vector<int32_t> rp0 = zrp, ci0 = zci;
        StrError e = to_csr(mat, true);
        if (e) return e;
        if (zrp != rp0 || zci != ci0) return "subsequent factorizations must use the same matrix (sparsity pattern differs)";
        value_map_set = g_backend.zset_value_map((InterfaceComplexHIPMF *)solver, (int32_t)mat.nnz, seg_ptr.data(), seg_idx.data()) == SUCCESSFUL_EXIT;
        if (value_map_set) {
            map_i.assign(mat.indices_i.begin(), mat.indices_i.begin() + (std::ptrdiff_t)mat.nnz);
            map_j.assign(mat.indices_j.begin(), mat.indices_j.begin() + (std::ptrdiff_t)mat.nnz);
        }
    }
    uint64_t t0 = now_ns();
    int32_t status;
    if (value_map_set) {
        status = g_backend.zfactorize_mapped((InterfaceComplexHIPMF *)solver, &effective_ordering, &effective_scaling, &perturbed_pivots, &rcond_estimate,
                                             verbose, mat.values.data());
        determinant_coefficient_real = 0.0, determinant_coefficient_imag = 0.0, determinant_exponent = 0.0;
        if (status == SUCCESSFUL_EXIT && compute_determinant)
            status = g_backend.zget_determinant((InterfaceComplexHIPMF *)solver, &determinant_coefficient_real, &determinant_coefficient_imag, &determinant_exponent);
    } else {
        // no value map (the backend refused it): the values go through a fresh conversion, and the pattern is checked on every call,
        // as the real SolverHIPMF does -- summing through the segments of the first call's triplet order would be silently wrong
        // for permuted triplets
        const std::vector<int32_t> rp0 = zrp, ci0 = zci;
        StrError e = to_csr(mat, true);
        if (e) return e;
        if (zrp != rp0 || zci != ci0) return "subsequent factorizations must use the same matrix (sparsity pattern differs)";
        status = g_backend.zfactorize((InterfaceComplexHIPMF *)solver, &effective_ordering, &effective_scaling, &perturbed_pivots, &rcond_estimate,
                                      &determinant_coefficient_real, &determinant_coefficient_imag, &determinant_exponent, compute_determinant ? 1 : 0,
                                      verbose, zvals.data());
    }
    if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
    time_factorize_ns = now_ns() - t0;
    factorized = true;
    int64_t istats[16];
    double dstats[16];
    if (g_backend.zget_stats((InterfaceComplexHIPMF *)solver, istats, dstats) == SUCCESSFUL_EXIT) effective_matching = istats[14] != 0;
    return nullptr;
}

StrError ComplexSolverHIPMF::solve(std::vector<double> &x, const std::vector<double> &rhs, bool verbose) {
    if (!factorized) return "the function factorize must be called before solve";
    if (x.size() != 2 * initialized_ndim) return "the dimension of the vector of unknown values x is incorrect";
    if (rhs.size() != 2 * initialized_ndim) return "the dimension of the right-hand side vector is incorrect";
    uint64_t t0 = now_ns();
    int32_t status = g_backend.zsolve((InterfaceComplexHIPMF *)solver, x.data(), rhs.data(), verbose ? 1 : 0);
    if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
    time_solve_ns = now_ns() - t0;
    return nullptr;
}

void ComplexSolverHIPMF::update_stats(StatsLinSol &stats) const {
    stats.solver = "HIPMF";
    stats.initialize_ns.push_back(time_initialize_ns);
    stats.factorize_ns.push_back(time_factorize_ns);
    stats.solve_ns.push_back(time_solve_ns);
    stats.effective_ordering = effective_ordering == HIPMF_ORDERING_NONE ? "No" : (effective_ordering == HIPMF_ORDERING_AMD ? "Amd" : "Nd"); // what ran: the backend's own minimum degree / nested dissection (none of the ordering libraries is used)
    stats.effective_scaling = effective_scaling == HIPMF_SCALE_MAX ? "Max" : (effective_scaling == HIPMF_SCALE_NONE ? "No" : "Sum");
    stats.rcond_estimate = rcond_estimate;
    // (complex_solver_umfpack.rs:411-414)
    stats.det_mantissa = determinant_coefficient_real, stats.det_mantissa_imag = determinant_coefficient_imag;
    stats.det_base = 10.0, stats.det_exponent = determinant_exponent;
    stats.perturbed_pivots = perturbed_pivots;
    stats.effective_matching = effective_matching ? "MaxProdScaled" : "None";
    stats.effective_pivoting = "LocalBlock";
}

StrError SolverHIPMF::solve_many(std::vector<double> &x, const std::vector<double> &rhs, size_t nrhs) {
    if (!factorized) return "the function factorize must be called before solve";
    if (nrhs < 1 || x.size() != initialized_ndim * nrhs) return "the dimension of the vector of unknown values x is incorrect";
    if (rhs.size() != initialized_ndim * nrhs) return "the dimension of the right-hand side vector is incorrect";
    uint64_t t0 = now_ns();
    int32_t status = g_backend.solve_many((InterfaceHIPMF *)solver, x.data(), rhs.data(), (int32_t)nrhs, (int32_t)initialized_ndim, 0);
    if (status != SUCCESSFUL_EXIT) return handle_hipmf_error_code(status);
    time_solve_ns = now_ns() - t0;
    return nullptr;
}

void SolverHIPMF::update_stats(StatsLinSol &stats) const {
    stats.solver = "HIPMF";
    stats.initialize_ns.push_back(time_initialize_ns);
    stats.factorize_ns.push_back(time_factorize_ns);
    stats.solve_ns.push_back(time_solve_ns);
    stats.effective_ordering = effective_ordering == HIPMF_ORDERING_NONE ? "No" : (effective_ordering == HIPMF_ORDERING_AMD ? "Amd" : "Nd"); // what ran: the backend's own minimum degree / nested dissection (none of the ordering libraries is used)
    stats.effective_scaling = effective_scaling == HIPMF_SCALE_MAX ? "Max" : (effective_scaling == HIPMF_SCALE_NONE ? "No" : "Sum");
    stats.rcond_estimate = rcond_estimate;
    stats.det_mantissa = determinant_coefficient;
    stats.det_base = 10.0;
    stats.det_exponent = determinant_exponent;
    stats.perturbed_pivots = perturbed_pivots;
    stats.effective_matching = effective_matching ? "MaxProdScaled" : "None";
    stats.effective_pivoting = "LocalBlock"; // (enums.rs Pivoting::LocalBlock: pivot search inside the diagonal block of the supernode)
}

StrError LinSolver::create(LinSolver &out, Genie genie) {
    switch (genie) {
    case Genie::Hipmf: {
        std::unique_ptr<SolverHIPMF> s;
        StrError e = SolverHIPMF::create(s);
        if (e) return e;
        out.actual = std::move(s);
        return nullptr;
    }
    case Genie::Umfpack: return "UMFPACK solver is not available";
    case Genie::Mumps: return "MUMPS solver is not available";
    case Genie::Cudss: return "cuDSS solver is not available";
    }
    return "unknown genie";
}

StrError LinSolver::compute(Genie genie, std::vector<double> &x, const CooMatrix &mat, const std::vector<double> &rhs, const LinSolParams *params) {
    LinSolver solver;
    StrError e = LinSolver::create(solver, genie);
    if (e) return e;
    e = solver.actual->factorize(mat, params);
    if (e) return e;
    return solver.actual->solve(x, rhs, false);
}

// ---- MatrixMarket (read_matrix_market.rs:44-184,346-475) -------------------------------------------------------
StrError read_matrix_market(MatrixMarketData &data, const std::string &full_path, MMsym handling) {
    CooMatrix &out = data.real;
    ComplexCooMatrix &zout = data.complex_matrix;
    data.complex = false;
    std::ifstream in(full_path);
    if (!in) return "cannot open file";
    std::string line;
    if (!std::getline(in, line)) return "the file is empty";
    bool complex = false, symmetric = false;
    {
        std::istringstream hs(line);
        std::string w;
        if (!(hs >> w)) return "cannot find the keyword %%MatrixMarket on the first line";
        if (w != "%%MatrixMarket") return "the header (first line) must start with %%MatrixMarket";
        if (!(hs >> w)) return "cannot find the first option in the header line";
        if (w != "matrix") return "after %%MatrixMarket, the first option must be \"matrix\"";
        if (!(hs >> w)) return "cannot find the second option in the header line";
        if (w != "coordinate") return "after %%MatrixMarket, the second option must be \"coordinate\"";
        if (!(hs >> w)) return "cannot find the third option in the header line";
        if (w == "real") complex = false;
        else if (w == "complex") complex = true;
        else return "after %%MatrixMarket, the third option must be \"real\" or \"complex\"";
        if (!(hs >> w)) return "cannot find the fourth option in the header line";
        if (w == "general") symmetric = false;
        else if (w == "symmetric") symmetric = true;
        else if (w == "Hermitian") {
            if (!complex) return "\"Hermitian\" keyword can only be used with the \"complex\" type";
            symmetric = true;
        } else
            return "after %%MatrixMarket, the fourth option must be either \"general\", \"symmetric\", or \"Hermitian\"";
    }
    long m = 0, n = 0, nnz = 0;
    bool have_dims = false;
    while (std::getline(in, line)) {
        size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos || line[b] == '%') continue;
        std::istringstream ds(line);
        std::string a, c, d;
        if (!(ds >> a)) continue;
        char *end;
        m = strtol(a.c_str(), &end, 10);
        if (*end) return "cannot parse number of rows";
        if (!(ds >> c)) return "cannot read number of columns";
        n = strtol(c.c_str(), &end, 10);
        if (*end) return "cannot parse number of columns";
        if (!(ds >> d)) return "cannot read number of non-zeros";
        nnz = strtol(d.c_str(), &end, 10);
        if (*end) return "cannot parse number of non-zeros";
        if (m < 1 || n < 1 || nnz < 1) return "found invalid (zero or negative) dimensions";
        have_dims = true;
        break;
    }
    if (!have_dims) return "cannot read the dimensions line";
    Sym sym = Sym::No;
    if (symmetric) {
        if (m != n) return "MatrixMarket data is invalid: the number of rows must equal the number of columns for symmetric matrices";
        sym = handling == MMsym::LeaveAsLower ? Sym::YesLower : (handling == MMsym::SwapToUpper ? Sym::YesUpper : Sym::YesFull);
    }
    size_t max = (size_t)nnz;
    if (symmetric && handling == MMsym::MakeItFull) max = 2 * (size_t)nnz;
    data.complex = complex;
    StrError e = complex ? ComplexCooMatrix::create(zout, (size_t)m, (size_t)n, max, sym) : CooMatrix::create(out, (size_t)m, (size_t)n, max, sym);
    if (e) return e;
    // one "put" for both value types (Hermitian files are mirrored WITHOUT conjugation, as read_matrix_market.rs:400-436 does)
    double bij = 0.0;
    // (the reference unwraps the put result, read_matrix_market.rs:450-463: an entry on the wrong side of a triangular storage
    // ends the read loudly instead of being dropped)
    StrError put_err = nullptr;
    auto put = [&](size_t i, size_t j, double aij) {
        StrError pe = complex ? zout.put(i, j, aij, bij) : out.put(i, j, aij);
        if (pe && !put_err) put_err = pe;
    };
    long pos = 0;
    while (std::getline(in, line)) {
        size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos || line[b] == '%') continue;
        if (pos == nnz) return "there are more values than specified";
        // tokens are parsed in place (strtol / strtod on the line buffer: the data section is millions of lines long)
        const char *q = line.c_str() + b;
        auto token = [&](const char *&tb, const char *&te) {
            while (*q == ' ' || *q == '\t' || *q == '\r') q++;
            tb = q;
            while (*q && *q != ' ' && *q != '\t' && *q != '\r' && *q != '\n') q++;
            te = q;
            return te > tb;
        };
        const char *tb, *te;
        char *end;
        if (!token(tb, te)) continue;
        long i = strtol(tb, &end, 10);
        if (end != te) return "cannot parse i";
        if (!token(tb, te)) return "cannot read j";
        long j = strtol(tb, &end, 10);
        if (end != te) return "cannot parse j";
        if (!token(tb, te)) return "cannot read aij";
        double aij = strtod(tb, &end);
        if (end != te) return "cannot parse aij";
        if (complex) {
            if (!token(tb, te)) return "cannot read bij";
            bij = strtod(tb, &end);
            if (end != te) return "cannot parse bij";
        }
        i -= 1, j -= 1; // MatrixMarket is one-based
        if (i < 0 || i >= m || j < 0 || j >= n) return "found an invalid index";
        pos++;
        if (symmetric) {
            if (handling == MMsym::LeaveAsLower) put((size_t)i, (size_t)j, aij);
            else if (handling == MMsym::SwapToUpper) put((size_t)j, (size_t)i, aij);
            else {
                put((size_t)i, (size_t)j, aij);
                if (i != j) put((size_t)j, (size_t)i, aij);
            }
        } else {
            put((size_t)i, (size_t)j, aij);
        }
        if (put_err) return put_err;
    }
    if (pos != nnz) return "not all values have been found";
    return nullptr;
}

// real files only (the form the flat C API and the older callers use)
StrError read_matrix_market(CooMatrix &out, const std::string &full_path, MMsym handling) {
    MatrixMarketData data;
    StrError e = read_matrix_market(data, full_path, handling);
    if (e) return e;
    if (data.complex) return "the file holds a complex matrix: use the MatrixMarketData form";
    out = std::move(data.real);
    return nullptr;
}

} // namespace russell

// ====================================================================================================
// flat C API (include/russell_host.h) for ctypes / other FFI users
// ====================================================================================================
using namespace russell;

extern "C" {

struct RhParams {
    int32_t ordering, scaling;
    int32_t has_pivot_epsilon;
    double pivot_epsilon;
    int32_t has_refinement_nstep, refinement_nstep;
    int32_t positive_definite, compute_determinant, verbose;
    int32_t matching, pivoting;
    int32_t has_hybrid_memory_factor;
    double hybrid_memory_factor;
    int32_t compute_error_estimates, compute_condition_numbers;
};

static LinSolParams to_params(const RhParams *p) {
    LinSolParams q;
    q.ordering = (Ordering)p->ordering;
    q.scaling = (Scaling)p->scaling;
    q.has_pivot_epsilon = p->has_pivot_epsilon != 0;
    q.pivot_epsilon = p->pivot_epsilon;
    q.has_refinement_nstep = p->has_refinement_nstep != 0;
    q.refinement_nstep = p->refinement_nstep;
    q.positive_definite = p->positive_definite != 0;
    q.compute_determinant = p->compute_determinant != 0;
    q.verbose = p->verbose != 0;
    q.matching = (Matching)p->matching;
    q.pivoting = (Pivoting)p->pivoting;
    q.has_hybrid_memory_factor = p->has_hybrid_memory_factor != 0;
    q.hybrid_memory_factor = p->hybrid_memory_factor;
    q.compute_error_estimates = p->compute_error_estimates != 0;
    q.compute_condition_numbers = p->compute_condition_numbers != 0;
    return q;
}

void rh_set_hipmf_library(const char *path) { set_hipmf_library_path(path ? path : ""); }

void *rh_coo_new(int64_t nrow, int64_t ncol, int64_t max_nnz, int32_t sym, const char **err) {
    CooMatrix *c = new CooMatrix();
    *err = (nrow < 0 || ncol < 0 || max_nnz < 0) ? "negative dimension" : CooMatrix::create(*c, (size_t)nrow, (size_t)ncol, (size_t)max_nnz, (Sym)sym);
    if (*err) {
        delete c;
        return nullptr;
    }
    return c;
}
void rh_coo_free(void *h) { delete (CooMatrix *)h; }
const char *rh_coo_put(void *h, int64_t i, int64_t j, double aij) {
    if (i < 0) return "COO matrix: index of row is outside range";
    if (j < 0) return "COO matrix: index of column is outside range";
    return ((CooMatrix *)h)->put((size_t)i, (size_t)j, aij);
}
void rh_coo_reset(void *h) { ((CooMatrix *)h)->reset(); }
void rh_coo_info(void *h, int64_t *nrow, int64_t *ncol, int64_t *nnz, int64_t *max_nnz, int32_t *sym) {
    CooMatrix *c = (CooMatrix *)h;
    *nrow = (int64_t)c->nrow, *ncol = (int64_t)c->ncol, *nnz = (int64_t)c->nnz, *max_nnz = (int64_t)c->max_nnz, *sym = (int32_t)c->symmetric;
}
void rh_coo_arrays(void *h, const int32_t **ai, const int32_t **aj, const double **ax) {
    CooMatrix *c = (CooMatrix *)h;
    *ai = c->indices_i.data(), *aj = c->indices_j.data(), *ax = c->values.data();
}
const char *rh_coo_mat_vec_mul(void *h, double *v, int64_t nv, double alpha, const double *u, int64_t nu) {
    std::vector<double> vv((size_t)nv), uu(u, u + nu);
    StrError e = ((CooMatrix *)h)->mat_vec_mul(vv, alpha, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}

void *rh_coo_from(int64_t nrow, int64_t ncol, int64_t nnz, const int32_t *row_indices, const int32_t *col_indices, const double *values, int32_t sym,
                  const char **err) {
    CooMatrix *c = new CooMatrix();
    const size_t k = (size_t)std::max<int64_t>(nnz, 0);
    *err = CooMatrix::from(*c, (size_t)std::max<int64_t>(nrow, 0), (size_t)std::max<int64_t>(ncol, 0), std::vector<int32_t>(row_indices, row_indices + k),
                           std::vector<int32_t>(col_indices, col_indices + k), std::vector<double>(values, values + k), (Sym)sym);
    if (*err) {
        delete c;
        return nullptr;
    }
    return c;
}
const char *rh_coo_mat_vec_mul_update(void *h, double *v, int64_t nv, double alpha, const double *u, int64_t nu) {
    std::vector<double> vv(v, v + nv), uu(u, u + nu);
    StrError e = ((CooMatrix *)h)->mat_vec_mul_update(vv, alpha, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}
const char *rh_coo_mat_t_vec_mul(void *h, double *v, int64_t nv, double alpha, const double *u, int64_t nu) {
    std::vector<double> vv((size_t)nv), uu(u, u + nu);
    StrError e = ((CooMatrix *)h)->mat_t_vec_mul(vv, alpha, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}
const char *rh_coo_assign(void *h, double alpha, void *other) { return ((CooMatrix *)h)->assign(alpha, *(CooMatrix *)other); }
const char *rh_coo_add(void *h, double alpha, void *other) { return ((CooMatrix *)h)->add(alpha, *(CooMatrix *)other); }
const char *rh_coo_put_lagrange_block(void *h, void *bb) { return ((CooMatrix *)h)->put_lagrange_block(*(CooMatrix *)bb); }
const char *rh_coo_to_dense(void *h, double *a, int64_t len) {
    std::vector<double> aa((size_t)std::max<int64_t>(len, 0));
    StrError e = ((CooMatrix *)h)->to_dense(aa);
    if (!e) std::copy(aa.begin(), aa.end(), a);
    return e;
}
int64_t rh_coo_actual_nnz(void *h) { return (int64_t)((CooMatrix *)h)->get_actual_nnz(); }
void *rh_csc_new(int64_t nrow, int64_t ncol, const int32_t *col_pointers, int64_t np, const int32_t *row_indices, const double *values, int64_t nv,
                 int32_t sym, const char **err) {
    CscMatrix *m = new CscMatrix();
    *err = CscMatrix::create(*m, (size_t)std::max<int64_t>(nrow, 0), (size_t)std::max<int64_t>(ncol, 0), std::vector<int32_t>(col_pointers, col_pointers + np),
                             std::vector<int32_t>(row_indices, row_indices + nv), std::vector<double>(values, values + nv), (Sym)sym);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}
void *rh_csr_new(int64_t nrow, int64_t ncol, const int32_t *row_pointers, int64_t np, const int32_t *col_indices, const double *values, int64_t nv,
                 int32_t sym, const char **err) {
    CsrMatrix *m = new CsrMatrix();
    *err = CsrMatrix::create(*m, (size_t)std::max<int64_t>(nrow, 0), (size_t)std::max<int64_t>(ncol, 0), std::vector<int32_t>(row_pointers, row_pointers + np),
                             std::vector<int32_t>(col_indices, col_indices + nv), std::vector<double>(values, values + nv), (Sym)sym);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}
const char *rh_csc_to_dense(void *h, double *a, int64_t len) {
    std::vector<double> aa((size_t)std::max<int64_t>(len, 0));
    StrError e = ((CscMatrix *)h)->to_dense(aa);
    if (!e) std::copy(aa.begin(), aa.end(), a);
    return e;
}
const char *rh_csr_to_dense(void *h, double *a, int64_t len) {
    std::vector<double> aa((size_t)std::max<int64_t>(len, 0));
    StrError e = ((CsrMatrix *)h)->to_dense(aa);
    if (!e) std::copy(aa.begin(), aa.end(), a);
    return e;
}
void *rh_csc_from_csr(void *csr, const char **err) {
    CscMatrix *m = new CscMatrix();
    *err = CscMatrix::from_csr(*m, *(CsrMatrix *)csr);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}
void *rh_csr_from_csc(void *csc, const char **err) {
    CsrMatrix *m = new CsrMatrix();
    *err = CsrMatrix::from_csc(*m, *(CscMatrix *)csc);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}

void *rh_csc_from_coo(void *coo, const char **err) {
    CscMatrix *m = new CscMatrix();
    *err = CscMatrix::from_coo(*m, *(CooMatrix *)coo);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}
const char *rh_csc_update_from_coo(void *h, void *coo) { return ((CscMatrix *)h)->update_from_coo(*(CooMatrix *)coo); }
void rh_csc_arrays(void *h, const int32_t **cp, const int32_t **ri, const double **vx, int64_t *ncol, int64_t *nnz) {
    CscMatrix *m = (CscMatrix *)h;
    *cp = m->col_pointers.data(), *ri = m->row_indices.data(), *vx = m->values.data(), *ncol = (int64_t)m->ncol, *nnz = (int64_t)m->nnz_final();
}
const char *rh_csc_mat_vec_mul(void *h, double *v, int64_t nv, double alpha, const double *u, int64_t nu) {
    std::vector<double> vv((size_t)nv), uu(u, u + nu);
    StrError e = ((CscMatrix *)h)->mat_vec_mul(vv, alpha, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}
void rh_csc_free(void *h) { delete (CscMatrix *)h; }

void *rh_csr_from_coo(void *coo, const char **err) {
    CsrMatrix *m = new CsrMatrix();
    *err = CsrMatrix::from_coo(*m, *(CooMatrix *)coo);
    if (*err) {
        delete m;
        return nullptr;
    }
    return m;
}
const char *rh_csr_update_from_coo(void *h, void *coo) { return ((CsrMatrix *)h)->update_from_coo(*(CooMatrix *)coo); }
void rh_csr_arrays(void *h, const int32_t **rp, const int32_t **cj, const double **vx, int64_t *nrow, int64_t *nnz) {
    CsrMatrix *m = (CsrMatrix *)h;
    *rp = m->row_pointers.data(), *cj = m->col_indices.data(), *vx = m->values.data(), *nrow = (int64_t)m->nrow, *nnz = (int64_t)m->nnz_final();
}
const char *rh_csr_mat_vec_mul(void *h, double *v, int64_t nv, double alpha, const double *u, int64_t nu) {
    std::vector<double> vv((size_t)nv), uu(u, u + nu);
    StrError e = ((CsrMatrix *)h)->mat_vec_mul(vv, alpha, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}
void rh_csr_free(void *h) { delete (CsrMatrix *)h; }

const char *rh_verify(void *coo, const double *x, int64_t nx, const double *rhs, int64_t nr, double *out4) {
    VerifyLinSys v;
    StrError e = VerifyLinSys::from(v, *(CooMatrix *)coo, std::vector<double>(x, x + nx), std::vector<double>(rhs, rhs + nr));
    if (!e) out4[0] = v.max_abs_a, out4[1] = v.max_abs_ax, out4[2] = v.max_abs_diff, out4[3] = v.relative_error;
    return e;
}

void *rh_read_matrix_market(const char *path, int32_t mmsym, const char **err) {
    CooMatrix *c = new CooMatrix();
    *err = read_matrix_market(*c, path, (MMsym)mmsym);
    if (*err) {
        delete c;
        return nullptr;
    }
    return c;
}

struct RhSolver {
    LinSolver ls;
    StatsLinSol stats;
    std::string json;
};

// bulk put (host-side convenience of the test / bench drivers: same checks as put, stops at the first error)
const char *rh_coo_put_many(void *c, int64_t n, const int32_t *ii, const int32_t *jj, const double *aa) {
    CooMatrix *m = (CooMatrix *)c;
    for (int64_t k = 0; k < n; k++) {
        if (ii[k] < 0 || jj[k] < 0) return "COO matrix: index of row is outside range";
        StrError e = m->put((size_t)ii[k], (size_t)jj[k], aa[k]);
        if (e) return e;
    }
    return nullptr;
}
const char *rh_ccoo_put_many(void *c, int64_t n, const int32_t *ii, const int32_t *jj, const double *reim) {
    ComplexCooMatrix *m = (ComplexCooMatrix *)c;
    for (int64_t k = 0; k < n; k++) {
        if (ii[k] < 0 || jj[k] < 0) return "COO matrix: index of row is outside range";
        StrError e = m->put((size_t)ii[k], (size_t)jj[k], reim[2 * k], reim[2 * k + 1]);
        if (e) return e;
    }
    return nullptr;
}
void *rh_ccoo_new(int64_t nrow, int64_t ncol, int64_t max_nnz, int32_t sym, const char **err) {
    ComplexCooMatrix *c = new ComplexCooMatrix();
    *err = ComplexCooMatrix::create(*c, (size_t)std::max<int64_t>(nrow, 0), (size_t)std::max<int64_t>(ncol, 0), (size_t)std::max<int64_t>(max_nnz, 0), (Sym)sym);
    if (*err) {
        delete c;
        return nullptr;
    }
    return c;
}
void rh_ccoo_free(void *c) { delete (ComplexCooMatrix *)c; }
const char *rh_ccoo_put(void *c, int64_t i, int64_t j, double re, double im) {
    if (i < 0 || j < 0) return "COO matrix: index of row is outside range";
    return ((ComplexCooMatrix *)c)->put((size_t)i, (size_t)j, re, im);
}
void rh_ccoo_reset(void *c) { ((ComplexCooMatrix *)c)->reset(); }
const char *rh_ccoo_mat_vec_mul(void *c, double *v, int64_t nv, double alpha_re, double alpha_im, const double *u, int64_t nu) {
    std::vector<double> vv((size_t)nv), uu(u, u + nu);
    StrError e = ((ComplexCooMatrix *)c)->mat_vec_mul(vv, alpha_re, alpha_im, uu);
    if (!e) std::copy(vv.begin(), vv.end(), v);
    return e;
}
struct RhComplexSolver {
    std::unique_ptr<ComplexSolverHIPMF> s;
};
void *rh_clinsolver_new(const char **err) {
    RhComplexSolver *h = new RhComplexSolver();
    *err = ComplexSolverHIPMF::create(h->s);
    if (*err) {
        delete h;
        return nullptr;
    }
    return h;
}
void rh_clinsolver_free(void *h) { delete (RhComplexSolver *)h; }
const char *rh_clinsolver_factorize(void *h, void *ccoo, const RhParams *params) {
    RhComplexSolver *s = (RhComplexSolver *)h;
    if (params) {
        LinSolParams p = to_params(params);
        return s->s->factorize(*(ComplexCooMatrix *)ccoo, &p);
    }
    return s->s->factorize(*(ComplexCooMatrix *)ccoo, nullptr);
}
const char *rh_clinsolver_solve(void *h, double *x, int64_t nx, const double *rhs, int64_t nr, int32_t verbose) {
    RhComplexSolver *s = (RhComplexSolver *)h;
    std::vector<double> xx((size_t)nx), rr(rhs, rhs + nr);
    StrError e = s->s->solve(xx, rr, verbose != 0);
    if (!e) std::copy(xx.begin(), xx.end(), x);
    return e;
}

void rh_clinsolver_outputs(void *h, double *det_re, double *det_im, double *det_exp, double *rcond, int32_t *npert) {
    RhComplexSolver *s = (RhComplexSolver *)h;
    s->s->get_determinant(*det_re, *det_im, *det_exp);
    *rcond = s->s->get_rcond(), *npert = s->s->get_perturbed_pivots();
}

void *rh_linsolver_new(int32_t genie, const char **err) {
    RhSolver *s = new RhSolver();
    *err = LinSolver::create(s->ls, (Genie)genie);
    if (*err) {
        delete s;
        return nullptr;
    }
    return s;
}
void rh_linsolver_free(void *h) { delete (RhSolver *)h; }
const char *rh_linsolver_factorize(void *h, void *coo, const RhParams *params) {
    RhSolver *s = (RhSolver *)h;
    if (params) {
        LinSolParams p = to_params(params);
        return s->ls.actual->factorize(*(CooMatrix *)coo, &p);
    }
    return s->ls.actual->factorize(*(CooMatrix *)coo, nullptr);
}
const char *rh_linsolver_solve(void *h, double *x, int64_t nx, const double *rhs, int64_t nr, int32_t verbose) {
    RhSolver *s = (RhSolver *)h;
    if (SolverHIPMF *a = dynamic_cast<SolverHIPMF *>(s->ls.actual.get())) return a->solve_slices(x, (size_t)nx, rhs, (size_t)nr, verbose != 0);
    std::vector<double> xx((size_t)nx), rr(rhs, rhs + nr);
    StrError e = s->ls.actual->solve(xx, rr, verbose != 0);
    if (!e) std::copy(xx.begin(), xx.end(), x);
    return e;
}
const char *rh_linsolver_solve_many(void *h, double *x, const double *rhs, int64_t n, int64_t nrhs) {
    RhSolver *s = (RhSolver *)h;
    SolverHIPMF *a = dynamic_cast<SolverHIPMF *>(s->ls.actual.get());
    if (!a) return "solve_many is only available with Genie::Hipmf";
    std::vector<double> xx((size_t)(n * nrhs)), rr(rhs, rhs + n * nrhs);
    StrError e = a->solve_many(xx, rr, (size_t)nrhs);
    if (!e) std::copy(xx.begin(), xx.end(), x);
    return e;
}
void rh_linsolver_times(void *h, uint64_t *ns3) {
    RhSolver *s = (RhSolver *)h;
    ns3[0] = s->ls.actual->get_ns_init(), ns3[1] = s->ls.actual->get_ns_fact(), ns3[2] = s->ls.actual->get_ns_solve();
}
void rh_linsolver_outputs(void *h, double *det_coef, double *det_exp, double *rcond, int32_t *eff_ordering, int32_t *eff_scaling, int32_t *npert) {
    RhSolver *s = (RhSolver *)h;
    SolverHIPMF *a = dynamic_cast<SolverHIPMF *>(s->ls.actual.get());
    if (!a) return;
    *det_coef = a->determinant_coefficient, *det_exp = a->determinant_exponent, *rcond = a->rcond_estimate;
    *eff_ordering = a->effective_ordering, *eff_scaling = a->effective_scaling, *npert = a->perturbed_pivots;
}
const char *rh_linsolver_stats_json(void *h, void *coo, const char *name, const double *x, const double *rhs) {
    RhSolver *s = (RhSolver *)h;
    s->stats = StatsLinSol();
    s->ls.actual->update_stats(s->stats);
    if (coo) {
        CooMatrix *c = (CooMatrix *)coo;
        s->stats.matrix_name = name ? name : "";
        s->stats.nrow = c->nrow, s->stats.ncol = c->ncol, s->stats.nnz = c->nnz;
        s->stats.symmetric = sym_name(c->symmetric);
        if (x && rhs) VerifyLinSys::from(s->stats.verify, *c, std::vector<double>(x, x + c->ncol), std::vector<double>(rhs, rhs + c->nrow));
    }
    s->json = s->stats.to_json();
    return s->json.c_str();
}
const char *rh_error_string(int32_t code) { return handle_hipmf_error_code(code); }
void rh_format_nanoseconds(uint64_t nanoseconds, char *buf, int32_t len) {
    if (!buf || len < 1) return;
    const std::string t = format_nanoseconds(nanoseconds);
    snprintf(buf, (size_t)len, "%s", t.c_str());
}
int32_t rh_is_memory_error(const char *message) { return is_memory_error(message) ? 1 : 0; }
void rh_ccoo_info(void *h, int64_t *nrow, int64_t *ncol, int64_t *nnz, int64_t *max_nnz, int32_t *sym) {
    ComplexCooMatrix *c = (ComplexCooMatrix *)h;
    *nrow = (int64_t)c->nrow, *ncol = (int64_t)c->ncol, *nnz = (int64_t)c->nnz, *max_nnz = (int64_t)c->max_nnz, *sym = (int32_t)c->symmetric;
}
void rh_ccoo_arrays(void *h, const int32_t **ai, const int32_t **aj, const double **ax) {
    ComplexCooMatrix *c = (ComplexCooMatrix *)h;
    *ai = c->indices_i.data(), *aj = c->indices_j.data(), *ax = c->values.data();
}
const char *rh_read_matrix_market_any(const char *path, int32_t mmsym, void **coo, void **ccoo) {
    *coo = nullptr, *ccoo = nullptr;
    MatrixMarketData data;
    StrError e = read_matrix_market(data, path, (MMsym)mmsym);
    if (e) return e;
    if (data.complex) *ccoo = new ComplexCooMatrix(std::move(data.complex_matrix));
    else *coo = new CooMatrix(std::move(data.real));
    return nullptr;
}
const char *rh_enum_name(int32_t which, int32_t value) {
    if (which == 0 && value >= 0 && value < 12) return ORDERING_NAMES[value];
    if (which == 1 && value >= 0 && value < 9) return SCALING_NAMES[value];
    if (which == 2) return genie_to_string((Genie)value);
    if (which == 3) return sym_name((Sym)value);
    return "";
}
int32_t rh_genie_get_sym(int32_t genie, int32_t symmetric) { return (int32_t)genie_get_sym((Genie)genie, symmetric != 0); }

} // extern "C"

// russell_host.hpp -- C++ mirror of the Rust host layer of russell_sparse for the solver path.
//
// The reference's host layer is Rust (no Rust toolchain in this image), so the layer that sits ABOVE the
// C-ABI of include/russell_hipmf.h is written here in C++ with the same names, argument meaning and error
// strings, so that a russell_sparse user finds what the trait boundary promises:
//   CooMatrix::new/put/reset/mat_vec_mul      /root/reference/russell_sparse/src/coo_matrix.rs:173-196,324-354,388,547-566
//   CscMatrix::from_coo/update_from_coo       .../csc_matrix.rs:337-356,365-505 ; mat_vec_mul :735-755
//   CsrMatrix::from_coo/update_from_coo       .../csr_matrix.rs:332-351,359-480 ; mat_vec_mul :709-729
//   LinSolParams                              .../lin_sol_params.rs:5-107
//   Genie / Sym / Ordering / Scaling / MMsym  .../enums.rs:5-20,27-39,45-66,71-155,159-222,334-366
//   LinSolTrait, LinSolver::new / compute     .../lin_solver.rs:12-64,116-142,212-224
//   SolverHIPMF (the new backend)             modelled on .../solver_cudss.rs:92-131,194-360,501-558
//   VerifyLinSys::from                        .../verify_lin_sys.rs:60-96
//   StatsLinSol (JSON)                        .../stats_lin_sol.rs:14-113,212-340
//   read_matrix_market                        .../read_matrix_market.rs:44-184,346-475
// Errors are `StrError` = static C strings (nullptr means Ok), the analogue of Result<(), &'static str>.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace russell {

typedef const char *StrError;

enum class Sym : int32_t { No = 0, YesFull = 1, YesLower = 2, YesUpper = 3 };
enum class Genie : int32_t { Hipmf = 0, Umfpack = 1, Mumps = 2, Cudss = 3 };
enum class Ordering : int32_t { Amd = 0, Amf, Auto, Best, BtfColamd, Cholmod, Colamd, Metis, No, Pord, Qamd, Scotch };
enum class Scaling : int32_t { Auto = 0, Column, Diagonal, Max, No, RowCol, RowColIter, RowColRig, Sum };
enum class MMsym : int32_t { LeaveAsLower = 0, SwapToUpper = 1, MakeItFull = 2 };

const char *genie_to_string(Genie g);
Genie genie_from(const std::string &name); // default Hipmf here (the reference defaults to umfpack, enums.rs:336-343)
Sym genie_get_sym(Genie g, bool symmetric); // enums.rs:355-365; Hipmf follows the cuDSS rule (YesLower)

// enums.rs:159-332 (the variants the GPU plug-ins read)
enum class Matching : int32_t { None = 0, Auto, MaxDiagCount, MaxMinDiag, MaxMinDiagAlt, MaxDiagSum, MaxDiagProduct };
enum class Pivoting : int32_t { Auto = 0, None, GlobalCol, GlobalRow, Diagonal, LocalBlock };

// lin_sol_params.rs:5-82; the MUMPS / UMFPACK-only fields have no meaning for this backend and are not mirrored.
//   ordering   Ordering::No -> natural order; Amd / Amf / Qamd -> this backend's approximate minimum degree; Best -> BOTH orderings, the one
//              whose column counts promise fewer factorisation flops is kept (UMFPACK's meaning of the word, solver_umfpack.rs:461; on
//              complex handles -- paired rows -- the nested dissection alone); every other variant (Auto, Cholmod, Colamd, Metis, Pord,
//              Scotch) -> its nested dissection with dense leaves (the reference maps variants a backend does not have to that
//              backend's default the same way, solver_umfpack.rs:457-472); the effective ordering ("No" / "Amd" / "Nd"; for Best: the
//              winner) is reported by update_stats
//   matching   None -> never; Auto -> when the diagonal is weak; any named variant -> always (there is one matching: maximum product
//              + scaling, what cuDSS calls MaxDiagProduct)
//   pivoting   a REQUEST, as for cuDSS (solver_cudss.rs:233,298: the effective strategy comes back from factorize): every value is accepted,
//              what runs is partial pivoting inside the pivot block (LocalBlock) + replaced pivots + Krylov rescue; StatsLinSol reports it
//   hybrid_memory_factor  recorded; a factor that does not fit HBM is refused with the "Not enough memory" string the reference's
//              harness recognises (stats_lin_sol.rs:334-340)
//   compute_error_estimates / compute_condition_numbers  always available from the actual solver (get_error_estimate / rcond_estimate)
struct LinSolParams {
    Ordering ordering = Ordering::Auto;
    Scaling scaling = Scaling::Auto;
    Matching matching = Matching::Auto;
    Pivoting pivoting = Pivoting::Auto;
    bool has_pivot_epsilon = false;
    double pivot_epsilon = 0.0;
    bool has_refinement_nstep = false;
    int32_t refinement_nstep = 0;
    bool has_hybrid_memory_factor = false;
    double hybrid_memory_factor = 0.0;
    bool positive_definite = false;
    bool compute_determinant = false;
    bool compute_error_estimates = false;
    bool compute_condition_numbers = false;
    bool verbose = false;
};

struct CooMatrix {
    Sym symmetric = Sym::No;
    size_t nrow = 0, ncol = 0, nnz = 0, max_nnz = 0;
    std::vector<int32_t> indices_i, indices_j;
    std::vector<double> values;
    static StrError create(CooMatrix &out, size_t nrow, size_t ncol, size_t max_nnz, Sym symmetric);
    // coo_matrix.rs:246-291 (NumCooMatrix::from): adopt ready triplet arrays (nnz = max_nnz = their length)
    static StrError from(CooMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> row_indices, std::vector<int32_t> col_indices,
                         std::vector<double> values, Sym symmetric);
    StrError put(size_t i, size_t j, double aij);
    void reset() { nnz = 0; }
    StrError mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const;
    // coo_matrix.rs:629-652 (v += alpha A u), :708-730 (v = alpha A^T u)
    StrError mat_vec_mul_update(std::vector<double> &v, double alpha, const std::vector<double> &u) const;
    StrError mat_t_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const;
    // coo_matrix.rs:738-757 (this = alpha other, triplets replaced), :779-797 (this += alpha other: triplets appended, as Radau5 /
    // BwEuler build K = gamma M - J on a fixed structure), :823-857 (Lagrange-multiplier block B / B^T)
    StrError assign(double alpha, const CooMatrix &other);
    StrError add(double alpha, const CooMatrix &other);
    StrError put_lagrange_block(const CooMatrix &bb);
    // coo_matrix.rs:468-497: dense row-major nrow x ncol copy (duplicates summed, triangular storage mirrored)
    StrError to_dense(std::vector<double> &a) const;
    size_t get_actual_nnz() const; // coo_matrix.rs:872-887
};

struct CsrMatrix;

struct CscMatrix {
    Sym symmetric = Sym::No;
    size_t nrow = 0, ncol = 0;
    std::vector<int32_t> col_pointers, row_indices;
    std::vector<double> values;
    // csc_matrix.rs:197-262: validated constructor from ready arrays (pointers ascending, indices in range and ascending per column)
    static StrError create(CscMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> col_pointers, std::vector<int32_t> row_indices,
                           std::vector<double> values, Sym symmetric);
    StrError to_dense(std::vector<double> &a_row_major) const; // csc_matrix.rs:702-729 (triangular storage mirrored)
    static StrError from_coo(CscMatrix &out, const CooMatrix &coo);
    static StrError from_csr(CscMatrix &out, const CsrMatrix &csr); // csc_matrix.rs:508-584
    StrError update_from_coo(const CooMatrix &coo);
    StrError mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const;
    size_t nnz_final() const { return col_pointers.empty() ? 0 : (size_t)col_pointers[ncol]; }

  private:
    std::vector<int32_t> temp_rp, temp_rj, temp_w;
    std::vector<double> temp_rx;
    std::vector<size_t> temp_rc;
};

struct CsrMatrix {
    Sym symmetric = Sym::No;
    size_t nrow = 0, ncol = 0;
    std::vector<int32_t> row_pointers, col_indices;
    std::vector<double> values;
    // csr_matrix.rs:193-257
    static StrError create(CsrMatrix &out, size_t nrow, size_t ncol, std::vector<int32_t> row_pointers, std::vector<int32_t> col_indices,
                           std::vector<double> values, Sym symmetric);
    StrError to_dense(std::vector<double> &a_row_major) const; // csr_matrix.rs:676-703
    static StrError from_coo(CsrMatrix &out, const CooMatrix &coo);
    static StrError from_csc(CsrMatrix &out, const CscMatrix &csc); // csr_matrix.rs:483-558
    StrError update_from_coo(const CooMatrix &coo);
    StrError mat_vec_mul(std::vector<double> &v, double alpha, const std::vector<double> &u) const;
    size_t nnz_final() const { return row_pointers.empty() ? 0 : (size_t)row_pointers[nrow]; }

  private:
    std::vector<int32_t> temp_rp, temp_w;
    std::vector<std::pair<int32_t, double>> temp_rjx;
    std::vector<size_t> temp_rc;
};

struct ComplexCooMatrix;

struct VerifyLinSys {
    double max_abs_a = 0, max_abs_ax = 0, max_abs_diff = 0, relative_error = 0;
    static StrError from(VerifyLinSys &out, const CooMatrix &mat, const std::vector<double> &x, const std::vector<double> &rhs);
    // verify_lin_sys.rs:103-143: x, rhs interleaved complex vectors; absolute values are complex moduli
    static StrError from_complex(VerifyLinSys &out, const ComplexCooMatrix &mat, const std::vector<double> &x, const std::vector<double> &rhs);
    // verify_lin_sys.rs:146-152
    VerifyLinSys max_relative_error(const VerifyLinSys &other) const { return other.relative_error > relative_error ? other : *this; }
};

// russell_lab base/formatters.rs:60-95: "250ns", "2.5µs", "25ms", "2.5s", "4m10s", "1h2m3s", "1h100.001µs"
std::string format_nanoseconds(uint64_t nanoseconds);

// stats_lin_sol.rs:334-340: does the error message indicate an out-of-memory condition?
bool is_memory_error(const char *message);

struct StatsLinSol {
    std::string solver = "Unknown", matrix_name = "Unknown", symmetric = "Unknown", ordering = "Unknown", scaling = "Unknown",
                matching = "Unknown", effective_ordering = "Unknown", effective_scaling = "Unknown", effective_matching = "Unknown";
    // round 6: the fields of stats_lin_sol.rs:36-60 the mirror used to leave out (requests.pivoting / mumps_num_threads /
    // hybrid_memory_factor, output.effective_pivoting / effective_mumps_num_threads / openmp_num_threads / umfpack_strategy /
    // umfpack_rcond_estimate) and the mumps_stats block (:100-113): a consumer of the reference's JSON finds every key
    std::string pivoting = "Unknown", effective_pivoting = "Unknown", umfpack_strategy = "Unknown";
    bool has_hybrid_memory_factor = false;
    double hybrid_memory_factor = 0.0;
    size_t nrow = 0, ncol = 0, nnz = 0, nnz_actual = 0;
    bool complex = false, positive_definite = false, out_of_memory = false;
    double rcond_estimate = 0.0, det_mantissa = 0.0, det_mantissa_imag = 0.0, det_base = 0.0, det_exponent = 0.0;
    int32_t perturbed_pivots = 0;
    VerifyLinSys verify;
    uint64_t read_matrix_ns = 0, verify_ns = 0;
    std::vector<uint64_t> initialize_ns, factorize_ns, solve_ns;
    // stats_lin_sol.rs:212-233
    void set_matrix_name_from_path(const std::string &filepath);
    void set_matrix_info_from_coo(const CooMatrix &coo);
    void set_matrix_info_from_coo(const ComplexCooMatrix &coo);
    // field names follow stats_lin_sol.rs:14-113 (main, matrix, requests, output, determinant, verify, time_human,
    // time_nanoseconds); averages and total_ifs as compute_derived_values does (:277-326)
    std::string to_json(bool pretty = false) const;
};

class LinSolTrait {
  public:
    virtual ~LinSolTrait() {}
    virtual StrError factorize(const CooMatrix &mat, const LinSolParams *params) = 0;
    virtual StrError solve(std::vector<double> &x, const std::vector<double> &rhs, bool verbose) = 0;
    virtual void update_stats(StatsLinSol &stats) const = 0;
    virtual uint64_t get_ns_init() const = 0;
    virtual uint64_t get_ns_fact() const = 0;
    virtual uint64_t get_ns_solve() const = 0;
};

// The new backend: COO -> CSR on the host, then the three C-ABI phases of include/russell_hipmf.h.
class SolverHIPMF : public LinSolTrait {
  public:
    static StrError create(std::unique_ptr<SolverHIPMF> &out);
    ~SolverHIPMF() override;
    StrError factorize(const CooMatrix &mat, const LinSolParams *params) override;
    StrError solve(std::vector<double> &x, const std::vector<double> &rhs, bool verbose) override;
    StrError solve_slices(double *x, size_t nx, const double *rhs, size_t nr, bool verbose); // (borrowed slices: no copies)
    void update_stats(StatsLinSol &stats) const override;
    uint64_t get_ns_init() const override { return time_initialize_ns; }
    uint64_t get_ns_fact() const override { return time_factorize_ns; }
    uint64_t get_ns_solve() const override { return time_solve_ns; }
    // extension: many right-hand sides, column-major n x nrhs
    StrError solve_many(std::vector<double> &x, const std::vector<double> &rhs, size_t nrhs);
    // extension: the backend's diagnostic counters (HIPMF_COUNTER_* of include/russell_hipmf.h; -1 before the first factorize)
    int64_t get_counter(int32_t which) const;

    bool factorized = false;
    bool value_map_set = false, first_call = false; // repeat factorizations refresh the values on the device through a map
    std::vector<int32_t> map_i, map_j;              // the triplet indices the map was built from (a repeat call with other triplets rebuilds it)
    int32_t effective_ordering = -1, effective_scaling = -1, perturbed_pivots = 0;
    bool effective_matching = false; // a maximum-product matching pre-permutation is in force (weak diagonal at initialize)
    double rcond_estimate = 0.0, determinant_coefficient = 0.0, determinant_exponent = 0.0;

  private:
    SolverHIPMF() {}
    void *solver = nullptr; // struct InterfaceHIPMF*
    CsrMatrix csr;
    bool initialized = false;
    Sym initialized_sym = Sym::No;
    size_t initialized_ndim = 0, initialized_nnz = 0;
    uint64_t time_initialize_ns = 0, time_factorize_ns = 0, time_solve_ns = 0;
    bool compute_determinant = false;
};

// ---- complex twin (ComplexLinSolTrait, complex_lin_solver.rs:12-104; ComplexCooMatrix, complex_coo_matrix.rs) ----
// Values are interleaved (re, im) pairs like the reference's Complex64 arrays (constants.h:18).  The backend solves the
// REAL-EQUIVALENT system of order 2n with unknowns (Re z_k, Im z_k) interleaved,
//     a + ib at (i, j)   ->   [ a  -b ; b  a ] at rows 2i, 2i+1 / columns 2j, 2j+1,
// on the same device path as the real solver (a native complex kernel set is a "next" row of DESIGN.md).
struct ComplexCooMatrix {
    Sym symmetric = Sym::No;
    size_t nrow = 0, ncol = 0, nnz = 0, max_nnz = 0;
    std::vector<int32_t> indices_i, indices_j;
    std::vector<double> values; // 2 * max_nnz: re, im
    static StrError create(ComplexCooMatrix &out, size_t nrow, size_t ncol, size_t max_nnz, Sym symmetric);
    StrError put(size_t i, size_t j, double re, double im);
    void reset() { nnz = 0; }
    // v := alpha * A * u with interleaved complex vectors (complex_coo_matrix.rs mat_vec_mul; symmetric storage mirrored)
    StrError mat_vec_mul(std::vector<double> &v, double alpha_re, double alpha_im, const std::vector<double> &u) const;
};

class ComplexSolverHIPMF {
  public:
    static StrError create(std::unique_ptr<ComplexSolverHIPMF> &out);
    ~ComplexSolverHIPMF();
    // complex_solver_umfpack.rs:232-329 with this backend's symmetry rule (Sym::No or Sym::YesLower): COO -> complex CSR on the host at
    // the first call (duplicates summed in COO order, as complex_csr_matrix from_coo does), then the complex C-ABI
    // (complex_solver_hipmf_*, include/russell_hipmf.h); repeat calls hand the raw triplet values to the device-side value refresh
    StrError factorize(const ComplexCooMatrix &mat, const LinSolParams *params);
    // x, rhs: interleaved complex vectors of length 2 n
    StrError solve(std::vector<double> &x, const std::vector<double> &rhs, bool verbose);
    bool factorized = false;
    void update_stats(StatsLinSol &stats) const; // complex_lin_solver.rs:12-104 (ComplexLinSolTrait::update_stats)
    uint64_t get_ns_init() const { return time_initialize_ns; }
    uint64_t get_ns_fact() const { return time_factorize_ns; }
    uint64_t get_ns_solve() const { return time_solve_ns; }
    // determinant = (re + i im) x 10^exponent when LinSolParams.compute_determinant was set (complex_solver_umfpack.rs:154-167, 411-414)
    void get_determinant(double &re, double &im, double &exponent) const { re = determinant_coefficient_real, im = determinant_coefficient_imag, exponent = determinant_exponent; }
    double get_rcond() const { return rcond_estimate; }
    int32_t get_perturbed_pivots() const { return perturbed_pivots; }
    int64_t get_counter(int32_t which) const; // (HIPMF_COUNTER_* of include/russell_hipmf.h)

  private:
    ComplexSolverHIPMF() {}
    void *solver = nullptr; // struct InterfaceComplexHIPMF*
    bool initialized = false, value_map_set = false;
    Sym initialized_sym = Sym::No;
    size_t initialized_ndim = 0, initialized_nnz = 0;
    std::vector<int32_t> map_i, map_j; // the triplet indices the value map was built from
    std::vector<int32_t> zrp, zci, seg_ptr, seg_idx;
    std::vector<double> zvals;          // summed CSR values (first call / fallback when the triplet order changes)
    int32_t effective_ordering = -1, effective_scaling = -1, perturbed_pivots = 0;
    double rcond_estimate = 0.0, determinant_coefficient_real = 0.0, determinant_coefficient_imag = 0.0, determinant_exponent = 0.0;
    bool compute_determinant = false;
    bool effective_matching = false;
    uint64_t time_initialize_ns = 0, time_factorize_ns = 0, time_solve_ns = 0;
    StrError to_csr(const ComplexCooMatrix &mat, bool pattern_too);
};

class LinSolver {
  public:
    std::unique_ptr<LinSolTrait> actual;
    static StrError create(LinSolver &out, Genie genie);
    // lin_solver.rs:212-224: allocate, factorize, solve in one go
    static StrError compute(Genie genie, std::vector<double> &x, const CooMatrix &mat, const std::vector<double> &rhs,
                            const LinSolParams *params);
};

// read_matrix_market.rs:346-475 returns (Option<CooMatrix>, Option<ComplexCooMatrix>): exactly one of the two is filled
struct MatrixMarketData {
    bool complex = false;
    CooMatrix real;
    ComplexCooMatrix complex_matrix;
};
StrError read_matrix_market(MatrixMarketData &out, const std::string &full_path, MMsym symmetric_handling);
StrError read_matrix_market(CooMatrix &out, const std::string &full_path, MMsym symmetric_handling); // real files only

StrError handle_hipmf_error_code(int32_t err);

// where the HIP library is looked for: $RUSSELL_HIPMF_LIB, else librussell_hipmf.so next to this library
void set_hipmf_library_path(const std::string &path);

} // namespace russell

// solve_matrix_market -- the benchmark harness of russell_sparse for the HIPMF backend.
//
// Mirrors /root/reference/russell_sparse/src/bin/solve_matrix_market.rs:10-305: read a MatrixMarket file,
// rhs = ones (complex: 1 + 1i), LinSolver::new(genie) -> factorize -> solve -> update_stats, VerifyLinSys,
// print the StatsLinSol JSON; `--nrun` repeats with a fresh solver and keeps the largest error; an out-of-memory
// factorize prints the JSON with main.out_of_memory = true and exits 0; the bfwb62 solution is checked at 1e-10.
// Options of the other backends that have no meaning here (MUMPS threads, cuDSS hybrid memory, vismatrix output,
// error estimates, condition numbers, UMFPACK's strategy switch) are accepted and ignored so that the reference's
// sweep scripts run unchanged.
//
// usage: solve_matrix_market [-g hipmf] [-o ORDERING] [-s SCALING] [-p] [-d] [-v] [-r NRUN] [--hide-json] FILE.mtx
#include "russell_host.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace russell;

namespace {

uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Options {
    std::string matrix_market_file, genie = "hipmf", ordering = "Auto", scaling = "Auto", matching_sym = "None", matching_gen = "Auto";
    bool positive_definite = false, verbose = false, determinant = false, hide_json = false;
    long nrun = 1;
    bool has_hybrid = false;
    double hybrid = 0.0;
};

// enums.rs:45-66,159-222 (Ordering::from / Scaling::from): case-insensitive names, anything else -> Auto
Ordering ordering_from(std::string s) {
    for (auto &c : s) c = (char)tolower(c);
    const std::pair<const char *, Ordering> names[] = {{"amd", Ordering::Amd},       {"amf", Ordering::Amf},         {"auto", Ordering::Auto},
                                                       {"best", Ordering::Best},     {"btfcolamd", Ordering::BtfColamd}, {"cholmod", Ordering::Cholmod},
                                                       {"colamd", Ordering::Colamd}, {"metis", Ordering::Metis},     {"no", Ordering::No},
                                                       {"pord", Ordering::Pord},     {"qamd", Ordering::Qamd},       {"scotch", Ordering::Scotch}};
    for (auto &n : names)
        if (s == n.first) return n.second;
    return Ordering::Auto;
}
Scaling scaling_from(std::string s) {
    for (auto &c : s) c = (char)tolower(c);
    const std::pair<const char *, Scaling> names[] = {{"auto", Scaling::Auto},     {"column", Scaling::Column},       {"diagonal", Scaling::Diagonal},
                                                      {"max", Scaling::Max},       {"no", Scaling::No},               {"rowcol", Scaling::RowCol},
                                                      {"rowcoliter", Scaling::RowColIter}, {"rowcolrig", Scaling::RowColRig}, {"sum", Scaling::Sum}};
    for (auto &n : names)
        if (s == n.first) return n.second;
    return Scaling::Auto;
}
const char *ordering_name(Ordering o) {
    static const char *names[] = {"Amd", "Amf", "Auto", "Best", "BtfColamd", "Cholmod", "Colamd", "Metis", "No", "Pord", "Qamd", "Scotch"};
    return names[(int)o];
}
const char *scaling_name(Scaling s) {
    static const char *names[] = {"Auto", "Column", "Diagonal", "Max", "No", "RowCol", "RowColIter", "RowColRig", "Sum"};
    return names[(int)s];
}

int usage(const char *msg) {
    if (msg) fprintf(stderr, "error: %s\n", msg);
    fprintf(stderr, "usage: solve_matrix_market [-g GENIE] [-o ORDERING] [-s SCALING] [-p] [-d] [-v] [-r NRUN] [--hide-json] FILE.mtx\n");
    return 2;
}

// golden solution of bfwb62 with rhs = ones (data: tests/golden/bfwb62_x.json holds the same 62 numbers)
#include "bfwb62_correct_x.inc"

} // namespace

int main(int argc, char **argv) {
    Options opt;
    for (int a = 1; a < argc; a++) {
        const std::string arg = argv[a];
        auto value = [&](std::string &dst) {
            if (a + 1 >= argc) return false;
            dst = argv[++a];
            return true;
        };
        std::string ignored;
        if (arg == "-g" || arg == "--genie") {
            if (!value(opt.genie)) return usage("missing value");
        } else if (arg == "-o" || arg == "--ordering") {
            if (!value(opt.ordering)) return usage("missing value");
        } else if (arg == "-s" || arg == "--scaling") {
            if (!value(opt.scaling)) return usage("missing value");
        } else if (arg == "--matching-sym") {
            if (!value(opt.matching_sym)) return usage("missing value");
        } else if (arg == "--matching-gen") {
            if (!value(opt.matching_gen)) return usage("missing value");
        } else if (arg == "-r" || arg == "--nrun") {
            std::string v;
            if (!value(v)) return usage("missing value");
            opt.nrun = strtol(v.c_str(), nullptr, 10);
            if (opt.nrun < 1) return usage("nrun must be >= 1");
        } else if (arg == "-h" || arg == "--hybrid-memory-factor" || arg == "-m" || arg == "--mumps-nt" || arg == "-n" || arg == "--nt") {
            if (!value(ignored)) return usage("missing value");
            if (arg == "-h" || arg == "--hybrid-memory-factor") {
                const double v = strtod(ignored.c_str(), nullptr);
                if (v < 0.01 || v > 0.99) {
                    fprintf(stderr, "hybrid memory factor must be in [0.01, 0.99]\n");
                    return 1;
                }
                opt.has_hybrid = true, opt.hybrid = v; // (solve_matrix_market.rs:132-138: handed to the solver and recorded in the requests)
            }
        } else if (arg == "-p" || arg == "--positive-definite") opt.positive_definite = true;
        else if (arg == "-v" || arg == "--verbose") opt.verbose = true;
        else if (arg == "-d" || arg == "--determinant") opt.determinant = true;
        else if (arg == "--hide-json") opt.hide_json = true;
        else if (arg == "-x" || arg == "--error-estimates" || arg == "-y" || arg == "--condition-numbers" || arg == "-u" ||
                 arg == "--enforce-unsymmetric-strategy" || arg == "--vismatrix" || arg == "--override-prevent-issue") {
            // options of UMFPACK / MUMPS: nothing to switch in this backend
        } else if (!arg.empty() && arg[0] == '-') return usage(("unknown option " + arg).c_str());
        else opt.matrix_market_file = arg;
    }
    if (opt.matrix_market_file.empty()) return usage("the MatrixMarket file is missing");

    const Genie genie = genie_from(opt.genie);
    // solve_matrix_market.rs:109-114: the storage the backend wants for symmetric files (Hipmf: lower, like cuDSS / MUMPS)
    const MMsym handling = genie == Genie::Umfpack ? MMsym::MakeItFull : MMsym::LeaveAsLower;

    LinSolParams params;
    params.ordering = ordering_from(opt.ordering);
    params.scaling = scaling_from(opt.scaling);
    params.positive_definite = opt.positive_definite;
    params.compute_determinant = opt.determinant;
    params.verbose = opt.verbose;

    StatsLinSol stats;
    stats.solver = genie_to_string(genie);
    stats.ordering = ordering_name(params.ordering);
    stats.scaling = scaling_name(params.scaling);
    stats.positive_definite = params.positive_definite;
    stats.pivoting = "Auto"; // (lin_sol_params.rs:16 default; the reference's harness has no switch for it either)
    if (opt.has_hybrid) {
        params.has_hybrid_memory_factor = true, params.hybrid_memory_factor = opt.hybrid;
        stats.has_hybrid_memory_factor = true, stats.hybrid_memory_factor = opt.hybrid;
    }

    uint64_t t0 = now_ns();
    MatrixMarketData data;
    if (StrError e = read_matrix_market(data, opt.matrix_market_file, handling)) {
        fprintf(stderr, "Error: %s\n", e);
        return 1;
    }
    stats.read_matrix_ns = now_ns() - t0;
    stats.set_matrix_name_from_path(opt.matrix_market_file);
    // the matching of this backend is automatic (weak diagonal at initialize); the request strings are recorded as given
    const bool symmetric = data.complex ? data.complex_matrix.symmetric != Sym::No : data.real.symmetric != Sym::No;
    stats.matching = symmetric ? opt.matching_sym : opt.matching_gen;

    auto fail = [&](StrError e) {
        if (is_memory_error(e)) {
            stats.out_of_memory = true;
            if (!opt.hide_json) printf("%s\n", stats.to_json(true).c_str());
            return 0;
        }
        fprintf(stderr, "Error: %s\n", e);
        return 1;
    };

    if (!data.complex) {
        const CooMatrix &coo = data.real;
        stats.set_matrix_info_from_coo(coo);
        if (coo.nrow != coo.ncol) return fail("the matrix must be square");
        std::vector<double> x(coo.nrow, 0.0), rhs(coo.nrow, 1.0);
        for (long run = 0; run < opt.nrun; run++) {
            LinSolver solver;
            if (StrError e = LinSolver::create(solver, genie)) return fail(e);
            if (StrError e = solver.actual->factorize(coo, &params)) return fail(e);
            if (StrError e = solver.actual->solve(x, rhs, opt.verbose)) return fail(e);
            solver.actual->update_stats(stats);
            t0 = now_ns();
            VerifyLinSys verify;
            if (StrError e = VerifyLinSys::from(verify, coo, x, rhs)) return fail(e);
            stats.verify_ns = now_ns() - t0;
            stats.verify = run == 0 ? verify : stats.verify.max_relative_error(verify);
            if (stats.matrix_name == "bfwb62") {
                for (size_t i = 0; i < coo.nrow && i < 62; i++) {
                    const double diff = std::fabs(x[i] - BFWB62_CORRECT_X[i]);
                    if (diff > 1e-10) printf("BFWB62 FAILED WITH NUMERICAL ERROR = %.2e @ %zu COMPONENT\n", diff, i);
                }
            }
        }
    } else {
        const ComplexCooMatrix &coo = data.complex_matrix;
        stats.set_matrix_info_from_coo(coo);
        if (coo.nrow != coo.ncol) return fail("the matrix must be square");
        std::vector<double> x(2 * coo.nrow, 0.0), rhs(2 * coo.nrow, 1.0); // cpx!(1.0, 1.0) in every component
        for (long run = 0; run < opt.nrun; run++) {
            std::unique_ptr<ComplexSolverHIPMF> solver;
            if (genie != Genie::Hipmf) return fail("only the HIPMF backend is available");
            if (StrError e = ComplexSolverHIPMF::create(solver)) return fail(e);
            LinSolParams zparams = params;
            if (StrError e = solver->factorize(coo, &zparams)) return fail(e);
            if (StrError e = solver->solve(x, rhs, opt.verbose)) return fail(e);
            solver->update_stats(stats);
            t0 = now_ns();
            VerifyLinSys verify;
            if (StrError e = VerifyLinSys::from_complex(verify, coo, x, rhs)) return fail(e);
            stats.verify_ns = now_ns() - t0;
            stats.verify = run == 0 ? verify : stats.verify.max_relative_error(verify);
        }
    }
    if (!opt.hide_json) printf("%s\n", stats.to_json(true).c_str());
    return 0;
}

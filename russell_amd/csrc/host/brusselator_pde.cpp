// brusselator_pde.cpp -- BASELINE config 5 end to end: the Brusselator PDE in 2D integrated by Radau5 (Radau IIA, order 5) with
// the repeat-factorise pattern of the reference, on the HIPMF backend.  Own restatement, in C++ above the host mirror of
// russell_sparse (russell_host.hpp), of
//   * the ODE system and its analytical Jacobian      /root/reference/russell_ode/src/samples.rs:497-612
//     (first book: alpha = 2e-3, zero-flux boundaries; second book: alpha = 0.1, periodic, inhomogeneity after t = 1.1),
//     with the five-point molecule and the mirrored / wrapped ghost indices of russell_pde/src/fdm_2d.rs:376-386,944-979;
//   * the Radau5 step (simplified Newton on the transformed variables W, error estimate, collocation polynomial, Gustafsson's
//     predictive controller, Jacobian / factorisation re-use)   russell_ode/src/radau5.rs:186-303 (assemble, factorize),
//     :336-586 (step), :588-651 (accept), :654-665 (reject); constants :697-725;
//   * the variable-step driver                        russell_ode/src/ode_solver.rs:273-378;
//   * the tolerance transformation and defaults        russell_ode/src/params.rs:265,285-298,377-382,481-510;
//   * the command line of the reference's harness      russell_ode/src/bin/brusselator_pde.rs:10-119.
// K_real = gamma I - J and K_comp = (alpha + i beta) I - J keep their structure for the whole run: every factorisation after the
// first goes through LinSolTrait::factorize(&coo, None), i.e. the device-side value refresh (solver_hipmf_factorize_mapped).  The
// real and the complex system are factorised and solved on two threads (radau5.rs:270-296,306-326) unless --serial.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "russell_host.hpp"

using namespace russell;
using Vec = std::vector<double>;

namespace {

// Radau5 constants (radau5.rs:697-725)
const double ALPHA = 2.6810828736277521338957907432111121010270319565630;
const double BETA = 3.0504301992474105694263776247875679044407041991795;
const double GAMMA = 3.6378342527444957322084185135777757979459360868739;
const double E0 = -2.7623054547485993983499285952820549558040707846130;
const double E1 = 0.37993559825272887786874736408712686858426119657697;
const double E2 = -0.091629609865225789249276201199804926431531138001387;
const double MU1 = 0.15505102572168219018027159252941086080340525193433;
const double MU2 = 0.64494897427831780981972840747058913919659474806567;
const double MU3 = -0.84494897427831780981972840747058913919659474806567;
const double MU4 = -0.35505102572168219018027159252941086080340525193433;
const double MU5 = -0.48989794855663561963945681494117827839318949613133;
const double SQRT_6 = 2.44948974278317809819728407470589139196594748065667;
const double C[3] = {(4.0 - SQRT_6) / 10.0, (4.0 + SQRT_6) / 10.0, 1.0};
const double T[3][3] = {{9.1232394870892942792e-02, -0.14125529502095420843, -3.0029194105147424492e-02},
                        {0.24171793270710701896, 0.20412935229379993199, 0.38294211275726193779},
                        {0.96604818261509293619, 1.0, 0.0}};
const double TI[3][3] = {{4.3255798900631553510, 0.33919925181580986954, 0.54177053993587487119},
                         {-4.1787185915519047273, -0.32768282076106238708, 0.47662355450055045196},
                         {-0.50287263494578687595, 2.5719269498556054292, -0.59603920482822492497}};
const double EPS = 2.220446049250313e-16;

// ---- the ODE system (samples.rs:497-612) ----
struct Brusselator {
    size_t npoint, s, ndim;
    bool second_book;
    double molecule[5]; // alpha, beta, beta, gamma, gamma of fdm_2d.rs:376-386 with kx = ky = -alpha_diffusion
    double dx;
    Brusselator(double alpha, size_t np, bool second) : npoint(np), s(np * np), ndim(2 * np * np), second_book(second) {
        dx = 1.0 / (double)(np - 1);
        const double kx = -alpha, ky = -alpha, dx2 = dx * dx;
        molecule[0] = 2.0 * (kx / dx2 + ky / dx2);
        molecule[1] = molecule[2] = -kx / dx2;
        molecule[3] = molecule[4] = -ky / dx2;
    }
    // column indices of row m (fdm_2d.rs:944-979): ghost indices mirrored (zero flux) or wrapped (periodic)
    void bandwidth(size_t m, size_t nn[5]) const {
        const size_t nx = npoint, fin = npoint - 1, i = m % nx, j = m / nx;
        nn[0] = m;
        if (second_book) {
            nn[1] = i != 0 ? m - 1 : m + fin;
            nn[2] = i != fin ? m + 1 : m - fin;
            nn[3] = j != 0 ? m - nx : m + fin * nx;
            nn[4] = j != fin ? m + nx : m - fin * nx;
        } else {
            nn[1] = i != 0 ? m - 1 : m + 1;
            nn[2] = i != fin ? m + 1 : m - 1;
            nn[3] = j != 0 ? m - nx : m + nx;
            nn[4] = j != fin ? m + nx : m - nx;
        }
    }
    void function(Vec &f, double t, const Vec &yy) const {
        for (size_t m = 0; m < s; m++) {
            const double um = yy[m], vm = yy[s + m], um2 = um * um;
            f[m] = 1.0 - 4.4 * um + um2 * vm;
            f[s + m] = 3.4 * um - um2 * vm;
            size_t nn[5];
            bandwidth(m, nn);
            for (int b = 0; b < 5; b++) {
                f[m] += molecule[b] * yy[nn[b]];
                f[s + m] += molecule[b] * yy[s + nn[b]];
            }
            if (second_book && t >= 1.1) {
                const double x = (double)(m % npoint) * dx, y = (double)(m / npoint) * dx;
                const double ddx = x - 0.3, ddy = y - 0.6;
                if (ddx * ddx + ddy * ddy <= 0.01) f[m] += 5.0;
            }
        }
    }
    size_t jac_nnz() const { return 4 * s + 2 * s * 5; }
    // triplets of aa * J in the order of samples.rs:549-571 (duplicates at mirrored boundary nodes are kept: the COO -> CSR step sums them)
    void jacobian(CooMatrix &jj, double aa, const Vec &yy) const {
        jj.reset();
        for (size_t m = 0; m < s; m++) {
            const double um = yy[m], vm = yy[s + m], um2 = um * um;
            jj.put(m, m, aa * (-4.4 + 2.0 * um * vm));
            jj.put(m, s + m, aa * um2);
            jj.put(s + m, m, aa * (3.4 - 2.0 * um * vm));
            jj.put(s + m, s + m, aa * (-um2));
            size_t nn[5];
            bandwidth(m, nn);
            for (int b = 0; b < 5; b++) {
                jj.put(m, nn[b], aa * molecule[b]);
                jj.put(s + m, s + nn[b], aa * molecule[b]);
            }
        }
    }
    void initial(Vec &yy0) const {
        for (size_t m = 0; m < s; m++) {
            const double x = (double)(m % npoint) * dx, y = (double)(m / npoint) * dx;
            if (second_book) {
                yy0[m] = 22.0 * y * std::pow(1.0 - y, 1.5);
                yy0[s + m] = 27.0 * x * std::pow(1.0 - x, 1.5);
            } else {
                yy0[m] = 0.5 + y;
                yy0[s + m] = 1.0 + 5.0 * x;
            }
        }
    }
};

struct Stats {
    size_t n_function = 0, n_jacobian = 0, n_factor = 0, n_lin_sol = 0, n_steps = 0, n_accepted = 0, n_rejected = 0, n_iterations = 0,
           n_iterations_max = 0;
    double h_accepted = 0.0;
    double ns_factor_max = 0, ns_factor_total = 0, ns_lin_sol_max = 0, ns_lin_sol_total = 0, ns_jacobian_total = 0, ns_total = 0;
};

struct Params {
    // params.rs:265 (n_iteration_max), :285-298 (step), :377-382 (radau5)
    size_t n_iteration_max = 7, n_step_max = 100000;
    double m_min = 0.125, m_max = 5.0, m_safety = 0.9, m_first_reject = 0.1, h_ini = 1e-4, rel_error_prev_min = 1e-2;
    double theta_max = 1e-3, c1h = 1.0, c2h = 1.2;
    bool zero_trial = false, use_pred_control = true, concurrent = true;
    double tol_abs = 1e-4, tol_rel = 1e-4, tol_newton = 0.0;
    // params.rs:481-510 (radau5 = true)
    void set_tolerances(double abs_tol, double rel_tol) {
        const double quot = abs_tol / rel_tol;
        tol_rel = 0.1 * std::pow(rel_tol, 2.0 / 3.0);
        tol_abs = tol_rel * quot;
        tol_newton = std::max(10.0 * EPS / tol_rel, std::min(0.03, std::sqrt(tol_rel)));
    }
};

double now_ns() { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double rms_norm(const Vec &err, const Vec &scaling) {
    double sum = 0.0;
    for (size_t m = 0; m < err.size(); m++) {
        const double r = err[m] / scaling[m];
        sum += r * r;
    }
    return std::max(1e-10, std::sqrt(sum / (double)err.size()));
}

struct Work { // workspace.rs
    bool follows_reject_step = false, iterations_diverging = false;
    double h_multiplier_diverging = 1.0, h_prev = 0.0, h_new = 0.0, rel_error_prev = 0.0, rel_error = 0.0;
    Stats stats;
};

class Radau5 {
  public:
    Radau5(const Params &p, const Brusselator &sys) : params(p), system(sys), ndim(sys.ndim) {
        const size_t nnz = sys.jac_nnz() + ndim; // no mass matrix: the diagonal gamma I / (alpha + i beta) I is appended
        CooMatrix::create(jj, ndim, ndim, sys.jac_nnz(), Sym::No);
        CooMatrix::create(kk_real, ndim, ndim, nnz, Sym::No);
        ComplexCooMatrix::create(kk_comp, ndim, ndim, nnz, Sym::No);
        for (Vec *v : {&k_accepted, &scaling, &v0, &v1, &v2, &k0, &k1, &k2, &z0, &z1, &z2, &yc0, &yc1, &yc2, &w0, &w1, &w2, &dw0, &dw1, &dw2})
            v->assign(ndim, 0.0);
        v12.assign(2 * ndim, 0.0);
        dw12.assign(2 * ndim, 0.0);
        theta = params.theta_max;
    }
    StrError init_solvers() {
        StrError e = SolverHIPMF::create(solver_real);
        if (e) return e;
        return ComplexSolverHIPMF::create(solver_comp);
    }

    StrError step(Work &work, double x, const Vec &y, double h) {
        if (work.stats.n_accepted == 0) initialize(work, x, y);
        // Jacobian, K_real, K_comp and their factorisations (for all iterations: simplified Newton)
        if (reuse_jacobian_kk_and_fact) {
            reuse_jacobian_kk_and_fact = false;
        } else {
            assemble(work, x, y, h);
            const double t0 = now_ns();
            work.stats.n_factor++;
            StrError er = nullptr, ec = nullptr;
            const LinSolParams *par = first_factorize ? &lin_sol_params : nullptr; // (lin_solver.rs: params only on the first call)
            if (params.concurrent) {
                std::thread tr([&] { er = solver_real->factorize(kk_real, par); });
                std::thread tc([&] { ec = solver_comp->factorize(kk_comp, par); });
                tr.join();
                tc.join();
            } else {
                er = solver_real->factorize(kk_real, par);
                ec = solver_comp->factorize(kk_comp, par);
            }
            first_factorize = false;
            if (er) return er;
            if (ec) return ec;
            const double dt = now_ns() - t0;
            work.stats.ns_factor_total += dt;
            work.stats.ns_factor_max = std::max(work.stats.ns_factor_max, dt);
        }
        const double u0 = x + C[0] * h, u1 = x + C[1] * h, u2 = x + C[2] * h;
        // starting values of the Newton iterations
        if (work.stats.n_accepted == 0 || params.zero_trial) {
            for (size_t m = 0; m < ndim; m++) z0[m] = z1[m] = z2[m] = w0[m] = w1[m] = w2[m] = 0.0;
        } else {
            const double c3q = h / work.h_prev, c1q = MU1 * c3q, c2q = MU2 * c3q;
            for (size_t m = 0; m < ndim; m++) {
                z0[m] = c1q * (yc0[m] + (c1q - MU4) * (yc1[m] + (c1q - MU3) * yc2[m]));
                z1[m] = c2q * (yc0[m] + (c2q - MU4) * (yc1[m] + (c2q - MU3) * yc2[m]));
                z2[m] = c3q * (yc0[m] + (c3q - MU4) * (yc1[m] + (c3q - MU3) * yc2[m]));
                w0[m] = TI[0][0] * z0[m] + TI[0][1] * z1[m] + TI[0][2] * z2[m];
                w1[m] = TI[1][0] * z0[m] + TI[1][1] * z1[m] + TI[1][2] * z2[m];
                w2[m] = TI[2][0] * z0[m] + TI[2][1] * z1[m] + TI[2][2] * z2[m];
            }
        }
        const double dim = (double)ndim, alpha = ALPHA / h, beta = BETA / h, gamma = GAMMA / h;
        eta = std::pow(std::max(eta, EPS), 0.8);
        theta = params.theta_max;
        double ldw_old = 0.0, thq_old = 0.0;
        bool success = false;
        work.iterations_diverging = false;
        work.stats.n_iterations = 0;
        for (size_t it = 0; it < params.n_iteration_max; it++) {
            work.stats.n_iterations++;
            for (size_t m = 0; m < ndim; m++) {
                v0[m] = y[m] + z0[m];
                v1[m] = y[m] + z1[m];
                v2[m] = y[m] + z2[m];
            }
            work.stats.n_function += 3;
            system.function(k0, u0, v0);
            system.function(k1, u1, v1);
            system.function(k2, u2, v2);
            for (size_t m = 0; m < ndim; m++) {
                v0[m] = TI[0][0] * k0[m] + TI[0][1] * k1[m] + TI[0][2] * k2[m] - gamma * w0[m];
                v1[m] = TI[1][0] * k0[m] + TI[1][1] * k1[m] + TI[1][2] * k2[m] - alpha * w1[m] + beta * w2[m];
                v2[m] = TI[2][0] * k0[m] + TI[2][1] * k1[m] + TI[2][2] * k2[m] - beta * w1[m] - alpha * w2[m];
            }
            for (size_t m = 0; m < ndim; m++) v12[2 * m] = v1[m], v12[2 * m + 1] = v2[m]; // complex_vec_zip
            const double t0 = now_ns();
            work.stats.n_lin_sol++;
            StrError er = nullptr, ec = nullptr;
            if (params.concurrent) {
                std::thread tr([&] { er = solver_real->solve(dw0, v0, false); });
                std::thread tc([&] { ec = solver_comp->solve(dw12, v12, false); });
                tr.join();
                tc.join();
            } else {
                er = solver_real->solve(dw0, v0, false);
                ec = solver_comp->solve(dw12, v12, false);
            }
            if (er) return er;
            if (ec) return ec;
            if (getenv("BRUS_DEBUG")) {
                VerifyLinSys vr, vc;
                VerifyLinSys::from(vr, kk_real, dw0, v0);
                VerifyLinSys::from_complex(vc, kk_comp, dw12, v12);
                fprintf(stderr, "   lin sys: real rel_err %.3e, complex rel_err %.3e\n", vr.relative_error, vc.relative_error);
            }
            const double dt = now_ns() - t0;
            work.stats.ns_lin_sol_total += dt;
            work.stats.ns_lin_sol_max = std::max(work.stats.ns_lin_sol_max, dt);
            double ldw = 0.0;
            for (size_t m = 0; m < ndim; m++) {
                w0[m] += dw0[m];
                w1[m] += dw12[2 * m];
                w2[m] += dw12[2 * m + 1];
                z0[m] = T[0][0] * w0[m] + T[0][1] * w1[m] + T[0][2] * w2[m];
                z1[m] = T[1][0] * w0[m] + T[1][1] * w1[m] + T[1][2] * w2[m];
                z2[m] = T[2][0] * w0[m] + T[2][1] * w1[m] + T[2][2] * w2[m];
                const double r0 = dw0[m] / scaling[m], r1 = dw12[2 * m] / scaling[m], r2 = dw12[2 * m + 1] / scaling[m];
                ldw += r0 * r0 + r1 * r1 + r2 * r2;
            }
            ldw = std::sqrt(ldw / (3.0 * dim));
            const size_t newt = work.stats.n_iterations, nit = params.n_iteration_max;
            if (newt > 1 && newt < nit) {
                const double thq = ldw / ldw_old;
                theta = newt == 2 ? thq : std::sqrt(thq * thq_old);
                thq_old = thq;
                if (theta < 0.99) {
                    eta = theta / (1.0 - theta);
                    const double rel_err = eta * ldw * std::pow(theta, (double)(nit - 1 - newt)) / params.tol_newton;
                    if (rel_err >= 1.0) { // diverging
                        const double q_newt = std::max(1.0e-4, std::min(20.0, rel_err));
                        work.h_multiplier_diverging = 0.8 * std::pow(q_newt, -1.0 / (double)(4 + nit - 1 - newt));
                        work.iterations_diverging = true;
                        return nullptr;
                    }
                } else { // diverging badly
                    work.h_multiplier_diverging = 0.5;
                    work.iterations_diverging = true;
                    return nullptr;
                }
            }
            ldw_old = ldw;
            if (eta * ldw < params.tol_newton) {
                success = true;
                break;
            }
        }
        work.stats.n_iterations_max = std::max(work.stats.n_iterations_max, work.stats.n_iterations);
        if (!success) return "Newton-Raphson method did not complete successfully";
        // error estimate (no mass matrix): err = K_real^{-1} (gamma ez + f0)
        Vec &ez = w0, &mez = w1, &rhs = w2, &err = dw0;
        for (size_t m = 0; m < ndim; m++) {
            ez[m] = E0 * z0[m] + E1 * z1[m] + E2 * z2[m];
            mez[m] = gamma * ez[m];
            rhs[m] = mez[m] + k_accepted[m];
        }
        StrError e = solver_real->solve(err, rhs, false);
        if (e) return e;
        work.rel_error = rms_norm(err, scaling);
        if (work.rel_error < 1.0) return nullptr;
        if (work.stats.n_accepted == 0 || work.follows_reject_step) {
            Vec &ype = dw1, &fpe = dw2;
            for (size_t m = 0; m < ndim; m++) ype[m] = y[m] + err[m];
            work.stats.n_function++;
            system.function(fpe, x, ype);
            for (size_t m = 0; m < ndim; m++) rhs[m] = mez[m] + fpe[m];
            e = solver_real->solve(err, rhs, false);
            if (e) return e;
            work.rel_error = rms_norm(err, scaling);
        }
        return nullptr;
    }

    void accept(Work &work, double &x, Vec &y, double h) {
        reuse_jacobian_kk_and_fact = false;
        reuse_jacobian = false;
        jacobian_computed = false;
        for (size_t m = 0; m < ndim; m++) {
            y[m] += z2[m];
            yc0[m] = (z1[m] - z2[m]) / MU4;
            yc1[m] = ((z0[m] - z1[m]) / MU5 - yc0[m]) / MU3;
            yc2[m] = yc1[m] - ((z0[m] - z1[m]) / MU5 - z0[m] / MU1) / MU2;
        }
        const size_t newt = work.stats.n_iterations;
        const double num = params.m_safety * (double)(1 + 2 * params.n_iteration_max), den = (double)(newt + 2 * params.n_iteration_max);
        const double fac = std::min(params.m_safety, num / den);
        double div = std::max(params.m_min, std::min(params.m_max, std::pow(work.rel_error, 0.25) / fac));
        double h_new = h / div;
        if (params.use_pred_control && work.stats.n_accepted > 1) { // Gustafsson
            const double r2 = work.rel_error * work.rel_error, rp = work.rel_error_prev;
            double f2 = (work.h_prev / h) * std::pow(r2 / rp, 0.25) / params.m_safety;
            f2 = std::max(params.m_min, std::min(params.m_max, f2));
            div = std::max(div, f2);
            h_new = h / div;
        }
        const double h_ratio = h_new / h;
        reuse_jacobian_kk_and_fact = theta <= params.theta_max && h_ratio >= params.c1h && h_ratio <= params.c2h;
        if (!reuse_jacobian_kk_and_fact) work.h_new = h_new;
        if (!reuse_jacobian_kk_and_fact) reuse_jacobian = theta <= params.theta_max;
        x += h;
        initialize(work, x, y);
    }

    void reject(Work &work, double h) {
        const size_t newt = work.stats.n_iterations;
        const double num = params.m_safety * (double)(1 + 2 * params.n_iteration_max), den = (double)(newt + 2 * params.n_iteration_max);
        const double fac = std::min(params.m_safety, num / den);
        const double div = std::max(params.m_min, std::min(params.m_max, std::pow(work.rel_error, 0.25) / fac));
        work.h_new = h / div;
    }

    LinSolParams lin_sol_params;

  private:
    void initialize(Work &work, double x, const Vec &y) {
        for (size_t i = 0; i < ndim; i++) scaling[i] = params.tol_abs + params.tol_rel * std::fabs(y[i]);
        work.stats.n_function++;
        system.function(k_accepted, x, y);
    }
    void assemble(Work &work, double, const Vec &y, double h) {
        if (reuse_jacobian) {
            reuse_jacobian = false;
        } else if (!jacobian_computed) {
            const double t0 = now_ns();
            work.stats.n_jacobian++;
            system.jacobian(jj, 1.0, y);
            jacobian_computed = true;
            work.stats.ns_jacobian_total += now_ns() - t0;
        }
        const double alpha = ALPHA / h, beta = BETA / h, gamma = GAMMA / h;
        kk_real.assign(-1.0, jj); // K_real = -J
        kk_comp.reset();          // K_comp = -J (assign_real of complex_coo_matrix.rs)
        for (size_t k = 0; k < jj.nnz; k++) kk_comp.put((size_t)jj.indices_i[k], (size_t)jj.indices_j[k], -jj.values[k], 0.0);
        for (size_t m = 0; m < ndim; m++) {
            kk_real.put(m, m, gamma);
            kk_comp.put(m, m, alpha, beta);
        }
    }

  public:
    // the backend's diagnostic counters of the two handles (HIPMF_COUNTER_* of include/russell_hipmf.h)
    long long counter_real(int which) const { return solver_real ? (long long)solver_real->get_counter(which) : -1; }
    long long counter_comp(int which) const { return solver_comp ? (long long)solver_comp->get_counter(which) : -1; }

  private:
    Params params;
    const Brusselator &system;
    size_t ndim;
    CooMatrix jj, kk_real;
    ComplexCooMatrix kk_comp;
    std::unique_ptr<SolverHIPMF> solver_real;
    std::unique_ptr<ComplexSolverHIPMF> solver_comp;
    bool reuse_jacobian = false, reuse_jacobian_kk_and_fact = false, jacobian_computed = false, first_factorize = true;
    double eta = 1.0, theta = 0.0;
    Vec k_accepted, scaling, v0, v1, v2, v12, k0, k1, k2, z0, z1, z2, yc0, yc1, yc2, w0, w1, w2, dw0, dw1, dw2, dw12;
};

// ode_solver.rs:273-378 (variable stepping)
StrError solve(Radau5 &actual, const Params &params, Work &work, Vec &y, double x0, double x1) {
    double h = std::min(params.h_ini, x1 - x0);
    work.follows_reject_step = false;
    work.iterations_diverging = false;
    work.h_multiplier_diverging = 1.0;
    work.h_prev = h, work.h_new = h, work.rel_error_prev = params.rel_error_prev_min, work.rel_error = 0.0;
    double x = x0;
    bool success = false, last_step = false;
    for (size_t iter = 0; iter < params.n_step_max; iter++) {
        const double dx = x1 - x;
        if (dx <= 10.0 * EPS) {
            success = true;
            break;
        }
        h = std::min(work.h_new, dx);
        if (h <= 10.0 * EPS) return "the stepsize becomes too small";
        work.stats.n_steps++;
        StrError e = actual.step(work, x, y, h);
        if (e) return e;
        if (getenv("BRUS_DEBUG")) fprintf(stderr, "step %zu x=%.6g h=%.6g div=%d rel_error=%.4g newt=%zu\n", work.stats.n_steps, x, h, (int)work.iterations_diverging, work.rel_error, work.stats.n_iterations);
        if (work.iterations_diverging) {
            work.iterations_diverging = false;
            work.follows_reject_step = true;
            last_step = false;
            work.h_new = h * work.h_multiplier_diverging;
            continue;
        }
        if (work.rel_error < 1.0) {
            work.stats.n_accepted++;
            actual.accept(work, x, y, h);
            for (double v : y)
                if (!std::isfinite(v)) return "an element of the vector is either infinite or NaN";
            if (work.follows_reject_step) work.h_new = std::min(work.h_new, h);
            work.follows_reject_step = false;
            work.h_prev = h;
            work.rel_error_prev = std::max(params.rel_error_prev_min, work.rel_error);
            work.stats.h_accepted = work.h_new;
            if (last_step) {
                success = true;
                break;
            }
            if (x + work.h_new >= x1) last_step = true;
        } else {
            if (work.stats.n_accepted > 0) work.stats.n_rejected++;
            work.follows_reject_step = true;
            last_step = false;
            if (work.stats.n_accepted == 0 && params.m_first_reject > 0.0) work.h_new = h * params.m_first_reject;
            else actual.reject(work, h);
        }
    }
    return success ? nullptr : "variable stepping did not converge";
}

} // namespace

int main(int argc, char **argv) {
    size_t npoint = 129;
    bool first_book = false, serial = false, json = false;
    int neg_exp_tol = 4;
    double t1 = 1.5;
    bool t1_given = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", what);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--npoint") npoint = (size_t)atol(next("--npoint"));
        else if (a == "--first-book") first_book = true;
        else if (a == "--serial") serial = true;
        else if (a == "--neg-exp-tol") neg_exp_tol = atoi(next("--neg-exp-tol"));
        else if (a == "--t1") t1 = atof(next("--t1")), t1_given = true;
        else if (a == "--json") json = true;
        else if (a == "-g" || a == "--genie") {
            const std::string g = next("--genie");
            if (g != "hipmf" && g != "Hipmf") {
                fprintf(stderr, "only the HIPMF genie is available in this build\n");
                return 2;
            }
        } else {
            fprintf(stderr, "usage: brusselator_pde [--npoint N] [--first-book] [--neg-exp-tol K] [--serial] [--t1 T] [--json] [-g hipmf]\n");
            return 2;
        }
    }
    if (npoint < 2) {
        fprintf(stderr, "npoint must be >= 2\n");
        return 2;
    }
    (void)t1_given;
    const double alpha = first_book ? 2e-3 : 0.1;
    Brusselator sys(alpha, npoint, !first_book);
    Params params;
    const double tol = std::pow(10.0, -neg_exp_tol);
    params.h_ini = 1e-4;
    params.concurrent = !serial;
    params.set_tolerances(tol, tol);
    Radau5 radau(params, sys);
    StrError e = radau.init_solvers();
    if (e) {
        fprintf(stderr, "ERROR: %s\n", e);
        return 1;
    }
    Vec yy(sys.ndim);
    sys.initial(yy);
    Work work;
    const double t_start = now_ns();
    e = solve(radau, params, work, yy, 0.0, t1);
    work.stats.ns_total = now_ns() - t_start;
    if (e) {
        fprintf(stderr, "ERROR: %s\n", e);
        return 1;
    }
    const Stats &st = work.stats;
    const size_t ij_mid = (npoint - 1) / 2, m_mid = ij_mid + ij_mid * npoint;
    // the backend's own counters of both handles (HIPMF_COUNTER_* of include/russell_hipmf.h): solves that fell back to the level-set
    // launches (2), factorisations that fell back from the chained tiled steps (7), solves that found the device's gate held by the other
    // handle (11: the real and the complex system are solved on two threads, radau5.rs:270-296) -- fallbacks are never expected
    const long long fb_real = radau.counter_real(2);
    const long long fb_comp = radau.counter_comp(2);
    const long long cf_real = radau.counter_real(7);
    const long long cf_comp = radau.counter_comp(7);
    const long long gw_real = radau.counter_real(11);
    const long long gw_comp = radau.counter_comp(11);
    if (json) {
        printf("{\"second_book\": %s, \"npoint\": %zu, \"ndim\": %zu, \"jac_nnz\": %zu, \"tolerance\": %.3e, \"concurrent\": %s, \"t1\": %.17g, "
               "\"n_function\": %zu, \"n_jacobian\": %zu, \"n_factor\": %zu, \"n_lin_sol\": %zu, \"n_steps\": %zu, \"n_accepted\": %zu, "
               "\"n_rejected\": %zu, \"n_iterations_max\": %zu, \"h_accepted\": %.17g, \"u_mid\": %.17g, \"v_mid\": %.17g, "
               "\"ms_total\": %.3f, \"ms_factor_max\": %.3f, \"ms_factor_avg\": %.3f, \"ms_lin_sol_max\": %.3f, \"ms_lin_sol_avg\": %.3f, "
               "\"ms_jacobian_total\": %.3f, \"fused_fallbacks\": [%lld, %lld], \"chain_fallbacks\": [%lld, %lld], \"gate_waits\": [%lld, %lld]}\n",
               first_book ? "false" : "true", npoint, sys.ndim, sys.jac_nnz(), tol, serial ? "false" : "true", t1, st.n_function, st.n_jacobian,
               st.n_factor, st.n_lin_sol, st.n_steps, st.n_accepted, st.n_rejected, st.n_iterations_max, st.h_accepted, yy[m_mid], yy[sys.s + m_mid],
               st.ns_total * 1e-6, st.ns_factor_max * 1e-6, st.n_factor ? st.ns_factor_total * 1e-6 / (double)st.n_factor : 0.0,
               st.ns_lin_sol_max * 1e-6, st.n_lin_sol ? st.ns_lin_sol_total * 1e-6 / (double)st.n_lin_sol : 0.0, st.ns_jacobian_total * 1e-6, fb_real,
               fb_comp, cf_real, cf_comp, gw_real, gw_comp);
    } else {
        printf("Second-book problem              = %s\n", first_book ? "false" : "true");
        printf("Number of points along x and y   = %zu\n", npoint);
        printf("Tolerance (abs_tol = rel_tol)    = %.2e\n", tol);
        printf("Concurrent real and complex sys  = %s\n", serial ? "false" : "true");
        printf("Problem dimension (ndim)         = %zu\n", sys.ndim);
        printf("Number of non-zeros (jac_nnz)    = %zu\n", sys.jac_nnz());
        printf("Linear solver                    = Hipmf\n");
        printf("Number of function evaluations   = %zu\n", st.n_function);
        printf("Number of Jacobian evaluations   = %zu\n", st.n_jacobian);
        printf("Number of factorizations         = %zu\n", st.n_factor);
        printf("Number of lin sys solutions      = %zu\n", st.n_lin_sol);
        printf("Number of performed steps        = %zu\n", st.n_steps);
        printf("Number of accepted steps         = %zu\n", st.n_accepted);
        printf("Number of rejected steps         = %zu\n", st.n_rejected);
        printf("Number of iterations (maximum)   = %zu\n", st.n_iterations_max);
        printf("Last accepted/suggested stepsize = %.17g\n", st.h_accepted);
        printf("Max time spent on factorization  = %.3f ms\n", st.ns_factor_max * 1e-6);
        printf("Max time spent on lin solution   = %.3f ms\n", st.ns_lin_sol_max * 1e-6);
        printf("Total time                       = %.3f ms\n", st.ns_total * 1e-6);
        printf("Solve fallbacks (real, complex)  = %lld, %lld\n", fb_real, fb_comp);
        printf("Chain fallbacks (real, complex)  = %lld, %lld\n", cf_real, cf_comp);
        printf("Waits at the device gate (r, c)  = %lld, %lld\n", gw_real, gw_comp);
        printf("u, v at the middle node          = %.15g %.15g\n", yy[m_mid], yy[sys.s + m_mid]);
    }
    return 0;
}

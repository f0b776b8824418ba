/* umfpack_probe.c -- CPU-baseline timing harness (test / benchmark infrastructure, never part of the product).
 *
 * Runs the call sequence of the reference's UMFPACK shim on a CSC matrix with the shim's controls
 * (/root/reference/russell_sparse/c_code/interface_umfpack.c: umfpack_di_defaults :47, Control[STRATEGY / ORDERING / SCALE]
 * :99-105, umfpack_di_symbolic :109, umfpack_di_numeric :167, umfpack_di_solve(UMFPACK_A) :229; ordering AMD and scaling SUM are
 * the values the Rust side sends for Ordering::Auto / Scaling::Auto, solver_umfpack.rs:457-487) and times each phase.
 * SuiteSparse is NOT part of this image: libumfpack is looked for at RUN time (dlopen), so the same binary reports the real
 * reference library's time on a box that has it and "not found" (return -1) elsewhere.
 * The numeric constants are those of SuiteSparse's umfpack.h (stable across 5.x / 6.x).  */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#define UMFPACK_CONTROL 20
#define UMFPACK_INFO 90
#define UMFPACK_STRATEGY 5
#define UMFPACK_ORDERING 10
#define UMFPACK_SCALE 16
#define UMFPACK_STRATEGY_AUTO 0
#define UMFPACK_ORDERING_AMD 1
#define UMFPACK_SCALE_SUM 1
#define UMFPACK_A 0

typedef void (*defaults_fn)(double *);
typedef int (*symbolic_fn)(int, int, const int *, const int *, const double *, void **, const double *, double *);
typedef int (*numeric_fn)(const int *, const int *, const double *, void *, void **, const double *, double *);
typedef int (*solve_fn)(int, const int *, const int *, const double *, double *, const double *, void *, const double *, double *);
typedef void (*free_fn)(void **);

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* returns 0 on success, -1 when no libumfpack can be loaded, a positive UMFPACK status otherwise;
 * seconds[0..2] = symbolic, numeric, solve; lib_used receives the name that loaded */
int umfpack_probe(int32_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax, const double *b, double *x, double *seconds,
                  char *lib_used, int32_t lib_used_len) {
    static const char *names[] = {"libumfpack.so", "libumfpack.so.6", "libumfpack.so.5", "libumfpack.so.7", NULL};
    void *dl = NULL;
    for (int i = 0; names[i] && !dl; i++) {
        dl = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (dl && lib_used && lib_used_len > 0) snprintf(lib_used, (size_t)lib_used_len, "%s", names[i]);
    }
    if (!dl) return -1;
    defaults_fn f_defaults = (defaults_fn)dlsym(dl, "umfpack_di_defaults");
    symbolic_fn f_symbolic = (symbolic_fn)dlsym(dl, "umfpack_di_symbolic");
    numeric_fn f_numeric = (numeric_fn)dlsym(dl, "umfpack_di_numeric");
    solve_fn f_solve = (solve_fn)dlsym(dl, "umfpack_di_solve");
    free_fn f_free_s = (free_fn)dlsym(dl, "umfpack_di_free_symbolic");
    free_fn f_free_n = (free_fn)dlsym(dl, "umfpack_di_free_numeric");
    if (!f_defaults || !f_symbolic || !f_numeric || !f_solve || !f_free_s || !f_free_n) return -1;
    double control[UMFPACK_CONTROL], info[UMFPACK_INFO];
    f_defaults(control);
    control[UMFPACK_STRATEGY] = UMFPACK_STRATEGY_AUTO;
    control[UMFPACK_ORDERING] = UMFPACK_ORDERING_AMD;
    control[UMFPACK_SCALE] = UMFPACK_SCALE_SUM;
    void *symbolic = NULL, *numeric = NULL;
    double t0 = now_s();
    int code = f_symbolic(n, n, Ap, Ai, Ax, &symbolic, control, info);
    double t1 = now_s();
    if (code != 0) return code > 0 ? code : 1000 - code;
    code = f_numeric(Ap, Ai, Ax, symbolic, &numeric, control, info);
    double t2 = now_s();
    if (code != 0) {
        f_free_s(&symbolic);
        return code > 0 ? code : 1000 - code;
    }
    code = f_solve(UMFPACK_A, Ap, Ai, Ax, x, b, numeric, control, info);
    double t3 = now_s();
    f_free_n(&numeric);
    f_free_s(&symbolic);
    seconds[0] = t1 - t0, seconds[1] = t2 - t1, seconds[2] = t3 - t2;
    return code == 0 ? 0 : (code > 0 ? code : 1000 - code);
}

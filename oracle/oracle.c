/*
 * oracle.c -- CPU restatement of the russell_sparse solver path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared library.  The product (russell_amd/) never links, imports or calls it.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   oracle_coo_to_csc      russell_sparse/src/csc_matrix.rs:365-505  (COO->CSC, duplicates summed
 *                          in COO order within a row, rows ascending inside a column)
 *   oracle_coo_to_csr      russell_sparse/src/csr_matrix.rs:359-480  (COO->CSR, same dedup, then a
 *                          per-row sort by column)
 *   oracle_coo_matvec      russell_sparse/src/coo_matrix.rs:547-566  (v := alpha*A*u, mirroring the
 *                          off-diagonal entries when the storage is triangular)
 *   oracle_csr_matvec      russell_sparse/src/csr_matrix.rs:709-729
 *   oracle_csc_matvec      russell_sparse/src/csc_matrix.rs:735-755
 *   oracle_verify          russell_sparse/src/verify_lin_sys.rs:60-96
 *   oracle_lu_*            the call sequence of russell_sparse/c_code/interface_umfpack.c:82-243
 *                          (symbolic -> numeric -> solve(UMFPACK_A) with the library defaults the
 *                          shim leaves in place: row scaling SUM, threshold partial pivoting,
 *                          <=2 steps of iterative refinement, determinant as mantissa*10^exp).
 *
 * The LU arithmetic of the reference lives in SuiteSparse UMFPACK, which is NOT in the reference
 * tree (russell_sparse/src/util.rs:168 pins it to "latest (from GitHub)"; it cannot be built here).
 * oracle_lu_* therefore restates the PUBLISHED algorithm family instead of UMFPACK's code: a
 * left-looking sparse LU with a depth-first symbolic reach per column (Gilbert & Peierls, SIAM
 * J. Sci. Stat. Comput. 9, 1988), threshold partial pivoting with UMFPACK's documented default
 * tolerance 0.1 and a preference for the diagonal, row scaling by the sum of absolute values
 * (UMFPACK_SCALE_SUM), and iterative refinement on the unscaled system.  Parity is pinned on the
 * reference's own golden vectors at the solver boundary (tests/golden/):
 * solutions, determinant, singular status.  Permutation vectors / pivot sequences are UNPINNED:
 * the reference never extracts them from UMFPACK and no reference test holds them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_SINGULAR 1 /* same value UMFPACK returns as UMFPACK_WARNING_singular_matrix */
#define ORACLE_OUT_OF_MEMORY (-1)
#define ORACLE_INVALID (-3)

/* ------------------------------------------------------------------------------------------------
 * COO -> CSC  (csc_matrix.rs:365-505).  bp has ncol+1 entries, bi/bx have nnz entries (capacity
 * including duplicates).  Returns the final nnz (= bp[ncol]) or a negative error.
 * ---------------------------------------------------------------------------------------------- */
int32_t oracle_coo_to_csc(int32_t nrow, int32_t ncol, int32_t nnz, const int32_t *ai, const int32_t *aj,
                          const double *ax, int32_t *bp, int32_t *bi, double *bx) {
    int32_t ndim = nrow > ncol ? nrow : ncol;
    int32_t *rp = (int32_t *)calloc((size_t)nrow + 1, sizeof(int32_t));
    int32_t *rj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    double *rx = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    int32_t *rc = (int32_t *)calloc((size_t)nrow + 1, sizeof(int32_t));
    int32_t *w = (int32_t *)calloc((size_t)ndim + 1, sizeof(int32_t));
    if (!rp || !rj || !rx || !rc || !w) {
        free(rp), free(rj), free(rx), free(rc), free(w);
        return ORACLE_OUT_OF_MEMORY;
    }
    /* rows counted with duplicates */
    for (int32_t k = 0; k < nnz; k++) w[ai[k]]++;
    rp[0] = 0;
    for (int32_t i = 0; i < nrow; i++) {
        rp[i + 1] = rp[i] + w[i];
        w[i] = rp[i];
    }
    /* row form in COO order */
    for (int32_t k = 0; k < nnz; k++) {
        int32_t p = w[ai[k]]++;
        rj[p] = aj[k];
        rx[p] = ax[k];
    }
    /* duplicates summed per row: w[j] remembers where column j landed in the current row */
    for (int32_t j = 0; j < ncol; j++) w[j] = -1;
    for (int32_t i = 0; i < nrow; i++) {
        int32_t p1 = rp[i], p2 = rp[i + 1], dest = p1;
        for (int32_t p = p1; p < p2; p++) {
            int32_t j = rj[p];
            if (w[j] >= p1) {
                rx[w[j]] += rx[p];
            } else {
                w[j] = dest;
                if (dest != p) {
                    rj[dest] = j;
                    rx[dest] = rx[p];
                }
                dest++;
            }
        }
        rc[i] = dest - p1;
    }
    /* column counts, pointers, scatter (row-by-row => rows ascending within each column) */
    for (int32_t j = 0; j < ncol; j++) w[j] = 0;
    for (int32_t i = 0; i < nrow; i++)
        for (int32_t p = rp[i]; p < rp[i] + rc[i]; p++) w[rj[p]]++;
    bp[0] = 0;
    for (int32_t j = 0; j < ncol; j++) bp[j + 1] = bp[j] + w[j];
    for (int32_t j = 0; j < ncol; j++) w[j] = bp[j];
    for (int32_t i = 0; i < nrow; i++)
        for (int32_t p = rp[i]; p < rp[i] + rc[i]; p++) {
            int32_t cp = w[rj[p]]++;
            bi[cp] = i;
            bx[cp] = rx[p];
        }
    int32_t final_nnz = bp[ncol];
    free(rp), free(rj), free(rx), free(rc), free(w);
    return final_nnz;
}

/* small insertion/shell sort of (col, val) pairs by col; rows are short */
static void sort_pairs(int32_t *j, double *x, int32_t n) {
    for (int32_t gap = n / 2; gap > 0; gap /= 2)
        for (int32_t a = gap; a < n; a++) {
            int32_t tj = j[a];
            double tx = x[a];
            int32_t b = a;
            while (b >= gap && j[b - gap] > tj) {
                j[b] = j[b - gap];
                x[b] = x[b - gap];
                b -= gap;
            }
            j[b] = tj;
            x[b] = tx;
        }
}

/* COO -> CSR (csr_matrix.rs:359-480) */
int32_t oracle_coo_to_csr(int32_t nrow, int32_t ncol, int32_t nnz, const int32_t *ai, const int32_t *aj,
                          const double *ax, int32_t *bp, int32_t *bj, double *bx) {
    int32_t ndim = nrow > ncol ? nrow : ncol;
    int32_t *rp = (int32_t *)calloc((size_t)nrow + 1, sizeof(int32_t));
    int32_t *rj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    double *rx = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    int32_t *w = (int32_t *)calloc((size_t)ndim + 1, sizeof(int32_t));
    if (!rp || !rj || !rx || !w) {
        free(rp), free(rj), free(rx), free(w);
        return ORACLE_OUT_OF_MEMORY;
    }
    for (int32_t k = 0; k < nnz; k++) w[ai[k]]++;
    rp[0] = 0;
    for (int32_t i = 0; i < nrow; i++) {
        rp[i + 1] = rp[i] + w[i];
        w[i] = rp[i];
    }
    for (int32_t k = 0; k < nnz; k++) {
        int32_t p = w[ai[k]]++;
        rj[p] = aj[k];
        rx[p] = ax[k];
    }
    for (int32_t j = 0; j < ncol; j++) w[j] = -1;
    int32_t k = 0;
    bp[0] = 0;
    for (int32_t i = 0; i < nrow; i++) {
        int32_t p1 = rp[i], p2 = rp[i + 1], dest = p1;
        for (int32_t p = p1; p < p2; p++) {
            int32_t j = rj[p];
            if (w[j] >= p1) {
                rx[w[j]] += rx[p];
            } else {
                w[j] = dest;
                if (dest != p) {
                    rj[dest] = j;
                    rx[dest] = rx[p];
                }
                dest++;
            }
        }
        int32_t cnt = dest - p1;
        sort_pairs(rj + p1, rx + p1, cnt);
        for (int32_t p = p1; p < p1 + cnt; p++) {
            bj[k] = rj[p];
            bx[k] = rx[p];
            k++;
        }
        bp[i + 1] = k;
    }
    free(rp), free(rj), free(rx), free(w);
    return k;
}

/* v := alpha * A * u for COO.  sym: 0 = no/full storage, 1 = triangular storage (mirror
 * off-diagonals), coo_matrix.rs:547-566 */
void oracle_coo_matvec(int32_t nrow, int32_t nnz, const int32_t *ai, const int32_t *aj, const double *ax,
                       int32_t sym_triangular, double alpha, const double *u, double *v) {
    for (int32_t i = 0; i < nrow; i++) v[i] = 0.0;
    for (int32_t p = 0; p < nnz; p++) {
        int32_t i = ai[p], j = aj[p];
        v[i] += alpha * ax[p] * u[j];
        if (sym_triangular && i != j) v[j] += alpha * ax[p] * u[i];
    }
}

/* v := alpha * A * u for CSR, csr_matrix.rs:709-729 */
void oracle_csr_matvec(int32_t nrow, const int32_t *rp, const int32_t *cj, const double *ax,
                       int32_t sym_triangular, double alpha, const double *u, double *v) {
    for (int32_t i = 0; i < nrow; i++) v[i] = 0.0;
    for (int32_t i = 0; i < nrow; i++)
        for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
            int32_t j = cj[p];
            v[i] += alpha * ax[p] * u[j];
            if (sym_triangular && i != j) v[j] += alpha * ax[p] * u[i];
        }
}

/* v := alpha * A * u for CSC, csc_matrix.rs:735-755 */
void oracle_csc_matvec(int32_t nrow, int32_t ncol, const int32_t *cp, const int32_t *ri, const double *ax,
                       int32_t sym_triangular, double alpha, const double *u, double *v) {
    for (int32_t i = 0; i < nrow; i++) v[i] = 0.0;
    for (int32_t j = 0; j < ncol; j++)
        for (int32_t p = cp[j]; p < cp[j + 1]; p++) {
            int32_t i = ri[p];
            v[i] += alpha * ax[p] * u[j];
            if (sym_triangular && i != j) v[j] += alpha * ax[p] * u[i];
        }
}

/* verify_lin_sys.rs:60-96: out = {max_abs_a, max_abs_ax, max_abs_diff, relative_error} */
int32_t oracle_verify(int32_t nrow, int32_t nnz, const int32_t *ai, const int32_t *aj, const double *ax,
                      int32_t sym_triangular, const double *x, const double *rhs, double *out) {
    if (nnz < 1) return ORACLE_INVALID;
    double max_abs_a = 0.0;
    for (int32_t p = 0; p < nnz; p++)
        if (fabs(ax[p]) > max_abs_a) max_abs_a = fabs(ax[p]);
    double *v = (double *)malloc(sizeof(double) * (size_t)nrow);
    if (!v) return ORACLE_OUT_OF_MEMORY;
    oracle_coo_matvec(nrow, nnz, ai, aj, ax, sym_triangular, 1.0, x, v);
    double max_abs_ax = 0.0, max_abs_diff = 0.0;
    for (int32_t i = 0; i < nrow; i++) {
        if (fabs(v[i]) > max_abs_ax) max_abs_ax = fabs(v[i]);
        double d = fabs(v[i] - rhs[i]);
        if (d > max_abs_diff) max_abs_diff = d;
    }
    free(v);
    out[0] = max_abs_a;
    out[1] = max_abs_ax;
    out[2] = max_abs_diff;
    out[3] = max_abs_diff / (max_abs_a + 1.0);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Sparse LU.   P * R * A * Q = L * U
 *   R     row scaling (UMFPACK_SCALE_SUM = 1: 1/sum|a_ij|, MAX = 2: 1/max|a_ij|, NONE = 0)
 *   Q     column pre-ordering supplied by the caller (NULL = natural), like umfpack_di_qsymbolic
 *   P     row permutation from threshold partial pivoting (tolerance 0.1, diagonal preferred)
 * ---------------------------------------------------------------------------------------------- */
typedef struct OracleLU {
    int32_t n;
    /* L: unit lower, stored by columns WITHOUT the unit diagonal; row indices are ORIGINAL rows */
    int64_t *lp;
    int32_t *li;
    double *lx;
    /* U: by columns, row indices are PIVOT STEP numbers (0..k), diagonal stored last */
    int64_t *up;
    int32_t *ui;
    double *ux;
    int32_t *pinv;  /* original row -> pivot step */
    int32_t *prow;  /* pivot step -> original row */
    int32_t *q;     /* pivot step (column position) -> original column */
    double *rs;     /* row scale factors (multiplicative) */
    int32_t status; /* ORACLE_OK or ORACLE_SINGULAR */
    int32_t nswaps_parity;
    double min_abs_pivot, max_abs_pivot;
    /* A kept for refinement (CSC copy) */
    int32_t *ap;
    int32_t *ai;
    double *ax;
} OracleLU;

void oracle_lu_free(OracleLU *f) {
    if (!f) return;
    free(f->lp), free(f->li), free(f->lx), free(f->up), free(f->ui), free(f->ux);
    free(f->pinv), free(f->prow), free(f->q), free(f->rs), free(f->ap), free(f->ai), free(f->ax);
    free(f);
}

int64_t oracle_lu_nnz_l(const OracleLU *f) { return f->lp[f->n]; }
int64_t oracle_lu_nnz_u(const OracleLU *f) { return f->up[f->n]; }
int32_t oracle_lu_status(const OracleLU *f) { return f->status; }

/* Depth-first reach of column pattern through the columns of L already computed
 * (Gilbert-Peierls).  Nodes are ORIGINAL row indices; a row that is already pivotal (pinv>=0)
 * has successors = the pattern of L(:, pinv[row]).  Output: topological order in xi[top..n). */
static int32_t reach(const OracleLU *f, int32_t n, const int32_t *bi, int32_t bnz, int32_t *xi, int32_t *stack,
                     int64_t *pos, int32_t *mark, int32_t stamp) {
    int32_t top = n;
    for (int32_t b = 0; b < bnz; b++) {
        int32_t root = bi[b];
        if (mark[root] == stamp) continue;
        int32_t head = 0;
        stack[0] = root;
        while (head >= 0) {
            int32_t i = stack[head];
            int32_t col = f->pinv[i];
            if (mark[i] != stamp) {
                mark[i] = stamp;
                pos[head] = (col < 0) ? 0 : f->lp[col];
            }
            int done = 1;
            if (col >= 0) {
                int64_t pend = f->lp[col + 1];
                for (int64_t p = pos[head]; p < pend; p++) {
                    int32_t r = f->li[p];
                    if (mark[r] == stamp) continue;
                    pos[head] = p + 1;
                    stack[++head] = r;
                    done = 0;
                    break;
                }
            }
            if (done) {
                head--;
                xi[--top] = i;
            }
        }
    }
    return top;
}

/* numeric factorisation; returns NULL only on allocation failure */
OracleLU *oracle_lu_factor(int32_t n, const int32_t *ap, const int32_t *ai, const double *ax, const int32_t *q,
                           int32_t scaling, double pivot_tol) {
    OracleLU *f = (OracleLU *)calloc(1, sizeof(OracleLU));
    if (!f) return NULL;
    f->n = n;
    int64_t annz = ap[n];
    int64_t lcap = 4 * annz + n, ucap = 4 * annz + n;
    f->lp = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    f->up = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    f->li = (int32_t *)malloc(sizeof(int32_t) * (size_t)lcap);
    f->lx = (double *)malloc(sizeof(double) * (size_t)lcap);
    f->ui = (int32_t *)malloc(sizeof(int32_t) * (size_t)ucap);
    f->ux = (double *)malloc(sizeof(double) * (size_t)ucap);
    f->pinv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    f->prow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    f->q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    f->rs = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    f->ap = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    f->ai = (int32_t *)malloc(sizeof(int32_t) * (size_t)(annz + 1));
    f->ax = (double *)malloc(sizeof(double) * (size_t)(annz + 1));
    double *x = (double *)calloc((size_t)n + 1, sizeof(double));
    int32_t *xi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int32_t *mark = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    if (!f->lp || !f->up || !f->li || !f->lx || !f->ui || !f->ux || !f->pinv || !f->prow || !f->q || !f->rs ||
        !f->ap || !f->ai || !f->ax || !x || !xi || !stack || !pos || !mark) {
        free(x), free(xi), free(stack), free(pos), free(mark);
        oracle_lu_free(f);
        return NULL;
    }
    memcpy(f->ap, ap, sizeof(int32_t) * (size_t)(n + 1));
    memcpy(f->ai, ai, sizeof(int32_t) * (size_t)annz);
    memcpy(f->ax, ax, sizeof(double) * (size_t)annz);
    for (int32_t i = 0; i < n; i++) {
        f->pinv[i] = -1;
        f->q[i] = q ? q[i] : i;
        f->rs[i] = 0.0;
    }
    /* row scale factors */
    for (int32_t j = 0; j < n; j++)
        for (int32_t p = ap[j]; p < ap[j + 1]; p++) {
            double a = fabs(ax[p]);
            if (scaling == 1) f->rs[ai[p]] += a;
            else if (scaling == 2 && a > f->rs[ai[p]]) f->rs[ai[p]] = a;
        }
    for (int32_t i = 0; i < n; i++) f->rs[i] = (scaling == 0 || f->rs[i] == 0.0) ? 1.0 : 1.0 / f->rs[i];

    f->status = ORACLE_OK;
    f->min_abs_pivot = INFINITY;
    f->max_abs_pivot = 0.0;
    int64_t lnz = 0, unz = 0;
    for (int32_t k = 0; k < n; k++) {
        int32_t col = f->q[k];
        f->lp[k] = lnz;
        f->up[k] = unz;
        if (lnz + n > lcap) {
            lcap = 2 * lcap + n;
            f->li = (int32_t *)realloc(f->li, sizeof(int32_t) * (size_t)lcap);
            f->lx = (double *)realloc(f->lx, sizeof(double) * (size_t)lcap);
        }
        if (unz + n > ucap) {
            ucap = 2 * ucap + n;
            f->ui = (int32_t *)realloc(f->ui, sizeof(int32_t) * (size_t)ucap);
            f->ux = (double *)realloc(f->ux, sizeof(double) * (size_t)ucap);
        }
        if (!f->li || !f->lx || !f->ui || !f->ux) {
            free(x), free(xi), free(stack), free(pos), free(mark);
            oracle_lu_free(f);
            return NULL;
        }
        /* x = L \ (R*A)(:,col), sparse */
        int32_t bnz = ap[col + 1] - ap[col];
        int32_t top = reach(f, n, ai + ap[col], bnz, xi, stack, pos, mark, k + 1);
        for (int32_t p = top; p < n; p++) x[xi[p]] = 0.0;
        for (int32_t p = ap[col]; p < ap[col + 1]; p++) x[ai[p]] = f->rs[ai[p]] * ax[p];
        for (int32_t px = top; px < n; px++) {
            int32_t i = xi[px];
            int32_t j = f->pinv[i];
            if (j < 0) continue; /* not yet pivotal: stays in the L part */
            double xj = x[i];
            for (int64_t p = f->lp[j]; p < f->lp[j + 1]; p++) x[f->li[p]] -= f->lx[p] * xj;
        }
        /* pivot search among non-pivotal rows */
        double amax = -1.0;
        int32_t ipiv = -1;
        for (int32_t px = top; px < n; px++) {
            int32_t i = xi[px];
            if (f->pinv[i] < 0) {
                double a = fabs(x[i]);
                if (a > amax) {
                    amax = a;
                    ipiv = i;
                }
            } else {
                f->ui[unz] = f->pinv[i];
                f->ux[unz++] = x[i];
            }
        }
        if (ipiv < 0 || amax <= 0.0) {
            /* structurally or numerically singular column: take any non-pivotal row, keep going
             * (UMFPACK also completes the factorisation and returns the singular warning) */
            f->status = ORACLE_SINGULAR;
            if (ipiv < 0)
                for (int32_t i = 0; i < n; i++)
                    if (f->pinv[i] < 0 && mark[i] != k + 1) {
                        ipiv = i;
                        x[i] = 0.0;
                        break;
                    }
            if (ipiv < 0) /* every unmarked row is taken: pick a marked non-pivotal one */
                for (int32_t px = top; px < n; px++)
                    if (f->pinv[xi[px]] < 0) {
                        ipiv = xi[px];
                        break;
                    }
        } else if (f->pinv[col] < 0 && mark[col] == k + 1 && fabs(x[col]) >= pivot_tol * amax) {
            ipiv = col; /* diagonal preferred when acceptable */
        }
        double pivot = x[ipiv];
        if (ipiv != col) f->nswaps_parity ^= 1;
        f->ui[unz] = k;
        f->ux[unz++] = pivot;
        f->pinv[ipiv] = k;
        f->prow[k] = ipiv;
        if (fabs(pivot) < f->min_abs_pivot) f->min_abs_pivot = fabs(pivot);
        if (fabs(pivot) > f->max_abs_pivot) f->max_abs_pivot = fabs(pivot);
        for (int32_t px = top; px < n; px++) {
            int32_t i = xi[px];
            if (f->pinv[i] < 0) {
                f->li[lnz] = i;
                f->lx[lnz++] = (pivot != 0.0) ? x[i] / pivot : 0.0;
            }
            x[i] = 0.0;
        }
    }
    f->lp[n] = lnz;
    f->up[n] = unz;
    free(x), free(xi), free(stack), free(pos), free(mark);
    return f;
}

/* one pass: x = Q * (U \ (L \ (P * R * b))) */
static void lu_apply(const OracleLU *f, const double *b, double *x, double *w) {
    int32_t n = f->n;
    /* w indexed by ORIGINAL row while eliminating with L (L rows are original indices) */
    for (int32_t i = 0; i < n; i++) w[i] = f->rs[i] * b[i];
    for (int32_t k = 0; k < n; k++) {
        double wk = w[f->prow[k]];
        for (int64_t p = f->lp[k]; p < f->lp[k + 1]; p++) w[f->li[p]] -= f->lx[p] * wk;
    }
    /* y[k] = w[prow[k]]; back substitution with U stored by columns (diagonal last) */
    double *y = x; /* reuse x as pivot-step-indexed storage, then permute in place via w */
    for (int32_t k = 0; k < n; k++) y[k] = w[f->prow[k]];
    for (int32_t k = n - 1; k >= 0; k--) {
        int64_t pd = f->up[k + 1] - 1;
        double d = f->ux[pd];
        y[k] = (d != 0.0) ? y[k] / d : y[k];
        for (int64_t p = f->up[k]; p < pd; p++) y[f->ui[p]] -= f->ux[p] * y[k];
    }
    for (int32_t k = 0; k < n; k++) w[f->q[k]] = y[k];
    for (int32_t i = 0; i < n; i++) x[i] = w[i];
}

/* solve with up to `nrefine` steps of iterative refinement on A x = b (UMFPACK's default is 2
 * steps, stopped early when the residual no longer shrinks) */
int32_t oracle_lu_solve(const OracleLU *f, const double *b, double *x, int32_t nrefine) {
    int32_t n = f->n;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    if (!w || !r || !d) {
        free(w), free(r), free(d);
        return ORACLE_OUT_OF_MEMORY;
    }
    lu_apply(f, b, x, w);
    double prev = INFINITY;
    for (int32_t it = 0; it < nrefine; it++) {
        for (int32_t i = 0; i < n; i++) r[i] = b[i];
        for (int32_t j = 0; j < n; j++)
            for (int32_t p = f->ap[j]; p < f->ap[j + 1]; p++) r[f->ai[p]] -= f->ax[p] * x[j];
        double rn = 0.0;
        for (int32_t i = 0; i < n; i++)
            if (fabs(r[i]) > rn) rn = fabs(r[i]);
        if (rn == 0.0 || rn >= prev) break;
        prev = rn;
        lu_apply(f, r, d, w);
        for (int32_t i = 0; i < n; i++) x[i] += d[i];
    }
    free(w), free(r), free(d);
    return f->status;
}

/* determinant as mantissa * 10^exponent (shape of umfpack_di_get_determinant's Mx, Ex used at
 * interface_umfpack.c:187-195).  det(A) = sign(P) sign(Q) prod(u_kk) / prod(rs_i) */
static int32_t perm_parity(const int32_t *p, int32_t n) {
    char *seen = (char *)calloc((size_t)n + 1, 1);
    int32_t parity = 0;
    for (int32_t i = 0; i < n; i++) {
        if (seen[i]) continue;
        int32_t len = 0;
        for (int32_t j = i; !seen[j]; j = p[j]) {
            seen[j] = 1;
            len++;
        }
        if ((len & 1) == 0) parity ^= 1;
    }
    free(seen);
    return parity;
}

void oracle_lu_determinant(const OracleLU *f, double *mantissa, double *exponent) {
    int32_t n = f->n;
    double m = 1.0, e = 0.0;
    for (int32_t k = 0; k < n; k++) {
        double d = f->ux[f->up[k + 1] - 1] / f->rs[f->prow[k]];
        if (d == 0.0) {
            *mantissa = 0.0;
            *exponent = 0.0;
            return;
        }
        m *= d;
        while (fabs(m) >= 10.0) m /= 10.0, e += 1.0;
        while (fabs(m) < 1.0) m *= 10.0, e -= 1.0;
    }
    if (perm_parity(f->prow, n) ^ perm_parity(f->q, n)) m = -m;
    *mantissa = m;
    *exponent = e;
}

/* rcond estimate in UMFPACK's cheap sense: min|u_kk| / max|u_kk| (UMFPACK_RCOND, user guide) */
double oracle_lu_rcond(const OracleLU *f) {
    if (f->max_abs_pivot == 0.0 || !isfinite(f->min_abs_pivot)) return 0.0;
    return f->min_abs_pivot / f->max_abs_pivot;
}

/* ---- finite-difference Laplacian: the K-bar / K-check triplets of Fdm2d::get_matrices_sps -----------------------------------
 * Plain restatement of /root/reference/russell_pde/src/fdm_2d.rs:603-649 (loop over the unknown nodes and their molecule),
 * :376-386 (molecule), :944-979 (mirrored / wrapped ghost nodes) and equation_handler.rs:153-190 (local numbers iu / ip).
 * nz > 1: the 7-point analogue with the same rules along z (the reference has no 3D operator; checker of the repo's extension).
 * Pinned on the dense matrices the reference's own tests print (fdm_2d.rs:1040-1207 -> tests/golden/fdm2d_reference_cases.json).
 * Output arrays must hold 7 entries per node; returns nnz(K-bar), *nnz_check = nnz(K-check); local (ntot int32) = iu or ip. */
int64_t oracle_fdm_sps(int32_t nx, int32_t ny, int32_t nz, int32_t px, int32_t py, int32_t pz, int32_t sym, const uint8_t *presc,
                       double dx, double dy, double dz, double kx, double ky, double kz, double alpha, int32_t *local,
                       int32_t *bar_i, int32_t *bar_j, double *bar_v, int32_t *chk_i, int32_t *chk_j, double *chk_v, int64_t *nnz_check) {
    const int64_t nxy = (int64_t)nx * ny, ntot = nxy * nz;
    int32_t iu = 0, ip = 0;
    for (int64_t e = 0; e < ntot; e++) local[e] = (presc && presc[e]) ? ip++ : iu++;
    const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
    double mol[7];
    mol[0] = 2.0 * (kx / dx2 + ky / dy2 + (nz > 1 ? kz / dz2 : 0.0));
    mol[1] = mol[2] = -kx / dx2;
    mol[3] = mol[4] = -ky / dy2;
    mol[5] = mol[6] = nz > 1 ? -kz / dz2 : 0.0;
    const int nb = nz > 1 ? 7 : 5;
    int64_t nbar = 0, nchk = 0;
    for (int64_t m = 0; m < ntot; m++) {
        if (presc && presc[m]) continue;
        const int32_t i = (int32_t)(m % nx), j = (int32_t)((m / nx) % ny), k = (int32_t)(m / nxy);
        int64_t nn[7];
        nn[0] = m;
        if (px) {
            nn[1] = i != 0 ? m - 1 : m + (nx - 1);
            nn[2] = i != nx - 1 ? m + 1 : m - (nx - 1);
        } else {
            nn[1] = i != 0 ? m - 1 : m + 1;
            nn[2] = i != nx - 1 ? m + 1 : m - 1;
        }
        if (py) {
            nn[3] = j != 0 ? m - nx : m + (int64_t)(ny - 1) * nx;
            nn[4] = j != ny - 1 ? m + nx : m - (int64_t)(ny - 1) * nx;
        } else {
            nn[3] = j != 0 ? m - nx : m + nx;
            nn[4] = j != ny - 1 ? m + nx : m - nx;
        }
        if (pz) {
            nn[5] = k != 0 ? m - nxy : m + (int64_t)(nz - 1) * nxy;
            nn[6] = k != nz - 1 ? m + nxy : m - (int64_t)(nz - 1) * nxy;
        } else {
            nn[5] = k != 0 ? m - nxy : m + nxy;
            nn[6] = k != nz - 1 ? m + nxy : m - nxy;
        }
        for (int b = 0; b < nb; b++) {
            const int64_t n = nn[b];
            double val = mol[b];
            if (m == n) val += alpha;
            if (!px && (i == 0 || i == nx - 1)) val /= 2.0;
            if (!py && (j == 0 || j == ny - 1)) val /= 2.0;
            if (nz > 1 && !pz && (k == 0 || k == nz - 1)) val /= 2.0;
            if (presc && presc[n]) {
                chk_i[nchk] = local[m], chk_j[nchk] = local[n], chk_v[nchk] = val;
                nchk++;
            } else if (!((sym == 1 && m < n) || (sym == 2 && m > n))) {
                bar_i[nbar] = local[m], bar_j[nbar] = local[n], bar_v[nbar] = val;
                nbar++;
            }
        }
    }
    *nnz_check = nchk;
    return nbar;
}

/* ---- finite-difference Laplacian, Lagrange-multiplier form: the triplets of Fdm2d::get_matrices_lmm --------------------------
 * Plain restatement of /root/reference/russell_pde/src/fdm_2d.rs:672-748: M = [K C^T; C 0] of order neq + nlag -- every node's
 * molecule (fdm_2d.rs:692-710: entries above / below the diagonal skipped for lower / upper storage, alpha on the diagonal, rows
 * of boundary nodes halved), then per prescribed node, ascending (equation_handler.rs:153-190), the entries of C (row neq + ip,
 * column m) and C^T (fdm_2d.rs:713-728: lower storage keeps C, upper C^T, general both, C first).
 * nz > 1: the 7-point analogue.  Pinned on the dense M / C the reference's own test prints (fdm_2d.rs:1094-1131 ->
 * tests/golden/fdm2d_reference_cases.json).  Output arrays must hold 7 ntot + 2 nlag entries; returns nnz(M); local = ip of the
 * prescribed nodes (others untouched). */
int64_t oracle_fdm_lmm(int32_t nx, int32_t ny, int32_t nz, int32_t px, int32_t py, int32_t pz, int32_t sym, const uint8_t *presc,
                       double dx, double dy, double dz, double kx, double ky, double kz, double alpha, int32_t *mm_i, int32_t *mm_j,
                       double *mm_v, int64_t *nlag_out) {
    const int64_t nxy = (int64_t)nx * ny, ntot = nxy * nz;
    const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
    double mol[7];
    mol[0] = 2.0 * (kx / dx2 + ky / dy2 + (nz > 1 ? kz / dz2 : 0.0));
    mol[1] = mol[2] = -kx / dx2;
    mol[3] = mol[4] = -ky / dy2;
    mol[5] = mol[6] = nz > 1 ? -kz / dz2 : 0.0;
    const int nb = nz > 1 ? 7 : 5;
    int64_t nnz = 0;
    for (int64_t m = 0; m < ntot; m++) {
        const int32_t i = (int32_t)(m % nx), j = (int32_t)((m / nx) % ny), k = (int32_t)(m / nxy);
        int64_t nn[7];
        nn[0] = m;
        nn[1] = px ? (i != 0 ? m - 1 : m + (nx - 1)) : (i != 0 ? m - 1 : m + 1);
        nn[2] = px ? (i != nx - 1 ? m + 1 : m - (nx - 1)) : (i != nx - 1 ? m + 1 : m - 1);
        nn[3] = py ? (j != 0 ? m - nx : m + (int64_t)(ny - 1) * nx) : (j != 0 ? m - nx : m + nx);
        nn[4] = py ? (j != ny - 1 ? m + nx : m - (int64_t)(ny - 1) * nx) : (j != ny - 1 ? m + nx : m - nx);
        nn[5] = pz ? (k != 0 ? m - nxy : m + (int64_t)(nz - 1) * nxy) : (k != 0 ? m - nxy : m + nxy);
        nn[6] = pz ? (k != nz - 1 ? m + nxy : m - (int64_t)(nz - 1) * nxy) : (k != nz - 1 ? m + nxy : m - nxy);
        for (int b = 0; b < nb; b++) {
            const int64_t n = nn[b];
            if ((sym == 1 && m < n) || (sym == 2 && m > n)) continue;
            double val = mol[b];
            if (m == n) val += alpha;
            if (!px && (i == 0 || i == nx - 1)) val /= 2.0;
            if (!py && (j == 0 || j == ny - 1)) val /= 2.0;
            if (nz > 1 && !pz && (k == 0 || k == nz - 1)) val /= 2.0;
            mm_i[nnz] = (int32_t)m, mm_j[nnz] = (int32_t)n, mm_v[nnz] = val;
            nnz++;
        }
    }
    int64_t ip = 0;
    for (int64_t m = 0; m < ntot; m++) {
        if (!(presc && presc[m])) continue;
        if (sym != 2) mm_i[nnz] = (int32_t)(ntot + ip), mm_j[nnz] = (int32_t)m, mm_v[nnz] = 1.0, nnz++; /* C */
        if (sym != 1) mm_i[nnz] = (int32_t)m, mm_j[nnz] = (int32_t)(ntot + ip), mm_v[nnz] = 1.0, nnz++; /* C^T */
        ip++;
    }
    *nlag_out = ip;
    return nnz;
}

// matching.cpp -- maximum-product bipartite matching with scaling (host side of the "initialize" phase).
//
// Static pivoting (pivots searched only inside a supernode's pivot block, tiny pivots perturbed) is safe when the
// matrix has a strong diagonal.  UMFPACK -- the reference's default backend -- reaches the same goal dynamically with
// threshold partial pivoting across the whole front; a solver that fixes its data layout before the numbers are known
// gets it from a pre-permutation instead: find the row permutation that maximises the product of the diagonal
// magnitudes, and the row / column scalings that make every diagonal entry 1 and every off-diagonal entry <= 1 in
// magnitude (Duff & Koster, "On algorithms for permuting large entries to the diagonal of a sparse matrix",
// SIAM J. Matrix Anal. Appl. 22(4), 2001: the MC64 "job 5" problem; cuDSS does its matching in the analysis phase of
// the reference's interface_cudss.cu:361 the same way).  This file restates that published algorithm:
// shortest augmenting paths (Dijkstra on reduced costs) for the assignment problem with costs
//     c_ij = log(max_k |a_kj|) - log|a_ij|  >= 0,
// dual variables u (rows), w (columns) with  c_ij - u_i - w_j >= 0  and equality on matched entries, from which
//     dr_i = exp(u_i),   dc_j = exp(w_j) / max_k |a_kj|.
// Deterministic: no hashing, ties broken by index.
#include "matching.hpp"

#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <queue>
#include <utility>
#include <vector>

namespace hipmf {

bool diagonal_is_weak(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, double threshold, bool pairs) {
    for (int32_t i = 0; i < n; i++) {
        double d = 0.0, mx = 0.0;
        for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
            const double a = std::fabs(v[p]);
            // (pairs: the real or the imaginary part of the complex diagonal entry -- the paired pivot search takes the larger)
            if (ci[p] == i || (pairs && ci[p] == (i ^ 1))) d += a; // duplicates were summed by the caller's COO -> CSR; a split diagonal is still counted
            mx = a > mx ? a : mx;
        }
        if (!(d >= threshold * mx) || d == 0.0) return true;
    }
    return false;
}

int32_t max_product_matching(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, std::vector<int32_t> &mrow,
                             std::vector<double> &dr, std::vector<double> &dc) {
    const double INF = std::numeric_limits<double>::infinity();
    const int64_t nnz = rp[n];
    // CSC copy with costs; exact zeros are not edges
    std::vector<int64_t> cp((size_t)n + 1, 0);
    for (int64_t k = 0; k < nnz; k++)
        if (v[k] != 0.0 && std::isfinite(v[k])) cp[ci[k] + 1]++;
    for (int32_t j = 0; j < n; j++) cp[j + 1] += cp[j];
    std::vector<int32_t> ri((size_t)cp[n]);
    std::vector<double> cost((size_t)cp[n]);
    std::vector<double> cmax((size_t)n, 0.0);
    {
        std::vector<int64_t> w(cp.begin(), cp.end() - 1);
        for (int32_t i = 0; i < n; i++)
            for (int32_t p = rp[i]; p < rp[i + 1]; p++)
                if (v[p] != 0.0 && std::isfinite(v[p])) {
                    const int32_t j = ci[p];
                    const double a = std::fabs(v[p]);
                    const int64_t q = w[j]++;
                    ri[q] = i;
                    cost[q] = a; // magnitude for now
                    cmax[j] = a > cmax[j] ? a : cmax[j];
                }
    }
    for (int32_t j = 0; j < n; j++) {
        if (cp[j + 1] == cp[j]) return -1; // empty column: structurally singular
        const double lm = std::log(cmax[j]);
        for (int64_t q = cp[j]; q < cp[j + 1]; q++) cost[q] = lm - std::log(cost[q]);
    }
    // Initial duals and greedy matching on tight edges.  Any u gives feasible duals with w_j = min_i (c_ij - u_i); two starts are tried
    // and the one that leaves fewer columns unmatched is kept (ties: the first):
    //   A  u_i = min_j c_ij                       (the costs are normalised by COLUMN maxima: the textbook start)
    //   B  u_i = -log max_j |a_ij| first, i.e. the same start on costs normalised by ROW maxima, then one row pass and one column pass.
    // On a matrix whose ROWS are scaled over many decades the column maxima are set by whichever large row touches a column, start A's
    // tight edges crowd onto a few rows (38 720 rows, 1.7 M entries, rows scaled 10^U(-6,6): 28 479 columns unmatched, 10 s of searches)
    // and start B leaves 7 380; on a matrix whose COLUMNS are badly scaled it is the other way round.
    for (int32_t j = 0; j < n; j++)
        for (int64_t q = cp[j]; q < cp[j + 1]; q++)
            if (!(cost[q] < INF)) return -1;
    std::vector<double> u, w;
    std::vector<int32_t> mcol;
    std::vector<int64_t> mptr; // CSC position of the matched entry of a column
    auto col_pass = [&](const std::vector<double> &uu, std::vector<double> &ww) {
        ww.assign((size_t)n, INF);
        for (int32_t j = 0; j < n; j++)
            for (int64_t q = cp[j]; q < cp[j + 1]; q++) {
                const double rc = cost[q] - uu[ri[q]];
                if (rc < ww[j]) ww[j] = rc;
            }
    };
    auto row_pass = [&](const std::vector<double> &ww, std::vector<double> &uu) {
        uu.assign((size_t)n, INF);
        for (int32_t j = 0; j < n; j++)
            for (int64_t q = cp[j]; q < cp[j + 1]; q++) {
                const double rc = cost[q] - ww[j];
                if (rc < uu[ri[q]]) uu[ri[q]] = rc;
            }
    };
    auto greedy = [&](const std::vector<double> &uu, const std::vector<double> &ww, std::vector<int32_t> &mr, std::vector<int32_t> &mc,
                      std::vector<int64_t> &mp) {
        mr.assign((size_t)n, -1), mc.assign((size_t)n, -1), mp.assign((size_t)n, -1);
        int32_t unmatched = 0;
        for (int32_t j = 0; j < n; j++) {
            int64_t best = -1;
            for (int64_t q = cp[j]; q < cp[j + 1]; q++)
                if (cost[q] - uu[ri[q]] == ww[j] && mc[ri[q]] < 0) {
                    best = q;
                    break;
                }
            if (best >= 0) mr[j] = ri[best], mc[ri[best]] = j, mp[j] = best;
            else unmatched++;
        }
        return unmatched;
    };
    {
        std::vector<double> zero((size_t)n, 0.0);
        row_pass(zero, u); // start A: u_i = min_j c_ij
        for (int32_t i = 0; i < n; i++)
            if (u[i] == INF) return -1; // empty row
        col_pass(u, w);
        const int32_t left_a = greedy(u, w, mrow, mcol, mptr);
        if (left_a > 0) {
            std::vector<double> ub((size_t)n, 0.0), wb, ub2;
            for (int32_t i = 0; i < n; i++) {
                double rmax = 0.0;
                for (int32_t p = rp[i]; p < rp[i + 1]; p++)
                    if (v[p] != 0.0 && std::isfinite(v[p])) rmax = std::fabs(v[p]) > rmax ? std::fabs(v[p]) : rmax;
                ub[i] = -std::log(rmax);
            }
            col_pass(ub, wb);
            row_pass(wb, ub2);
            col_pass(ub2, wb);
            std::vector<int32_t> mrow_b, mcol_b;
            std::vector<int64_t> mptr_b;
            const int32_t left_b = greedy(ub2, wb, mrow_b, mcol_b, mptr_b);
            if (left_b < left_a) u.swap(ub2), w.swap(wb), mrow.swap(mrow_b), mcol.swap(mcol_b), mptr.swap(mptr_b);
        }
    }
    // shortest augmenting paths
    std::vector<double> d((size_t)n, INF);
    std::vector<int32_t> pred((size_t)n, -1); // column from which a row was reached
    std::vector<int64_t> predq((size_t)n, -1); // CSC position of that edge
    std::vector<char> done((size_t)n, 0);
    std::vector<int32_t> touched, finalised;
    typedef std::pair<double, int32_t> Item;
    for (int32_t j0 = 0; j0 < n; j0++) {
        if (mrow[j0] >= 0) continue;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        touched.clear();
        finalised.clear();
        int32_t j = j0, sink = -1;
        double lsp = 0.0;
        // csp: length of the shortest path to a FREE row seen so far (Duff & Koster's pruning): a free row is a candidate sink as soon as
        // an edge reaches it, and nothing at or beyond csp is worth a place in the heap -- without it every search ran until a free row
        // was POPPED, with every row closer than that one pushed and finalised first (38 720 rows, 1.7 M entries, shuffled and badly
        // scaled: 6.7 s of a 6.9 s initialize)
        double csp = INF;
        for (;;) {
            for (int64_t q = cp[j]; q < cp[j + 1]; q++) {
                const int32_t i = ri[q];
                if (done[i]) continue;
                const double dn = lsp + (cost[q] - u[i] - w[j]);
                if (dn < csp && dn < d[i]) {
                    if (d[i] == INF) touched.push_back(i);
                    d[i] = dn;
                    pred[i] = j;
                    predq[i] = q;
                    if (mcol[i] < 0) csp = dn, sink = i; // (ties keep the first free row found: deterministic)
                    else heap.push(Item(dn, i));
                }
            }
            int32_t i = -1;
            while (!heap.empty()) {
                Item it = heap.top();
                if (it.first >= csp) break; // the best free row is at least as close as anything left
                heap.pop();
                if (!done[it.second] && it.first == d[it.second]) {
                    i = it.second;
                    break;
                }
            }
            if (i < 0) break; // the search is over: sink (if any) ends the shortest augmenting path
            done[i] = 1;
            finalised.push_back(i);
            lsp = d[i];
            j = mcol[i];
        }
        lsp = csp;
        if (sink < 0) {
            for (int32_t i : touched) d[i] = INF, done[i] = 0;
            return -1;
        }
        // dual update of the finalised rows, then augment, then re-tighten the columns of the tree
        for (int32_t i : finalised) u[i] += d[i] - lsp;
        for (int32_t i = sink; i >= 0;) {
            const int32_t jj = pred[i];
            const int32_t inext = mrow[jj]; // row that column jj gives up (-1 at the root j0)
            mrow[jj] = i;
            mcol[i] = jj;
            mptr[jj] = predq[i];
            i = inext;
        }
        for (int32_t i : finalised) {
            const int32_t jj = mcol[i];
            if (jj >= 0) w[jj] = cost[mptr[jj]] - u[i];
        }
        w[mcol[sink]] = cost[mptr[mcol[sink]]] - u[sink]; // (the sink is not among the finalised rows: it ended the search unpopped)
        for (int32_t i : touched) d[i] = INF, done[i] = 0;
    }
    dr.resize((size_t)n);
    dc.resize((size_t)n);
    for (int32_t i = 0; i < n; i++) dr[i] = std::exp(u[i]);
    for (int32_t j = 0; j < n; j++) dc[j] = std::exp(w[j]) / cmax[j];
    return 0;
}

int32_t paired_matching(int32_t n, const int32_t *rp, const int32_t *ci, const double *v, std::vector<int32_t> &mrow, std::vector<double> &dr,
                        std::vector<double> &dc) {
    if (n % 2 != 0) return -1;
    const int32_t nc = n / 2;
    std::vector<int32_t> rpc((size_t)nc + 1, 0), cic;
    std::vector<double> vc;
    cic.reserve((size_t)rp[n] / 4 + 1), vc.reserve((size_t)rp[n] / 4 + 1);
    for (int32_t i = 0; i < nc; i++) {
        // row 2 i of the real-equivalent form holds (Re, -Im) of complex entry (i, j) in columns 2 j, 2 j + 1 (ascending)
        for (int32_t p = rp[2 * i]; p < rp[2 * i + 1]; p++) {
            const int32_t j = ci[p] / 2;
            if (!cic.empty() && (int32_t)cic.size() > rpc[i] && cic.back() == j) vc.back() = std::hypot(vc.back(), v[p]);
            else cic.push_back(j), vc.push_back(std::fabs(v[p]));
        }
        rpc[(size_t)i + 1] = (int32_t)cic.size();
    }
    std::vector<int32_t> mc;
    std::vector<double> drc, dcc;
    const int32_t rc = max_product_matching(nc, rpc.data(), cic.data(), vc.data(), mc, drc, dcc);
    if (rc != 0) return rc;
    mrow.resize((size_t)n), dr.resize((size_t)n), dc.resize((size_t)n);
    for (int32_t k = 0; k < nc; k++) {
        mrow[2 * k] = 2 * mc[k], mrow[2 * k + 1] = 2 * mc[k] + 1;
        dr[2 * k] = dr[2 * k + 1] = drc[k];
        dc[2 * k] = dc[2 * k + 1] = dcc[k];
    }
    return 0;
}

} // namespace hipmf

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
    // initial duals and greedy matching on tight edges
    std::vector<double> u((size_t)n, INF), w((size_t)n, INF);
    for (int32_t j = 0; j < n; j++)
        for (int64_t q = cp[j]; q < cp[j + 1]; q++) u[ri[q]] = cost[q] < u[ri[q]] ? cost[q] : u[ri[q]];
    for (int32_t i = 0; i < n; i++)
        if (u[i] == INF) return -1; // empty row
    mrow.assign((size_t)n, -1);
    std::vector<int32_t> mcol((size_t)n, -1);
    std::vector<int64_t> mptr((size_t)n, -1); // CSC position of the matched entry of a column
    for (int32_t j = 0; j < n; j++) {
        int64_t best = -1;
        for (int64_t q = cp[j]; q < cp[j + 1]; q++) {
            const double rc = cost[q] - u[ri[q]];
            if (rc < w[j]) w[j] = rc;
        }
        for (int64_t q = cp[j]; q < cp[j + 1]; q++)
            if (cost[q] - u[ri[q]] == w[j] && mcol[ri[q]] < 0) {
                best = q;
                break;
            }
        if (best >= 0) {
            mrow[j] = ri[best];
            mcol[ri[best]] = j;
            mptr[j] = best;
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
        for (;;) {
            for (int64_t q = cp[j]; q < cp[j + 1]; q++) {
                const int32_t i = ri[q];
                if (done[i]) continue;
                const double dn = lsp + (cost[q] - u[i] - w[j]);
                if (dn < d[i]) {
                    if (d[i] == INF) touched.push_back(i);
                    d[i] = dn;
                    pred[i] = j;
                    predq[i] = q;
                    heap.push(Item(dn, i));
                }
            }
            int32_t i = -1;
            while (!heap.empty()) {
                Item it = heap.top();
                heap.pop();
                if (!done[it.second] && it.first == d[it.second]) {
                    i = it.second;
                    break;
                }
            }
            if (i < 0) break; // no augmenting path: structurally singular
            done[i] = 1;
            finalised.push_back(i);
            lsp = d[i];
            if (mcol[i] < 0) {
                sink = i;
                break;
            }
            j = mcol[i];
        }
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

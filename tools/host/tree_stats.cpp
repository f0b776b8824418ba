// Host-only: per-level statistics of the assembly tree (fronts, pivots, bytes of the solve panels) for a 2D / 3D Poisson grid.
//   g++ -O2 -std=c++17 -pthread tools/host/tree_stats.cpp russell_amd/csrc/symbolic.cpp -o build/tree_stats && build/tree_stats 2 1000
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../russell_amd/csrc/symbolic.hpp"
using namespace hipmf;

int main(int argc, char **argv) {
    const int dim = argc > 1 ? atoi(argv[1]) : 2, N = argc > 2 ? atoi(argv[2]) : 1000, lower = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t n = dim == 2 ? (int64_t)N * N : (int64_t)N * N * N;
    std::vector<int32_t> rp((size_t)n + 1, 0), ci;
    const int Z = dim == 2 ? 1 : N;
    for (int z = 0; z < Z; z++)
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) {
                const int64_t i = ((int64_t)z * N + y) * N + x;
                if (dim == 3 && z > 0) ci.push_back((int32_t)(i - (int64_t)N * N));
                if (y > 0) ci.push_back((int32_t)(i - N));
                if (x > 0) ci.push_back((int32_t)(i - 1));
                ci.push_back((int32_t)i);
                if (!lower) {
                    if (x + 1 < N) ci.push_back((int32_t)(i + 1));
                    if (y + 1 < N) ci.push_back((int32_t)(i + N));
                    if (dim == 3 && z + 1 < N) ci.push_back((int32_t)(i + (int64_t)N * N));
                }
                rp[(size_t)i + 1] = (int32_t)ci.size();
            }
    SymbolicOptions so;
    so.nd_leaf = 16, so.dense_leaves = true, so.symmetric_ldlt = lower != 0;
    Symbolic S;
    const int rc = analyse((int32_t)n, rp.data(), ci.data(), lower != 0, so, S);
    printf("rc %d n %lld supernodes %d levels %d nnzL %lld\n", rc, (long long)n, S.nsuper, S.nlevels, (long long)S.nnz_l);
    printf("level fronts small big | sum_p pmax fmax | smallMB E_MB Ep_MB | sum_f_big sum_m_big nchild_max\n");
    double tot_s = 0, tot_e = 0, tot_ep = 0;
    for (int l = 0; l < S.nlevels; l++) {
        int64_t ns = 0, nb = 0, sp = 0, pmax = 0, fmax = 0, sf = 0, sm = 0, ncm = 0;
        double sb = 0, eb = 0, epb = 0;
        for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; k++) {
            const int s = S.level_sn[k];
            const int64_t p = S.npiv(s), m = S.nrow(s), f = p + m;
            sp += p, pmax = std::max(pmax, p), fmax = std::max(fmax, f);
            ncm = std::max<int64_t>(ncm, S.child_ptr[s + 1] - S.child_ptr[s]);
            if (f <= 64) ns++, sb += 8.0 * f * p + (m > 0 ? 8.0 * f * p : 8.0 * f * p);
            else nb++, eb += 8.0 * f * p, epb += 8.0 * f * p, sf += f, sm += m;
        }
        tot_s += sb, tot_e += eb, tot_ep += epb;
        printf("%3d %7lld %7lld %6lld | %8lld %5lld %5lld | %8.2f %8.2f %8.2f | %9lld %9lld %4lld\n", l, (long long)(ns + nb), (long long)ns, (long long)nb, (long long)sp,
               (long long)pmax, (long long)fmax, sb / 1e6, eb / 1e6, epb / 1e6, (long long)sf, (long long)sm, (long long)ncm);
    }
    printf("big fronts per level: count | f<=128&p<=32 | f<=192&p<=48 | f<=256&p<=64 | avg f, avg p, avg children\n");
    for (int l = 0; l < S.nlevels; l++) {
        int64_t nb = 0, c1 = 0, c2 = 0, c3 = 0, sf = 0, sp = 0, sc = 0;
        for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; k++) {
            const int s = S.level_sn[k];
            const int64_t p = S.npiv(s), f = p + S.nrow(s);
            if (f <= 64) continue;
            nb++, sf += f, sp += p, sc += S.child_ptr[s + 1] - S.child_ptr[s];
            c1 += f <= 128 && p <= 32, c2 += f <= 192 && p <= 48, c3 += f <= 256 && p <= 64;
        }
        if (nb) printf("%3d %6lld | %6lld %6lld %6lld | %6.1f %6.1f %5.2f\n", l, (long long)nb, (long long)c1, (long long)c2, (long long)c3, (double)sf / nb, (double)sp / nb, (double)sc / nb);
    }
    printf("total small (both passes) %.1f MB, E %.1f MB, E' %.1f MB\n", tot_s / 1e6, tot_e / 1e6, tot_ep / 1e6);
    return 0;
}

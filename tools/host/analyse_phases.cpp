// Host-only driver of the symbolic analysis (russell_amd/csrc/symbolic.cpp): phases of analyse() on a 2D / 3D Poisson grid, no device needed.
//   g++ -O2 -std=c++17 -pthread tools/host/analyse_phases.cpp russell_amd/csrc/symbolic.cpp -o build/analyse_phases && build/analyse_phases 3 200 1
// arguments: dimension (2 | 3), points per side, lower triangle only (1: symmetric-lower storage, the L D L^T layout of BASELINE config 4)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../russell_amd/csrc/symbolic.hpp"
using namespace hipmf;

int main(int argc, char **argv) {
    const int dim = argc > 1 ? atoi(argv[1]) : 2, N = argc > 2 ? atoi(argv[2]) : 1000, lower = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t n = dim == 2 ? (int64_t)N * N : (int64_t)N * N * N;
    std::vector<int32_t> rp((size_t)n + 1, 0), ci;
    ci.reserve((size_t)n * (dim == 2 ? 5 : 7));
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
    so.nd_leaf = 16, so.dense_leaves = true, so.symmetric_ldlt = lower != 0; // (what Solver::initialize sets)
    if (const char *e = getenv("HIPMF_ND_THREADS")) so.nd_threads = atoi(e);
    for (int rep = 0; rep < 2; rep++) {
        Symbolic S;
        auto t0 = std::chrono::steady_clock::now();
        const int rc = analyse((int32_t)n, rp.data(), ci.data(), lower != 0, so, S);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("analyse rc %d: %.3f s | graph %.3f, ordering %.3f, etree + postorder %.3f, column counts + supernodes %.3f, row structures %.3f, levels + layout %.3f, assembly map %.3f | "
               "n %lld, supernodes %d, levels %d, max front %d, nnz(L) %lld\n",
               rc, t, S.seconds_phase[0], S.seconds_phase[1], S.seconds_phase[2], S.seconds_phase[3], S.seconds_phase[4], S.seconds_phase[5], S.seconds_phase[6],
               (long long)n, S.nsuper, S.nlevels, S.max_front, (long long)S.nnz_l);
    }
    return 0;
}

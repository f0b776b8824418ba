// Differential / sanity fuzzing of the orderings without a device: random small patterns (empty, dense, hubs, two components, rows without a
// diagonal entry, up to 3 000 vertices) through analyse() with the dissection, Ordering::Amd and Ordering::Best -- every result must be a
// permutation, and the analysis with host threads (subtree-parallel elimination tree, column counts, row structures) must equal the serial one.
//   g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -I russell_amd/csrc tools/host/fuzz_orderings.cpp russell_amd/csrc/symbolic.cpp -o build/fuzz_orderings && build/fuzz_orderings 3000
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include "../../russell_amd/csrc/symbolic.hpp"
using namespace hipmf;
int main(int argc, char **argv) {
    const int cases = atoi(argv[1]);
    std::mt19937 rng(12345);
    int bad = 0;
    for (int c = 0; c < cases; c++) {
        const int kind = rng() % 6;
        int n = 1 + rng() % (kind == 5 ? 3000 : 90);
        std::vector<std::vector<int32_t>> rows(n);
        const double dens = kind == 0 ? 0.0 : (kind == 1 ? 1.0 : (kind == 2 ? 0.5 : (double)(1 + rng() % 6) / n));
        for (int i = 0; i < n; i++) {
            if (rng() % 7 != 0) rows[i].push_back(i); // (some rows without a diagonal entry)
            for (int j = 0; j < n; j++) if (j != i && (rng() % 100000) < dens * 100000) rows[i].push_back(j);
        }
        if (kind == 3 && n > 3) for (int j = 1; j < n; j++) rows[0].push_back(j); // a hub row
        if (kind == 4 && n > 10) { for (int i = 0; i < n / 2; i++) { rows[i].erase(std::remove_if(rows[i].begin(), rows[i].end(), [&](int32_t j){ return j >= n / 2; }), rows[i].end()); } for (int i = n / 2; i < n; i++) rows[i].erase(std::remove_if(rows[i].begin(), rows[i].end(), [&](int32_t j){ return j < n / 2; }), rows[i].end()); } // two components
        std::vector<int32_t> rp(n + 1, 0), ci;
        for (int i = 0; i < n; i++) { std::sort(rows[i].begin(), rows[i].end()); rows[i].erase(std::unique(rows[i].begin(), rows[i].end()), rows[i].end()); for (int32_t j : rows[i]) ci.push_back(j); rp[i + 1] = (int32_t)ci.size(); }
        if (ci.empty()) ci.push_back(0);
        for (int ord : {0, 2, 3}) {
            SymbolicOptions so; so.nd_leaf = 16, so.dense_leaves = true; so.ordering = ord; so.parallel_min_n = 0; so.nd_threads = 2 + c % 4;
            so.parallel_chunk_min = 1 + c % 40; // (small subtrees: many chunks, records across them)
            if (c % 5 == 0) so.dense_row_factor = 0.0;
            Symbolic S, S1;
            int rc = analyse(n, rp.data(), ci.data(), false, so, S);
            {
                // the same analysis without host threads: elimination tree, column counts and row structures in their serial form
                SymbolicOptions s1 = so;
                s1.nd_threads = 1;
                const int rc1 = analyse(n, rp.data(), ci.data(), false, s1, S1);
                const bool same = rc1 == rc && (rc != 0 || (S.perm == S1.perm && S.sn_first == S1.sn_first && S.sn_rowptr == S1.sn_rowptr && S.sn_rows == S1.sn_rows &&
                                                            S.rel == S1.rel && S.front_off == S1.front_off && S.sn_parent == S1.sn_parent));
                if (!same) { bad++; printf("case %d kind %d n %d ordering %d: threaded analysis differs from the serial one\n", c, kind, n, ord); }
            }
            bool ok = rc == 0;
            std::vector<char> seen(n, 0);
            if (ok) for (int k = 0; k < n; k++) { int32_t v = S.perm[k]; if (v < 0 || v >= n || seen[v]) { ok = false; break; } seen[v] = 1; }
            if (!ok) { bad++; printf("case %d kind %d n %d ordering %d: rc %d\n", c, kind, n, ord, rc); }
        }
    }
    printf("%d cases, %d bad\n", cases, bad);
    return bad != 0;
}

// symbolic.cpp -- ordering, elimination tree, supernodes, frontal index sets, assembly maps.
// See symbolic.hpp for the role this plays at the reference's solver boundary.
#include "symbolic.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <iterator>
#include <map>
#include <mutex>
#include <set>
#include <numeric>
#include <thread>

namespace hipmf {
namespace {

using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// fn(begin, end) over [0, n) in contiguous chunks on up to `threads` host threads (index-independent work only: the result
// does not depend on the number of threads)
template <typename Fn> static void parallel_ranges(int32_t n, int threads, Fn fn) {
    threads = std::max(1, std::min(threads, n / 65536));
    if (threads <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        const int32_t b = (int32_t)((int64_t)n * t / threads), e = (int32_t)((int64_t)n * (t + 1) / threads);
        pool.emplace_back([=]() { fn(b, e); });
    }
    for (auto &th : pool) th.join();
}
static int host_threads(const SymbolicOptions &opt) {
    return opt.nd_threads > 0 ? opt.nd_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
}

struct Graph {
    int32_t n = 0;
    std::vector<int64_t> ptr;
    std::vector<int32_t> adj;
};

// pattern of A + A^T without the diagonal; adjacency lists ascending and duplicate-free
static int build_graph(int32_t n, const int32_t *rp, const int32_t *ci, Graph &g, int threads) {
    g.n = n;
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; i++) {
        if (rp[i + 1] < rp[i]) return -1;
        for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
            int32_t j = ci[p];
            if (j < 0 || j >= n) return -2;
            if (j != i) {
                cnt[i + 1]++;
                cnt[j + 1]++;
            }
        }
    }
    for (int32_t i = 0; i < n; i++) cnt[i + 1] += cnt[i];
    std::vector<int32_t> raw((size_t)cnt[n]);
    std::vector<int64_t> w(cnt.begin(), cnt.end() - 1);
    for (int32_t i = 0; i < n; i++)
        for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
            int32_t j = ci[p];
            if (j != i) {
                raw[w[i]++] = j;
                raw[w[j]++] = i;
            }
        }
    // sort + dedupe every list in place, then compact
    g.ptr.assign((size_t)n + 1, 0);
    parallel_ranges(n, threads, [&](int32_t i0, int32_t i1) {
        for (int32_t i = i0; i < i1; i++) {
            auto b = raw.begin() + cnt[i], e = raw.begin() + cnt[i + 1];
            std::sort(b, e);
            g.ptr[i + 1] = (int64_t)(std::unique(b, e) - b);
        }
    });
    for (int32_t i = 0; i < n; i++) g.ptr[i + 1] += g.ptr[i];
    g.adj.resize((size_t)g.ptr[n]);
    parallel_ranges(n, threads, [&](int32_t i0, int32_t i1) {
        for (int32_t i = i0; i < i1; i++) std::copy(raw.begin() + cnt[i], raw.begin() + cnt[i] + (g.ptr[i + 1] - g.ptr[i]), g.adj.begin() + g.ptr[i]);
    });
    return 0;
}

static void permute_graph(const Graph &g, const std::vector<int32_t> &perm, const std::vector<int32_t> &pinv, Graph &out, int threads) {
    int32_t n = g.n;
    out.n = n;
    out.ptr.assign((size_t)n + 1, 0);
    out.adj.resize(g.adj.size());
    for (int32_t k = 0; k < n; k++) out.ptr[k + 1] = out.ptr[k] + (g.ptr[perm[k] + 1] - g.ptr[perm[k]]);
    parallel_ranges(n, threads, [&](int32_t k0, int32_t k1) {
        for (int32_t k = k0; k < k1; k++) {
            int32_t v = perm[k];
            int64_t o = out.ptr[k];
            for (int64_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) out.adj[o++] = pinv[g.adj[p]];
            std::sort(out.adj.begin() + out.ptr[k], out.adj.begin() + o);
        }
    });
}

// ------------------------------------------------------------------------------------------------
// Nested dissection by breadth-first level structures.  A region is split by the level set of a
// BFS from a pseudo-peripheral vertex that balances the two sides with the fewest vertices; the
// separator is numbered last.  Regions of <= 64 vertices are numbered by minimum degree on a
// one-word-per-row bitset elimination graph.
//
// The regions of the dissection tree are independent once their parent has been split, so they are
// processed by a small pool of host threads (the first splits are serial, the rest of the tree is wide).
// The result does not depend on the number of threads or on the schedule: a region's outcome is a function
// of its own vertex list only; region ids and visit stamps are merely unique, never compared for order.
// ------------------------------------------------------------------------------------------------
// per-vertex state of the dissection in ONE record: a breadth-first search tests the region and the visit stamp of every neighbour
// and writes stamp and level of the ones it takes -- one cache line per neighbour instead of three
struct NDVertex {
    std::atomic<int32_t> part; // region id, -1 once numbered (read across regions: atomic, relaxed)
    int32_t stamp;             // visit stamp (vertices of the own region only)
    int32_t lev;               // BFS level / local index (vertices of the own region only)
    int32_t pad;
};
struct NDShared {
    const Graph &g;
    std::unique_ptr<NDVertex[]> vx;
    std::vector<int32_t> verts;                   // region vertex lists (disjoint segments)
    std::atomic<int32_t> cur_stamp{0}, next_id{1};
    // Regions of at most `interval_cutoff` vertices whose parent is larger, as (first position, size) in the new numbering.  Every neighbour
    // of a region's vertex that is numbered BEFORE it lies in the region itself (what is outside are separators and hubs, numbered after):
    // the elimination tree of such an interval can be built without looking at anything else (etree below).
    int32_t interval_cutoff = 0;
    std::mutex interval_mu;
    std::vector<std::pair<int32_t, int32_t>> intervals;
    explicit NDShared(const Graph &gr) : g(gr) {}
    int32_t region_of(int32_t v) const { return vx[v].part.load(std::memory_order_relaxed); }
    void set_region(int32_t v, int32_t id) { vx[v].part.store(id, std::memory_order_relaxed); }
};

// per-thread scratch, sized by the largest region the thread has seen
struct NDScratch {
    std::vector<int32_t> queue, tmp, lvl_ptr, comp_ptr;
    void reserve(int32_t size) {
        if ((int32_t)queue.size() < size) {
            queue.resize((size_t)size);
            tmp.resize((size_t)size);
        }
    }
};

struct NDRegion {
    int32_t begin, end, pos, id;
    bool connected;
    int32_t parent_size; // vertices of the region this one was cut out of (the whole graph: INT32_MAX)
    // Round 6: side A of a split is the head of the parent's level structure (levels below the separator + the thinned separator
    // vertices), and its first vertex is the parent's root: the breadth-first search from that vertex inside A would visit exactly that
    // head again, in the same order.  The split therefore hands A what that search would have produced -- its eccentricity and the
    // candidate of the pseudo-peripheral iteration (the minimum-degree vertex of its last level) -- and A starts with its SECOND search.
    // Same permutation bit for bit (tests/test_tree_solve_cpu.py's golden hash), one search of three less on every A side.
    int32_t first_ecc = -1, first_cand = -1;
};

// BFS inside region `id` from `root`; fills t.queue[0..count) in visit order, w.lev, t.lvl_ptr.
// Returns the eccentricity (number of levels - 1).
static int32_t bfs_region(NDShared &w, NDScratch &t, int32_t root, int32_t id, int32_t &count) {
    const Graph &g = w.g;
    const int32_t st = w.cur_stamp.fetch_add(1, std::memory_order_relaxed) + 1;
    int32_t head = 0, tail = 0;
    NDVertex *vx = w.vx.get();
    int32_t *queue = t.queue.data();
    queue[tail++] = root;
    vx[root].stamp = st;
    vx[root].lev = 0;
    t.lvl_ptr.clear();
    t.lvl_ptr.push_back(0);
    int32_t curlev = 0, lev_end = 1; // the current level is queue[lvl_ptr.back(), lev_end)
    while (head < tail) {
        if (head == lev_end) { // first vertex of the next level
            curlev++;
            t.lvl_ptr.push_back(head);
            lev_end = tail;
        }
        const int32_t v = queue[head++];
        for (int64_t p = g.ptr[v], pe = g.ptr[v + 1]; p < pe; p++) {
            const int32_t u = g.adj[p];
            NDVertex &x = vx[u];
            if (x.part.load(std::memory_order_relaxed) == id && x.stamp != st) {
                x.stamp = st;
                x.lev = curlev + 1;
                queue[tail++] = u;
            }
        }
    }
    t.lvl_ptr.push_back(tail);
    count = tail;
    return (int32_t)t.lvl_ptr.size() - 2;
}

static void leaf_min_degree(NDShared &w, int32_t begin, int32_t end, int32_t id, int32_t pos, std::vector<int32_t> &perm) {
    const Graph &g = w.g;
    int32_t s = end - begin;
    uint64_t a[64];
    int32_t ext[64];
    for (int32_t i = 0; i < s; i++) w.vx[w.verts[begin + i]].lev = i; // local index
    for (int32_t i = 0; i < s; i++) {
        int32_t v = w.verts[begin + i];
        a[i] = 0;
        ext[i] = 0;
        for (int64_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
            int32_t u = g.adj[p];
            if (w.region_of(u) == id) a[i] |= (uint64_t)1 << w.vx[u].lev;
            else ext[i]++;
        }
    }
    uint64_t alive = (s == 64) ? ~(uint64_t)0 : (((uint64_t)1 << s) - 1);
    for (int32_t step = 0; step < s; step++) {
        int32_t best = -1, bestdeg = 1 << 30;
        for (int32_t i = 0; i < s; i++) {
            if (!((alive >> i) & 1)) continue;
            int32_t deg = __builtin_popcountll(a[i] & alive) + ext[i];
            if (deg < bestdeg) {
                bestdeg = deg;
                best = i;
            }
        }
        int32_t v = w.verts[begin + best];
        perm[pos + step] = v;
        w.set_region(v, -1);
        alive &= ~((uint64_t)1 << best);
        uint64_t nb = a[best] & alive;
        for (uint64_t m = nb; m; m &= m - 1) {
            int32_t u = __builtin_ctzll(m);
            a[u] |= nb;
            a[u] &= ~((uint64_t)1 << u);
            if (ext[best] > ext[u]) ext[u] = ext[best];
        }
    }
}

// One region: number it (leaf / clique) or split it; the sub-regions are appended to `out` (last = to be processed first
// by a serial driver, which reproduces the depth-first order; any order gives the same permutation).
static void nd_process(NDShared &w, NDScratch &t, const SymbolicOptions &opt, int32_t leaf, NDRegion R, std::vector<int32_t> &perm,
                       std::vector<int32_t> &leaf_of, std::vector<NDRegion> &out) {
    const Graph &g = w.g;
    const int32_t size = R.end - R.begin;
    if (size <= 0) return;
    if (size <= w.interval_cutoff && R.parent_size > w.interval_cutoff) {
        std::lock_guard<std::mutex> lk(w.interval_mu);
        w.intervals.emplace_back(R.pos, size);
    }
    t.reserve(size);
    // A region that is not known to be connected and is going to be split: the first breadth-first search of the pseudo-peripheral
    // iteration (from the region's first vertex) visits exactly what the component sweep below would visit first, in the same order.
    // When it reaches every vertex the region is connected and the sweep -- one more pass over the region's adjacency -- is skipped.
    bool have_bfs = false;
    int32_t count0 = 0, ecc0 = 0;
    const bool known_first = R.first_cand >= 0 && size > leaf; // (side A of a split: connected, first search known -- see NDRegion)
    if (known_first) R.connected = true;
    if (!R.connected && size > leaf) {
        ecc0 = bfs_region(w, t, w.verts[R.begin], R.id, count0);
        if (count0 == size) R.connected = true, have_bfs = true;
    }
    if (!R.connected) {
        // connected components of the region (discovery order, deterministic)
        const int32_t st = w.cur_stamp.fetch_add(1, std::memory_order_relaxed) + 1;
        int32_t outn = 0;
        t.comp_ptr.clear();
        t.comp_ptr.push_back(0);
        for (int32_t k = R.begin; k < R.end; k++) {
            int32_t r = w.verts[k];
            if (w.vx[r].stamp == st) continue;
            int32_t head = outn;
            t.tmp[outn++] = r;
            w.vx[r].stamp = st;
            while (head < outn) {
                int32_t v = t.tmp[head++];
                for (int64_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
                    int32_t u = g.adj[p];
                    if (w.region_of(u) == R.id && w.vx[u].stamp != st) {
                        w.vx[u].stamp = st;
                        t.tmp[outn++] = u;
                    }
                }
            }
            t.comp_ptr.push_back(outn);
        }
        std::copy(t.tmp.begin(), t.tmp.begin() + size, w.verts.begin() + R.begin);
        int32_t ncomp = (int32_t)t.comp_ptr.size() - 1;
        if (ncomp > 1) {
            // independent subtrees: number them one after the other
            for (int32_t c = ncomp - 1; c >= 0; c--) {
                int32_t b = R.begin + t.comp_ptr[c], e = R.begin + t.comp_ptr[c + 1];
                int32_t id = w.next_id.fetch_add(1, std::memory_order_relaxed);
                for (int32_t k = b; k < e; k++) w.set_region(w.verts[k], id);
                out.push_back({b, e, R.pos + t.comp_ptr[c], id, true, size});
            }
            return;
        }
    }
    if (size <= leaf) {
        if (opt.dense_leaves && size > 1)
            for (int32_t k = R.begin; k < R.end; k++) leaf_of[w.verts[k]] = R.pos; // leaf label: its first position (unique)
        leaf_min_degree(w, R.begin, R.end, R.id, R.pos, perm);
        return;
    }
    // pseudo-peripheral vertex: repeat BFS from a minimum-degree vertex of the last level
    int32_t root = w.verts[R.begin], count = count0;
    int32_t ecc = known_first ? R.first_ecc : (have_bfs ? ecc0 : bfs_region(w, t, root, R.id, count));
    for (int32_t it = 0; it < 4; it++) {
        int32_t cand;
        if (known_first && it == 0) {
            cand = R.first_cand; // (what the loop below would pick from the last level of the search from `root`)
        } else {
            int32_t lb = t.lvl_ptr[t.lvl_ptr.size() - 2], le = t.lvl_ptr.back();
            cand = t.queue[lb];
            int64_t cdeg = g.ptr[cand + 1] - g.ptr[cand];
            for (int32_t k = lb + 1; k < le; k++) {
                int32_t v = t.queue[k];
                int64_t d = g.ptr[v + 1] - g.ptr[v];
                if (d < cdeg || (d == cdeg && v < cand)) {
                    cand = v;
                    cdeg = d;
                }
            }
        }
        if (cand == root) break;
        int32_t e2 = bfs_region(w, t, cand, R.id, count);
        // the structure rooted at `cand` is kept in both cases (same or larger eccentricity, valid structure)
        const bool grew = e2 > ecc;
        ecc = e2;
        root = cand;
        if (!grew) break;
    }
    if (ecc < 2) {
        // (nearly) a clique: cannot be dissected; number in BFS order
        for (int32_t k = 0; k < size; k++) {
            perm[R.pos + k] = t.queue[k];
            w.set_region(t.queue[k], -1);
        }
        return;
    }
    // choose the separating level
    int32_t best = -1;
    int64_t best_sz = 0, best_diff = 0;
    bool best_ok = false;
    for (int32_t l = 1; l < ecc; l++) {
        int64_t a = t.lvl_ptr[l], s = t.lvl_ptr[l + 1] - t.lvl_ptr[l], b = size - a - s;
        bool ok = std::min(a, b) * 20 >= (int64_t)size * 7; // both sides >= 35 %
        int64_t diff = a > b ? a - b : b - a;
        bool better;
        if (best < 0) better = true;
        else if (ok != best_ok) better = ok;
        else if (ok) better = (s < best_sz) || (s == best_sz && diff < best_diff);
        else better = (diff < best_diff) || (diff == best_diff && s < best_sz);
        if (better) {
            best = l;
            best_sz = s;
            best_diff = diff;
            best_ok = ok;
        }
    }
    int32_t lb = t.lvl_ptr[best], le = t.lvl_ptr[best + 1];
    // queue layout: [0,lb) = side A, [lb,le) = separator level, [le,size) = side B.
    // thin the separator: a vertex with no neighbour in level best+1 can join side A
    int32_t idA = w.next_id.fetch_add(2, std::memory_order_relaxed), idB = idA + 1;
    int32_t nA = 0, nS = 0;
    for (int32_t k = 0; k < lb; k++) t.tmp[nA++] = t.queue[k];
    int32_t sep_begin = size; // separator collected at the back of tmp (reverse)
    for (int32_t k = lb; k < le; k++) {
        int32_t v = t.queue[k];
        bool up = false;
        for (int64_t p = g.ptr[v]; p < g.ptr[v + 1] && !up; p++) {
            int32_t u = g.adj[p];
            up = (w.region_of(u) == R.id && w.vx[u].lev == best + 1);
        }
        if (up) {
            t.tmp[--sep_begin] = v;
            nS++;
        } else {
            t.tmp[nA++] = v;
        }
    }
    int32_t nB = size - le;
    // tmp: [0,nA) = A ; B goes to [nA, nA+nB) ; separator currently at [sep_begin,size) == [nA+nB,size)
    for (int32_t k = 0; k < nB; k++) t.tmp[nA + k] = t.queue[le + k];
    for (int32_t k = 0; k < nA; k++) w.set_region(t.tmp[k], idA);
    for (int32_t k = nA; k < nA + nB; k++) w.set_region(t.tmp[k], idB);
    // separator numbered last, in ascending BFS order
    for (int32_t k = 0; k < nS; k++) {
        int32_t v = t.tmp[size - 1 - k];
        perm[R.pos + nA + nB + k] = v;
        w.set_region(v, -1);
    }
    std::copy(t.tmp.begin(), t.tmp.begin() + nA + nB, w.verts.begin() + R.begin);
    out.push_back({R.begin + nA, R.begin + nA + nB, R.pos + nA, idB, false, size});
    NDRegion RA = {R.begin, R.begin + nA, R.pos, idA, false, size};
    if (t.tmp[0] == root && lb > 0) {
        // A's own search from `root`: levels 0 .. best - 1 as here, then the thinned separator vertices (level `best`, if any) -- its last
        // level is t.tmp[lb, nA) when that is not empty, else this structure's level best - 1
        const int32_t l0 = nA > lb ? lb : t.lvl_ptr[best - 1], l1 = nA > lb ? nA : lb;
        const int32_t *src = nA > lb ? t.tmp.data() : t.queue.data();
        int32_t cand = src[l0];
        int64_t cdeg = g.ptr[cand + 1] - g.ptr[cand];
        for (int32_t k = l0 + 1; k < l1; k++) {
            const int32_t v = src[k];
            const int64_t d = g.ptr[v + 1] - g.ptr[v];
            if (d < cdeg || (d == cdeg && v < cand)) cand = v, cdeg = d;
        }
        RA.first_ecc = nA > lb ? best : best - 1;
        RA.first_cand = cand;
    }
    out.push_back(RA);
}

static void nested_dissection(const Graph &g, const SymbolicOptions &opt, std::vector<int32_t> &perm, std::vector<int32_t> &leaf_of,
                              std::vector<std::pair<int32_t, int32_t>> *intervals = nullptr) {
    int32_t n = g.n;
    perm.assign((size_t)n, -1);
    leaf_of.assign((size_t)n, -1);
    NDShared w(g);
    if (intervals && n >= opt.parallel_min_n) w.interval_cutoff = std::max<int32_t>(std::min(64, opt.parallel_chunk_min), n / (8 * std::max(1, host_threads(opt))));
    w.vx.reset(new NDVertex[(size_t)n]);
    for (int32_t v = 0; v < n; v++) {
        w.vx[v].part.store(0, std::memory_order_relaxed);
        w.vx[v].stamp = 0, w.vx[v].lev = 0, w.vx[v].pad = 0;
    }
    // Dense rows / columns (hub vertices: the supply nets of circuit matrices, Lagrange multipliers tied to many unknowns) are
    // taken out first and numbered last, as AMD / COLAMD do: left inside, one hub makes every breadth-first level structure three
    // levels deep with its whole neighbourhood as the "separator".  Threshold: degree > max(32, min(10 sqrt(n), 40 x the average
    // degree)) -- AMD's rule, tightened for graphs whose ordinary vertices have few neighbours (meshes: the largest degree of a
    // finite-element or stencil graph stays within a small multiple of the average).
    const double avg_deg = n > 0 ? (double)g.ptr[n] / (double)n : 0.0;
    const int64_t dense_deg =
        std::max<int64_t>(32, (int64_t)std::min(opt.dense_row_factor * std::sqrt((double)n), 4.0 * opt.dense_row_factor * std::max(avg_deg, 1.0)));
    int32_t ndense = 0;
    w.verts.clear();
    w.verts.reserve((size_t)n);
    for (int32_t v = 0; v < n; v++) {
        if (opt.dense_row_factor > 0.0 && g.ptr[v + 1] - g.ptr[v] > dense_deg) ndense++;
        else w.verts.push_back(v);
    }
    if (ndense > 0 && ndense < n) {
        int32_t k = n - ndense;
        for (int32_t v = 0; v < n; v++)
            if (g.ptr[v + 1] - g.ptr[v] > dense_deg) {
                perm[k++] = v; // ascending vertex order
                w.set_region(v, -1);
            }
    } else if (ndense == n) { // (everything is "dense": a small full matrix; nothing to take out)
        w.verts.resize((size_t)n);
        std::iota(w.verts.begin(), w.verts.end(), 0);
        ndense = 0;
    }
    const int32_t nsparse = n - ndense;
    const int32_t leaf = std::min<int32_t>(64, std::max<int32_t>(1, opt.nd_leaf));

    int nthreads = host_threads(opt);
    if (n < 20000) nthreads = 1;
    // shared LIFO of regions; a region below `serial_size` vertices is finished by the thread that took it
    const int32_t serial_size = std::max<int32_t>(2048, n / (64 * nthreads));
    std::mutex mu;
    std::condition_variable cv;
    std::vector<NDRegion> shared;
    int busy = 0;
    shared.push_back({0, nsparse, 0, 0, false, INT32_MAX});
    auto worker = [&]() {
        NDScratch t;
        std::vector<NDRegion> local, out;
        for (;;) {
            NDRegion R;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !shared.empty() || busy == 0; });
                if (shared.empty()) return; // nothing queued and nobody can produce more
                R = shared.back();
                shared.pop_back();
                busy++;
            }
            local.clear();
            local.push_back(R);
            while (!local.empty()) {
                NDRegion Q = local.back();
                local.pop_back();
                out.clear();
                nd_process(w, t, opt, leaf, Q, perm, leaf_of, out);
                bool gave = false;
                for (const NDRegion &c : out) {
                    if (nthreads > 1 && c.end - c.begin >= serial_size && (!local.empty() || &c != &out.back())) {
                        // keep the last (first-to-process) child, hand the other large ones to the pool
                        std::lock_guard<std::mutex> lk(mu);
                        shared.push_back(c);
                        gave = true;
                    } else
                        local.push_back(c);
                }
                if (gave) cv.notify_all();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                busy--;
            }
            cv.notify_all();
        }
    };
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nthreads; i++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    if (intervals) {
        std::sort(w.intervals.begin(), w.intervals.end());
        intervals->swap(w.intervals);
    }
}

// ------------------------------------------------------------------------------------------------
// Approximate minimum degree on the pattern of A + A^T (Ordering::Amd of the reference's enum, enums.rs:71-155).
// The method of Amestoy, Davis & Duff (SIAM J. Matrix Anal. Appl. 17, 1996), written for this code's data structures: the
// elimination graph is kept as a QUOTIENT graph in one workspace -- a variable's list holds the elements (eliminated pivots) it
// touches, then the variables it is still directly adjacent to; an element's list holds its variables -- with
//   * element absorption (an element reachable through the new pivot, or whose variables all lie in the new element, dies),
//   * approximate external degrees  d_i = min(n - k, d_i + |L_me \ i|, |A_i \ i| + |L_me \ i| + sum_e |L_e \ L_me|),
//   * mass elimination (a variable whose only neighbour is the new element is eliminated with it),
//   * indistinguishable variables merged into supervariables (found by hashing the lists of the variables of the new element).
// Ties are broken by list position only (no hashing of pointers, no threads): the permutation is reproducible.
// Hub vertices are taken out first and ordered last, as in the dissection.
// ------------------------------------------------------------------------------------------------
static void approximate_minimum_degree(const Graph &g, const SymbolicOptions &opt, std::vector<int32_t> &perm) {
    const int32_t n = g.n;
    perm.assign((size_t)n, -1);
    const double avg_deg = n > 0 ? (double)g.ptr[n] / (double)n : 0.0;
    const int64_t dense_deg =
        std::max<int64_t>(32, (int64_t)std::min(opt.dense_row_factor * std::sqrt((double)n), 4.0 * opt.dense_row_factor * std::max(avg_deg, 1.0)));
    enum : uint8_t { VAR = 0, ELEM = 1, DEAD = 2, HUB = 3 };
    std::vector<uint8_t> kind((size_t)n, VAR);
    int32_t nhub = 0;
    if (opt.dense_row_factor > 0.0)
        for (int32_t v = 0; v < n; v++)
            if (g.ptr[v + 1] - g.ptr[v] > dense_deg) kind[v] = HUB, nhub++;
    if (nhub == n) std::fill(kind.begin(), kind.end(), (uint8_t)VAR), nhub = 0;
    const int32_t nact = n - nhub;

    // workspace: the lists never grow in total (a variable that gains the new element loses the pivot or an absorbed element; a new
    // element is no longer than the lists it replaces), so the initial size + room for one element built at the end is enough
    std::vector<int64_t> pe((size_t)n, 0);
    std::vector<int32_t> len((size_t)n, 0), elen((size_t)n, 0), nv((size_t)n, 1), degree((size_t)n, 0), parent((size_t)n, -1);
    int64_t nz = 0;
    for (int32_t v = 0; v < n; v++)
        if (kind[v] == VAR)
            for (int64_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) nz += kind[g.adj[p]] == VAR;
    const int64_t iwlen = nz + nz / 5 + 2 * (int64_t)n + 16;
    std::vector<int32_t> iw((size_t)iwlen);
    int64_t pfree = 0;
    for (int32_t v = 0; v < n; v++) {
        if (kind[v] != VAR) continue;
        pe[v] = pfree;
        for (int64_t p = g.ptr[v]; p < g.ptr[v + 1]; p++)
            if (kind[g.adj[p]] == VAR) iw[(size_t)pfree++] = g.adj[p];
        len[v] = (int32_t)(pfree - pe[v]);
        degree[v] = len[v];
    }
    // degree lists
    std::vector<int32_t> head((size_t)n + 1, -1), nxt((size_t)n, -1), prv((size_t)n, -1);
    auto dl_insert = [&](int32_t i, int32_t d) {
        nxt[i] = head[d], prv[i] = -1;
        if (head[d] >= 0) prv[head[d]] = i;
        head[d] = i;
    };
    auto dl_remove = [&](int32_t i, int32_t d) {
        if (prv[i] >= 0) nxt[prv[i]] = nxt[i];
        else head[d] = nxt[i];
        if (nxt[i] >= 0) prv[nxt[i]] = prv[i];
    };
    // (inserted in descending vertex order: the head of a list is its smallest vertex)
    for (int32_t v = n - 1; v >= 0; v--)
        if (kind[v] == VAR) dl_insert(v, degree[v]);
    std::vector<int32_t> inl((size_t)n, -1);            // inl[i] == pivot step: variable i lies in the current element
    std::vector<int32_t> wstep((size_t)n, -1), wext((size_t)n, 0); // element e, at the current step: |L_e \ L_me| (weighted)
    std::vector<int32_t> esize((size_t)n, 0);            // weighted size of an element when it was formed
    std::vector<int32_t> hhead((size_t)n, -1), hnext((size_t)n, -1), cmp((size_t)n, -1);
    std::vector<uint32_t> hval((size_t)n, 0);
    std::vector<int32_t> order;
    order.reserve((size_t)nact);
    std::vector<char> is_pivot((size_t)n, 0);

    auto collect_garbage = [&]() {
        // live lists keep their order in the workspace; the first entry of each is swapped for a tag while the workspace is swept
        for (int32_t i = 0; i < n; i++)
            if ((kind[i] == VAR || kind[i] == ELEM) && len[i] > 0) {
                const int32_t first = iw[(size_t)pe[i]];
                iw[(size_t)pe[i]] = -(i + 1);
                pe[i] = first; // (parked)
            }
        int64_t dst = 0;
        for (int64_t src = 0; src < pfree;) {
            const int32_t tag = iw[(size_t)src];
            if (tag >= 0) {
                src++;
                continue;
            }
            const int32_t i = -tag - 1;
            iw[(size_t)dst] = (int32_t)pe[i];
            pe[i] = dst;
            for (int32_t k = 1; k < len[i]; k++) iw[(size_t)(dst + k)] = iw[(size_t)(src + k)];
            dst += len[i], src += len[i];
        }
        pfree = dst;
    };

    int32_t nel = 0, mindeg = 0;
    for (int32_t step = 0; nel < nact; step++) {
        while (mindeg < n && head[mindeg] < 0) mindeg++;
        const int32_t me = head[mindeg];
        dl_remove(me, mindeg);
        int32_t nvpiv = nv[me];
        nel += nvpiv;
        order.push_back(me);
        is_pivot[me] = 1;
        inl[me] = step; // (keeps the pivot itself out of its element)
        // ---- the new element: variables of the pivot's lists and of the elements it touches --------------------------------------
        bool has_elem = false;
        for (int64_t p = pe[me]; p < pe[me] + elen[me]; p++) has_elem |= kind[iw[(size_t)p]] == ELEM;
        int64_t pme1, pme2;
        int32_t degme = 0;
        auto take = [&](int32_t i, int64_t &wr) {
            if (kind[i] != VAR || inl[i] == step) return;
            inl[i] = step;
            iw[(size_t)wr++] = i;
            degme += nv[i];
            dl_remove(i, degree[i]);
        };
        if (!has_elem) {
            // in place: the pivot's own variable list, filtered
            pme1 = pe[me];
            int64_t wr = pme1;
            for (int64_t p = pe[me] + elen[me]; p < pe[me] + len[me]; p++) take(iw[(size_t)p], wr);
            pme2 = wr;
        } else {
            if (pfree + (int64_t)(nact - nel) + 1 > iwlen) collect_garbage();
            pme1 = pfree;
            int64_t wr = pfree;
            const int64_t pb = pe[me]; // (re-read after a possible compaction)
            for (int64_t p = pb; p < pb + elen[me]; p++) {
                const int32_t e = iw[(size_t)p];
                if (kind[e] != ELEM) continue;
                for (int64_t q = pe[e]; q < pe[e] + len[e]; q++) take(iw[(size_t)q], wr);
                kind[e] = DEAD, parent[e] = me; // absorbed
            }
            for (int64_t p = pb + elen[me]; p < pb + len[me]; p++) take(iw[(size_t)p], wr);
            pme2 = wr;
            pfree = wr;
        }
        kind[me] = ELEM;
        pe[me] = pme1, len[me] = (int32_t)(pme2 - pme1), elen[me] = 0;
        // ---- |L_e \ L_me| for every element a variable of L_me touches -------------------------------------------------------------
        for (int64_t pm = pme1; pm < pme2; pm++) {
            const int32_t i = iw[(size_t)pm];
            for (int64_t p = pe[i]; p < pe[i] + elen[i]; p++) {
                const int32_t e = iw[(size_t)p];
                if (kind[e] != ELEM) continue;
                if (wstep[e] != step) wstep[e] = step, wext[e] = esize[e];
                wext[e] -= nv[i];
            }
        }
        // ---- degrees, list clean-up, mass elimination, hash ------------------------------------------------------------------------
        for (int64_t pm = pme1; pm < pme2; pm++) {
            const int32_t i = iw[(size_t)pm];
            const int64_t p1 = pe[i];
            int64_t pn = p1;
            int64_t deg = 0;
            uint32_t h = 0;
            for (int64_t p = p1; p < p1 + elen[i]; p++) {
                const int32_t e = iw[(size_t)p];
                if (kind[e] != ELEM) continue;
                const int32_t ext = std::max(wext[e], 0);
                if (ext > 0) {
                    deg += ext;
                    iw[(size_t)pn++] = e;
                    h += (uint32_t)e;
                } else {
                    kind[e] = DEAD, parent[e] = me; // every variable of e lies in the new element: aggressive absorption
                }
            }
            const int64_t p3 = pn;
            for (int64_t p = p1 + elen[i]; p < p1 + len[i]; p++) {
                const int32_t j = iw[(size_t)p];
                if (kind[j] != VAR || inl[j] == step) continue; // (the pivot, dead variables, variables the new element covers)
                deg += nv[j];
                iw[(size_t)pn++] = j;
                h += (uint32_t)j;
            }
            if (deg == 0) {
                // nothing outside the new element: eliminated together with the pivot
                kind[i] = DEAD, parent[i] = me;
                nvpiv += nv[i], degme -= nv[i], nel += nv[i];
                len[i] = 0;
                continue;
            }
            degree[i] = (int32_t)std::min<int64_t>(degree[i], deg);
            // the new element leads the list (one entry at least was dropped above: the pivot or an absorbed element)
            iw[(size_t)pn] = iw[(size_t)p3];
            iw[(size_t)p3] = iw[(size_t)p1];
            iw[(size_t)p1] = me;
            elen[i] = (int32_t)(p3 - p1) + 1;
            len[i] = (int32_t)(pn - p1) + 1;
            h += (uint32_t)me;
            hval[i] = h;
            const int32_t b = (int32_t)(h % (uint32_t)n);
            hnext[i] = hhead[b], hhead[b] = i;
        }
        // ---- supervariables: variables of the new element with identical lists ------------------------------------------------------
        for (int64_t pm = pme1; pm < pme2; pm++) {
            const int32_t i0 = iw[(size_t)pm];
            if (kind[i0] != VAR) continue;
            const int32_t b = (int32_t)(hval[i0] % (uint32_t)n);
            int32_t i = hhead[b];
            hhead[b] = -1; // (the bucket is worked off once)
            for (; i >= 0 && hnext[i] >= 0; i = hnext[i]) {
                if (kind[i] != VAR) continue;
                for (int64_t p = pe[i] + 1; p < pe[i] + len[i]; p++) cmp[iw[(size_t)p]] = i;
                int32_t jprev = i;
                for (int32_t j = hnext[i]; j >= 0;) {
                    bool same = kind[j] == VAR && hval[j] == hval[i] && len[j] == len[i] && elen[j] == elen[i];
                    for (int64_t p = pe[j] + 1; same && p < pe[j] + len[j]; p++) same = cmp[iw[(size_t)p]] == i;
                    if (same) {
                        nv[i] += nv[j];
                        kind[j] = DEAD, parent[j] = i, len[j] = 0;
                        j = hnext[j];
                        hnext[jprev] = j;
                    } else {
                        jprev = j;
                        j = hnext[j];
                    }
                }
            }
        }
        // ---- the element's final list, the degrees of its variables ----------------------------------------------------------------
        int64_t wr = pme1;
        for (int64_t pm = pme1; pm < pme2; pm++) {
            const int32_t i = iw[(size_t)pm];
            if (kind[i] != VAR) continue;
            int64_t d = (int64_t)degree[i] + degme - nv[i];
            d = std::max<int64_t>(0, std::min<int64_t>(d, (int64_t)nact - nel - nv[i]));
            degree[i] = (int32_t)d;
            dl_insert(i, degree[i]);
            if (degree[i] < mindeg) mindeg = degree[i];
            iw[(size_t)wr++] = i;
        }
        nv[me] = nvpiv;
        esize[me] = degme;
        len[me] = (int32_t)(wr - pme1);
        if (len[me] == 0) kind[me] = DEAD; // (a root of the assembly forest: no variable refers to it)
        if (has_elem) pfree = wr;
    }
    // ---- permutation: pivots in elimination order, each followed by the variables eliminated with it (ascending) --------------------
    std::vector<int32_t> owner((size_t)n, -1), cnt((size_t)n + 1, 0);
    for (int32_t v = 0; v < n; v++) {
        if (kind[v] == HUB || is_pivot[v]) continue;
        int32_t r = v;
        while (!is_pivot[r]) r = parent[r];
        owner[v] = r;
        cnt[r]++;
    }
    int32_t k = 0;
    std::vector<int32_t> start((size_t)n, 0);
    for (int32_t me : order) {
        perm[(size_t)k++] = me;
        start[me] = k;
        k += cnt[me];
    }
    for (int32_t v = 0; v < n; v++)
        if (owner[v] >= 0) perm[(size_t)start[owner[v]]++] = v;
    for (int32_t v = 0; v < n; v++)
        if (kind[v] == HUB) perm[(size_t)k++] = v;
}

// elimination tree of the permuted symmetric pattern (Liu's algorithm with path compression)
// `closed`: disjoint intervals (first column, size), ascending, whose columns have no neighbour with a smaller number outside the
// interval (regions of the dissection): their parts of the tree are built on host threads, the remaining columns (separators above
// them, hubs) afterwards in ascending order.  The tree does not depend on the split -- it is a function of the pattern.
static void etree(const Graph &gp, std::vector<int32_t> &parent, const std::vector<std::pair<int32_t, int32_t>> *closed = nullptr, int threads = 1) {
    int32_t n = gp.n;
    parent.assign((size_t)n, -1);
    std::vector<int32_t> anc((size_t)n, -1);
    int32_t *par = parent.data(), *an = anc.data();
    auto columns = [&](int32_t j0, int32_t j1) {
        for (int32_t j = j0; j < j1; j++)
            for (int64_t p = gp.ptr[j]; p < gp.ptr[j + 1]; p++) {
                int32_t r = gp.adj[p];
                if (r >= j) break; // ascending lists
                while (an[r] != -1 && an[r] != j) {
                    int32_t next = an[r];
                    an[r] = j;
                    r = next;
                }
                if (an[r] == -1) {
                    an[r] = j;
                    par[r] = j;
                }
            }
    };
    if (!closed || closed->empty() || threads <= 1) {
        columns(0, n);
        return;
    }
    {
        std::atomic<size_t> next{0};
        auto body = [&]() {
            for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < closed->size();) columns((*closed)[i].first, (*closed)[i].first + (*closed)[i].second);
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(body);
        for (auto &th : pool) th.join();
    }
    int32_t j = 0;
    for (const auto &iv : *closed) {
        columns(j, iv.first);
        j = iv.first + iv.second;
    }
    columns(j, n);
}

static void postorder(const std::vector<int32_t> &parent, std::vector<int32_t> &post) {
    int32_t n = (int32_t)parent.size();
    std::vector<int32_t> head((size_t)n, -1), next((size_t)n, -1), stack;
    for (int32_t j = n - 1; j >= 0; j--)
        if (parent[j] >= 0) {
            next[j] = head[parent[j]];
            head[parent[j]] = j;
        }
    post.clear();
    post.reserve((size_t)n);
    for (int32_t r = 0; r < n; r++) {
        if (parent[r] != -1) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            int32_t v = stack.back();
            int32_t c = head[v];
            if (c == -1) {
                post.push_back(v);
                stack.pop_back();
            } else {
                head[v] = next[c];
                stack.push_back(c);
            }
        }
    }
}

// column counts of the Cholesky factor of the (postordered) symmetric pattern
// (skeleton / least-common-ancestor method of Gilbert, Ng & Peyton, 1994)
//
// threads > 1: in the postorder a subtree is a contiguous range of columns [first[r], r], and everything the method does for a column of
// the subtree stays inside it -- its own counts, the leaves of the row subtrees of rows INSIDE the range, the disjoint-set links -- except
// for the rows above the subtree's root: there the first leaf the subtree contributes needs the row's previous leaf (in an earlier
// subtree) and the last one must be left behind for later columns.  So the maximal subtrees below a size bound are processed on host
// threads with private state for those rows, and each hands back, per row above it, (first leaf, last leaf, its `first`); the columns
// above the subtrees are then processed in order, a subtree's records applied where its columns stand.  Same counts as the serial sweep.
static void column_counts(const Graph &gp, const std::vector<int32_t> &parent, std::vector<int64_t> &cc, int threads = 1, int32_t parallel_min_n = 0,
                          int32_t chunk_min = 4096) {
    int32_t n = gp.n;
    std::vector<int32_t> first((size_t)n, -1), maxfirst((size_t)n, -1), prevleaf((size_t)n, -1), anc((size_t)n);
    std::iota(anc.begin(), anc.end(), 0);
    cc.assign((size_t)n, 0);
    for (int32_t k = 0; k < n; k++) {
        int32_t j = k;
        cc[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    auto find = [&](int32_t x) {
        int32_t q = x;
        while (q != anc[q]) q = anc[q];
        for (int32_t s = x; s != q;) {
            int32_t sp = anc[s];
            anc[s] = q;
            s = sp;
        }
        return q;
    };
    // one column with the shared (serial) state
    auto column = [&](int32_t j) {
        if (parent[j] != -1) cc[parent[j]]--;
        for (int64_t p = gp.ptr[j]; p < gp.ptr[j + 1]; p++) {
            int32_t i = gp.adj[p];
            if (i <= j || first[j] <= maxfirst[i]) continue;
            maxfirst[i] = first[j];
            int32_t jprev = prevleaf[i];
            prevleaf[i] = j;
            cc[j]++;
            if (jprev != -1) cc[find(jprev)]--;
        }
        if (parent[j] != -1) anc[j] = parent[j];
    };
    struct Rec {
        int32_t row, first_leaf, last_leaf, last_first;
    };
    struct Chunk {
        int32_t lo, hi;
        std::vector<Rec> recs;
    };
    std::vector<Chunk> chunks;
    if (threads > 1 && n >= parallel_min_n && n > 1) {
        const int64_t bound = std::max<int64_t>(chunk_min, (int64_t)n / (8 * threads));
        auto weight = [&](int32_t j) { return (int64_t)j - first[j] + 1; };
        for (int32_t j = 0; j < n; j++)
            if (weight(j) <= bound && (parent[j] < 0 || weight(parent[j]) > bound)) chunks.push_back({first[j], j, {}});
    }
    if (chunks.empty()) {
        for (int32_t j = 0; j < n; j++) column(j);
    } else {
        std::atomic<size_t> next{0};
        std::atomic<bool> oom{false};
        auto body = [&]() {
            try {
                struct Out {
                    int32_t maxfirst, prevleaf, first_leaf;
                };
                std::vector<int32_t> mf_in, pl_in; // rows inside the chunk, by (row - lo)
                std::vector<std::pair<int32_t, Out>> outs;  // rows above the chunk's root: few (open-addressing table below)
                std::vector<int32_t> slot;
                for (size_t ci; (ci = next.fetch_add(1, std::memory_order_relaxed)) < chunks.size();) {
                    Chunk &C = chunks[ci];
                    const int32_t lo = C.lo, hi = C.hi, w = hi - lo + 1;
                    mf_in.assign((size_t)w, -1), pl_in.assign((size_t)w, -1);
                    outs.clear();
                    size_t cap = 64;
                    slot.assign(cap, -1);
                    auto out_of = [&](int32_t row) -> Out & { // find or insert
                        for (;;) {
                            size_t h = ((size_t)(uint32_t)row * 2654435761u) & (cap - 1);
                            while (slot[h] >= 0 && outs[(size_t)slot[h]].first != row) h = (h + 1) & (cap - 1);
                            if (slot[h] >= 0) return outs[(size_t)slot[h]].second;
                            if (2 * (outs.size() + 1) > cap) { // grow and rehash
                                cap *= 2;
                                slot.assign(cap, -1);
                                for (size_t q = 0; q < outs.size(); q++) {
                                    size_t g = ((size_t)(uint32_t)outs[q].first * 2654435761u) & (cap - 1);
                                    while (slot[g] >= 0) g = (g + 1) & (cap - 1);
                                    slot[g] = (int32_t)q;
                                }
                                continue;
                            }
                            slot[h] = (int32_t)outs.size();
                            outs.push_back({row, Out{-1, -1, -1}});
                            return outs.back().second;
                        }
                    };
                    for (int32_t j = lo; j <= hi; j++) {
                        if (j != hi && parent[j] != -1) cc[parent[j]]--; // (the root's parent lies outside: left to the serial part)
                        for (int64_t p = gp.ptr[j]; p < gp.ptr[j + 1]; p++) {
                            const int32_t i = gp.adj[p];
                            if (i <= j) continue;
                            int32_t jprev;
                            if (i <= hi) {
                                if (first[j] <= mf_in[(size_t)(i - lo)]) continue;
                                mf_in[(size_t)(i - lo)] = first[j];
                                jprev = pl_in[(size_t)(i - lo)];
                                pl_in[(size_t)(i - lo)] = j;
                            } else {
                                Out &o = out_of(i);
                                if (first[j] <= o.maxfirst) continue;
                                o.maxfirst = first[j];
                                jprev = o.prevleaf;
                                o.prevleaf = j;
                                if (jprev == -1) o.first_leaf = j; // its least common ancestor with the row's earlier leaf: serial part
                            }
                            cc[j]++;
                            if (jprev != -1) cc[find(jprev)]--; // (both leaves inside the chunk: so is their least common ancestor)
                        }
                        if (parent[j] != -1) anc[j] = parent[j];
                    }
                    C.recs.reserve(outs.size());
                    for (const auto &o : outs) C.recs.push_back({o.first, o.second.first_leaf, o.second.prevleaf, o.second.maxfirst});
                }
            } catch (const std::bad_alloc &) { // (an exception must not leave a thread)
                oom.store(true);
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(body);
        for (auto &th : pool) th.join();
        if (oom.load()) throw std::bad_alloc();
        size_t c = 0;
        for (int32_t j = 0; j < n; j++) {
            if (c < chunks.size() && chunks[c].lo == j) {
                const Chunk &C = chunks[c++];
                for (const Rec &r : C.recs) {
                    const int32_t jprev = prevleaf[r.row];
                    if (jprev != -1) cc[find(jprev)]--;
                    prevleaf[r.row] = r.last_leaf, maxfirst[r.row] = r.last_first;
                }
                if (parent[C.hi] != -1) cc[parent[C.hi]]--;
                j = C.hi;
                continue;
            }
            column(j);
        }
    }
    for (int32_t j = 0; j < n; j++)
        if (parent[j] != -1) cc[parent[j]] += cc[j];
}

} // namespace

int validate_csr(int32_t n, const int32_t *rp, const int32_t *ci) {
    if (n < 1 || !rp || !ci || rp[0] != 0) return -1;
    for (int32_t i = 0; i < n; i++) {
        if (rp[i + 1] < rp[i]) return -1;
        int32_t prev = -1;
        for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
            const int32_t j = ci[p];
            if (j < 0 || j >= n) return -2;
            if (j <= prev) return -3;
            prev = j;
        }
    }
    return 0;
}


int analyse(int32_t n, const int32_t *rp, const int32_t *ci, bool sym_lower, const SymbolicOptions &opt, Symbolic &S) {
    auto t_all = clk::now(), t_phase = t_all;
    if (n < 1 || !rp || !ci) return -1;
    S = Symbolic();
    S.n = n;
    S.nnz_a = rp[n];
    S.sym_lower = sym_lower;

    Graph g;
    const int threads = host_threads(opt);
    int rc;
    const bool pairs = opt.pair_blocks;
    if (pairs) {
        // the structure of the pair (2 k, 2 k + 1) is made explicit: both rows hold both columns of every pair either of them touches, the
        // pair's own 2 x 2 block included (whatever the caller stored; entries that are not in A are structural zeros of the fronts).  Then
        // 2 k + 1 is the parent of 2 k in the elimination tree and its only child, the two columns have the same structure below them, and
        // the postorder keeps them next to each other: one fundamental supernode holds both.
        if (n % 2 != 0 || sym_lower) return -1;
        std::vector<int32_t> rp2((size_t)n + 1, 0), ci2, mark((size_t)n / 2, -1);
        ci2.reserve((size_t)rp[n] * 2 + (size_t)n * 2);
        for (int32_t k = 0; k < n / 2; k++) {
            const size_t b = ci2.size();
            mark[k] = k, ci2.push_back(k);
            for (int32_t i = 2 * k; i < 2 * k + 2; i++) {
                if (rp[i + 1] < rp[i]) return -1;
                for (int32_t q = rp[i]; q < rp[i + 1]; q++) {
                    const int32_t j = ci[q];
                    if (j < 0 || j >= n) return -2;
                    if (mark[j / 2] != k) mark[j / 2] = k, ci2.push_back(j / 2);
                }
            }
            std::sort(ci2.begin() + b, ci2.end());
            const size_t cnt = ci2.size() - b;
            // (pair columns -> both columns, twice: rows 2 k and 2 k + 1)
            ci2.resize(b + 4 * cnt);
            for (size_t e = cnt; e-- > 0;) {
                const int32_t c = ci2[b + e];
                ci2[b + 2 * e] = 2 * c, ci2[b + 2 * e + 1] = 2 * c + 1;
            }
            std::copy(ci2.begin() + b, ci2.begin() + b + 2 * cnt, ci2.begin() + b + 2 * cnt);
            rp2[2 * k + 1] = (int32_t)(b + 2 * cnt), rp2[2 * k + 2] = (int32_t)(b + 4 * cnt);
            if (ci2.size() > 0x7fffffffULL) return -1;
        }
        rc = build_graph(n, rp2.data(), ci2.data(), g, threads);
    } else
        rc = build_graph(n, rp, ci, g, threads);
    if (rc != 0) return rc;

    S.seconds_phase[0] = since(t_phase), t_phase = clk::now();
    // ---- ordering --------------------------------------------------------------------------
    auto t_ord = clk::now();
    std::vector<int32_t> perm0((size_t)n), pinv0((size_t)n), leaf_of;
    std::vector<std::pair<int32_t, int32_t>> closed; // intervals of the dissection's numbering that the elimination tree can take apart
    bool single_front = n <= opt.dense_n;
    if (single_front || opt.ordering == ORDERING_NATURAL) {
        std::iota(perm0.begin(), perm0.end(), 0);
    } else if (pairs) {
        // the graph of the pairs is ordered, every pair takes two consecutive places
        Graph gc;
        gc.n = n / 2;
        gc.ptr.assign((size_t)n / 2 + 1, 0);
        for (int32_t k = 0; k < n / 2; k++) gc.ptr[(size_t)k + 1] = gc.ptr[k] + (g.ptr[2 * k + 1] - g.ptr[2 * k] - 1) / 2;
        gc.adj.resize((size_t)gc.ptr[n / 2]);
        for (int32_t k = 0; k < n / 2; k++) {
            int64_t w = gc.ptr[k];
            for (int64_t q = g.ptr[2 * k]; q < g.ptr[2 * k + 1]; q++)
                if ((g.adj[q] & 1) == 0 && g.adj[q] != 2 * k) gc.adj[(size_t)w++] = g.adj[q] / 2; // (ascending, one per pair: 2 k's list holds 2 k + 1 and both columns of every other pair)
            if (w != gc.ptr[(size_t)k + 1]) return -11;
        }
        std::vector<int32_t> permc, leafc;
        if (opt.ordering == ORDERING_MIN_DEGREE) approximate_minimum_degree(gc, opt, permc);
        else nested_dissection(gc, opt, permc, leafc, &closed);
        for (auto &iv : closed) iv.first *= 2, iv.second *= 2; // (pairs: two columns per vertex of the ordered graph)
        for (int32_t k = 0; k < n / 2; k++) perm0[2 * k] = 2 * permc[k], perm0[2 * k + 1] = 2 * permc[k] + 1;
        if (!leafc.empty()) {
            leaf_of.resize((size_t)n);
            for (int32_t k = 0; k < n / 2; k++) leaf_of[2 * k] = leaf_of[2 * k + 1] = leafc[k];
        }
    } else if (opt.ordering == ORDERING_MIN_DEGREE) {
        approximate_minimum_degree(g, opt, perm0);
    } else if (opt.ordering == ORDERING_BEST) {
        // Ordering::Best (UMFPACK's meaning, umfpack_ordering in solver_umfpack.rs:457-472: try several, keep the sparsest): both orderings,
        // the one whose factorisation needs fewer flops by the column counts wins (the minimum degree runs on a thread of its own beside
        // the dissection; the winner's elimination tree is built once more below)
        std::vector<int32_t> perm_md;
        std::atomic<bool> md_oom{false};
        std::thread md([&]() {
            try {
                approximate_minimum_degree(g, opt, perm_md);
            } catch (const std::bad_alloc &) {
                md_oom.store(true);
            }
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() {
                if (t.joinable()) t.join();
            }
        } joiner{md};
        nested_dissection(g, opt, perm0, leaf_of);
        md.join();
        if (md_oom.load()) return -41;
        auto flops_of = [&](const std::vector<int32_t> &pm) -> double {
            std::vector<int32_t> pi((size_t)n);
            for (int32_t k = 0; k < n; k++) {
                if (pm[k] < 0 || pm[k] >= n) return -1.0;
                pi[pm[k]] = k;
            }
            Graph q;
            permute_graph(g, pm, pi, q, threads);
            std::vector<int32_t> par, po;
            etree(q, par);
            postorder(par, po);
            std::vector<int32_t> pm2((size_t)n), pi2((size_t)n), poi((size_t)n), par2((size_t)n);
            for (int32_t k = 0; k < n; k++) pm2[k] = pm[po[k]], poi[po[k]] = k;
            for (int32_t k = 0; k < n; k++) pi2[pm2[k]] = k, par2[k] = par[po[k]] < 0 ? -1 : poi[par[po[k]]];
            permute_graph(g, pm2, pi2, q, threads);
            std::vector<int64_t> c;
            column_counts(q, par2, c);
            double fl = 0.0;
            for (int32_t k = 0; k < n; k++) fl += (double)c[k] * (double)c[k];
            return fl;
        };
        const double f_nd = flops_of(perm0), f_md = flops_of(perm_md);
        if (f_nd < 0.0 || f_md < 0.0) return -10;
        S.best_chose_min_degree = f_md < f_nd;
        if (S.best_chose_min_degree) perm0.swap(perm_md), leaf_of.clear();
    } else {
        nested_dissection(g, opt, perm0, leaf_of, &closed);
    }
    for (int32_t k = 0; k < n; k++) {
        if (perm0[k] < 0 || perm0[k] >= n) return -10;
        pinv0[perm0[k]] = k;
    }
    S.seconds_ordering = since(t_ord);

    S.seconds_phase[1] = since(t_phase), t_phase = clk::now();
    // ---- etree, postorder, final permutation -------------------------------------------------
    Graph gp;
    permute_graph(g, perm0, pinv0, gp, threads);
    std::vector<int32_t> parent0, post;
    etree(gp, parent0, &closed, threads);
    postorder(parent0, post);
    S.perm.resize((size_t)n);
    S.pinv.resize((size_t)n);
    for (int32_t k = 0; k < n; k++) S.perm[k] = perm0[post[k]];
    for (int32_t k = 0; k < n; k++) S.pinv[S.perm[k]] = k;
    std::vector<int32_t> postinv((size_t)n), parent((size_t)n);
    for (int32_t k = 0; k < n; k++) postinv[post[k]] = k;
    for (int32_t k = 0; k < n; k++) parent[k] = parent0[post[k]] < 0 ? -1 : postinv[parent0[post[k]]];
    permute_graph(g, S.perm, S.pinv, gp, threads);
    { Graph().ptr.swap(g.ptr); std::vector<int32_t>().swap(g.adj); }

    S.seconds_phase[2] = since(t_phase), t_phase = clk::now();
    // ---- column counts and supernodes --------------------------------------------------------
    std::vector<int64_t> cc;
    try {
        column_counts(gp, parent, cc, threads, opt.parallel_min_n, opt.parallel_chunk_min);
    } catch (const std::bad_alloc &) {
        return -41;
    }

    std::vector<int32_t> fs_first; // fundamental supernodes
    if (single_front) {
        fs_first.push_back(0);
    } else {
        std::vector<int32_t> nchild((size_t)n, 0);
        for (int32_t j = 0; j < n; j++)
            if (parent[j] >= 0) nchild[parent[j]]++;
        // columns of one nested-dissection leaf region form ONE dense supernode when opt.dense_leaves is set:
        // fewer, fatter fronts (the solves and the factorisation are latency-bound, not bandwidth-bound)
        std::vector<int32_t> lid((size_t)n, -1);
        if (!leaf_of.empty())
            for (int32_t j = 0; j < n; j++) lid[j] = leaf_of[S.perm[j]];
        fs_first.push_back(0);
        for (int32_t j = 1; j < n; j++) {
            bool same;
            if (lid[j] >= 0 || lid[j - 1] >= 0) same = lid[j] == lid[j - 1] && parent[j - 1] >= 0;
            else same = parent[j - 1] == j && cc[j - 1] == cc[j] + 1 && nchild[j] == 1;
            if (!same) fs_first.push_back(j);
        }
    }
    int32_t nfs = (int32_t)fs_first.size();
    fs_first.push_back(n);
    std::vector<int32_t> s_first((size_t)nfs), s_last((size_t)nfs), s_ncol((size_t)nfs);
    std::vector<int64_t> s_m((size_t)nfs), s_nz((size_t)nfs);
    std::vector<char> alive((size_t)nfs, 1);
    std::vector<int32_t> col2fs((size_t)n);
    for (int32_t s = 0; s < nfs; s++) {
        s_first[s] = fs_first[s];
        s_last[s] = fs_first[s + 1] - 1;
        s_ncol[s] = fs_first[s + 1] - fs_first[s];
        s_m[s] = single_front ? 0 : cc[s_last[s]] - 1;
        int64_t nz = 0;
        for (int32_t j = s_first[s]; j <= s_last[s]; j++) {
            col2fs[j] = s;
            nz += single_front ? (int64_t)(n - j) : cc[j];
        }
        s_nz[s] = nz;
    }
    // relaxed amalgamation: a supernode may absorb the child whose columns end right before its own
    for (int32_t s = 0; s < nfs && !single_front; s++) {
        int32_t pj = parent[s_last[s]];
        if (pj < 0) continue;
        int32_t t = col2fs[pj];
        if (s_last[s] + 1 != s_first[t]) continue;
        int64_t nc = (int64_t)s_ncol[s] + s_ncol[t];
        int64_t size = nc * (nc + 1) / 2 + nc * s_m[t];
        int64_t tru = s_nz[s] + s_nz[t];
        double z = size > 0 ? (double)(size - tru) / (double)size : 0.0;
        bool accept;
        if (nc <= opt.relax_ncol[0]) accept = true;
        else if (nc <= opt.relax_ncol[1]) accept = z < opt.relax_zeros[0];
        else if (nc <= opt.relax_ncol[2]) accept = z < opt.relax_zeros[1];
        else accept = z < opt.relax_zeros[2] && (opt.relax_big_front <= 0 || nc + s_m[t] >= opt.relax_big_front);
        if (accept) {
            s_first[t] = s_first[s];
            s_ncol[t] = (int32_t)nc;
            s_nz[t] = tru;
            alive[s] = 0;
        }
    }
    S.sn_first.clear();
    for (int32_t s = 0; s < nfs; s++)
        if (alive[s]) S.sn_first.push_back(s_first[s]);
    std::sort(S.sn_first.begin(), S.sn_first.end());
    // A supernode with more than opt.split_pivots pivots becomes a CHAIN of supernodes of (nearly) equal width, a multiple
    // of 32 columns each: link k owns the next block of pivot columns and has the rest of the original front as its rows, so
    // its contribution block is the next link's whole front.  The arithmetic is that of a blocked right-looking LU of the
    // original front; what changes is the cost of the augmentation (section "front pool layout" below): an augmented front
    // of p pivots and f rows does 2 p f^2 flops instead of 2/3 p^3 + 2 p^2 m + 2 p m^2, i.e. three times the work for a root
    // separator -- with links of b pivots the overhead drops to about 3 b / (2 p).  (3D problems: the top separators hold
    // most of the flops.)  Parents, children and row structures below follow from the column elimination tree as for any
    // other supernode partition.
    if (opt.split_pivots > 0 && !single_front) {
        const int32_t nsn = (int32_t)S.sn_first.size();
        std::vector<int32_t> extra;
        for (int32_t s = 0; s < nsn; s++) {
            const int32_t first = S.sn_first[s], next = s + 1 < nsn ? S.sn_first[s + 1] : n;
            const int32_t np = next - first;
            if (np <= opt.split_pivots) continue;
            const int32_t nlinks = (np + opt.split_pivots - 1) / opt.split_pivots;
            const int32_t width = ((np + nlinks - 1) / nlinks + 31) / 32 * 32;
            for (int32_t c = first + width; c < next; c += width) extra.push_back(c);
        }
        S.sn_first.insert(S.sn_first.end(), extra.begin(), extra.end());
        std::sort(S.sn_first.begin(), S.sn_first.end());
    }
    S.nsuper = (int32_t)S.sn_first.size();
    S.sn_first.push_back(n);
    // Memory guard BEFORE the row structures are built: the column counts already tell how large the fronts will be.  A graph
    // without small separators (random sparse matrices) would otherwise ask the host for hundreds of GB of row indices and the
    // device for a pool it does not have; the caller gets the out-of-memory status instead (analyse returns -40).
    if (!single_front) {
        double rows_total = 0.0, pool_total = 0.0;
        for (int32_t s = 0; s < S.nsuper; s++) {
            const double p = (double)(S.sn_first[s + 1] - S.sn_first[s]);
            const double m = std::max((double)cc[S.sn_first[s + 1] - 1] - 1.0, (double)cc[S.sn_first[s]] - p);
            const double f = p + m;
            rows_total += m;
            // persistent part only (a lower bound: the arena of the working blocks is planned once the levels are known)
            pool_total += f > (double)opt.augment_above ? f * p * (opt.symmetric_ldlt ? 1.0 : 2.0) : f * f + p * f;
        }
        S.pool_estimate_bytes = 8.0 * pool_total;
        const double limit = opt.pool_limit_live ? opt.pool_limit_live->load() : opt.pool_limit_bytes;
        if (rows_total > 1.0e9 || (limit > 0.0 && 8.0 * pool_total > limit)) return -40;
    }
    S.sn_of.resize((size_t)n);
    for (int32_t s = 0; s < S.nsuper; s++)
        for (int32_t j = S.sn_first[s]; j < S.sn_first[s + 1]; j++) S.sn_of[j] = s;
    S.sn_parent.assign((size_t)S.nsuper, -1);
    for (int32_t s = 0; s < S.nsuper; s++) {
        int32_t pj = parent[S.sn_first[s + 1] - 1];
        S.sn_parent[s] = pj < 0 ? -1 : S.sn_of[pj];
    }
    // children lists (ascending)
    S.child_ptr.assign((size_t)S.nsuper + 1, 0);
    for (int32_t s = 0; s < S.nsuper; s++)
        if (S.sn_parent[s] >= 0) S.child_ptr[S.sn_parent[s] + 1]++;
    for (int32_t s = 0; s < S.nsuper; s++) S.child_ptr[s + 1] += S.child_ptr[s];
    S.child_idx.resize((size_t)S.child_ptr[S.nsuper]);
    {
        std::vector<int32_t> w(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int32_t s = 0; s < S.nsuper; s++)
            if (S.sn_parent[s] >= 0) S.child_idx[w[S.sn_parent[s]]++] = s;
    }

    S.seconds_phase[3] = since(t_phase), t_phase = clk::now();
    // ---- row structure of every supernode ----------------------------------------------------
    // A supernode's rows are the entries of A below its pivots merged with its children's rows: a subtree needs nothing from outside,
    // and in the postorder it is a contiguous range of supernodes.  The maximal subtrees below a size bound ("chunks") are built on host
    // threads into buffers of their own, the few supernodes above them afterwards, and the pieces are copied to their places once all sizes
    // are known (the lists themselves are what the serial sweep produced: sorted unions).
    S.sn_rowptr.assign((size_t)S.nsuper + 1, 0);
    S.sn_rows.clear();
    {
        struct Chunk {
            int32_t lo, hi; // supernodes [lo, hi]
            std::vector<int32_t> rows;
            std::vector<int64_t> ptr; // hi - lo + 2 offsets into rows
        };
        std::vector<Chunk> chunks;
        std::vector<int32_t> chunk_of((size_t)S.nsuper, -1);
        if (threads > 1 && n >= opt.parallel_min_n) {
            std::vector<int32_t> desc_first((size_t)S.nsuper);
            std::iota(desc_first.begin(), desc_first.end(), 0);
            for (int32_t s = 0; s < S.nsuper; s++)
                if (S.sn_parent[s] >= 0) desc_first[S.sn_parent[s]] = std::min(desc_first[S.sn_parent[s]], desc_first[s]);
            const int64_t bound = std::max<int64_t>(opt.parallel_chunk_min, (int64_t)n / (8 * threads)); // columns of a chunk
            auto weight = [&](int32_t s) { return (int64_t)S.sn_first[s + 1] - S.sn_first[desc_first[s]]; };
            for (int32_t s = 0; s < S.nsuper; s++)
                if (weight(s) <= bound && (S.sn_parent[s] < 0 || weight(S.sn_parent[s]) > bound)) {
                    for (int32_t k = desc_first[s]; k <= s; k++) chunk_of[k] = (int32_t)chunks.size();
                    chunks.push_back({desc_first[s], s, {}, {}});
                }
        }
        struct Scratch {
            std::vector<int32_t> mark, rows, merged;
        };
        // rows of supernode s into t.rows; child(ch) gives the [begin, end) of a child's finished list
        auto sn_struct = [&](int32_t s, Scratch &t, auto &&child) {
            if (t.mark.empty()) t.mark.assign((size_t)n, -1);
            std::vector<int32_t> &mark = t.mark, &rows = t.rows, &merged = t.merged;
            const int32_t last = S.sn_first[s + 1] - 1;
            rows.clear();
            for (int32_t j = S.sn_first[s]; j <= last; j++)
                for (int64_t p = gp.ptr[j + 1] - 1; p >= gp.ptr[j]; p--) {
                    int32_t i = gp.adj[p];
                    if (i <= last) break;
                    if (mark[i] != s) {
                        mark[i] = s;
                        rows.push_back(i);
                    }
                }
            const int32_t nchild = S.child_ptr[s + 1] - S.child_ptr[s];
            if (nchild >= 1 && nchild <= 8) {
                // Few children (the separators of a dissection have two): their row lists are sorted, so the union is a merge --
                // the entries of A first (few, sorted here), then child after child; no sort of the whole list (the sorts of the
                // large supernodes were most of this phase: 0.8 s at 200^3).
                std::sort(rows.begin(), rows.end());
                for (int32_t c = S.child_ptr[s]; c < S.child_ptr[s + 1]; c++) {
                    const std::pair<const int32_t *, const int32_t *> cr = child(S.child_idx[c]);
                    const int32_t *cb = std::upper_bound(cr.first, cr.second, last), *ce = cr.second; // the child's rows beyond this supernode's pivots
                    if (cb == ce) continue;
                    merged.clear();
                    merged.reserve(rows.size() + (size_t)(ce - cb));
                    std::set_union(rows.begin(), rows.end(), cb, ce, std::back_inserter(merged));
                    rows.swap(merged);
                }
            } else {
                for (int32_t c = S.child_ptr[s]; c < S.child_ptr[s + 1]; c++) {
                    const std::pair<const int32_t *, const int32_t *> cr = child(S.child_idx[c]);
                    for (const int32_t *q = cr.first; q < cr.second; q++) {
                        const int32_t i = *q;
                        if (i > last && mark[i] != s) {
                            mark[i] = s;
                            rows.push_back(i);
                        }
                    }
                }
                std::sort(rows.begin(), rows.end());
            }
        };
        std::atomic<bool> oom{false};
        if (!chunks.empty()) {
            std::atomic<size_t> next{0};
            auto body = [&]() {
                try {
                    Scratch t;
                    for (size_t ci; (ci = next.fetch_add(1, std::memory_order_relaxed)) < chunks.size();) {
                        Chunk &C = chunks[ci];
                        C.ptr.assign(1, 0);
                        auto child = [&](int32_t ch) { // (a chunk is closed under children)
                            return std::make_pair((const int32_t *)C.rows.data() + C.ptr[(size_t)(ch - C.lo)], (const int32_t *)C.rows.data() + C.ptr[(size_t)(ch - C.lo) + 1]);
                        };
                        for (int32_t s = C.lo; s <= C.hi; s++) {
                            sn_struct(s, t, child);
                            C.rows.insert(C.rows.end(), t.rows.begin(), t.rows.end());
                            C.ptr.push_back((int64_t)C.rows.size());
                        }
                    }
                } catch (const std::bad_alloc &) { // (an exception must not leave a thread)
                    oom.store(true);
                }
            };
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; i++) pool.emplace_back(body);
            for (auto &th : pool) th.join();
            if (oom.load()) return -41;
        }
        // the supernodes above the chunks, in order (all of them when nothing was chunked)
        std::vector<int32_t> top_rows;
        std::vector<int64_t> top_ptr((size_t)S.nsuper + 1, 0); // offsets of the top supernodes (entries of chunked ones unused)
        {
            Scratch t;
            auto child = [&](int32_t ch) {
                const int32_t c = chunk_of[ch];
                if (c >= 0) {
                    const Chunk &C = chunks[(size_t)c];
                    return std::make_pair((const int32_t *)C.rows.data() + C.ptr[(size_t)(ch - C.lo)], (const int32_t *)C.rows.data() + C.ptr[(size_t)(ch - C.lo) + 1]);
                }
                return std::make_pair((const int32_t *)top_rows.data() + top_ptr[ch], (const int32_t *)top_rows.data() + top_ptr[ch] + S.sn_rowptr[ch + 1]);
            };
            // (S.sn_rowptr[s + 1] holds the SIZE of s until the prefix sum below)
            for (int32_t s = 0; s < S.nsuper; s++) {
                const int32_t c = chunk_of[s];
                if (c >= 0) {
                    const Chunk &C = chunks[(size_t)c];
                    S.sn_rowptr[s + 1] = C.ptr[(size_t)(s - C.lo) + 1] - C.ptr[(size_t)(s - C.lo)];
                    continue;
                }
                top_ptr[s] = (int64_t)top_rows.size();
                sn_struct(s, t, child);
                top_rows.insert(top_rows.end(), t.rows.begin(), t.rows.end());
                S.sn_rowptr[s + 1] = (int64_t)t.rows.size();
            }
        }
        for (int32_t s = 0; s < S.nsuper; s++) S.sn_rowptr[s + 1] += S.sn_rowptr[s];
        S.sn_rows.resize((size_t)S.sn_rowptr[S.nsuper]);
        // copies: every chunk is one contiguous piece of the final array; the top supernodes one piece each
        auto place_top = [&]() {
            for (int32_t s = 0; s < S.nsuper; s++)
                if (chunk_of[s] < 0 && S.sn_rowptr[s + 1] > S.sn_rowptr[s])
                    std::memcpy(S.sn_rows.data() + S.sn_rowptr[s], top_rows.data() + top_ptr[s], sizeof(int32_t) * (size_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]));
        };
        if (!chunks.empty()) {
            std::atomic<size_t> next{0};
            auto body = [&]() {
                for (size_t ci; (ci = next.fetch_add(1, std::memory_order_relaxed)) < chunks.size();) {
                    Chunk &C = chunks[ci];
                    if (!C.rows.empty()) std::memcpy(S.sn_rows.data() + S.sn_rowptr[C.lo], C.rows.data(), sizeof(int32_t) * C.rows.size());
                    std::vector<int32_t>().swap(C.rows);
                }
            };
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; i++) pool.emplace_back(body);
            place_top();
            for (auto &th : pool) th.join();
        } else
            place_top();
    }
    // relative indices into the parent's front
    S.rel.resize(S.sn_rows.size()); // (every entry is written below, or the analysis fails)
    {
        // (every supernode writes its own range of rel and reads its parent's rows: independent, spread over the host threads)
        std::atomic<int> rel_err{0};
        auto rel_range = [&](int32_t s0, int32_t s1) {
            for (int32_t s = s0; s < s1; s++) {
                int32_t t = S.sn_parent[s];
                if (t < 0) {
                    if (S.nrow(s) != 0) rel_err.store(-20); // a root must have an empty off-diagonal structure
                    continue;
                }
                int32_t tf = S.sn_first[t], tl = S.sn_first[t + 1] - 1, tp = S.npiv(t);
                int64_t q = S.sn_rowptr[t], qe = S.sn_rowptr[t + 1];
                for (int64_t p = S.sn_rowptr[s]; p < S.sn_rowptr[s + 1]; p++) {
                    int32_t i = S.sn_rows[p];
                    if (i <= tl) {
                        if (i < tf) {
                            rel_err.store(-21);
                            break;
                        }
                        S.rel[p] = i - tf;
                    } else {
                        while (q < qe && S.sn_rows[q] < i) q++;
                        if (q >= qe || S.sn_rows[q] != i) {
                            rel_err.store(-22);
                            break;
                        }
                        S.rel[p] = tp + (int32_t)(q - S.sn_rowptr[t]);
                    }
                }
            }
        };
        if (S.sn_rows.size() > (size_t)1 << 20 && threads > 1) {
            // ranges of supernodes with about the same number of rows
            const int nt = threads;
            std::vector<std::thread> pool;
            int32_t s0 = 0;
            for (int t = 0; t < nt; t++) {
                const int64_t target = (int64_t)S.sn_rows.size() * (t + 1) / nt;
                int32_t s1 = (int32_t)(std::upper_bound(S.sn_rowptr.begin(), S.sn_rowptr.end(), target) - S.sn_rowptr.begin()) - 1;
                s1 = t == nt - 1 ? S.nsuper : std::max(s0, std::min(s1, S.nsuper));
                if (s1 > s0) pool.emplace_back([=]() { rel_range(s0, s1); });
                s0 = s1;
            }
            for (auto &th : pool) th.join();
        } else
            rel_range(0, S.nsuper);
        if (rel_err.load() != 0) return rel_err.load();
    }

    S.seconds_phase[4] = since(t_phase), t_phase = clk::now();
    // ---- levels -----------------------------------------------------------------------------
    S.sn_level.assign((size_t)S.nsuper, 0);
    for (int32_t s = 0; s < S.nsuper; s++) {
        int32_t t = S.sn_parent[s];
        if (t >= 0 && S.sn_level[t] < S.sn_level[s] + 1) S.sn_level[t] = S.sn_level[s] + 1;
    }
    S.nlevels = 0;
    for (int32_t s = 0; s < S.nsuper; s++) S.nlevels = std::max(S.nlevels, S.sn_level[s] + 1);
    S.level_ptr.assign((size_t)S.nlevels + 1, 0);
    for (int32_t s = 0; s < S.nsuper; s++) S.level_ptr[S.sn_level[s] + 1]++;
    for (int32_t l = 0; l < S.nlevels; l++) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_sn.resize((size_t)S.nsuper);
    {
        std::vector<int32_t> w(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (int32_t s = 0; s < S.nsuper; s++) S.level_sn[w[S.sn_level[s]]++] = s;
    }

    // ---- front pool layout and statistics ----------------------------------------------------
    // Fronts larger than opt.augment_above are factorised AUGMENTED: the partial LU runs on [F Ic; Ir 0] and leaves
    //   E  = [inv(L11) P ; -L21 inv(L11) P]   f x p   (forward-solve panel)
    //   E' = [inv(U11) , -inv(U11) U12]       p x f   (backward-solve panel; absent in symmetric mode, which applies E^T)
    // so the triangular solves of big supernodes become dependency-free GEMVs (see kernels_common.hpp).  E / E' and the small
    // fronts are persistent; the f x f working block of a big front lives in an arena from its level until its parent's
    // level has consumed the contribution block (fronts are processed level by level), then the storage is re-used.
    S.sym_mode = opt.symmetric_ldlt && sym_lower;
    S.front_off.assign((size_t)S.nsuper, 0);
    S.front_ld.assign((size_t)S.nsuper, 0);
    S.e_off.assign((size_t)S.nsuper, -1);
    S.ep_off.assign((size_t)S.nsuper, -1);
    S.front_ldp.assign((size_t)S.nsuper, 0);
    const bool arena_reuse = !(getenv("HIPMF_ARENA_REUSE") && atoi(getenv("HIPMF_ARENA_REUSE")) == 0); // (debug knob: 0 = every block keeps its own storage)
    auto round16 = [](int64_t v) { return (v + 15) / 16 * 16; }; // 128-byte granules for the large blocks
    // column stride of a big front's working block and of its E panel (measured: padding the stride to f + p, to a multiple of 16 or to
    // an odd multiple of 16 changes nothing on MI355X; the kernels take any stride >= f)
    // Round 5: the strides of a big front's E (and working block) and -- above 64 pivots -- of its E' are multiples of 16 doubles: the
    // top-level solve slabs read 8 rows (64 bytes) of every column, and with odd strides those segments straddled 128-byte lines that
    // the neighbouring slabs (other XCDs) fetched again -- the pass moved 1.76x the stored factor (profiles/r04_sptrsv_traffic.json).
    // Fronts with few pivots keep E' packed (stride p): their backward slabs read whole columns, i.e. the block end to end.
    const bool align_panels = !(getenv("HIPMF_ALIGN_PANELS") && atoi(getenv("HIPMF_ALIGN_PANELS")) == 0);
    auto ld_of = [&](int64_t f, int64_t) -> int64_t { return align_panels ? (f + 15) / 16 * 16 : f; };
    int64_t pers = 0;
    for (int32_t s = 0; s < S.nsuper; s++) {
        int64_t p = S.npiv(s), m = S.nrow(s), f = p + m;
        S.front_ld[s] = (int32_t)(f > opt.augment_above ? ld_of(f, p) : f);
        S.front_ldp[s] = (int32_t)((f > opt.augment_above && align_panels && p > 64) ? (p + 15) / 16 * 16 : p);
        if (f > opt.augment_above) {
            S.e_off[s] = pers, pers += round16((int64_t)S.front_ld[s] * p);
            if (!S.sym_mode) S.ep_off[s] = pers, pers += round16((int64_t)S.front_ldp[s] * f);
        } else {
            // (both blocks start on a 128-byte line: the wave-subtree solves fetch them as flat 512-byte pieces, and a piece that
            //  straddles lines costs a fifth line -- measured as HBM fetch bytes of k_wt_fwd / k_wt_bwd; +64 bytes per front on average)
            pers = round16(pers);
            S.front_off[s] = pers, pers += f * f;
            // the rows of U of a small front once more, packed (p x f, stride p): the backward solve reads [U11 | U12] as one
            // contiguous block instead of p-entry pieces of f columns (which drags the whole f x f block through the cache lines)
            if (p > 0 && m > 0) pers = round16(pers), S.ep_off[s] = pers, pers += p * f;
        }
        S.nnz_l += p * (p - 1) / 2 + p * m;
        S.nnz_u += p * (p + 1) / 2 + p * m;
        double dp = (double)p, dm = (double)m;
        S.flops += 2.0 / 3.0 * dp * dp * dp + 2.0 * dp * dp * dm + 2.0 * dp * dm * dm;
        S.flops_gemm += 2.0 * dp * dm * dm;
        S.max_front = std::max<int32_t>(S.max_front, (int32_t)f);
        S.max_pivots = std::max<int32_t>(S.max_pivots, (int32_t)p);
    }
    pers = round16(pers);
    S.persist_doubles = pers;
    {
        // arena plan: best fit over a coalescing free list, levels in execution order
        // (free blocks by offset for coalescing and by (size, offset) for the best-fit search: the same choices as a linear scan over a
        //  list sorted by offset -- smallest sufficient block, lowest offset among equals -- in O(log) per request; the scan took
        //  0.6 - 0.8 s of the analysis of the 200^3 matrix, 190 000 tiled fronts)
        std::map<int64_t, int64_t> by_off;             // offset -> size
        std::set<std::pair<int64_t, int64_t>> by_size; // (size, offset)
        int64_t top = 0;
        auto take = [&](int64_t sz) -> int64_t {
            auto it = by_size.lower_bound(std::make_pair(sz, (int64_t)INT64_MIN));
            if (it != by_size.end()) {
                const int64_t bsz = it->first, off = it->second;
                by_size.erase(it);
                by_off.erase(off);
                if (bsz > sz) {
                    by_off.emplace(off + sz, bsz - sz);
                    by_size.emplace(bsz - sz, off + sz);
                }
                return off;
            }
            if (!by_off.empty()) { // grow the free block at the end of the arena
                auto last = std::prev(by_off.end());
                if (last->first + last->second == top) {
                    const int64_t off = last->first;
                    by_size.erase(std::make_pair(last->second, last->first));
                    by_off.erase(last);
                    top = off + sz;
                    return off;
                }
            }
            const int64_t off = top;
            top += sz;
            return off;
        };
        auto give = [&](int64_t off, int64_t sz) {
            auto nx = by_off.lower_bound(off);
            if (nx != by_off.end() && off + sz == nx->first) { // merge with the block behind
                sz += nx->second;
                by_size.erase(std::make_pair(nx->second, nx->first));
                nx = by_off.erase(nx);
            }
            if (nx != by_off.begin()) {
                auto pv = std::prev(nx);
                if (pv->first + pv->second == off) { // merge with the block in front
                    off = pv->first, sz += pv->second;
                    by_size.erase(std::make_pair(pv->second, pv->first));
                    by_off.erase(pv);
                }
            }
            by_off.emplace(off, sz);
            by_size.emplace(sz, off);
        };
        std::vector<std::vector<int32_t>> expire((size_t)S.nlevels);
        std::vector<int32_t> bigs;
        for (int32_t l = 0; l < S.nlevels; l++) {
            bigs.clear();
            for (int32_t k = S.level_ptr[l]; k < S.level_ptr[l + 1]; k++)
                if (S.fsize(S.level_sn[k]) > opt.augment_above) bigs.push_back(S.level_sn[k]);
            std::stable_sort(bigs.begin(), bigs.end(), [&](int32_t a, int32_t b) { return S.fsize(a) > S.fsize(b); });
            for (int32_t s : bigs) {
                const int64_t f = S.fsize(s);
                S.front_off[s] = pers + take(round16((int64_t)S.front_ld[s] * f));
                const int32_t t = S.sn_parent[s];
                // The block is needed until the parent's level has pulled the contribution block.  The blocks released after level l
                // are handed out from level l + 1 on, whose blocks are zero-filled while level l is still being factorised (side
                // stream, after level l's extend-add): a front without a parent / without a block therefore keeps its storage one
                // level longer than its own.
                const int32_t last = (t < 0 || S.nrow(s) == 0) ? l + 1 : S.sn_level[t];
                if (last < S.nlevels) expire[(size_t)last].push_back(s);
            }
            if (arena_reuse)
                for (int32_t s : expire[(size_t)l]) give(S.front_off[s] - pers, round16((int64_t)S.front_ld[s] * S.fsize(s)));
        }
        S.temp_doubles = top;
    }
    S.pool_estimate_bytes = 8.0 * (double)(S.persist_doubles + S.temp_doubles);
    {
        const double limit = opt.pool_limit_live ? opt.pool_limit_live->load() : opt.pool_limit_bytes;
        if (limit > 0.0 && S.pool_estimate_bytes > limit) return -40;
    }

    S.seconds_phase[5] = since(t_phase), t_phase = clk::now();
    // ---- assembly map: where every input entry lands ------------------------------------------
    S.amap.assign((size_t)S.nnz_a, -1);
    S.amap_sn.assign((size_t)S.nnz_a, -1);
    if (sym_lower) S.amap2.assign((size_t)S.nnz_a, -1);
    auto local = [&](int32_t s, int32_t i) -> int64_t {
        int32_t first = S.sn_first[s], last = S.sn_first[s + 1] - 1;
        if (i <= last) return i - first;
        const int32_t *b = S.sn_rows.data() + S.sn_rowptr[s], *e = S.sn_rows.data() + S.sn_rowptr[s + 1];
        const int32_t *it = std::lower_bound(b, e, i);
        if (it == e || *it != i) return -1;
        return (int64_t)S.npiv(s) + (it - b);
    };
    std::atomic<int> amap_err{0};
    parallel_ranges(n, threads, [&](int32_t r0, int32_t r1) {
        for (int32_t r = r0; r < r1; r++)
            for (int32_t p = rp[r]; p < rp[r + 1]; p++) {
                int32_t i = S.pinv[r], j = S.pinv[ci[p]];
                if (sym_lower && ci[p] > r) { // lower storage promised
                    amap_err.store(-30);
                    return;
                }
                int32_t s = S.sn_of[std::min(i, j)];
                int64_t f = S.front_ld[s];
                int64_t li = local(s, i), lj = local(s, j);
                if (li < 0 || lj < 0) {
                    amap_err.store(-31);
                    return;
                }
                S.amap_sn[p] = s;
                if (S.sym_mode && S.fsize(s) > opt.augment_above) {
                    // L D L^T front: the entry goes to the lower triangle only
                    S.amap[p] = S.front_off[s] + std::max(li, lj) + std::min(li, lj) * f;
                } else {
                    S.amap[p] = S.front_off[s] + li + lj * f;
                    if (sym_lower && i != j) S.amap2[p] = S.front_off[s] + lj + li * f;
                }
            }
    });
    if (amap_err.load() != 0) return amap_err.load();
    S.seconds_phase[6] = since(t_phase);
    S.seconds_total = since(t_all);
    return 0;
}

} // namespace hipmf

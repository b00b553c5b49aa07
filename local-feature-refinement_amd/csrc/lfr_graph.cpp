// Graph stage of the solver path, host side (exact greedy semantics of the reference):
//   tracks      solve.cc:489-549  constrained maximum spanning forest
//   roots       solve.cc:552-582
//   components  solve.cc:252-373  meta graph, connected components, size cap + (substitute) cut
//   assembly    solve.cc:79-143   reduced programs -> 80-byte edge records in residual-block order
// Image names are interned to integers at ingest: every use in the reference is an equality
// test (std::set<std::string> intersection/union, solve.cc:493-519).
#include <algorithm>
#include <climits>
#include <future>
#include <atomic>
#include <chrono>
#include <cstring>
#include <numeric>
#include <queue>
#include <cstdlib>
#include <functional>
#include <thread>

#include <immintrin.h>

#include "lfr_internal.hpp"

namespace lfr {

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static inline uint32_t sim_key(float s) {   // order-preserving float -> uint32 (as the double compare of solve.cc:489)
    if (s == 0.f) s = 0.f;                   // -0.0 == +0.0
    uint32_t b; memcpy(&b, &s, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct SortKey { uint64_t hi; uint32_t lo; uint32_t m; };   // (sim, n1) | n2 | match id

// host worker threads for the embarrassingly parallel parts (sort, assembly): LFR_HOST_THREADS or
// min(hardware threads, 32)
static int host_threads() {
    if (const char *e = getenv("LFR_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 256); }
    const unsigned h = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(h ? h : 1u, 32u));
}
// run f(chunk, lo, hi) over [0, n) split at `cuts` (size chunks+1) on one thread per chunk
static void parallel_chunks(const std::vector<int64_t> &cuts, const std::function<void(int, int64_t, int64_t)> &f) {
    const int chunks = (int)cuts.size() - 1;
    if (chunks <= 1) { if (chunks == 1) f(0, cuts[0], cuts[1]); return; }
    std::vector<std::thread> th;
    for (int c = 1; c < chunks; ++c) th.emplace_back(f, c, cuts[c], cuts[c + 1]);
    f(0, cuts[0], cuts[1]);
    for (auto &t : th) t.join();
}
template <class T, class Cmp>
static void parallel_sort(std::vector<T> &v, Cmp cmp, int threads) {
    const int64_t n = (int64_t)v.size();
    int chunks = 1;
    while (chunks * 2 <= threads && n / (chunks * 2) >= 65536) chunks *= 2;
    std::vector<int64_t> cuts(chunks + 1);
    for (int c = 0; c <= chunks; ++c) cuts[c] = n * c / chunks;
    parallel_chunks(cuts, [&](int, int64_t lo, int64_t hi) { std::sort(v.begin() + lo, v.begin() + hi, cmp); });
    for (int width = 1; width < chunks; width *= 2) {            // pairwise merges, one thread per pair
        std::vector<std::thread> th;
        for (int c = 0; c + width < chunks; c += 2 * width)
            th.emplace_back([&, c, width] { std::inplace_merge(v.begin() + cuts[c], v.begin() + cuts[c + width], v.begin() + cuts[std::min(c + 2 * width, chunks)], cmp); });
        for (auto &t : th) t.join();
    }
}

static inline int32_t uf_root(std::vector<int32_t> &parent, int32_t i) {
    int32_t r = i;
    while (parent[r] >= 0) r = parent[r];
    while (parent[i] >= 0) { const int32_t nx = parent[i]; parent[i] = r; i = nx; }
    return r;
}

// ----------------------------------------------------------------------------------------------
// Deterministic bisection used where the reference calls Graclus through COLMAP
// (colmap::ComputeNormalizedMinGraphCut, solve.cc:192).  Graclus' multilevel kernel k-means is
// third-party, heuristic and not reproducible, so results on inputs that need a cut are NOT
// claimed to match the reference.  Method: maximum-adjacency region growing from
// the smallest node id until half of the total edge-weight volume is absorbed, then one pass of
// boundary refinement on the normalized-cut objective.  All sums are sums of integers held in doubles (exact),
// so the result does not depend on the order of `edges`.
// ----------------------------------------------------------------------------------------------
namespace {
// A (sub)graph of the cut recursion in compact form: node i is ids[i] (ascending), edges join local indices.
// Hash maps keyed by track id and per-edge binary searches were most of the host time of a cut (config 5: 0.6 M meta
// edges, twelve levels): the recursion hands local indices down instead.
struct SubGraph {
    std::vector<int> ids;
    std::vector<int> ea, eb, w;
};

// Scratch of bisect_core, one per thread, grown and kept: a cut recursion calls it thousands of times, and fresh multi-megabyte vectors
// (mapped, faulted in, unmapped every call) cost more than the arithmetic on them.
#ifndef LFR_CUT_LOPSIDED_GAIN
#define LFR_CUT_LOPSIDED_GAIN 0.6
#endif
#ifndef LFR_CUT_SPAWN_MIN
#define LFR_CUT_SPAWN_MIN 2500
#endif
struct BisectScratch {
    struct Nb { int node, w; };                                   // (integer weights: every sum below is exact in doubles)
    std::vector<uint32_t> off, fill;
    std::vector<Nb> nbr;
    struct HeapEntry { int64_t key; int id; };
    std::vector<double> deg, ext;
    std::vector<char> in;
    std::vector<HeapEntry> heap;
    std::vector<int> pos, order;
    std::vector<int32_t> att;
    void release() { *this = BisectScratch(); }
};
BisectScratch &bisect_scratch() { static thread_local BisectScratch ws; return ws; }

// Arg-max of (attachment, smallest id) over a plain array: what the heap below answers, for the dense graphs of a cut recursion.  The
// meta graph of config 5 has 2878 nodes of mean degree 73 and stays that dense down the recursion: absorbing a node raised 73 heap keys
// (14 ns each with the sifts) to save a scan of 2878 int32 values that AVX2 does in 0.15 us.  Absorbed nodes hold INT32_MIN.
__attribute__((target("avx2"))) int argmax_first_avx2(const int32_t *a, int n) {
    __m256i vmax = _mm256_set1_epi32(INT32_MIN);
    int i = 0;
    for (; i + 8 <= n; i += 8) vmax = _mm256_max_epi32(vmax, _mm256_loadu_si256(reinterpret_cast<const __m256i *>(a + i)));
    alignas(32) int32_t lanes[8];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), vmax);
    int32_t m = lanes[0];
    for (int k = 1; k < 8; ++k) m = std::max(m, lanes[k]);
    for (; i < n; ++i) m = std::max(m, a[i]);
    const __m256i vm = _mm256_set1_epi32(m);
    for (i = 0; i + 8 <= n; i += 8) {
        const unsigned mask = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(vm, _mm256_loadu_si256(reinterpret_cast<const __m256i *>(a + i)))));
        if (mask) return i + __builtin_ctz(mask);
    }
    for (; i < n; ++i) if (a[i] == m) return i;
    return -1;
}
// (LFR_BISECT_HEAP_ONLY: tests run the heap against the scan; the growth order - hence the partition - is the same either way)
bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2") && !getenv("LFR_BISECT_HEAP_ONLY"); return v; }

void bisect_core(const SubGraph &g, std::vector<char> &side) {
    const int n = (int)g.ids.size();
    side.assign(n, 1);
    if (n == 0) return;
    const size_t E = g.ea.size();
    BisectScratch &ws = bisect_scratch();
    using Nb = BisectScratch::Nb;
    std::vector<uint32_t> &off = ws.off, &fill = ws.fill;
    off.assign(n + 1, 0);                                         // CSR adjacency, neighbours in edge order
    for (size_t k = 0; k < E; ++k) { ++off[g.ea[k] + 1]; ++off[g.eb[k] + 1]; }
    for (int i = 0; i < n; ++i) off[i + 1] += off[i];
    if (ws.nbr.size() < 2 * E) ws.nbr.resize(2 * E);
    Nb *nbr = ws.nbr.data();
    std::vector<double> &deg = ws.deg;
    deg.assign(n, 0.0);
    double volume = 0.0;
    {
        fill.assign(off.begin(), off.end() - 1);
        for (size_t k = 0; k < E; ++k) {
            const int a = g.ea[k], b = g.eb[k], wi = std::max(g.w[k], 1);
            const double w = (double)wi;
            nbr[fill[a]++] = Nb{b, wi};
            nbr[fill[b]++] = Nb{a, wi};
            deg[a] += w; deg[b] += w; volume += 2 * w;
        }
    }
    std::vector<char> &in = ws.in;
    in.assign(n, 0);
    double vol0 = 0.0, cut = 0.0;
    int n0 = 0;
    // region growing: always absorb the outside node with the largest attachment to the region (ties ->
    // smallest id; a node nobody is attached to yet has attachment 0, so an emptied frontier restarts from the
    // smallest unvisited id).  Indexed binary max-heap over the n nodes keyed (attachment, -id) with increase-key:
    // O(E log n) with n entries (a lazy-deletion queue held 2 E of them; the O(n^2) arg-max scan before that).
    // Round 4: the heap holds its keys (the attachments are sums of integers: exact in an int64) - every comparison of a sift read
    // attach[heap[slot]], two dependent loads; the growing is half of a bisection's time and the bisections of a cut are a chain.
    std::vector<BisectScratch::HeapEntry> &heap = ws.heap;
    std::vector<int> &pos = ws.pos;
    // Round 6: dense graphs keep the attachments in a plain int32 array and take the arg-max with a vector scan (argmax_first_avx2): the
    // same (largest attachment, smallest id) the heap pops, so the growth order - and the partition - is the same.  Chosen where the
    // scans (n / 8 vector steps per absorbed node) are cheaper than raising a heap key per visited neighbour, and only when every
    // attachment fits an int32 (the volume does).
    const bool scan_mode = have_avx2() && volume < 2e9 && (double)n * (double)n < 640.0 * (double)E;
    std::vector<int32_t> &att = ws.att;
    if (scan_mode) att.assign(n, 0);
    else {
        heap.resize(n); pos.resize(n);
        for (int i = 0; i < n; ++i) { heap[i] = BisectScratch::HeapEntry{0, i}; pos[i] = i; }      // all keys (0, -i): already a heap
    }
    int hn = n;
    auto above = [](const BisectScratch::HeapEntry &a, const BisectScratch::HeapEntry &b) { return a.key > b.key || (a.key == b.key && a.id < b.id); };
    auto sift_up = [&](int i) {
        const BisectScratch::HeapEntry v = heap[i];
        while (i > 0) { const int p = (i - 1) >> 1; if (!above(v, heap[p])) break; heap[i] = heap[p]; pos[heap[i].id] = i; i = p; }
        heap[i] = v; pos[v.id] = i;
    };
    auto sift_down = [&](int i) {
        const BisectScratch::HeapEntry v = heap[i];
        for (;;) {
            int c = 2 * i + 1;
            if (c >= hn) break;
            if (c + 1 < hn && above(heap[c + 1], heap[c])) ++c;
            if (!above(heap[c], v)) break;
            heap[i] = heap[c]; pos[heap[i].id] = i; i = c;
        }
        heap[i] = v; pos[v.id] = i;
    };
    // Round 3: the region does not stop at half of the volume - it grows on to three quarters and the prefix of the growth order
    // with the SMALLEST normalized cut between a quarter and three quarters of the volume wins (ties -> the earliest); a graph
    // that is a chain of clusters is then cut at a bottleneck instead of wherever half of the volume happens to be reached.
    // scripts/cut_quality.py measures what that buys against a spectral sweep cut (profiles/r03_cut_quality.json).
    auto ncut = [&](double c, double v0) { const double v1 = volume - v0; return (v0 > 0 && v1 > 0) ? c / v0 + c / v1 : 1e300; };
    std::vector<int> &order = ws.order;
    order.clear();
    order.reserve(n);
    int best_len = -1, half_len = -1;
    double best_val = 1e300, best_cut = 0.0, best_vol = 0.0, half_cut = 0.0, half_vol = 0.0;
    while (vol0 * 4 < 3 * volume && n0 < n - 1 && hn > 0) {
        int best;
        double attach_best;
        if (scan_mode) {
            best = argmax_first_avx2(att.data(), n);
            attach_best = (double)att[best];
            att[best] = INT32_MIN;
            --hn;
        } else {
            best = heap[0].id;
            attach_best = (double)heap[0].key;
            heap[0] = heap[--hn]; pos[heap[0].id] = 0;
            if (hn > 0) sift_down(0);
        }
        in[best] = 1; vol0 += deg[best]; ++n0;
        order.push_back(best);
        cut += deg[best] - 2.0 * attach_best;             // its edges to the outside enter the cut, those to the region leave it
        if (scan_mode) {
            for (uint32_t q = off[best]; q < off[best + 1]; ++q) {
                const int v = nbr[q].node;
                if (!in[v]) att[v] += nbr[q].w;           // (nobody reads the attachment of a node inside the region again)
            }
        } else {
            for (uint32_t q = off[best]; q < off[best + 1]; ++q) {
                const int v = nbr[q].node;
                if (in[v]) continue;
                const int at = pos[v];
                heap[at].key += nbr[q].w;
                sift_up(at);
            }
        }
        if (half_len < 0 && vol0 * 2 >= volume) { half_len = n0; half_cut = cut; half_vol = vol0; }
        if (vol0 * 4 >= volume && vol0 * 4 <= 3 * volume) {
            const double val = ncut(cut, vol0);
            if (val < best_val) { best_val = val; best_len = n0; best_cut = cut; best_vol = vol0; }
        }
    }
    if (best_len < 0) {                                   // no prefix inside the window (a node of huge degree): the round-2 rule
        if (half_len < 0) { half_len = n0; half_cut = cut; half_vol = vol0; }
        best_len = half_len; best_cut = half_cut; best_vol = half_vol;
    } else if (half_len > 0) {
        // a lopsided prefix has to EARN its place: unless its normalized cut beats the balanced (half-volume) prefix by the factor
        // below, the balanced one wins - a graph without a bottleneck (config 5's meta graph: random wrong matches between dense tracks)
        // otherwise recursed 27 levels deep for 4 % less dropped similarity
        // (a build-time constant, -DLFR_CUT_LOPSIDED_GAIN=x for experiments: the partition - and with it labels and positions - is a pure
        // function of the input, not of the process environment)
        constexpr double need = LFR_CUT_LOPSIDED_GAIN;
        const double half_val = ncut(half_cut, half_vol);
        if (!(best_val < need * half_val)) { best_len = half_len; best_cut = half_cut; best_vol = half_vol; }
    }
    for (int i = 0; i < best_len; ++i) side[order[i]] = 0;
    cut = best_cut; vol0 = best_vol; n0 = best_len;
    // refinement sweeps in node order: move a node if that lowers cut/vol0 + cut/vol1 (never emptying a side); up to eight sweeps,
    // stopping with the first one that moves nothing.  ext[i] = weight of i's edges to the other side, kept up to date move by move
    // (integers in doubles: the same values a fresh count over the neighbours would give).
    std::vector<double> &ext = ws.ext;
    ext.assign(n, 0.0);
    for (size_t k = 0; k < E; ++k) {
        const int a = g.ea[k], b = g.eb[k];
        if (side[a] != side[b]) { const double w = (double)std::max(g.w[k], 1); ext[a] += w; ext[b] += w; }
    }
    double ncut_now = ncut(cut, vol0);                      // (a function of (cut, vol0): recomputed when a move changes them, not per node)
    for (int sweep = 0; sweep < 8; ++sweep) {
        bool moved = false;
        for (int i = 0; i < n; ++i) {
            const double to_other = ext[i], to_same = deg[i] - ext[i];
            const double c2 = cut + to_same - to_other;
            const double v2 = side[i] == 0 ? vol0 - deg[i] : vol0 + deg[i];
            const int cnt0 = side[i] == 0 ? n0 - 1 : n0 + 1;
            if (cnt0 <= 0 || cnt0 >= n) continue;
            if (ncut(c2, v2) < ncut_now) {
                const char was = side[i];
                side[i] ^= 1; cut = c2; vol0 = v2; n0 = cnt0; moved = true;
                ncut_now = ncut(cut, vol0);
                ext[i] = to_same;
                for (uint32_t q = off[i]; q < off[i + 1]; ++q) {
                    const int v = nbr[q].node;
                    if (v == i) continue;
                    ext[v] += side[v] == was ? (double)nbr[q].w : -(double)nbr[q].w;
                }
            }
        }
        if (!moved) break;
    }
}

// nodes = endpoints of the edges, ascending; edges in local indices
SubGraph compact(const std::vector<std::pair<int, int>> &edges, const std::vector<int> &weights) {
    SubGraph g;
    g.ea.resize(edges.size()); g.eb.resize(edges.size()); g.w = weights;
    int lo = INT32_MAX, hi = -1;
    for (auto &e : edges) { lo = std::min({lo, e.first, e.second}); hi = std::max({hi, e.first, e.second}); }
    if (!edges.empty() && lo >= 0 && (size_t)hi - lo <= 8 * edges.size() + 1024) {
        // ids in a dense range (track ids are): mark, rank, look up - sorting 2 E ids and two binary searches per edge were a
        // quarter of the whole cut
        std::vector<int> rank((size_t)hi - lo + 2, 0);
        for (auto &e : edges) { rank[e.first - lo + 1] = 1; rank[e.second - lo + 1] = 1; }
        for (size_t i = 1; i < rank.size(); ++i) {
            if (rank[i]) g.ids.push_back((int)(i - 1) + lo);
            rank[i] += rank[i - 1];
        }
        for (size_t k = 0; k < edges.size(); ++k) { g.ea[k] = rank[edges[k].first - lo]; g.eb[k] = rank[edges[k].second - lo]; }
        return g;
    }
    g.ids.reserve(2 * edges.size());
    for (auto &e : edges) { g.ids.push_back(e.first); g.ids.push_back(e.second); }
    std::sort(g.ids.begin(), g.ids.end());
    g.ids.erase(std::unique(g.ids.begin(), g.ids.end()), g.ids.end());
    auto local = [&](int id) { return (int)(std::lower_bound(g.ids.begin(), g.ids.end(), id) - g.ids.begin()); };
    for (size_t k = 0; k < edges.size(); ++k) { g.ea[k] = local(edges[k].first); g.eb[k] = local(edges[k].second); }
    return g;
}

// recursive_graph_cut of solve.cc:185-250 with bisect_core in place of Graclus: out[i] = subset of node i.
// The two halves of a bisection are independent sub-problems: the second one runs on another thread while this one
// handles the first (big halves only, a bounded number of tasks); the results are merged in the reference's order
// (solve.cc:210-246: subset 0 first), so the numbering does not depend on the schedule.
std::atomic<int> g_cut_tasks{0};
void cut_rec(const SubGraph &g, const std::vector<int64_t> &node_weights, int64_t max_weight, std::vector<int> &out) {
    const int n = (int)g.ids.size();
    std::vector<char> side;
    bisect_core(g, side);
    int64_t subset_w[2] = {0, 0};
    for (int i = 0; i < n; ++i) subset_w[(int)side[i]] += node_weights[g.ids[i]];
    // the half of an oversized side that still has edges: nodes with an internal edge, in ascending order
    SubGraph child[2];
    std::vector<int> up[2];                                        // child index -> index here
    const bool over[2] = {subset_w[0] > max_weight, subset_w[1] > max_weight};
    if (over[0] || over[1]) {                                      // one pass over the edges for both sides (a node has one side)
        std::vector<int> down(n, -1);
        for (int s = 0; s < 2; ++s) if (over[s]) { child[s].ea.reserve(g.ea.size() / 3); child[s].eb.reserve(g.ea.size() / 3); child[s].w.reserve(g.ea.size() / 3); }
        for (size_t k = 0; k < g.ea.size(); ++k) {
            const int a = g.ea[k], b = g.eb[k], sa = side[a];
            if (sa != side[b] || !over[sa]) continue;
            down[a] = 0; down[b] = 0;
            child[sa].ea.push_back(a); child[sa].eb.push_back(b); child[sa].w.push_back(g.w[k]);
        }
        for (int i = 0; i < n; ++i) if (down[i] == 0) { const int s = side[i]; down[i] = (int)up[s].size(); up[s].push_back(i); child[s].ids.push_back(g.ids[i]); }
        for (int s = 0; s < 2; ++s)
            for (size_t k = 0; k < child[s].ea.size(); ++k) { child[s].ea[k] = down[child[s].ea[k]]; child[s].eb[k] = down[child[s].eb[k]]; }
    }
    std::vector<int> sub[2];
    std::shared_ptr<PoolTask> second;
    bool spawned = false;
    // (a thread costs 0.1-1 ms to start: only halves whose own bisection takes longer than that: -DLFR_CUT_SPAWN_MIN; the labels do not depend on it)
    static const size_t spawn_min = [] { const char *e = getenv("LFR_CUT_SPAWN_MIN"); return e ? (size_t)atoll(e) : (size_t)LFR_CUT_SPAWN_MIN; }();   // (experiments)
    if (!child[0].ea.empty() && child[1].ea.size() >= spawn_min) {
        if (g_cut_tasks.fetch_add(1) < 64) {
            second = pool_async([&] { cut_rec(child[1], node_weights, max_weight, sub[1]); });
            spawned = true;
        } else g_cut_tasks.fetch_sub(1);
    }
    {
        // the task reads this frame's locals: should the first half throw, it must have finished before the frame unwinds
        struct Joiner {
            std::shared_ptr<PoolTask> &t; bool pending;
            ~Joiner() { if (pending) { try { pool_wait(t); } catch (...) {} g_cut_tasks.fetch_sub(1); } }
        } joiner{second, spawned};
        if (!child[0].ea.empty()) cut_rec(child[0], node_weights, max_weight, sub[0]);
        if (spawned) { joiner.pending = false; g_cut_tasks.fetch_sub(1); pool_wait(second); }       // (rethrows what the second half threw)
        else if (!child[1].ea.empty()) cut_rec(child[1], node_weights, max_weight, sub[1]);
    }
    out.assign(n, -1);
    int max_idx = 0;
    for (int s = 0; s < 2; ++s) {
        if (subset_w[s] <= max_weight) {
            for (int i = 0; i < n; ++i) if (side[i] == s) out[i] = max_idx;
            ++max_idx;
            continue;
        }
        if (!child[s].ea.empty()) {
            int new_max = max_idx;
            for (size_t j = 0; j < sub[s].size(); ++j) { out[up[s][j]] = max_idx + sub[s][j]; new_max = std::max(new_max, max_idx + sub[s][j]); }
            max_idx = new_max + 1;
        }
        for (int i = 0; i < n; ++i) if (side[i] == s && out[i] < 0) out[i] = max_idx++;
    }
}
}  // namespace

void bisect_graph(const std::vector<std::pair<int, int>> &edges, const std::vector<int> &weights,
                  std::unordered_map<int, int> &part) {
    part.clear();
    const SubGraph g = compact(edges, weights);
    std::vector<char> side;
    bisect_core(g, side);
    for (size_t i = 0; i < g.ids.size(); ++i) part[g.ids[i]] = side[i];
}

std::unordered_map<int, int> recursive_cut(const std::vector<std::pair<int, int>> &edges,
                                                  const std::vector<int> &weights,
                                                  const std::vector<int64_t> &node_weights, int64_t max_weight) {
    const SubGraph g = compact(edges, weights);
    std::vector<int> out;
    cut_rec(g, node_weights, max_weight, out);
    bisect_scratch().release();                                   // (the calling thread would otherwise keep multi-megabyte vectors for the life of the process)
    std::unordered_map<int, int> final_map;
    final_map.reserve(g.ids.size() * 2);
    for (size_t i = 0; i < g.ids.size(); ++i) final_map.emplace(g.ids[i], out[i]);
    return final_map;
}


// out-edge CSR of the match graph: out_eid[out_off[n] .. out_off[n+1]) = directed edge ids of node n, ascending =
// the reference's insertion order (graph.cc:17-23)
void build_out_csr(const Graph &g, std::vector<int64_t> &out_off, std::vector<int64_t> &out_eid) {
    const int64_t N = g.n_nodes(), M = g.n_matches();
    out_off.assign(N + 1, 0);
    for (int64_t m = 0; m < M; ++m) { ++out_off[g.m_node1[m] + 1]; ++out_off[g.m_node2[m] + 1]; }
    for (int64_t i = 0; i < N; ++i) out_off[i + 1] += out_off[i];
    out_eid.resize(2 * M);
    std::vector<int64_t> cur(out_off.begin(), out_off.end() - 1);
    for (int64_t m = 0; m < M; ++m) { out_eid[cur[g.m_node1[m]]++] = 2 * m; out_eid[cur[g.m_node2[m]]++] = 2 * m + 1; }
}

// separate_meta_graph (solve.cc:252-373) given the tracks: meta graph, connected components numbered by their
// smallest track (= the BFS order of solve.cc:292-300), size cap by recursive cut, cut edges dropped, re-split.
// Also the continuation of the DEVICE graph stage when a component exceeds the cap: the priority-queue region
// growing of bisect_graph is sequential, the meta graph is small, so the cut stays on the host.
void components_from_tracks(const Graph &g, const std::vector<int64_t> &track, int64_t n_tracks, const std::vector<int64_t> &tsize,
                            int64_t max_nodes, const std::vector<int64_t> *out_off_in, const std::vector<int64_t> *out_eid_in,
                            std::vector<int64_t> &comp, int64_t &n_components, int64_t &n_cut) {
    const int64_t N = g.n_nodes(), M = g.n_matches();
    comp.assign(N, -1);
    n_cut = 0;
    auto edge_dst = [&](int64_t e) -> uint32_t { return (e & 1) ? g.m_node1[e >> 1] : g.m_node2[e >> 1]; };
    std::vector<int32_t> mp(n_tracks, -1);
    for (int64_t m = 0; m < M; ++m) {
        const int64_t ta = track[g.m_node1[m]], tb = track[g.m_node2[m]];
        if (ta == tb) continue;
        const int32_t ra = uf_root(mp, (int32_t)ta), rb = uf_root(mp, (int32_t)tb);
        if (ra != rb) mp[std::max(ra, rb)] = std::min(ra, rb);
    }
    std::vector<int64_t> label(n_tracks, -1);
    int64_t nc = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {      // BFS labelling order of solve.cc:292-300
        const int32_t r = uf_root(mp, (int32_t)t);
        if (label[r] < 0) label[r] = nc++;
        label[t] = label[r];
    }
    std::vector<int64_t> csize(nc, 0);
    for (int64_t t = 0; t < n_tracks; ++t) csize[label[t]] += tsize[t];
    bool any_over = false;
    for (int64_t c = 0; c < nc; ++c) if (csize[c] > max_nodes) { any_over = true; ++n_cut; }
    std::vector<int64_t> final_label = label;
    int64_t n_final = nc;
    if (any_over) {
        std::vector<int64_t> off_local, eid_local;
        if (!out_off_in || !out_eid_in) build_out_csr(g, off_local, eid_local);
        const std::vector<int64_t> &out_off = out_off_in ? *out_off_in : off_local, &out_eid = out_eid_in ? *out_eid_in : eid_local;
        // meta edges of the oversized components (solve.cc:268-289, 322-332)
        std::vector<std::unordered_map<int64_t, double>> meta(n_tracks);
        for (int64_t i = 0; i < N; ++i) {
            const int64_t ts = track[i];
            if (csize[label[ts]] <= max_nodes) continue;
            for (int64_t k = out_off[i]; k < out_off[i + 1]; ++k) {
                const int64_t e = out_eid[k], tt = track[edge_dst(e)];
                if (tt != ts) meta[ts][tt] += (double)g.m_sim[e >> 1];
            }
        }
        std::vector<std::vector<int64_t>> members(nc);
        for (int64_t t = 0; t < n_tracks; ++t) if (csize[label[t]] > max_nodes) members[label[t]].push_back(t);
        std::vector<int64_t> gc(n_tracks, 0);
        int64_t ngc = 0;
        for (int64_t c = 0; c < nc; ++c) {
            if (csize[c] <= max_nodes) { ++ngc; continue; }      // ids only need to differ between components
            std::vector<std::pair<int, int>> e; std::vector<int> w;
            for (int64_t t : members[c]) {
                std::vector<std::pair<int64_t, double>> nb(meta[t].begin(), meta[t].end());
                std::sort(nb.begin(), nb.end());
                for (auto &it : nb) if (t < it.first) { e.push_back({(int)t, (int)it.first}); w.push_back(static_cast<int>(100 * it.second)); }
            }
            auto split = recursive_cut(e, w, tsize, max_nodes);
            int64_t top = 0;
            for (auto &it : split) { gc[it.first] = ngc + it.second; top = std::max(top, gc[it.first]); }
            ngc = top + 1;
        }
        // drop cut meta edges, re-split (solve.cc:345-364): union-find restricted to kept edges
        std::vector<int32_t> mp2(n_tracks, -1);
        for (int64_t m = 0; m < M; ++m) {
            const int64_t ta = track[g.m_node1[m]], tb = track[g.m_node2[m]];
            if (ta == tb) continue;
            const bool over = csize[label[ta]] > max_nodes;
            if (over && gc[ta] != gc[tb]) continue;
            const int32_t ra = uf_root(mp2, (int32_t)ta), rb = uf_root(mp2, (int32_t)tb);
            if (ra != rb) mp2[std::max(ra, rb)] = std::min(ra, rb);
        }
        std::fill(final_label.begin(), final_label.end(), -1);
        n_final = 0;
        for (int64_t t = 0; t < n_tracks; ++t) {
            const int32_t r = uf_root(mp2, (int32_t)t);
            if (final_label[r] < 0) final_label[r] = n_final++;
            final_label[t] = final_label[r];
        }
    }
    for (int64_t i = 0; i < N; ++i) comp[i] = final_label[track[i]];
    n_components = n_final;
}

int block_max_rows() {
    static const int v = [] {
        if (const char *e = getenv("LFR_BLOCK_MAX_ROWS")) { const int x = atoi(e); if (x >= 0 && x <= kBlockMaxRows) return x; }
        return kBlockMaxRows;
    }();
    return v;
}

static int classify(int rows, int64_t n_edges) {
    if (rows <= 8 && n_edges <= 24) return KC_G8;
    if (rows <= 16 && n_edges <= 96) return KC_G16;
    if (rows <= 24 && n_edges <= 192) return KC_G64_2;
    if (rows <= 32 && n_edges <= 320) return KC_G64_4;
    const int lds_max = block_max_rows();
    if (rows <= std::min(kBlockRowsS, lds_max)) return KC_BLOCK;
    if (rows <= std::min(kBlockRowsM, lds_max)) return KC_BLOCK_M;
    if (rows <= lds_max) return KC_BLOCK_L;
    return KC_GLOBAL;
}

int build_problem(const Graph &g, int64_t max_nodes, const int64_t *component_override, Problem &p, bool host_batch) {
    using clock = std::chrono::steady_clock;
    p.g = &g;
    const int64_t N = g.n_nodes(), M = g.n_matches();
    p.track.assign(N, -1); p.comp.assign(N, -1); p.is_root.assign(N, 0);
    p.stats = lfr_problem_stats{};
    if (N == 0) return LFR_OK;
    if (N >= (int64_t)1 << 31) { set_error("more than 2^31 nodes"); return LFR_ERR_UNSUPPORTED; }
    if (max_nodes <= 0) max_nodes = (int64_t)g.image_names.size();          // solve.cc:586

    // ------------------------------------------------------------------ tracks (solve.cc:489-541)
    auto t0 = clock::now();
    std::vector<SortKey> keys(M);
    for (int64_t m = 0; m < M; ++m)
        keys[m] = SortKey{((uint64_t)sim_key(g.m_sim[m]) << 32) | g.m_node1[m], g.m_node2[m], (uint32_t)m};
    parallel_sort(keys, [](const SortKey &a, const SortKey &b) {   // descending (sort + reverse); keys are unique up to duplicates
        if (a.hi != b.hi) return a.hi > b.hi;
        return a.lo > b.lo;
    }, host_threads());
    std::vector<int32_t> parent(N, -1), next(N, -1), tail(N), count(N, 1);
    std::vector<int64_t> stamp(g.image_names.size(), -1);
    for (int64_t i = 0; i < N; ++i) tail[i] = (int32_t)i;
    for (int64_t k = 0; k < M; ++k) {
        const int32_t r1 = uf_root(parent, (int32_t)(keys[k].hi & 0xffffffffu));
        const int32_t r2 = uf_root(parent, (int32_t)keys[k].lo);
        if (r1 == r2) continue;
        // images_in_track[root] == images of the member nodes (all distinct by construction)
        bool conflict = false;
        for (int32_t i = r1; i >= 0; i = next[i]) stamp[g.node_image[i]] = k;
        for (int32_t i = r2; i >= 0; i = next[i]) if (stamp[g.node_image[i]] == k) { conflict = true; break; }
        if (conflict) continue;                                              // solve.cc:509-511
        int32_t big = r1, small = r2;
        if (count[r1] < count[r2]) { big = r2; small = r1; }                  // solve.cc:513-521
        parent[small] = big;
        next[tail[big]] = small; tail[big] = tail[small]; count[big] += count[small];
    }
    std::vector<SortKey>().swap(keys);
    int64_t n_tracks = 0;
    for (int64_t i = 0; i < N; ++i) if (parent[i] < 0) p.track[i] = n_tracks++;       // solve.cc:528-533
    for (int64_t i = 0; i < N; ++i) if (p.track[i] < 0) p.track[i] = p.track[uf_root(parent, (int32_t)i)];
    std::vector<int64_t> tsize(n_tracks, 0);
    for (int64_t i = 0; i < N; ++i) ++tsize[p.track[i]];
    p.stats.n_tracks = n_tracks;
    p.stats.max_track_size = *std::max_element(tsize.begin(), tsize.end());
    p.stats.tracks_ms = ms_since(t0);

    // ------------------------------------------------------------------ roots (solve.cc:552-582)
    t0 = clock::now();
    {
        std::vector<double> score(N, 0.0);
        for (int64_t m = 0; m < M; ++m) {            // per node: out-edges in insertion order
            const uint32_t a = g.m_node1[m], b = g.m_node2[m];
            if (p.track[a] == p.track[b]) { score[a] += (double)g.m_sim[m]; score[b] += (double)g.m_sim[m]; }
        }
        std::vector<int64_t> best(n_tracks, -1);
        for (int64_t i = 0; i < N; ++i) {            // max (score, node_idx): ties -> larger node idx
            int64_t &b = best[p.track[i]];
            if (b < 0 || score[i] >= score[b]) b = i;
        }
        for (int64_t t = 0; t < n_tracks; ++t) p.is_root[best[t]] = 1;
    }
    p.stats.roots_ms = ms_since(t0);

    // out-edge CSR: directed edge ids ascending per source node == insertion order
    std::vector<int64_t> out_off, out_eid;
    if (host_batch) build_out_csr(g, out_off, out_eid);        // (labels only: built on demand by the size cap)
    auto edge_dst = [&](int64_t e) -> uint32_t { return (e & 1) ? g.m_node1[e >> 1] : g.m_node2[e >> 1]; };
    auto edge_src = [&](int64_t e) -> uint32_t { return (e & 1) ? g.m_node2[e >> 1] : g.m_node1[e >> 1]; };

    // ------------------------------------------------------------------ components (solve.cc:252-373)
    t0 = clock::now();
    int64_t n_components = 0;
    if (component_override) {
        for (int64_t i = 0; i < N; ++i) {
            if (component_override[i] < 0) { set_error("negative component id in override"); return LFR_ERR_ARG; }
            p.comp[i] = component_override[i];
            n_components = std::max(n_components, p.comp[i] + 1);
        }
    } else {
        components_from_tracks(g, p.track, n_tracks, tsize, max_nodes, host_batch ? &out_off : nullptr, host_batch ? &out_eid : nullptr,
                               p.comp, n_components, p.stats.n_cut_components);
    }
    p.stats.n_components = n_components;
    p.stats.graph_cut_ms = ms_since(t0);
    p.host_batch = host_batch;
    if (host_batch && g.dev_disp1) {
        set_error("the flows of this graph live on the GPU: use lfr_problem_build_labels or lfr_problem_build_hip");
        return LFR_ERR_ARG;
    }
    if (!host_batch) {                    // labels only: lfr_batch_create assembles the batch on the GPU
        std::vector<int64_t> csz(n_components, 0);
        for (int64_t i = 0; i < N; ++i) ++csz[p.comp[i]];
        p.stats.max_component_size = *std::max_element(csz.begin(), csz.end());
        return LFR_OK;
    }

    // ------------------------------------------------------------------ assembly (solve.cc:594-606, 94-143)
    t0 = clock::now();
    std::vector<int64_t> comp_off(n_components + 1, 0);
    for (int64_t i = 0; i < N; ++i) ++comp_off[p.comp[i] + 1];
    for (int64_t c = 0; c < n_components; ++c) {
        p.stats.max_component_size = std::max(p.stats.max_component_size, comp_off[c + 1]);
        comp_off[c + 1] += comp_off[c];
    }
    std::vector<int64_t> comp_nodes(N);
    {
        std::vector<int64_t> cur(comp_off.begin(), comp_off.end() - 1);
        for (int64_t i = 0; i < N; ++i) comp_nodes[cur[p.comp[i]]++] = i;      // ascending node idx
    }
    // pass 1 (parallel over components): which nodes are variables, how many edges are kept
    std::vector<uint8_t> is_var(N, 0);
    struct Meta { int64_t comp; int64_t n_edges; int32_t n_var, n_nodes, cls, n_tracks; };
    std::vector<int64_t> seen_track(n_tracks, -1);
    const int T = host_threads();
    std::vector<int64_t> ccuts;                       // component ranges balanced by node count
    {
        int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(T, N / 20000));
        if (component_override) chunks = 1;       // a side-car may split a track over components: seen_track[] would be shared between threads
        ccuts.push_back(0);
        for (int c = 1; c < chunks; ++c) {
            const int64_t target = N * c / chunks;
            const int64_t pos = std::lower_bound(comp_off.begin(), comp_off.end(), target) - comp_off.begin();
            ccuts.push_back(std::max(ccuts.back(), std::min<int64_t>(pos, n_components)));
        }
        ccuts.push_back(n_components);
    }
    std::vector<std::vector<Meta>> metas_t(ccuts.size() - 1);
    std::vector<int> err_t(ccuts.size() - 1, 0);
    parallel_chunks(ccuts, [&](int chunk, int64_t clo, int64_t chi) {
        std::vector<Meta> &out = metas_t[chunk];
        for (int64_t c = clo; c < chi; ++c) {
            const int64_t lo = comp_off[c], hi = comp_off[c + 1];
            if (hi - lo <= 1) continue;                                           // solve.cc:619-622
            int32_t n_var = 0;
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t n = comp_nodes[k];
                bool opt = false;
                for (int64_t q = out_off[n]; q < out_off[n + 1] && !opt; ++q) {
                    const uint32_t d = edge_dst(out_eid[q]);
                    opt = p.track[n] == p.track[d] || p.comp[n] == p.comp[d];       // solve.cc:105,114,127
                }
                if (opt && !p.is_root[n]) { is_var[n] = 1; ++n_var; }               // solve.cc:133-141
            }
            if (n_var == 0) continue;                     // "No non-constant parameter blocks": nothing moves
            int64_t n_edges = 0;
            int32_t n_tr = 0;
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t n = comp_nodes[k];
                if (tsize[p.track[n]] >= 2 && seen_track[p.track[n]] != c) { seen_track[p.track[n]] = c; ++n_tr; }
                for (int64_t q = out_off[n]; q < out_off[n + 1]; ++q) {
                    const uint32_t d = edge_dst(out_eid[q]);
                    if (!(p.track[n] == p.track[d] || p.comp[n] == p.comp[d])) continue;
                    if (!is_var[n] && !is_var[d]) continue;       // both constant: not in the reduced program
                    ++n_edges;
                }
            }
            if (hi - lo > 32767) { err_t[chunk] = 1; return; }
            out.push_back(Meta{c, n_edges, n_var, (int32_t)(hi - lo), classify(2 * n_var, n_edges), n_tr});
        }
    });
    for (int e : err_t) if (e) { set_error("a component exceeds the 32767-node batch limit"); return LFR_ERR_UNSUPPORTED; }
    std::vector<Meta> metas;
    for (auto &v : metas_t) metas.insert(metas.end(), v.begin(), v.end());
    std::stable_sort(metas.begin(), metas.end(), [](const Meta &a, const Meta &b) {
        if (a.cls != b.cls) return a.cls < b.cls;
        if (a.n_edges != b.n_edges) return a.n_edges > b.n_edges;
        return a.n_var > b.n_var;
    });
    std::vector<int64_t> eoff(metas.size() + 1, 0), noff(metas.size() + 1, 0);
    for (size_t i = 0; i < metas.size(); ++i) { eoff[i + 1] = eoff[i] + metas[i].n_edges; noff[i + 1] = noff[i] + metas[i].n_nodes; }
    const int64_t total_edges = eoff.back(), total_nodes = noff.back();
    if (total_edges >= ((int64_t)1 << 32) || total_nodes >= ((int64_t)1 << 32)) { set_error("batch exceeds 2^32 edges/nodes"); return LFR_ERR_UNSUPPORTED; }
    p.descs.resize(metas.size()); p.desc_component.resize(metas.size()); p.desc_class.resize(metas.size());
    p.desc_tracks.resize(metas.size());
    p.edges.resize(total_edges); p.node_ids.resize(total_nodes);
    p.node_inc.assign(total_nodes, NodeInc{0, 0, 0, 0}); p.in_idx.resize(total_edges);
    std::vector<int32_t> local_of(N, -1);
    // pass 2 (parallel over the sorted components, balanced by edges): emit the batch
    std::vector<int64_t> dcuts;
    {
        int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(T, total_edges / 50000));
        dcuts.push_back(0);
        for (int c = 1; c < chunks; ++c) {
            const int64_t pos = std::lower_bound(eoff.begin(), eoff.end(), total_edges * c / chunks) - eoff.begin();
            dcuts.push_back(std::max(dcuts.back(), std::min<int64_t>(pos, (int64_t)metas.size())));
        }
        dcuts.push_back((int64_t)metas.size());
    }
    std::atomic<int> pair_err{0};
    parallel_chunks(dcuts, [&](int, int64_t dlo, int64_t dhi) {
        std::vector<int64_t> kept;
        for (int64_t di = dlo; di < dhi; ++di) {
            const Meta &mt = metas[di];
            const int64_t lo = comp_off[mt.comp], hi = comp_off[mt.comp + 1];
            const int64_t no = noff[di], eo0 = eoff[di];
            int64_t eo = eo0;
            int32_t nv = 0, nc2 = mt.n_var;
            for (int64_t k = lo; k < hi; ++k) {           // variable nodes first, then constants
                const int64_t n = comp_nodes[k];
                const int32_t l = is_var[n] ? nv++ : nc2++;
                local_of[n] = l;
                p.node_ids[no + l] = (uint32_t)n;
            }
            CompDesc &d = p.descs[di];
            d.edge_off = (uint32_t)eo0; d.n_edges = (uint32_t)mt.n_edges; d.node_off = (uint32_t)no;
            d.n_nodes = (uint16_t)mt.n_nodes; d.n_var = (uint16_t)mt.n_var;
            p.desc_component[di] = mt.comp; p.desc_class[di] = mt.cls; p.desc_tracks[di] = mt.n_tracks;
            // Workgroup classes: residual-block order of solve.cc:98-102 (by source node: the owner-computes
            // kernels walk out-edge runs).  Packed classes: edge-id order, so that the two directions of a
            // match (ids 2m, 2m+1; both are kept or both dropped) sit in neighbouring records / lanes.
            const bool by_edge_id = mt.cls < KC_BLOCK;
            kept.clear();
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t n = comp_nodes[k];
                p.node_inc[no + local_of[n]].out_begin = (uint32_t)(eo - eo0 + (int64_t)kept.size());
                for (int64_t q = out_off[n]; q < out_off[n + 1]; ++q) {
                    const int64_t e = out_eid[q];
                    const uint32_t dn = edge_dst(e);
                    if (!(p.track[n] == p.track[dn] || p.comp[n] == p.comp[dn])) continue;
                    if (!is_var[n] && !is_var[dn]) continue;
                    kept.push_back(e);
                }
            }
            if (by_edge_id) {
                std::sort(kept.begin(), kept.end());
                // the packed kernel's pair exchange relies on it: both directions of every match are present
                bool paired = kept.size() % 2 == 0;
                for (size_t q = 0; paired && q < kept.size(); q += 2) paired = (kept[q] ^ 1) == kept[q + 1];
                if (!paired) { pair_err.store(1); return; }
            }
            for (const int64_t e : kept) {
                const uint32_t n = edge_src(e), dn = edge_dst(e);
                const int kind = p.track[n] == p.track[dn] ? 0 : 1;
                EdgeRec &r = p.edges[eo++];
                const float *fl = ((e & 1) ? g.m_disp1.data() : g.m_disp2.data()) + 18 * (e >> 1);
                memcpy(r.flow, fl, sizeof r.flow);
                r.sim = g.m_sim[e >> 1];
                r.src = (uint16_t)local_of[n];
                r.dst_kind = (uint16_t)(local_of[dn] | (kind << 15));
                ++p.node_inc[no + local_of[n]].out_count;
                ++p.node_inc[no + local_of[dn]].in_count;
            }
            {   // in-edge lists: counting sort of the component's edges by destination (stable)
                uint32_t acc = 0;
                for (int32_t l = 0; l < mt.n_nodes; ++l) { p.node_inc[no + l].in_begin = acc; acc += p.node_inc[no + l].in_count; p.node_inc[no + l].in_count = 0; }
                for (int64_t e = eo0; e < eo; ++e) {
                    NodeInc &ni = p.node_inc[no + (p.edges[e].dst_kind & 0x7fff)];
                    p.in_idx[eo0 + ni.in_begin + ni.in_count++] = (uint32_t)(e - eo0);
                }
            }
        }
    });
    if (pair_err.load()) { set_error("internal: a kept edge without its opposite direction"); return LFR_ERR_UNSUPPORTED; }
    for (const Meta &mt : metas) p.stats.n_solved_tracks += mt.n_tracks;
    p.stats.n_solved_components = (int64_t)metas.size();
    p.stats.n_solved_edges = total_edges;
    p.stats.n_solved_nodes = total_nodes;
    p.stats.assemble_ms = ms_since(t0);
    return LFR_OK;
}

std::vector<int32_t> assign_shards(const Problem &p, int world) {
    std::vector<int32_t> shard(p.descs.size(), 0);
    if (world <= 1) return shard;
    for (size_t i = 0; i < shard.size(); ++i) {          // snake_shard() of lfr_assemble.hpp (host-only file: restated)
        const size_t round = i / (size_t)world, pos = i % (size_t)world;
        shard[i] = (int32_t)((round & 1) ? (size_t)world - 1 - pos : pos);
    }
    return shard;
}

}  // namespace lfr

using namespace lfr;

extern "C" {

static int problem_build(const lfr_graph *g, int64_t max_nodes_in_component, const int64_t *component_override,
                         bool host_batch, lfr_problem **out) {
    if (!g || !out) { set_error("bad argument"); return LFR_ERR_ARG; }
    lfr_problem *h = new lfr_problem();
    const int rc = build_problem(g->g, max_nodes_in_component, component_override, h->p, host_batch);
    if (rc != LFR_OK) { delete h; *out = nullptr; return rc; }
    *out = h;
    return LFR_OK;
}

int lfr_problem_build(const lfr_graph *g, int64_t max_nodes_in_component, const int64_t *component_override,
                      lfr_problem **out) {
    return problem_build(g, max_nodes_in_component, component_override, true, out);
}

int lfr_problem_build_labels(const lfr_graph *g, int64_t max_nodes_in_component, const int64_t *component_override,
                             lfr_problem **out) {
    return problem_build(g, max_nodes_in_component, component_override, false, out);
}

void lfr_problem_free(lfr_problem *p) { delete p; }

int64_t lfr_bisect_graph(int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b, const int32_t *weights, int32_t *nodes,
                         int32_t *part) {
    if (n_edges < 0 || (n_edges > 0 && (!edge_a || !edge_b || !weights))) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::vector<std::pair<int, int>> e((size_t)n_edges);
    std::vector<int> w((size_t)n_edges);
    for (int64_t k = 0; k < n_edges; ++k) { e[k] = {edge_a[k], edge_b[k]}; w[k] = weights[k]; }
    std::unordered_map<int, int> split;
    bisect_graph(e, w, split);
    std::vector<int> keys;
    for (auto &it : split) keys.push_back(it.first);
    std::sort(keys.begin(), keys.end());
    for (size_t k = 0; k < keys.size(); ++k) { if (nodes) nodes[k] = keys[k]; if (part) part[k] = split[keys[k]]; }
    return (int64_t)keys.size();
}

int64_t lfr_debug_recursive_cut(int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b, const int32_t *weights,
                                int64_t n_node_weights, const int64_t *node_weights, int64_t max_weight, int32_t *nodes, int32_t *subset) {
    if (n_edges < 0 || n_node_weights < 0 || (n_edges > 0 && (!edge_a || !edge_b || !weights || !node_weights))) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::vector<std::pair<int, int>> e((size_t)n_edges);
    std::vector<int> w(weights, weights + n_edges);
    for (int64_t k = 0; k < n_edges; ++k) {
        if (edge_a[k] < 0 || edge_b[k] < 0 || edge_a[k] >= n_node_weights || edge_b[k] >= n_node_weights) { set_error("node id without a weight"); return LFR_ERR_ARG; }
        e[k] = {edge_a[k], edge_b[k]};
    }
    const std::vector<int64_t> nw(node_weights, node_weights + n_node_weights);
    const auto split = recursive_cut(e, w, nw, max_weight);
    std::vector<int> keys;
    keys.reserve(split.size());
    for (auto &it : split) keys.push_back(it.first);
    std::sort(keys.begin(), keys.end());
    for (size_t k = 0; k < keys.size(); ++k) { if (nodes) nodes[k] = keys[k]; if (subset) subset[k] = split.at(keys[k]); }
    return (int64_t)keys.size();
}

int lfr_problem_get_stats(const lfr_problem *p, lfr_problem_stats *stats) {
    if (!p || !stats) return LFR_ERR_ARG;
    *stats = p->p.stats;
    return LFR_OK;
}

int64_t lfr_problem_shard_components(const lfr_problem *p, int shard_rank, int shard_world, int64_t *components,
                                     int64_t *n_edges) {
    if (!p || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) return LFR_ERR_ARG;
    if (!p->p.host_batch) { set_error("lfr_problem_shard_components needs a host-assembled problem (lfr_problem_build)"); return LFR_ERR_ARG; }
    const std::vector<int32_t> shard = assign_shards(p->p, shard_world);
    int64_t n = 0;
    for (size_t i = 0; i < shard.size(); ++i)
        if (shard[i] == shard_rank) {
            if (components) components[n] = p->p.desc_component[i];
            if (n_edges) n_edges[n] = p->p.descs[i].n_edges;
            ++n;
        }
    return n;
}

int lfr_problem_get_labels(const lfr_problem *p, int64_t *track, uint8_t *is_root, int64_t *component) {
    if (!p) return LFR_ERR_ARG;
    const int rc = p->p.ensure_host_labels();       // after the device graph stage the labels are still in HBM
    if (rc != LFR_OK) return rc;
    const size_t n = p->p.track.size();
    if (track && n) memcpy(track, p->p.track.data(), sizeof(int64_t) * n);
    if (is_root && n) memcpy(is_root, p->p.is_root.data(), n);
    if (component && n) memcpy(component, p->p.comp.data(), sizeof(int64_t) * n);
    return LFR_OK;
}

}  // extern "C"

// Elimination-tree plan for the components whose normal matrices do not fit LDS (kernel class KC_GLOBAL, > 192 rows).
//
// The reference asks Ceres for SPARSE_NORMAL_CHOLESKY (solve.cc:147): a supernodal factorization along an elimination tree.  The
// normal matrix of a component is as sparse as its match graph - one 2x2 block per matched node pair - and a component at the size
// cap (#images nodes, solve.cc:586) is a few hundred short tracks (small cliques / lattices) hanging together through inter-track
// matches.  Round 3 kept such a matrix inside the envelope of a postorder and factored it as ONE chain of n/16 dependent panels on
// one workgroup.  This plan exposes the parallelism the structure has:
//
//   * ORDER: nested dissection of the tracks' meta graph (tracks = connected pieces of the intra-track edges; separators = the
//     lighter of a middle level of a breadth-first level structure and the centroid of its spanning tree), separators last.  A chain
//     of 230 tracks becomes a tree of depth ~8 instead of a path of 230.
//   * BLOCKS: the node sequence is cut into blocks of <= 8 nodes = 16 rows (one fp64 MFMA tile) that never straddle independent
//     subtrees: a block holds whole tracks of one separator / leaf, or whole small subtrees of siblings (and their parent) - merges
//     that add no coupling between blocks.  Position of a node = 8 * block + slot; the slots a block does not use are padding (inert:
//     never a pivot, zero everywhere).
//   * STRUCTURE: symbolic factorization at block level (struct(J) = blocks I > J with L(I, J) != 0, parent(J) = min struct(J));
//     tiles are stored by column (the diagonal tile first, then the tiles below it).
//   * SCHEDULE: columns by LEVEL of the block elimination tree (leaves = level 0).  All columns of a level are independent.  The
//     factorization is LEFT-LOOKING so that every tile is written by one wave in a fixed order (bitwise reproducible, no atomics):
//     tile (I, J) -= sum over k in rows(I) & rows(J) of U(I, k) D_k^-1 U(J, k)^T, the list of (tile(I, k), tile(J, k), k) per tile
//     precomputed here.  The right-hand side is a vector that rides through the diagonal tile's task.
//   * SWEEP ITEMS: (node, neighbour) pairs in matrix order, each with the records between the two nodes in both directions: one lane
//     evaluates them and STORES the pair's 2x2 cross block (no atomics, duplicates summed in record order); the node's diagonal block
//     and gradient are summed over its items in a fixed order.
//
// Everything is a deterministic function of the component's record list.  The blob written here is what the kernel reads
// (solve_tree_component in lfr_solve.hip); tests/test_tree_plan.py executes it on the CPU against a dense solve.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "lfr_internal.hpp"

namespace lfr {

namespace {

struct Csr {
    std::vector<int32_t> off, adj;
    std::vector<uint8_t> kind;
};

// adjacency among the variable nodes (both directions of a match are records of the component: every undirected pair is seen twice)
Csr build_adj(int nv, int64_t ne, const uint32_t *w) {
    Csr g;
    g.off.assign(nv + 1, 0);
    for (int64_t e = 0; e < ne; ++e) {
        const int s = (int)(w[e] & 0xffffu), d = (int)((w[e] >> 16) & 0x7fffu);
        if (s < nv && d < nv && s != d) ++g.off[s + 1];
    }
    for (int i = 0; i < nv; ++i) g.off[i + 1] += g.off[i];
    g.adj.resize(g.off[nv]); g.kind.resize(g.off[nv]);
    std::vector<int32_t> cur(g.off.begin(), g.off.end() - 1);
    for (int64_t e = 0; e < ne; ++e) {
        const int s = (int)(w[e] & 0xffffu), d = (int)((w[e] >> 16) & 0x7fffu);
        if (s < nv && d < nv && s != d) { g.adj[cur[s]] = d; g.kind[cur[s]] = (uint8_t)(w[e] >> 31); ++cur[s]; }
    }
    return g;
}

int find_root(std::vector<int32_t> &p, int x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------
// nested dissection of the meta graph: a forest of SEGMENTS (a separator's or a leaf's tracks, in order), children = the pieces the
// separator leaves.  Elimination order = children before parents.
// ---------------------------------------------------------------------------------------------------------------------
struct Meta {
    int T = 0;
    std::vector<int32_t> off, adj;           // deduplicated, sorted
    std::vector<int32_t> weight;             // variable nodes of the track
};
struct Segment {
    std::vector<int32_t> tracks;             // the segment's units (variable nodes), in elimination order
    int parent = -1;
};
// Pieces of up to this many nodes are not dissected further: their blocks form a chain in breadth-first (banded) order.  The
// factorization deals a level's columns to the workgroup's waves, so below ~4 blocks per leaf more levels buy nothing
// (scripts/tree_plan_stats.py: column rounds of the cap-sized sparse workload).
#ifndef LFR_TREE_LEAF_NODES
#define LFR_TREE_LEAF_NODES 32
#endif
constexpr int kLeafNodes = LFR_TREE_LEAF_NODES;

struct Dissector {
    const Meta &m;
    std::vector<Segment> &segs;
    std::vector<int32_t> in_set, level, tparent, order, sub;      // scratch indexed by track
    int stamp = 0;
    explicit Dissector(const Meta &mm, std::vector<Segment> &s) : m(mm), segs(s), in_set(mm.T, 0), level(mm.T, 0), tparent(mm.T, -1), sub(mm.T, 0) {}

    // breadth-first search inside the marked set (in_set[t] == tag) from `start`; fills order / level / tparent
    void bfs(int start, int tag) {
        order.clear();
        order.push_back(start); level[start] = 0; tparent[start] = -1;
        in_set[start] = -tag;                                     // visited = negated tag
        for (size_t h = 0; h < order.size(); ++h) {
            const int t = order[h];
            for (int k = m.off[t]; k < m.off[t + 1]; ++k) {
                const int u = m.adj[k];
                if (in_set[u] != tag) continue;
                in_set[u] = -tag; level[u] = level[t] + 1; tparent[u] = t; order.push_back(u);
            }
        }
        for (int t : order) in_set[t] = tag;
    }

    void run(std::vector<int32_t> all) {
        struct Job { std::vector<int32_t> set; int parent; };
        std::vector<Job> jobs;
        // connected pieces of the whole component (the variable nodes of a component can fall apart: constants link them)
        split(all, -1, jobs);
        while (!jobs.empty()) {
            Job job = std::move(jobs.back());
            jobs.pop_back();
            dissect(job.set, job.parent, jobs);
        }
    }

    // pushes the connected pieces of `set` as jobs (largest last = processed first; the order only affects block numbering)
    template <typename Jobs>
    void split(const std::vector<int32_t> &set, int parent, Jobs &jobs) {
        const int tag = ++stamp;
        for (int t : set) in_set[t] = tag;
        const int done = ++stamp;
        for (int t : set) {
            if (in_set[t] != tag) continue;
            bfs(t, tag);
            std::vector<int32_t> piece(order.begin(), order.end());
            for (int u : piece) in_set[u] = done;
            jobs.push_back({std::move(piece), parent});
        }
    }

    template <typename Jobs>
    void dissect(const std::vector<int32_t> &set, int parent, Jobs &jobs) {
        int64_t W = 0;
        for (int t : set) W += m.weight[t];
        const int seg_id = (int)segs.size();
        if (set.size() == 1 || W <= kLeafNodes) {                 // a leaf: a chain of <= kLeafNodes / 8 blocks in breadth-first order
            if (set.size() <= 8) { segs.push_back({set, parent}); return; }
            const int tag = ++stamp;
            for (int t : set) in_set[t] = tag;
            int start = set[0];
            { int best = INT32_MAX; for (int t : set) { int d = 0; for (int k = m.off[t]; k < m.off[t + 1]; ++k) d += in_set[m.adj[k]] == tag ? 1 : 0; if (d < best) { best = d; start = t; } } }
            bfs(start, tag);
            bfs(order.back(), tag);
            segs.push_back({std::vector<int32_t>(order.begin(), order.end()), parent});
            return;
        }
        const int tag = ++stamp;
        for (int t : set) in_set[t] = tag;
        // pseudo-peripheral start: the lightest-degree track, then the far end of a search from it
        int start = set[0];
        auto deg_in = [&](int t) { int d = 0; for (int k = m.off[t]; k < m.off[t + 1]; ++k) d += in_set[m.adj[k]] == tag ? 1 : 0; return d; };
        { int best = deg_in(start); for (int t : set) { const int d = deg_in(t); if (d < best || (d == best && t < start)) { best = d; start = t; } } }
        bfs(start, tag);
        bfs(order.back(), tag);
        const int L = level[order.back()] + 1;
        // (A) a middle level of the level structure
        std::vector<int64_t> lw(L, 0);
        for (int t : order) lw[level[t]] += m.weight[t];
        int best_l = -1;
        int64_t score_a = INT64_MAX;
        {
            int64_t before = 0;
            for (int l = 0; l < L; ++l) {
                const int64_t after = W - before - lw[l];
                if (l >= 1 && l + 1 < L) {
                    const int64_t sc = std::max(before, after) + 2 * lw[l];
                    if (sc < score_a) { score_a = sc; best_l = l; }
                }
                before += lw[l];
            }
        }
        // (B) the centroid of the search tree, judged by the pieces its removal REALLY leaves (non-tree edges can tie them together)
        for (int t : order) sub[t] = m.weight[t];
        for (size_t h = order.size(); h-- > 1;) sub[tparent[order[h]]] += sub[order[h]];
        int c = order[0];
        for (;;) {
            int heavy = -1;
            for (int k = m.off[c]; k < m.off[c + 1]; ++k) {
                const int u = m.adj[k];
                if (in_set[u] == tag && tparent[u] == c && 2 * (int64_t)sub[u] > W) { heavy = u; break; }
            }
            if (heavy < 0) break;
            c = heavy;
        }
        std::vector<int32_t> tree_order(order.begin(), order.end());     // (bfs below overwrites `order`)
        std::vector<int32_t> tree_level(set.size());
        for (size_t i = 0; i < tree_order.size(); ++i) tree_level[i] = level[tree_order[i]];
        int64_t score_b;
        {
            const int tb = ++stamp;
            for (int t : set) in_set[t] = tb;
            in_set[c] = 0;
            int64_t largest = 0;
            const int seen = ++stamp;
            for (int t : set) {
                if (in_set[t] != tb) continue;
                bfs(t, tb);
                int64_t w = 0;
                for (int u : order) { w += m.weight[u]; in_set[u] = seen; }
                largest = std::max(largest, w);
            }
            score_b = largest + 2 * (int64_t)m.weight[c];
        }
        std::vector<int32_t> sep, rest;
        const bool use_a = best_l >= 0 && score_a <= score_b;
        const int64_t sep_weight = use_a ? lw[best_l] : (int64_t)m.weight[c];
        const int64_t largest_piece = (use_a ? score_a : score_b) - 2 * sep_weight;
        if ((double)largest_piece > 0.92 * (double)W && set.size() > 2) {
            // no separator worth its name (a dense meta graph): one segment in reverse search order - the far end first
            std::vector<int32_t> seq(tree_order.rbegin(), tree_order.rend());
            segs.push_back({std::move(seq), parent});
            return;
        }
        if (use_a) {
            for (size_t i = 0; i < tree_order.size(); ++i) (tree_level[i] == best_l ? sep : rest).push_back(tree_order[i]);
        } else {
            for (int t : tree_order) (t == c ? sep : rest).push_back(t);
        }
        segs.push_back({std::move(sep), parent});
        split(rest, seg_id, jobs);
    }
};

template <typename T>
void put(std::vector<uint32_t> &blob, uint32_t hdr_slot, const std::vector<T> &v) {
    static_assert(sizeof(T) == 4, "u32 arrays only");
    while (blob.size() % 4) blob.push_back(0);                   // (the kernel reads some of the arrays in 16-byte words)
    blob[hdr_slot] = (uint32_t)blob.size();
    const size_t o = blob.size();
    blob.resize(o + v.size());
    if (!v.empty()) memcpy(&blob[o], v.data(), 4 * v.size());
}

}  // namespace

void tree_plan(int n_var, int64_t n_edges, const uint32_t *words, TreePlan &out) {
    static const bool plan_timing = getenv("LFR_PLAN_TIMING") != nullptr;            // laps of one plan to stderr (where does batch creation go?)
    auto lap_t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!plan_timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "lfr plan (%d nodes, %lld records): %8.3f ms  %s\n", n_var, (long long)n_edges, std::chrono::duration<double, std::milli>(t - lap_t0).count(), what);
        lap_t0 = t;
    };
    const Csr g = build_adj(n_var, n_edges, words);
    lap("adjacency");
    // ---- nested dissection of the variable nodes' graph -> segments ----
    // (The unit is the NODE, not the track: the tracks of the graph stage can be long - chains of short feature tracks joined by wrong
    // matches that the image-disjoint rule accepted - and a lattice of 30 nodes dissects as well as a chain of tracks does.  A short
    // all-pairs track is a clique: it never separates and stays together in its leaf.)
    Meta meta;
    meta.T = n_var;
    meta.weight.assign(n_var, 1);
    meta.off.assign(n_var + 1, 0);
    {
        std::vector<int32_t> row;
        meta.adj.reserve(g.adj.size());
        for (int v = 0; v < n_var; ++v) {
            row.assign(g.adj.begin() + g.off[v], g.adj.begin() + g.off[v + 1]);
            std::sort(row.begin(), row.end());
            row.erase(std::unique(row.begin(), row.end()), row.end());
            meta.adj.insert(meta.adj.end(), row.begin(), row.end());
            meta.off[v + 1] = (int32_t)meta.adj.size();
        }
    }
    int T = 0;                                                   // tracks (connected pieces of the intra-track edges): reported only
    {
        std::vector<int32_t> uf(n_var);
        std::iota(uf.begin(), uf.end(), 0);
        for (int v = 0; v < n_var; ++v)
            for (int k = g.off[v]; k < g.off[v + 1]; ++k)
                if (g.kind[k] == 0) { const int a = find_root(uf, v), b = find_root(uf, g.adj[k]); if (a != b) uf[std::max(a, b)] = std::min(a, b); }
        for (int v = 0; v < n_var; ++v) T += find_root(uf, v) == v ? 1 : 0;
    }
    std::vector<Segment> segs;
    {
        Dissector d(meta, segs);
        std::vector<int32_t> all(n_var);
        std::iota(all.begin(), all.end(), 0);
        d.run(std::move(all));
    }
    lap("meta graph + nested dissection");
    const int S = (int)segs.size();
    // ---- blocks: children before parents (segments were created parent first: reverse index order) ----
    std::vector<int64_t> sub_weight(S, 0);
    std::vector<std::vector<int32_t>> unit(S);                 // nodes of a collapsed subtree (<= 8), handed to the parent
    std::vector<std::vector<int32_t>> pending(S);              // collapsed children of a segment, in the order they finish
    std::vector<std::vector<int32_t>> blocks;
    std::vector<int> last_block(S, -1);                        // the last block a (non-collapsed) segment emitted: the top of its chain
    std::vector<std::vector<int>> kids(S);                     // non-collapsed children
    for (int s = S - 1; s >= 0; --s) {
        sub_weight[s] += (int64_t)segs[s].tracks.size();
        const int par = segs[s].parent;
        if (par >= 0) sub_weight[par] += sub_weight[s];        // (children have larger indices: the parent's sum is complete when its turn comes)
        if (par >= 0 && sub_weight[s] <= 8) {                  // the whole subtree is one small unit
            std::vector<int32_t> u;
            for (int c : pending[s]) u.insert(u.end(), unit[c].begin(), unit[c].end());
            u.insert(u.end(), segs[s].tracks.begin(), segs[s].tracks.end());
            unit[s] = std::move(u);
            pending[par].push_back(s);
            continue;
        }
        std::vector<int32_t> cur;
        auto close = [&] { if (!cur.empty()) { blocks.push_back(cur); cur.clear(); } };
        for (int c : pending[s]) {                             // small subtrees of siblings share blocks
            if (cur.size() + unit[c].size() > 8) close();
            cur.insert(cur.end(), unit[c].begin(), unit[c].end());
        }
        // A small separator would be a column of its own with one to three pivots (every column costs the factorization the same
        // latencies).  If it fits, it takes in the TOP blocks of its children instead - their last, often half-empty blocks move up
        // into the separator's block (amalgamation of a parent with the tops of its children: no new coupling between blocks, the
        // children's other blocks still come first).
        if (cur.size() + segs[s].tracks.size() <= 8 && !kids[s].empty()) {
            size_t room = 8 - cur.size() - segs[s].tracks.size();
            std::vector<int> cand;
            for (int c : kids[s]) if (last_block[c] >= 0 && !blocks[last_block[c]].empty() && blocks[last_block[c]].size() <= room) cand.push_back(c);
            std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return blocks[last_block[a]].size() > blocks[last_block[b]].size(); });
            for (int c : cand) {
                std::vector<int32_t> &top = blocks[last_block[c]];
                if (top.size() > room) continue;
                room -= top.size();
                cur.insert(cur.end(), top.begin(), top.end());
                top.clear();                                   // (empty blocks are dropped below)
            }
        }
        for (int v : segs[s].tracks) {                         // ... and the last of them may take in the segment's first nodes
            if (cur.size() == 8) close();
            cur.push_back(v);
        }
        close();
        last_block[s] = (int)blocks.size() - 1;
        if (par >= 0) kids[par].push_back(s);
    }
    blocks.erase(std::remove_if(blocks.begin(), blocks.end(), [](const std::vector<int32_t> &b) { return b.empty(); }), blocks.end());
    const int NB = (int)blocks.size();
    std::vector<int32_t> blk(n_var, -1), slot(n_var, 0);
    for (int b = 0; b < NB; ++b) for (size_t i = 0; i < blocks[b].size(); ++i) { blk[blocks[b][i]] = b; slot[blocks[b][i]] = (int32_t)i; }
    lap("blocks");
    // ---- block-level symbolic factorization ----
    std::vector<std::vector<int32_t>> st(NB);                   // struct(J), sorted
    {
        std::vector<std::vector<int32_t>> higher(NB), children(NB);
        for (int v = 0; v < n_var; ++v)
            for (int k = g.off[v]; k < g.off[v + 1]; ++k) { const int a = blk[v], b = blk[g.adj[k]]; if (a < b) higher[a].push_back(b); }
        std::vector<int32_t> tmp;
        for (int J = 0; J < NB; ++J) {
            std::vector<int32_t> &s = higher[J];
            std::sort(s.begin(), s.end());
            s.erase(std::unique(s.begin(), s.end()), s.end());
            for (int c : children[J]) {
                tmp.clear();
                const std::vector<int32_t> &sc = st[c];
                std::set_union(s.begin(), s.end(), std::upper_bound(sc.begin(), sc.end(), J), sc.end(), std::back_inserter(tmp));
                s.swap(tmp);
            }
            st[J] = std::move(s);
            if (!st[J].empty()) children[st[J][0]].push_back(J);
        }
    }
    std::vector<uint32_t> colptr(NB + 1, 0);
    for (int J = 0; J < NB; ++J) colptr[J + 1] = colptr[J] + 1u + (uint32_t)st[J].size();
    const uint32_t n_tiles = colptr[NB];
    std::vector<uint32_t> rowsof(n_tiles);
    for (int J = 0; J < NB; ++J) { rowsof[colptr[J]] = (uint32_t)J; for (size_t i = 0; i < st[J].size(); ++i) rowsof[colptr[J] + 1 + i] = (uint32_t)st[J][i]; }
    // levels of the block elimination tree
    std::vector<int32_t> lvl(NB, 0);
    int n_levels = 0;
    for (int J = 0; J < NB; ++J) { if (!st[J].empty()) lvl[st[J][0]] = std::max(lvl[st[J][0]], lvl[J] + 1); n_levels = std::max(n_levels, lvl[J] + 1); }
    std::vector<uint32_t> level_ptr(n_levels + 1, 0), level_cols(NB);
    for (int J = 0; J < NB; ++J) ++level_ptr[lvl[J] + 1];
    for (int l = 0; l < n_levels; ++l) level_ptr[l + 1] += level_ptr[l];
    {
        // inside a level: the columns with the most tiles first (the longest tasks start first)
        std::vector<uint32_t> cur(level_ptr.begin(), level_ptr.end() - 1);
        std::vector<int32_t> by(NB);
        std::iota(by.begin(), by.end(), 0);
        std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return st[a].size() > st[b].size(); });
        for (int J : by) level_cols[cur[lvl[J]]++] = (uint32_t)J;
    }
    std::vector<uint32_t> nreal(NB);
    for (int b = 0; b < NB; ++b) nreal[b] = (uint32_t)blocks[b].size();
    // rows of every block row: rowlist[I] = columns k < I with a tile (I, k), ascending, with the tile's id
    std::vector<std::vector<std::pair<int32_t, uint32_t>>> rowlist(NB);
    for (int k = 0; k < NB; ++k) for (size_t i = 0; i < st[k].size(); ++i) rowlist[st[k][i]].push_back({k, colptr[k] + 1u + (uint32_t)i});
    lap("symbolic factorization, levels");
    // ---- rows the column task carries through the diagonal tile's elimination; the rest are "extra row" tasks ----
    // lanes 16 = right-hand side, 17 .. 63 = rows 0-15 of the first two tiles below the diagonal and rows 0-14 of the third
    std::vector<uint32_t> ncarry(NB), x_ptr(n_levels + 1, 0), x_tasks;
    for (int l = 0; l < n_levels; ++l) {
        for (uint32_t q = level_ptr[l]; q < level_ptr[l + 1]; ++q) {
            const int J = (int)level_cols[q];
            const int ns = (int)st[J].size();
            int nc = std::min(ns, 2);
            if (ns >= 3 && nreal[st[J][2]] <= 7) nc = 3;
            ncarry[J] = (uint32_t)nc;
            for (int i = nc; i < ns; i += 4) { x_tasks.push_back((uint32_t)J); x_tasks.push_back((uint32_t)i); x_tasks.push_back((uint32_t)std::min(4, ns - i)); }
        }
        x_ptr[l + 1] = (uint32_t)(x_tasks.size() / 3);
    }
    // ---- left-looking update lists ----
    // The column task of J applies the updates of its diagonal tile, of the right-hand side and of the tiles it carries itself:
    // col_upd[J] = per column k of row J (ascending) the quintuple (k, tile (J, k), tile (I_0, k), tile (I_1, k), tile (I_2, k)) with
    // I_i the carried rows (none: 0xffffffff).  The other tiles of a column (rows beyond the carried ones) are separate tile tasks:
    // (target tile, first update, end, J) over triples (tile (I, k), tile (J, k), k).
    constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> col_upd_ptr(NB + 1, 0), col_upd, p1_first(NB, 0);
    std::vector<uint32_t> upd, p1_tasks, p1_ptr(n_levels + 1, 0);
    uint64_t n_upd = 0;
    {
        std::vector<std::vector<uint32_t>> col_entries(NB);
        std::vector<std::vector<uint32_t>> per_target;
        for (int l = 0; l < n_levels; ++l) {
            for (uint32_t q = level_ptr[l]; q < level_ptr[l + 1]; ++q) {
                const int J = (int)level_cols[q];
                const std::vector<int32_t> &sJ = st[J];
                const size_t nc = ncarry[J];
                per_target.assign(sJ.size(), {});
                std::vector<uint32_t> &ce = col_entries[J];
                for (auto &rk : rowlist[J]) {
                    const int k = rk.first;
                    const uint32_t tJk = rk.second;
                    const std::vector<int32_t> &sk = st[k];
                    uint32_t carried[3] = {kNone, kNone, kNone};
                    // the rows of column k below J are rows of column J as well (fill closure): walk both lists
                    size_t i = std::lower_bound(sk.begin(), sk.end(), J) - sk.begin() + 1, j = 0;
                    for (; i < sk.size(); ++i) {
                        const int I = sk[i];
                        const uint32_t tIk = colptr[k] + 1u + (uint32_t)i;
                        while (j < sJ.size() && sJ[j] < I) ++j;              // (sJ[j] == I)
                        if (j < nc) carried[j] = tIk;
                        else { per_target[j].push_back(tIk); per_target[j].push_back(tJk); per_target[j].push_back((uint32_t)k); }
                        ++n_upd;
                    }
                    ce.push_back((uint32_t)k); ce.push_back(tJk); ce.push_back(carried[0]); ce.push_back(carried[1]); ce.push_back(carried[2]);
                    ++n_upd;
                }
                p1_first[J] = (uint32_t)(p1_tasks.size() / 4);
                for (size_t t = nc; t < per_target.size(); ++t) {      // (also without updates: the task moves the tile from A to the factor)
                    const uint32_t b0 = (uint32_t)(upd.size() / 3);
                    upd.insert(upd.end(), per_target[t].begin(), per_target[t].end());
                    p1_tasks.push_back(colptr[J] + 1u + (uint32_t)t); p1_tasks.push_back(b0); p1_tasks.push_back((uint32_t)(upd.size() / 3));
                    p1_tasks.push_back((uint32_t)J);
                }
            }
            p1_ptr[l + 1] = (uint32_t)(p1_tasks.size() / 4);
        }
        for (int J = 0; J < NB; ++J) {
            col_upd_ptr[J] = (uint32_t)(col_upd.size() / 5);
            col_upd.insert(col_upd.end(), col_entries[J].begin(), col_entries[J].end());
        }
        col_upd_ptr[NB] = (uint32_t)(col_upd.size() / 5);
    }
    lap("update lists");
    // ---- column descriptors in execution order (level by level): everything a column task needs to start its loads comes with ONE
    //      pair of scalar loads - {J, diagonal tile, carried tiles, pivots, update entries, first further entry, entry 0 (5 words), entry 1,
    //      tiles below the diagonal, rows of the first four of them} - instead of a chain of dependent lookups ----
    std::vector<uint32_t> n_children(NB, 0);
    for (int J = 0; J < NB; ++J) if (!st[J].empty()) ++n_children[st[J][0]];
    std::vector<uint32_t> col_desc(32 * (size_t)NB, 0);
    for (int q = 0; q < NB; ++q) {
        const int J = (int)level_cols[q];
        uint32_t *dsc = &col_desc[32 * (size_t)q];
        const uint32_t e0 = col_upd_ptr[J], ne = col_upd_ptr[J + 1] - e0;
        dsc[0] = (uint32_t)J; dsc[1] = colptr[J]; dsc[2] = ncarry[J]; dsc[3] = 2u * nreal[J]; dsc[4] = ne; dsc[5] = e0 + 2u;
        for (uint32_t i = 0; i < 2; ++i)
            for (int w5 = 0; w5 < 5; ++w5) dsc[6 + 5 * i + w5] = i < ne ? col_upd[5 * (size_t)(e0 + i) + w5] : (w5 == 0 ? (uint32_t)J : w5 == 1 ? colptr[J] : kNone);
        dsc[16] = (uint32_t)st[J].size();
        for (size_t i = 0; i < 4; ++i) dsc[17 + i] = i < st[J].size() ? (uint32_t)st[J][i] : (uint32_t)J;
        dsc[21] = st[J].empty() ? kNone : (uint32_t)st[J][0];          // parent in the elimination tree
        dsc[22] = n_children[J];
        dsc[23] = p1_first[J];                                         // the tile tasks of the tiles this column does not carry: p1_first .. + (tiles below - carried)
    }
    lap("column descriptors");
    // ---- sweep items ----
    const uint32_t n_pad = 16u * (uint32_t)NB;
    std::vector<uint32_t> ipos(8 * (size_t)NB, 0xffffffffu), node_items(8 * (size_t)NB + 1, 0), items, item_edges;
    for (int v = 0; v < n_var; ++v) ipos[8 * (size_t)blk[v] + slot[v]] = (uint32_t)v;
    {
        struct Inc { int32_t nbr; uint32_t rec; uint8_t dir; };
        std::vector<uint32_t> inc_off(n_var + 1, 0);
        for (int64_t e = 0; e < n_edges; ++e) {
            const int s = (int)(words[e] & 0xffffu), d = (int)((words[e] >> 16) & 0x7fffu);
            if (s == d) continue;
            if (s < n_var) ++inc_off[s + 1];
            if (d < n_var) ++inc_off[d + 1];
        }
        for (int v = 0; v < n_var; ++v) inc_off[v + 1] += inc_off[v];
        std::vector<Inc> inc(inc_off[n_var]);
        std::vector<uint32_t> cur(inc_off.begin(), inc_off.end() - 1);
        for (int64_t e = 0; e < n_edges; ++e) {
            const int s = (int)(words[e] & 0xffffu), d = (int)((words[e] >> 16) & 0x7fffu);
            if (s == d) continue;
            if (s < n_var) inc[cur[s]++] = {d, (uint32_t)e, 0};
            if (d < n_var) inc[cur[d]++] = {s, (uint32_t)e, 1};
        }
        auto tile_of = [&](int I, int J) -> uint32_t {            // I >= J
            if (I == J) return colptr[J];
            const auto it = std::lower_bound(st[J].begin(), st[J].end(), I);
            return colptr[J] + 1u + (uint32_t)(it - st[J].begin());
        };
        for (size_t p = 0; p < ipos.size(); ++p) {
            node_items[p] = (uint32_t)(items.size() / 8);
            if (ipos[p] == 0xffffffffu) continue;
            const int v = (int)ipos[p];
            std::sort(inc.begin() + inc_off[v], inc.begin() + inc_off[v + 1], [](const Inc &a, const Inc &b) {
                return a.nbr != b.nbr ? a.nbr < b.nbr : a.dir != b.dir ? a.dir < b.dir : a.rec < b.rec; });
            for (uint32_t i = inc_off[v]; i < inc_off[v + 1];) {
                const int u = inc[i].nbr;
                uint32_t cross = 0xffffffffu, xu = n_pad;          // constants read the zero slot behind the vector
                if (u < n_var) {
                    xu = 16u * (uint32_t)blk[u] + 2u * (uint32_t)slot[u];
                    if (blk[u] < blk[v] || (blk[u] == blk[v] && slot[u] < slot[v]))
                        cross = tile_of(blk[v], blk[u]) * 256u + (2u * (uint32_t)slot[v]) * 16u + 2u * (uint32_t)slot[u];
                }
                // an item = 8 words: {row of the node, row of the neighbour (or the zero slot), offset of the pair's 2x2 block in A (or none),
                // records, the first two records inline (record << 2 | direction << 1 | count the cost here), first further record in
                // item_edges, 0} - the usual pair (one record per direction) needs no second lookup
                uint32_t ew[2] = {0u, 0u}, n_e = 0;
                const uint32_t ext = (uint32_t)item_edges.size();
                for (; i < inc_off[v + 1] && inc[i].nbr == u; ++i, ++n_e) {
                    const uint32_t cost_flag = inc[i].dir == 0 ? 1u : (u >= n_var ? 1u : 0u);   // every record's cost is counted once
                    const uint32_t w = inc[i].rec << 2 | (uint32_t)inc[i].dir << 1 | cost_flag;
                    if (n_e < 2) ew[n_e] = w; else item_edges.push_back(w);
                }
                items.push_back(2u * (uint32_t)p); items.push_back(xu); items.push_back(cross); items.push_back(n_e);
                items.push_back(ew[0]); items.push_back(ew[1]); items.push_back(ext); items.push_back(0u);
            }
        }
        node_items[ipos.size()] = (uint32_t)(items.size() / 8);
        for (int pad = 0; pad < 8; ++pad) items.push_back(pad == 2 ? 0xffffffffu : 0u);        // one item past the end (the sweep reads one ahead)
    }
    const uint32_t n_items = (uint32_t)(items.size() / 8) - 1;
    lap("sweep items");
    // ---- the blob ----
    std::vector<uint32_t> &blob = out.blob;
    blob.assign(kTreeHdrWords, 0);
    put(blob, 8, colptr); put(blob, 9, rowsof); put(blob, 10, nreal); put(blob, 11, level_ptr); put(blob, 12, level_cols);
    put(blob, 13, p1_ptr); put(blob, 14, p1_tasks); put(blob, 15, upd); put(blob, 16, x_ptr); put(blob, 17, x_tasks); put(blob, 18, ncarry);
    put(blob, 19, items); put(blob, 20, item_edges); put(blob, 21, node_items); put(blob, 22, ipos);
    put(blob, 25, col_upd_ptr); put(blob, 26, col_upd); put(blob, 27, col_desc);
    while (blob.size() % 64) blob.push_back(0);                  // the tiles start at a multiple of 32 doubles
    out.n_var = n_var; out.NB = NB; out.n_levels = n_levels; out.n_tiles = n_tiles; out.n_items = n_items; out.n_updates = n_upd;
    out.n_tracks = T; out.n_segments = S;
    blob[0] = (uint32_t)NB; blob[1] = n_tiles; blob[2] = (uint32_t)(blob.size() / 2); blob[4] = n_pad; blob[5] = (uint32_t)n_levels;
    blob[6] = n_items; blob[7] = (uint32_t)(p1_tasks.size() / 4);
    // "thin" plan: the kernel runs the columns on dependency counters (a column starts when its children are done) instead of a barrier
    // per level.
    // (A column with a few tiles beyond the three it carries finishes them itself, behind its elimination; plans with many such tiles -
    // dense components - keep the barrier schedule, where tile tasks are dealt to all waves.)
    {
        uint32_t max_extra = 0;
        for (int J = 0; J < NB; ++J) max_extra = std::max<uint32_t>(max_extra, (uint32_t)st[J].size() - ncarry[J]);
        blob[28] = (NB <= kTreeMaxFlagColumns && max_extra <= 6u && p1_tasks.size() / 4 <= (size_t)NB / 8 + 4) ? 1u : 0u;
    }
    // two sets of tiles - A (the sweep stores J^T J there, always the same entries: zeroed once per solve, never by a sweep) and U (the
    // factor) -, then 6 doubles of partial sums per item, then the vectors
    const uint64_t off_part = (uint64_t)blob[2] + 512ull * n_tiles;
    const uint64_t off_vec = (off_part + 6ull * n_items + 31) / 32 * 32;
    // The kernel addresses A's tiles, the factor's tiles and the vectors through three buffer descriptors with 32-bit BYTE offsets
    // (tile << 11): fewer than 2^21 tiles (4 GB per array) and 32-bit double offsets for the rest - a dense 16 k-node component
    // still fits, a dense 32 k-node one (34 GB of tiles) is refused with an error instead of wrapping around.
    if (n_tiles >= (1u << 21) || n_edges >= (1ll << 30) || off_vec + (uint64_t)kTreeVectors * out.vec_stride() + out.team_doubles() >= (1ull << 32)) { out.blob.clear(); return; }
    blob[3] = (uint32_t)off_vec; blob[23] = (uint32_t)off_part; blob[24] = (uint32_t)out.vec_stride();
    // behind the vectors: the words several workgroups working on ONE component meet at (solve_tree_component<.., TEAM>): 16 words
    // (the bad-pivot flag first) + one dependency counter per column, touched by agent-scope atomics only
    blob[29] = (uint32_t)(off_vec + (uint64_t)kTreeVectors * out.vec_stride());
    // a model of the factorization's critical path: per level, the column tasks dealt to 8 waves
    {
        uint64_t rounds = 0;
        for (int l = 0; l < n_levels; ++l) rounds += (level_ptr[l + 1] - level_ptr[l] + 7) / 8;
        out.column_rounds = rounds;
        size_t mf = 0;
        for (int J = 0; J < NB; ++J) mf = std::max(mf, st[J].size());
        out.max_front = (int)mf + 1;
    }
}

uint64_t TreePlan::vec_stride() const { return 16ull * NB + 16ull; }
uint64_t TreePlan::header_doubles() const { return blob.size() / 2; }
uint64_t TreePlan::team_doubles() const { return (16ull + (uint64_t)NB + 1) / 2; }
uint64_t TreePlan::doubles() const { return blob.empty() ? 0 : (uint64_t)blob[3] + (uint64_t)kTreeVectors * vec_stride() + team_doubles(); }

}  // namespace lfr

// The plan of one component as the kernel reads it (tests/test_tree_plan.py executes it on the CPU).  words[e] = src | (dst | kind << 15) << 16
// of the component's records.  Returns the number of 32-bit words of the blob (copied to `blob` when cap is large enough), < 0 on error;
// info[0..7] = blocks, tiles, levels, items, updates, tracks, segments, column rounds.
extern "C" int64_t lfr_debug_tree_plan(int32_t n_var, int64_t n_edges, const uint32_t *words, uint32_t *blob, int64_t cap, int64_t *info) {
    if (n_var < 0 || n_var > 32767 || n_edges < 0 || (n_edges > 0 && !words)) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::TreePlan pl;
    lfr::tree_plan(n_var, n_edges, words, pl);
    if (pl.blob.empty()) { lfr::set_error("component too large for the plan's 32-bit offsets"); return LFR_ERR_UNSUPPORTED; }
    if (blob && cap >= (int64_t)pl.blob.size()) memcpy(blob, pl.blob.data(), 4 * pl.blob.size());
    if (info) { info[0] = pl.NB; info[1] = pl.n_tiles; info[2] = pl.n_levels; info[3] = pl.n_items; info[4] = (int64_t)pl.n_updates; info[5] = pl.n_tracks; info[6] = pl.n_segments; info[7] = (int64_t)pl.column_rounds; }
    return (int64_t)pl.blob.size();
}

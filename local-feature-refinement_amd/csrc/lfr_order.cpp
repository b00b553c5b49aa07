// Fill-reducing order + block-envelope plan for the components whose normal matrices do not fit LDS (kernel class KC_GLOBAL).
//
// The reference asks Ceres for SPARSE_NORMAL_CHOLESKY (solve.cc:147): the normal matrix of a component is as sparse as its match
// graph - one 2x2 block per matched node pair.  A component at the size cap (#images nodes, solve.cc:586) is a few hundred short
// tracks (cliques / sparse lattices of a handful of nodes) hanging together through inter-track matches, tree-plus-few-cycles at
// the level of tracks.  Round 2 factored those 2.7 k-row systems DENSE in HBM (6.5 GFLOP each).  Here the variable nodes are
// renumbered so that the matrix has a small ENVELOPE (every row's nonzeros start close to the diagonal) and the kernel stores and
// factors only the 16x16 tiles inside the envelope of each block row (an LDL^T without pivoting never fills outside the envelope).
//
// Order: tracks = connected pieces of the intra-track (Cauchy) edges among the variable nodes; the tracks' meta graph (inter-track
// edges) gets a spanning forest; tracks are numbered in POSTORDER of that forest with the heaviest subtree first, a track's nodes
// contiguous (Cuthill-McKee inside the track).  A node's lower-numbered neighbours are then its own track (distance <= track
// length), child tracks (distance <= the sizes of the lighter sibling subtrees: heavy-first keeps the sum over the tree at
// O(n log n)) and the few non-tree inter-track matches.  Plain reverse Cuthill-McKee over the whole component is computed as
// well; the plan with fewer tiles wins (both are deterministic functions of the component's edge list).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <queue>
#include <vector>

#include "lfr_internal.hpp"

namespace lfr {

namespace {

struct Csr {
    std::vector<int32_t> off, adj;
    std::vector<uint8_t> kind;
};

// adjacency among the variable nodes (both directions of a match are records of the component: every undirected pair is seen twice)
Csr build_adj(int nv, int64_t ne, const uint32_t *src_dst_kind) {
    Csr g;
    g.off.assign(nv + 1, 0);
    for (int64_t e = 0; e < ne; ++e) {
        const uint32_t w = src_dst_kind[e];
        const int s = (int)(w & 0xffffu), d = (int)((w >> 16) & 0x7fffu);
        if (s < nv && d < nv && s != d) ++g.off[s + 1];
    }
    for (int i = 0; i < nv; ++i) g.off[i + 1] += g.off[i];
    g.adj.resize(g.off[nv]); g.kind.resize(g.off[nv]);
    std::vector<int32_t> cur(g.off.begin(), g.off.end() - 1);
    for (int64_t e = 0; e < ne; ++e) {
        const uint32_t w = src_dst_kind[e];
        const int s = (int)(w & 0xffffu), d = (int)((w >> 16) & 0x7fffu);
        if (s < nv && d < nv && s != d) { g.adj[cur[s]] = d; g.kind[cur[s]] = (uint8_t)(w >> 31); ++cur[s]; }
    }
    return g;
}

int find_root(std::vector<int32_t> &p, int x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

// Cuthill-McKee over the nodes `members` (a connected piece w.r.t. `use_edge`), appended to `out`; start = a pseudo-peripheral node
template <typename UseEdge>
void cuthill_mckee(const Csr &g, const std::vector<int32_t> &members, std::vector<int32_t> &mark, int stamp, UseEdge use_edge,
                   std::vector<int32_t> &out) {
    if (members.empty()) return;
    auto degree = [&](int v) { int d = 0; for (int k = g.off[v]; k < g.off[v + 1]; ++k) d += use_edge(v, k) ? 1 : 0; return d; };
    auto bfs = [&](int start, std::vector<int32_t> &order, int st) {
        order.clear();
        order.push_back(start); mark[start] = st;
        std::vector<std::pair<int, int>> nb;
        for (size_t h = 0; h < order.size(); ++h) {
            const int v = order[h];
            nb.clear();
            for (int k = g.off[v]; k < g.off[v + 1]; ++k) {
                const int w = g.adj[k];
                if (!use_edge(v, k) || mark[w] == st) continue;
                mark[w] = st;
                nb.push_back({degree(w), w});
            }
            std::sort(nb.begin(), nb.end());
            for (auto &q : nb) order.push_back(q.second);
        }
    };
    int start = members[0];
    for (int v : members) if (std::make_pair(degree(v), v) < std::make_pair(degree(start), start)) start = v;
    std::vector<int32_t> order;
    bfs(start, order, stamp);                          // first sweep: its last node is far from the start
    const int far = order.back();
    bfs(far, order, stamp + 1);
    // nodes of `members` the edges do not reach from `far` (cannot happen for a connected piece) are appended as they are
    for (int v : members) if (mark[v] != stamp + 1) { mark[v] = stamp + 1; order.push_back(v); }
    out.insert(out.end(), order.begin(), order.end());
}

// positions by tracks in heavy-first postorder of the meta forest
void order_by_tracks(const Csr &g, int nv, std::vector<int32_t> &seq) {
    std::vector<int32_t> uf(nv);
    std::iota(uf.begin(), uf.end(), 0);
    for (int v = 0; v < nv; ++v)
        for (int k = g.off[v]; k < g.off[v + 1]; ++k)
            if (g.kind[k] == 0) { const int a = find_root(uf, v), b = find_root(uf, g.adj[k]); if (a != b) uf[std::max(a, b)] = std::min(a, b); }
    std::vector<int32_t> track_of(nv), track_root;           // tracks numbered by their smallest node
    std::vector<int32_t> idx(nv, -1);
    for (int v = 0; v < nv; ++v) { const int r = find_root(uf, v); if (idx[r] < 0) { idx[r] = (int)track_root.size(); track_root.push_back(r); } track_of[v] = idx[r]; }
    const int T = (int)track_root.size();
    std::vector<std::vector<int32_t>> members(T);
    for (int v = 0; v < nv; ++v) members[track_of[v]].push_back(v);
    // meta adjacency (inter-track edges), deduplicated
    std::vector<std::vector<int32_t>> madj(T);
    for (int v = 0; v < nv; ++v)
        for (int k = g.off[v]; k < g.off[v + 1]; ++k) {
            const int a = track_of[v], b = track_of[g.adj[k]];
            if (a != b) madj[a].push_back(b);
        }
    for (auto &l : madj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    // spanning forest by BFS from the largest track of every meta component; subtree weights in nodes
    std::vector<int32_t> parent(T, -2), bfs_order;
    std::vector<int64_t> weight(T);
    for (int t = 0; t < T; ++t) weight[t] = (int64_t)members[t].size();
    std::vector<int32_t> by_size(T);
    std::iota(by_size.begin(), by_size.end(), 0);
    std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) { return members[a].size() > members[b].size(); });
    std::vector<int32_t> roots;
    for (int s : by_size) {
        if (parent[s] != -2) continue;
        parent[s] = -1; roots.push_back(s);
        const size_t h0 = bfs_order.size();
        bfs_order.push_back(s);
        for (size_t h = h0; h < bfs_order.size(); ++h) {
            const int t = bfs_order[h];
            for (int u : madj[t]) if (parent[u] == -2) { parent[u] = t; bfs_order.push_back(u); }
        }
    }
    for (size_t h = bfs_order.size(); h-- > 0;) { const int t = bfs_order[h]; if (parent[t] >= 0) weight[parent[t]] += weight[t]; }
    std::vector<std::vector<int32_t>> children(T);
    for (int t : bfs_order) if (parent[t] >= 0) children[parent[t]].push_back(t);
    for (auto &c : children) std::stable_sort(c.begin(), c.end(), [&](int a, int b) { return weight[a] > weight[b]; });   // heaviest first = farthest
    // postorder, iterative
    std::vector<int32_t> track_seq;
    track_seq.reserve(T);
    std::vector<std::pair<int32_t, size_t>> stack;
    for (int r : roots) {
        stack.push_back({r, 0});
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second < children[top.first].size()) { const int c = children[top.first][top.second++]; stack.push_back({c, 0}); }
            else { track_seq.push_back(top.first); stack.pop_back(); }
        }
    }
    std::vector<int32_t> mark(nv, 0);
    int stamp = 1;
    seq.clear();
    seq.reserve(nv);
    for (int t : track_seq) {
        cuthill_mckee(g, members[t], mark, stamp, [&](int v, int k) { return track_of[g.adj[k]] == track_of[v]; }, seq);
        stamp += 2;
    }
}

void order_rcm(const Csr &g, int nv, std::vector<int32_t> &seq) {
    std::vector<int32_t> uf(nv);
    std::iota(uf.begin(), uf.end(), 0);
    for (int v = 0; v < nv; ++v)
        for (int k = g.off[v]; k < g.off[v + 1]; ++k) { const int a = find_root(uf, v), b = find_root(uf, g.adj[k]); if (a != b) uf[std::max(a, b)] = std::min(a, b); }
    std::vector<std::vector<int32_t>> pieces;
    std::vector<int32_t> idx(nv, -1);
    for (int v = 0; v < nv; ++v) { const int r = find_root(uf, v); if (idx[r] < 0) { idx[r] = (int)pieces.size(); pieces.emplace_back(); } pieces[idx[r]].push_back(v); }
    std::vector<int32_t> mark(nv, 0);
    int stamp = 1;
    seq.clear();
    for (auto &m : pieces) {
        const size_t b = seq.size();
        cuthill_mckee(g, m, mark, stamp, [](int, int) { return true; }, seq);
        std::reverse(seq.begin() + b, seq.end());
        stamp += 2;
    }
}

// block envelope of the order `seq` (seq[p] = node at position p): first block column of every block row; returns the tile count
uint64_t envelope(const Csr &g, int nv, const std::vector<int32_t> &seq, std::vector<uint16_t> &fb) {
    std::vector<int32_t> pos(nv);
    for (int p = 0; p < nv; ++p) pos[seq[p]] = p;
    const int n = 2 * nv, RT = (n + 1 + 15) >> 4;
    fb.assign(RT, 0);
    for (int R = 0; R < RT; ++R) fb[R] = (uint16_t)R;
    for (int v = 0; v < nv; ++v) {
        int lo = pos[v];
        for (int k = g.off[v]; k < g.off[v + 1]; ++k) lo = std::min(lo, pos[g.adj[k]]);
        const int R = (2 * pos[v]) >> 4, J = (2 * lo) >> 4;
        fb[R] = std::min<uint16_t>(fb[R], (uint16_t)J);
    }
    fb[n >> 4] = 0;                                            // the block row of the right-hand side (row n) spans every column
    uint64_t tiles = 0;
    for (int R = 0; R < RT; ++R) tiles += (uint64_t)(R - fb[R] + 1);
    return tiles;
}

}  // namespace

void sky_plan(int n_var, int64_t n_edges, const uint32_t *src_dst_kind, SkyPlan &out) {
    const Csr g = build_adj(n_var, n_edges, src_dst_kind);
    std::vector<int32_t> seq_a, seq_b;
    std::vector<uint16_t> fb_a, fb_b;
    order_by_tracks(g, n_var, seq_a);
    order_rcm(g, n_var, seq_b);
    const uint64_t ta = envelope(g, n_var, seq_a, fb_a), tb = envelope(g, n_var, seq_b, fb_b);
    const bool use_a = ta <= tb;
    const std::vector<int32_t> &seq = use_a ? seq_a : seq_b;
    out.n_var = n_var;
    out.n = 2 * n_var;
    out.RT = (out.n + 1 + 15) >> 4;
    out.fb = use_a ? fb_a : fb_b;
    out.order_used = use_a ? 0 : 1;
    out.tiles_by_tracks = ta; out.tiles_rcm = tb;
    out.pos.resize(n_var); out.ipos.resize(n_var);
    for (int p = 0; p < n_var; ++p) { out.ipos[p] = (uint16_t)seq[p]; out.pos[seq[p]] = (uint16_t)p; }
    out.tilebase.assign(out.RT + 1, 0);
    for (int R = 0; R < out.RT; ++R) out.tilebase[R + 1] = out.tilebase[R] + (uint32_t)(R - out.fb[R] + 1);
}

// Workspace image of a plan (what the kernel reads; see solve_sky_component in lfr_solve.hip):
//   u32 hdr[8] = {RT, n_tiles, off_tiles (doubles from the base), off_vec (doubles), n_pad, 0, 0, 0}, u32 tilebase[RT + 1], u16 fb[RT],
//   u16 pos[n_var], u16 ipos[n_var]; padded to 8 bytes; the tiles start at a multiple of 32 doubles
uint64_t SkyPlan::header_doubles() const {
    const uint64_t bytes = 32 + 4ull * (RT + 1) + 2ull * RT + 4ull * n_var;
    return ((bytes + 7) / 8 + 31) / 32 * 32;
}
uint64_t SkyPlan::n_pad() const { return 16ull * RT + 16ull; }
uint64_t SkyPlan::doubles() const { return header_doubles() + 256ull * tilebase[RT] + kSkyVectors * n_pad(); }
void SkyPlan::write_header(void *dst) const {
    uint8_t *p = (uint8_t *)dst;
    memset(p, 0, header_doubles() * 8);
    uint32_t hdr[8] = {(uint32_t)RT, tilebase[RT], (uint32_t)header_doubles(), (uint32_t)(header_doubles() + 256ull * tilebase[RT]), (uint32_t)n_pad(), 0, 0, 0};
    memcpy(p, hdr, 32); p += 32;
    memcpy(p, tilebase.data(), 4 * (RT + 1)); p += 4 * (RT + 1);
    memcpy(p, fb.data(), 2 * RT); p += 2 * RT;
    memcpy(p, pos.data(), 2 * n_var); p += 2 * n_var;
    memcpy(p, ipos.data(), 2 * n_var);
}

}  // namespace lfr

extern "C" int64_t lfr_debug_sky_plan(int32_t n_var, int64_t n_edges, const uint32_t *src_dst_kind, uint16_t *pos, uint16_t *first_block, int64_t *info) {
    if (n_var < 0 || n_var > 32767 || n_edges < 0 || (n_edges > 0 && !src_dst_kind)) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::SkyPlan pl;
    lfr::sky_plan(n_var, n_edges, src_dst_kind, pl);
    if (pos) memcpy(pos, pl.pos.data(), 2 * (size_t)n_var);
    if (first_block) memcpy(first_block, pl.fb.data(), 2 * (size_t)pl.RT);
    if (info) { info[0] = pl.RT; info[1] = pl.tilebase[pl.RT]; info[2] = (int64_t)pl.tiles_by_tracks; info[3] = (int64_t)pl.tiles_rcm; info[4] = pl.order_used; info[5] = (int64_t)pl.doubles(); }
    return pl.tilebase[pl.RT];
}

// Device-side batch assembly (reference: create_and_solve_problem's problem construction,
// solve.cc:94-143, and the largest-first ordering of solve.cc:599-604).
//
// Input: the match graph in HBM (DevGraph: per-match endpoints, similarity, two 3x3x2 flow grids) and the
// per-node labels in HBM (DevProblem: track, component, is_root - left there by the device graph stage).
// Output, built entirely on the GPU and bit-identical to the host assembly of lfr_graph.cpp: CompDesc[],
// 80-byte EdgeRec[] in the reference's residual-block order, node_ids[], NodeInc[], in_idx[] - sorted by
// kernel class, then edge count descending - plus the per-component workspace offsets and the launch
// geometry of the solve kernels (AsmSummary).  The flows are gathered into edge records at HBM speed from
// their staged copy (uploaded on the copy stream while the graph stage ran), from the caller's device
// arrays, or - sharded batches - zero-copy from pinned host memory (only the shard's rows cross PCIe).
//
// No host round trips: every launch is sized by an upper bound (N, 2M, C) and bounded on the device by the
// counts the previous kernels left there; one read-back of the 160-byte summary ends the stage.
// Everything is integer/byte work: radix sorts (rocPRIM), scans, histograms with integer atomics
// (exact and order-independent), gathers.  HBM-bound; no LDS tiling to speak of.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstring>
#include <vector>

#include "lfr_assemble.hpp"

namespace lfr {

namespace {

constexpr int kThreads = kPipeThreads;
inline dim3 grid_for(int64_t n) { return pipe_grid(n); }

__device__ __forceinline__ void edge_ends(const uint32_t *node1, const uint32_t *node2, int64_t e, uint32_t &src, uint32_t &dst) {
    const int64_t m = e >> 1;
    const uint32_t a = node1[m], b = node2[m];
    src = (e & 1) ? b : a;      // directed edge 2m = node1->node2, 2m+1 = node2->node1 (solve.cc:477-478)
    dst = (e & 1) ? a : b;
}

// kept = same track or same component (solve.cc:105,114); marks nodes with a kept out-edge.  The test is symmetric: both directions
// of a match are kept or dropped together.  match_key (optional): the key of the match-level sort - the component of the match
// (a track lies inside one component, so both ends agree) with the edge kind (inter-track, solve.cc:114-123) in bit 31, dropped_key for
// a dropped match.
__global__ void k_mark_kept(int64_t M, const uint32_t *node1, const uint32_t *node2, const int32_t *track, const int32_t *comp, uint8_t *kept,
                            uint8_t *opt, uint32_t dropped_key, uint32_t *match_key) {
    // Two threads per match, one per direction, and no gather that is not needed (same track: the components are not compared; only the
    // even direction writes the key and needs its end's component).  Measured on config 4: 51 us this way, 65 us with all four gathers
    // unconditionally, 54 us with one thread per match doing both ends; without the key 37 us, but then k_match_keys_comp gathers the
    // component again (29 us instead of 15): the kernel is a chain of two dependent gathers and lives on the number of them in flight.
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * M) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const bool same_track = track[s] == track[d], keyed = match_key && !(e & 1);
    int32_t cs = 0;
    bool k = same_track;
    if (!same_track) { cs = comp[s]; k = cs == comp[d]; }
    else if (keyed) cs = comp[s];
    kept[e] = k;
    if (k) opt[s] = 1;
    if (keyed) match_key[e >> 1] = k ? ((uint32_t)cs | (same_track ? 0u : 0x80000000u)) : dropped_key;
}

// variable = has a kept out-edge and is not its track's root (solve.cc:127,133-141); sizes of the tracks.  node_key (optional): the key
// of the node sort - component, variables before constants - whose runs are the components' node counts (k_node_runs); without it the
// counts are taken here, one atomic per node and counter.
__global__ void k_mark_var(int64_t n_nodes, const uint8_t *opt, const uint8_t *is_root, const int32_t *track, const int32_t *comp,
                           uint8_t *is_var, uint32_t *c_nodes, uint32_t *c_var, uint32_t *t_size, int32_t *t_comp, uint32_t *node_key, uint32_t *node_id) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const bool v = opt[n] && !is_root[n];
    const int32_t c = comp[n], t = track[n];
    is_var[n] = v;
    if (node_key) { node_key[n] = ((uint32_t)c << 1) | (v ? 0u : 1u); node_id[n] = (uint32_t)n; }
    else { atomicAdd(&c_nodes[c], 1u); if (v) atomicAdd(&c_var[c], 1u); }
    atomicAdd(&t_size[t], 1u);
    t_comp[t] = c;
}
// nodes sorted by (component, variables first): where a component's run begins, where its variables end, where it ends (all start at
// zero; k_comp_keys turns them into the counts)
__global__ void k_node_runs(int64_t N, const uint32_t *keys, uint32_t *run_begin, uint32_t *var_end, uint32_t *run_end) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t k = keys[i], c = k >> 1;
    const bool first = i == 0 || (keys[i - 1] >> 1) != c, last = i + 1 == N || (keys[i + 1] >> 1) != c;
    if (first) run_begin[c] = (uint32_t)i;
    if ((k & 1u) && (first || keys[i - 1] != k)) var_end[c] = (uint32_t)i;      // the first constant
    if (last) { run_end[c] = (uint32_t)(i + 1); if (!(k & 1u)) var_end[c] = (uint32_t)(i + 1); }      // (no constant at all)
}
// ... and every node of a solved component to its place in the batch: position in the run = local index
__global__ void k_place_nodes(int64_t N, const uint32_t *keys, const uint32_t *node_sorted, const int32_t *di_of_comp, const uint32_t *node_off,
                              const uint32_t *run_begin, uint32_t *node_ids, uint32_t *local_of) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const uint32_t c = keys[p] >> 1;
    const int32_t di = di_of_comp[c];
    if (di < 0) return;
    const uint32_t l = (uint32_t)p - run_begin[c], n = node_sorted[p];
    node_ids[node_off[di] + l] = n;
    local_of[n] = l;
}

// one thread per MATCH: both directions are kept or dropped together (same track / same component and "an end is variable" are
// symmetric), so one atomic of 2 per match instead of one per directed edge (5 M atomics on 147 k addresses were 0.38 ms)
__global__ void k_count_edges(int64_t n_matches, const uint32_t *node1, const uint32_t *node2, const int32_t *comp,
                              const uint8_t *is_var, uint8_t *kept, uint32_t *c_edges) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_matches) return;
    const uint32_t a = node1[m], b = node2[m];
    const bool var = is_var[a] || is_var[b];                 // both ends constant: not in the reduced program
    const bool k0 = kept[2 * m] && var, k1 = kept[2 * m + 1] && var;
    kept[2 * m] = k0; kept[2 * m + 1] = k1;
    if (k0) atomicAdd(&c_edges[comp[a]], 1u + (k1 && comp[b] == comp[a] ? 1u : 0u));
    if (k1 && !(k0 && comp[b] == comp[a])) atomicAdd(&c_edges[comp[b]], 1u);
}

// Few, large components (a giant component cut into a few thousand parts: config 5, 9.3 M edges over 2152 counters - 0.6 ms of
// atomics taking turns): the same count through per-node counters (65 per word instead of 4300), then one add per node.
__global__ void k_count_edges_by_node(int64_t n_matches, const uint32_t *node1, const uint32_t *node2, const uint8_t *is_var, uint8_t *kept, uint32_t *n_edges) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_matches) return;
    const uint32_t a = node1[m], b = node2[m];
    const bool var = is_var[a] || is_var[b];
    const bool k0 = kept[2 * m] && var, k1 = kept[2 * m + 1] && var;
    kept[2 * m] = k0; kept[2 * m + 1] = k1;
    if (k0) atomicAdd(&n_edges[a], 1u);                       // edge 2m leaves node1, edge 2m+1 leaves node2
    if (k1) atomicAdd(&n_edges[b], 1u);
}
__global__ void k_sum_node_edges(int64_t n_nodes, const int32_t *comp, const uint32_t *n_edges, uint32_t *c_edges) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes && n_edges[n]) atomicAdd(&c_edges[comp[n]], n_edges[n]);
}

__device__ __forceinline__ int classify_dev(uint32_t rows, uint32_t n_edges, uint32_t block_max) {
    if (rows <= 8 && n_edges <= 24) return KC_G8;
    if (rows <= 16 && n_edges <= 96) return KC_G16;
    if (rows <= 24 && n_edges <= 192) return KC_G64_2;
    if (rows <= 32 && n_edges <= 320) return KC_G64_4;
    if (rows <= min((uint32_t)kBlockRowsS, block_max)) return KC_BLOCK;
    if (rows <= min((uint32_t)kBlockRowsM, block_max)) return KC_BLOCK_M;
    if (rows <= block_max) return KC_BLOCK_L;
    return KC_GLOBAL;
}

// per component: solvable? class; the three sort keys of the batch order
// batch order = class, then edges descending, then variables descending, then id: one 52-bit key (a stable sort keeps the ids
// ascending inside ties).  Three LSD passes over 32-bit keys cost three block sorts and thirty merge launches of ~6 us each.
__global__ void k_comp_keys(int64_t n_comp, uint32_t *c_nodes, uint32_t *c_var, uint32_t *c_edges, const uint32_t *run_begin,
                            const uint32_t *run_end, const uint32_t *node_begin, const uint32_t *node_var_end, const uint32_t *node_end,
                            unsigned long long *key, uint32_t *ids, uint32_t *too_big, uint32_t block_max,
                            int64_t n_tracks, const uint32_t *t_size, const int32_t *t_comp, uint32_t *c_tracks,
                            uint32_t e_max, uint32_t v_max, int e_bits, int v_bits) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_tracks && t_size[c] >= 2) atomicAdd(&c_tracks[t_comp[c]], 1u);      // (thread t doubles as track t: k_count_tracks' launch saved)
    if (c >= n_comp) return;
    if (node_begin) { c_nodes[c] = node_end[c] - node_begin[c]; c_var[c] = node_var_end[c] - node_begin[c]; }      // nodes sorted before the counts (k_node_runs)
    if (run_begin) c_edges[c] = 2u * (run_end[c] - run_begin[c]);       // matches sorted before the counts (k_match_keys_comp): both directions of every match of the run
    const bool solvable = c_nodes[c] >= 2 && c_var[c] >= 1;   // solve.cc:619-622; no variable: nothing to solve
    if (solvable && c_nodes[c] > 32767) *too_big = 1u;
    // (the fields are as wide as the host's bounds need - e_max >= every edge count, v_max >= min(variables, 0xffff): the same order
    // in 18 key bits on config 4 instead of 52, i.e. two radix passes instead of a block sort and eight merges)
    const unsigned long long kv = v_max - min(min(c_var[c], 0xffffu), v_max);   // descending
    const unsigned long long ke = e_max - min(c_edges[c], e_max);               // descending
    const unsigned long long kc = solvable ? (uint32_t)classify_dev(2 * c_var[c], c_edges[c], block_max) : kNoClass;
    key[c] = (kc << (e_bits + v_bits)) | (ke << v_bits) | kv;
    ids[c] = (uint32_t)c;
}
__global__ void k_class_of_key(int64_t n, const unsigned long long *key, uint32_t *cls, int class_shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cls[i] = (uint32_t)(key[i] >> class_shift);
}


// after the sort: per desc sizes (0 for unsolvable) + inverse permutation
// (sorted_keys != nullptr: the class comes straight from the sorted keys and is written to class_sorted here - k_class_of_key's launch saved
// on the unsharded road)
__global__ void k_desc_sizes(int64_t n_comp, const uint32_t *perm, uint32_t *class_sorted, const unsigned long long *sorted_keys, const uint32_t *c_nodes,
                             const uint32_t *c_edges, uint32_t *d_nodes, uint32_t *d_edges, int32_t *di_of_comp, int class_shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_comp) return;
    const uint32_t c = perm[i];
    uint32_t cls;
    if (sorted_keys) { cls = (uint32_t)(sorted_keys[i] >> class_shift); class_sorted[i] = cls; }
    else cls = class_sorted[i];
    const bool solvable = cls != kNoClass;
    d_nodes[i] = solvable ? c_nodes[c] : 0u;
    d_edges[i] = solvable ? c_edges[c] : 0u;
    di_of_comp[c] = solvable ? (int32_t)i : -1;
}

__global__ void k_node_keys(int64_t n_nodes, const int32_t *comp, const int32_t *di_of_comp, const uint8_t *is_var,
                            uint32_t dropped_key, uint32_t *keys, uint32_t *ids) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const int32_t di = di_of_comp[comp[n]];
    keys[n] = di < 0 ? dropped_key : (((uint32_t)di << 1) | (is_var[n] ? 0u : 1u));     // variables first, then constants
    ids[n] = (uint32_t)n;
}

__global__ void k_node_locals(int64_t cap, const uint32_t *total_nodes, const uint32_t *node_sorted, const int32_t *comp, const int32_t *di_of_comp,
                              const uint32_t *node_off, uint32_t *node_ids, uint32_t *local_of) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap || p >= (int64_t)*total_nodes) return;
    const uint32_t n = node_sorted[p];
    node_ids[p] = n;
    local_of[n] = (uint32_t)p - node_off[di_of_comp[comp[n]]];
}

__global__ void k_edge_keys(int64_t n_dir, const uint32_t *node1, const uint32_t *node2, const int32_t *comp,
                            const int32_t *di_of_comp, const uint32_t *class_of_desc, const uint8_t *kept, int node_bits,
                            uint64_t dropped_key, uint64_t *keys, uint32_t *ids) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_dir) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const int32_t di = kept[e] ? di_of_comp[comp[s]] : -1;
    // workgroup classes: residual-block order (by source node, then edge id); packed classes: edge id only
    // (the sort is stable), so the two directions 2m, 2m+1 of a match become neighbouring records
    // (written with an early return + mask: the one-expression form `(di >= 0 && class < KC_BLOCK) ? 0 : s` was
    // miscompiled by hipcc 7.2 -O3 for gfx950 - the register holding s was reused before the select)
    // (keys are packed into node_bits + bits(C) bits so that the radix sort runs 5 passes instead of 8)
    if (di < 0) { keys[e] = dropped_key; ids[e] = (uint32_t)e; return; }
    const uint32_t by_source = class_of_desc[di] >= (uint32_t)KC_BLOCK ? 0xffffffffu : 0u;
    keys[e] = ((uint64_t)(uint32_t)di << node_bits) | (s & by_source);
    ids[e] = (uint32_t)e;
}

// the same order when no workgroup class is expected: the key is the descriptor index alone - 32 bits, a third less to move per radix pass
__global__ void k_edge_keys_packed(int64_t n_dir, const uint32_t *node1, const uint32_t *node2, const int32_t *comp, const int32_t *di_of_comp,
                                   const uint8_t *kept, uint32_t dropped_key, uint32_t *keys, uint32_t *ids) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_dir) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const int32_t di = kept[e] ? di_of_comp[comp[s]] : -1;
    keys[e] = di < 0 ? dropped_key : (uint32_t)di;
    ids[e] = (uint32_t)e;
}

// Round 5: ... and per MATCH.  The two directions of a match (edge ids 2 m, 2 m + 1) are kept or dropped together and sort next to each other
// (same component, consecutive ids), so the order of the edges is the order of the matches with every entry doubled: half the keys through
// the radix passes, and a kernel that writes the pairs out.
__global__ void k_match_keys_packed(int64_t M, const uint32_t *node1, const int32_t *comp, const int32_t *di_of_comp, const uint8_t *kept,
                                    uint32_t dropped_key, uint32_t *keys, uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int32_t di = kept[2 * m] ? di_of_comp[comp[node1[m]]] : -1;
    keys[m] = di < 0 ? dropped_key : (uint32_t)di;
    ids[m] = (uint32_t)m;
}
// ... and the record words of both directions with it (words != nullptr: the fused gather's k_edge_words, one walk through the match arrays
// per match instead of one per directed edge)
__global__ void k_expand_match_order(int64_t M, const uint32_t *total_edges_p, const uint32_t *match_sorted, const uint32_t *node1, const uint32_t *node2,
                                     const int32_t *track, const uint32_t *local_of, uint32_t *edge_sorted, uint32_t *words) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const uint32_t m = match_sorted[p];
    reinterpret_cast<uint2 *>(edge_sorted)[p] = make_uint2(2u * m, 2u * m + 1u);
    if (words && 2 * p < (int64_t)*total_edges_p) {
        const uint32_t a = node1[m], b = node2[m];
        const uint32_t kind = track[a] != track[b] ? 1u : 0u, la = local_of[a], lb = local_of[b];
        reinterpret_cast<uint2 *>(words)[p] = make_uint2(la | ((lb | (kind << 15)) << 16), lb | ((la | (kind << 15)) << 16));
    }
}

// Round 6: the match-level order BEFORE the counts.  k_count_edges' 2.5 M atomics on 147 k counters (config 4) were 0.106 ms of a 0.76 ms
// assembly; the matches are sorted by component anyway - by the component's RANK in the batch, which needs the counts.  Sorted by the
// component's id instead (known from the graph stage), the runs of equal keys ARE the counts (no atomic), and the rank order is a
// permutation of whole runs: position of a match in the batch = its component's edge offset + its position in the run.  The sort is
// stable on the match id, so the records of a component come out in the order they always had.
// A match is in the reduced program when it is kept and one end is a variable; both directions go together and - a track lies inside
// one component - both ends are in one component: anything else is flagged like an unpaired record.
__global__ void k_match_keys_comp(int64_t M, const uint32_t *node1, const uint32_t *node2, const uint8_t *is_var, uint8_t *kept,
                                  uint32_t dropped_key, uint32_t *keys, uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t pk = keys[m];                             // k_mark_kept: component | kind << 31, or dropped_key
    const bool in = pk != dropped_key && (is_var[node1[m]] || is_var[node2[m]]);     // both ends constant: not in the reduced program
    if (!in && pk != dropped_key) reinterpret_cast<uint16_t *>(kept)[m] = 0;
    keys[m] = in ? (pk & 0x7fffffffu) : dropped_key;
    ids[m] = (uint32_t)m | (pk & 0x80000000u);               // the kind travels with the id (M < 2^30)
}
// one thread per sorted match: the first of a run writes where it begins, the last where it ends (both start at zero: no run, no edges);
// k_comp_keys turns the pair into the component's edge count
__global__ void k_match_runs(int64_t M, const uint32_t *keys, uint32_t n_comp, uint32_t *run_begin, uint32_t *run_end) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint32_t k = keys[i];
    if (k >= n_comp) return;
    if (i == 0 || keys[i - 1] != k) run_begin[k] = (uint32_t)i;
    if (i + 1 == M || keys[i + 1] != k) run_end[k] = (uint32_t)(i + 1);
}
// ... and every match to its place: both directions' edge ids (and record words, k_expand_match_order) at the component's offset
__global__ void k_place_matches(int64_t M, const uint32_t *keys, const uint32_t *match_sorted, uint32_t n_comp, const int32_t *di_of_comp,
                                const uint32_t *edge_off, const uint32_t *run_begin, const uint32_t *node1, const uint32_t *node2,
                                const uint32_t *local_of, uint32_t *edge_sorted, uint32_t *words) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const uint32_t c = keys[p];
    if (c >= n_comp) return;
    const int32_t di = di_of_comp[c];
    if (di < 0) return;
    const uint32_t q = (edge_off[di] >> 1) + ((uint32_t)p - run_begin[c]);
    const uint32_t mk = match_sorted[p], m = mk & 0x7fffffffu, kind = mk >> 31;
    reinterpret_cast<uint2 *>(edge_sorted)[q] = make_uint2(2u * m, 2u * m + 1u);
    if (words) {
        const uint32_t la = local_of[node1[m]], lb = local_of[node2[m]];
        reinterpret_cast<uint2 *>(words)[q] = make_uint2(la | ((lb | (kind << 15)) << 16), lb | ((la | (kind << 15)) << 16));
    }
}

// packed classes: records 2i, 2i+1 of a component must be the two directions of one match (the solve
// kernel's pair exchange relies on it)
__global__ void k_check_pairs(int64_t cap, const uint32_t *total_edges_p, const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2,
                              const int32_t *comp, const int32_t *di_of_comp, const uint32_t *class_of_desc,
                              const uint32_t *edge_off, uint32_t *flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total_edges = (int64_t)*total_edges_p;
    if (p >= cap || p >= total_edges) return;
    const uint32_t e = edge_sorted[p];
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const uint32_t di = (uint32_t)di_of_comp[comp[s]];
    if (class_of_desc[di] >= (uint32_t)KC_BLOCK) return;
    const uint32_t local = (uint32_t)p - edge_off[di];
    const int64_t q = (local & 1u) ? p - 1 : p + 1;
    if (q < 0 || q >= total_edges || edge_sorted[q] != (e ^ 1u)) *flag = 1u;
}

// one thread per (edge record, 16-byte chunk): writes the 80-byte EdgeRec of every kept edge whose match lies in
// [row_lo, row_hi) - the flows arrive in chunks of matches on the copy stream and each chunk is gathered as soon
// as it has landed
template <bool ALIGNED8>     // flow rows are 72 bytes: 8-byte aligned when the arrays are (ours always are; a caller's device flows may not be)
__global__ void k_emit_edges(int64_t cap, const uint32_t *total_edges_p, const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2,
                             const float *sim, const float *disp1, const float *disp2, const int32_t *track,
                             const uint32_t *local_of, const uint32_t *flow_row, int64_t row_lo, int64_t row_hi, uint4 *records,
                             const uint32_t *skip_below_p, const uint32_t *words) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t / 5;
    const int chunk = (int)(t - 5 * p);
    if (p >= cap || p >= (int64_t)*total_edges_p) return;
    if (skip_below_p && p < (int64_t)*skip_below_p) return;       // fused gather: the packed classes' records are never read
    const uint32_t e = edge_sorted[p];
    const int64_t m = (int64_t)(e >> 1);
    if (m < row_lo || m >= row_hi) return;
    const size_t frow = flow_row ? flow_row[m] : (size_t)m;
    const float *fl = ((e & 1) ? disp1 : disp2) + 18 * frow;
    uint4 q;
    if (chunk < 4) {
        if (ALIGNED8) {
            const uint2 a = reinterpret_cast<const uint2 *>(fl)[2 * chunk], b = reinterpret_cast<const uint2 *>(fl)[2 * chunk + 1];
            q.x = a.x; q.y = a.y; q.z = b.x; q.w = b.y;
        } else {
            q.x = __float_as_uint(fl[4 * chunk]); q.y = __float_as_uint(fl[4 * chunk + 1]);
            q.z = __float_as_uint(fl[4 * chunk + 2]); q.w = __float_as_uint(fl[4 * chunk + 3]);
        }
    } else {
        q.x = __float_as_uint(fl[16]); q.y = __float_as_uint(fl[17]);
        q.z = __float_as_uint(sim[m]);
        if (words) q.w = words[p];                                // (the fused gather's word array holds exactly this: k_edge_words)
        else {
            uint32_t s, d;
            edge_ends(node1, node2, e, s, d);
            const uint32_t ls = local_of[s], ld = local_of[d];
            const uint32_t kind = track[s] != track[d] ? 1u : 0u;
            q.w = ls | ((ld | (kind << 15)) << 16);
        }
    }
    records[5 * p + chunk] = q;
}

// fused gather: the 4-byte word of every record (local source | (local destination | kind << 15) << 16)
__global__ void k_edge_words(int64_t cap, const uint32_t *total_edges_p, const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2,
                             const int32_t *track, const uint32_t *local_of, uint32_t *words) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap || p >= (int64_t)*total_edges_p) return;
    uint32_t s, d;
    edge_ends(node1, node2, edge_sorted[p], s, d);
    const uint32_t kind = track[s] != track[d] ? 1u : 0u;
    words[p] = local_of[s] | ((local_of[d] | (kind << 15)) << 16);
}

// Incidence lists of the workgroup-class components (the owner-computes assembly of solve_block_kernel walks a
// node's out-edge run and its in-edge list; the packed kernels use neither): degrees, run starts, and the
// (component, destination, edge index) keys of the in-edge sort.  Packed-class edges carry their key too (the sorted
// position of a key is its position in the batch's edge array: solve_block_kernel indexes in_idx from the
// component's edge offset); only the padding behind total_edges gets a key that sorts last.
__global__ void k_incidence(int64_t cap, const uint32_t *total_edges_p, const uint64_t *keys_sorted, int node_bits, const uint32_t *words,
                            const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2, const uint32_t *class_of_desc,
                            const uint32_t *edge_off, const uint32_t *node_off, const uint32_t *local_of, NodeInc *inc, uint64_t *in_keys, uint32_t *in_vals) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap) return;
    in_keys[p] = 0x0000ffffffffffffull; in_vals[p] = 0u;
    const int64_t total = (int64_t)*total_edges_p;
    if (p >= total) return;
    // component and source node come with the sorted key (component << node_bits | source, k_edge_keys), the local ids with the record
    // words when the fused gather wrote them: no walk back through the match arrays (six dependent gathers per edge were 0.8 ms of
    // config 5's assembly, with or without the atomics that used to count the runs)
    const uint64_t key = keys_sorted[p];
    const uint32_t di = (uint32_t)(key >> node_bits);
    uint32_t ls, ld;
    if (words) { const uint32_t w = words[p]; ls = w & 0xffffu; ld = (w >> 16) & 0x7fffu; }
    else { uint32_t s, d; edge_ends(node1, node2, edge_sorted[p], s, d); ls = local_of[s]; ld = local_of[d]; }
    const uint32_t eo = edge_off[di], no = node_off[di];
    const uint32_t local_edge = (uint32_t)p - eo;
    in_keys[p] = ((uint64_t)di << 16) | ld;            // in-edge lists: by component, destination, edge index
    in_vals[p] = local_edge;
    if (class_of_desc[di] < (uint32_t)KC_BLOCK) return;
    // a node's out-edges are a run of this list (workgroup classes: sorted by component, then source): the first of the run writes
    // where it begins, the last where it ends - k_inc_counts turns the ends into counts (the in-edge runs: k_in_begin)
    const bool first = p == 0 || keys_sorted[p - 1] != key, last = p + 1 >= total || keys_sorted[p + 1] != key;
    if (first) inc[no + ls].out_begin = local_edge;
    if (last) inc[no + ls].out_count = local_edge + 1u;
}

// (every component, packed classes included - as the host builder does, lfr_graph.cpp: in_begin / in_count of its counting sort.  The OUT
// words of packed components stay zero here while the host counts them: NodeInc is read by the workgroup kernels only (solve_component,
// solve_tree_component); the packed kernel walks its record words - ADVICE r4)
__global__ void k_in_begin(int64_t cap, const uint32_t *total_edges_p, const uint64_t *in_keys_sorted, const uint32_t *edge_off, const uint32_t *node_off,
                           NodeInc *inc) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap || p >= (int64_t)*total_edges_p) return;
    const uint64_t k = in_keys_sorted[p];
    if (k == 0x0000ffffffffffffull) return;
    const bool first = p == 0 || in_keys_sorted[p - 1] != k, last = p + 1 >= (int64_t)*total_edges_p || in_keys_sorted[p + 1] != k;
    if (first || last) {
        const uint32_t di = (uint32_t)(k >> 16), ld = (uint32_t)(k & 0xffffu);
        if (first) inc[node_off[di] + ld].in_begin = (uint32_t)p - edge_off[di];
        if (last) inc[node_off[di] + ld].in_count = (uint32_t)p + 1u - edge_off[di];
    }
}
// ends of the runs -> lengths (nodes without out- or in-edges hold zeros in both words)
__global__ void k_inc_counts(int64_t n, NodeInc *inc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NodeInc v = inc[i];
    v.out_count -= v.out_begin; v.in_count -= v.in_begin;
    inc[i] = v;
}

__global__ void k_fill_descs(int64_t cap, const uint32_t *class_sorted, const uint32_t *perm, const uint32_t *edge_off, const uint32_t *node_off,
                             const uint32_t *c_nodes, const uint32_t *c_var, const uint32_t *c_edges, const uint32_t *c_tracks,
                             CompDesc *descs, uint32_t *desc_tracks, uint32_t *desc_class, uint32_t *desc_component) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint32_t c = perm[i];
    desc_class[i] = class_sorted[i]; desc_component[i] = c;       // the device copies behind the lazily fetched host mirrors (two copy launches before)
    if (class_sorted[i] == kNoClass) return;
    CompDesc d;
    d.edge_off = edge_off[i]; d.n_edges = c_edges[c]; d.node_off = node_off[i];
    d.n_nodes = (uint16_t)c_nodes[c]; d.n_var = (uint16_t)c_var[c];
    descs[i] = d;
    desc_tracks[i] = c_tracks[c];
}

// sharded batches: the i-th solvable component in batch order goes to shard snake_shard(i); the others are
// marked unsolvable (class 7) for this shard and drop out of everything downstream
__global__ void k_shard_class(int64_t n_comp, const uint32_t *class_sorted, int shard_rank, int shard_world, uint32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_comp) return;
    const uint32_t c = class_sorted[i];
    out[i] = (c != kNoClass && snake_shard(i, shard_world) == shard_rank) ? c : kNoClass;
}

// class ranges, solved tracks, largest workgroup-class systems, per-edge scratch sizes.  Sums and maxima are
// reduced over the workgroup first: one atomic per workgroup instead of 147 k atomics on one address (1.7 ms -> 10 us).
__global__ void k_summary(int64_t n_comp, const uint32_t *class_sorted, const uint32_t *perm, const uint32_t *c_var,
                          const uint32_t *c_edges, const uint32_t *c_tracks, AsmSummary *sum, unsigned long long *es_size) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tracks = 0, rows_wg = 0;
    int my_class = KC_COUNT;
    if (i <= n_comp) {
        const int prev = i > 0 ? min((int)class_sorted[i - 1], (int)KC_COUNT) : -1;
        const int cur = min(i < n_comp ? (int)class_sorted[i] : KC_COUNT, (int)KC_COUNT);   // (unsolved components: behind every class)
        for (int kc = prev + 1; kc <= cur; ++kc) sum->class_begin[kc] = (uint32_t)i;     // first index with class >= kc
        if (i < n_comp) {
            unsigned long long es = 0ull;
            if (cur != KC_COUNT) {
                const uint32_t c = perm[i];
                tracks = c_tracks[c];
                if (cur >= KC_BLOCK) { rows_wg = 2u * c_var[c]; es = 8ull * c_edges[c]; }   // 64 B of Jacobian scratch per edge
            }
            es_size[i] = es;
            my_class = cur;
        }
    }
    // (one atomic per workgroup: on one address they take their turns at the L2, ~2-3 ns each)
    __shared__ uint32_t s_tracks[kPipeThreads / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tracks += __shfl_xor(tracks, m, 64);
    if ((threadIdx.x & 63) == 0) s_tracks[threadIdx.x >> 6] = tracks;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kPipeThreads / 64; ++w) tracks += s_tracks[w];
        if (tracks) atomicAdd(&sum->n_tracks, tracks);
    }
    // largest system per workgroup class (sizes the launch's LDS): few such components, one atomic each
    if (rows_wg) atomicMax(&sum->class_max_rows[my_class], rows_wg);
}
// global-matrix class: packed lower triangle + the vectors of the largest system of the class, per component
__global__ void k_ws_sizes(int64_t n_comp, const uint32_t *class_sorted, const uint32_t *perm, const uint32_t *c_var,
                           const AsmSummary *sum, unsigned long long *ws_size) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_comp) return;
    unsigned long long v = 0ull;
    if (class_sorted[i] == (uint32_t)KC_GLOBAL) {
        const unsigned long long rows = 2ull * c_var[perm[i]], mat = rows * (rows + 1) / 2;
        v = mat + (mat & 1ull) + (10ull * sum->class_max_rows[KC_GLOBAL] + 4ull);    // block_vector_doubles(largest global system)
    }
    ws_size[i] = v;
}
// es_scan / ws_scan: exclusive scans over C+1 entries (the last entry holds the total)
__global__ void k_offsets(int64_t n_comp, const uint32_t *class_sorted, const unsigned long long *es_scan, const unsigned long long *ws_scan,
                          const uint32_t *node_off, const uint32_t *edge_off, AsmSummary *sum, uint64_t *es_off, uint64_t *ws_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_comp) return;
    if (i == n_comp) {
        sum->es_doubles = es_scan[n_comp]; sum->ws_doubles = ws_scan[n_comp];
        sum->packed_edges = edge_off[min((int64_t)sum->class_begin[KC_BLOCK], n_comp)];
        sum->n_desc = sum->class_begin[KC_COUNT];
        sum->total_nodes = node_off[n_comp]; sum->total_edges = edge_off[n_comp];
        for (int kc = 0; kc < KC_COUNT; ++kc)           // descs are sorted by class: a class is one range of the edge scan
            sum->class_edges[kc] = (uint64_t)(edge_off[sum->class_begin[kc + 1]] - edge_off[sum->class_begin[kc]]);
        return;
    }
    es_off[i] = es_scan[i];
    ws_off[i] = class_sorted[i] == (uint32_t)KC_GLOBAL ? es_scan[n_comp] + ws_scan[i] : 0ull;
}

struct ArenaMark {                                 // temporaries: released at scope end (reuse is stream ordered)
    DevArena &a; size_t m;
    explicit ArenaMark(DevArena &ar) : a(ar), m(ar.top) {}
    ~ArenaMark() { a.top = m; }
};
#define TAKE(ptr, T, count)                                                                                   \
    T *ptr = arena.take_n<T>((size_t)(count));                                                                \
    if (!ptr) { set_error("assembly: device arena exhausted (%s)", #ptr); return LFR_ERR_NOMEM; }
#define TAKE_OUT(dst, T, count)                                                                               \
    dst = slab.take_n<T>((size_t)(count));                                                                    \
    if (!dst) { set_error("assembly: batch slab exhausted (%s)", #dst); return LFR_ERR_NOMEM; }

template <class K, class V>
int sort_pairs(DevArena &arena, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int begin_bit, int end_bit, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    LFR_HIP_TRY(sort_pairs_raw(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st));
    ArenaMark mark(arena);
    void *tmp = arena.take(bytes);
    if (!tmp) { set_error("assembly: device arena exhausted (sort of %lld items)", (long long)n); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(sort_pairs_raw(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st));
    return LFR_OK;
}
template <class T>
int exclusive_sum(DevArena &arena, const T *in, T *out, int64_t n, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    LFR_HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, std::remove_cv_t<std::remove_reference_t<decltype(*out)>>(0), (size_t)n, rocprim::plus<std::remove_cv_t<std::remove_reference_t<decltype(*out)>>>(), st));
    ArenaMark mark(arena);
    void *tmp = arena.take(bytes);
    if (!tmp) { set_error("assembly: device arena exhausted (scan of %lld items)", (long long)n); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(rocprim::exclusive_scan(tmp, bytes, in, out, std::remove_cv_t<std::remove_reference_t<decltype(*out)>>(0), (size_t)n, rocprim::plus<std::remove_cv_t<std::remove_reference_t<decltype(*out)>>>(), st));
    return LFR_OK;
}

}  // namespace

int warm_assembly_primitives(DevCtx *ctx) {
    hipStream_t st = ctx->s_main;
    LFR_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t sizes[4] = {100000, 300000, 1000000, (int64_t)4 << 20};   // (merge-sort levels and radix paths differ by size)
    DevArena arena;
    if (!arena.init(ctx, (size_t)sizes[3] * 64 + ((size_t)64 << 20))) return LFR_ERR_NOMEM;
    const int64_t nmax = sizes[3];
    uint64_t *k64a = arena.take_n<uint64_t>(nmax), *k64b = arena.take_n<uint64_t>(nmax);
    uint32_t *k32a = arena.take_n<uint32_t>(nmax), *k32b = arena.take_n<uint32_t>(nmax), *v32a = arena.take_n<uint32_t>(nmax), *v32b = arena.take_n<uint32_t>(nmax);
    if (!k64a || !k64b || !k32a || !k32b || !v32a || !v32b) { set_error("warm-up arena exhausted"); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(hipMemsetAsync(k64a, 0x5a, 8 * (size_t)nmax, st));
    LFR_HIP_TRY(hipMemsetAsync(k32a, 0x3c, 4 * (size_t)nmax, st));
    LFR_HIP_TRY(hipMemsetAsync(v32a, 0, 4 * (size_t)nmax, st));
    int rc = LFR_OK;
    for (const int64_t n : sizes) {
        if ((rc = sort_pairs(arena, k32a, k32b, v32a, v32b, n, 0, 20, st)) != LFR_OK) return rc;
        if ((rc = sort_pairs(arena, k64a, k64b, v32a, v32b, n, 0, 44, st)) != LFR_OK) return rc;
        if ((rc = sort_pairs(arena, reinterpret_cast<unsigned long long *>(k64a), reinterpret_cast<unsigned long long *>(k64b), v32a, v32b, n, 0, 52, st)) != LFR_OK) return rc;
        if ((rc = exclusive_sum(arena, v32a, v32b, n, st)) != LFR_OK) return rc;
        if ((rc = exclusive_sum(arena, reinterpret_cast<unsigned long long *>(k64a), reinterpret_cast<unsigned long long *>(k64b), n, st)) != LFR_OK) return rc;
    }
    LFR_HIP_TRY(stream_wait(st));
    return rc;
}

size_t assembly_output_bytes(int64_t N, int64_t M, int64_t C) {
    const size_t E2 = (size_t)2 * M, n = (size_t)N, c = (size_t)C + 1;
    return sizeof(CompDesc) * c + sizeof(EdgeRec) * E2 + 4 * n + sizeof(NodeInc) * n + 4 * E2 + 8 * E2 + 16 * c + 12 * c + 256 * 18;
}

static inline int nbits(uint64_t x) { int b = 1; while (x >>= 1) ++b; return b; }      // bits needed for values 0..x

__global__ void k_fill_regions(FillRegions r) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (int j = 0; j < r.n; ++j) {
        const unsigned int b = r.byte_value[j] * 0x01010101u;
        const size_t n16 = r.bytes[j] / 16, tail = r.bytes[j] - 16 * n16;
        uint4 *q = static_cast<uint4 *>(r.ptr[j]);
        for (size_t i = tid; i < n16; i += stride) q[i] = make_uint4(b, b, b, b);
        if (tid < tail) static_cast<unsigned char *>(r.ptr[j])[16 * n16 + tid] = (unsigned char)r.byte_value[j];
    }
}

int assemble_on_device(const Problem &p, const DevProblem &dp, int shard_rank, int shard_world, DevArena &slab, DeviceAssembly &out) {
    const DevGraph &dg = *dp.graph;
    DevCtx *ctx = dp.ctx;
    hipStream_t st = ctx->s_main;
    const int64_t N = dg.N, M = dg.M, E2 = 2 * M;
    const int64_t C = p.stats.n_components, T = p.stats.n_tracks;
    if (E2 >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31)) { set_error("graph too large for the device assembly"); return LFR_ERR_UNSUPPORTED; }
    LFR_HIP_TRY(hipSetDevice(ctx->device));

    DevArena arena;                                   // temporaries of this call
    if (!arena.init(ctx, (size_t)96 * M + (size_t)52 * N + (size_t)128 * (C + 1) + ((size_t)32 << 20))) return LFR_ERR_NOMEM;
    size_t pin_bytes = 0;
    AsmSummary *h_sum = (AsmSummary *)ctx->pinned_acquire(sizeof(AsmSummary), &pin_bytes);
    if (!h_sum) return LFR_ERR_NOMEM;
    struct PinGuard { DevCtx *c; void *p; size_t b; ~PinGuard() { c->pinned_release(p, b); } } pin_guard{ctx, h_sum, pin_bytes};

    const uint32_t *node1 = dg.n1, *node2 = dg.n2;
    const int32_t *track = dp.track, *comp = dp.comp;
    const int node_bits = nbits((uint64_t)std::max<int64_t>(N - 1, 0)), comp_bits = nbits((uint64_t)C);

    // ---- outputs (capacities are upper bounds: every edge kept, every node in a solved component) ----
    TAKE_OUT(out.d_descs, CompDesc, C + 1); TAKE_OUT(out.d_edges, EdgeRec, E2);
    TAKE_OUT(out.d_node_ids, uint32_t, N); TAKE_OUT(out.d_node_inc, NodeInc, N); TAKE_OUT(out.d_in_idx, uint32_t, E2);
    TAKE_OUT(out.d_ws_off, uint64_t, C + 1); TAKE_OUT(out.d_es_off, uint64_t, C + 1);
    TAKE_OUT(out.d_desc_class, uint32_t, C + 1); TAKE_OUT(out.d_desc_tracks, uint32_t, C + 1); TAKE_OUT(out.d_desc_component, uint32_t, C + 1);
    const bool aligned8 = (((uintptr_t)dg.disp1 | (uintptr_t)dg.disp2) & 7u) == 0;
    // Fused gather: a whole batch over flows that live in HBM lets the packed kernel read its edges from the graph's own arrays
    // (72-byte flow rows, 8-byte aligned) - the 400 MB of records for config 4 are neither written nor re-read.  Shards gather zero-copy
    // from pinned host memory (once, into records); LFR_FUSED_GATHER=0 switches the path off.
    // Caller-owned device flows (lfr_graph_from_arrays_device_flows) are only promised to live until the batch exists (lfr.h): such a
    // batch gathers its records now and never looks at the caller's memory again.
    bool fused = shard_world == 1 && !dg.flows_zero_copy && !dg.flows_external && aligned8 && dg.disp1 && dg.disp2;
    if (const char *e = getenv("LFR_FUSED_GATHER")) fused = fused && e[0] != '0';
    out.fused = fused;
    if (fused) { TAKE_OUT(out.d_edge_ref, uint32_t, E2); TAKE_OUT(out.d_edge_word, uint32_t, E2); }

    // ---- which edges are kept, which nodes are variables, per-component sizes ----
    // (everything that starts at zero sits in one block: one memset instead of a dozen)
    const size_t zero_mark = arena.top;
    TAKE(sum, AsmSummary, 1);
    TAKE(opt, uint8_t, N);
    TAKE(cn, uint32_t, C + 1); TAKE(cv, uint32_t, C + 1); TAKE(ce, uint32_t, C + 1); TAKE(ct, uint32_t, C + 1);
    TAKE(ts, uint32_t, T + 1); TAKE(dn, uint32_t, C + 1); TAKE(de, uint32_t, C + 1);
    TAKE(ws_size, unsigned long long, C + 1);
    TAKE(run_begin, uint32_t, C + 1); TAKE(run_end, uint32_t, C + 1);       // (k_match_runs)
    TAKE(node_begin, uint32_t, C + 1); TAKE(node_var_end, uint32_t, C + 1); TAKE(node_end, uint32_t, C + 1);      // (k_node_runs)
    const size_t scan_words = scan_state_words(C + 1);                      // look-back states of the four prefix sums below (exclusive_sum_one_launch)
    TAKE(scan_state, unsigned long long, 4 * scan_words);
    {   FillRegions fr; fr.add(arena.base + zero_mark, arena.top - zero_mark, 0); LFR_HIP_TRY(fill_regions(fr, st)); }
    TAKE(kept, uint8_t, E2); TAKE(is_var, uint8_t, N); TAKE(tc, int32_t, T + 1);
    TAKE(nk0, uint32_t, N); TAKE(nk1, uint32_t, N); TAKE(ni0, uint32_t, N); TAKE(ni1, uint32_t, N); TAKE(local, uint32_t, N);
    // Packed classes only (the graph stage's largest component says that no workgroup class can exist): the matches go through their
    // sort now, by component id, and the runs are the edge counts (k_match_keys_comp).  LFR_EDGE_SORT_BY_EDGE / LFR_MATCH_SORT_LATE keep
    // the older orders for A/B.
    TAKE(ek0, uint64_t, E2); TAKE(ek1, uint64_t, E2); TAKE(ei0, uint32_t, E2); TAKE(ei1, uint32_t, E2);      // edge keys / ids of the edge order below
    if (fused) ei1 = out.d_edge_ref;                  // the sorted edge ids ARE the packed kernel's gather list
    const bool expect_workgroup_classes = p.stats.max_component_size > 17;
    const bool match_sort_first = !expect_workgroup_classes && !getenv("LFR_EDGE_SORT_BY_EDGE") && !getenv("LFR_MATCH_SORT_LATE");
    // The nodes go through THEIR sort before the counts as well - by (component, variables first): the runs are the components' node and
    // variable counts (k_mark_var's two atomics per node on 147 k counters were half of its 54 us), a node's position in its run is its
    // local index.  LFR_NODE_SORT_LATE keeps the older order (by the component's rank in the batch) for A/B.
    const bool node_sort_first = !getenv("LFR_NODE_SORT_LATE");
    uint32_t *mkey_sorted = nullptr, *match_sorted = nullptr;
    // (keys and ids of the M matches: the four quarters of the first 64-bit edge-key buffer)
    uint32_t *mk0 = reinterpret_cast<uint32_t *>(ek0), *mk1 = mk0 + M, *mi0 = mk1 + M, *mi1 = mi0 + M;
    int rc;
    hipLaunchKernelGGL(k_mark_kept, grid_for(E2), dim3(kThreads), 0, st, M, node1, node2, track, comp, kept, opt, (uint32_t)C,
                       match_sort_first ? mk0 : nullptr);
    hipLaunchKernelGGL(k_mark_var, grid_for(N), dim3(kThreads), 0, st, N, opt, dp.is_root, track, comp, is_var, cn, cv, ts, tc,
                       node_sort_first ? nk0 : nullptr, ni0);
    if (node_sort_first) {
        if ((rc = sort_pairs(arena, nk0, nk1, ni0, ni1, N, 0, std::min(32, nbits((uint64_t)2 * C)), st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_node_runs, grid_for(N), dim3(kThreads), 0, st, N, nk1, node_begin, node_var_end, node_end);
    }
    if (match_sort_first) {
        hipLaunchKernelGGL(k_match_keys_comp, grid_for(M), dim3(kThreads), 0, st, M, node1, node2, is_var, kept, (uint32_t)C, mk0, mi0);
        if ((rc = sort_pairs(arena, mk0, mk1, mi0, mi1, M, 0, comp_bits, st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_match_runs, grid_for(M), dim3(kThreads), 0, st, M, mk1, (uint32_t)C, run_begin, run_end);
        mkey_sorted = mk1; match_sorted = mi1;
    } else if (M > 256 * C) {                         // (see k_count_edges_by_node)
        TAKE(ne, uint32_t, N);
        LFR_HIP_TRY(hipMemsetAsync(ne, 0, 4 * (size_t)N, st));
        hipLaunchKernelGGL(k_count_edges_by_node, grid_for(M), dim3(kThreads), 0, st, M, node1, node2, is_var, kept, ne);
        hipLaunchKernelGGL(k_sum_node_edges, grid_for(N), dim3(kThreads), 0, st, N, comp, ne, ce);
    } else {
        hipLaunchKernelGGL(k_count_edges, grid_for(M), dim3(kThreads), 0, st, M, node1, node2, comp, is_var, kept, ce);
    }

    // ---- batch order of the components: class, then edges descending, then variables descending, then id ----
    TAKE(key64, unsigned long long, C + 1); TAKE(key64s, unsigned long long, C + 1);
    TAKE(id0, uint32_t, C + 1); TAKE(id1, uint32_t, C + 1); TAKE(k0, uint32_t, C + 1); TAKE(k1, uint32_t, C + 1);
    // field widths of the key from what the host knows: a component's matches are matches of ONE connected component of the match graph
    // (dp.max_cc_matches, from the graph stage; 0 = unknown: 32 bits), its variables are nodes of the component (stats.max_component_size)
    uint32_t e_max = 0xffffffffu, v_max = 0xffffu;
    if (dp.max_cc_matches > 0 && dp.max_cc_matches < (1u << 30)) e_max = 2u * dp.max_cc_matches;
    if (dp.max_cc_matches > 0 && p.stats.max_component_size > 0 && p.stats.max_component_size < 0xffff) v_max = (uint32_t)p.stats.max_component_size;   // (the same stage counted both)
    int e_bits = 1, v_bits = 1;
    while (e_bits < 32 && (e_max >> e_bits) != 0u) ++e_bits;
    while (v_bits < 16 && (v_max >> v_bits) != 0u) ++v_bits;
    const int class_shift = e_bits + v_bits;
    hipLaunchKernelGGL(k_comp_keys, grid_for(std::max(C, T)), dim3(kThreads), 0, st, C, cn, cv, ce, match_sort_first ? run_begin : nullptr, run_end,
                       node_sort_first ? node_begin : nullptr, node_var_end, node_end, key64, id0, &sum->too_big, (uint32_t)block_max_rows(),
                       T, ts, tc, ct, e_max, v_max, e_bits, v_bits);
    if ((rc = sort_pairs(arena, key64, key64s, id0, id1, C, 0, class_shift + kClassBits, st)) != LFR_OK) return rc;
    uint32_t *perm = id1;                  // perm[i] = component of desc i
    uint32_t *class_sorted = k1;
    if (shard_world > 1) {                 // keep this shard's components (same relative order), the rest becomes class 7
        hipLaunchKernelGGL(k_class_of_key, grid_for(C), dim3(kThreads), 0, st, C, key64s, k1, class_shift);
        hipLaunchKernelGGL(k_shard_class, grid_for(C), dim3(kThreads), 0, st, C, class_sorted, shard_rank, shard_world, k0);
        if ((rc = sort_pairs(arena, k0, k1, id1, id0, C, 0, kClassBits, st)) != LFR_OK) return rc;
        perm = id0; class_sorted = k1;     // (k1 is rewritten by the sort after k_shard_class has read it: stream ordered)
    }

    TAKE(no, uint32_t, C + 1); TAKE(eo, uint32_t, C + 1); TAKE(di, int32_t, C + 1);
    hipLaunchKernelGGL(k_desc_sizes, grid_for(C), dim3(kThreads), 0, st, C, perm, class_sorted, shard_world > 1 ? nullptr : key64s, cn, ce, dn, de, di, class_shift);
    LFR_HIP_TRY(exclusive_sum_one_launch(dn, no, C + 1, scan_state + 0 * scan_words, st));
    LFR_HIP_TRY(exclusive_sum_one_launch(de, eo, C + 1, scan_state + 1 * scan_words, st));
    const uint32_t *total_nodes_p = no + C, *total_edges_p = eo + C;

    // ---- launch geometry + workspace offsets ----
    TAKE(es_size, unsigned long long, C + 1); TAKE(es_scan, unsigned long long, C + 1); TAKE(ws_scan, unsigned long long, C + 1);
    hipLaunchKernelGGL(k_summary, grid_for(C + 1), dim3(kThreads), 0, st, C, class_sorted, perm, cv, ce, ct, sum, es_size);
    hipLaunchKernelGGL(k_ws_sizes, grid_for(C), dim3(kThreads), 0, st, C, class_sorted, perm, cv, sum, ws_size);
    LFR_HIP_TRY(exclusive_sum_one_launch(es_size, es_scan, C + 1, scan_state + 2 * scan_words, st));
    LFR_HIP_TRY(exclusive_sum_one_launch(ws_size, ws_scan, C + 1, scan_state + 3 * scan_words, st));
    hipLaunchKernelGGL(k_offsets, grid_for(C + 1), dim3(kThreads), 0, st, C, class_sorted, es_scan, ws_scan, no, eo, sum, out.d_es_off, out.d_ws_off);

    // ---- local node numbering: nodes by (desc, variable first, node id) ----
    if (node_sort_first) {
        hipLaunchKernelGGL(k_place_nodes, grid_for(N), dim3(kThreads), 0, st, N, nk1, ni1, di, no, node_begin, out.d_node_ids, local);
    } else {
        hipLaunchKernelGGL(k_node_keys, grid_for(N), dim3(kThreads), 0, st, N, comp, di, is_var, (uint32_t)(2 * C), nk0, ni0);
        if ((rc = sort_pairs(arena, nk0, nk1, ni0, ni1, N, 0, std::min(32, nbits((uint64_t)2 * C)), st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_node_locals, grid_for(N), dim3(kThreads), 0, st, N, total_nodes_p, ni1, comp, di, no, out.d_node_ids, local);
    }

    // ---- edge order: kept edges by (desc, source node, edge id); packed classes by (desc, edge id) ----
    // Packed classes carry zeros in the source-node bits (their order is component, then edge id - the sort is stable): when the graph
    // stage's largest component says that no workgroup class can exist, only the component bits are sorted - three radix passes over the
    // 5 M keys of config 4 instead of five.  A small component with > 320 edges still lands in a workgroup class: the summary below has
    // the last word and the full sort is redone then.
    bool words_done = false;                          // the record words came with the match-level order
    if (match_sort_first) {                           // (sorted before the counts: every match to its component's place)
        hipLaunchKernelGGL(k_place_matches, grid_for(M), dim3(kThreads), 0, st, M, mkey_sorted, match_sorted, (uint32_t)C, di, eo, run_begin, node1, node2,
                           local, ei1, fused ? out.d_edge_word : nullptr);
        words_done = fused;
    } else if (expect_workgroup_classes) {
        hipLaunchKernelGGL(k_edge_keys, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, comp, di, class_sorted, kept, node_bits,
                           (uint64_t)C << node_bits, ek0, ei0);
        if ((rc = sort_pairs(arena, ek0, ek1, ei0, ei1, E2, 0, node_bits + comp_bits, st)) != LFR_OK) return rc;
    } else if (!getenv("LFR_EDGE_SORT_BY_EDGE")) {    // (32-bit keys in the front halves of the 64-bit key buffers; one key per MATCH: k_match_keys_packed)
        uint32_t *k32a = reinterpret_cast<uint32_t *>(ek0), *k32b = reinterpret_cast<uint32_t *>(ek1);
        uint32_t *mi0 = ei0, *mi1 = ei0 + M;
        hipLaunchKernelGGL(k_match_keys_packed, grid_for(M), dim3(kThreads), 0, st, M, node1, comp, di, kept, (uint32_t)C, k32a, mi0);
        if ((rc = sort_pairs(arena, k32a, k32b, mi0, mi1, M, 0, comp_bits, st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_expand_match_order, grid_for(M), dim3(kThreads), 0, st, M, total_edges_p, mi1, node1, node2, track, local,
                           ei1, fused ? out.d_edge_word : nullptr);
        words_done = fused;
    } else {                                          // (the same per directed edge: rounds 3-4, kept for A/B)
        uint32_t *k32a = reinterpret_cast<uint32_t *>(ek0), *k32b = reinterpret_cast<uint32_t *>(ek1);
        hipLaunchKernelGGL(k_edge_keys_packed, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, comp, di, kept, (uint32_t)C, k32a, ei0);
        if ((rc = sort_pairs(arena, k32a, k32b, ei0, ei1, E2, 0, comp_bits, st)) != LFR_OK) return rc;
    }
    // (both directions of a match are kept or dropped together by construction - k_count_edges - and the sort is stable on the edge
    // id, so the pair check only runs on request or on the path that materialises records)
    if (!fused || getenv("LFR_CHECK_PAIRS"))
        hipLaunchKernelGGL(k_check_pairs, grid_for(E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2, comp, di, class_sorted, eo, &sum->unpaired);
    if (fused && !words_done) hipLaunchKernelGGL(k_edge_words, grid_for(E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2, track, local, out.d_edge_word);

    // ---- descriptors + the device copies behind the lazily fetched host mirrors ----
    hipLaunchKernelGGL(k_fill_descs, grid_for(C), dim3(kThreads), 0, st, C, class_sorted, perm, eo, no, cn, cv, ce, ct, out.d_descs, out.d_desc_tracks,
                       out.d_desc_class, out.d_desc_component);

    // ---- incidence lists: workgroup classes only.  The graph stage's largest component tells whether such a class can
    // exist (packed classes take up to 16 variable nodes); when it says no, the lists are skipped and the summary
    // below has the last word (a small component with > 320 edges - duplicated matches - still lands there).
    auto build_incidence = [&]() -> int {
        LFR_HIP_TRY(hipMemsetAsync(out.d_node_inc, 0, std::max<size_t>(sizeof(NodeInc) * (size_t)N, 16), st));
        hipLaunchKernelGGL(k_incidence, grid_for(E2), dim3(kThreads), 0, st, E2, total_edges_p, ek1, node_bits, fused ? out.d_edge_word : nullptr,
                           ei1, node1, node2, class_sorted, eo, no, local, out.d_node_inc, ek0, ei0);      // (the unsorted key buffers are free again)
        // (component << 16 | destination: 16 + comp_bits bits; the padding key's ones in that range exceed every real component index)
        const int r = sort_pairs(arena, ek0, ek1, ei0, out.d_in_idx, E2, 0, 16 + comp_bits, st);
        if (r != LFR_OK) return r;
        hipLaunchKernelGGL(k_in_begin, grid_for(E2), dim3(kThreads), 0, st, E2, total_edges_p, ek1, eo, no, out.d_node_inc);
        hipLaunchKernelGGL(k_inc_counts, grid_for(N), dim3(kThreads), 0, st, N, out.d_node_inc);
        return LFR_OK;
    };
    if (expect_workgroup_classes && (rc = build_incidence()) != LFR_OK) return rc;

    // ---- records: gather the flows (their first consumer); staged flows arrive in chunks on the copy stream ----
    // Every launch walks the whole edge list and keeps the edges of its rows, so chunks whose upload has already finished
    // (all of them when the graph is resident) go out as ONE launch: four filtered passes were 40 % of the gather.
    // A connected-component shard (dg.parent) carries the PARENT's upload events and chunk bounds, but its match indices are compacted:
    // match m of the shard reads flow row flow_row[m] >= m, which can lie in a later chunk than chunk_row[] says for m.
    // It waits for every chunk and emits in one pass.
    const bool shard = (bool)dg.parent;
    const bool chunked = dg.flows_staged && !shard;
    const int n_chunks = chunked ? kFlowChunks : 1;
    const bool emit = !fused || expect_workgroup_classes;          // (fused and no workgroup class expected: nothing to write; the summary has the last word)
    if ((!emit || shard) && dg.flows_staged)                          // the solve reads the flows themselves / a shard gathers any row: every chunk must have landed
        for (int c = 0; c < kFlowChunks; ++c) if (dg.ev_flows[c]) LFR_HIP_TRY(hipStreamWaitEvent(st, dg.ev_flows[c], 0));
    for (int c = 0; emit && c < n_chunks;) {
        int c2 = c + 1;
        if (chunked) {
            auto landed = [&](int k) {
                if (!dg.ev_flows[k]) return true;
                const hipError_t q = hipEventQuery(dg.ev_flows[k]);
                if (q != hipSuccess && q != hipErrorNotReady) (void)hipGetLastError();
                return q == hipSuccess;
            };
            if (landed(c)) while (c2 < n_chunks && landed(c2)) ++c2;
            else if (dg.ev_flows[c]) LFR_HIP_TRY(hipStreamWaitEvent(st, dg.ev_flows[c], 0));
        }
        const int64_t lo = chunked ? dg.chunk_row[c] : 0, hi = chunked ? dg.chunk_row[c2] : M;
        if (hi > lo || (c == 0 && n_chunks == 1)) {
            if (aligned8)
                hipLaunchKernelGGL(k_emit_edges<true>, grid_for(5 * E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2,
                                   dg.sim, dg.disp1, dg.disp2, track, local, dg.flow_row, lo, hi, reinterpret_cast<uint4 *>(out.d_edges), fused ? &sum->packed_edges : nullptr, fused ? out.d_edge_word : nullptr);
            else
                hipLaunchKernelGGL(k_emit_edges<false>, grid_for(5 * E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2,
                                   dg.sim, dg.disp1, dg.disp2, track, local, dg.flow_row, lo, hi, reinterpret_cast<uint4 *>(out.d_edges), fused ? &sum->packed_edges : nullptr, fused ? out.d_edge_word : nullptr);
        }
        c = c2;
    }
    LFR_HIP_TRY(hipGetLastError());

    // the one read-back of the stage
    LFR_HIP_TRY(hipMemcpyAsync(h_sum, sum, sizeof(AsmSummary), hipMemcpyDeviceToHost, st));
    LFR_HIP_TRY(stream_wait(st));
    out.summary = *h_sum;
    if (out.summary.too_big) { set_error("a component exceeds the 32767-node batch limit"); return LFR_ERR_UNSUPPORTED; }
    if (out.summary.unpaired) { set_error("internal: a kept edge without its opposite direction"); return LFR_ERR_UNSUPPORTED; }
    if (!expect_workgroup_classes && out.summary.class_begin[KC_BLOCK] < out.summary.n_desc) {
        // the unexpected workgroup classes need their edges by source node: the 64-bit keys and the full sort (packed classes keep their
        // order), the words again
        hipLaunchKernelGGL(k_edge_keys, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, comp, di, class_sorted, kept, node_bits,
                           (uint64_t)C << node_bits, ek0, ei0);
        if ((rc = sort_pairs(arena, ek0, ek1, ei0, ei1, E2, 0, node_bits + comp_bits, st)) != LFR_OK) return rc;
        if (fused) hipLaunchKernelGGL(k_edge_words, grid_for(E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2, track, local, out.d_edge_word);
        if ((rc = build_incidence()) != LFR_OK) return rc;
        {                                         // the records again, in the new order (fused: the unexpected workgroup classes' only)
            if (dg.flows_staged) for (int c = 0; c < n_chunks; ++c) if (dg.ev_flows[c]) LFR_HIP_TRY(hipStreamWaitEvent(st, dg.ev_flows[c], 0));
            if (aligned8)
                hipLaunchKernelGGL(k_emit_edges<true>, grid_for(5 * E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2,
                                   dg.sim, dg.disp1, dg.disp2, track, local, dg.flow_row, (int64_t)0, M, reinterpret_cast<uint4 *>(out.d_edges), &sum->packed_edges, fused ? out.d_edge_word : nullptr);
            else
                hipLaunchKernelGGL(k_emit_edges<false>, grid_for(5 * E2), dim3(kThreads), 0, st, E2, total_edges_p, ei1, node1, node2,
                                   dg.sim, dg.disp1, dg.disp2, track, local, dg.flow_row, (int64_t)0, M, reinterpret_cast<uint4 *>(out.d_edges), &sum->packed_edges, fused ? out.d_edge_word : nullptr);
        }
        LFR_HIP_TRY(stream_wait(st));         // the temporaries go back to the cache at return
    }
    return LFR_OK;
}

}  // namespace lfr

namespace {
template <class K>
int debug_sort(int64_t n, const void *keys, const uint32_t *vals, int begin_bit, int end_bit, int use_library, void *keys_out, uint32_t *vals_out) {
    K *kin = nullptr, *kout = nullptr;
    uint32_t *vin = nullptr, *vout = nullptr;
    void *tmp = nullptr;
    size_t bytes = 0;
    int rc = LFR_OK;
    auto body = [&]() -> int {
        LFR_HIP_TRY(hipMalloc(&kin, sizeof(K) * n));
        LFR_HIP_TRY(hipMalloc(&kout, sizeof(K) * n));
        LFR_HIP_TRY(hipMalloc(&vin, 4 * n));
        LFR_HIP_TRY(hipMalloc(&vout, 4 * n));
        LFR_HIP_TRY(hipMemcpy(kin, keys, sizeof(K) * n, hipMemcpyHostToDevice));
        LFR_HIP_TRY(hipMemcpy(vin, vals, 4 * n, hipMemcpyHostToDevice));
        if (use_library) {
            LFR_HIP_TRY(rocprim::radix_sort_pairs<LfrRadixSortConfig>(nullptr, bytes, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, nullptr));
            LFR_HIP_TRY(hipMalloc(&tmp, bytes));
            LFR_HIP_TRY(rocprim::radix_sort_pairs<LfrRadixSortConfig>(tmp, bytes, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, nullptr));
        } else {
            LFR_HIP_TRY(lfr::sort_pairs_raw(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, nullptr));
            LFR_HIP_TRY(hipMalloc(&tmp, bytes));
            LFR_HIP_TRY(hipMemset(tmp, 0xff, bytes));          // the driver must clear what it relies on
            LFR_HIP_TRY(lfr::sort_pairs_raw(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, nullptr));
        }
        LFR_HIP_TRY(hipDeviceSynchronize());
        LFR_HIP_TRY(hipMemcpy(keys_out, kout, sizeof(K) * n, hipMemcpyDeviceToHost));
        LFR_HIP_TRY(hipMemcpy(vals_out, vout, 4 * n, hipMemcpyDeviceToHost));
        return LFR_OK;
    };
    rc = body();
    (void)hipFree(kin); (void)hipFree(kout); (void)hipFree(vin); (void)hipFree(vout); (void)hipFree(tmp);
    return rc;
}
}  // namespace

template <class T>
static int debug_scan(int64_t n, const void *in, void *out) {
    T *din = nullptr, *dout = nullptr;
    unsigned long long *state = nullptr;
    auto body = [&]() -> int {
        LFR_HIP_TRY(hipMalloc(&din, sizeof(T) * n));
        LFR_HIP_TRY(hipMalloc(&dout, sizeof(T) * n));
        LFR_HIP_TRY(hipMalloc(&state, 8 * lfr::scan_state_words(n)));
        LFR_HIP_TRY(hipMemcpy(din, in, sizeof(T) * n, hipMemcpyHostToDevice));
        LFR_HIP_TRY(hipMemset(dout, 0xee, sizeof(T) * n));
        LFR_HIP_TRY(hipMemset(state, 0, 8 * lfr::scan_state_words(n)));
        LFR_HIP_TRY(lfr::exclusive_sum_one_launch(din, dout, n, state, nullptr));
        LFR_HIP_TRY(hipDeviceSynchronize());
        LFR_HIP_TRY(hipMemcpy(out, dout, sizeof(T) * n, hipMemcpyDeviceToHost));
        return LFR_OK;
    };
    const int rc = body();
    (void)hipFree(din); (void)hipFree(dout); (void)hipFree(state);
    return rc;
}
extern "C" int lfr_debug_exclusive_sum(int device, int64_t n, int item_bytes, const void *in, void *out) {
    if (n <= 0 || !in || !out || (item_bytes != 4 && item_bytes != 8)) { lfr::set_error("lfr_debug_exclusive_sum: bad arguments"); return LFR_ERR_ARG; }
    if (hipSetDevice(device) != hipSuccess) { lfr::set_error("lfr_debug_exclusive_sum: no device %d", device); return LFR_ERR_HIP; }
    return item_bytes == 4 ? debug_scan<uint32_t>(n, in, out) : debug_scan<unsigned long long>(n, in, out);
}

extern "C" int lfr_debug_sort_pairs(int device, int64_t n, int key_bytes, const void *keys, const uint32_t *vals, int begin_bit, int end_bit,
                                    int use_library, void *keys_out, uint32_t *vals_out) {
    if (n <= 0 || !keys || !vals || !keys_out || !vals_out || (key_bytes != 4 && key_bytes != 8) || begin_bit < 0 || end_bit <= begin_bit ||
        end_bit > 8 * key_bytes) {
        lfr::set_error("lfr_debug_sort_pairs: bad arguments");
        return LFR_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { lfr::set_error("lfr_debug_sort_pairs: no device %d", device); return LFR_ERR_HIP; }
    return key_bytes == 4 ? debug_sort<uint32_t>(n, keys, vals, begin_bit, end_bit, use_library, keys_out, vals_out)
                          : debug_sort<unsigned long long>(n, keys, vals, begin_bit, end_bit, use_library, keys_out, vals_out);
}

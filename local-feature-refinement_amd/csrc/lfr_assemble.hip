// Device-side batch assembly (reference: create_and_solve_problem's problem construction,
// solve.cc:94-143, and the largest-first ordering of solve.cc:599-604).
//
// Input: the match graph (per-match endpoints, similarity, two 3x3x2 flow grids) and the host
// graph stage's per-node labels (track, component, is_root).  Output, built entirely on the GPU
// and bit-identical to the host assembly of lfr_graph.cpp: CompDesc[], 80-byte EdgeRec[] in the
// reference's residual-block order, node_ids[], NodeInc[], in_idx[] — sorted by kernel class, then
// edge count descending.  The 2 x 180 MB of flows cross PCIe once, in match order, and are gathered
// into edge records at HBM speed; no 400 MB host-side record array is ever built.
//
// Everything is integer/byte work: radix sorts (hipCUB), scans, histograms with integer atomics
// (exact and order-independent), gathers.  HBM-bound; no LDS tiling to speak of.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "lfr_assemble.hpp"

namespace lfr {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                         \
            return LFR_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

namespace {

constexpr int kThreads = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + kThreads - 1) / kThreads)); }

__device__ __forceinline__ void edge_ends(const uint32_t *node1, const uint32_t *node2, int64_t e, uint32_t &src, uint32_t &dst) {
    const int64_t m = e >> 1;
    const uint32_t a = node1[m], b = node2[m];
    src = (e & 1) ? b : a;      // directed edge 2m = node1->node2, 2m+1 = node2->node1 (solve.cc:477-478)
    dst = (e & 1) ? a : b;
}

// kept = same track or same component (solve.cc:105,114); marks nodes with a kept out-edge
__global__ void k_mark_kept(int64_t n_dir, const uint32_t *node1, const uint32_t *node2, const int32_t *track,
                            const int32_t *comp, uint8_t *kept, uint8_t *opt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_dir) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const bool k = track[s] == track[d] || comp[s] == comp[d];
    kept[e] = k;
    if (k) opt[s] = 1;
}

__global__ void k_mark_var(int64_t n_nodes, const uint8_t *opt, const uint8_t *is_root, const int32_t *track, const int32_t *comp,
                           uint8_t *is_var, uint32_t *c_nodes, uint32_t *c_var, uint32_t *t_size, int32_t *t_comp) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const bool v = opt[n] && !is_root[n];                    // solve.cc:127,133-141
    is_var[n] = v;
    atomicAdd(&c_nodes[comp[n]], 1u);
    if (v) atomicAdd(&c_var[comp[n]], 1u);
    atomicAdd(&t_size[track[n]], 1u);
    t_comp[track[n]] = comp[n];
}

__global__ void k_count_edges(int64_t n_dir, const uint32_t *node1, const uint32_t *node2, const int32_t *comp,
                              const uint8_t *is_var, uint8_t *kept, uint32_t *c_edges) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_dir) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const bool k = kept[e] && (is_var[s] || is_var[d]);      // both ends constant: not in the reduced program
    kept[e] = k;
    if (k) atomicAdd(&c_edges[comp[s]], 1u);
}

__global__ void k_count_tracks(int64_t n_tracks, const uint32_t *t_size, const int32_t *t_comp, uint32_t *c_tracks) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tracks) return;
    if (t_size[t] >= 2) atomicAdd(&c_tracks[t_comp[t]], 1u);
}

__device__ __forceinline__ int classify_dev(uint32_t rows, uint32_t n_edges) {
    if (rows <= 8 && n_edges <= 24) return KC_G8;
    if (rows <= 16 && n_edges <= 96) return KC_G16;
    if (rows <= 24 && n_edges <= 192) return KC_G64_2;
    if (rows <= 32 && n_edges <= 320) return KC_G64_4;
    if (rows <= (uint32_t)kBlockMaxRows) return KC_BLOCK;
    return KC_GLOBAL;
}

// per component: solvable? class; the three sort keys of the batch order
__global__ void k_comp_keys(int64_t n_comp, const uint32_t *c_nodes, const uint32_t *c_var, const uint32_t *c_edges,
                            uint32_t *key_var, uint32_t *key_edges, uint32_t *key_class, uint32_t *ids, int *too_big) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_comp) return;
    const bool solvable = c_nodes[c] >= 2 && c_var[c] >= 1;   // solve.cc:619-622; no variable: nothing to solve
    if (solvable && c_nodes[c] > 32767) *too_big = 1;
    key_var[c] = 0xffffu - min(c_var[c], 0xffffu);             // descending
    key_edges[c] = 0xffffffffu - c_edges[c];                   // descending
    key_class[c] = solvable ? (uint32_t)classify_dev(2 * c_var[c], c_edges[c]) : 7u;
    ids[c] = (uint32_t)c;
}

__global__ void k_gather_u32(int64_t n, const uint32_t *idx, const uint32_t *src, uint32_t *dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// after the sort: per desc sizes (0 for unsolvable) + inverse permutation
__global__ void k_desc_sizes(int64_t n_comp, const uint32_t *perm, const uint32_t *key_class_sorted, const uint32_t *c_nodes,
                             const uint32_t *c_edges, uint32_t *d_nodes, uint32_t *d_edges, int32_t *di_of_comp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_comp) return;
    const uint32_t c = perm[i];
    const bool solvable = key_class_sorted[i] != 7u;
    d_nodes[i] = solvable ? c_nodes[c] : 0u;
    d_edges[i] = solvable ? c_edges[c] : 0u;
    di_of_comp[c] = solvable ? (int32_t)i : -1;
}

__global__ void k_node_keys(int64_t n_nodes, const int32_t *comp, const int32_t *di_of_comp, const uint8_t *is_var,
                            uint32_t *keys, uint32_t *ids) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const int32_t di = di_of_comp[comp[n]];
    keys[n] = di < 0 ? 0xffffffffu : (((uint32_t)di << 1) | (is_var[n] ? 0u : 1u));     // variables first, then constants
    ids[n] = (uint32_t)n;
}

__global__ void k_node_locals(int64_t total_nodes, const uint32_t *node_sorted, const int32_t *comp, const int32_t *di_of_comp,
                              const uint32_t *node_off, uint32_t *node_ids, uint32_t *local_of) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total_nodes) return;
    const uint32_t n = node_sorted[p];
    node_ids[p] = n;
    local_of[n] = (uint32_t)p - node_off[di_of_comp[comp[n]]];
}

__global__ void k_edge_keys(int64_t n_dir, const uint32_t *node1, const uint32_t *node2, const int32_t *comp,
                            const int32_t *di_of_comp, const uint32_t *class_of_desc, const uint8_t *kept, uint64_t *keys,
                            uint32_t *ids) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_dir) return;
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const int32_t di = kept[e] ? di_of_comp[comp[s]] : -1;
    // workgroup classes: residual-block order (by source node, then edge id); packed classes: edge id only
    // (the sort is stable), so the two directions 2m, 2m+1 of a match become neighbouring records
    // (written with an early return + mask: the one-expression form `(di >= 0 && class < KC_BLOCK) ? 0 : s` was
    // miscompiled by hipcc 7.2 -O3 for gfx950 - the register holding s was reused before the select)
    if (di < 0) { keys[e] = ~0ull; ids[e] = (uint32_t)e; return; }
    const uint32_t by_source = class_of_desc[di] >= (uint32_t)KC_BLOCK ? 0xffffffffu : 0u;
    keys[e] = ((uint64_t)(uint32_t)di << 32) | (s & by_source);
    ids[e] = (uint32_t)e;
}

// packed classes: records 2i, 2i+1 of a component must be the two directions of one match (the solve
// kernel's pair exchange relies on it)
__global__ void k_check_pairs(int64_t total_edges, const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2,
                              const int32_t *comp, const int32_t *di_of_comp, const uint32_t *class_of_desc,
                              const uint32_t *edge_off, int *flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total_edges) return;
    const uint32_t e = edge_sorted[p];
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const uint32_t di = (uint32_t)di_of_comp[comp[s]];
    if (class_of_desc[di] >= (uint32_t)KC_BLOCK) return;
    const uint32_t local = (uint32_t)p - edge_off[di];
    const int64_t q = (local & 1u) ? p - 1 : p + 1;
    if (q < 0 || q >= total_edges || edge_sorted[q] != (e ^ 1u)) *flag = 1;
}

// one thread per (edge record, 16-byte chunk): writes EdgeRec, counts degrees, records run starts
__global__ void k_emit_edges(int64_t total_edges, const uint32_t *edge_sorted, const uint32_t *node1, const uint32_t *node2,
                             const float *sim, const float *disp1, const float *disp2, const int32_t *track,
                             const int32_t *comp, const int32_t *di_of_comp, const uint32_t *edge_off, const uint32_t *node_off,
                             const uint32_t *local_of, const uint32_t *flow_row, uint4 *records, NodeInc *inc, uint64_t *in_keys,
                             uint32_t *in_vals) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t / 5;
    const int chunk = (int)(t - 5 * p);
    if (p >= total_edges) return;
    const uint32_t e = edge_sorted[p];
    uint32_t s, d;
    edge_ends(node1, node2, e, s, d);
    const size_t frow = flow_row ? flow_row[e >> 1] : (size_t)(e >> 1);
    const float *fl = ((e & 1) ? disp1 : disp2) + 18 * frow;
    uint4 q;
    if (chunk < 4) {
        q.x = __float_as_uint(fl[4 * chunk]); q.y = __float_as_uint(fl[4 * chunk + 1]);
        q.z = __float_as_uint(fl[4 * chunk + 2]); q.w = __float_as_uint(fl[4 * chunk + 3]);
    } else {
        const uint32_t ls = local_of[s], ld = local_of[d];
        const uint32_t kind = track[s] != track[d] ? 1u : 0u;
        q.x = __float_as_uint(fl[16]); q.y = __float_as_uint(fl[17]);
        q.z = __float_as_uint(sim[e >> 1]);
        q.w = ls | ((ld | (kind << 15)) << 16);
        const uint32_t di = (uint32_t)di_of_comp[comp[s]];
        const uint32_t eo = edge_off[di], no = node_off[di];
        const uint32_t local_edge = (uint32_t)p - eo;
        atomicAdd(&inc[no + ls].out_count, 1u);
        atomicAdd(&inc[no + ld].in_count, 1u);
        bool first = p == 0;
        if (!first) {
            uint32_t ps, pd;
            edge_ends(node1, node2, edge_sorted[p - 1], ps, pd);
            first = ps != s;
        }
        if (first) inc[no + ls].out_begin = local_edge;
        in_keys[p] = ((uint64_t)di << 16) | ld;            // in-edge lists: by component, destination, edge index
        in_vals[p] = local_edge;
    }
    records[5 * p + chunk] = q;
}

__global__ void k_in_begin(int64_t total_edges, const uint64_t *in_keys_sorted, const uint32_t *edge_off, const uint32_t *node_off,
                           NodeInc *inc) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total_edges) return;
    const uint64_t k = in_keys_sorted[p];
    if (p == 0 || in_keys_sorted[p - 1] != k) {
        const uint32_t di = (uint32_t)(k >> 16), ld = (uint32_t)(k & 0xffffu);
        inc[node_off[di] + ld].in_begin = (uint32_t)p - edge_off[di];
    }
}

__global__ void k_fill_descs(int64_t n_desc, const uint32_t *perm, const uint32_t *edge_off, const uint32_t *node_off,
                             const uint32_t *c_nodes, const uint32_t *c_var, const uint32_t *c_edges, const uint32_t *c_tracks,
                             CompDesc *descs, uint32_t *desc_tracks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_desc) return;
    const uint32_t c = perm[i];
    CompDesc d;
    d.edge_off = edge_off[i]; d.n_edges = c_edges[c]; d.node_off = node_off[i];
    d.n_nodes = (uint16_t)c_nodes[c]; d.n_var = (uint16_t)c_var[c];
    descs[i] = d;
    desc_tracks[i] = c_tracks[c];
}

#define DEV_ALLOC(buf, bytes) HIP_TRY(dev_alloc(arena, buf, (size_t)(bytes), false))

template <class K, class V>
int sort_pairs(DevArena *arena, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int begin_bit, int end_bit, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, (int)n, begin_bit, end_bit, st));
    DevBuf tmp;
    HIP_TRY(dev_alloc(arena, tmp, bytes, true));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin, kout, vin, vout, (int)n, begin_bit, end_bit, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LFR_OK;
}

int exclusive_sum(DevArena *arena, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, st));
    DevBuf tmp;
    HIP_TRY(dev_alloc(arena, tmp, bytes, true));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LFR_OK;
}

}  // namespace

int assemble_on_device(const Graph &g, const Problem &p, hipStream_t st, const float *dev_disp1, const float *dev_disp2,
                       DeviceAssembly &out) {
    const int64_t N = g.n_nodes(), M = g.n_matches(), E2 = 2 * M;
    const int64_t C = p.stats.n_components, T = p.stats.n_tracks;
    if (E2 >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31)) { set_error("graph too large for the device assembly"); return LFR_ERR_UNSUPPORTED; }

    DevArena slab;                                   // declared first: the buffers below must die before it
    DevArena *arena = &slab;
    if (slab.init((size_t)240 * M + (size_t)64 * N + (size_t)64 * C + ((size_t)16 << 20)) != hipSuccess) { (void)hipGetLastError(); slab.base = nullptr; arena = nullptr; }

    // ---- uploads: endpoints, similarities, labels (small) and the flows (2 x 72 B per match) ----
    std::vector<int32_t> track32(N), comp32(N);
    for (int64_t i = 0; i < N; ++i) { track32[i] = (int32_t)p.track[i]; comp32[i] = (int32_t)p.comp[i]; }
    DevBuf b_n1, b_n2, b_sim, b_track, b_comp, b_root, b_d1, b_d2;
    DEV_ALLOC(b_n1, 4 * M); DEV_ALLOC(b_n2, 4 * M); DEV_ALLOC(b_sim, 4 * M);
    DEV_ALLOC(b_track, 4 * N); DEV_ALLOC(b_comp, 4 * N); DEV_ALLOC(b_root, N);
    HIP_TRY(hipMemcpyAsync(b_n1.p, g.m_node1.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_n2.p, g.m_node2.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_sim.p, g.m_sim.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_track.p, track32.data(), 4 * N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_comp.p, comp32.data(), 4 * N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_root.p, p.is_root.data(), N, hipMemcpyHostToDevice, st));
    const float *disp1 = dev_disp1 ? dev_disp1 : g.dev_disp1, *disp2 = dev_disp2 ? dev_disp2 : g.dev_disp2;
    DevBuf b_frow;
    const uint32_t *flow_row = nullptr;
    if (disp1 && disp2 && !g.m_flow_row.empty()) {      // caller-owned device flows, indexed by their original row
        DEV_ALLOC(b_frow, 4 * M);
        HIP_TRY(hipMemcpyAsync(b_frow.p, g.m_flow_row.data(), 4 * M, hipMemcpyHostToDevice, st));
        flow_row = b_frow.as<uint32_t>();
    }
    if (!disp1 || !disp2) {
        DEV_ALLOC(b_d1, 72 * M); DEV_ALLOC(b_d2, 72 * M);
        HIP_TRY(hipMemcpyAsync(b_d1.p, g.m_disp1.data(), 72 * M, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b_d2.p, g.m_disp2.data(), 72 * M, hipMemcpyHostToDevice, st));
        disp1 = b_d1.as<float>(); disp2 = b_d2.as<float>();
    }
    const uint32_t *node1 = b_n1.as<uint32_t>(), *node2 = b_n2.as<uint32_t>();
    const int32_t *track = b_track.as<int32_t>(), *comp = b_comp.as<int32_t>();

    // ---- which edges are kept, which nodes are variables, per-component sizes ----
    DevBuf b_kept, b_opt, b_var, b_cn, b_cv, b_ce, b_ct, b_ts, b_tc, b_flag;
    DEV_ALLOC(b_kept, E2); DEV_ALLOC(b_opt, N); DEV_ALLOC(b_var, N);
    DEV_ALLOC(b_cn, 4 * C); DEV_ALLOC(b_cv, 4 * C); DEV_ALLOC(b_ce, 4 * C); DEV_ALLOC(b_ct, 4 * C);
    DEV_ALLOC(b_ts, 4 * T); DEV_ALLOC(b_tc, 4 * T); DEV_ALLOC(b_flag, 4);
    HIP_TRY(hipMemsetAsync(b_opt.p, 0, N, st));
    HIP_TRY(hipMemsetAsync(b_cn.p, 0, 4 * C, st)); HIP_TRY(hipMemsetAsync(b_cv.p, 0, 4 * C, st));
    HIP_TRY(hipMemsetAsync(b_ce.p, 0, 4 * C, st)); HIP_TRY(hipMemsetAsync(b_ct.p, 0, 4 * C, st));
    HIP_TRY(hipMemsetAsync(b_ts.p, 0, 4 * T, st)); HIP_TRY(hipMemsetAsync(b_flag.p, 0, 4, st));
    hipLaunchKernelGGL(k_mark_kept, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, track, comp, b_kept.as<uint8_t>(), b_opt.as<uint8_t>());
    hipLaunchKernelGGL(k_mark_var, grid_for(N), dim3(kThreads), 0, st, N, b_opt.as<uint8_t>(), b_root.as<uint8_t>(), track, comp,
                       b_var.as<uint8_t>(), b_cn.as<uint32_t>(), b_cv.as<uint32_t>(), b_ts.as<uint32_t>(), b_tc.as<int32_t>());
    hipLaunchKernelGGL(k_count_edges, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, comp, b_var.as<uint8_t>(), b_kept.as<uint8_t>(), b_ce.as<uint32_t>());
    hipLaunchKernelGGL(k_count_tracks, grid_for(T), dim3(kThreads), 0, st, T, b_ts.as<uint32_t>(), b_tc.as<int32_t>(), b_ct.as<uint32_t>());

    // ---- batch order of the components: class, then edges descending, then variables descending, then id ----
    DevBuf b_kv, b_ke, b_kc, b_id0, b_id1, b_k0, b_k1;
    DEV_ALLOC(b_kv, 4 * C); DEV_ALLOC(b_ke, 4 * C); DEV_ALLOC(b_kc, 4 * C);
    DEV_ALLOC(b_id0, 4 * C); DEV_ALLOC(b_id1, 4 * C); DEV_ALLOC(b_k0, 4 * C); DEV_ALLOC(b_k1, 4 * C);
    hipLaunchKernelGGL(k_comp_keys, grid_for(C), dim3(kThreads), 0, st, C, b_cn.as<uint32_t>(), b_cv.as<uint32_t>(), b_ce.as<uint32_t>(),
                       b_kv.as<uint32_t>(), b_ke.as<uint32_t>(), b_kc.as<uint32_t>(), b_id0.as<uint32_t>(), b_flag.as<int>());
    int rc;
    // LSD over the three keys (each pass stable): variables, edges, class
    if ((rc = sort_pairs(arena, b_kv.as<uint32_t>(), b_k0.as<uint32_t>(), b_id0.as<uint32_t>(), b_id1.as<uint32_t>(), C, 0, 16, st)) != LFR_OK) return rc;
    hipLaunchKernelGGL(k_gather_u32, grid_for(C), dim3(kThreads), 0, st, C, b_id1.as<uint32_t>(), b_ke.as<uint32_t>(), b_k0.as<uint32_t>());
    if ((rc = sort_pairs(arena, b_k0.as<uint32_t>(), b_k1.as<uint32_t>(), b_id1.as<uint32_t>(), b_id0.as<uint32_t>(), C, 0, 32, st)) != LFR_OK) return rc;
    hipLaunchKernelGGL(k_gather_u32, grid_for(C), dim3(kThreads), 0, st, C, b_id0.as<uint32_t>(), b_kc.as<uint32_t>(), b_k0.as<uint32_t>());
    if ((rc = sort_pairs(arena, b_k0.as<uint32_t>(), b_k1.as<uint32_t>(), b_id0.as<uint32_t>(), b_id1.as<uint32_t>(), C, 0, 3, st)) != LFR_OK) return rc;
    const uint32_t *perm = b_id1.as<uint32_t>();             // perm[i] = component of desc i
    const uint32_t *class_sorted = b_k1.as<uint32_t>();

    DevBuf b_dn, b_de, b_no, b_eo, b_di;
    DEV_ALLOC(b_dn, 4 * (C + 1)); DEV_ALLOC(b_de, 4 * (C + 1)); DEV_ALLOC(b_no, 4 * (C + 1)); DEV_ALLOC(b_eo, 4 * (C + 1)); DEV_ALLOC(b_di, 4 * C);
    HIP_TRY(hipMemsetAsync(b_dn.p, 0, 4 * (C + 1), st)); HIP_TRY(hipMemsetAsync(b_de.p, 0, 4 * (C + 1), st));
    hipLaunchKernelGGL(k_desc_sizes, grid_for(C), dim3(kThreads), 0, st, C, perm, class_sorted, b_cn.as<uint32_t>(), b_ce.as<uint32_t>(),
                       b_dn.as<uint32_t>(), b_de.as<uint32_t>(), b_di.as<int32_t>());
    if ((rc = exclusive_sum(arena, b_dn.as<uint32_t>(), b_no.as<uint32_t>(), C + 1, st)) != LFR_OK) return rc;
    if ((rc = exclusive_sum(arena, b_de.as<uint32_t>(), b_eo.as<uint32_t>(), C + 1, st)) != LFR_OK) return rc;
    uint32_t totals[2];
    int too_big = 0;
    HIP_TRY(hipMemcpy(&totals[0], b_no.as<uint32_t>() + C, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&totals[1], b_eo.as<uint32_t>() + C, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&too_big, b_flag.p, 4, hipMemcpyDeviceToHost));
    if (too_big) { set_error("a component exceeds the 32767-node batch limit"); return LFR_ERR_UNSUPPORTED; }
    const int64_t total_nodes = totals[0], total_edges = totals[1];
    std::vector<uint32_t> h_class(C);
    HIP_TRY(hipMemcpy(h_class.data(), class_sorted, 4 * C, hipMemcpyDeviceToHost));
    int64_t n_desc = 0;
    while (n_desc < C && h_class[n_desc] != 7u) ++n_desc;

    // ---- local node numbering: nodes by (desc, variable first, node id) ----
    DevBuf b_nk0, b_nk1, b_ni0, b_ni1, b_local;
    DEV_ALLOC(b_nk0, 4 * N); DEV_ALLOC(b_nk1, 4 * N); DEV_ALLOC(b_ni0, 4 * N); DEV_ALLOC(b_ni1, 4 * N); DEV_ALLOC(b_local, 4 * N);
    hipLaunchKernelGGL(k_node_keys, grid_for(N), dim3(kThreads), 0, st, N, comp, b_di.as<int32_t>(), b_var.as<uint8_t>(), b_nk0.as<uint32_t>(), b_ni0.as<uint32_t>());
    if ((rc = sort_pairs(arena, b_nk0.as<uint32_t>(), b_nk1.as<uint32_t>(), b_ni0.as<uint32_t>(), b_ni1.as<uint32_t>(), N, 0, 32, st)) != LFR_OK) return rc;
    HIP_TRY(hipMalloc(&out.d_node_ids, std::max<size_t>(4 * total_nodes, 16)));
    hipLaunchKernelGGL(k_node_locals, grid_for(total_nodes), dim3(kThreads), 0, st, total_nodes, b_ni1.as<uint32_t>(), comp, b_di.as<int32_t>(),
                       b_no.as<uint32_t>(), out.d_node_ids, b_local.as<uint32_t>());

    // ---- edge order: kept edges by (desc, source node, edge id); packed classes by (desc, edge id) ----
    DevBuf b_ek0, b_ek1, b_ei0, b_ei1;
    DEV_ALLOC(b_ek0, 8 * E2); DEV_ALLOC(b_ek1, 8 * E2); DEV_ALLOC(b_ei0, 4 * E2); DEV_ALLOC(b_ei1, 4 * E2);
    hipLaunchKernelGGL(k_edge_keys, grid_for(E2), dim3(kThreads), 0, st, E2, node1, node2, comp, b_di.as<int32_t>(), class_sorted, b_kept.as<uint8_t>(),
                       b_ek0.as<uint64_t>(), b_ei0.as<uint32_t>());
    if ((rc = sort_pairs(arena, b_ek0.as<uint64_t>(), b_ek1.as<uint64_t>(), b_ei0.as<uint32_t>(), b_ei1.as<uint32_t>(), E2, 0, 64, st)) != LFR_OK) return rc;

    HIP_TRY(hipMemsetAsync(b_flag.p, 0, 4, st));
    hipLaunchKernelGGL(k_check_pairs, grid_for(total_edges), dim3(kThreads), 0, st, total_edges, b_ei1.as<uint32_t>(), node1, node2, comp,
                       b_di.as<int32_t>(), class_sorted, b_eo.as<uint32_t>(), b_flag.as<int>());
    {
        int unpaired = 0;
        HIP_TRY(hipMemcpyAsync(&unpaired, b_flag.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (unpaired) { set_error("internal: a kept edge without its opposite direction"); return LFR_ERR_UNSUPPORTED; }
    }

    // ---- records, degrees, in-edge lists ----
    HIP_TRY(hipMalloc(&out.d_edges, std::max<size_t>(sizeof(EdgeRec) * total_edges, 16)));
    HIP_TRY(hipMalloc(&out.d_node_inc, std::max<size_t>(sizeof(NodeInc) * total_nodes, 16)));
    HIP_TRY(hipMalloc(&out.d_in_idx, std::max<size_t>(4 * total_edges, 16)));
    HIP_TRY(hipMemsetAsync(out.d_node_inc, 0, std::max<size_t>(sizeof(NodeInc) * total_nodes, 16), st));
    uint64_t *in_k0 = b_ek0.as<uint64_t>(), *in_k1 = b_ek1.as<uint64_t>();   // reuse the key buffers
    uint32_t *in_v0 = b_ei0.as<uint32_t>();
    hipLaunchKernelGGL(k_emit_edges, grid_for(5 * total_edges), dim3(kThreads), 0, st, total_edges, b_ei1.as<uint32_t>(), node1, node2,
                       b_sim.as<float>(), disp1, disp2, track, comp, b_di.as<int32_t>(), b_eo.as<uint32_t>(), b_no.as<uint32_t>(),
                       b_local.as<uint32_t>(), flow_row, reinterpret_cast<uint4 *>(out.d_edges), out.d_node_inc, in_k0, in_v0);
    if ((rc = sort_pairs(arena, in_k0, in_k1, in_v0, out.d_in_idx, total_edges, 0, 48, st)) != LFR_OK) return rc;
    hipLaunchKernelGGL(k_in_begin, grid_for(total_edges), dim3(kThreads), 0, st, total_edges, in_k1, b_eo.as<uint32_t>(), b_no.as<uint32_t>(), out.d_node_inc);

    // ---- descriptors ----
    DevBuf b_dt;
    DEV_ALLOC(b_dt, 4 * std::max<int64_t>(n_desc, 1));
    HIP_TRY(hipMalloc(&out.d_descs, std::max<size_t>(sizeof(CompDesc) * n_desc, 16)));
    hipLaunchKernelGGL(k_fill_descs, grid_for(n_desc), dim3(kThreads), 0, st, n_desc, perm, b_eo.as<uint32_t>(), b_no.as<uint32_t>(),
                       b_cn.as<uint32_t>(), b_cv.as<uint32_t>(), b_ce.as<uint32_t>(), b_ct.as<uint32_t>(), out.d_descs, b_dt.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));

    // ---- host mirrors the batch needs (descriptors, component ids, node ids) ----
    out.descs.resize(n_desc); out.desc_component.resize(n_desc); out.desc_class.resize(n_desc); out.desc_tracks.resize(n_desc);
    out.node_ids.resize(total_nodes);
    std::vector<uint32_t> h_perm(n_desc), h_tracks(n_desc);
    if (n_desc) {
        HIP_TRY(hipMemcpy(out.descs.data(), out.d_descs, sizeof(CompDesc) * n_desc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_perm.data(), perm, 4 * n_desc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_tracks.data(), b_dt.p, 4 * n_desc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out.node_ids.data(), out.d_node_ids, 4 * total_nodes, hipMemcpyDeviceToHost));
    }
    for (int64_t i = 0; i < n_desc; ++i) {
        out.desc_component[i] = h_perm[i]; out.desc_class[i] = (int32_t)h_class[i]; out.desc_tracks[i] = (int32_t)h_tracks[i];
    }
    out.n_edges = total_edges; out.n_nodes = total_nodes;
    return LFR_OK;
}

void DeviceAssembly::release() {
    if (d_descs) (void)hipFree(d_descs);
    if (d_edges) (void)hipFree(d_edges);
    if (d_node_ids) (void)hipFree(d_node_ids);
    if (d_node_inc) (void)hipFree(d_node_inc);
    if (d_in_idx) (void)hipFree(d_in_idx);
    d_descs = nullptr; d_edges = nullptr; d_node_ids = nullptr; d_node_inc = nullptr; d_in_idx = nullptr;
}

}  // namespace lfr

// Device-side graph stage: tracks (solve.cc:489-549), roots (solve.cc:552-582) and components
// (solve.cc:252-308) on the GPU, bit-identical to the host stage of lfr_graph.cpp.
//
// The constrained maximum spanning forest is greedy over the globally sorted match list, i.e.
// order dependent — but only INSIDE a connected component of the match graph: two matches of
// different connected components never interact.  So (hard part 4):
//   1. radix-sort the matches by (similarity, n1, n2) descending            (hipCUB)
//   2. plain connected components of the match graph, ignoring image conflicts (lock-free union-find)
//   3. stable-sort the ordered matches by connected component
//   4. one thread per connected component runs the reference's sequential union-find with the
//      image-conflict test over its own matches, in order
//   5. track ids = rank of the root nodes; roots = arg-max (score, node) per track; components =
//      connected components of the track meta-graph, numbered by their smallest track.
// Components above the size cap need the graph cut, and very large connected components would
// serialise step 4 on one thread: both cases return LFR_GRAPHSTAGE_USE_HOST and the caller runs the
// host stage instead.  Integer work throughout; the only floating-point accumulation (root scores,
// sums of float32 similarities in fp64) is exact for any realistic input, hence order independent.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
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
constexpr int64_t kMaxSegmentEdges = 1 << 16;     // larger connected components: host stage
inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + kThreads - 1) / kThreads)); }

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

__device__ __forceinline__ uint32_t sim_key(float s) {       // order-preserving float -> uint32
    if (s == 0.f) s = 0.f;                                    // -0.0 == +0.0
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// descending order of (sim, n1, n2) == ascending order of the complemented keys
__global__ void k_match_keys(int64_t M, const uint32_t *n1, const uint32_t *n2, const float *sim, uint64_t *k_hi, uint32_t *k_lo,
                             uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    k_hi[m] = ((uint64_t)(~sim_key(sim[m])) << 32) | (uint32_t)(~n1[m]);
    k_lo[m] = ~n2[m];
    ids[m] = (uint32_t)m;
}
__global__ void k_gather_u64(int64_t n, const uint32_t *idx, const uint64_t *src, uint64_t *dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_iota(int64_t n, uint32_t *p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// ---- lock-free union-find (hook the larger root under the smaller): labels = smallest member ----
// Every access to parent[] is an agent-scope atomic: a plain load may be served from the CU's
// non-coherent vector L1, and a stale "parent[a] == a" there makes the CAS loop spin forever.
__device__ __forceinline__ uint32_t uf_load(const uint32_t *parent, uint32_t x) {
    return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    uint32_t p = uf_load(parent, x);
    while (p != x) {                                           // path halving (parents only ever decrease)
        const uint32_t gp = uf_load(parent, p);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p; p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }      // a > b: hook a under b
        const uint32_t seen = atomicCAS(&parent[a], a, b);
        if (seen == a) return;
        a = seen;                                               // a was hooked meanwhile: continue from its new parent
    }
}
__global__ void k_cc_union(int64_t M, const uint32_t *n1, const uint32_t *n2, uint32_t *parent) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) uf_union(parent, n1[m], n2[m]);
}
__global__ void k_cc_flatten(int64_t n, uint32_t *parent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = uf_find(parent, (uint32_t)i);
}
__global__ void k_cc_keys(int64_t M, const uint32_t *order, const uint32_t *n1, const uint32_t *cc, uint32_t *keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) keys[i] = cc[n1[order[i]]];
}
__global__ void k_seg_flags(int64_t M, const uint32_t *keys, uint32_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_seg_starts(int64_t M, const uint32_t *flags, const uint32_t *seg_id, uint32_t *starts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M && flags[i]) starts[seg_id[i]] = (uint32_t)i;
}

// ---- the reference's greedy constrained union-find, one thread per connected component ----
// parent: -1 = root (solve.cc:492); lists of member nodes replace images_in_track (every member has
// a distinct image, so |images_in_track[root]| == count[root]).
__device__ __forceinline__ int32_t seq_root(int32_t *parent, int32_t i) {
    int32_t r = i;
    while (parent[r] >= 0) r = parent[r];
    while (parent[i] >= 0) { const int32_t nx = parent[i]; parent[i] = r; i = nx; }
    return r;
}
__global__ void k_kruskal(int64_t n_seg, const uint32_t *starts, int64_t M, const uint32_t *order, const uint32_t *n1,
                          const uint32_t *n2, const int32_t *node_image, int32_t *parent, int32_t *next, int32_t *tail,
                          int32_t *count) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int64_t lo = starts[s], hi = (s + 1 < n_seg) ? starts[s + 1] : M;
    for (int64_t k = lo; k < hi; ++k) {
        const uint32_t m = order[k];
        const int32_t r1 = seq_root(parent, (int32_t)n1[m]), r2 = seq_root(parent, (int32_t)n2[m]);
        if (r1 == r2) continue;
        bool conflict = false;                               // solve.cc:506-511
        for (int32_t i = r1; i >= 0 && !conflict; i = next[i]) {
            const int32_t im = node_image[i];
            for (int32_t j = r2; j >= 0; j = next[j]) if (node_image[j] == im) { conflict = true; break; }
        }
        if (conflict) continue;
        int32_t big = r1, small = r2;
        if (count[r1] < count[r2]) { big = r2; small = r1; }  // solve.cc:513-521 (ties: root2 under root1)
        parent[small] = big;
        next[tail[big]] = small; tail[big] = tail[small]; count[big] += count[small];
    }
}
__global__ void k_init_nodes(int64_t n, int32_t *parent, int32_t *next, int32_t *tail, int32_t *count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = -1; next[i] = -1; tail[i] = (int32_t)i; count[i] = 1; }
}
__global__ void k_root_flags(int64_t n, const int32_t *parent, uint32_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = parent[i] < 0 ? 1u : 0u;
}
__global__ void k_track_ids(int64_t n, const int32_t *parent, const uint32_t *rank, int32_t *track, uint32_t *tsize) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t r = (int32_t)i;
    while (parent[r] >= 0) r = parent[r];
    track[i] = (int32_t)rank[r];                              // solve.cc:528-541
    atomicAdd(&tsize[rank[r]], 1u);
}

// ---- roots: score = sum of similarities over intra-track out-edges; arg-max (score, node) ----
__global__ void k_scores(int64_t M, const uint32_t *n1, const uint32_t *n2, const float *sim, const int32_t *track, double *score) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t a = n1[m], b = n2[m];
    if (track[a] == track[b]) { atomicAdd(&score[a], (double)sim[m]); atomicAdd(&score[b], (double)sim[m]); }
}
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ void k_best_score(int64_t n, const int32_t *track, const double *score, unsigned long long *best) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(&best[track[i]], ordered_bits(score[i]));
}
__global__ void k_best_node(int64_t n, const int32_t *track, const double *score, const unsigned long long *best, int32_t *node) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ordered_bits(score[i]) == best[track[i]]) atomicMax(&node[track[i]], (int32_t)i);   // ties: larger node idx
}
__global__ void k_mark_roots(int64_t n_tracks, const int32_t *node, uint8_t *is_root) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tracks) is_root[node[t]] = 1;
}

// ---- components of the track meta-graph ----
__global__ void k_meta_union(int64_t M, const uint32_t *n1, const uint32_t *n2, const int32_t *track, uint32_t *parent) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int32_t ta = track[n1[m]], tb = track[n2[m]];
    if (ta != tb) uf_union(parent, (uint32_t)ta, (uint32_t)tb);
}
__global__ void k_comp_flags(int64_t n_tracks, const uint32_t *parent, uint32_t *flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tracks) flags[t] = parent[t] == (uint32_t)t ? 1u : 0u;      // representative = smallest track of the component
}
__global__ void k_comp_sizes(int64_t n_tracks, const uint32_t *parent, const uint32_t *rank, const uint32_t *tsize, uint32_t *csize) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tracks) atomicAdd(&csize[rank[parent[t]]], tsize[t]);
}
__global__ void k_node_comp(int64_t n, const int32_t *track, const uint32_t *parent, const uint32_t *rank, int32_t *comp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = (int32_t)rank[parent[track[i]]];
}
__global__ void k_max_u32(int64_t n, const uint32_t *v, uint32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(out, v[i]);
}

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int graph_stage_on_device(const Graph &g, int64_t max_nodes, int device, Problem &p) {
    using clock = std::chrono::steady_clock;
    const int64_t N = g.n_nodes(), M = g.n_matches();
    p.g = &g;
    p.track.assign(N, -1); p.comp.assign(N, -1); p.is_root.assign(N, 0);
    p.stats = lfr_problem_stats{};
    p.host_batch = false;
    if (N == 0) return LFR_OK;
    if (N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 30)) return LFR_GRAPHSTAGE_USE_HOST;
    if (max_nodes <= 0) max_nodes = (int64_t)g.image_names.size();
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) { set_error("HIP device %d not available (%d devices)", device, n_dev); return LFR_ERR_HIP; }
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    auto t0 = clock::now();
    DevArena slab;                                   // declared first: the buffers below must die before it
    DevArena *arena = &slab;
    if (slab.init((size_t)80 * M + (size_t)96 * N + ((size_t)16 << 20)) != hipSuccess) { (void)hipGetLastError(); slab.base = nullptr; arena = nullptr; }

    DevBuf b_n1, b_n2, b_sim, b_img;
    DEV_ALLOC(b_n1, 4 * M); DEV_ALLOC(b_n2, 4 * M); DEV_ALLOC(b_sim, 4 * M); DEV_ALLOC(b_img, 4 * N);
    HIP_TRY(hipMemcpyAsync(b_n1.p, g.m_node1.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_n2.p, g.m_node2.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_sim.p, g.m_sim.data(), 4 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_img.p, g.node_image.data(), 4 * N, hipMemcpyHostToDevice, st));
    const uint32_t *n1 = b_n1.as<uint32_t>(), *n2 = b_n2.as<uint32_t>();
    const float *sim = b_sim.as<float>();

    // 1. matches in the reference's order: descending (sim, n1, n2)
    DevBuf b_khi, b_khi2, b_klo, b_klo2, b_id0, b_id1;
    DEV_ALLOC(b_khi, 8 * M); DEV_ALLOC(b_khi2, 8 * M); DEV_ALLOC(b_klo, 4 * M); DEV_ALLOC(b_klo2, 4 * M);
    DEV_ALLOC(b_id0, 4 * M); DEV_ALLOC(b_id1, 4 * M);
    hipLaunchKernelGGL(k_match_keys, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, sim, b_khi.as<uint64_t>(), b_klo.as<uint32_t>(), b_id0.as<uint32_t>());
    int rc;
    if ((rc = sort_pairs(arena, b_klo.as<uint32_t>(), b_klo2.as<uint32_t>(), b_id0.as<uint32_t>(), b_id1.as<uint32_t>(), M, 0, 32, st)) != LFR_OK) return rc;
    hipLaunchKernelGGL(k_gather_u64, grid_for(M), dim3(kThreads), 0, st, M, b_id1.as<uint32_t>(), b_khi.as<uint64_t>(), b_khi2.as<uint64_t>());
    if ((rc = sort_pairs(arena, b_khi2.as<uint64_t>(), b_khi.as<uint64_t>(), b_id1.as<uint32_t>(), b_id0.as<uint32_t>(), M, 0, 64, st)) != LFR_OK) return rc;
    uint32_t *order = b_id0.as<uint32_t>();

    // 2. connected components of the match graph (conflicts ignored)
    DevBuf b_cc;
    DEV_ALLOC(b_cc, 4 * N);
    hipLaunchKernelGGL(k_iota, grid_for(N), dim3(kThreads), 0, st, N, b_cc.as<uint32_t>());
    hipLaunchKernelGGL(k_cc_union, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, b_cc.as<uint32_t>());
    hipLaunchKernelGGL(k_cc_flatten, grid_for(N), dim3(kThreads), 0, st, N, b_cc.as<uint32_t>());

    // 3. ordered matches grouped by connected component (stable)
    DevBuf b_ck0, b_ck1, b_flags, b_segid, b_starts;
    DEV_ALLOC(b_ck0, 4 * M); DEV_ALLOC(b_ck1, 4 * M); DEV_ALLOC(b_flags, 4 * (M + 1)); DEV_ALLOC(b_segid, 4 * (M + 1));
    hipLaunchKernelGGL(k_cc_keys, grid_for(M), dim3(kThreads), 0, st, M, order, n1, b_cc.as<uint32_t>(), b_ck0.as<uint32_t>());
    if ((rc = sort_pairs(arena, b_ck0.as<uint32_t>(), b_ck1.as<uint32_t>(), order, b_id1.as<uint32_t>(), M, 0, 32, st)) != LFR_OK) return rc;
    order = b_id1.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(b_flags.p, 0, 4 * (M + 1), st));
    hipLaunchKernelGGL(k_seg_flags, grid_for(M), dim3(kThreads), 0, st, M, b_ck1.as<uint32_t>(), b_flags.as<uint32_t>());
    if ((rc = exclusive_sum(arena, b_flags.as<uint32_t>(), b_segid.as<uint32_t>(), M + 1, st)) != LFR_OK) return rc;
    uint32_t n_seg = 0;
    HIP_TRY(hipMemcpy(&n_seg, b_segid.as<uint32_t>() + M, 4, hipMemcpyDeviceToHost));
    DEV_ALLOC(b_starts, 4 * ((int64_t)n_seg + 1));
    hipLaunchKernelGGL(k_seg_starts, grid_for(M), dim3(kThreads), 0, st, M, b_flags.as<uint32_t>(), b_segid.as<uint32_t>(), b_starts.as<uint32_t>());
    {   // a huge connected component would run on ONE thread: let the host do those inputs
        std::vector<uint32_t> h_starts(n_seg);
        if (n_seg) HIP_TRY(hipMemcpy(h_starts.data(), b_starts.p, 4 * (size_t)n_seg, hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < n_seg; ++s) {
            const int64_t len = (s + 1 < n_seg ? (int64_t)h_starts[s + 1] : M) - h_starts[s];
            if (len > kMaxSegmentEdges) return LFR_GRAPHSTAGE_USE_HOST;
        }
    }

    // 4. greedy constrained union-find per connected component
    DevBuf b_par, b_next, b_tail, b_cnt;
    DEV_ALLOC(b_par, 4 * N); DEV_ALLOC(b_next, 4 * N); DEV_ALLOC(b_tail, 4 * N); DEV_ALLOC(b_cnt, 4 * N);
    hipLaunchKernelGGL(k_init_nodes, grid_for(N), dim3(kThreads), 0, st, N, b_par.as<int32_t>(), b_next.as<int32_t>(), b_tail.as<int32_t>(), b_cnt.as<int32_t>());
    hipLaunchKernelGGL(k_kruskal, grid_for(n_seg), dim3(kThreads), 0, st, (int64_t)n_seg, b_starts.as<uint32_t>(), M, order, n1, n2,
                       b_img.as<int32_t>(), b_par.as<int32_t>(), b_next.as<int32_t>(), b_tail.as<int32_t>(), b_cnt.as<int32_t>());

    // 5. track ids (roots in ascending node index), sizes
    DevBuf b_rflag, b_rrank, b_track, b_tsize, b_max;
    DEV_ALLOC(b_rflag, 4 * (N + 1)); DEV_ALLOC(b_rrank, 4 * (N + 1)); DEV_ALLOC(b_track, 4 * N); DEV_ALLOC(b_max, 8);
    HIP_TRY(hipMemsetAsync(b_rflag.p, 0, 4 * (N + 1), st));
    hipLaunchKernelGGL(k_root_flags, grid_for(N), dim3(kThreads), 0, st, N, b_par.as<int32_t>(), b_rflag.as<uint32_t>());
    if ((rc = exclusive_sum(arena, b_rflag.as<uint32_t>(), b_rrank.as<uint32_t>(), N + 1, st)) != LFR_OK) return rc;
    uint32_t n_tracks = 0;
    HIP_TRY(hipMemcpy(&n_tracks, b_rrank.as<uint32_t>() + N, 4, hipMemcpyDeviceToHost));
    DEV_ALLOC(b_tsize, 4 * (int64_t)n_tracks);
    HIP_TRY(hipMemsetAsync(b_tsize.p, 0, 4 * (size_t)n_tracks, st));
    HIP_TRY(hipMemsetAsync(b_max.p, 0, 8, st));
    hipLaunchKernelGGL(k_track_ids, grid_for(N), dim3(kThreads), 0, st, N, b_par.as<int32_t>(), b_rrank.as<uint32_t>(), b_track.as<int32_t>(), b_tsize.as<uint32_t>());
    hipLaunchKernelGGL(k_max_u32, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_tsize.as<uint32_t>(), b_max.as<uint32_t>());
    HIP_TRY(hipStreamSynchronize(st));
    p.stats.tracks_ms = ms_since(t0);

    // roots
    t0 = clock::now();
    DevBuf b_score, b_best, b_bnode, b_root;
    DEV_ALLOC(b_score, 8 * N); DEV_ALLOC(b_best, 8 * (int64_t)n_tracks); DEV_ALLOC(b_bnode, 4 * (int64_t)n_tracks); DEV_ALLOC(b_root, N);
    HIP_TRY(hipMemsetAsync(b_score.p, 0, 8 * N, st));
    HIP_TRY(hipMemsetAsync(b_best.p, 0, 8 * (size_t)n_tracks, st));
    HIP_TRY(hipMemsetAsync(b_bnode.p, 0xff, 4 * (size_t)n_tracks, st));          // -1
    HIP_TRY(hipMemsetAsync(b_root.p, 0, N, st));
    hipLaunchKernelGGL(k_scores, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, sim, b_track.as<int32_t>(), b_score.as<double>());
    hipLaunchKernelGGL(k_best_score, grid_for(N), dim3(kThreads), 0, st, N, b_track.as<int32_t>(), b_score.as<double>(), b_best.as<unsigned long long>());
    hipLaunchKernelGGL(k_best_node, grid_for(N), dim3(kThreads), 0, st, N, b_track.as<int32_t>(), b_score.as<double>(), b_best.as<unsigned long long>(), b_bnode.as<int32_t>());
    hipLaunchKernelGGL(k_mark_roots, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_bnode.as<int32_t>(), b_root.as<uint8_t>());
    HIP_TRY(hipStreamSynchronize(st));
    p.stats.roots_ms = ms_since(t0);

    // components of the track meta-graph, numbered by their smallest track (solve.cc:292-300)
    t0 = clock::now();
    DevBuf b_mp, b_cflag, b_crank, b_csize, b_comp;
    DEV_ALLOC(b_mp, 4 * (int64_t)n_tracks); DEV_ALLOC(b_cflag, 4 * ((int64_t)n_tracks + 1)); DEV_ALLOC(b_crank, 4 * ((int64_t)n_tracks + 1));
    DEV_ALLOC(b_comp, 4 * N);
    hipLaunchKernelGGL(k_iota, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_mp.as<uint32_t>());
    hipLaunchKernelGGL(k_meta_union, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, b_track.as<int32_t>(), b_mp.as<uint32_t>());
    hipLaunchKernelGGL(k_cc_flatten, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_mp.as<uint32_t>());
    HIP_TRY(hipMemsetAsync(b_cflag.p, 0, 4 * ((size_t)n_tracks + 1), st));
    hipLaunchKernelGGL(k_comp_flags, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_mp.as<uint32_t>(), b_cflag.as<uint32_t>());
    if ((rc = exclusive_sum(arena, b_cflag.as<uint32_t>(), b_crank.as<uint32_t>(), (int64_t)n_tracks + 1, st)) != LFR_OK) return rc;
    uint32_t n_comp = 0;
    HIP_TRY(hipMemcpy(&n_comp, b_crank.as<uint32_t>() + n_tracks, 4, hipMemcpyDeviceToHost));
    DEV_ALLOC(b_csize, 4 * (int64_t)n_comp);
    HIP_TRY(hipMemsetAsync(b_csize.p, 0, 4 * (size_t)n_comp, st));
    hipLaunchKernelGGL(k_comp_sizes, grid_for(n_tracks), dim3(kThreads), 0, st, (int64_t)n_tracks, b_mp.as<uint32_t>(), b_crank.as<uint32_t>(),
                       b_tsize.as<uint32_t>(), b_csize.as<uint32_t>());
    hipLaunchKernelGGL(k_max_u32, grid_for(n_comp), dim3(kThreads), 0, st, (int64_t)n_comp, b_csize.as<uint32_t>(), b_max.as<uint32_t>() + 1);
    hipLaunchKernelGGL(k_node_comp, grid_for(N), dim3(kThreads), 0, st, N, b_track.as<int32_t>(), b_mp.as<uint32_t>(), b_crank.as<uint32_t>(), b_comp.as<int32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t maxes[2] = {0, 0};
    HIP_TRY(hipMemcpy(maxes, b_max.p, 8, hipMemcpyDeviceToHost));
    if ((int64_t)maxes[1] > max_nodes) return LFR_GRAPHSTAGE_USE_HOST;      // needs the graph cut (solve.cc:311-343)

    // labels back to the host (the Problem's public labels; the assembly re-uploads 10 MB)
    std::vector<int32_t> h_track(N), h_comp(N);
    HIP_TRY(hipMemcpy(h_track.data(), b_track.p, 4 * N, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_comp.data(), b_comp.p, 4 * N, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(p.is_root.data(), b_root.p, N, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) { p.track[i] = h_track[i]; p.comp[i] = h_comp[i]; }
    p.stats.n_tracks = n_tracks; p.stats.max_track_size = maxes[0];
    p.stats.n_components = n_comp; p.stats.max_component_size = maxes[1];
    p.stats.graph_cut_ms = ms_since(t0);
    return LFR_OK;
}

}  // namespace lfr

// Device pipeline between the parsed match graph and the solve kernels:
//   DevGraph    the match graph in HBM (endpoints, similarities, node images; flows staged, caller-owned or zero-copy)
//   DevProblem  tracks / roots / components in HBM (lfr_graphstage.hip)
//   device assembly of the batch layout (lfr_assemble.hip)
// One stream (DevCtx::s_main) carries the whole chain without host round trips: the graph stage ends with one
// 64-byte read-back (counts for the stdout lines + the host-fallback decision), the assembly with one
// (launch geometry of the solve kernels).  The flows travel on DevCtx::s_copy beside it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <memory>

#include "lfr_devctx.hpp"
#include "lfr_internal.hpp"

// rocPRIM directly (not through the hipCUB compatibility layer).  Radix sort: rocPRIM's default takes a merge sort (log2(n / block) launches of
// ~6 us) up to 2^20 keys - the node order of config 4 (0.88 M keys of 19 bits) was 21 launches, 150 us; from 256 K keys the one-sweep radix
// passes (one launch per 8 key bits) are the shorter road.  Only .hip translation units see this.
#ifdef __HIPCC__
#include <rocprim/rocprim.hpp>
#include "lfr_sort.hpp"          // LfrRadixSortConfig, sort_pairs_raw: one fill per radix sort instead of 1 + 2 per digit place
using lfr::LfrRadixSortConfig;
#endif

namespace lfr {

#define LFR_HIP_TRY(expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::lfr::set_error("%s failed: %s", #expr, hipGetErrorString(_e));                  \
            return LFR_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

constexpr int kFlowChunks = 4;
struct DevGraph {
    DevCtx *ctx = nullptr;
    int64_t N = 0, M = 0;
    DevArena slab;                       // endpoints, similarities, node images [, flow rows] [, staged flows]
    uint32_t *n1 = nullptr, *n2 = nullptr;
    float *sim = nullptr;
    int32_t *node_image = nullptr;
    uint32_t *flow_row = nullptr;        // caller-owned device flows are indexed by their original row
    const float *disp1 = nullptr, *disp2 = nullptr;   // flows in match order: HBM (staged / caller-owned) or pinned host (zero copy)
    bool flows_staged = false, flows_zero_copy = false, flows_external = false;
    bool endpoints_pending = false;      // created by prestage_flows while the scanner was still numbering nodes: n1/n2/sim/node_image not sent yet
    int64_t N_cap = 0;                   // nodes the slab has room for (prestage_flows sizes it from a bound)
    // staged flows travel in kFlowChunks chunks of matches [chunk_row[c], chunk_row[c+1]) on s_copy; ev_flows[c] fires
    // when chunk c has landed, so the assembly gathers a chunk while the next one is still on the wire
    hipEvent_t ev_flows[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t chunk_row[5] = {0, 0, 0, 0, 0};
    // A SHARD of another device graph (graph_stage_on_device with shard_world > 1): the matches of the connected components dealt to one
    // rank, in their original order, with flow_row pointing into the parent's flows; node images, flows and their events are the
    // parent's (kept alive here, never destroyed by the shard).
    std::shared_ptr<DevGraph> parent;
    ~DevGraph();
};
// The graph's device copy (created on first use, cached on the graph until lfr_graph_evict_device).
// stage_flows: copy the flows to HBM asynchronously on s_copy; otherwise leave them where they are
// (pinned host memory is read zero-copy by the assembly; pageable memory is staged after all).
int ensure_dev_graph(const Graph &g, int device, bool stage_flows, std::shared_ptr<DevGraph> &out);
// prestage_flows (lfr_internal.hpp; lfr_graph_from_matches_file_device): called by the scanner the moment the flow arrays are complete - the
// node numbering is still running - so that the 144 B per match are in HBM when the parse returns.  n_bound >= the final node count.
// Failures are not errors: the graph simply is not resident yet and ensure_dev_graph does everything later.

struct DevProblem {
    DevCtx *ctx = nullptr;
    std::shared_ptr<DevGraph> graph;
    int64_t N = 0;
    DevArena slab;
    int32_t *track = nullptr, *comp = nullptr;
    uint8_t *is_root = nullptr;
    uint32_t max_cc_matches = 0;       // most matches in one connected component of the match graph (0: unknown) - bounds a component's edge count for the batch-order keys
    ~DevProblem();
};
// labels computed on the host (graph cut, side-car, fallbacks) -> HBM
int upload_labels(const Problem &p, int device, bool stage_flows, std::shared_ptr<DevProblem> &out);

// Tracks, roots and components on the GPU (lfr_graphstage.hip): fills p.dev and the stage statistics; the
// host copies of the labels are fetched on demand (Problem::ensure_host_labels).  Returns
// LFR_GRAPHSTAGE_USE_HOST when the input needs something only the host stage has.
constexpr int LFR_GRAPHSTAGE_USE_HOST = 1;
// shard_world > 1: only the connected components of the match graph dealt to shard_rank (the k-th component in node order goes to rank
// k mod shard_world): tracks, roots, components and - later - the batch of this rank cover exactly those; the other ranks' nodes stay
// unmatched singletons here.  When one connected component holds most of the matches the deal cannot balance: the stage then runs over
// the whole graph (p.cc_sharded stays false) and the caller shards the COMPONENTS at batch assembly as before.
int graph_stage_on_device(const Graph &g, int64_t max_nodes, int device, bool stage_flows, Problem &p, int shard_rank = 0, int shard_world = 1);

// What the host needs to launch the solve kernels, read back once at the end of the assembly.
struct AsmSummary {
    uint32_t n_desc, total_nodes, total_edges, n_tracks;
    uint32_t class_begin[KC_COUNT + 1];
    uint32_t class_max_rows[KC_COUNT];   // largest system (rows) of every workgroup class
    uint32_t too_big, unpaired;
    uint32_t packed_edges;               // records of the packed classes = the head of the edge order (the workgroup classes follow)
    uint64_t class_edges[KC_COUNT];
    uint64_t es_doubles, ws_doubles;     // per-edge scratch, + HBM matrices of the global class
};

struct DeviceAssembly {                  // arrays inside the batch's slab (capacities are upper bounds)
    CompDesc *d_descs = nullptr;
    EdgeRec *d_edges = nullptr;
    uint32_t *d_node_ids = nullptr;
    NodeInc *d_node_inc = nullptr;
    uint32_t *d_in_idx = nullptr;
    uint64_t *d_ws_off = nullptr, *d_es_off = nullptr;
    uint32_t *d_desc_component = nullptr, *d_desc_class = nullptr, *d_desc_tracks = nullptr;
    // fused gather (whole batches over device-resident flows): the packed kernel reads its edges straight from the match-ordered
    // arrays of the graph through edge_ref[p] = directed edge id of record p and edge_word[p] = src | (dst | kind << 15) << 16; the
    // 80-byte records are only written for the workgroup classes
    bool fused = false;
    uint32_t *d_edge_ref = nullptr, *d_edge_word = nullptr;
    AsmSummary summary{};
};
// bytes of batch slab the assembly's outputs need for a graph of N nodes, M matches, C components
size_t assembly_output_bytes(int64_t N, int64_t M, int64_t C);
// Builds the batch layout of shard `shard_rank` of `shard_world` (snake deal in batch order, see assign_shards)
// out of `slab`; synchronises s_main once, at the end.
int assemble_on_device(const Problem &p, const DevProblem &labels, int shard_rank, int shard_world, DevArena &slab, DeviceAssembly &out);

// Warm-up: rocPRIM picks other kernels once its inputs are large (one-sweep radix sort above ~1 M items, block + merge sort below,
// look-back scans) and HIP loads every kernel's code on its first launch; a toy graph never reaches the large-input variants, so a
// one-shot caller's "Total time" was spent loading them (tracks 28 ms against 2 ms).  These run every sort / scan instantiation of
// their translation unit once at both sizes over scratch memory (a few ms of GPU time, no host-side graph).
int warm_graphstage_primitives(DevCtx *ctx);
int warm_assembly_primitives(DevCtx *ctx);

// shard of the i-th solvable component in batch order (class, edges descending, ...): dealt out and back
// (0..W-1, W-1..0, ...), so every shard receives the same mix of kernel classes and sizes
__host__ __device__ inline int snake_shard(int64_t i, int world) {
    const int64_t round = i / world;
    const int pos = (int)(i - round * world);
    return (round & 1) ? world - 1 - pos : pos;
}

// small helpers shared by the two pipeline stages (all asynchronous on `st`)
constexpr int kPipeThreads = 256;
inline dim3 pipe_grid(int64_t n) { return dim3((unsigned)((n < 1 ? 1 : n) + kPipeThreads - 1) / kPipeThreads); }

#ifdef __HIPCC__
// Several byte fills in ONE launch.  hipMemsetAsync is a launch of its own per call - two when the size is not a multiple of its fill
// width (the N bytes of is_root) - at 4-5 us each on an otherwise busy stream; a stage's initial values (a zero block, a block of -1,
// a byte array) go out together instead.  Regions start 16-byte aligned (arena blocks are 256-byte aligned).
struct FillRegions {
    static constexpr int kMax = 4;
    void *ptr[kMax];
    unsigned long long bytes[kMax];
    unsigned int byte_value[kMax];
    int n = 0;
    void add(void *p, size_t b, unsigned int v) { if (b && n < kMax) { ptr[n] = p; bytes[n] = b; byte_value[n] = v & 0xffu; ++n; } else if (b) std::abort(); }       // (more than kMax regions: a programming error)
};
__global__ void k_fill_regions(FillRegions r);
inline hipError_t fill_regions(const FillRegions &r, hipStream_t st) {
    if (r.n == 0) return hipSuccess;
    unsigned long long words = 0;
    for (int j = 0; j < r.n; ++j) words += r.bytes[j] / 16 + 1;
    const unsigned long long per_block = (unsigned long long)kPipeThreads * 8;       // eight 16-byte stores per thread
    const unsigned int blocks = (unsigned int)((words + per_block - 1) / per_block < 1 ? 1 : ((words + per_block - 1) / per_block > 65536 ? 65536 : (words + per_block - 1) / per_block));
    hipLaunchKernelGGL(k_fill_regions, dim3(blocks), dim3(kPipeThreads), 0, st, r);
    return hipGetLastError();
}
#endif

}  // namespace lfr

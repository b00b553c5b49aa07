// Internal data model of liblfr_hip.so (not part of the C ABI; see include/lfr.h).
#pragma once
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <exception>
#include <functional>
#include <vector>

#include "lfr.h"
#include "lfr_devctx.hpp"

namespace lfr {

void set_error(const char *fmt, ...);
struct DevGraph;
struct DevProblem;

// ------------------------------------------------------------------------------------------
// Match graph (solve.cc:405-481).  Directed edge 2m   = node1(m) -> node2(m), flow disp2(m);
//                                  directed edge 2m+1 = node2(m) -> node1(m), flow disp1(m)
// (solve.cc:477-478).  A node's out-edges in the reference's insertion order are exactly its
// directed edges in ascending id.
// ------------------------------------------------------------------------------------------
struct Graph {
    std::vector<std::string> image_names;      // seen (non-banned) images, order of first appearance
    std::vector<float> image_fact;             // first-wins (solve.cc:449,451)
    std::unordered_map<std::string, int32_t> image_index;
    // per-match arrays: pinned host memory when a GPU is present (lfr_devctx.hpp), so that they cross PCIe at
    // link speed and the flows can be gathered zero-copy by a sharded assembly
    HostBuf<uint32_t> m_node1, m_node2;
    HostBuf<float> m_sim;
    HostBuf<float> m_disp1, m_disp2;           // 18 floats per match, zero padded (solve.cc:460-472)
    // flows that never left the GPU (lfr_graph_from_arrays_device_flows): device arrays of
    // n_rows x 18 floats owned by the caller; match m uses row m_flow_row[m] (identity when empty)
    const float *dev_disp1 = nullptr, *dev_disp2 = nullptr;
    int dev_flows_device = -1;
    std::vector<uint32_t> m_flow_row;
    std::vector<int32_t> node_image;
    std::vector<uint32_t> node_feat;

    // open-addressing (image, feature) -> node map, used only while building
    std::vector<uint64_t> hkeys;
    std::vector<int64_t> hvals;
    uint64_t hmask = 0;
    int64_t hcount = 0;

    // nodes grouped by image (built by finish(); lfr_apply_displacements)
    std::vector<int64_t> img_off;
    std::vector<uint32_t> img_nodes;

    // ingest temporaries kept mapped until the graph dies (lfr_wire.cpp: large unmaps right before GPU work stall the queues)
    std::vector<std::shared_ptr<void>> ingest_keepalive;

    // the graph's copy in HBM (lfr_assemble.hpp), created on first use by the device pipeline or by
    // lfr_graph_to_device, dropped by lfr_graph_evict_device
    mutable std::vector<std::shared_ptr<DevGraph>> devgs;      // indexed by HIP device ordinal
    int prefetch_device = -1;          // ingest straight to this device: the scanner starts the flows' upload when they are in place
    mutable std::mutex dev_mu;

    // The device graph stage sums similarities with atomics (root scores, meta-edge weights): order independent - and therefore
    // bit-identical to the reference's sequential sums (solve.cc:557-562, 268-289) - only while the fp64 sums of the float32
    // values are EXACT, i.e. while the similarities share a narrow exponent range.  finish() checks it (max |sim| / min |sim| of
    // the non-zero, finite values <= 2^10: then 2^18 terms still sum exactly in 53 bits); otherwise the host stage runs.
    bool sims_sum_exactly = true;

    int64_t n_nodes() const { return (int64_t)node_image.size(); }
    int64_t n_matches() const { return (int64_t)m_sim.size(); }
    int32_t intern_image(const std::string &name, float fact);
    uint32_t find_or_create_node(int32_t image, uint32_t feature);
    void add_match(int32_t img1, int32_t img2, uint32_t f1, uint32_t f2, float sim, const float *d1, int n1,
                   const float *d2, int n2);
    void finish();   // drop the hash map, group the nodes by image
};

// ------------------------------------------------------------------------------------------
// Device batch layout (host copy).  One 80-byte record per kept directed edge, components
// contiguous, records in the reference's residual-block order (solve.cc:98-102).
// ------------------------------------------------------------------------------------------
struct alignas(16) EdgeRec {
    float flow[18];      // grid_idx*2 + {di,dj}
    float sim;           // ScaledLoss weight (solve.cc:111,120)
    uint16_t src;        // local node index inside the component (variable nodes first)
    uint16_t dst_kind;   // bit 15: 1 = inter-track (Tukey), 0 = intra-track (Cauchy); bits 0-14: local dst
};
static_assert(sizeof(EdgeRec) == 80, "EdgeRec must be 80 bytes");

struct CompDesc {
    uint32_t edge_off;       // first EdgeRec
    uint32_t n_edges;
    uint32_t node_off;       // into node_ids
    uint16_t n_nodes;        // variable + constant
    uint16_t n_var;          // variable nodes (local indices [0, n_var))
};
static_assert(sizeof(CompDesc) == 16, "CompDesc must be 16 bytes");

// incidence of one local node: its out-edges are contiguous in the component's edge list
// (records [out_begin, out_begin+out_count)); its in-edges are in_idx[in_begin .. in_begin+in_count)
// (component-local edge indices, ascending).  Used by the owner-computes assembly of the
// workgroup kernels: every row of J^T J is summed by one thread in a fixed order (deterministic).
struct NodeInc { uint32_t out_begin, out_count, in_begin, in_count; };
static_assert(sizeof(NodeInc) == 16, "NodeInc must be 16 bytes");

// kernel classes
enum KernelClass : int {
    KC_G8 = 0,      // packed <NV=8, LPR=1, EPL=3>:  8 comps/wave, <=8 rows,  <=24 edges
    KC_G16,         // packed <16,1,6>: 4 comps/wave, <=16 rows, <=96 edges (3 slots resident, 3 re-read per sweep)
    KC_G32,         // retired (was <16,2,3>: 2 comps/wave); kept so that the class indices of the C ABI stay put
    KC_G64_2,       // packed <32,1,6>: 2 comps/wave, <=24 rows, <=192 edges (2 slots resident, 4 re-read per sweep)
    KC_G64_4,       // packed <32,2,5>: 1 comp/wave,  <=32 rows, <=320 edges (2 slots resident, 3 re-read per sweep)
    // workgroup per component, normal matrix in LDS - three classes by LDS footprint so that small systems share a CU:
    KC_BLOCK,       //   <= 88 rows : <= 39.6 KB of LDS, 128 threads -> 4 workgroups per CU
    KC_BLOCK_M,     //   <= 130 rows: <= 79 KB, 256 threads        -> 2 workgroups per CU
    KC_BLOCK_L,     //   <= 192 rows: <= 160 KB, 256 threads       -> 1 workgroup per CU
    KC_GLOBAL,      // workgroup per component, normal matrix in HBM workspace
    KC_COUNT
};
constexpr uint32_t kNoClass = 15;            // sort key of components that are not solved (by this shard)
constexpr int kClassBits = 4;
#ifndef LFR_ROWS_S
#define LFR_ROWS_S 88
#endif
#ifndef LFR_ROWS_M
#define LFR_ROWS_M 130
#endif
constexpr int kBlockRowsS = LFR_ROWS_S, kBlockRowsM = LFR_ROWS_M;      // (overridable for class-boundary experiments)
inline bool is_lds_class(int cls) { return cls >= KC_BLOCK && cls <= KC_BLOCK_L; }
constexpr int kBlockMaxRows = 192;   // packed lower triangle 192*193/2*8 B = 148.2 KB + 15.4 KB of vectors <= 160 KiB of LDS
// rows above which a component's matrix lives in the HBM workspace instead of LDS: kBlockMaxRows, or LFR_BLOCK_MAX_ROWS (0..192)
int block_max_rows();
// `work` on `threads` threads (the caller is one of them), the others PERSISTENT workers of the process: a batch with elimination-tree
// components makes ~130 plans of ~1 ms each, and creating as many threads every time cost more than the plans.  `work` pulls its items from
// a counter of its own.  One call at a time (a second caller runs its work on threads of its own).
void run_on_pool(int threads, const std::function<void()> &work);
// One task for the same workers (the size cap's recursion hands the second half of a bisection to one: creating a thread per half cost
// 0.1-1 ms each).  pool_wait returns when the task has run; while it waits the caller runs queued tasks itself, so tasks may spawn and wait
// for tasks of their own.
// An exception thrown by the task is kept and rethrown by pool_wait (on a detached worker it would end the process).
struct PoolTask { std::function<void()> fn; bool done = false; std::exception_ptr err; };
std::shared_ptr<PoolTask> pool_async(std::function<void()> fn);
void pool_wait(const std::shared_ptr<PoolTask> &t);

struct Problem {
    const Graph *g = nullptr;
    // per-node labels.  After the device graph stage they live in HBM (`dev`) and these host copies are
    // fetched on first use (ensure_host_labels: lfr_problem_get_labels, host fallbacks).
    mutable std::vector<int64_t> track, comp;
    mutable std::vector<uint8_t> is_root;
    mutable bool host_labels_valid = true;
    mutable std::mutex label_mu;
    mutable std::vector<std::shared_ptr<DevProblem>> devs;   // labels in HBM per device ordinal (device graph stage, or uploaded by the first device assembly there)
    std::shared_ptr<DevProblem> dev_get(int device) const {
        std::lock_guard<std::mutex> lk(label_mu);
        return device >= 0 && device < (int)devs.size() ? devs[device] : nullptr;
    }
    void dev_set(int device, const std::shared_ptr<DevProblem> &d) const {
        std::lock_guard<std::mutex> lk(label_mu);
        if ((int)devs.size() <= device) devs.resize(device + 1);
        devs[device] = d;
    }
    int ensure_host_labels() const;            // lfr_graphstage.hip
    lfr_problem_stats stats{};
    bool host_batch = true;                // false: labels only, the batch is assembled on the device
    bool cc_sharded = false;               // the device graph stage ran over one rank's connected components only (graph_stage_on_device)
    int64_t shard_matches = 0, shard_matches_max = 0;     // cc_sharded: matches of this rank / of the largest rank
    // batch (all solvable components, sorted by kernel class then size descending)
    std::vector<CompDesc> descs;
    std::vector<int64_t> desc_component;   // original component id per desc
    std::vector<int32_t> desc_class;
    std::vector<int32_t> desc_tracks;      // tracks (>=2 nodes) inside
    std::vector<EdgeRec> edges;
    std::vector<uint32_t> node_ids;        // global node id per local slot
    std::vector<NodeInc> node_inc;         // parallel to node_ids
    std::vector<uint32_t> in_idx;          // parallel to edges (per component: edge indices sorted by dst)
};

// Deal of the solvable components to `world` shards: the batch order (kernel class, then edge count
// descending - the largest-first task order of solve.cc:599-634) dealt out and back (snake_shard in
// lfr_assemble.hpp), so every shard gets the same mix of classes and sizes.  Returns shard per desc.
std::vector<int32_t> assign_shards(const Problem &p, int world);

int build_problem(const Graph &g, int64_t max_nodes, const int64_t *component_override, Problem &p, bool host_batch = true);
void build_out_csr(const Graph &g, std::vector<int64_t> &out_off, std::vector<int64_t> &out_eid);
// separate_meta_graph (solve.cc:252-373) given the tracks (lfr_graph.cpp); out_off/out_eid: optional out-edge CSR
void components_from_tracks(const Graph &g, const std::vector<int64_t> &track, int64_t n_tracks, const std::vector<int64_t> &tsize,
                            int64_t max_nodes, const std::vector<int64_t> *out_off, const std::vector<int64_t> *out_eid,
                            std::vector<int64_t> &comp, int64_t &n_components, int64_t &n_cut);

// recursive_graph_cut (solve.cc:185-250) around bisect_graph: {meta node: subset index} (lfr_graph.cpp)
std::unordered_map<int, int> recursive_cut(const std::vector<std::pair<int, int>> &edges, const std::vector<int> &weights,
                                           const std::vector<int64_t> &node_weights, int64_t max_weight);

// deterministic substitute for colmap::ComputeNormalizedMinGraphCut(edges, weights, 2)
// (solve.cc:192): returns part (0/1) per node id appearing in `edges`.
void bisect_graph(const std::vector<std::pair<int, int>> &edges, const std::vector<int> &weights,
                  std::unordered_map<int, int> &part);

// device residency of a graph, callable from host-only translation units (defined in lfr_graphstage.hip)
void prestage_flows(const Graph &g, int device, int64_t n_bound);   // scanner: the flows are complete, start their upload (best effort)
int graph_make_resident(const Graph &g, int device);                // = lfr_graph_to_device

// ------------------------------------------------------------------------------------------
// Elimination-tree plan of a KC_GLOBAL component (lfr_treeplan.cpp): nested dissection of the tracks, blocks of <= 8 nodes,
// block-level symbolic factorization, columns by level of the elimination tree, left-looking update lists, sweep items.
// `blob` is what the kernel reads (u32 words): hdr[32] = {NB, tiles, offset of the tiles (doubles from the base), offset of the
// vectors, n_pad = 16 NB, levels, items, phase-1 tasks, then the word offsets of colptr[NB+1], rowsof[tiles], nreal[NB],
// level_ptr[levels+1], level_cols[NB], p1_ptr[levels+1], p1_tasks[][4], upd[][3], x_ptr[levels+1], x_tasks[][3], ncarry[NB],
// items[items+1][8], item_edges[], node_items[8 NB + 1], ipos[8 NB]; [23] = offset of the items' partial sums, [24] = vector stride,
// [28] = 1: every column carries all its tiles (dependency-counter schedule), [25] [26] [27] = word offsets of col_upd_ptr[NB+1], col_upd[][5], col_desc[NB][32] (columns in level order)}.  Behind the words: the
// tiles of A, the tiles of the factor, the items' partial sums, the vectors.
// ------------------------------------------------------------------------------------------
constexpr int kTreeHdrWords = 32;
constexpr int kTreeMaxFlagColumns = 4096;   // dependency counters of the barrier-free schedule live in LDS (one int per column)
constexpr int kTreeVectors = 12;             // x, trial x, g, trial g, scale, diag, step, D, diag(A), delta, 1/d, w (the right-hand side riding through the factorization)
struct TreePlan {
    int n_var = 0, NB = 0, n_levels = 0, n_tracks = 0, n_segments = 0, max_front = 0;
    uint32_t n_tiles = 0, n_items = 0;
    uint64_t n_updates = 0, column_rounds = 0;
    std::vector<uint32_t> blob;              // empty: the component is beyond the plan's 32-bit offsets
    uint64_t vec_stride() const, header_doubles() const, team_doubles() const, doubles() const;
};
// words[e] = src | (dst | kind << 15) << 16 of the component's records (the last word of EdgeRec)
void tree_plan(int n_var, int64_t n_edges, const uint32_t *words, TreePlan &out);

}  // namespace lfr

struct lfr_graph { lfr::Graph g; };
struct lfr_problem { lfr::Problem p; };

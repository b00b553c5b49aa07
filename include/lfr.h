/* lfr.h — C ABI of liblfr_hip.so: MI355X-native multi-view keypoint-refinement solver.
 *
 * Drop-in scope: the reference's `solve` executable, multi-view-refinement/solve.cc:375-682
 * (+ cost.cc, graph.{h,cc}).  The reference has no library API for this path — its only
 * interface is the CLI + the two protobuf files (types.proto) — so the entry points below are
 * the stages of that main(), cut where a host program (the `solve` launcher, bench.py, a
 * cgo/ctypes binding) needs to hold data between them.  Each declaration cites the reference
 * lines it replaces.  Plain C types only, caller-owned output buffers, int return codes
 * (0 = ok, <0 = error, text via lfr_last_error()), no exceptions across the boundary.
 *
 * Units/axes: positions[2n] = di (row / y), positions[2n+1] = dj (col / x) of node n, in the
 * solver's unit (16 px * fact at extraction resolution; colmap_utils.py:133-136).
 *
 * Undefined inputs (the reference defines none of them; tests/test_gpu_undefined_inputs.py pins the library against the oracle):
 *   - similarity == 0 is a legal weight: the edge drops out of the cost (ScaledLoss(.., 0), solve.cc:111,120);
 *   - similarity < 0, +-inf or NaN, or a flow entry that is not finite, makes the evaluation of ITS component non-finite: no LM
 *     step of that component is ever valid, it terminates LFR_TERM_FAILURE after ten invalid steps and keeps zero displacements
 *     (Ceres: a non-finite evaluation fails, IsSolutionUsable() is false and solve.cc:609-612 left the positions at 0).  Every other
 *     component - also one packed into the same wavefront - is solved exactly as if the bad match were not there;
 *   - a NaN similarity additionally leaves the order-dependent graph stage (tracks and roots sort by similarity, solve.cc:489-582)
 *     undefined in the reference itself; the library never emits a non-finite displacement, but which node of a tie becomes a
 *     root may differ from a given Ceres build;
 *   - all-zero flow grids (SKIP_REFINEMENT, compute_match_graph.py:150-152): every component converges at iteration 0, all zeros.
 */
#ifndef LFR_H
#define LFR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFR_VERSION 1

/* error codes */
#define LFR_OK 0
#define LFR_ERR_ARG (-1)
#define LFR_ERR_IO (-2)
#define LFR_ERR_PARSE (-3)       /* "Failed to parse proto object."  solve.cc:433-436 */
#define LFR_ERR_HIP (-4)
#define LFR_ERR_UNSUPPORTED (-5)
#define LFR_ERR_NOMEM (-6)

/* Tukey loss flavour: Ceres changed TukeyLoss by a factor 2 between 1.14 and 2.0 and the
 * reference pins no Ceres version (CMakeLists.txt:9). */
#define LFR_TUKEY_CERES1 1
#define LFR_TUKEY_CERES2 2

/* termination types (ceres::TerminationType subset that ceres::Solve can return here) */
#define LFR_TERM_CONVERGENCE 0
#define LFR_TERM_NO_CONVERGENCE 1
#define LFR_TERM_FAILURE 2

typedef struct lfr_graph lfr_graph;       /* parsed match graph: nodes + directed edges (host) */
typedef struct lfr_problem lfr_problem;   /* tracks, roots, components + device batch layout (host) */
typedef struct lfr_batch lfr_batch;       /* a problem (or one shard of it) resident in HBM */

int lfr_version(void);
const char *lfr_last_error(void);          /* thread-local, valid until the next failing call */

/* ---------------------------------------------------------------------------------------------
 * A0/A3  MatchingFile ingest + graph construction.                    solve.cc:426-481, graph.cc
 * ------------------------------------------------------------------------------------------- */

/* Parse serialized MatchingFile(s) (types.proto:3-28) in the given order, skipping image pairs
 * that touch a banned image (solve.cc:444-446).  Nodes are numbered in order of first
 * appearance, node1 before node2 (solve.cc:474-475). */
int lfr_graph_from_files(const char *const *paths, int n_paths, const char *const *banned, int n_banned,
                         lfr_graph **out);

/* Resolve `path`, or `path.part.0`, `path.part.1`, ... up to the first gap (solve.cc:416-424),
 * then behave as lfr_graph_from_files. */
int lfr_graph_from_matches_file(const char *path, const char *const *banned, int n_banned, lfr_graph **out);

/* lfr_graph_from_matches_file + lfr_graph_to_device(device) in one call, with the two overlapped: the flows (144 of the 164 bytes per
 * match) start their way to HBM as soon as the scanner has them in place, while it still numbers the nodes; endpoints, similarities and
 * node images follow before the call returns (asynchronously: the pipeline's stream waits for them, the caller does not).  The graph and
 * every result are identical to the two-call form; a device that cannot be initialised is LFR_ERR_HIP.  This is the ingest the `solve`
 * drop-in uses: the reference's "Total time" (solve.cc:487) starts after the file is in memory, ours after it is in HBM or on its way. */
int lfr_graph_from_matches_file_device(const char *path, const char *const *banned, int n_banned, int device, lfr_graph **out);

/* Same graph from flat arrays — the producer contract of compute_match_graph.py:163-187 without
 * the protobuf hop.  pair_img1/2[p]: image index of ImagePair p; matches of pair p are
 * [pair_off[p], pair_off[p+1]); disp1/disp2: n_matches x 18 float32 (grid_idx*2 + {di,dj}),
 * disp2 = flow image1->image2, disp1 = flow image2->image1 (solve.cc:477-478). */
int lfr_graph_from_arrays(int32_t n_images, const char *const *image_names, const float *image_facts,
                          int64_t n_pairs, const int32_t *pair_img1, const int32_t *pair_img2,
                          const int64_t *pair_off, const uint32_t *feat1, const uint32_t *feat2,
                          const float *sim, const float *disp1, const float *disp2,
                          const char *const *banned, int n_banned, lfr_graph **out);

/* Producer contract without the host hop for the flows (compute_match_graph.py:163-187 keeps
 * grid_displacements12/21 on the GPU): as lfr_graph_from_arrays, but disp1/disp2 are DEVICE pointers
 * (n_matches x 18 float32 on HIP device `device`, owned by the caller, alive until the batch has been
 * created).  Such a graph is solved through lfr_problem_build_labels / lfr_problem_build_hip +
 * lfr_batch_create on the same device; lfr_problem_build (host assembly) rejects it. */
int lfr_graph_from_arrays_device_flows(int32_t n_images, const char *const *image_names, const float *image_facts,
                                       int64_t n_pairs, const int32_t *pair_img1, const int32_t *pair_img2,
                                       const int64_t *pair_off, const uint32_t *feat1, const uint32_t *feat2,
                                       const float *sim, const void *disp1_device, const void *disp2_device, int device,
                                       const char *const *banned, int n_banned, lfr_graph **out);

/* Device residency of the graph (the GPU-side counterpart of the Graph object solve.cc:405-481 builds while it
 * parses).  lfr_graph_to_device starts copying endpoints, similarities and flows to HBM
 * asynchronously and returns; the device pipeline (lfr_problem_build_hip, lfr_batch_create) continues from that
 * copy, and creates it itself - inside the caller's "Total time" span - when it does not exist.  The copy stays
 * cached on the handle until lfr_graph_evict_device / lfr_graph_free. */
int lfr_graph_to_device(const lfr_graph *g, int device);
int lfr_graph_evict_device(const lfr_graph *g);

void lfr_graph_free(lfr_graph *g);
int64_t lfr_graph_num_nodes(const lfr_graph *g);      /* "# graph nodes"  solve.cc:484 */
int64_t lfr_graph_num_edges(const lfr_graph *g);      /* "# graph edges" = 2 x matches  solve.cc:485 */
int32_t lfr_graph_num_images(const lfr_graph *g);     /* images_set.size()  solve.cc:448,450,586 */
/* node_image[n] = index into the seen-image list; node_feature[n] = feature_idx */
int lfr_graph_get_nodes(const lfr_graph *g, int32_t *node_image, uint32_t *node_feature);
const char *lfr_graph_image_name(const lfr_graph *g, int32_t image);
float lfr_graph_image_fact(const lfr_graph *g, int32_t image);

/* Serialize a MatchingFile from flat arrays (native counterpart of compute_match_graph.py:163-205;
 * used by the generators and by callers that want to keep the file hand-off). */
int lfr_write_matching_file(const char *path, int32_t n_images, const char *const *image_names,
                            const float *image_facts, int64_t n_pairs, const int32_t *pair_img1,
                            const int32_t *pair_img2, const int64_t *pair_off, const uint32_t *feat1,
                            const uint32_t *feat2, const float *sim, const float *disp1, const float *disp2);

/* ---------------------------------------------------------------------------------------------
 * A4-A7, A9, A11  tracks, roots, components, problem assembly.          solve.cc:487-606, 79-143
 * ------------------------------------------------------------------------------------------- */
typedef struct lfr_problem_stats {
    int64_t n_tracks;              /* "# tracks"            solve.cc:534 */
    int64_t max_track_size;        /* "max track size"      solve.cc:549 */
    int64_t n_components;          /* "# components"        solve.cc:591 */
    int64_t max_component_size;    /* "max component size"  solve.cc:606 */
    int64_t n_cut_components;      /* components above the cap that went through the graph cut */
    int64_t n_solved_components;   /* components handed to the solver (>1 node, >=1 variable) */
    int64_t n_solved_tracks;       /* tracks with >=2 nodes inside solved components */
    int64_t n_solved_edges;        /* directed edges = residual blocks of the reduced programs */
    int64_t n_solved_nodes;        /* nodes (variable + constant) inside solved components */
    double tracks_ms, roots_ms, graph_cut_ms, assemble_ms;
    double kruskal_rounds;         /* device graph stage: parallel rounds spent on large connected components (0: none) */
    double tie_resorts;            /* device graph stage: 1 if a long run of equal similarities made it order the matches with the three
                                      stable sorts instead of the one sort + in-place tie fix (same order either way) */
} lfr_problem_stats;

/* max_nodes_in_component <= 0: use the number of seen images (solve.cc:586).
 * component_override: NULL, or n_nodes component ids that replace separate_meta_graph()
 * (solve.cc:586) — the side-car for exact parity with a reference run on inputs whose components
 * exceeded the cap (Graclus' cut is not reproducible). */
int lfr_problem_build(const lfr_graph *g, int64_t max_nodes_in_component, const int64_t *component_override,
                      lfr_problem **out);
/* Same graph stage (tracks, roots, components) without the host-side batch assembly: the 80-byte
 * edge records, descriptors and incidence lists are then built on the GPU by lfr_batch_create - the
 * whole problem (shard_world = 1: the flows cross PCIe once in match order) or one shard of it
 * (shard_world > 1: the device filters its components and gathers only their flow rows); no host-side
 * record array is ever materialised.  Results are bit-identical to the host-assembled batch / shard. */
int lfr_problem_build_labels(const lfr_graph *g, int64_t max_nodes_in_component, const int64_t *component_override,
                             lfr_problem **out);
/* As lfr_problem_build_labels, with the graph stage itself on HIP device `device`: radix sort of the
 * matches, connected components, the order-dependent constrained union-find (solve.cc:499-523) by one
 * GPU thread per small connected component and by exact parallel rounds over prefix blocks for large
 * ones, roots, components; components above the size cap are cut on the host from meta edges the
 * device compacts and sums, and re-labelled on the device.  Bit-identical labels.  Falls back to the
 * host stage only when component_override is given, for >= 2^31 nodes or >= 2^30 matches, when the
 * similarities span more than 2^10 in magnitude or hold inf/nan (the device sums them with atomics, exact and
 * therefore order independent only in a narrow exponent range; the host sums in the reference's order), when the
 * image bitsets of the parallel rounds would exceed 24 GB, or after 100000 rounds (a path-shaped
 * dependency chain). */
int lfr_problem_build_hip(const lfr_graph *g, int device, int64_t max_nodes_in_component,
                          const int64_t *component_override, lfr_problem **out);
/* flags: LFR_BUILD_FLOWS_STAY_ON_HOST - do not stage the flows in HBM; a sharded lfr_batch_create then gathers
 * only its shard's rows zero-copy from the graph's pinned host arrays (multi-GPU: every GPU pulls 1/world of the
 * flow bytes over its own PCIe link). */
#define LFR_BUILD_FLOWS_STAY_ON_HOST 1
int lfr_problem_build_hip_ex(const lfr_graph *g, int device, int64_t max_nodes_in_component,
                             const int64_t *component_override, int flags, lfr_problem **out);
/* Multi-GPU, one process per GPU (solve.cc:594-597: components are independent; solve.cc:489-541: the constrained spanning forest
 * never joins two connected components of the match graph, so tracks, roots and components decompose by connected component too):
 * the graph stage of rank shard_rank of shard_world over ITS connected components only - the k-th connected component in node order
 * belongs to rank k mod shard_world.  lfr_batch_create(p, device, 0, 1) then assembles this rank's components; the union over the ranks
 * is the whole problem, every component bit-identical to the unsharded run.  lfr_problem_cc_sharded(p) returns 0 when one connected
 * component dominates (or the host stage ran): the problem then covers the whole graph and the caller shards the COMPONENTS at
 * assembly, lfr_batch_create(p, device, shard_rank, shard_world), as with lfr_problem_build_hip_ex. */
int lfr_problem_build_hip_shard(const lfr_graph *g, int device, int64_t max_nodes_in_component, int flags, int shard_rank, int shard_world,
                                lfr_problem **out);
int lfr_problem_cc_sharded(const lfr_problem *p);
void lfr_problem_free(lfr_problem *p);
/* The two-way cut this library substitutes for colmap::ComputeNormalizedMinGraphCut(edges, weights, 2)
 * (solve.cc:192; COLMAP wraps Graclus there, a third-party multilevel heuristic that cannot be restated: see
 * DESIGN.md).  Exposed so that a checker can run the reference's recursion around the same primitive
 * (tests/, oracle/).  edges (edge_a[k], edge_b[k]) with integer weights; writes the distinct node ids in
 * ascending order and their side (0/1) - 2 * n_edges entries are always enough - and returns their count. */
int64_t lfr_bisect_graph(int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b, const int32_t *weights,
                         int32_t *nodes, int32_t *part);
/* The whole size-cap recursion (the product's restatement of recursive_graph_cut, solve.cc:185-250, around lfr_bisect_graph): subset index
 * of every node that has an edge, nodes ascending; node_weights[id] for every id below n_node_weights.  For checkers (the oracles run their
 * own literal recursion around lfr_bisect_graph; the two must agree) and for timing the host part of the cut.  Returns the node count. */
int64_t lfr_debug_recursive_cut(int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b, const int32_t *weights,
                                int64_t n_node_weights, const int64_t *node_weights, int64_t max_weight, int32_t *nodes, int32_t *subset);
/* The elimination-tree plan the solver gives a component whose normal matrix does not fit LDS (the reference hands such systems to
 * Ceres' SPARSE_NORMAL_CHOLESKY, solve.cc:147): nested dissection of the tracks, 16-row blocks, block-level symbolic factorization,
 * columns by level of the elimination tree, left-looking update lists and the sweep's (node, neighbour) items - exactly the words
 * the kernel reads, so that a checker can execute the plan on the CPU (tests/test_tree_plan.py).  words[e] as above.  Returns the
 * number of 32-bit words of the plan (copied to `blob` if cap is large enough; < 0: error); info[8] = {blocks, tiles, levels,
 * items, updates, tracks, segments, column rounds}. */
int64_t lfr_debug_tree_plan(int32_t n_var, int64_t n_edges, const uint32_t *words, uint32_t *blob, int64_t cap, int64_t *info);
int lfr_problem_get_stats(const lfr_problem *p, lfr_problem_stats *stats);
/* per node: track_idx_container, is_root, component_idx_container of solve.cc:526,570,586 */
int lfr_problem_get_labels(const lfr_problem *p, int64_t *track, uint8_t *is_root, int64_t *component);

/* Solvable components dealt to shard `shard_rank` of `shard_world`: the batch order (kernel class, then
 * edge count descending - the largest-first task order of solve.cc:599-634) dealt out and back
 * (0..W-1, W-1..0, ...), so every shard gets the same mix of classes and sizes.  Needs a host-assembled problem.
 * Fills original component ids / edge counts (either may be NULL); returns the count. */
int64_t lfr_problem_shard_components(const lfr_problem *p, int shard_rank, int shard_world, int64_t *components,
                                     int64_t *n_edges);

/* ---------------------------------------------------------------------------------------------
 * A1, A2, A8, A10  batched Levenberg-Marquardt on the GPU.    solve.cc:79-160,614-635 + cost.cc
 * ------------------------------------------------------------------------------------------- */
typedef struct lfr_solve_stats {
    int64_t n_components, n_edges, n_nodes, n_tracks;    /* of this batch/shard */
    int64_t n_converged, n_no_convergence, n_failed;
    int64_t sum_iterations;
    int64_t ref_jacobian_passes_edges;   /* sum_c E_c * (jacobian evaluations Ceres performs)  */
    int64_t ref_cost_passes_edges;       /* sum_c E_c * (cost-only evaluations Ceres performs) */
    int64_t exec_passes_edges;           /* sum_c E_c * (edge sweeps the kernels executed)     */
    int64_t ref_passes_nodes;            /* sum_c N_c * (all evaluations Ceres performs)       */
    double sum_final_cost;
    double kernel_ms;                    /* HIP-event time of the solve kernels, last solve */
    double h2d_ms, d2h_ms;
    double dominant_kernel_ms;           /* HIP-event time of the largest kernel launch */
    int64_t dominant_kernel_edges;       /* edges processed by that launch */
    int64_t dominant_kernel_nodes;
    int64_t dominant_ref_passes_edges;   /* (jacobian + cost passes) * edges for that launch */
    int64_t dominant_ref_passes_nodes;
} lfr_solve_stats;

/* Create the HIP context of `device` (a few hundred ms the first time) and run a toy graph through the
 * device pipeline once, so that every kernel is resolved before the real input arrives; safe to call
 * from a side thread while the caller parses its input. */
int lfr_hip_warmup(int device);
/* Pre-populate the per-device caches (device slabs, pinned staging) with what a pipeline run over a graph of
 * n_nodes / n_matches will ask for, so that a one-shot caller's timed span pays no allocation; lfr_hip_trim
 * returns every cached slab to the driver. */
int lfr_hip_reserve(int device, int64_t n_nodes, int64_t n_matches);
/* Wait for everything the library has in flight on its own streams of `device` (e.g. the upload started by
 * lfr_graph_to_device). */
int lfr_hip_synchronize(int device);
int lfr_hip_trim(int device);

/* Shard `shard_rank` of `shard_world` (see lfr_problem_shard_components) resident on HIP device `device`:
 * assembled there from the labels (lfr_problem_build_labels / _hip), or uploaded (lfr_problem_build).
 * Limits: <= 32767 nodes per component (16-bit local indices in the 80-byte edge record), < 2^30 matches; a component above 192 rows
 * whose factor needs 2^21 or more 16x16 tiles (4 GB: a DENSE component beyond ~16 k nodes) or that holds 2^30 or more records (the
 * plan's sweep items pack record << 2 | direction << 1 | flag into 32 bits) is refused (LFR_ERR_UNSUPPORTED). */
int lfr_batch_create(const lfr_problem *p, int device, int shard_rank, int shard_world, int tukey_variant,
                     lfr_batch **out);
void lfr_batch_free(lfr_batch *b);
/* Run every solve kernel of the batch on `hip_stream` (a hipStream_t, NULL = default stream).
 * The position array is zeroed once, when the batch is created (solve.cc:609-612); every solve rewrites every
 * variable node (a failed solve writes 0), roots and nodes outside solved components stay 0.  Asynchronous
 * unless stats != NULL, in which case the call synchronizes the stream and fills stats. */
int lfr_batch_solve(lfr_batch *b, void *hip_stream, lfr_solve_stats *stats);
/* HIP-event timings of one of the last 64 lfr_batch_solve calls (solves_back = 0: the latest);
 * waits for that solve to finish.  class_ms / class_edges: LFR_NUM_KERNEL_CLASSES entries, one
 * per kernel launch (packed <8,1,3>, <16,1,6>, a retired slot, <32,1,6>, <32,2,5>; workgroup per component with the matrix
 * in LDS: <= 88 rows, <= 130 rows, <= 192 rows; workgroup per component with the matrix in HBM). */
#define LFR_NUM_KERNEL_CLASSES 9
int lfr_batch_timing(lfr_batch *b, int solves_back, double *total_ms, double *class_ms, int64_t *class_edges);
/* Per component (order of lfr_batch_component_info): columns, tiles, 16x16x16 updates per factorization, levels and sweep items of the
 * elimination-tree plan a component above 192 rows is solved with (the reference: Ceres SPARSE_NORMAL_CHOLESKY, solve.cc:147); zeros
 * for the other components.  Any pointer may be NULL.  Returns the number of components (< 0: error). */
int64_t lfr_batch_tree_stats(lfr_batch *b, int64_t *columns, int64_t *tiles, int64_t *updates, int64_t *levels, int64_t *items);
/* Diagnostics: bounded spin-waits inside the workgroup kernels (wave hand-offs of the factorizations) that ran out during the
 * latest solve of the batch - each one rejected an LM step instead of hanging the GPU.  0 on a healthy run; < 0: error. */
int64_t lfr_batch_spin_timeouts(lfr_batch *b);
/* Components above 192 rows whose expected work reaches a threshold (LFR_TREE_TEAM="w2,w4": rows x [joins several tracks]; "0" = off)
 * are solved by a TEAM of 2 or 4 workgroups on one XCD (the reference: one Ceres SPARSE_NORMAL_CHOLESKY problem per pool thread,
 * solve.cc:147,617-635 - here the elimination tree's columns, the sweeps and the vector passes of ONE problem are spread over several
 * CUs).  Returns how many components the latest solve of the batch handed to a team (< 0: error). */
int64_t lfr_batch_team_runs(lfr_batch *b);
/* The teams of one launch form from workgroups that are resident on an XCD at the same time (the reference: a pool thread per problem,
 * solve.cc:617-635 - no such constraint).  When the CUs are not there - other kernels, another process, a smaller partition - and nobody
 * is at work and nothing moves for LFR_TEAM_PATIENCE_MS (default 50), the launch goes on with ONE workgroup per component: nothing
 * fails, no wait runs out (lfr_batch_spin_timeouts stays 0); the positions of such a component are those of a team of one (equal to the
 * team's to rounding, not bit for bit).  Returns how many components of the latest solve were solved that way (< 0: error). */
int64_t lfr_batch_team_fallbacks(lfr_batch *b);
/* positions: 2 * n_nodes doubles of the WHOLE graph; only this shard's nodes are written.  Waits for the
 * latest lfr_batch_solve of this batch, whatever stream it was issued on. */
int lfr_batch_download(lfr_batch *b, double *positions);
/* The same without the final host copy: *positions points at the batch's pinned staging buffer (2 * n_nodes
 * doubles, nodes outside the shard read 0), valid until the next solve / download / free of this batch. */
int lfr_batch_positions_view(lfr_batch *b, const double **positions);
/* The same in the precision the reference's SolutionFile holds (solve.cc:661-664 casts every displacement to float): converted on
 * the device, so half the bytes cross PCIe.  *positions: 2 * n_nodes floats in a pinned buffer of the batch, each the float nearest
 * to the double lfr_batch_positions_view would return; valid until the next solve / view / free of this batch. */
int lfr_batch_positions_view_f32(lfr_batch *b, const float **positions);
/* per solved component of the shard, in batch order: original component id, iterations,
 * termination, final cost (any pointer may be NULL). Returns the count. */
int64_t lfr_batch_component_info(lfr_batch *b, int64_t *component, int32_t *iterations, int32_t *termination,
                                 double *final_cost, int32_t *n_var_nodes, int32_t *n_edges);

/* Unit-level probe of the device arithmetic (cost.cc:13-48,78-90 + the loss / corrector of solve.cc:111,120): for
 * each of n edges (flows n x 18 float32, sim, kind 0 = intra-track/Cauchy 1 = inter-track/Tukey, x1 = source and
 * x2 = destination position) the kernels' eval_edge on the GPU: out8[8i..] = 0.5*rho, corrected residual r0 r1,
 * corrected d r / d x1 (j00 j01 j10 j11), sqrt(rho') (= d r / d x2 diagonal); cost_only[i] = 0.5*rho from the
 * cost-only variant the line search uses.  Test infrastructure for parity at 1e-12, not part of the solve path. */
int lfr_debug_eval_edges(int device, int64_t n, const float *flows, const float *sim, const int32_t *kind, const double *x1,
                         const double *x2, int tukey_variant, double *out8, double *cost_only);

/* Unit-level probe of the Armijo line search's step contraction (Ceres line_search.cc ArmijoLineSearch::DoSearch +
 * polynomial.cc MinimizeInterpolatingPolynomial): for each of n cases, samples holds 15 doubles = (x, value, gradient,
 * value_valid, gradient_valid) of the initial, the previous and the current sample; step[i] = the next step size the kernels
 * compute (negative: the search gives up); register_version 0 = the loop version the packed kernel calls, 1 = the unrolled
 * version of the workgroup-per-component kernel, 2 = the wave-cooperative form of the elimination-tree kernel (the pieces of the
 * root isolation a lane each; one case per wave).  Test infrastructure, not part of the solve path. */
int lfr_debug_ls_next_step(int device, int64_t n, const double *samples, const double *dir_max, int register_version, double *step);
/* The library's persistent host workers (they make the elimination-tree plans of a batch, solve.cc:79-143 for components above 192 rows:
 * the reference builds one problem per pool thread, solve.cc:617-635): `items` increments of one counter dealt to `threads` threads, the
 * caller among them, `reps` times.  Returns the number of increments performed (items * reps when nothing was lost).  Callable from
 * several threads at once (a caller that finds the workers busy runs on threads of its own).  Test infrastructure. */
int64_t lfr_debug_pool_selftest(int threads, int64_t items, int reps);

/* The pipeline's stable device sort of (key, value) pairs (graph stage: matches by similarity, solve.cc:489-497; assembly: nodes and
 * edges by component, solve.cc:563-597 - the reference sorts on the host with std::sort / visits in order): n pairs of key_bytes (4 or 8)
 * byte unsigned keys and 32-bit values, ascending by key bits [begin_bit, end_bit), equal keys in input order.  use_library 0 = the
 * library's driver (one-sweep radix passes with ONE clearing fill per sort above 256 K pairs, lfr_sort.hpp), 1 = rocprim::radix_sort_pairs.
 * Test infrastructure, not part of the solve path. */
int lfr_debug_sort_pairs(int device, int64_t n, int key_bytes, const void *keys, const uint32_t *vals, int begin_bit, int end_bit,
                         int use_library, void *keys_out, uint32_t *vals_out);

/* The pipeline's one-launch exclusive prefix sum (lfr_sort.hpp: decoupled look-back over states that lie in a stage's zero block): n items
 * of item_bytes (4 or 8) byte unsigned integers; out[i] = in[0] + ... + in[i-1] (32-bit sums wrap, 64-bit sums must stay below 2^62).
 * Test infrastructure, not part of the solve path. */
int lfr_debug_exclusive_sum(int device, int64_t n, int item_bytes, const void *in, void *out);

/* Keeps `workgroups` CUs of the device busy for `milliseconds` (512-thread workgroups at the elimination-tree kernel's register budget, on
 * a stream of their own; returns when they have started): the tests take CUs away from a solve with it (lfr_batch_team_fallbacks).
 * Test infrastructure, not part of the solve path. */
int lfr_debug_occupy(int device, int workgroups, double milliseconds);

/* One-call convenience used by the `solve` launcher: upload, solve, download on one device. */
int lfr_solve_hip(const lfr_problem *p, int device, int tukey_variant, double *positions,
                  lfr_solve_stats *stats);

/* The same over several GPUs of one node from ONE process: the solvable components are dealt to
 * `n_devices` shards (lfr_problem_shard_components), one host thread per device assembles (on its GPU, for a
 * labels-only problem) or uploads (host-assembled problem), solves and downloads its shard; shards write
 * disjoint node sets of `positions` (the thread pool of solve.cc:617-635 with GPUs as workers). */
int lfr_solve_hip_multi(const lfr_problem *p, const int *devices, int n_devices, int tukey_variant, double *positions,
                        lfr_solve_stats *stats);

/* solve.cc:487-641 over several GPUs from ONE process, the graph stage included: device k runs tracks / roots / components, the assembly
 * and the solve over the connected components of the match graph dealt to shard k (lfr_problem_build_hip_shard) - nothing is computed
 * on one GPU for the others.  problem_stats: what the stdout lines of solve.cc:534-606 need, summed / maximised over the shards.
 * A graph that is one connected component cannot be dealt out: every device then holds the whole problem and solves its share of the
 * components (as lfr_solve_hip_multi).  positions: 2 * n_nodes doubles, every node written (zeros where nothing is solved). */
int lfr_solve_graph_hip_multi(const lfr_graph *g, const int *devices, int n_devices, int64_t max_nodes_in_component, int tukey_variant,
                              double *positions, lfr_problem_stats *problem_stats, lfr_solve_stats *stats);

/* ---------------------------------------------------------------------------------------------
 * A12  SolutionFile emit.                                                     solve.cc:644-679
 * ------------------------------------------------------------------------------------------- */
/* Writes types.proto:30-46; returns through n_outside the count printed at solve.cc:666-670. */
int lfr_write_solution(const lfr_graph *g, const double *positions, const char *path, int64_t *n_outside);

/* The consumer's arithmetic as a library call (colmap_utils.py:126-137) — lets a caller skip the
 * SolutionFile round trip: for every node of `image_name`, keypoints[feature_idx][0] += dj*fact*16,
 * keypoints[feature_idx][1] += di*fact*16 (float32, same operation order), then +0.5 on both
 * coordinates of all `num_features` rows.  An image the graph does not know only gets the +0.5
 * (colmap_utils.py:127-128).  keypoints: num_features rows of `stride` floats (x, y, ...). */
int lfr_apply_displacements(const lfr_graph *g, const double *positions, const char *image_name, float *keypoints,
                            int64_t num_features, int64_t stride);

#ifdef __cplusplus
}
#endif
#endif /* LFR_H */

// Device-side graph stage: tracks (solve.cc:489-549), roots (solve.cc:552-582) and components
// (solve.cc:252-308) on the GPU, bit-identical to the host stage of lfr_graph.cpp; plus the device copies of
// the match graph (DevGraph) and of the labels (DevProblem) that the batch assembly continues from.
//
// The constrained maximum spanning forest is greedy over the globally sorted match list, i.e.
// order dependent — but only INSIDE a connected component of the match graph: two matches of
// different connected components never interact.  So (hard part 4):
//   1. plain connected components of the match graph, ignoring image conflicts (lock-free union-find)
//   2. ONE radix sort (rocPRIM) by (connected component, similarity descending), then the runs of equal similarities inside a
//      component put into the reference's (n1, n2) descending order in place (k_tie_fix).  Only when such a run is longer than
//      kMaxTieRun: the round-3 scheme, radix sorts by n2, by (similarity, n1) and - stable - by connected component.
//   3. (the segments of that list = the connected components)
//   4. small connected components: one thread each replays the reference's sequential union-find with the
//      image-conflict test over its own matches, in order.  Large ones (real match graphs are typically ONE
//      giant connected component: wrong matches link the tracks) run the same greedy rule in parallel ROUNDS:
//      a match whose two roots are equal or share an image can be retired at once (sets only grow, so the verdict
//      can never change); of the others, a match is accepted in this round iff no earlier pending match touches
//      either of its roots - such matches have pairwise disjoint roots and commute with everything before them,
//      so every union happens with exactly the operands it has in the sequential order (same roots, same sizes,
//      same tie rule).  Image sets are per-root bitsets (#images bits).
//   5. track ids = rank of the root nodes; roots = arg-max (score, node) per track; components =
//      connected components of the track meta-graph, numbered by their smallest track - these ARE the connected components of
//      step 1 (k_cc_min_track), a second union-find only runs after a cut (k_meta_union_cut).
// The whole chain is enqueued on ONE stream without host round trips: counts that size later steps
// (segments, tracks, components) stay on the device and bound the kernels there; every array is sized
// by its upper bound (N or M).  One 64-byte read-back at the end delivers the counts for the stdout lines
// (a second one, mid-way, tells whether large connected components exist and whether a long tie asks for the three sorts).  A component above the size cap
// needs the graph cut: the sequential priority-queue bisection stays on the host, fed with the device's tracks
// (lfr_graph.cpp: components_from_tracks), and the labels go back to HBM.
// Integer work throughout; the only floating-point accumulation (root scores, sums of float32
// similarities in fp64) is exact for any realistic input, hence order independent.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lfr_assemble.hpp"

namespace lfr {

namespace {

constexpr int kThreads = kPipeThreads;
constexpr int64_t kSerialSegmentEdges = 2048;     // connected components up to this many matches: one thread each
                                                   // (LFR_SERIAL_SEGMENT_EDGES overrides it: tests push everything through the rounds)
constexpr int kMaxTieRun = 64;                     // runs of equal similarities inside a connected component up to this long are ordered in place (k_tie_fix)
constexpr int kMaxRounds = 100000;                 // larger ones: parallel rounds (a path-shaped dependency chain this long: host stage)
constexpr size_t kMaxBitsetBytes = (size_t)24 << 30;
inline dim3 grid_for(int64_t n) { return pipe_grid(n); }

struct ArenaMark {                                 // temporaries: released at scope end (reuse is stream ordered)
    DevArena &a; size_t m;
    explicit ArenaMark(DevArena &ar) : a(ar), m(ar.top) {}
    ~ArenaMark() { a.top = m; }
};
#define TAKE(ptr, T, count)                                                                                   \
    T *ptr = arena.take_n<T>((size_t)(count));                                                                \
    if (!ptr) { set_error("graph stage: device arena exhausted (%s)", #ptr); return LFR_ERR_NOMEM; }

template <class K, class V>
int sort_pairs(DevArena &arena, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int begin_bit, int end_bit, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    LFR_HIP_TRY(sort_pairs_raw(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st));
    ArenaMark mark(arena);
    void *tmp = arena.take(bytes);
    if (!tmp) { set_error("graph stage: device arena exhausted (sort of %lld items)", (long long)n); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(sort_pairs_raw(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st));
    return LFR_OK;
}
// sums of the values of equal adjacent keys (rocPRIM reduce_by_key); *n_runs (device) = number of distinct runs
int sum_by_key(DevArena &arena, const unsigned long long *keys, unsigned long long *unique, const double *vals, double *sums, uint32_t *n_runs,
               int64_t n, hipStream_t st) {
    size_t bytes = 0;
    LFR_HIP_TRY(rocprim::reduce_by_key(nullptr, bytes, keys, vals, (size_t)n, unique, sums, n_runs, rocprim::plus<double>(), rocprim::equal_to<unsigned long long>(), st));
    void *tmp = arena.take(bytes);
    if (!tmp) { set_error("graph stage: device arena exhausted (reduce-by-key)"); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(rocprim::reduce_by_key(tmp, bytes, keys, vals, (size_t)n, unique, sums, n_runs, rocprim::plus<double>(), rocprim::equal_to<unsigned long long>(), st));
    return LFR_OK;
}

int exclusive_sum(DevArena &arena, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st) {
    if (n <= 0) return LFR_OK;
    size_t bytes = 0;
    LFR_HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, std::remove_cv_t<std::remove_reference_t<decltype(*out)>>(0), (size_t)n, rocprim::plus<std::remove_cv_t<std::remove_reference_t<decltype(*out)>>>(), st));
    ArenaMark mark(arena);
    void *tmp = arena.take(bytes);
    if (!tmp) { set_error("graph stage: device arena exhausted (scan of %lld items)", (long long)n); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(rocprim::exclusive_scan(tmp, bytes, in, out, std::remove_cv_t<std::remove_reference_t<decltype(*out)>>(0), (size_t)n, rocprim::plus<std::remove_cv_t<std::remove_reference_t<decltype(*out)>>>(), st));
    return LFR_OK;
}

__device__ __forceinline__ uint32_t sim_key(float s) {       // order-preserving float -> uint32
    if (s == 0.f) s = 0.f;                                    // -0.0 == +0.0
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sim_from_key(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// descending order of (sim, n1, n2) == ascending order of the complemented keys; node ids are complemented inside
// their node_bits bits (N-1-n) so that the radix sorts run over 32 + 2 x node_bits key bits instead of 96
__global__ void k_match_keys(int64_t M, uint32_t n_minus_1, int node_bits, const uint32_t *n1, const uint32_t *n2, const float *sim,
                             uint64_t *k_hi, uint32_t *k_lo, uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    k_hi[m] = ((uint64_t)(~sim_key(sim[m])) << node_bits) | (uint64_t)(n_minus_1 - n1[m]);
    k_lo[m] = n_minus_1 - n2[m];
    ids[m] = (uint32_t)m;
}
// The usual case needs ONE sort: connected components are known before any order is (the union-find of step 2 does not read one), so
// the key (component | ~similarity) groups and orders the matches in seven radix passes; the thirteen passes of the three-sort
// scheme above (n2, then (sim, n1), then the component, each stable) only break ties between EQUAL similarities inside a component, and
// k_tie_fix does that in place for the short runs real similarities produce.
__global__ void k_cc_sim_keys(int64_t M, const uint32_t *n1, const float *sim, const uint32_t *cc, uint64_t *key, uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    key[m] = ((uint64_t)cc[n1[m]] << 32) | (uint64_t)(~sim_key(sim[m]));
    ids[m] = (uint32_t)m;
}
// Round 6: the usual case needs no 52-bit sort either.  Real match graphs fall into small connected components (config 4: 147 k of
// 6..136 matches) - or into one giant one, which keeps the sort above.  The matches are grouped by component with a sort over the
// component bits alone (two 10-bit passes instead of six 9-bit ones over 64-bit keys), and inside a component every match finds its
// place by COUNTING the matches that precede it in the reference's order (descending (sim, n1, n2), equal triples by match id): the
// comparison is the full one, so there are no ties to repair and no long-tie fallback.  A workgroup stages the records of the segments
// that touch its 256 positions in LDS (16 B per match: ~sim key, n1, n2, id), so a record is gathered about once.
constexpr int kRankSortMax = 1024;                   // longest segment the counting handles (LDS window: 256 + 2 x longest records, 36 KB at most)
__global__ void k_match_records(int64_t M, const uint32_t *n1, const uint32_t *n2, const float *sim, const uint32_t *cc, uint4 *rec, uint32_t *key, uint32_t *ids) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t a = n1[m];
    rec[m] = make_uint4(~sim_key(sim[m]), a, n2[m], (uint32_t)m);
    key[m] = cc[a];
    ids[m] = (uint32_t)m;
}
__device__ __forceinline__ bool rank_precedes(const uint4 &q, const uint4 &me) {          // q comes before me in the reference's order
    return q.x < me.x || (q.x == me.x && (q.y > me.y || (q.y == me.y && (q.z > me.z || (q.z == me.z && q.w < me.w)))));
}
__global__ __launch_bounds__(kThreads) void k_rank_sort(int64_t M, const uint32_t *grouped, const uint4 *rec, const uint32_t *flags, const uint32_t *seg_id,
                                                        const uint32_t *starts, uint32_t *order, uint4 *sorted) {
    extern __shared__ uint4 s_rec[];                  // kThreads + 2 x (longest segment) records: the host knows the longest segment by now
    const int64_t b0 = (int64_t)blockIdx.x * kThreads;
    if (b0 >= M) return;
    const int64_t b1 = b0 + kThreads < M ? b0 + kThreads : M;
    const int64_t w_lo = starts[seg_id[b0] + flags[b0] - 1u], w_hi = starts[seg_id[b1 - 1] + flags[b1 - 1]];     // the segments that touch [b0, b1)
    for (int64_t j = w_lo + threadIdx.x; j < w_hi; j += kThreads) s_rec[j - w_lo] = rec[grouped[j]];
    __syncthreads();
    const int64_t i = b0 + threadIdx.x;
    if (i >= M) return;
    const uint32_t sg = seg_id[i] + flags[i] - 1u;
    const int64_t lo = starts[sg], hi = starts[sg + 1];
    const uint4 me = s_rec[i - w_lo];
    uint32_t rank = 0;
    for (int64_t j = lo; j < hi; ++j) rank += rank_precedes(s_rec[j - w_lo], me) ? 1u : 0u;
    order[lo + rank] = me.w;
    sorted[lo + rank] = me;                           // the record itself in the reference's order: k_kruskal_stream and k_scores_grouped read their matches as this stream
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
// The walks of a find read parent[] with WORKGROUP-scope atomic loads: they may be served from the CU's vector L1, i.e. be stale.  Sound:
// every value parent[x] ever held is an ancestor of x (or x itself) and, if not x, smaller than x, so a stale walk ends at an ancestor, and
// what decides a union is the agent-scope CAS - on a node that is no root any more it fails and hands back the current parent, from which
// uf_union continues (strictly smaller: it terminates); on a true root `a` with a stale, former root `b` < a it hooks a under a member of
// b's tree, which still joins the two trees, still points downwards, and leaves the root the smallest member.  Why not agent scope
// everywhere: in a giant connected component every find ends at the same root, and a single L2 channel serves ~2 G requests/s - the
// finds of config 5's 5.3 M matches queued on that one word (k_cc_union: 1.4 ms; the stages below and the gated halving of uf_find are the rest of that story).
__device__ __forceinline__ uint32_t uf_load(const uint32_t *parent, uint32_t x) {
    return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kHalveAfterHops = 3;
__device__ __forceinline__ uint32_t uf_load_near(const uint32_t *parent, uint32_t x) {
    return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    uint32_t p = uf_load_near(parent, x);
    int hops = 0;
    while (p != x) {
        // path halving, but only on walks that turn out long: thousands of threads see the same short stale path below a hot root, and
        // their (agent-scope) halving stores of the same words queued at the L2 - config 5: 1.0-1.6 ms with them, 0.15 without
        const uint32_t gp = uf_load_near(parent, p);
        if (gp != p && ++hops > kHalveAfterHops) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// Real match graphs are ONE giant connected component: millions of unions then meet at a handful of roots.  So the unions run in
// stages (the sampling idea of Afforest, Sutton et al. 2018): every 32nd match first, then every 4th, then the rest, the parents
// flattened in between (k_uf_flatten).  The giant component forms in a stage with few threads, and from the second stage on a match
// whose ends already show the same parent is skipped without touching an atomic.  That test reads parent[] with plain loads: any
// value parent[x] ever held is an ancestor of x in the final forest (a root is hooked under another root, a halving store writes a
// grandparent, the flattening writes the root), so equal parents - however stale - prove that the ends are connected; unequal ones fall
// through to the lock-free union.  The labels (smallest member) do not depend on the schedule.  Config 5 (5.3 M matches, one giant
// component): 1.4 ms as one pass with agent-scope walks, 0.15 ms now; config 4 (147 k small components): 0.14 ms either way.
constexpr int kUnionStages = 3;
constexpr int kUnionStrides[kUnionStages] = {32, 4, 1};
__device__ __forceinline__ bool uf_connected_hint(const uint32_t *parent, uint32_t a, uint32_t b) { return parent[a] == parent[b]; }
__global__ void k_uf_flatten(int64_t n, uint32_t *parent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i, p = uf_load(parent, x);
    if (p == x) return;
    while (p != x) { x = p; p = uf_load(parent, x); }
    __hip_atomic_store(&parent[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (a walker through i sees the old ancestor or the root)
}
// one stage: the matches 0, stride, 2 stride, ... except those an earlier stage (multiples of done_stride, 0 = none) has united already
__global__ void k_cc_union(int64_t M, int stride, int done_stride, const uint32_t *n1, const uint32_t *n2, uint32_t *parent) {
    const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (m >= M || (done_stride && m % done_stride == 0)) return;
    const uint32_t a = n1[m], b = n2[m];
    if (done_stride && uf_connected_hint(parent, a, b)) return;
    uf_union(parent, a, b);
}
// Labels = roots, written to a SEPARATE array by a read-only walk.  (Flattening in place - parent[i] = find(i) with
// path halving - is wrong under concurrency: another thread's halving store, computed from an older read of
// parent[i], can land after the flattened value and leave an intermediate ancestor there; a giant connected
// component then appears split into two labels.)
__global__ void k_cc_labels(int64_t n, const uint32_t *parent, uint32_t *label) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i, p = uf_load(parent, x);
    while (p != x) { x = p; p = uf_load(parent, x); }
    label[i] = x;
}
__global__ void k_cc_keys(int64_t M, const uint32_t *order, const uint32_t *n1, const uint32_t *cc, uint32_t *keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) keys[i] = cc[n1[order[i]]];
}
__global__ void k_seg_flags(int64_t M, const uint32_t *keys, uint32_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_seg_flags_hi(int64_t M, const uint64_t *keys, uint32_t *flags) {                // (component | ~similarity) keys
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flags[i] = (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32)) ? 1u : 0u;
}
// counts[] layout (device): what the host reads back at the end of the stage
enum { CNT_SEG = 0, CNT_TRACKS, CNT_COMPS, CNT_MAX_TRACK, CNT_MAX_COMP, CNT_MAX_SEG, CNT_LONG_TIE, CNT_WORDS = 16 };
// Runs of equal (component, similarity) in the one-sort order: the reference's order inside a run is descending (n1, n2) (solve.cc:489,
// std::sort + reverse of (sim, n1, n2) tuples; equal triples are the same union whichever comes first - match id ascending, as the
// stable three-sort scheme leaves them).  The head of a run sorts it by insertion; a run above max_run (quantized similarities in a
// giant component) sets CNT_LONG_TIE and the host orders the list with the three sorts instead.
__global__ void k_tie_fix(int64_t M, const uint64_t *key, uint32_t *order, const uint32_t *n1, const uint32_t *n2, int max_run, uint32_t *counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint64_t k = key[i];
    if ((i > 0 && key[i - 1] == k) || i + 1 >= M || key[i + 1] != k) return;      // inside a run / a run of one
    int64_t j = i + 2;
    while (j < M && j - i <= max_run && key[j] == k) ++j;
    if (j - i > max_run) { counts[CNT_LONG_TIE] = 1u; return; }
    for (int64_t a = i + 1; a < j; ++a) {
        const uint32_t m = order[a], m1 = n1[m], m2 = n2[m];
        int64_t b = a;
        while (b > i) {
            const uint32_t q = order[b - 1], q1 = n1[q], q2 = n2[q];
            if (q1 > m1 || (q1 == m1 && (q2 > m2 || (q2 == m2 && q < m)))) break;  // q stays in front of m
            order[b] = q; --b;
        }
        order[b] = m;
    }
}
// seg_id[i] = exclusive count of flags; seg_id[M] = number of segments
__global__ void k_seg_starts(int64_t M, const uint32_t *flags, const uint32_t *seg_id, uint32_t *starts, uint32_t *counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M && flags[i]) starts[seg_id[i]] = (uint32_t)i;
    if (i == 0) { starts[seg_id[M]] = (uint32_t)M; counts[CNT_SEG] = seg_id[M]; }
}
// A maximum into ONE word: atomics on one address take their turns at the L2 (14 k waves, one atomic each, were 28 us of a 5-us kernel;
// looking first with an agent-scope load made it 44), so these kernels run as a few hundred workgroups that stride over the items and
// send one maximum each.
constexpr int kFewBlocks = 512;
inline dim3 grid_few(int64_t n) { const dim3 g = grid_for(n); return dim3(g.x < (unsigned)kFewBlocks ? g.x : (unsigned)kFewBlocks); }
__device__ __forceinline__ void block_atomic_max(uint32_t v, uint32_t *dst) {          // every thread of the workgroup calls it
    __shared__ uint32_t s_max[kThreads / 64];
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kThreads / 64; ++i) v = max(v, s_max[i]);
        if (v) atomicMax(dst, v);
    }
}
__global__ void k_seg_maxlen(int64_t cap, const uint32_t *starts, uint32_t *counts) {
    const int64_t n = cap < (int64_t)counts[CNT_SEG] ? cap : (int64_t)counts[CNT_SEG];
    uint32_t v = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) v = max(v, starts[s + 1] - starts[s]);
    block_atomic_max(v, &counts[CNT_MAX_SEG]);
}

// ---- the reference's greedy constrained union-find, one thread per connected component ----
// parent: -1 = root (solve.cc:492); lists of member nodes replace images_in_track (every member has
// a distinct image, so |images_in_track[root]| == count[root]).
// Image sets as 128-bit signatures (one hashed bit per image, OR-ed on union): two roots whose signatures do not intersect share no
// image, so the exact test - a walk over both member lists, the bulk of this kernel's dependent loads - only runs when the signatures
// collide (config 4: ~6-node tracks in 1344 images, a few per cent of the unions).
__device__ __forceinline__ ulonglong2 image_signature(int32_t im) {
    const uint32_t h = (uint32_t)im * 0x9E3779B1u;
    const uint32_t b = h >> 25;                               // 7 bits
    ulonglong2 m;
    m.x = b < 64u ? 1ull << b : 0ull;
    m.y = b < 64u ? 0ull : 1ull << (b - 64u);
    return m;
}
// One thread replays its component's matches in order; what it waits for is memory - 2-3 waves per SIMD, every access a miss of the CU's
// L1, and each match used to be a chain of ~8 dependent loads (order -> ends -> one root -> the other -> signatures -> counts -> tail).  So
// the loads that do not depend on each other go out together: the next matches' ends two iterations ahead, the two walks to the roots hop
// for hop, everything the union reads from its two roots at once.  Three trips per match instead of eight: 0.42 -> 0.28 ms on config 4
// (an LDS copy of the segment and a records pre-pass had not paid - the chain, not the gathers, was the time; guessing the root one trip
// early does not pay either: what is left is the lockstep of 64 components of different lengths per wave).
__device__ __forceinline__ void kruskal_segment(const int64_t lo, const int64_t hi, const uint32_t *order, const uint32_t *n1,
                                                const uint32_t *n2, const int32_t *node_image, int32_t *parent, int32_t *next, int32_t *tail,
                                                int32_t *count, ulonglong2 *sig) {
    uint32_t m1 = order[lo], m2 = lo + 1 < hi ? order[lo + 1] : 0u;           // matches k and k + 1
    uint32_t a1 = n1[m1], b1 = n2[m1];
    for (int64_t k = lo; k < hi; ++k) {
        const int32_t a = (int32_t)a1, b = (int32_t)b1;
        if (k + 1 < hi) { a1 = n1[m2]; b1 = n2[m2]; }         // the ends of match k + 1, the id of match k + 2: in flight during this union
        if (k + 2 < hi) m2 = order[k + 2];
        int32_t r1 = a, r2 = b, p1 = parent[a], p2 = parent[b];
        while (p1 >= 0 || p2 >= 0) {                          // (either walk has a hop left)
            const bool h1 = p1 >= 0, h2 = p2 >= 0;
            if (h1) r1 = p1;
            if (h2) r2 = p2;
            const int32_t q1 = h1 ? parent[r1] : -1, q2 = h2 ? parent[r2] : -1;
            p1 = q1; p2 = q2;
        }
        if (r1 != a) parent[a] = r1;                          // (compression of the two starting nodes: sizes decide the unions, not the paths)
        if (r2 != b) parent[b] = r2;
        if (r1 == r2) continue;
        const ulonglong2 s1 = sig[r1], s2 = sig[r2];
        const int32_t c1 = count[r1], c2 = count[r2], t1 = tail[r1], t2 = tail[r2];
        bool conflict = false;                               // solve.cc:506-511
        if ((s1.x & s2.x) | (s1.y & s2.y)) {                 // the signatures collide: the exact test
            for (int32_t i = r1; i >= 0 && !conflict; i = next[i]) {
                const int32_t im = node_image[i];
                for (int32_t j = r2; j >= 0; j = next[j]) if (node_image[j] == im) { conflict = true; break; }
            }
        }
        if (conflict) continue;
        const bool swap = c1 < c2;                           // solve.cc:513-521 (ties: root2 under root1)
        const int32_t big = swap ? r2 : r1, small = swap ? r1 : r2;
        parent[small] = big;
        next[swap ? t2 : t1] = small; tail[big] = swap ? t1 : t2; count[big] = c1 + c2;
        sig[big] = make_ulonglong2(s1.x | s2.x, s1.y | s2.y);
    }
}
__global__ void k_kruskal(int64_t cap, int64_t serial_limit, const uint32_t *counts, const uint32_t *starts, const uint32_t *order, const uint32_t *n1,
                          const uint32_t *n2, const int32_t *node_image, int32_t *parent, int32_t *next, int32_t *tail,
                          int32_t *count, ulonglong2 *sig) {
    // (Round 6, measured and dropped: dealing the segments to the threads BY LENGTH.  In the order they come in, config 4's 147 k components
    // of 6..136 matches keep a wave busy 3.3 times the mean length; dealt by length over the whole list the factor is 1.00 and the kernel
    // ran 613 us instead of 288, dealt inside a workgroup of 256 / 512 / 1024 threads through LDS 324 / 329 / 375 us.  Neighbouring
    // threads replaying neighbouring components - one stretch of the ordered list, nodes numbered close together - is worth more than
    // equal lengths: the kernel is bound by memory transactions, not by its longest lane.  profiles/r06_ab/kruskal_by_length.txt)
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap || s >= (int64_t)counts[CNT_SEG]) return;
    const int64_t lo = starts[s], hi = starts[s + 1];
    if (hi - lo > serial_limit) return;                      // large connected component: parallel rounds (k_round_*)
    kruskal_segment(lo, hi, order, n1, n2, node_image, parent, next, tail, count, sig);
}
// The same rule fed by the stream of sorted records (round 6, counting road: k_rank_sort writes every record to its place in the
// reference's order, 16 bytes per match): no id from `order`, no gathered ends - two of the dozen scattered 32-byte sectors a match
// costs the kernel above.  Config 4: 284 -> 206 us, k_rank_sort + 4 us for the records.  Measured against it and dropped: the component's
// whole state in LDS under local node numbers (one thread per component, [slot][thread] columns, images and parents out once per
// node): 363 us - a wave's lanes diverge in every inner loop (look-up among the known ids, root walks, the image test), so every match
// costs the longest path of 64 components, where the scattered version hides its latency behind 32 waves per CU
// (profiles/r06_ab/kruskal_variants.txt).
__global__ void k_kruskal_stream(int64_t cap, int64_t serial_limit, const uint32_t *counts, const uint32_t *starts, const uint4 *sorted,
                                 const int32_t *node_image, int32_t *parent, int32_t *next, int32_t *tail, int32_t *count, ulonglong2 *sig) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap || s >= (int64_t)counts[CNT_SEG]) return;
    const int64_t lo = starts[s], hi = starts[s + 1];
    if (hi - lo > serial_limit) return;
    uint4 q = sorted[lo];
    for (int64_t k = lo; k < hi; ++k) {
        const int32_t a = (int32_t)q.y, b = (int32_t)q.z;
        if (k + 1 < hi) q = sorted[k + 1];
        int32_t r1 = a, r2 = b, p1 = parent[a], p2 = parent[b];
        while (p1 >= 0 || p2 >= 0) {
            const bool h1 = p1 >= 0, h2 = p2 >= 0;
            if (h1) r1 = p1;
            if (h2) r2 = p2;
            const int32_t q1 = h1 ? parent[r1] : -1, q2 = h2 ? parent[r2] : -1;
            p1 = q1; p2 = q2;
        }
        if (r1 != a) parent[a] = r1;
        if (r2 != b) parent[b] = r2;
        if (r1 == r2) continue;
        const ulonglong2 s1 = sig[r1], s2 = sig[r2];
        const int32_t c1 = count[r1], c2 = count[r2], t1 = tail[r1], t2 = tail[r2];
        bool conflict = false;
        if ((s1.x & s2.x) | (s1.y & s2.y)) {
            for (int32_t i = r1; i >= 0 && !conflict; i = next[i]) {
                const int32_t im = node_image[i];
                for (int32_t j = r2; j >= 0; j = next[j]) if (node_image[j] == im) { conflict = true; break; }
            }
        }
        if (conflict) continue;
        const bool swap = c1 < c2;
        const int32_t big = swap ? r2 : r1, small = swap ? r1 : r2;
        parent[small] = big;
        next[swap ? t2 : t1] = small; tail[big] = swap ? t1 : t2; count[big] = c1 + c2;
        sig[big] = make_ulonglong2(s1.x | s2.x, s1.y | s2.y);
    }
}
// parent[] is shared with k_kruskal (-1 = root).  Finds compress by halving; within the evaluation kernel no union
// happens, so a stale pointer is still an ancestor.
__device__ __forceinline__ int32_t par_load(const int32_t *parent, int32_t x) {
    return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t par_find(int32_t *parent, int32_t x) {
    for (;;) {
        const int32_t p = par_load(parent, x);
        if (p < 0) return x;
        const int32_t gp = par_load(parent, p);
        if (gp >= 0) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
    }
}
// append to a list with one atomic per wave
__device__ __forceinline__ uint32_t wave_append(bool keep, uint32_t *counter) {
    const unsigned long long mask = __ballot(keep);
    const int lane = (int)(threadIdx.x & 63);
    uint32_t base = 0;
    if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, leader, 64);
    }
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// ... and with one atomic per WORKGROUP, ITEMS candidates per thread: a returning atomic on one address costs ~10 ns at the L2 whoever
// sends it, so a list of millions built wave by wave (83 k atomics for config 5's 5.3 M matches) spent 0.25-1 ms per pass on its counter.
// Every thread of the (kThreads-wide) workgroup must call it.
template <int ITEMS>
__device__ __forceinline__ void block_append(const bool (&keep)[ITEMS], uint32_t *counter, uint32_t (&at)[ITEMS]) {
    __shared__ uint32_t s_wave[kThreads / 64];
    __shared__ uint32_t s_base;
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned long long mask = __ballot(keep[j]);
        at[j] = run + (uint32_t)__popcll(mask & below);
        run += (uint32_t)__popcll(mask);
    }
    if (lane == 0) s_wave[w] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int i = 0; i < kThreads / 64; ++i) { const uint32_t c = s_wave[i]; s_wave[i] = total; total += c; }
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t base = s_base + s_wave[w];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) at[j] += base;
}
constexpr int kAppendItems = 8;
inline dim3 grid_for_items(int64_t n, int items) { return grid_for((n + items - 1) / items); }
struct Pending { uint32_t k; int32_t ra, rb; };     // position in the ordered match list, roots of n1 / n2
// a bit that many threads set: a plain (possibly stale) look saves the atomic once the bit is visible - a node of config 5 has ~70 matches
__device__ __forceinline__ void set_bit(unsigned long long *word, unsigned long long mask) {
    if (!(*word & mask)) atomicOr(word, mask);
}
// matches of large connected components -> the first pending list; bit of the own image for their nodes
__global__ void k_large_pending(int64_t k_lo, int64_t k_hi, int64_t serial_limit, const uint32_t *flags, const uint32_t *seg_id, const uint32_t *starts, const uint32_t *order,
                                const uint32_t *n1, const uint32_t *n2, const int32_t *node_image, int W, unsigned long long *bits,
                                Pending *pend, uint32_t *n_pend) {
    const int64_t k0 = k_lo + (int64_t)blockIdx.x * (kThreads * kAppendItems) + threadIdx.x;      // positions [k_lo, k_hi) of the ordered list
    bool large[kAppendItems];
    uint32_t at[kAppendItems];
#pragma unroll
    for (int j = 0; j < kAppendItems; ++j) {
        const int64_t k = k0 + (int64_t)j * kThreads;
        large[j] = false;
        if (k < k_hi) {
            const uint32_t s = seg_id[k] + flags[k] - 1u;           // segment of position k
            large[j] = (int64_t)(starts[s + 1] - starts[s]) > serial_limit;
        }
    }
    block_append(large, n_pend, at);
#pragma unroll
    for (int j = 0; j < kAppendItems; ++j) {
        if (!large[j]) continue;
        const int64_t k = k0 + (int64_t)j * kThreads;
        const uint32_t m = order[k], a = n1[m], b = n2[m];
        pend[at[j]] = Pending{(uint32_t)k, (int32_t)a, (int32_t)b};
        if (bits) {                                                  // (idempotent: every match of a node sets the same bit - look first)
            set_bit(&bits[(size_t)a * W + (node_image[a] >> 6)], 1ull << (node_image[a] & 63));
            set_bit(&bits[(size_t)b * W + (node_image[b] >> 6)], 1ull << (node_image[b] & 63));
        }
    }
}
// retire what can never be accepted (same root / shared image: monotone), bid for the roots with the rest
// (one call per wave-wide slice of the pending list: every lane of the wave calls it, `valid` says whether it holds an entry)
__device__ __forceinline__ bool round_evaluate(bool valid, Pending &q, int32_t *parent, const unsigned long long *bits, int W) {
    if (!valid) return false;
    q.ra = par_find(parent, q.ra); q.rb = par_find(parent, q.rb);
    if (q.ra == q.rb) return false;
    const unsigned long long *A = bits + (size_t)q.ra * W, *B = bits + (size_t)q.rb * W;
    unsigned long long any = 0ull;
    for (int w = 0; w < W; ++w) any |= A[w] & B[w];               // solve.cc:506-511
    return any == 0ull;
}
__device__ __forceinline__ void round_bid(const Pending q, unsigned long long round_hi, unsigned long long *minpos) {
    const unsigned long long key = round_hi | q.k;                // a later round's bid beats every stale entry
    // thousands of matches bid for the root of a grown track and all but one lose: look before bidding (the value only
    // decreases, so a bid that does not beat what is already there could not have won)
    if (key < __hip_atomic_load(&minpos[q.ra], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&minpos[q.ra], key);
    if (key < __hip_atomic_load(&minpos[q.rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&minpos[q.rb], key);
}
__device__ __forceinline__ void round_eval_one(bool valid, Pending q, int32_t *parent, const unsigned long long *bits, int W,
                                               unsigned long long round_hi, unsigned long long *minpos, Pending *out, uint32_t *n_out) {
    const bool keep = round_evaluate(valid, q, parent, bits, W);
    const uint32_t at = wave_append(keep, n_out);
    if (!keep) return;
    out[at] = q;
    round_bid(q, round_hi, minpos);
}
constexpr int kEvalItems = 4;
__global__ void k_round_eval(const uint32_t *n_in_p, const Pending *in, int32_t *parent, const unsigned long long *bits, int W,
                             unsigned long long round_hi, unsigned long long *minpos, Pending *out, uint32_t *n_out) {
    const uint32_t i0 = blockIdx.x * (kThreads * kEvalItems) + threadIdx.x;
    const uint32_t n_in = *n_in_p;                                // (the launch is sized by an upper bound: the count of an earlier round)
    if (blockIdx.x * (kThreads * kEvalItems) >= n_in) return;     // (uniform over the workgroup)
    Pending q[kEvalItems];
    bool keep[kEvalItems];
    uint32_t at[kEvalItems];
#pragma unroll
    for (int j = 0; j < kEvalItems; ++j) {
        const uint32_t i = i0 + j * kThreads;
        const bool valid = i < n_in;
        q[j] = valid ? in[i] : Pending{0, 0, 0};
        keep[j] = round_evaluate(valid, q[j], parent, bits, W);
    }
    block_append(keep, n_out, at);
#pragma unroll
    for (int j = 0; j < kEvalItems; ++j) if (keep[j]) { out[at[j]] = q[j]; round_bid(q[j], round_hi, minpos); }
}
// counters of a batch of rounds: c[j] = matches pending before round j of the batch; the last count of the previous batch moves to the front
__global__ void k_round_counters_shift(uint32_t *c, int R) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { c[0] = c[R]; for (int j = 1; j <= R; ++j) c[j] = 0u; }
}
// a bidder that holds both of its roots is the earliest pending match touching either: accept (solve.cc:513-521)
__device__ __forceinline__ void round_accept_one(const Pending q, unsigned long long round_hi, const unsigned long long *minpos,
                                                 int32_t *parent, int32_t *count, unsigned long long *bits, int W, uint32_t *n_accepted) {
    const unsigned long long key = round_hi | q.k;
    if (minpos[q.ra] != key || minpos[q.rb] != key) return;
    int32_t big = q.ra, small = q.rb;                              // r1 = root of n1, r2 = root of n2
    if (count[q.ra] < count[q.rb]) { big = q.rb; small = q.ra; }   // ties: root2 under root1
    __hip_atomic_store(&parent[small], big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    count[big] += count[small];
    unsigned long long *A = bits + (size_t)big * W;
    const unsigned long long *B = bits + (size_t)small * W;
    for (int w = 0; w < W; ++w) A[w] |= B[w];
    atomicAdd(n_accepted, 1u);
}
__global__ void k_round_accept(const uint32_t *n_p, const Pending *pend, unsigned long long round_hi, const unsigned long long *minpos,
                               int32_t *parent, int32_t *count, unsigned long long *bits, int W, uint32_t *n_accepted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_p) return;
    round_accept_one(pend[i], round_hi, minpos, parent, count, bits, W, n_accepted);
}

// Every prefix block and every round of the parallel greedy rule in ONE cooperative launch (opt-in, LFR_ROUNDS_COOPERATIVE=1: see the
// measurements at its launch site): the phases that are kernels in the default path (first pending list of a block, evaluation,
// acceptance) are separated by grid-wide barriers instead of launches and read-backs.  c[0], c[1]: entries in list A / list B; c[2]: rounds run; c[3]: matches accepted; c[4]: 1 = the round limit was hit.
// Grid barrier of k_rounds_all (the launch is cooperative, so every workgroup is resident): one monotonic arrival counter per
// eighth of the grid (one cache line each) and a release word; the last arrival of the last group releases.  Agent-scope fences on both
// sides: the XCDs' L2 caches are not coherent with each other inside a kernel.  (cooperative_groups' grid.sync() took ~130 us per
// barrier here - 20 ms for config 5's 76 rounds against 4.4 ms for launch-per-round.)
__device__ __forceinline__ void rounds_barrier(uint32_t *bar, uint32_t &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t grp = blockIdx.x & 7u, in_grp = (gridDim.x - grp + 7u) >> 3;
        if (__hip_atomic_fetch_add(&bar[16 * (1 + grp)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gen * in_grp) {
            const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u;
            if (__hip_atomic_fetch_add(&bar[16 * 9], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gen * groups)
                __hip_atomic_store(&bar[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
#ifndef LFR_ROUNDS_TAIL_PENDING
#define LFR_ROUNDS_TAIL_PENDING 0
#endif
#ifndef LFR_ROUNDS_TAIL_WGS
#define LFR_ROUNDS_TAIL_WGS 4
#endif
struct RoundsArgs {
    int64_t M, first_block, serial_limit;
    const uint32_t *flags, *seg_id, *starts, *order, *n1, *n2;
    const int32_t *node_image;
    int W, max_rounds;
    unsigned long long *bits, *minpos;
    Pending *pa, *pb;
    int32_t *parent, *count;
    uint32_t *c;
    uint32_t *bar;           // 160 zeroed words
    int resume, launched0;   // resume = 1: the rounds of ONE prefix block whose pending list (pa, c[0] entries) exists already; launched0 rounds have run
};
// Round 5: the same loop on ONE XCD (k_rounds_all<true>, a plain launch).  A round of the greedy rule is a chain of dependent accesses to
// words everybody shares (parents, bids, image sets); through agent-scope fences every hop leaves the XCD (~2 us) and a barrier writes the
// whole L2 back, and as separate launches a round costs two kernel boundaries (~43 us for a few thousand pending matches: 76 rounds = 3.3
// ms of config 5's graph stage).  The workgroups that land on one XCD share its L2: the barrier there is an arrival counter, `s_waitcnt
// vmcnt(0)` before it (every store has reached L2) and an L1 invalidate behind it - no write-back, nothing leaves the XCD.  The launch
// is eight times the workgroups that are wanted; every workgroup registers, the ones on other XCDs than the first registrant's leave, the
// others learn their number and their count when the whole grid has registered.  bar[150]: chosen XCC + 1, [151]: participants,
// [152]: registered, [153]: arrivals.
__device__ __forceinline__ void rounds_barrier_xcd(uint32_t *bar, uint32_t &gen, const uint32_t n_blk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        __hip_atomic_fetch_add(&bar[153], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; (int)(__hip_atomic_load(&bar[153], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen * n_blk) < 0 && spins < (1 << 24); ++spins) __builtin_amdgcn_s_sleep(1);   // (bounded: a miscount ends as garbage labels the tests catch, not as a hung GPU)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // (buffer_inv sc1: this CU's L1; the L2 is the XCD's own)
    }
    __syncthreads();
}
template <bool XCD>
__global__ void __launch_bounds__(kPipeThreads) k_rounds_all(RoundsArgs a) {
    uint32_t gen = 0;
    uint32_t n_blk = gridDim.x, blk = blockIdx.x;
    if (XCD) {
        __shared__ uint32_t s_reg[2];
        if (threadIdx.x == 0) {
            const uint32_t xcc = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u) + 1u;
            uint32_t chosen = atomicCAS(&a.bar[150], 0u, xcc);
            if (chosen == 0u) chosen = xcc;
            uint32_t idx = 0xffffffffu;
            if (chosen == xcc) idx = __hip_atomic_fetch_add(&a.bar[151], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the participant count is out before the total says "everyone has registered")
            __hip_atomic_fetch_add(&a.bar[152], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx != 0xffffffffu) for (int spins = 0; __hip_atomic_load(&a.bar[152], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins < (1 << 24); ++spins) __builtin_amdgcn_s_sleep(4);
            s_reg[0] = idx; s_reg[1] = __hip_atomic_load(&a.bar[151], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        blk = s_reg[0]; n_blk = s_reg[1];
        if (blk == 0xffffffffu) return;
    }
    auto barrier = [&]() { if (XCD) rounds_barrier_xcd(a.bar, gen, n_blk); else rounds_barrier(a.bar, gen); };
    const uint32_t n_thr = n_blk * blockDim.x, tid = blk * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    Pending *pa = a.pa, *pb = a.pb;
    int cur = 0;                                                  // c[cur] counts pa; both counters are 0 between blocks
    int launched = a.launched0;
    for (int64_t k_lo = 0, size = a.first_block; k_lo < a.M; k_lo += size, size *= 2) {
        const int64_t k_hi = k_lo + size < a.M ? k_lo + size : a.M;
        for (int64_t b = k_lo + (tid - lane); !a.resume && b < k_hi; b += n_thr) {          // (wave-uniform trip count: wave_append is a wave operation)
            const int64_t k = b + lane;
            bool large = false;
            uint32_t m = 0;
            if (k < k_hi) {
                const uint32_t sg = a.seg_id[k] + a.flags[k] - 1u;
                large = (int64_t)(a.starts[sg + 1] - a.starts[sg]) > a.serial_limit;
                m = a.order[k];
            }
            const uint32_t at = wave_append(large, &a.c[cur]);
            if (large) {
                const uint32_t x = a.n1[m], y = a.n2[m];
                pa[at] = Pending{(uint32_t)k, (int32_t)x, (int32_t)y};
                atomicOr(&a.bits[(size_t)x * a.W + (a.node_image[x] >> 6)], 1ull << (a.node_image[x] & 63));
                atomicOr(&a.bits[(size_t)y * a.W + (a.node_image[y] >> 6)], 1ull << (a.node_image[y] & 63));
            }
        }
        barrier();
        for (;;) {
            const uint32_t n_in = __hip_atomic_load(&a.c[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_in == 0) break;
            if (launched >= a.max_rounds) { if (tid == 0) a.c[4] = 1u; return; }        // (uniform: every thread read the same count)
            const unsigned long long round_hi = (unsigned long long)(kMaxRounds - launched) << 32;
            for (uint32_t b = tid - lane; b < n_in; b += n_thr) {
                const uint32_t i = b + lane;
                const bool valid = i < n_in;
                round_eval_one(valid, valid ? pa[i] : Pending{0, 0, 0}, a.parent, a.bits, a.W, round_hi, a.minpos, pb, &a.c[cur ^ 1]);
            }
            barrier();
            const uint32_t n_out = __hip_atomic_load(&a.c[cur ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t i = tid; i < n_out; i += n_thr) round_accept_one(pb[i], round_hi, a.minpos, a.parent, a.count, a.bits, a.W, &a.c[3]);
            if (tid == 0) a.c[cur] = 0u;                          // (everybody read n_in before the barrier above)
            barrier();
            Pending *t = pa; pa = pb; pb = t;
            cur ^= 1;
            ++launched;
        }
        if (a.resume) break;                                      // (one block: the host has the next one's pending list made by the whole chip)
    }
    if (tid == 0) a.c[2] = (uint32_t)launched;
}

// ---- size cap (solve.cc:311-364): the cut itself runs on the host, on the handful of inter-track matches it needs ----
// component (pre-cut) of every track, and the inter-track matches of components above the cap, compacted as
// (unordered track pair, similarity): key = min << 32 | max
__global__ void k_track_comp(int64_t cap, const uint32_t *counts, const uint32_t *mlabel, const uint32_t *rank, int32_t *tcomp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cap && t < (int64_t)counts[CNT_TRACKS]) tcomp[t] = (int32_t)rank[mlabel[t]];
}
__global__ void k_cut_edges(int64_t M, const uint32_t *n1, const uint32_t *n2, const float *sim, const int32_t *track, const int32_t *tcomp,
                            const uint32_t *csize, uint32_t max_nodes, unsigned long long *pair_key, double *pair_sim, uint32_t *n_list) {
    const int64_t m0 = (int64_t)blockIdx.x * (kThreads * kAppendItems) + threadIdx.x;
    bool keep[kAppendItems];
    uint32_t at[kAppendItems];
    int32_t ta[kAppendItems], tb[kAppendItems];
#pragma unroll
    for (int j = 0; j < kAppendItems; ++j) {
        const int64_t m = m0 + (int64_t)j * kThreads;
        keep[j] = false; ta[j] = tb[j] = 0;
        if (m < M) {
            ta[j] = track[n1[m]]; tb[j] = track[n2[m]];
            keep[j] = ta[j] != tb[j] && csize[tcomp[ta[j]]] > max_nodes;
        }
    }
    block_append(keep, n_list, at);
#pragma unroll
    for (int j = 0; j < kAppendItems; ++j) {
        if (!keep[j]) continue;
        pair_key[at[j]] = ((unsigned long long)(uint32_t)min(ta[j], tb[j]) << 32) | (uint32_t)max(ta[j], tb[j]);
        pair_sim[at[j]] = (double)sim[m0 + (int64_t)j * kThreads];
    }
}
// meta union without the cut edges (solve.cc:346-353): gc[t] = subset of track t inside its oversized component, -1 elsewhere
__global__ void k_meta_union_cut(int64_t M, const uint32_t *n1, const uint32_t *n2, const int32_t *track, const int32_t *gc, uint32_t *parent) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int32_t ta = track[n1[m]], tb = track[n2[m]];
    if (ta == tb) return;
    if (gc[ta] >= 0 && gc[ta] != gc[tb]) return;        // both tracks are in the same oversized component: the edge was cut
    uf_union(parent, (uint32_t)ta, (uint32_t)tb);
}
__global__ void k_init_nodes(int64_t n, const int32_t *node_image, int32_t *parent, int32_t *next, int32_t *tail, int32_t *count, ulonglong2 *sig) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = -1; next[i] = -1; tail[i] = (int32_t)i; count[i] = 1; sig[i] = image_signature(node_image[i]); }
}
__global__ void k_root_flags(int64_t n, const int32_t *parent, uint32_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = parent[i] < 0 ? 1u : 0u;
}
// rank[] = exclusive scan of the root flags over N+1 entries: rank[N] = number of tracks
__global__ void k_track_ids(int64_t n, const int32_t *parent, const uint32_t *rank, int32_t *track, uint32_t *tsize, uint32_t *counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) counts[CNT_TRACKS] = rank[n];
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
// The same sums with the matches taken in the ORDERED list (grouped by connected component): the 64 matches of a wave then touch a few
// dozen nodes, each several times, and the wave adds them up in LDS first - 128 slots per wave, claimed by compare-and-swap on the node id, a
// contribution that finds its slot taken by another node goes to memory directly - and sends one fp64 atomic per node instead of two per
// match (agent-scope fp64 atomics leave the XCD: 5 M of them were 0.21 ms on config 4; 0.10 ms this way).  Exact sums: the order of the additions is free.
constexpr int kScoreSlots = 128;
__global__ void k_scores_grouped(int64_t M, const uint4 *sorted, const uint32_t *order, const uint32_t *n1, const uint32_t *n2, const float *sim, const int32_t *track, double *score) {
    __shared__ uint32_t s_key[kThreads / 64][kScoreSlots];
    __shared__ double s_val[kThreads / 64][kScoreSlots];
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    for (int i = lane; i < kScoreSlots; i += 64) { s_key[w][i] = 0xffffffffu; s_val[w][i] = 0.0; }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t node[2] = {0u, 0u};
    double s = 0.0;
    bool live = false;
    if (i < M) {
        if (sorted) {                                                            // (counting road: the record is there, in order - no gathers)
            const uint4 q = sorted[i];
            node[0] = q.y; node[1] = q.z;
            s = (double)sim_from_key(~q.x);                                      // (-0.0 comes back as +0.0: the same sum)
        } else {
            const uint32_t m = order[i];
            node[0] = n1[m]; node[1] = n2[m];
            s = (double)sim[m];
        }
        live = track[node[0]] == track[node[1]];
    }
    // (a wave's LDS operations complete in order: the zeroing above is visible to its own lanes without a barrier, other waves use other rows)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!live) continue;
        const uint32_t h = (node[c] * 0x9E3779B1u) >> 25;                       // 7 bits
        const uint32_t seen = atomicCAS(&s_key[w][h], 0xffffffffu, node[c]);
        if (seen == 0xffffffffu || seen == node[c]) atomicAdd(&s_val[w][h], s);
        else atomicAdd(&score[node[c]], s);
    }
    for (int k = lane; k < kScoreSlots; k += 64) {
        const uint32_t key = s_key[w][k];
        if (key != 0xffffffffu) atomicAdd(&score[key], s_val[w][k]);
    }
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
__global__ void k_mark_roots(int64_t cap, const uint32_t *counts, const int32_t *node, uint8_t *is_root) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cap && t < (int64_t)counts[CNT_TRACKS]) is_root[node[t]] = 1;
}

// ---- components of the track meta-graph ----
// Two tracks are in one component of the meta-graph iff their nodes are in one connected component of the MATCH graph (a track is
// connected through its accepted matches; every other match joins two tracks and is a meta edge, solve.cc:262-290), and those labels exist
// since step 1.  So no second union-find over the matches (config 5: 2.3 ms, every union at the root of the one giant component): the
// component of track t is named by the smallest track of its connected component.
__global__ void k_cc_min_track(int64_t n, const uint32_t *cc, const int32_t *track, uint32_t *min_track) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const uint32_t c = valid ? cc[i] : 0xffffffffu;
    uint32_t t = valid ? (uint32_t)track[i] : 0xffffffffu;
    // the lanes that share the first lane's component send one minimum (a giant component: the whole wave), the others their own;
    // nobody sends what cannot lower the word (it only decreases, so a stale read errs on the side of sending)
    const uint32_t c0 = (uint32_t)__shfl((int)c, 0, 64);
    const bool same = c == c0;
    uint32_t ts = same ? t : 0xffffffffu;
    for (int o = 32; o > 0; o >>= 1) ts = min(ts, (uint32_t)__shfl_xor((int)ts, o, 64));
    if (same) { if ((threadIdx.x & 63) != 0) return; t = ts; }
    if (valid && t < min_track[c]) atomicMin(&min_track[c], t);
}
__global__ void k_track_label(int64_t n, const uint32_t *cc, const int32_t *track, const uint32_t *min_track, uint32_t *label) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) label[track[i]] = min_track[cc[i]];                // (every node of a track writes the same value)
}
__global__ void k_comp_flags(int64_t cap, const uint32_t *counts, const uint32_t *parent, uint32_t *flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cap && t < (int64_t)counts[CNT_TRACKS]) flags[t] = parent[t] == (uint32_t)t ? 1u : 0u;      // representative = smallest track of the component
}
// rank[] = exclusive scan of the component flags over N+1 entries (flags beyond the tracks are 0): rank[N] = #components
__global__ void k_comp_sizes(int64_t cap, uint32_t *counts, const uint32_t *parent, const uint32_t *rank, const uint32_t *tsize, uint32_t *csize) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) counts[CNT_COMPS] = rank[cap];
    const int64_t n = cap < (int64_t)counts[CNT_TRACKS] ? cap : (int64_t)counts[CNT_TRACKS];
    uint32_t v = 0;
    for (int64_t i = t; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        atomicAdd(&csize[rank[parent[i]]], tsize[i]);
        v = max(v, tsize[i]);
    }
    block_atomic_max(v, &counts[CNT_MAX_TRACK]);
}
__global__ void k_node_comp(int64_t n, const int32_t *track, const uint32_t *parent, const uint32_t *rank, int32_t *comp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = (int32_t)rank[parent[track[i]]];
}
__global__ void k_max_csize(int64_t cap, uint32_t *counts, const uint32_t *csize) {
    const int64_t n = cap < (int64_t)counts[CNT_COMPS] ? cap : (int64_t)counts[CNT_COMPS];
    uint32_t v = 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) v = max(v, csize[c]);
    block_atomic_max(v, &counts[CNT_MAX_COMP]);
}

}  // namespace

// =================================================================================================
// device copies of the graph and of the labels
// =================================================================================================
DevGraph::~DevGraph() {
    if (ctx) {      // nothing of ours may still be reading the slab when it goes back to the cache
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->s_copy);
        (void)hipStreamSynchronize(ctx->s_main);
    }
    if (!parent) for (auto &e : ev_flows) if (e) (void)hipEventDestroy(e);      // (a shard borrows its parent's events)
}
DevProblem::~DevProblem() {
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->s_main); }
}

static int stage_flows_now(const Graph &g, DevGraph &dg) {
    // 2 x 72 B per match, in match order, on the copy stream: travels beside the graph stage
    DevCtx *ctx = dg.ctx;
    const int64_t M = dg.M;
    float *d1 = dg.slab.take_n<float>((size_t)18 * M), *d2 = dg.slab.take_n<float>((size_t)18 * M);
    if (!d1 || !d2) { set_error("device graph: slab too small for the flows"); return LFR_ERR_NOMEM; }
    for (int c = 0; c < kFlowChunks; ++c) {
        const int64_t lo = M * c / kFlowChunks, hi = M * (c + 1) / kFlowChunks;
        dg.chunk_row[c] = lo; dg.chunk_row[c + 1] = hi;
        if (hi > lo) {
            LFR_HIP_TRY(hipMemcpyAsync(d1 + 18 * lo, g.m_disp1.data() + 18 * lo, (size_t)72 * (hi - lo), hipMemcpyHostToDevice, ctx->s_copy));
            LFR_HIP_TRY(hipMemcpyAsync(d2 + 18 * lo, g.m_disp2.data() + 18 * lo, (size_t)72 * (hi - lo), hipMemcpyHostToDevice, ctx->s_copy));
        }
        if (!dg.ev_flows[c]) LFR_HIP_TRY(hipEventCreateWithFlags(&dg.ev_flows[c], hipEventDisableTiming));
        LFR_HIP_TRY(hipEventRecord(dg.ev_flows[c], ctx->s_copy));
    }
    dg.disp1 = d1; dg.disp2 = d2;
    dg.flows_staged = true; dg.flows_zero_copy = false;
    return LFR_OK;
}

static int send_endpoints(const Graph &g, DevGraph &dg) {
    const int64_t N = g.n_nodes(), M = dg.M;
    hipStream_t st = dg.ctx->s_main;
    if (M > 0) {
        LFR_HIP_TRY(hipMemcpyAsync(dg.n1, g.m_node1.data(), (size_t)4 * M, hipMemcpyHostToDevice, st));
        LFR_HIP_TRY(hipMemcpyAsync(dg.n2, g.m_node2.data(), (size_t)4 * M, hipMemcpyHostToDevice, st));
        LFR_HIP_TRY(hipMemcpyAsync(dg.sim, g.m_sim.data(), (size_t)4 * M, hipMemcpyHostToDevice, st));
    }
    if (N > 0) LFR_HIP_TRY(hipMemcpyAsync(dg.node_image, g.node_image.data(), (size_t)4 * N, hipMemcpyHostToDevice, st));
    dg.N = N;
    dg.endpoints_pending = false;
    return LFR_OK;
}

void prestage_flows(const Graph &g, int device, int64_t n_bound) {
    const int64_t M = (int64_t)(g.m_disp1.size() / 18);
    if (device < 0 || M <= 0 || M >= ((int64_t)1 << 30) || n_bound >= ((int64_t)1 << 31)) return;
    if (g.dev_disp1) return;                                     // (pageable flows are fine: the copies below are staged by the runtime, on the scanner's helper thread)
    std::lock_guard<std::mutex> lk(g.dev_mu);
    if ((int)g.devgs.size() <= device) g.devgs.resize(device + 1);
    if (g.devgs[device]) return;
    DevCtx *ctx = dev_ctx(device);
    if (!ctx || hipSetDevice(device) != hipSuccess) return;
    std::shared_ptr<DevGraph> dg(new DevGraph());
    dg->ctx = ctx; dg->N = 0; dg->N_cap = n_bound; dg->M = M;
    const size_t bytes = (size_t)16 * M + (size_t)4 * n_bound + (size_t)144 * M + ((size_t)1 << 16);
    if (!dg->slab.init(ctx, bytes)) return;
    dg->n1 = dg->slab.take_n<uint32_t>(M); dg->n2 = dg->slab.take_n<uint32_t>(M);
    dg->sim = dg->slab.take_n<float>(M); dg->node_image = dg->slab.take_n<int32_t>(n_bound);
    if (stage_flows_now(g, *dg) != LFR_OK) return;
    dg->endpoints_pending = true;
    g.devgs[device] = dg;
}

int ensure_dev_graph(const Graph &g, int device, bool stage_flows, std::shared_ptr<DevGraph> &out) {
    if (device < 0) { set_error("bad device ordinal %d", device); return LFR_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g.dev_mu);
    if ((int)g.devgs.size() <= device) g.devgs.resize(device + 1);
    if (g.devgs[device] && g.devgs[device]->endpoints_pending) {          // the scanner sent the flows ahead
        DevGraph &dg = *g.devgs[device];
        if (dg.M != g.n_matches() || g.n_nodes() > dg.N_cap) g.devgs[device].reset();     // (cannot happen; rebuild rather than trust it)
        else {
            LFR_HIP_TRY(hipSetDevice(device));
            const int rc = send_endpoints(g, dg);
            if (rc != LFR_OK) { g.devgs[device].reset(); return rc; }
        }
    }
    if (g.devgs[device]) {
        if (stage_flows && g.devgs[device]->flows_zero_copy) g.devgs[device].reset();   // a whole-problem batch after a sharded one: rebuild with staged flows
        else { out = g.devgs[device]; return LFR_OK; }
    }
    if (g.dev_disp1 && g.dev_flows_device != device) {
        set_error("the graph's flows live on device %d, the pipeline was requested on device %d", g.dev_flows_device, device);
        return LFR_ERR_ARG;
    }
    DevCtx *ctx = dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    LFR_HIP_TRY(hipSetDevice(device));
    const int64_t N = g.n_nodes(), M = g.n_matches();
    if (N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 30)) { set_error("graph too large for the device pipeline"); return LFR_ERR_UNSUPPORTED; }
    std::shared_ptr<DevGraph> dg(new DevGraph());
    dg->ctx = ctx; dg->N = N; dg->N_cap = N; dg->M = M;
    const bool external = g.dev_disp1 && g.dev_disp2;
    const bool host_pinned = g.m_disp1.pinned() && g.m_disp2.pinned();
    const bool stage = !external && (stage_flows || !host_pinned);      // pageable flows cannot be read zero-copy
    const size_t bytes = (size_t)16 * M + (size_t)4 * N + (stage ? (size_t)144 * M : 0) + ((size_t)1 << 16);
    if (!dg->slab.init(ctx, bytes)) return LFR_ERR_NOMEM;
    dg->n1 = dg->slab.take_n<uint32_t>(M); dg->n2 = dg->slab.take_n<uint32_t>(M);
    dg->sim = dg->slab.take_n<float>(M); dg->node_image = dg->slab.take_n<int32_t>(N);
    hipStream_t st = ctx->s_main;
    {
        const int rc = send_endpoints(g, *dg);
        if (rc != LFR_OK) return rc;
    }
    if (external) {
        dg->disp1 = g.dev_disp1; dg->disp2 = g.dev_disp2; dg->flows_external = true;
        if (!g.m_flow_row.empty()) {
            dg->flow_row = dg->slab.take_n<uint32_t>(M);
            LFR_HIP_TRY(hipMemcpyAsync(dg->flow_row, g.m_flow_row.data(), (size_t)4 * M, hipMemcpyHostToDevice, st));
        }
    } else if (stage) {
        const int rc = stage_flows_now(g, *dg);
        if (rc != LFR_OK) return rc;
    } else {
        dg->disp1 = g.m_disp1.data(); dg->disp2 = g.m_disp2.data(); dg->flows_zero_copy = true;   // pinned + portable: device visible
    }
    g.devgs[device] = dg;
    out = dg;
    return LFR_OK;
}

int upload_labels(const Problem &p, int device, bool stage_flows, std::shared_ptr<DevProblem> &out) {
    std::shared_ptr<DevGraph> dg;
    int rc = ensure_dev_graph(*p.g, device, stage_flows, dg);
    if (rc != LFR_OK) return rc;
    DevCtx *ctx = dg->ctx;
    const int64_t N = dg->N;
    std::shared_ptr<DevProblem> dp(new DevProblem());
    dp->ctx = ctx; dp->graph = dg; dp->N = N;
    if (!dp->slab.init(ctx, (size_t)9 * N + 4096)) return LFR_ERR_NOMEM;
    dp->track = dp->slab.take_n<int32_t>(N); dp->comp = dp->slab.take_n<int32_t>(N); dp->is_root = dp->slab.take_n<uint8_t>(N);
    // Labels that already live on ANOTHER GPU (the device graph stage ran there) come over the GPU-GPU link - xGMI on an 8-GPU node -
    // instead of through the host: 9 bytes per node, no D2H, no widening to the host's 64-bit labels and back.
    std::shared_ptr<DevProblem> peer;
    {
        std::lock_guard<std::mutex> lk(p.label_mu);
        for (auto &d : p.devs) if (d && d->track && d->N == N && d->ctx->device != device) { peer = d; break; }
    }
    if (peer && N > 0) {
        const int src = peer->ctx->device;
        LFR_HIP_TRY(hipMemcpyPeerAsync(dp->track, device, peer->track, src, (size_t)4 * N, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyPeerAsync(dp->comp, device, peer->comp, src, (size_t)4 * N, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyPeerAsync(dp->is_root, device, peer->is_root, src, (size_t)N, ctx->s_main));
        LFR_HIP_TRY(hipStreamSynchronize(ctx->s_main));
        out = dp;
        return LFR_OK;
    }
    if (N > 0) {
        if (!p.host_labels_valid) { const int rc2 = p.ensure_host_labels(); if (rc2 != LFR_OK) return rc2; }
        std::vector<int32_t> t32(N), c32(N);
        for (int64_t i = 0; i < N; ++i) { t32[i] = (int32_t)p.track[i]; c32[i] = (int32_t)p.comp[i]; }
        LFR_HIP_TRY(hipMemcpyAsync(dp->track, t32.data(), (size_t)4 * N, hipMemcpyHostToDevice, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyAsync(dp->comp, c32.data(), (size_t)4 * N, hipMemcpyHostToDevice, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyAsync(dp->is_root, p.is_root.data(), (size_t)N, hipMemcpyHostToDevice, ctx->s_main));
        LFR_HIP_TRY(hipStreamSynchronize(ctx->s_main));         // the staging vectors die here
    }
    out = dp;
    return LFR_OK;
}

int Problem::ensure_host_labels() const {
    std::lock_guard<std::mutex> lk(label_mu);
    if (host_labels_valid) return LFR_OK;
    std::shared_ptr<DevProblem> dev;
    for (auto &d : devs) if (d) { dev = d; break; }
    if (!dev) { set_error("internal: labels neither on the host nor on the device"); return LFR_ERR_ARG; }
    const int64_t N = dev->N;
    DevCtx *ctx = dev->ctx;
    LFR_HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int32_t> t32(N), c32(N);
    track.resize(N); comp.resize(N); is_root.resize(N);
    if (N > 0) {
        LFR_HIP_TRY(hipMemcpyAsync(t32.data(), dev->track, (size_t)4 * N, hipMemcpyDeviceToHost, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyAsync(c32.data(), dev->comp, (size_t)4 * N, hipMemcpyDeviceToHost, ctx->s_main));
        LFR_HIP_TRY(hipMemcpyAsync(is_root.data(), dev->is_root, (size_t)N, hipMemcpyDeviceToHost, ctx->s_main));
        LFR_HIP_TRY(hipStreamSynchronize(ctx->s_main));
    }
    for (int64_t i = 0; i < N; ++i) { track[i] = t32[i]; comp[i] = c32[i]; }
    host_labels_valid = true;
    return LFR_OK;
}

// =================================================================================================
// the stage
// =================================================================================================
int warm_graphstage_primitives(DevCtx *ctx) {
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
        if ((rc = sort_pairs(arena, k64a, k64b, v32a, v32b, n, 0, 52, st)) != LFR_OK) return rc;
        if ((rc = exclusive_sum(arena, v32a, v32b, n, st)) != LFR_OK) return rc;
    }
    // the two opt-in forms of the union-find rounds (the default is one launch per round and needs no warm-up of its own): an empty list each
    const char *e_coop = getenv("LFR_ROUNDS_COOPERATIVE"), *e_xcd = getenv("LFR_ROUNDS_XCD");
    const bool coop = e_coop && e_coop[0] == '1', xcd = e_xcd && e_xcd[0] == '1' && !coop;
    if (coop || xcd) {
        uint32_t *c = arena.take_n<uint32_t>(16 + 160);
        if (c) {
            LFR_HIP_TRY(hipMemsetAsync(c, 0, 4 * (16 + 160), st));
            RoundsArgs ra{0, 1024, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, kMaxRounds, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c, c + 16, 0, 0};
            void *kargs[1] = {&ra};
            if (coop) { if (hipLaunchCooperativeKernel((const void *)k_rounds_all<false>, dim3(256), dim3(kThreads), kargs, 0, st) != hipSuccess) (void)hipGetLastError(); }
            else hipLaunchKernelGGL(k_rounds_all<true>, dim3(64), dim3(kThreads), 0, st, ra);      // (LFR_ROUNDS_XCD=1: the rounds on one XCD - measured slower, not the default)
        }
    }
    LFR_HIP_TRY(stream_wait(st));
    return rc;
}

// ---- sharding by connected component (multi-GPU: every rank's graph stage, assembly and solve over its own components only) ----
__global__ void k_cc_is_root(int64_t n, const uint32_t *cc, uint32_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) flag[i] = i < n && cc[i] == (uint32_t)i ? 1u : 0u;
}
// keep[m] = the connected component of match m is dealt to `rank`; per-rank match counts (one atomic per workgroup and rank present)
__global__ void k_shard_keep(int64_t M, const uint32_t *n1, const uint32_t *cc, const uint32_t *cc_index, int rank, int world, uint32_t *keep, unsigned long long *per_rank) {
    __shared__ unsigned int cnt[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) {
        const int owner = (int)(cc_index[cc[n1[m]]] % (uint32_t)world);
        keep[m] = owner == rank ? 1u : 0u;
        atomicAdd(&cnt[owner], 1u);
    } else if (m == M) keep[m] = 0u;
    __syncthreads();
    if (threadIdx.x < world && cnt[threadIdx.x]) atomicAdd(&per_rank[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}
__global__ void k_shard_scatter(int64_t M, const uint32_t *keep, const uint32_t *pos, const uint32_t *n1, const uint32_t *n2, const float *sim,
                                const uint32_t *flow_row, uint32_t *o_n1, uint32_t *o_n2, float *o_sim, uint32_t *o_row) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || !keep[m]) return;
    const uint32_t q = pos[m];
    o_n1[q] = n1[m]; o_n2[q] = n2[m]; o_sim[q] = sim[m];
    o_row[q] = flow_row ? flow_row[m] : (uint32_t)m;
}
// The shard of `full` for (rank, world): connected components (the same union-find kernels as the stage itself), the k-th component in
// node order to rank k mod world, a STABLE compaction of the matches (their relative order is what the batch layout of a component
// depends on: the shard's components come out bit-identical to the whole graph's).  per_rank_h[r] = matches of rank r.
static int shard_dev_graph(const std::shared_ptr<DevGraph> &full, int rank, int world, std::shared_ptr<DevGraph> &out, std::vector<unsigned long long> &per_rank_h) {
    DevCtx *ctx = full->ctx;
    hipStream_t st = ctx->s_main;
    const int64_t N = full->N, M = full->M;
    DevArena arena;
    if (!arena.init(ctx, (size_t)12 * N + (size_t)12 * M + ((size_t)8 << 20))) return LFR_ERR_NOMEM;
    uint32_t *parent = arena.take_n<uint32_t>(N), *cc = arena.take_n<uint32_t>(N), *rflag = arena.take_n<uint32_t>(N + 1), *cc_index = arena.take_n<uint32_t>(N + 1);
    uint32_t *keep = arena.take_n<uint32_t>(M + 1), *pos = arena.take_n<uint32_t>(M + 1);
    unsigned long long *per_rank = arena.take_n<unsigned long long>(64);
    if (!parent || !cc || !rflag || !cc_index || !keep || !pos || !per_rank) { set_error("shard arena exhausted"); return LFR_ERR_NOMEM; }
    LFR_HIP_TRY(hipMemsetAsync(per_rank, 0, 64 * 8, st));
    hipLaunchKernelGGL(k_iota, grid_for(N), dim3(kThreads), 0, st, N, parent);
    for (int i = 0, done = 0; i < kUnionStages; done = kUnionStrides[i], ++i) {
        if (i) hipLaunchKernelGGL(k_uf_flatten, grid_for(N), dim3(kThreads), 0, st, N, parent);
        hipLaunchKernelGGL(k_cc_union, grid_for((M + kUnionStrides[i] - 1) / kUnionStrides[i]), dim3(kThreads), 0, st, M, kUnionStrides[i], done, full->n1, full->n2, parent);
    }
    hipLaunchKernelGGL(k_cc_labels, grid_for(N), dim3(kThreads), 0, st, N, parent, cc);
    hipLaunchKernelGGL(k_cc_is_root, grid_for(N + 1), dim3(kThreads), 0, st, N, cc, rflag);
    int rc;
    if ((rc = exclusive_sum(arena, rflag, cc_index, N + 1, st)) != LFR_OK) return rc;
    hipLaunchKernelGGL(k_shard_keep, grid_for(M + 1), dim3(kThreads), 0, st, M, full->n1, cc, cc_index, rank, world, keep, per_rank);
    if ((rc = exclusive_sum(arena, keep, pos, M + 1, st)) != LFR_OK) return rc;
    per_rank_h.assign(64, 0);
    LFR_HIP_TRY(hipMemcpyAsync(per_rank_h.data(), per_rank, 64 * 8, hipMemcpyDeviceToHost, st));
    LFR_HIP_TRY(stream_wait(st));
    const int64_t Ms = (int64_t)per_rank_h[rank];
    std::shared_ptr<DevGraph> dg(new DevGraph());
    dg->ctx = ctx; dg->N = N; dg->M = Ms; dg->N_cap = N; dg->parent = full;
    if (!dg->slab.init(ctx, (size_t)16 * std::max<int64_t>(Ms, 1) + 4096)) return LFR_ERR_NOMEM;
    dg->n1 = dg->slab.take_n<uint32_t>(Ms); dg->n2 = dg->slab.take_n<uint32_t>(Ms); dg->sim = dg->slab.take_n<float>(Ms); dg->flow_row = dg->slab.take_n<uint32_t>(Ms);
    if (!dg->n1 || !dg->n2 || !dg->sim || !dg->flow_row) { set_error("shard slab exhausted"); return LFR_ERR_NOMEM; }
    dg->node_image = full->node_image;
    dg->disp1 = full->disp1; dg->disp2 = full->disp2;
    dg->flows_staged = full->flows_staged; dg->flows_zero_copy = full->flows_zero_copy; dg->flows_external = full->flows_external;
    for (int c = 0; c < 4; ++c) dg->ev_flows[c] = full->ev_flows[c];
    for (int c = 0; c < 5; ++c) dg->chunk_row[c] = full->chunk_row[c];
    hipLaunchKernelGGL(k_shard_scatter, grid_for(M), dim3(kThreads), 0, st, M, keep, pos, full->n1, full->n2, full->sim, full->flow_row,
                       dg->n1, dg->n2, dg->sim, dg->flow_row);
    LFR_HIP_TRY(hipGetLastError());
    LFR_HIP_TRY(stream_wait(st));                      // (the temporaries go back to the cache)
    out = dg;
    return LFR_OK;
}

int graph_stage_on_device(const Graph &g, int64_t max_nodes, int device, bool stage_flows, Problem &p, int shard_rank, int shard_world) {
    const int64_t N = g.n_nodes();
    int64_t M = g.n_matches();
    const char *vb = getenv("LFR_VERBOSE");
    const int trace = vb ? atoi(vb) >= 2 ? atoi(vb) : 0 : 0;      // LFR_VERBOSE=2: lap times, 3: + every round of the parallel greedy
    const auto tr0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "lfr graph stage: %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), what);
    };
    p.g = &g;
    p.stats = lfr_problem_stats{};
    p.host_batch = false;
    { std::lock_guard<std::mutex> lk(p.label_mu); p.devs.clear(); }
    if (N == 0) { p.track.clear(); p.comp.clear(); p.is_root.clear(); p.host_labels_valid = true; return LFR_OK; }
    // (every hand-back to the host graph stage says why under LFR_VERBOSE / LFR_TIMING: results are the same, the time is not - ADVICE r4)
    auto use_host = [&](const char *why) {
        if (vb || getenv("LFR_TIMING")) fprintf(stderr, "lfr graph stage: handed back to the host (%s)\n", why);
        return LFR_GRAPHSTAGE_USE_HOST;
    };
    if (N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 30)) return use_host("2^31 nodes / 2^30 matches or more");
    if (!g.sims_sum_exactly) return use_host("similarity sums would round");      // atomics would round in an order-dependent way: the reference's sequential sums run on the host
    if (max_nodes <= 0) max_nodes = (int64_t)g.image_names.size();
    std::shared_ptr<DevGraph> dg;
    int rc = ensure_dev_graph(g, device, stage_flows, dg);       // endpoints on s_main now; the flows start travelling on s_copy
    if (rc != LFR_OK) return rc;
    DevCtx *ctx = dg->ctx;
    LFR_HIP_TRY(hipSetDevice(device));
    hipStream_t st = ctx->s_main;
    lap("device graph ready");
    p.cc_sharded = false;
    if (shard_world > 1 && shard_world <= 64 && M > 0) {
        std::shared_ptr<DevGraph> dgs;
        std::vector<unsigned long long> per_rank;
        if ((rc = shard_dev_graph(dg, shard_rank, shard_world, dgs, per_rank)) != LFR_OK) return rc;
        unsigned long long mx = 0;
        for (int r = 0; r < shard_world; ++r) mx = std::max(mx, per_rank[r]);
        // balanced enough (every rank decides the same from the same counts): the largest rank within 5/4 of the mean
        if (4ull * mx * (unsigned long long)shard_world <= 5ull * (unsigned long long)M + 4096ull * shard_world) {
            dg = dgs; M = dg->M;
            p.cc_sharded = true; p.shard_matches = M; p.shard_matches_max = (int64_t)mx;
        }
        lap(p.cc_sharded ? "sharded by connected component" : "one connected component dominates: the whole graph on every rank");
    }

    std::shared_ptr<DevProblem> dp(new DevProblem());
    dp->ctx = ctx; dp->graph = dg; dp->N = N;
    if (!dp->slab.init(ctx, (size_t)9 * N + 4096)) return LFR_ERR_NOMEM;
    dp->track = dp->slab.take_n<int32_t>(N); dp->comp = dp->slab.take_n<int32_t>(N); dp->is_root = dp->slab.take_n<uint8_t>(N);

    DevArena arena;                                   // temporaries of this call
    if (!arena.init(ctx, (size_t)112 * M + (size_t)112 * N + ((size_t)32 << 20))) return LFR_ERR_NOMEM;
    size_t pin_bytes = 0;
    uint32_t *h_counts = (uint32_t *)ctx->pinned_acquire(4 * (CNT_WORDS + 16), &pin_bytes);     // + the counters of a batch of union-find rounds
    if (!h_counts) return LFR_ERR_NOMEM;
    struct PinGuard { DevCtx *c; void *p; size_t b; ~PinGuard() { c->pinned_release(p, b); } } pin_guard{ctx, h_counts, pin_bytes};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 4; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ev};
    for (auto &e : ev) LFR_HIP_TRY(hipEventCreate(&e));
    lap("slabs, pinned block, events");

    const uint32_t *n1 = dg->n1, *n2 = dg->n2;
    const float *sim = dg->sim;
    int node_bits = 1;
    while (((int64_t)1 << node_bits) < N) ++node_bits;
    hipEvent_t ev_begin = nullptr;
    if (trace) { LFR_HIP_TRY(hipEventCreate(&ev_begin)); LFR_HIP_TRY(hipEventRecord(ev_begin, st)); }
    // everything that starts at zero sits in one block: one memset instead of a dozen
    const size_t zero_mark = arena.top;
    TAKE(counts, uint32_t, CNT_WORDS);
    TAKE(flags, uint32_t, M + 1); TAKE(rflag, uint32_t, N + 1); TAKE(tsize, uint32_t, N);
    TAKE(score, double, N); TAKE(best, unsigned long long, N); TAKE(cflag, uint32_t, N + 1); TAKE(csize, uint32_t, N);
    // look-back states of the stage's prefix sums (exclusive_sum_one_launch: no init launch for states that start in the zero block)
    const size_t scan_words_m = scan_state_words(M + 1), scan_words_n = scan_state_words(N + 1);
    TAKE(scan_state, unsigned long long, scan_words_m + 2 * scan_words_n);
    const size_t zero_bytes = arena.top - zero_mark;
    TAKE(bnode, int32_t, N); TAKE(mp_parent, uint32_t, N);                // (next to each other: one region of -1 / of "no track yet")
    {   // the stage's initial values in one launch (the memsets were 5 launches before the first kernel, 2 more before the meta-components)
        FillRegions fr;
        fr.add(arena.base + zero_mark, zero_bytes, 0);
        fr.add(dp->is_root, (size_t)N, 0);
        fr.add(bnode, (size_t)(reinterpret_cast<char *>(mp_parent + N) - reinterpret_cast<char *>(bnode)), 0xff);
        LFR_HIP_TRY(fill_regions(fr, st));
    }
    LFR_HIP_TRY(hipEventRecord(ev[0], st));

    // 1. connected components of the match graph (conflicts ignored)
    TAKE(cc_parent, uint32_t, N); TAKE(cc, uint32_t, N);
    hipLaunchKernelGGL(k_iota, grid_for(N), dim3(kThreads), 0, st, N, cc_parent);
    {
        for (int i = 0, done = 0; i < kUnionStages; done = kUnionStrides[i], ++i) {
            if (i) hipLaunchKernelGGL(k_uf_flatten, grid_for(N), dim3(kThreads), 0, st, N, cc_parent);
            hipLaunchKernelGGL(k_cc_union, grid_for((M + kUnionStrides[i] - 1) / kUnionStrides[i]), dim3(kThreads), 0, st, M, kUnionStrides[i], done, n1, n2, cc_parent);
        }
    }
    hipLaunchKernelGGL(k_cc_labels, grid_for(N), dim3(kThreads), 0, st, N, cc_parent, cc);

    // 2. matches grouped by connected component, inside a component in the reference's order: descending (sim, n1, n2).
    //    One sort over (component | ~sim) + the tie fix; the exact three-sort order only if a run of equal similarities is too long for it
    TAKE(rec, uint4, M);                              // (the 64-bit keys of the one-sort road share these 16 M bytes with the records of the counting road)
    uint64_t *const khi = reinterpret_cast<uint64_t *>(rec), *const khi2 = khi + M;
    TAKE(klo, uint32_t, M); TAKE(klo2, uint32_t, M);
    TAKE(srec, uint4, M);                             // counting road: the records in the reference's order
    TAKE(id0, uint32_t, M); TAKE(id1, uint32_t, M);
    TAKE(ck0, uint32_t, M); TAKE(ck1, uint32_t, M); TAKE(segid, uint32_t, M + 1);
    TAKE(starts, uint32_t, std::min(N, M) + 2);
    uint32_t *order = id1;
    int max_tie_run = kMaxTieRun;
    if (const char *e = getenv("LFR_MAX_TIE_RUN")) max_tie_run = std::max(0, atoi(e));     // (tests: 0 = always the three sorts)
    auto three_sorts = [&]() -> int {
        int r;
        hipLaunchKernelGGL(k_match_keys, grid_for(M), dim3(kThreads), 0, st, M, (uint32_t)(N - 1), node_bits, n1, n2, sim, khi, klo, id0);
        if ((r = sort_pairs(arena, klo, klo2, id0, id1, M, 0, node_bits, st)) != LFR_OK) return r;
        hipLaunchKernelGGL(k_gather_u64, grid_for(M), dim3(kThreads), 0, st, M, id1, khi, khi2);
        if ((r = sort_pairs(arena, khi2, khi, id1, id0, M, 0, 32 + node_bits, st)) != LFR_OK) return r;
        hipLaunchKernelGGL(k_cc_keys, grid_for(M), dim3(kThreads), 0, st, M, id0, n1, cc, ck0);
        return sort_pairs(arena, ck0, ck1, id0, id1, M, 0, node_bits, st);                // stable: -> order
    };
    const int64_t seg_cap = std::min(N, M) + 1;       // a segment has >= 1 match and >= 2 nodes
    bool scan_state_used = false;
    auto segments = [&]() -> int {                    // flags -> segment ids, starts, longest segment
        // (the one long prefix sum of the stage: rocPRIM's larger tiles keep the look-back chain short - 20.6 us with its init launch
        // against 26.9 us for the one-launch kernel at 2.5 M flags; the short sums below are the other way round.  The one-launch sum's
        // states are good for one use: a second pass over the segments takes the library's.)
        if (M + 1 > (int64_t)1 << 20 || scan_state_used) { const int r = exclusive_sum(arena, flags, segid, M + 1, st); if (r != LFR_OK) return r; }
        else { LFR_HIP_TRY(exclusive_sum_one_launch(flags, segid, M + 1, scan_state, st)); scan_state_used = true; }
        hipLaunchKernelGGL(k_seg_starts, grid_for(std::max<int64_t>(M, 1)), dim3(kThreads), 0, st, M, flags, segid, starts, counts);
        hipLaunchKernelGGL(k_seg_maxlen, grid_few(seg_cap), dim3(kThreads), 0, st, seg_cap, starts, counts);
        return LFR_OK;
    };
    // The counting road (k_rank_sort) first: group by component, look at the longest segment (the stage's first read-back, which used to
    // follow the small-component union-find), count inside the segments.  A giant component (or LFR_ONE_SORT_ORDER=1, or the tests'
    // LFR_MAX_TIE_RUN=0) takes the roads of rounds 3-5: one sort over (component | ~similarity) + tie repair, or the three sorts.
    bool counts_on_host = false, order_done = false;
    if (max_tie_run > 0 && !getenv("LFR_ONE_SORT_ORDER")) {
        hipLaunchKernelGGL(k_match_records, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, sim, cc, rec, ck0, id0);
        if ((rc = sort_pairs(arena, ck0, ck1, id0, id1, M, 0, node_bits, st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_seg_flags, grid_for(M), dim3(kThreads), 0, st, M, ck1, flags);
        if ((rc = segments()) != LFR_OK) return rc;
        lap("grouped by connected component");
        LFR_HIP_TRY(hipMemcpyAsync(h_counts, counts, 4 * CNT_WORDS, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(stream_wait(st));
        lap("first read-back");
        if ((int64_t)h_counts[CNT_MAX_SEG] <= kRankSortMax) {
            const size_t window = sizeof(uint4) * ((size_t)kThreads + 2 * (size_t)h_counts[CNT_MAX_SEG]);     // (8 KB for config 4's 136: eight workgroups per CU)
            hipLaunchKernelGGL(k_rank_sort, grid_for(M), dim3(kThreads), window, st, M, id1, rec, flags, segid, starts, id0, srec);
            order = id0;
            counts_on_host = true; order_done = true;
        }
    }
    if (!order_done) {
        if (max_tie_run > 0) {
            hipLaunchKernelGGL(k_cc_sim_keys, grid_for(M), dim3(kThreads), 0, st, M, n1, sim, cc, khi, id0);
            if ((rc = sort_pairs(arena, khi, khi2, id0, id1, M, 0, 32 + node_bits, st)) != LFR_OK) return rc;
            hipLaunchKernelGGL(k_tie_fix, grid_for(M), dim3(kThreads), 0, st, M, khi2, order, n1, n2, max_tie_run, counts);
            hipLaunchKernelGGL(k_seg_flags_hi, grid_for(M), dim3(kThreads), 0, st, M, khi2, flags);
        } else {
            if ((rc = three_sorts()) != LFR_OK) return rc;
            hipLaunchKernelGGL(k_seg_flags, grid_for(M), dim3(kThreads), 0, st, M, ck1, flags);
        }
        if ((rc = segments()) != LFR_OK) return rc;
    }

    // 3. greedy constrained union-find per connected component: small ones one thread each ...
    TAKE(par, int32_t, N); TAKE(next, int32_t, N); TAKE(tail, int32_t, N); TAKE(cnt, int32_t, N); TAKE(sig, ulonglong2, N);
    hipLaunchKernelGGL(k_init_nodes, grid_for(N), dim3(kThreads), 0, st, N, dg->node_image, par, next, tail, cnt, sig);
    int64_t serial_limit = kSerialSegmentEdges;
    if (const char *e = getenv("LFR_SERIAL_SEGMENT_EDGES")) serial_limit = std::max<int64_t>(0, atoll(e));
    // (counting road: the matches as a stream of sorted records - k_kruskal_stream; LFR_KRUSKAL_GLOBAL=1: the gathering kernel, A/B and tests)
    const bool streamed = counts_on_host && order_done && !getenv("LFR_KRUSKAL_GLOBAL");
    if (streamed)
        hipLaunchKernelGGL(k_kruskal_stream, grid_for(seg_cap), dim3(kThreads), 0, st, seg_cap, serial_limit, counts, starts, srec, dg->node_image, par, next, tail, cnt, sig);
    else
        hipLaunchKernelGGL(k_kruskal, grid_for(seg_cap), dim3(kThreads), 0, st, seg_cap, serial_limit, counts, starts, order, n1, n2,
                           dg->node_image, par, next, tail, cnt, sig);
    // ... large ones in parallel rounds (first read-back: is there any? - the counting road has read the counts already)
    lap("sorts / connected components / small-component union-find enqueued");
    if (!counts_on_host) {
        LFR_HIP_TRY(hipMemcpyAsync(h_counts, counts, 4 * CNT_WORDS, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(stream_wait(st));
        lap("first read-back");
    }
    if (h_counts[CNT_LONG_TIE]) {
        // a long run of equal similarities: the exact order from the three sorts (the segments - which matches, where - are the same:
        // flags, segment ids and starts stay), the union-find of the small components again from scratch
        if ((rc = three_sorts()) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_init_nodes, grid_for(N), dim3(kThreads), 0, st, N, dg->node_image, par, next, tail, cnt, sig);
        hipLaunchKernelGGL(k_kruskal, grid_for(seg_cap), dim3(kThreads), 0, st, seg_cap, serial_limit, counts, starts, order, n1, n2,
                           dg->node_image, par, next, tail, cnt, sig);
        p.stats.tie_resorts = 1.0;
        lap("three-sort order after a long tie");
    }
    if (trace && ev_begin) {
        float a = 0.f;
        (void)hipEventElapsedTime(&a, ev_begin, ev[0]);
        fprintf(stderr, "lfr graph stage: device time of the memsets before the first kernel: %.3f ms (zero block %zu bytes)\n", a, zero_bytes);
        (void)hipEventDestroy(ev_begin);
    }
    DevArena rounds_arena;
    if ((int64_t)h_counts[CNT_MAX_SEG] > serial_limit) {
        const int W = (int)((g.image_names.size() + 63) / 64);
        const size_t bits_bytes = (size_t)N * W * 8;
        if (bits_bytes > kMaxBitsetBytes) return use_host("image bitsets above the cap");
        if (!rounds_arena.init(ctx, bits_bytes + 2 * (size_t)M * sizeof(Pending) + 8 * (size_t)N + 65536)) return LFR_ERR_NOMEM;
        unsigned long long *bits = rounds_arena.take_n<unsigned long long>((size_t)N * W);
        unsigned long long *minpos = rounds_arena.take_n<unsigned long long>(N);
        Pending *pa = rounds_arena.take_n<Pending>(M), *pb = rounds_arena.take_n<Pending>(M);
        uint32_t *ctr = rounds_arena.take_n<uint32_t>(64);
        if (!bits || !minpos || !pa || !pb || !ctr) { set_error("graph stage: rounds arena exhausted"); return LFR_ERR_NOMEM; }
        LFR_HIP_TRY(hipMemsetAsync(bits, 0, bits_bytes, st));
        LFR_HIP_TRY(hipMemsetAsync(minpos, 0xff, 8 * (size_t)N, st));
        LFR_HIP_TRY(hipMemsetAsync(ctr, 0, 4 * 64, st));
        uint32_t *h_ctr = h_counts + CNT_WORDS;        // (16 more words behind the counts proper)
        // The ordered list goes through the rounds in PREFIX BLOCKS (2 N positions, then doubling): the sequential rule finishes
        // a prefix before it looks at the rest, so rounds over a block alone are exact, and a match is only carried through the
        // rounds of its own block.  With the whole list pending at once every match was re-evaluated each round until its
        // endpoints had met - 1.2e8 evaluations for config 5's 5.3 M matches (58 rounds); the tracks assemble from the first
        // few hundred thousand matches and what comes later retires in its block's first round: a twelfth of the work in a
        // simulation of this schedule, a few rounds more.
        int64_t first_block = std::max<int64_t>(2 * N, 1024);
        if (const char *e = getenv("LFR_ROUNDS_FIRST_BLOCK")) first_block = std::max<int64_t>(1, atoll(e));
        int64_t rounds = 0;
        // LFR_ROUNDS_COOPERATIVE=1: one cooperative launch runs every prefix block and every round (k_rounds_all, grid barriers instead
        // of launches).  Measured on config 5 (76 rounds, profiles/r03_rounds_cooperative.txt): 4.4 ms with one workgroup per CU, 5.8 / 8.0
        // with two / four, against 4.3 ms for the launch-per-round loop below - a barrier needs the same agent-scope write-back and
        // invalidate of the eight L2s as a kernel boundary and costs ~28 us; with cooperative_groups' grid.sync() ~130 us (20 ms).  So
        // the loop stays the default and this path is kept as the measured alternative.
        bool done = false;
        // Round 5, opt-in (LFR_ROUNDS_XCD=1), measured and NOT the default: the first round of a prefix block - every match of the block -
        // on the whole chip, the block's remaining rounds - a few thousand pending matches each, dozens of rounds - in ONE launch on ONE
        // XCD (k_rounds_all<true>): three L2-local barriers per round instead of two kernel boundaries.  Config 5's graph stage: 14.4-15.3
        // ms against 10.9-12.6 for the launch-per-round loop (and 16.4-17.8 with every round on one XCD, the big first rounds included):
        // 128 workgroups meeting at one L2 word three times per round, and an eighth of the chip's issue slots for the evaluation, cost
        // more than the kernel boundaries they replace.
        bool xcd_tail = false;
        { const char *hx = getenv("LFR_ROUNDS_XCD"); if (hx && hx[0] == '1' && !(getenv("LFR_ROUNDS_COOPERATIVE") && getenv("LFR_ROUNDS_COOPERATIVE")[0] == '1')) xcd_tail = true; }
        // Round 6: the one-XCD loop for SHORT lists only.  Below a few thousand pending matches a round is two kernel
        // boundaries around a chain of ~10 dependent accesses (~26 us whatever the count, 40 of config 5's 76 rounds), and a batch of
        // eight rounds is launched whole even when the list empties in its second; a handful of workgroups on one XCD then runs the
        // rest of the block's rounds in ONE launch (L2-local barriers, no read-back in between).  LFR_ROUNDS_TAIL="pending,workgroups"
        // (0 = off): the list length below which the block is finished that way, and how many workgroups take part.
        const std::pair<uint32_t, int> tail_cfg = [] {
            unsigned t = LFR_ROUNDS_TAIL_PENDING, w = LFR_ROUNDS_TAIL_WGS;
            if (const char *e = getenv("LFR_ROUNDS_TAIL")) { unsigned a = 0, b = 0; const int got = sscanf(e, "%u,%u", &a, &b); if (got >= 1) t = a; if (got >= 2 && b >= 1 && b <= 32) w = b; }
            return std::make_pair((uint32_t)t, (int)w);
        }();
        uint32_t *xbar = rounds_arena.take_n<uint32_t>(160);
        if (!xbar) { set_error("graph stage: rounds arena exhausted"); return LFR_ERR_NOMEM; }
        const char *hl = getenv("LFR_ROUNDS_COOPERATIVE");
        if (hl && hl[0] == '1') {
            static int blocks_per_cu = -1, n_cu = 0;
            if (blocks_per_cu < 0) {
                int nb = 0;
                hipDeviceProp_t prop;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_rounds_all<false>, kThreads, 0) == hipSuccess && nb > 0 &&
                    hipGetDeviceProperties(&prop, device) == hipSuccess && prop.cooperativeLaunch) { const char *eb = getenv("LFR_ROUNDS_BLOCKS_PER_CU"); blocks_per_cu = std::max(1, std::min(nb, eb ? atoi(eb) : 1)); n_cu = prop.multiProcessorCount; }
                else blocks_per_cu = 0;
            }
            if (blocks_per_cu > 0) {
                uint32_t *bar = rounds_arena.take_n<uint32_t>(160);
                if (!bar) { set_error("graph stage: rounds arena exhausted"); return LFR_ERR_NOMEM; }
                LFR_HIP_TRY(hipMemsetAsync(bar, 0, 4 * 160, st));
                RoundsArgs ra{M, first_block, serial_limit, flags, segid, starts, order, n1, n2, dg->node_image, W, kMaxRounds, bits, minpos, pa, pb, par, cnt, ctr + 8, bar, 0, 0};
                void *kargs[1] = {&ra};
                const hipError_t e = hipLaunchCooperativeKernel((const void *)k_rounds_all<false>, dim3((unsigned)(blocks_per_cu * n_cu)), dim3(kThreads), kargs, 0, st);
                if (e == hipSuccess) {
                    LFR_HIP_TRY(hipMemcpyAsync(h_ctr, ctr + 8, 4 * 5, hipMemcpyDeviceToHost, st));
                    LFR_HIP_TRY(stream_wait(st));
                    if (h_ctr[4]) return use_host("a path-shaped dependency chain");                      // a path-shaped dependency chain: sequential anyway
                    rounds = h_ctr[2];
                    if (trace > 2) fprintf(stderr, "lfr graph stage:   %u rounds in one cooperative launch, %u matches accepted\n", h_ctr[2], h_ctr[3]);
                    done = true;
                } else {
                    (void)hipGetLastError();
                    blocks_per_cu = 0;
                }
            }
        }
        // Rounds run in batches of kRoundBatch without a host round trip: every round takes its input count from the device
        // (the launches are sized by the count of the last read-back: counts only shrink), a round with nothing pending is two
        // empty launches.  Round 2 read the count back after every round: 76 synchronisations for config 5's giant component.
        constexpr int kRoundBatch = 8;
        uint32_t *rc_ = rounds_arena.take_n<uint32_t>(kRoundBatch + 2);
        if (!rc_) { set_error("graph stage: rounds arena exhausted"); return LFR_ERR_NOMEM; }
        int64_t launched = 0;
        if (!done) LFR_HIP_TRY(hipMemsetAsync(rc_, 0, 4 * (kRoundBatch + 2), st));
        for (int64_t k_lo = 0, size = first_block; !done && k_lo < M; k_lo += size, size *= 2) {
            const int64_t k_hi = std::min(M, k_lo + size);
            LFR_HIP_TRY(hipMemsetAsync(rc_ + kRoundBatch, 0, 4, st));
            hipLaunchKernelGGL(k_large_pending, grid_for_items(k_hi - k_lo, kAppendItems), dim3(kThreads), 0, st, k_lo, k_hi, serial_limit, flags, segid, starts, order, n1, n2,
                               dg->node_image, W, bits, pa, rc_ + kRoundBatch);
            LFR_HIP_TRY(hipMemcpyAsync(h_ctr, rc_ + kRoundBatch, 4, hipMemcpyDeviceToHost, st));
            LFR_HIP_TRY(stream_wait(st));
            uint32_t bound = h_ctr[0];
            if (xcd_tail && bound > 0) {
                if (launched + 1 > kMaxRounds) return use_host("round limit");
                hipLaunchKernelGGL(k_round_counters_shift, dim3(1), dim3(64), 0, st, rc_, kRoundBatch);         // rc_[0] = pending, rc_[1..] = 0
                const unsigned long long round_hi = (unsigned long long)(kMaxRounds - launched) << 32;
                hipLaunchKernelGGL(k_round_eval, grid_for_items(bound, kEvalItems), dim3(kThreads), 0, st, rc_, pa, par, bits, W, round_hi, minpos, pb, rc_ + 1);
                hipLaunchKernelGGL(k_round_accept, grid_for(bound), dim3(kThreads), 0, st, rc_ + 1, pb, round_hi, minpos, par, cnt, bits, W, ctr + 3);
                std::swap(pa, pb);
                ++launched;
                // the survivors (pa, rc_[1] entries) finish their rounds on one XCD
                LFR_HIP_TRY(hipMemsetAsync(xbar, 0, 4 * 160, st));
                LFR_HIP_TRY(hipMemsetAsync(ctr + 8, 0, 4 * 8, st));
                LFR_HIP_TRY(hipMemcpyAsync(ctr + 8, rc_ + 1, 4, hipMemcpyDeviceToDevice, st));
                RoundsArgs ra{M, first_block, serial_limit, flags, segid, starts, order, n1, n2, dg->node_image, W, kMaxRounds, bits, minpos, pa, pb, par, cnt, ctr + 8, xbar, 1, (int)launched};
                // (every participant must be resident at once - they meet at barriers: never more workgroups per CU than half of what the
                // occupancy query admits; a larger request hung the launch until its timeout)
                static int occ = -1;
                if (occ < 0) { int nb = 0; occ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_rounds_all<true>, kThreads, 0) == hipSuccess ? nb : 2; }
                int per_cu = std::max(1, std::min(4, occ / 2));
                if (const char *eb = getenv("LFR_ROUNDS_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(eb)));
                hipLaunchKernelGGL(k_rounds_all<true>, dim3((unsigned)(per_cu * ctx->n_cu)), dim3(kThreads), 0, st, ra);
                LFR_HIP_TRY(hipGetLastError());
                LFR_HIP_TRY(hipMemcpyAsync(h_ctr, ctr + 8, 4 * 5, hipMemcpyDeviceToHost, st));
                LFR_HIP_TRY(hipMemsetAsync(rc_, 0, 4 * (kRoundBatch + 2), st));
                LFR_HIP_TRY(stream_wait(st));
                if (h_ctr[4]) return use_host("a path-shaped dependency chain");                          // a path-shaped dependency chain: sequential anyway
                if (((int64_t)h_ctr[2] - launched) & 1) std::swap(pa, pb);              // (the kernel swapped its lists once per round)
                rounds += 1 + ((int64_t)h_ctr[2] - launched);
                launched = h_ctr[2];
                if (trace > 2) fprintf(stderr, "lfr graph stage:   block [%lld, %lld): %lld rounds so far\n", (long long)k_lo, (long long)k_hi, (long long)rounds);
                bound = 0;
            }
            while (bound > 0) {
                if (tail_cfg.first && bound <= tail_cfg.first && !xcd_tail) {
                    // the rest of this block's rounds on one XCD: pa holds `bound` entries (rc_[0] after the shift below), launched rounds have run
                    hipLaunchKernelGGL(k_round_counters_shift, dim3(1), dim3(64), 0, st, rc_, kRoundBatch);
                    LFR_HIP_TRY(hipMemsetAsync(xbar, 0, 4 * 160, st));
                    LFR_HIP_TRY(hipMemsetAsync(ctr + 8, 0, 4 * 8, st));
                    LFR_HIP_TRY(hipMemcpyAsync(ctr + 8, rc_, 4, hipMemcpyDeviceToDevice, st));
                    RoundsArgs ra{M, first_block, serial_limit, flags, segid, starts, order, n1, n2, dg->node_image, W, kMaxRounds, bits, minpos, pa, pb, par, cnt, ctr + 8, xbar, 1, (int)launched};
                    hipLaunchKernelGGL(k_rounds_all<true>, dim3((unsigned)(8 * tail_cfg.second)), dim3(kThreads), 0, st, ra);
                    LFR_HIP_TRY(hipGetLastError());
                    LFR_HIP_TRY(hipMemcpyAsync(h_ctr, ctr + 8, 4 * 5, hipMemcpyDeviceToHost, st));
                    LFR_HIP_TRY(hipMemsetAsync(rc_, 0, 4 * (kRoundBatch + 2), st));
                    LFR_HIP_TRY(stream_wait(st));
                    if (h_ctr[4]) return use_host("a path-shaped dependency chain");
                    if (((int64_t)h_ctr[2] - launched) & 1) std::swap(pa, pb);              // (the kernel swapped its lists once per round)
                    rounds += (int64_t)h_ctr[2] - launched;
                    launched = h_ctr[2];
                    if (trace > 2) fprintf(stderr, "lfr graph stage:   block [%lld, %lld): pending %u finished on one XCD, %lld rounds so far\n", (long long)k_lo, (long long)k_hi, bound, (long long)rounds);
                    bound = 0;
                    break;
                }
                if (launched + kRoundBatch > kMaxRounds) return use_host("round limit");   // a path-shaped dependency chain: sequential anyway
                hipLaunchKernelGGL(k_round_counters_shift, dim3(1), dim3(64), 0, st, rc_, kRoundBatch);
                for (int j = 0; j < kRoundBatch; ++j, ++launched) {
                    const unsigned long long round_hi = (unsigned long long)(kMaxRounds - launched) << 32;
                    hipLaunchKernelGGL(k_round_eval, grid_for_items(bound, kEvalItems), dim3(kThreads), 0, st, rc_ + j, pa, par, bits, W, round_hi, minpos, pb, rc_ + j + 1);
                    hipLaunchKernelGGL(k_round_accept, grid_for(bound), dim3(kThreads), 0, st, rc_ + j + 1, pb, round_hi, minpos, par, cnt, bits, W, ctr + 3);
                    std::swap(pa, pb);
                }
                LFR_HIP_TRY(hipMemcpyAsync(h_ctr, rc_, 4 * (kRoundBatch + 1), hipMemcpyDeviceToHost, st));
                LFR_HIP_TRY(stream_wait(st));
                for (int j = 0; j < kRoundBatch; ++j) rounds += h_ctr[j] > 0 ? 1 : 0;
                if (trace > 2) fprintf(stderr, "lfr graph stage:   block [%lld, %lld) rounds ..%lld: pending %u -> %u\n", (long long)k_lo, (long long)k_hi, (long long)rounds, h_ctr[0], h_ctr[kRoundBatch]);
                bound = h_ctr[kRoundBatch];
            }
        }
        p.stats.kruskal_rounds = (double)rounds;
    }

    // 5. track ids (roots in ascending node index), sizes
    TAKE(rrank, uint32_t, N + 1);
    hipLaunchKernelGGL(k_root_flags, grid_for(N), dim3(kThreads), 0, st, N, par, rflag);
    LFR_HIP_TRY(exclusive_sum_one_launch(rflag, rrank, N + 1, scan_state + scan_words_m, st));
    hipLaunchKernelGGL(k_track_ids, grid_for(N), dim3(kThreads), 0, st, N, par, rrank, dp->track, tsize, counts);
    LFR_HIP_TRY(hipEventRecord(ev[1], st));

    // roots
    // (small connected components only: in the ordered list of a giant one a wave's matches share no nodes - config 5: 0.40 ms grouped, 0.12 plain)
    if ((int64_t)h_counts[CNT_MAX_SEG] > serial_limit) hipLaunchKernelGGL(k_scores, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, sim, dp->track, score);
    else hipLaunchKernelGGL(k_scores_grouped, grid_for(M), dim3(kThreads), 0, st, M, streamed ? srec : nullptr, order, n1, n2, sim, dp->track, score);
    hipLaunchKernelGGL(k_best_score, grid_for(N), dim3(kThreads), 0, st, N, dp->track, score, best);
    hipLaunchKernelGGL(k_best_node, grid_for(N), dim3(kThreads), 0, st, N, dp->track, score, best, bnode);
    hipLaunchKernelGGL(k_mark_roots, grid_for(N), dim3(kThreads), 0, st, N, counts, bnode, dp->is_root);
    LFR_HIP_TRY(hipEventRecord(ev[2], st));

    // components of the track meta-graph, numbered by their smallest track (solve.cc:292-300)
    TAKE(mp, uint32_t, N); TAKE(crank, uint32_t, N + 1);                                 // (mp_parent - here: smallest track per connected component - starts at ~0: the stage's first fill)
    hipLaunchKernelGGL(k_cc_min_track, grid_for(N), dim3(kThreads), 0, st, N, cc, dp->track, mp_parent);
    hipLaunchKernelGGL(k_track_label, grid_for(N), dim3(kThreads), 0, st, N, cc, dp->track, mp_parent, mp);
    hipLaunchKernelGGL(k_comp_flags, grid_for(N), dim3(kThreads), 0, st, N, counts, mp, cflag);
    LFR_HIP_TRY(exclusive_sum_one_launch(cflag, crank, N + 1, scan_state + scan_words_m + scan_words_n, st));
    hipLaunchKernelGGL(k_comp_sizes, grid_few(N), dim3(kThreads), 0, st, N, counts, mp, crank, tsize, csize);
    hipLaunchKernelGGL(k_max_csize, grid_few(N), dim3(kThreads), 0, st, N, counts, csize);
    hipLaunchKernelGGL(k_node_comp, grid_for(N), dim3(kThreads), 0, st, N, dp->track, mp, crank, dp->comp);
    LFR_HIP_TRY(hipGetLastError());
    LFR_HIP_TRY(hipEventRecord(ev[3], st));

    // the one read-back of the stage
    lap("tracks / roots / components enqueued");
    LFR_HIP_TRY(hipMemcpyAsync(h_counts, counts, 4 * CNT_WORDS, hipMemcpyDeviceToHost, st));
    LFR_HIP_TRY(stream_wait(st));
    lap("final read-back");
    const bool needs_cut = (int64_t)h_counts[CNT_MAX_COMP] > max_nodes;      // solve.cc:311-343
    if (const char *dd = getenv("LFR_DEBUG_DUMP")) {      // intermediates of the stage as raw files (debugging aid)
        auto dump = [&](const char *name, const void *dptr, size_t bytes) {
            std::vector<char> h(bytes);
            (void)hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost);
            const std::string path = std::string(dd) + "/" + name + ".bin";
            if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
        };
        dump("order", order, 4 * (size_t)M); dump("cc", cc, 4 * (size_t)N); dump("starts", starts, 4 * (size_t)(std::min(N, M) + 2));
        dump("par", par, 4 * (size_t)N); dump("track", dp->track, 4 * (size_t)N); dump("counts", counts, 64);
        dump("segid", segid, 4 * (size_t)(M + 1)); dump("cnt", cnt, 4 * (size_t)N);
    }
    float ms = 0.f;
    LFR_HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); p.stats.tracks_ms = ms;
    LFR_HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); p.stats.roots_ms = ms;
    LFR_HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); p.stats.graph_cut_ms = ms;
    p.stats.n_tracks = h_counts[CNT_TRACKS]; p.stats.max_track_size = h_counts[CNT_MAX_TRACK];
    p.stats.n_components = h_counts[CNT_COMPS]; p.stats.max_component_size = h_counts[CNT_MAX_COMP];
    dp->max_cc_matches = h_counts[CNT_MAX_SEG];
    {
        std::lock_guard<std::mutex> lk(p.label_mu);
        p.track.clear(); p.comp.clear(); p.is_root.clear();
        p.host_labels_valid = false;                  // fetched from HBM on demand
        p.devs.assign((size_t)device + 1, nullptr);
        p.devs[device] = dp;
    }
    if (needs_cut) {
        // The size cap.  The deterministic bisection grows a region with a priority queue - sequential, and the meta
        // graph is tiny next to the match graph - so it runs on the host (lfr_graph.cpp: recursive_cut), but only on what
        // it needs: the device compacts the inter-track matches of the oversized components (a few per cent of a per cent
        // of the matches), the host cuts, the subset of every affected track goes back, and the device re-labels.
        hipEvent_t c0 = nullptr, c1 = nullptr;
        struct EvPair { hipEvent_t &a, &b; ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evp{c0, c1};
        LFR_HIP_TRY(hipEventCreate(&c0)); LFR_HIP_TRY(hipEventCreate(&c1));
        LFR_HIP_TRY(hipEventRecord(c0, st));
        const int64_t T = h_counts[CNT_TRACKS];
        TAKE(tcomp, int32_t, T + 1); TAKE(cut_n, uint32_t, 16); TAKE(gc_dev, int32_t, T + 1);
        DevArena cut_arena;                               // (pair, similarity) of every inter-track match of an oversized component
        if (!cut_arena.init(ctx, 16 * (size_t)M + 65536)) return LFR_ERR_NOMEM;
        unsigned long long *pair_key = cut_arena.take_n<unsigned long long>(M);
        double *pair_sim = cut_arena.take_n<double>(M);
        if (!pair_key || !pair_sim) { set_error("graph stage: cut arena exhausted"); return LFR_ERR_NOMEM; }
        LFR_HIP_TRY(hipMemsetAsync(cut_n, 0, 64, st));
        hipLaunchKernelGGL(k_track_comp, grid_for(T), dim3(kThreads), 0, st, T, counts, mp, crank, tcomp);
        hipLaunchKernelGGL(k_cut_edges, grid_for_items(M, kAppendItems), dim3(kThreads), 0, st, M, n1, n2, sim, dp->track, tcomp, csize, (uint32_t)max_nodes, pair_key, pair_sim, cut_n);
        uint32_t *h_n = h_counts + 8;
        LFR_HIP_TRY(hipMemcpyAsync(h_n, cut_n, 4, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(stream_wait(st));
        const size_t n_cut_edges = h_n[0];
        // Meta edges (solve.cc:268-289,322-332): per unordered track pair the sum of the similarities -> int weight 100 * sum.
        // Sorted and summed on the device (radix sort + reduce-by-key: 0.7 M matches of config 5 took a host std::sort 60 ms);
        // sums of float32 values in fp64 are exact here, hence independent of the order of the additions.
        DevArena sum_arena;
        size_t n_pairs = 0;
        std::vector<unsigned long long> h_pair;
        std::vector<double> h_sum;
        std::vector<int32_t> h_tcomp((size_t)T), h_gc((size_t)T, -1);
        std::vector<uint32_t> h_tsize((size_t)T);
        if (n_cut_edges) {
            if (!sum_arena.init(ctx, 64 * n_cut_edges + ((size_t)16 << 20))) return LFR_ERR_NOMEM;
            unsigned long long *key_s = sum_arena.take_n<unsigned long long>(n_cut_edges), *key_u = sum_arena.take_n<unsigned long long>(n_cut_edges);
            double *sim_s = sum_arena.take_n<double>(n_cut_edges), *sim_u = sum_arena.take_n<double>(n_cut_edges);
            if (!key_s || !key_u || !sim_s || !sim_u) { set_error("graph stage: cut arena exhausted"); return LFR_ERR_NOMEM; }
            int tbits = 1;
            while (((int64_t)1 << tbits) < T) ++tbits;
            if ((rc = sort_pairs(sum_arena, pair_key, key_s, pair_sim, sim_s, (int64_t)n_cut_edges, 0, 32 + tbits, st)) != LFR_OK) return rc;
            if ((rc = sum_by_key(sum_arena, key_s, key_u, sim_s, sim_u, cut_n + 1, (int64_t)n_cut_edges, st)) != LFR_OK) return rc;
            LFR_HIP_TRY(hipMemcpyAsync(h_n + 1, cut_n + 1, 4, hipMemcpyDeviceToHost, st));
            LFR_HIP_TRY(stream_wait(st));
            n_pairs = h_n[1];
            h_pair.resize(n_pairs); h_sum.resize(n_pairs);
            LFR_HIP_TRY(hipMemcpyAsync(h_pair.data(), key_u, 8 * n_pairs, hipMemcpyDeviceToHost, st));
            LFR_HIP_TRY(hipMemcpyAsync(h_sum.data(), sim_u, 8 * n_pairs, hipMemcpyDeviceToHost, st));
        }
        LFR_HIP_TRY(hipMemcpyAsync(h_tcomp.data(), tcomp, 4 * (size_t)T, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(hipMemcpyAsync(h_tsize.data(), tsize, 4 * (size_t)T, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(stream_wait(st));
        lap("cut: meta edges of the oversized components on the host");
        {
            // the pairs arrive sorted by (t, u); group them by component, keeping that order (what the host stage feeds the cut): a counting
            // sort over the component ids (a std::stable_sort through two indirections was 2-3 ms of config 5's cut for 105 k pairs)
            int32_t n_comp_ids = 0;
            for (int64_t t = 0; t < T; ++t) n_comp_ids = std::max(n_comp_ids, h_tcomp[t] + 1);
            std::vector<uint32_t> first((size_t)n_comp_ids + 1, 0u), order(n_pairs);
            for (size_t k = 0; k < n_pairs; ++k) ++first[(size_t)h_tcomp[h_pair[k] >> 32] + 1];
            for (int32_t c = 0; c < n_comp_ids; ++c) first[c + 1] += first[c];
            {
                std::vector<uint32_t> next(first.begin(), first.end() - 1);
                for (size_t k = 0; k < n_pairs; ++k) order[next[h_tcomp[h_pair[k] >> 32]]++] = (uint32_t)k;
            }
            std::vector<int64_t> tsize64((size_t)T);
            for (int64_t t = 0; t < T; ++t) tsize64[t] = h_tsize[t];
            std::vector<std::pair<int, int>> e;
            std::vector<int> w;
            for (int32_t c = 0; c < n_comp_ids; ++c) {
                const size_t lo = first[c], hi = first[c + 1];
                if (hi == lo) continue;
                e.clear(); w.clear();
                e.reserve(hi - lo); w.reserve(hi - lo);
                for (size_t q = lo; q < hi; ++q) {
                    const unsigned long long key = h_pair[order[q]];
                    e.push_back({(int)(key >> 32), (int)(key & 0xffffffffull)});
                    w.push_back(static_cast<int>(100 * h_sum[order[q]]));          // solve.cc:329
                }
                const auto split = recursive_cut(e, w, tsize64, max_nodes);
                for (auto &it : split) h_gc[it.first] = it.second;
            }
            lap("cut: recursive bisection");
            // (an oversized component without any meta edge is a single track: nothing to cut, it stays whole)
            p.stats.n_cut_components = 0;
            {   // count the components above the cap the way the host stage does (with or without meta edges)
                std::vector<int64_t> csz;
                for (int64_t t = 0; t < T; ++t) { if ((size_t)h_tcomp[t] >= csz.size()) csz.resize(h_tcomp[t] + 1, 0); csz[h_tcomp[t]] += h_tsize[t]; }
                for (int64_t c : csz) if (c > max_nodes) ++p.stats.n_cut_components;
            }
        }
        LFR_HIP_TRY(hipMemcpyAsync(gc_dev, h_gc.data(), 4 * (size_t)T, hipMemcpyHostToDevice, st));
        // drop the cut edges and re-label (solve.cc:345-372), same kernels as the first labelling
        hipLaunchKernelGGL(k_iota, grid_for(N), dim3(kThreads), 0, st, N, mp_parent);
        hipLaunchKernelGGL(k_meta_union_cut, grid_for(M), dim3(kThreads), 0, st, M, n1, n2, dp->track, gc_dev, mp_parent);
        hipLaunchKernelGGL(k_cc_labels, grid_for(N), dim3(kThreads), 0, st, N, mp_parent, mp);
        LFR_HIP_TRY(hipMemsetAsync(cflag, 0, 4 * (size_t)(N + 1), st));
        LFR_HIP_TRY(hipMemsetAsync(csize, 0, 4 * (size_t)N, st));
        LFR_HIP_TRY(hipMemsetAsync(counts + CNT_MAX_COMP, 0, 4, st));
        hipLaunchKernelGGL(k_comp_flags, grid_for(N), dim3(kThreads), 0, st, N, counts, mp, cflag);
        if ((rc = exclusive_sum(arena, cflag, crank, N + 1, st)) != LFR_OK) return rc;
        hipLaunchKernelGGL(k_comp_sizes, grid_few(N), dim3(kThreads), 0, st, N, counts, mp, crank, tsize, csize);
        hipLaunchKernelGGL(k_max_csize, grid_few(N), dim3(kThreads), 0, st, N, counts, csize);
        hipLaunchKernelGGL(k_node_comp, grid_for(N), dim3(kThreads), 0, st, N, dp->track, mp, crank, dp->comp);
        LFR_HIP_TRY(hipGetLastError());
        LFR_HIP_TRY(hipEventRecord(c1, st));
        LFR_HIP_TRY(hipMemcpyAsync(h_counts, counts, 4 * CNT_WORDS, hipMemcpyDeviceToHost, st));
        LFR_HIP_TRY(stream_wait(st));          // (the staging vectors above die at scope end)
        lap("cut: re-labelled");
        p.stats.n_components = h_counts[CNT_COMPS]; p.stats.max_component_size = h_counts[CNT_MAX_COMP];
    dp->max_cc_matches = h_counts[CNT_MAX_SEG];
        float cms = 0.f;
        LFR_HIP_TRY(hipEventElapsedTime(&cms, c0, c1));
        p.stats.graph_cut_ms += cms;
    }
    return LFR_OK;
}

int graph_make_resident(const Graph &g, int device) {
    std::shared_ptr<DevGraph> dg;
    return ensure_dev_graph(g, device, true, dg);
}

}  // namespace lfr

// =================================================================================================
// C ABI: device residency of the graph
// =================================================================================================
extern "C" {

int lfr_graph_to_device(const lfr_graph *g, int device) {
    if (!g) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    return lfr::graph_make_resident(g->g, device);
}

int lfr_graph_evict_device(const lfr_graph *g) {
    if (!g) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g->g.dev_mu);
    g->g.devgs.clear();                               // problems / batches built from it keep their own reference
    return LFR_OK;
}

}  // extern "C"

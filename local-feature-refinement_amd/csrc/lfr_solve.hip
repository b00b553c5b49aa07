// Batched per-component Levenberg-Marquardt on MI355X (gfx950) — replaces the hot loop of the
// reference, solve.cc:614-635 -> create_and_solve_problem (solve.cc:79-160) -> ceres::Solve,
// with the cost model of cost.cc:13-48,78-90.  Control flow mirrors Ceres' trust-region loop
// decision for decision; fp64 throughout.  The packed kernel uses no MFMA (its 2N-variable blocks are tiny);
// the workgroup kernels run the trailing updates of their factorizations on the fp64 matrix cores.
//
// Kernels:
//   solve_packed_kernel        ONE launch for every component of up to 32 rows: a wave64 hosts 64/S components
//                              (S = 8, 16, 32 or 64 lanes each), one wave per workgroup, no barriers.  Edge slots
//                              live in VGPRs (or are re-read per sweep beyond the resident ones), the two directions
//                              of a match sit in neighbouring lanes and exchange their terms through DPP before
//                              7 ds_add_f64 per edge assemble J^T J in LDS, the damped system is eliminated in
//                              registers (lane = row) with ds_swizzle broadcasts, sized per wave.
//   solve_block_kernel         persistent 128/256/512-thread workgroups over the components of up to 192 rows (three LDS-footprint
//                              classes): packed J^T J in LDS, fused evaluate-and-assemble sweep (four to eight lanes per node, the
//                              records re-streamed from HBM and nothing else), blocked LDL^T with 16-column panels - the diagonal
//                              tiles on a wave that runs ahead of the others, trailing updates AND the tiles' substitution (tile x M,
//                              M = the diagonal tile's substitution as a matrix) on the fp64 matrix cores.
//   solve_tree_kernel          one 512-thread workgroup per component above 192 rows: sparse LDL^T along the elimination tree of a
//                              nested-dissection order (lfr_treeplan.cpp), 16x16 tiles in an HBM workspace, columns of one tree level
//                              factored side by side by the workgroup's waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <array>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lfr_assemble.hpp"
#include "lfr_device.hpp"
#include "lfr_internal.hpp"

using namespace lfrdev;
using lfr::CompDesc;
using lfr::EdgeRec;

namespace {

struct CompInfoDev {
    int32_t iterations, termination, n_successful, n_ls_evals, n_cand_evals, exec_passes;
    double final_cost;
};
static_assert(sizeof(CompInfoDev) == 32, "CompInfoDev layout");

struct KernelArgs {
    const CompDesc *descs;
    const EdgeRec *edges;
    const uint32_t *node_ids;
    double *positions;          // 2 * n_nodes of the whole graph
    CompInfoDev *infos;
    const lfr::NodeInc *node_inc;   // parallel to node_ids
    const uint32_t *in_idx;     // parallel to edges
    double *workspace;          // workgroup kernels: per-edge scratch (+ packed matrices for the HBM variant)
    const uint64_t *ws_off;     // per desc: packed-matrix offset (HBM variant)
    const uint64_t *es_off;     // per desc: per-edge scratch offset (8 doubles per edge)
    unsigned long long *prof;   // -DLFR_PROFILE_PHASES: per-class cycle counters [cls*8 + phase]
    unsigned int *queue;        // workgroup classes: next component of the class (one counter per class, zeroed per solve)
    const uint32_t *wg_order;   // workgroup classes: descriptors in the order the queue hands them out (longest expected first)
    int wg_begin;               // first descriptor of the workgroup classes (wg_order[0] belongs to it)
    int desc_begin, desc_end;
    int tukey_variant;
    int scratch_sweep;         // 1: the workgroup kernels use the scratch sweep of rounds 1-2 (LFR_SCRATCH_SWEEP=1 at batch creation; A/B and tests)
    int cls;
    // fused gather (packed classes of a device-assembled whole batch): record p is directed edge edge_ref[p] of the graph - flow row
    // (f_row ? f_row[m] : m) of f_disp2 (even ids) / f_disp1 (odd ids), similarity f_sim[m], m = id >> 1 - with local indices edge_word[p]
    const uint32_t *edge_ref, *edge_word, *f_row;
    const float *f_disp1, *f_disp2, *f_sim;
    // elimination-tree class: teams of workgroups per component (nullptr: one workgroup per component); team_work[k]: smallest
    // hand-out key (k_wg_order_keys' `work`) solved by 2 << k workgroups
    unsigned int *team_ctl;
    double *team_red;
    uint32_t team_work[3];
    uint32_t team_epoch;        // differs between launches that share the reduction slots
    uint32_t team_patience_us;  // teams that cannot form for this long with nobody at work: the launch goes on one workgroup per component (LFR_TEAM_PATIENCE_MS)
    unsigned long long *trace;  // -DLFR_TRACE_TREE: [0] = words used, then {s_memtime, type << 56 | wave of the team << 48 | iteration << 32 | column} pairs
};

// =============================================================================================
// packed sub-group kernel: a wave64 hosts G = 64/S components, S = NV*LPR lanes each.
//   evaluation : lane = edge slot, EPL edge slots per lane: the first RES resident in VGPRs, the rest re-read
//                from L2/HBM by every sweep
//   solve      : NV rows; a row is held by LPR lanes with its columns interleaved (h[c] = column LPR*c + part),
//                Gauss-Jordan with ds_swizzle / v_readlane broadcasts inside the group, instantiated per
//                live-column count
// Groups advance through the same LM state machine in lockstep rounds (solve -> evaluate ->
// decide); rare paths (invalid step, line-search contraction, rejected step) just take extra
// rounds for their group.  Reductions are DPP butterflies; no barriers (a wave is its own
// synchronisation domain), no cross-group traffic.
// =============================================================================================
#ifdef LFR_PROFILE_PHASES
#define PROF_DECL unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long pt0_ = __builtin_amdgcn_s_memtime();
#define PROF_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_[i] += t_ - pt0_; pt0_ = t_; } while (0)
#define PROF_FLUSH() do { if (lane == 0) for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&a.prof[a.cls * 8 + i_], pt_[i_]); if (lane == 0) atomicAdd(&a.prof[a.cls * 8 + 7], 1ull); } while (0)
#elif defined(LFR_ISA_MARKS)
// -DLFR_ISA_MARKS (scripts/isa_account.py, compile only): comments in the ISA that delimit the phases of the packed kernel, so that the
// instructions of each phase can be counted by kind
#define PROF_DECL
#define PROF_MARK(i) asm volatile("; LFR_MARK " #i)
#define PROF_FLUSH()
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH()
#endif
#ifdef LFR_ISA_MARKS
#define ISA_MARK(name) asm volatile("; LFR_MARK " name)
#else
#define ISA_MARK(name)
#endif
// -DLFR_PROFILE_SWEEP (with LFR_PROFILE_PHASES): slot 3 = edge evaluation of the workgroup kernel's sweeps, slots 0/2 keep their
// assembly walks, the factorization reports as one number in slot 1
#if defined(LFR_PROFILE_PHASES) && defined(LFR_PROFILE_SWEEP)
#define PROF_SWEEP_MARK(i) PROF_MARK(i)
#define PROF_FACTOR_MARK(i) PROF_MARK(1)
#else
#define PROF_SWEEP_MARK(i)
#define PROF_FACTOR_MARK(i) PROF_MARK(i)
#endif
// Experiments (scripts/ab_packed.py builds them as variants): what a phase of the packed kernel costs, measured by running it TWICE
// with unchanged results (round 6, config 4, 0.45 ms: atomics +15 %, evaluation +21 %, matrix build + elimination +16 %,
// reductions +2 %, zeroing +2 % - profiles/r06_ablation_packed_kernel.txt)
#ifdef LFR_DOUBLE_ATOMICS          // every term of the assembly added as two halves
#define LFR_ASM_ADD(p, v) do { const double v_ = 0.5 * (v); atomicAdd(p, v_); atomicAdd(p, v_); } while (0)
#else
#define LFR_ASM_ADD(p, v) atomicAdd(p, v)
#endif
#ifndef LFR_LOG_REGS
#define LFR_LOG_REGS 1             // 8- / 16-row classes: the logarithm's series coefficients live in VGPRs (0: literals moved into SGPRs at every use)
#endif
#ifndef LFR_GJ_DPP
#define LFR_GJ_DPP 2               // elimination of the 16-row (1) and also the 8-row (2) packed class with DPP row broadcasts; 0: ds_swizzle
#endif
enum : int { PH_SOLVE = 0, PH_EVAL_INIT = 1, PH_EVAL_LS = 2, PH_EVAL_CAND = 3, PH_REEVAL = 4, PH_DONE = 5 };

// Gauss-Jordan elimination of the damped normal equations held one row per lane, no pivoting (SPD);
// padded rows are identity.  The LPR lanes of a row hold its columns INTERLEAVED (lane part p owns
// columns LPR*c + p in h[c]), so the columns already eliminated (j <= K) fall off both parts evenly
// and the register loops of step K start at c0 = (K+1)/LPR.  Step K: ONE burst of ds_swizzle
// broadcasts of the pivot row into distinct registers (a single LDS-crossbar round trip), a Newton
// reciprocal of the pivot, then the rank-1 update.  Written as a compile-time recursion of nested
// `if (K+1 < n)` so the unrolled steps share one exit and no copies of h[] are made at merges.
// (Skipping the columns beyond the wave's largest system behind scalar branches inside the steps was
// measured: the extra control flow costs ~150 spilled VGPRs and 25 % of the kernel - hence the CL
// instantiations below, selected once per step.)
// CL = registers per lane that can hold live columns: the caller picks the smallest instantiation that
// covers the largest system of the wave (wave-uniform), so columns beyond it cost nothing.
template <int NV, int LPR, int K, int CL>
struct GaussJordan {
    static constexpr int S = NV * LPR;
    static constexpr int kPartAnd = (NV == 8) ? 0x18 : (NV == 16) ? 0x10 : 0x00;
    static __device__ __forceinline__ void run(double (&h)[CL], double &rhs, double &piv_own, double &minpiv,
                                               int row, int part, int n_steps) {
        constexpr int pk = K % LPR, ck = K / LPR;                      // part / register holding column K
        constexpr int c0 = (K + 1) / LPR;                              // first register with a live column (> K)
#if LFR_GJ_DPP
        if constexpr (LPR == 1 && (NV == 16 || (NV == 8 && LFR_GJ_DPP >= 2))) {
            // One row per lane, the group inside one 16-lane DPP row: the pivot row reaches the rank-1 update as the DPP operand of
            // the v_fmac_f64 itself (row_newbcast:K) - one instruction per column and no LDS crossbar (until round 6: two ds_swizzle
            // and an fma per column, 75 % of the kernel's LDS instructions).  NV == 8: two groups per DPP row, the update is issued
            // per half with a bank mask (lanes 0-7 take lane K, lanes 8-15 lane K + 8).
            const double piv = (NV == 16) ? bcast16_f64<K>(h[K]) : bcast8_f64<K>(h[K]);
            minpiv = fmin(minpiv, piv);
            double nf = -(h[K] * fast_rcp(piv));
            const bool is_k = row == K;
            nf = is_k ? 0.0 : nf;                                      // (the pivot lane's own row stays: h += 0 * h)
            piv_own = is_k ? piv : piv_own;
            if constexpr (NV == 16) fmac_bcast_row<K, 0xf, c0, CL>(nf, h, rhs);
            else { fmac_bcast_row<K, 0x3, c0, CL>(nf, h, rhs); fmac_bcast_row<K + 8, 0xc, c0, CL>(nf, h, rhs); }
            if constexpr (K + 1 < CL && K + 1 < NV) {
                if (K + 1 < n_steps) GaussJordan<NV, LPR, K + 1, CL>::run(h, rhs, piv_own, minpiv, row, part, n_steps);
            }
            return;
        }
#endif
        double pr[CL];
#pragma unroll
        for (int c = c0; c < CL; ++c) pr[c] = swz_bcast<kPartAnd, K % 32>(h[c]);
        const double prhs = swz_bcast<kPartAnd, K % 32>(rhs);
        const double piv = (S == 64) ? readlane_f64(h[ck], K + NV * pk)
                                     : swz_bcast<(S == 8) ? 0x18 : (S == 16) ? 0x10 : 0x00, (K + NV * pk) % 32>(h[ck]);
        double f = h[ck];                                              // valid in the lanes of part pk
        if (LPR == 2) {
            const double fx = (NV == 16) ? swizzle_f64<kSwizzleXor16>(f) : xor32_f64(f);
            f = (part == pk) ? f : fx;
        }
        minpiv = fmin(minpiv, piv);
        f *= fast_rcp(piv);
        const bool is_k = row == K;
        f = is_k ? 0.0 : f;
        piv_own = is_k ? piv : piv_own;
#pragma unroll
        for (int c = c0; c < CL; ++c) h[c] = fma(-f, pr[c], h[c]);
        rhs = fma(-f, prhs, rhs);
        if constexpr (K + 1 < CL * LPR && K + 1 < NV) {
            if (K + 1 < n_steps) GaussJordan<NV, LPR, K + 1, CL>::run(h, rhs, piv_own, minpiv, row, part, n_steps);
        }
    }
};

template <int NV>
struct alignas(16) GroupLds {
    static constexpr int LD = NV + 1;
    double A[NV * LD];         // J^T J (lower triangle) of the last evaluation
    double g[NV];              // J^T r of the last evaluation
    double x[NV + 2];          // evaluation point; slots 2*n_var, 2*n_var+1 stay 0 (constants)
    // cold per-group state (line-search bookkeeping, counters): lives here, not in VGPRs
    double ls_prev_x, ls_prev_value, ls_prev_gradient;
    double step_norm2, xnorm2;  // |x - x_trial|^2 and |x_trial|^2 of the current trial point (written with it, read by the decision after the sweep)
    int ls_prev_flags, ls_iter, n_successful, n_ls_evals, n_cand, exec_passes;
};

#ifndef LFR_GROUP_WAVES
#define LFR_GROUP_WAVES 2          // waves per SIMD the packed kernel is register-budgeted for
#endif
#ifndef LFR_PACKED_WAVES
#define LFR_PACKED_WAVES 1         // waves per workgroup of the packed kernel: the waves are independent, and a
                                  // one-wave workgroup frees its slot the moment it finishes (4 -> 1: -7 %)
#endif
constexpr int kPackedWaves = LFR_PACKED_WAVES;
// one record of a packed class: 18 flow values, similarity, src | (dst | kind << 15) << 16.  FUSED: gathered from the graph's
// match-ordered arrays (9 + 3 loads); otherwise the 80-byte record of the batch (5 loads)
template <bool FUSED>
__device__ __forceinline__ void load_packed_edge(const KernelArgs &a, const uint32_t rec, float (&fl)[18], float &sm, uint32_t &word) {
    if constexpr (FUSED) {
        const uint32_t eid = a.edge_ref[rec], m = eid >> 1;
        const size_t row = a.f_row ? (size_t)a.f_row[m] : (size_t)m;
        const uint2 *fp = reinterpret_cast<const uint2 *>(((eid & 1u) ? a.f_disp1 : a.f_disp2) + 18 * row);
        uint2 q[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) q[i] = fp[i];
        sm = a.f_sim[m];
        word = a.edge_word[rec];
#pragma unroll
        for (int i = 0; i < 9; ++i) { fl[2 * i] = __uint_as_float(q[i].x); fl[2 * i + 1] = __uint_as_float(q[i].y); }
    } else {
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.edges + rec);
        uint4 q[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) q[i] = rp[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fl[4 * i] = __uint_as_float(q[i].x); fl[4 * i + 1] = __uint_as_float(q[i].y);
            fl[4 * i + 2] = __uint_as_float(q[i].z); fl[4 * i + 3] = __uint_as_float(q[i].w);
        }
        fl[16] = __uint_as_float(q[4].x); fl[17] = __uint_as_float(q[4].y);
        sm = __uint_as_float(q[4].z);
        word = q[4].w;
    }
}

template <int NV, int LPR, int EPL, bool FUSED>
__device__ __forceinline__ void solve_group_body(const KernelArgs &a, const int block_in_class, unsigned char *lds_raw) {
    constexpr int S = NV * LPR, G = 64 / S, CPL = NV / LPR, LD = NV + 1;
    static_assert(S <= 64 && (LPR == 1 || LPR == 2), "group geometry");
#ifdef LFR_ISA_MARKS
    asm volatile("; LFR_CLASS %0 %1 %2" :: "n"(NV), "n"(LPR), "n"(EPL));
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gid = lane / S, sl = lane % S;
    const int row = sl % NV, part = sl / NV;
    const int ci0 = a.desc_begin + (block_in_class * kPackedWaves + wave) * G;
    if (ci0 >= a.desc_end) return;                    // wave-uniform
#if defined(LFR_PROFILE_WGTIME) && LFR_PROFILE_WGTIME == 3
    const unsigned long long wave_r0_ = wall_clock64();
#endif
    const int ci = ci0 + gid;
    const bool have = ci < a.desc_end;
    GroupLds<NV> &L = reinterpret_cast<GroupLds<NV> *>(lds_raw)[wave * G + gid];

    CompDesc d;
    d.edge_off = 0; d.n_edges = 0; d.node_off = 0; d.n_nodes = 0; d.n_var = 0;
    if (have) d = a.descs[ci];
    const int n_var = d.n_var, nv2 = 2 * n_var, E = (int)d.n_edges;
    const int tv = a.tukey_variant;
    const bool is_row = row < nv2;
    const bool own = is_row && part == 0;             // one lane per row: reductions, LDS writes
    int nv2_max = nv2;                                // largest row count of the wave (wave-uniform)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nv2_max = max(nv2_max, __shfl_xor(nv2_max, m, 64));
    nv2_max = __builtin_amdgcn_readfirstlane(nv2_max);

    // ---- edges -> registers (the only HBM read of the solve) ----
    // Slots beyond the third are re-read from HBM/L2 by every sweep instead of being held: only the
    // <=256-edge class has them, and 80 resident flow registers there cost more in spills than the loads.
#ifndef LFR_RES4
#define LFR_RES4 2
#endif
#ifndef LFR_RES16
#define LFR_RES16 3
#endif
    constexpr int RES = EPL <= 3 ? EPL : (NV == 16 ? LFR_RES16 : LFR_RES4);
    float flow[RES][18];
    float sim[RES];
    uint32_t idx[RES];          // src | (dst|kind<<15) << 16, decoded at every use (keeps 5 VGPRs/slot free)
#pragma unroll
    for (int k = 0; k < RES; ++k) {
        const int e = sl + S * k;
        const bool on = e < E;
        load_packed_edge<FUSED>(a, d.edge_off + (on ? e : 0), flow[k], sim[k], idx[k]);
        if (!on) {
#pragma unroll
            for (int i = 0; i < 18; ++i) flow[k][i] = 0.f;
            sim[k] = 0.f; idx[k] = 0u;
        }
    }
    if (sl < NV) L.x[sl] = 0.0;
    if (sl < 2) L.x[NV + sl] = 0.0;
    for (int i = sl; i < NV * LD; i += S) L.A[i] = 0.0;           // rows >= nv2 stay zero for the whole solve (the sweeps re-zero rows < nv2)

    // ---- group state (uniform inside a group; 'row' values are identical in the LPR lanes of a row) ----
    int phase = have ? PH_EVAL_INIT : PH_DONE;
    double xi = 0.0, gi = 0.0, scale = 1.0, diag = 1.0;            // row
    double inv_scale = 1.0;                                          // row: 1 / scale = 1 + sqrt(a_ii at iteration 0)
    double xt = 0.0, delta = 0.0;                                    // row
    double cost = 0.0, radius = kInitialRadius, x_norm = 0.0, gmax = 0.0;
    double g_dot_delta = 0.0, model_cost_change = 0.0, alpha = 1.0;
    int n_reject = 0;                                 // decrease_factor = 2^(1 + n_reject)  (Ceres: 2, 4, 8, ...)
    bool reuse_diagonal = false, step_successful = true, a_dirty = false;
    bool cost_only = false;                           // this round evaluates the cost only (predicted last iteration)
    int n_invalid = 0, iteration = 0, term = LFR_TERM_CONVERGENCE;
    if (sl == 0) { L.n_successful = 0; L.n_ls_evals = 0; L.n_cand = 0; L.exec_passes = 0; L.ls_iter = 0; L.ls_prev_flags = 0; }

    bool at_zero = true;                              // wave-uniform: the first sweep evaluates every edge at the origin (PH_EVAL_INIT)
    // the logarithm's series coefficients in 22 VGPRs for the whole solve where the class has them to spare (8- and 16-row classes)
    constexpr bool kLogRegs = NV <= 16 && LPR == 1 && LFR_LOG_REGS;
    LogCoef logc;
    if constexpr (kLogRegs) logc.load();
    const LogCoef *lcp = kLogRegs ? &logc : nullptr;
    PROF_DECL
    PROF_MARK(0);                                     // 0: prologue (edge load)
    for (;;) {
        ISA_MARK("loop_top");
        if (!__any(phase != PH_DONE)) break;

        // ======================= A: iteration entry + LM step =======================
        if (__any(phase == PH_SOLVE)) {
            bool ps = phase == PH_SOLVE;
            if (ps) {   // FinalizeIterationAndCheckIfMinimizerCanContinue
                if (iteration >= kMaxIterations) { term = LFR_TERM_NO_CONVERGENCE; phase = PH_DONE; ps = false; }
                else if ((step_successful && gmax <= kGradientTol) || radius <= kMinRadius) { phase = PH_DONE; ps = false; }
                else if (a_dirty) { phase = PH_REEVAL; ps = false; if (own) L.x[row] = xi; }     // J^T J at x was overwritten
            }
            if (__any(ps)) {
                wave_lds_sync();
                if (ps) { ++iteration; step_successful = false; }
                const double *A = L.A;
                const double aii = is_row ? A[row * LD + row] : 1.0;
                if (ps && !reuse_diagonal) diag = fmin(fmax(scale * scale * aii, kMinLmDiag), kMaxLmDiag);
                // (S A S + D^2) y = S g  <=>  (A + S^-1 D^2 S^-1) (S y) = g: the unscaled system with the same Jacobi-
                // scaled LM diagonal gives the step directly and saves two multiplies and an LDS read per element.
                // D^2 = diag / radius (Ceres squares sqrt(diag / radius): the same up to two roundings; radius is in [1e-32, 1e16])
                const double dd = diag * fast_rcp(radius) * (inv_scale * inv_scale);
                const double dd_lane = is_row ? dd : 1.0;             // padded rows are identity
                double rhs = is_row ? gi : 0.0;
                const double rhs0 = rhs;
                double piv_own = 1.0;
                bool fail = false;
                double minpiv = 1.0;
                // build the lane's columns (LPR*c + part of the own row) and eliminate, in the instantiation
                // sized for the largest system of the wave
                auto lm_solve = [&](auto cl_tag) {
                    constexpr int CL = decltype(cl_tag)::value;
                    double h[CL];
#pragma unroll
                    for (int c = 0; c < CL; ++c) {
                        const int j = LPR * c + part;
                        if constexpr (NV == 32) {     // (the unconditional form below lets the 24 loads run ahead together: 40-60 spilled VGPRs here)
                            double v = 0.0;
                            if (is_row && j < nv2) v = (j <= row ? A[row * LD + j] : A[j * LD + row]);
                            if (j == row) v = is_row ? v + dd : 1.0;
                            h[c] = v;
                        } else {
                            // rows >= nv2 of A are zero (prologue), so columns beyond the system and padded rows read 0 without a mask
                            const double v = (j <= row ? A[row * LD + j] : A[j * LD + row]);
                            h[c] = (j == row) ? v + dd_lane : v;
                        }
                    }
                    PROF_MARK(5);                     // 5: step setup (diagonal, h build)
                    ISA_MARK("gauss_jordan");
                    GaussJordan<NV, LPR, 0, CL>::run(h, rhs, piv_own, minpiv, row, part, nv2_max);
                };
                const int c_hi = (nv2_max + LPR - 1) / LPR;          // wave-uniform
#ifdef LFR_ABL_GJ              // matrix build + elimination twice (the first result discarded through an opaque reset)
#define LFR_CL(n) do { lm_solve(std::integral_constant<int, n>{}); asm volatile("" : "+v"(rhs), "+v"(piv_own), "+v"(minpiv)); rhs = rhs0; piv_own = 1.0; minpiv = 1.0; lm_solve(std::integral_constant<int, n>{}); } while (0)
#else
#define LFR_CL(n) lm_solve(std::integral_constant<int, n>{})
#endif
                if constexpr (CPL == 8 && LPR == 1) {             // rows come in pairs: even sizes only
                    if (c_hi <= 2) LFR_CL(2); else if (c_hi <= 4) LFR_CL(4); else if (c_hi <= 6) LFR_CL(6); else LFR_CL(8);
                } else if constexpr (CPL == 8) {
                    if (c_hi <= 4) LFR_CL(4); else if (c_hi <= 6) LFR_CL(6); else if (c_hi <= 7) LFR_CL(7); else LFR_CL(8);
                } else if constexpr (CPL == 32) {                  // two <=32-row systems per wave
                    if (c_hi <= 18) LFR_CL(18); else if (c_hi <= 20) LFR_CL(20); else if (c_hi <= 22) LFR_CL(22);
                    else LFR_CL(24);                              // the class holds <= 24 rows (classify())
                } else if constexpr (LPR == 1) {
                    if (c_hi <= 10) LFR_CL(10); else if (c_hi <= 12) LFR_CL(12); else if (c_hi <= 14) LFR_CL(14); else LFR_CL(16);
                } else {
                    if (c_hi <= 10) LFR_CL(10); else if (c_hi <= 12) LFR_CL(12); else if (c_hi <= 14) LFR_CL(14); else LFR_CL(16);
                }
#undef LFR_CL
                ISA_MARK("step_reductions");
                fail = !(minpiv > 0.0);
                const double step = is_row ? -(rhs * fast_rcp(piv_own)) : 0.0;
                const unsigned long long badmask = __ballot(is_row && !isfinite(step));
                const unsigned long long gmask = (S == 64) ? ~0ull : ((1ull << (S & 63)) - 1);
                const bool bad = ((badmask >> ((gid * S) & 63)) & gmask) != 0;
                const double dl = step;
                // the four sums of an LM step in one transposed butterfly: model change, g . delta, and |x - x_trial|^2, |x_trial|^2 of
                // the full step's trial point (the decision after the sweep reads the last two from LDS; a line-search contraction
                // recomputes them for its point, and only it needs max |delta| and g_new . delta)
                const double xt_full = clampb(__dadd_rn(xi, dl));
                double mcc = own ? (-rhs0 * step + dd * step * step) : 0.0, gdd = own ? gi * dl : 0.0;
                double sn2 = own ? (xi - xt_full) * (xi - xt_full) : 0.0, xn2 = own ? xt_full * xt_full : 0.0;
#ifdef LFR_ABL_RED             // every reduction twice
                { double a_ = mcc, b_ = gdd, c_ = sn2, d_ = xn2; asm volatile("" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_)); group_sum4<S>(a_, b_, c_, d_);
                  asm volatile("" :: "v"(a_), "v"(b_), "v"(c_), "v"(d_)); }
#endif
                group_sum4<S>(mcc, gdd, sn2, xn2);
                mcc *= 0.5;
                if (ps) {
                    reuse_diagonal = true;
                    const bool valid = !fail && !bad && mcc > 0.0;
                    if (!valid) {
                        if (++n_invalid >= kMaxInvalid) { term = LFR_TERM_FAILURE; phase = PH_DONE; }
                        else { radius = ldexp(radius, -(1 + n_reject)); ++n_reject; }          // StepIsInvalid -> StepRejected(0): radius / 2^(1 + n_reject)
                    } else {
                        n_invalid = 0;
                        model_cost_change = mcc; delta = dl; g_dot_delta = gdd;
                        alpha = 1.0;
                        // A predicted decrease below the function tolerance almost always ends the solve at the
                        // next test (config 4: every final iteration, 0.1 % of the others): evaluate the cost only;
                        // if the solve does not stop there, the same point is evaluated again in full.
                        cost_only = mcc <= kFunctionTol * cost;
                        if (sl == 0) { L.step_norm2 = sn2; L.xnorm2 = xn2; L.ls_iter = 0; L.ls_prev_flags = 0; }
                        xt = xt_full;
                        if (own) L.x[row] = xt;
                        phase = PH_EVAL_LS;
                    }
                }
            }
        }

        PROF_MARK(1);                                 // 1: iteration entry + LM step
        // ======================= B: one sweep over the edges of the evaluating groups =======================
        const bool pe = phase == PH_EVAL_INIT || phase == PH_EVAL_LS || phase == PH_EVAL_CAND || phase == PH_REEVAL;
        if (!__any(pe)) continue;
        const bool jac = !cost_only;                  // (cost_only is only ever set in PH_EVAL_LS)
        if (pe && jac) {
            // rows >= nv2 are never touched.  16-byte stores: as many LDS instructions went into this zeroing as into the atomics
            double2 *A2 = reinterpret_cast<double2 *>(L.A);
            for (int i = sl; i < (nv2 * LD + 1) / 2; i += S) A2[i] = make_double2(0.0, 0.0);
            if (sl < NV / 2) reinterpret_cast<double2 *>(L.g)[sl] = make_double2(0.0, 0.0);
#ifdef LFR_ABL_ZERO            // the zeroing twice
            wave_lds_sync();
            for (int i = sl; i < (nv2 * LD + 1) / 2; i += S) A2[i] = make_double2(0.0, 0.0);
#endif
        }
        wave_lds_sync();
        PROF_MARK(6);                                 // 6: zero J^T J
        double cost_l = 0.0;
        if (__any(pe && jac)) {
            ISA_MARK("sweep_setup");
            // full sweep (a cost-only group riding in this wave evaluates in full too, but assembles nothing)
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                if (!(pe && sl + S * k < E)) continue;
                float flow_k[18]; float sim_k; uint32_t pk;
                if (k < RES) {
#pragma unroll
                    for (int i = 0; i < 18; ++i) flow_k[i] = flow[k < RES ? k : 0][i];
                    sim_k = sim[k < RES ? k : 0]; pk = idx[k < RES ? k : 0];
                } else {
                    load_packed_edge<FUSED>(a, d.edge_off + (sl + S * k), flow_k, sim_k, pk);
                }
                asm volatile("" : "+v"(pk));              // decode here, do not hoist 5 derived values per slot
                const int es = (int)(pk & 0xffffu), ed = (int)((pk >> 16) & 0x7fffu), ekind = (int)(pk >> 31);
                const int xa = 2 * min(es, n_var), xb = 2 * min(ed, n_var);      // constants read the zero slot
                const int ra = es < n_var ? 2 * es : -1, rb = ed < n_var ? 2 * ed : -1;
                EdgeOut o;
                ISA_MARK("eval");
                eval_edge<true>(flow_k, sim_k, ekind, tv, L.x[xa], L.x[xa + 1], L.x[xb], L.x[xb + 1], o, at_zero, lcp);
#ifdef LFR_ABL_EVAL            // the evaluation twice (the second on opaque copies of the positions), results averaged (= unchanged)
                {
                    double y0 = L.x[xa], y1 = L.x[xa + 1], y2 = L.x[xb], y3 = L.x[xb + 1];
                    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
                    EdgeOut o2;
                    eval_edge<true>(flow_k, sim_k, ekind, tv, y0, y1, y2, y3, o2, at_zero, lcp);
                    o.cost = 0.5 * (o.cost + o2.cost); o.r0 = 0.5 * (o.r0 + o2.r0); o.r1 = 0.5 * (o.r1 + o2.r1); o.sq = 0.5 * (o.sq + o2.sq);
                    o.j00 = 0.5 * (o.j00 + o2.j00); o.j01 = 0.5 * (o.j01 + o2.j01); o.j10 = 0.5 * (o.j10 + o2.j10); o.j11 = 0.5 * (o.j11 + o2.j11);
                }
#endif
                ISA_MARK("assemble");
                cost_l += o.cost;
                if (!jac) continue;
                double *A = L.A, *g = L.g;
                // The neighbouring lane (lane ^ 1) holds the opposite direction of the same match (packed classes
                // are assembled in edge-id order: records 2m, 2m+1), so its d r / d x_dst = sq' * I terms land on
                // THIS lane's source block and the two cross blocks coincide: exchange them through DPP and issue
                // 7 LDS atomics per edge instead of 17.
                const int q = sl & 1;
                const double p_w = dpp_f64<kDppQuadXor1>(o.sq * o.sq);
                const double p_g0 = dpp_f64<kDppQuadXor1>(o.sq * o.r0);
                const double p_g1 = dpp_f64<kDppQuadXor1>(o.sq * o.r1);
                // cross block M[rb+i][ra+j] = sq*J_ij + sq'*J'_ji: the even lane owns (0,0),(1,1), the odd lane (1,0),(0,1)
                const double c_send1 = o.sq * (q ? o.j00 : o.j01), c_send2 = o.sq * (q ? o.j11 : o.j10);
                const double c_own1 = o.sq * (q ? o.j10 : o.j00), c_own2 = o.sq * (q ? o.j01 : o.j11);
                const double c1 = c_own1 + dpp_f64<kDppQuadXor1>(c_send1);
                const double c2 = c_own2 + dpp_f64<kDppQuadXor1>(c_send2);
                if (ra >= 0) {
                    LFR_ASM_ADD(&A[ra * LD + ra], o.j00 * o.j00 + o.j10 * o.j10 + p_w);
                    LFR_ASM_ADD(&A[(ra + 1) * LD + ra], o.j01 * o.j00 + o.j11 * o.j10);
                    LFR_ASM_ADD(&A[(ra + 1) * LD + ra + 1], o.j01 * o.j01 + o.j11 * o.j11 + p_w);
                    LFR_ASM_ADD(&g[ra], o.j00 * o.r0 + o.j10 * o.r1 + p_g0);
                    LFR_ASM_ADD(&g[ra + 1], o.j01 * o.r0 + o.j11 * o.r1 + p_g1);
                }
                if (ra >= 0 && rb >= 0) {
                    const int r1 = rb + q, k1 = ra, r2 = rb + 1 - q, k2 = ra + 1;
                    LFR_ASM_ADD(&A[rb > ra ? r1 * LD + k1 : k1 * LD + r1], c1);
                    LFR_ASM_ADD(&A[rb > ra ? r2 * LD + k2 : k2 * LD + r2], c2);
                }
            }
        } else {
            // every evaluating group of the wave wants the cost only: no derivatives, no assembly
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                if (!(pe && sl + S * k < E)) continue;
                float flow_k[18]; float sim_k; uint32_t pk;
                if (k < RES) {
#pragma unroll
                    for (int i = 0; i < 18; ++i) flow_k[i] = flow[k < RES ? k : 0][i];
                    sim_k = sim[k < RES ? k : 0]; pk = idx[k < RES ? k : 0];
                } else {
                    load_packed_edge<FUSED>(a, d.edge_off + (sl + S * k), flow_k, sim_k, pk);
                }
                asm volatile("" : "+v"(pk));              // decode here, do not hoist 5 derived values per slot
                const int es = (int)(pk & 0xffffu), ed = (int)((pk >> 16) & 0x7fffu), ekind = (int)(pk >> 31);
                const int xa = 2 * min(es, n_var), xb = 2 * min(ed, n_var);      // constants read the zero slot
                const int ra = es < n_var ? 2 * es : -1, rb = ed < n_var ? 2 * ed : -1;
                EdgeOut o;
                (void)ra; (void)rb;
                ISA_MARK("eval_cost_only");
                eval_edge<false>(flow_k, sim_k, ekind, tv, L.x[xa], L.x[xa + 1], L.x[xb], L.x[xb + 1], o, false, lcp);
                ISA_MARK("cost_only_loop");
                cost_l += o.cost;
            }
        }
        at_zero = false;
        wave_lds_sync();
        PROF_MARK(2);                                 // 2: edge sweep (evaluate + assemble)
        // cross-lane quantities of every possible transition (uniform control flow)
#ifdef LFR_ABL_RED
        { double a_ = cost_l; asm volatile("" : "+v"(a_)); a_ = group_sum<S>(a_); asm volatile("" :: "v"(a_)); a_ = group_max<S>(a_); asm volatile("" :: "v"(a_)); }
#endif
        const double cost_e = group_sum<S>(cost_l);
        const double xe = (phase == PH_EVAL_INIT || phase == PH_REEVAL) ? xi : xt;     // row: the evaluated point
        const double gnew = is_row ? L.g[row] : 0.0;
        const double gmax_new = group_max<S>(is_row ? fabs(xe - clampb(xe - gnew)) : 0.0);
        const double step_norm2 = L.step_norm2, xnorm2_new = L.xnorm2;     // of the trial point (group-uniform LDS reads)

        PROF_MARK(3);                                 // 3: post-sweep reductions
        // ======================= C: transitions (no cross-lane operations below) =======================
        bool decide = false;
        double cost_cand = 0.0;
        if (pe && sl == 0) atomicAdd(&L.exec_passes, 1);          // statistics live in LDS: fire-and-forget ds_add
        if (phase == PH_EVAL_INIT) {
            cost = cost_e; gi = gnew; gmax = gmax_new;
            inv_scale = is_row ? 1.0 + sqrt(L.A[row * LD + row]) : 1.0;
            scale = 1.0 / inv_scale;                                                // jacobi scaling, once
            a_dirty = false; phase = PH_SOLVE;
        } else if (phase == PH_REEVAL) {
            a_dirty = false; phase = PH_SOLVE;
        } else if (phase == PH_EVAL_LS && cost_only) {
            // Speculative cost-only round: without the jacobian only the two "converged, candidate discarded"
            // outcomes can be taken.  Anything else (accept, reject, Armijo contraction) repeats the
            // evaluation of the same point in full next round; nothing is counted for this one.
            cost_only = false;
            const bool armijo = isfinite(cost_e) && !(cost_e > cost + kLsSufficientDecrease * g_dot_delta * alpha);
            const double ptol = kParameterTol * (x_norm + kParameterTol);
            const bool stop = armijo && (step_norm2 <= ptol * ptol || fabs(cost - cost_e) <= kFunctionTol * cost);
            if (stop) {
                if (sl == 0) atomicAdd(&L.n_ls_evals, 1);
                decide = true; cost_cand = cost_e;
            }
        } else if (phase == PH_EVAL_LS) {
            if (sl == 0) atomicAdd(&L.n_ls_evals, 1);
            const bool value_valid = isfinite(cost_e);
            if (value_valid && !(cost_e > cost + kLsSufficientDecrease * g_dot_delta * alpha)) {
                decide = true; cost_cand = cost_e;                                   // candidate == this sample
            } else {
                // rare: contraction.  Every lane of the group reads the same bookkeeping from LDS,
                // computes the same next step; lane 0 writes the bookkeeping back.
                // (cross-lane work inside this group-uniform branch: whole groups take it, the butterflies stay inside the group)
                const double gdc = group_sum<S>(own ? delta * gnew : 0.0);
                const double dir_max = group_max<S>(fabs(delta));
                LsSample initial{0.0, cost, g_dot_delta, true, true}, previous, current;
                const int pf = L.ls_prev_flags;
                previous.x = L.ls_prev_x; previous.value = L.ls_prev_value; previous.gradient = L.ls_prev_gradient;
                previous.value_valid = (pf & 1) != 0; previous.gradient_valid = (pf & 2) != 0;
                current.x = alpha; current.value = cost_e; current.value_valid = value_valid;
                current.gradient = value_valid ? gdc : 0.0;
                current.gradient_valid = value_valid && isfinite(gdc);
                int ls_iter = L.ls_iter;
                const double nstep = ls_next_step(initial, previous, current, dir_max, ls_iter);
                wave_lds_sync();
                if (sl == 0) {
                    L.ls_iter = ls_iter;
                    L.ls_prev_x = current.x; L.ls_prev_value = current.value; L.ls_prev_gradient = current.gradient;
                    L.ls_prev_flags = (current.value_valid ? 1 : 0) | (current.gradient_valid ? 2 : 0);
                }
                if (nstep < 0.0) {                                                  // search failed: full step
                    xt = clampb(__dadd_rn(xi, delta));
                    phase = PH_EVAL_CAND;
                } else {
                    alpha = nstep;
                    xt = clampb(__dadd_rn(xi, __dmul_rn(alpha, delta)));
                }
                if (own) L.x[row] = xt;
                const double sn2 = group_sum<S>(own ? (xi - xt) * (xi - xt) : 0.0), xn2 = group_sum<S>(own ? xt * xt : 0.0);
                if (sl == 0) { L.step_norm2 = sn2; L.xnorm2 = xn2; }
            }
        } else if (phase == PH_EVAL_CAND) {
            decide = true; cost_cand = isfinite(cost_e) ? cost_e : DBL_MAX;
        }
        if (decide) {
            if (sl == 0) atomicAdd(&L.n_cand, 1);
            const double cost_change = cost - cost_cand;
            const double ptol = kParameterTol * (x_norm + kParameterTol);
            if (step_norm2 <= ptol * ptol) phase = PH_DONE;                                      // candidate discarded (|step| <= tol, squared)
            else if (fabs(cost_change) <= kFunctionTol * cost) phase = PH_DONE;                  // candidate discarded
            else {
                const double rel = cost_change * fast_rcp(model_cost_change);       // (model_cost_change > 0: the step was valid)
                if (rel > kMinRelDecrease) {
                    xi = xt; x_norm = fast_sqrt(xnorm2_new); cost = cost_cand; gi = gnew; gmax = gmax_new;
                    step_successful = true;
                    if (sl == 0) atomicAdd(&L.n_successful, 1);
                    const double t = 2.0 * rel - 1.0;
                    radius = fmin(kMaxRadius, radius * fast_rcp(fmax(1.0 / 3.0, 1.0 - t * t * t)));
                    n_reject = 0; reuse_diagonal = false; a_dirty = false;
                } else {
                    radius = ldexp(radius, -(1 + n_reject)); ++n_reject;             // StepRejected: radius / 2^(1 + n_reject)
                    a_dirty = true;
                }
                phase = PH_SOLVE;
            }
        }
        PROF_MARK(4);                                 // 4: transitions
    }
    PROF_FLUSH();

    if (have) {
        if (own)       // every variable is written by every solve (a failed solve leaves 0, solve.cc:609-612)
            a.positions[2 * (size_t)a.node_ids[d.node_off + (row >> 1)] + (row & 1)] = term != LFR_TERM_FAILURE ? xi : 0.0;
        if (sl == 0) {
            CompInfoDev inf;
            inf.iterations = iteration; inf.termination = term; inf.n_successful = L.n_successful;
            inf.n_ls_evals = L.n_ls_evals; inf.n_cand_evals = L.n_cand; inf.exec_passes = L.exec_passes;
            inf.final_cost = cost;
#if defined(LFR_PROFILE_WGTIME) && LFR_PROFILE_WGTIME == 3      // diagnostic builds: when and where the wave ran (scripts/c4_timeline.py)
            inf.final_cost = (double)wave_r0_;
            inf.n_ls_evals = inf.iterations;
            inf.iterations = (int)(wall_clock64() - wave_r0_);
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
            inf.termination = (int)((hw & 0xffffu) | ((xcc & 0xfu) << 16));     // wave [3:0] simd [5:4] cu [11:8] sh [12] se [15:13] xcc [19:16]
#endif
            a.infos[ci] = inf;
        }
    }
}

constexpr size_t kPackedLdsBytes = kPackedWaves * (sizeof(GroupLds<16>) * 4 > 2 * sizeof(GroupLds<32>) ? sizeof(GroupLds<16>) * 4 : 2 * sizeof(GroupLds<32>));
static_assert(kPackedLdsBytes >= kPackedWaves * 8 * sizeof(GroupLds<8>) && kPackedLdsBytes >= kPackedWaves * 2 * sizeof(GroupLds<16>), "LDS budget");

// one class per launch (diagnostics: LFR_SERIAL_CLASSES=1 gives per-class timings)
template <int NV, int LPR, int EPL, bool FUSED>
__global__ __launch_bounds__(64 * kPackedWaves, LFR_GROUP_WAVES) void solve_group_kernel(const KernelArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kPackedWaves * (64 / (NV * LPR)) * sizeof(GroupLds<NV>)];
    solve_group_body<NV, LPR, EPL, FUSED>(a, (int)blockIdx.x, lds_raw);
}

// All packed classes in ONE launch: blocks [blk_begin[i], blk_begin[i+1]) belong to class i, the
// long-running one-component-per-wave classes first so their tail overlaps the bulk of the small
// classes (workgroups are dispatched in index order).  The classes are compiled into one kernel;
// its register/LDS budget is the maximum over the classes (all are built for 2 waves per SIMD).
struct PackedRanges {
    int blk_begin[6];          // G64_4, G64_2, G32, G16, G8 in dispatch order
    int desc_begin[5], desc_end[5];
};
template <bool FUSED>
__global__ __launch_bounds__(64 * kPackedWaves, LFR_GROUP_WAVES) void solve_packed_kernel(KernelArgs a, const PackedRanges r) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kPackedLdsBytes];
    // (one workgroup per block, dealt by the hardware: the wave timeline of config 4 shows the chip full from the first
    // microsecond, 95 % of the 2048 wave slots busy in the steady state and a 15-us tail; resident waves pulling blocks from an
    // atomic queue - what pays for the 160-KB workgroups of the long-track classes - cost this kernel 76 spilled VGPRs
    // and 38 %: 0.650 against 0.470 ms)
    const int b = (int)blockIdx.x;
    if (b < r.blk_begin[1]) {
        a.desc_begin = r.desc_begin[0]; a.desc_end = r.desc_end[0]; a.cls = lfr::KC_G64_4;
        solve_group_body<32, 2, 5, FUSED>(a, b - r.blk_begin[0], lds_raw);
    } else if (b < r.blk_begin[2]) {
        a.desc_begin = r.desc_begin[1]; a.desc_end = r.desc_end[1]; a.cls = lfr::KC_G64_2;
        solve_group_body<32, 1, 6, FUSED>(a, b - r.blk_begin[1], lds_raw);
    } else if (b < r.blk_begin[3]) {
        a.desc_begin = r.desc_begin[2]; a.desc_end = r.desc_end[2]; a.cls = lfr::KC_G32;
        // (KC_G32, formerly <16,2,3>, is retired: <16,1,6> takes every <=16-row component up to 96 edges)
    } else if (b < r.blk_begin[4]) {
        a.desc_begin = r.desc_begin[3]; a.desc_end = r.desc_end[3]; a.cls = lfr::KC_G16;
        solve_group_body<16, 1, 6, FUSED>(a, b - r.blk_begin[3], lds_raw);
    } else {
        a.desc_begin = r.desc_begin[4]; a.desc_end = r.desc_end[4]; a.cls = lfr::KC_G8;
        solve_group_body<8, 1, 3, FUSED>(a, b - r.blk_begin[4], lds_raw);
    }
}

// =============================================================================================
// workgroup-per-component kernel (packed lower-triangular normal matrix in LDS or HBM)
// =============================================================================================
// One workgroup per component; its size is a template parameter.  The launch is wait-bound (72 % of the wave cycles on config
// 5), so what counts is how many components share a CU, and that is set by LDS and by the 8 waves of 256 VGPRs a CU
// holds: 128 threads x 4 workgroups (<= 88 rows), 256 x 2 (<= 130 rows), 512 x 1 (<= 192 rows: the whole CU; with the
// barrier-per-column phases gone the second wave per SIMD pays - 10.15 ms against 10.8 ms with 256), 256 for the HBM-matrix variant.

// Accumulate into a matrix entry its row's thread owns.  A read-modify-write in program order made every edge of the walk wait
// for an LDS round trip (a row may hit the same entry twice, so the compiler cannot overlap them); the no-return atomic is
// fire-and-forget, and the adds of one thread to one address still apply in issue order: the sum stays bitwise reproducible.
__device__ __forceinline__ void mat_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Packed lower triangle: element (i, j), j <= i.  Rows are < 2^16 (32767 nodes per component), so i (i + 1) fits 32 bits and the
// product is a full-rate 24-bit multiply (v_mul_u32_u24) instead of the quarter-rate v_mul_lo_u32: the index arithmetic of the
// factorization's tile loop was costing more issue cycles than its MFMAs.
__device__ __forceinline__ uint32_t tri(int i, int j) { return (__umul24((unsigned)i, (unsigned)i + 1u) >> 1) + (unsigned)j; }

struct BlockShared {
    double red[16];
    double bcast[4];
    int flag;
    int ready;               // factor_lds: last panel whose diagonal block wave 0 has factored and published
    int lead;                // factor_lds: column blocks c for which wave 1 has substituted tile (c+2, c) and applied panel c to tile (c+2, c+1)
    int lead_t;              // factor_lds: panels whose update wave 1 has applied to the next two tiles wave 0 will take
    int wbar;                // factor_lds: arrivals at the worker waves' own barrier
    int ovf;                 // fused sweep: a term left the fixed-point range of the in-edge accumulators
};

using f64x4 = __attribute__((ext_vector_type(4))) double;

// A value every lane of the wave holds (a workgroup reduction read back from LDS, the scalars of the LM loop derived from such): moved
// to SGPRs, bit for bit.  The compiler keeps what the VALU produced in VGPRs whether or not the lanes agree; the state of the
// trust-region loop - two dozen doubles that live across the inlined sweeps - was what spilled to scratch around every sweep.
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int kBlockThreads>
__device__ __forceinline__ double block_sum(double v, BlockShared &sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlockThreads / 64; ++w) s += sh.red[w];
    return uni(s);
}
template <int kBlockThreads>
__device__ __forceinline__ double block_max(double v, BlockShared &sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = sh.red[0];
#pragma unroll
    for (int w = 1; w < kBlockThreads / 64; ++w) s = fmax(s, sh.red[w]);
    return uni(s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Blocked LDL^T of the damped normal matrix in LDS (packed lower triangle, the right-hand side riding as row n), in place.
// Column k keeps the UNSCALED entries a_ik (L_ik = a_ik / d_k) of the rows BELOW its diagonal tile, 1/d_k goes to vinv[k]; the strictly lower part
// of a diagonal tile holds, transposed, M = (I + D^-1 U^T)^-1 - the tile's substitution as a matrix (round 5, factor_diag) - and its diagonal
// is left as assembled (nobody reads either again).
//
// Right-looking over panels of 16 columns (round 2: 8 columns, three barriers, the diagonal block on one thread).  The serial pieces
// of a panel - the diagonal tile's elimination and the substitution of the rows below it, 16 dependent steps of ~300 cycles each (the
// kernel issues one VALU instruction per ~4.8 cycles and wave, scripts/probes/diag16_probe.hip) - are kept off each other's path:
//   * WAVE 0 owns the diagonal tiles and runs AHEAD of the other waves (no barrier inside the factorization).  Step k: apply panel k to
//     the diagonal tile (k+1, k+1) (fp64 MFMA, K = 16; rows k+1 of panel k are its own), then factor it with lane = row in lanes 0-15
//     while lanes 16-31 carry the rows of the tile below, (k+2, k+1), through the same steps - a_ic -= (a_ik / d_k) a_ck with a_ck read
//     from lane c: the diagonal tile's elimination IS their substitution - so the rows the NEXT step's update needs come out final with
//     the diagonal tile, not ~3000 cycles after it - and lanes 32-47 the rows of the identity, which come out as M.  Then it stores the tile below, M
//     and 1/d and publishes `ready`.
//   * WAVE 1 feeds wave 0: in every phase it first applies the panel to the two tiles wave 0 takes next (`lead_t`), and as soon as the
//     diagonal tile is out it substitutes tile (k+3, k+1), applies panel k+1 to tile (k+3, k+2) - the tile wave 0 carries next - and
//     says so (`lead`).  With three or more workers it takes no other trailing tile.
//   * the OTHER waves apply panel k to the tiles right of column block k+1 (two 16x16 tiles in flight per wave, 4 + 4
//     v_mfma_f64_16x16x4_f64), then to their tiles of column block k+1, wait for `ready`, substitute their tiles (tile <- tile M: four MFMAs;
//     until round 5 fifteen dependent steps with lane = row), and meet at a barrier of their own (an LDS counter; wave 0 is not part of it).
// Cycles per panel of the 190-row class: 14.3 k with one workgroup barrier per panel and the rows' substitution after the diagonal
// tile (round 3, first half), 12.6 k now (wave 0: update 1.8 k, waiting for wave 1 3.7 k, elimination + loads/stores 7.0 k) - wave 1's
// substitution of ONE tile plus its update is what wave 0 still waits for; the other workers are as loaded as wave 0 (11-12 k per phase).
// (Wave 1 running its OWN copy of the elimination with its tiles carried along - so that its first tile is final when wave 0 publishes -
// was built too: wave 0 waits 2.4 k instead of 3.7 k per panel, and its elimination takes 8.6 k instead of 7.0 k; no gain, removed.)
// (The pivot column through LDS instead of v_readlane - 3.8 k against 4.8 k cycles per tile in the isolated probe - DOUBLES the elimination here:
// 14 k cycles per panel with seven other waves loading and storing tiles through the same LDS.)
// Round 5 (M): wave 0 waits for wave 1 17.9 k instead of 42.9 k cycles per 190-row factorization and its update of the diagonal tile takes 17 k instead
// of 25 k (-DLFR_PROFILE_FACTOR), but carrying and storing M costs its elimination + stores 45 k: the launch gains 2 % (4.35 against 4.44 ms per
// config-5 solve), the back substitution 7 %.  The stores must stay free of per-store lane-varying branches (see factor_diag).
// scripts/emul_factor_v2.py is a lane-level CPU model of the FIRST round-3 schedule (one barrier per panel; random wave order + a race
// detector); scripts/factor_lds_sync_model.py models the hand-off words of this one at tile level (tests/test_factor_lds_sync_model.py).
// A non-positive pivot only raises sh.flag: the phases run to the end on whatever values there are (no data-dependent
// exit, so no wave can miss a barrier), the caller rejects the step.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef LFR_PROFILE_FACTOR      // diagnostic builds: where the waves of a factorization spend their cycles (waves 0 and 1, six slots each)
#define FPROF_DECL unsigned long long ft_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long ft0_ = __builtin_amdgcn_s_memtime();
#define FPROF_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ft_[i] += t_ - ft0_; ft0_ = t_; } while (0)
#define FPROF_RESET() do { ft0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define FPROF_FLUSH() do { if (lane == 0 && wave < 2 && fprof) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&fprof[wave * 8 + i_], ft_[i_]); atomicAdd(&fprof[wave * 8 + 7], 1ull); } } while (0)
#else
#define FPROF_DECL unsigned long long ft_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long ft0_ = 0; (void)ft_; (void)ft0_;
#define FPROF_MARK(i)
#define FPROF_RESET()
#define FPROF_FLUSH()
#endif
// (A substitution that followed the diagonal tile column by column - a poll per group of g columns inside the unrolled steps - cost registers:
// g = 1: 414 spilled VGPRs, g = 4: 396, g = 8: 297, one wait up front: 67.  With M there are no steps to follow.)
template <int kBlockThreads>
__device__ __forceinline__ void factor_lds(double *Mat, double *vinv, const int n, BlockShared &sh, unsigned long long *fprof, unsigned int *spin_timeouts) {
    constexpr int kWaves = kBlockThreads / 64;
    static_assert(kWaves >= 2, "factor_lds needs a wave beside the one that factors the diagonal blocks");
    constexpr int kWorkers = kWaves - 1;        // waves 1.. : everything but the diagonal blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (an SGPR: tile indices and loops are SALU)
    const int r16 = lane & 15, kq = lane >> 4;
    const int n1 = n + 1;                       // rows carried (row n = right-hand side)
    const int P = (n + 15) >> 4;                // panels
    const int RT = (n1 + 15) >> 4;              // 16-row tiles
    FPROF_DECL
    volatile int *ready = &sh.ready;            // 16 * panel + 16: that panel's diagonal tile (the tile below it, M, 1 / d) is published
    const __attribute__((address_space(3))) int *ready_lds = (const __attribute__((address_space(3))) int *)&sh.ready;

    // diagonal tile at kb (up to date in LDS): lane = row in lanes 0-15.  Lanes 16-31 hold the rows of the tile BELOW it, (panel+1, panel),
    // up to date as well: the elimination steps that factor the diagonal tile are exactly the rows' substitution for them
    // (a_ic -= (a_ik / d_k) a_ck with a_ck read from lane c), so the tile the NEXT diagonal tile's update needs comes out final together
    // with the diagonal tile instead of ~3000 cycles after it.  Lanes 32-47 carry the rows of the IDENTITY through the same steps: what
    // comes out is the substitution as a matrix, M = (I + D^-1 U^T)^-1 (unit upper triangular; U = the tile's strictly lower part, unscaled):
    // the substitution of any row vector r is r M, and the tile's part of the back substitution is y = M (D^-1 z).  M[i][c], c > i, is
    // stored where U[c][i] was (nothing reads U again): the other waves substitute their tiles with four MFMAs each instead of fifteen
    // dependent steps (finish_rows), the back substitution loses its dependent chain inside a tile.  Everything is stored, then `ready` says so.
    auto factor_diag = [&](const int kb, const int panel) {
        const int nbp = min(16, n - kb);                            // pivots; a row beyond them is the right-hand side or padding
        const int row = kb + 16 * kq + r16;                         // group 0: the diagonal tile, group 1: the tile below, group 2: identity
        const bool rv = row < n1 && kq <= 1;
        const uint32_t base = tri(rv ? row : kb, kb);
        double a[16];
        const bool ident = kq == 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // (one address register + immediate offsets: a row of group 0 reads past its diagonal - the next rows' entries, at the very end
            // of the matrix up to 15 doubles past it, still inside the LDS - and drops what it read; clamping the column per lane cost
            // sixteen address registers, spilled)
            const double v = Mat[base + j];
            a[j] = kq == 0 ? ((rv && j <= r16) ? v : (j == r16 ? 1.0 : 0.0)) : (rv ? v : 0.0);
            a[j] = (ident && j == r16) ? 1.0 : a[j];
        }
        bool bad = false;
        double my_inv = 0.0;                                        // lane k keeps 1 / d_k: one store after the loop instead of a masked one per step
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < nbp) {                                          // wave-uniform
                const double dk = readlane_f64(a[k], k);
                bad = bad || !(dk > 0.0);
                const double ik = fast_rcp(dk);
                my_inv = lane == k ? ik : my_inv;
                const double lik = a[k] * ik;                       // rows below the pivot (the others only touch their padding)
#pragma unroll
                for (int j = k + 1; j < 16; ++j) a[j] = fma(-lik, readlane_f64(a[k], j), a[j]);
            }
        }
        if (lane < nbp) vinv[kb + lane] = my_inv;
        FPROF_MARK(0);                                              // 0 (wave 0): loads + elimination; the stores report in slot 1
        // Stores, branch-free (as lane-varying `if`s they became one basic block each, every one reloading its spilled address from scratch
        // memory, on the factorization's critical path: 6.3 against 4.45 ms per config-5 solve): a lane without an entry writes to scratch
        // words - sh.red, which every reduction rewrites behind a barrier before reading it.
        // group 1, the tile below: columns 1-15 (column 0 is unchanged).  Group 0 stores nothing: the pivots are in vinv, the tile's strictly
        // lower part gives way to M, and nobody reads the diagonal again ...
        if (kq == 1 && rv) {                                        // (ONE lane-varying branch around fifteen plain stores)
#pragma unroll
            for (int c = 1; c < 16; ++c) Mat[base + c] = a[c];
        }
        // ... except the right-hand-side row when it lies in this tile (the last panel): its entries are the solve's w = L^-1 g
        if (kb + 16 > n) {                                          // wave-uniform
            if (kq == 0 && row == n) {
#pragma unroll
                for (int c = 0; c < 16; ++c) if (c < nbp) Mat[base + c] = a[c];
            }
        }
        // group 2: row i = r16 of M, entries c = i + 1 .. nbp - 1, to (row kb + c, column kb + i)
        if (ident) {                                                // (sixteen lanes, sixteen scratch words: no two lanes on one address)
            double *const dummy = &sh.red[r16];
            const unsigned span = (unsigned)max(nbp - r16 - 1, 0);  // entries of this lane's row (one lane-varying test per store: `c < nbp` on
            uint32_t at = tri(kb, kb) + (uint32_t)r16;              //  its own became a scalar branch per store)
#pragma unroll
            for (int c = 1; c < 16; ++c) {
                at += (uint32_t)(kb + c);                           // tri(kb + c, kb) + r16
                double *const dst = (unsigned)(c - r16 - 1) < span ? Mat + at : dummy;
                *dst = a[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (compiler only: the counter goes out after the data; the LDS keeps a wave's order)
        if (lane == 0) *ready = 16 * panel + 16;
        if (bad && lane == 0) sh.flag = 1;
    };
    // the tiles (R0, panel) and (R1, panel) (R1 = -1: none), rows 16 R .. 16 R + 15, substituted against the diagonal tile at kb (a full
    // panel): tile <- tile M on the matrix cores, M read from the diagonal tile's strictly lower part once wave 0 has published it
    // (lane l feeds A[l&15][4kk + (l>>4)] = tile[r16][4kk + kq] and B[4kk + (l>>4)][l&15] = M[4kk + kq][r16], holds D[(l>>4) + 4r][l&15]).
    // Two accumulator chains per tile: the first pair of wave 1 is on the factorization's critical path.
    auto finish_rows = [&](const int kb, const int panel, const int R0, const int R1) {
        const bool two = R1 >= 0;
        const int ia0 = min(16 * R0 + r16, n1 - 1), ia1 = min(16 * (two ? R1 : R0) + r16, n1 - 1);      // (clamped: a row beyond the matrix is not stored)
        const uint32_t oa0 = tri(ia0, kb + kq), oa1 = tri(ia1, kb + kq);
        double a0[4], a1[4], bm[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { a0[kk] = Mat[oa0 + 4 * kk]; a1[kk] = Mat[oa1 + 4 * kk]; }
        {   // spin until wave 0 has published the diagonal tile (one opaque instruction sequence: a C++ loop here made the compiler keep
            // the operands of both tiles across a call-like region)
            int tmp, have;
            const int need = 16 * panel + 16;
            asm volatile("LFR_POLL_%=:\n\t"
                         "ds_read_b32 %0, %2\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_readfirstlane_b32 %1, %0\n\t"
                         "s_cmp_ge_i32 %1, %3\n\t"
                         "s_cbranch_scc1 LFR_POLLED_%=\n\t"
                         "s_sleep 1\n\t"
                         "s_branch LFR_POLL_%=\n"
                         "LFR_POLLED_%=:"
                         : "=&v"(tmp), "=&s"(have)
                         : "v"((uint32_t)(uintptr_t)ready_lds), "s"(need)
                         : "memory", "scc");
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int i = 4 * kk + kq;                              // row of M this lane feeds, column r16
            const double v = Mat[tri(kb + max(r16, i), kb + min(r16, i))];   // (i < r16: M[i][r16]; otherwise a valid address, value replaced)
            bm[kk] = i < r16 ? v : (i == r16 ? 1.0 : 0.0);
        }
        f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, d0 = c0, d1 = c0;
        if (two) {                                                  // wave-uniform
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], bm[0], c0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[2], bm[2], d0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], bm[0], c1, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[2], bm[2], d1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], bm[1], c0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[3], bm[3], d0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], bm[1], c1, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[3], bm[3], d1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], bm[0], c0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[2], bm[2], d0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], bm[1], c0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[3], bm[3], d0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row0 = 16 * R0 + kq + 4 * r, row1 = 16 * R1 + kq + 4 * r;
            if (r16 > 0 && row0 < n1) Mat[tri(row0, kb + r16)] = c0[r] + d0[r];                     // (column 0 of M is e_0: unchanged)
            if (two && r16 > 0 && row1 < n1) Mat[tri(row1, kb + r16)] = c1[r] + d1[r];
        }
    };
    // tile (R, J) -= (rows R of panel kb) (rows J of panel kb / d)^T on the matrix cores: lane l feeds A[l&15][4kk + (l>>4)] and
    // B[4kk + (l>>4)][l&15] of chunk kk and holds D[(l>>4) + 4r][l&15], r = 0..3
    struct Tile { double a[4], b[4]; f64x4 c; bool ok[4]; int row0, jb; };
    double ninv[4];
    auto load_tile = [&](const int kb, const int R, const int J, Tile &T) {        // any tile: clamped loads, values selected afterwards
        const int ia = 16 * R + r16, jb = 16 * J + r16, iac = min(ia, n1 - 1), jbc = min(jb, n1 - 1);
        T.row0 = 16 * R + kq; T.jb = jb;
        const uint32_t oa = tri(iac, kb + kq), ob = tri(jbc, kb + kq);
        double av[4], bv[4], cv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { av[kk] = Mat[oa + 4 * kk]; bv[kk] = Mat[ob + 4 * kk]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = T.row0 + 4 * r, rc = min(row, n1 - 1);
            T.ok[r] = row < n1 && jb <= row;
            cv[r] = Mat[tri(rc, min(jbc, rc))];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { T.a[kk] = ia < n1 ? av[kk] : 0.0; T.b[kk] = jb < n1 ? bv[kk] * ninv[kk] : 0.0; }
#pragma unroll
        for (int r = 0; r < 4; ++r) T.c[r] = T.ok[r] ? cv[r] : 0.0;
    };
    auto load_full = [&](const int kb, const int R, const int J, Tile &T) {        // every row < n1, strictly below the diagonal
        const int ia = 16 * R + r16, jb = 16 * J + r16;
        T.row0 = 16 * R + kq; T.jb = jb;
        const uint32_t oa = tri(ia, kb + kq), ob = tri(jb, kb + kq);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { T.a[kk] = Mat[oa + 4 * kk]; T.b[kk] = Mat[ob + 4 * kk] * ninv[kk]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { T.ok[r] = true; T.c[r] = Mat[tri(T.row0 + 4 * r, jb)]; }
    };
    auto store_tile = [&](const Tile &T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (T.ok[r]) Mat[tri(T.row0 + 4 * r, T.jb)] = T.c[r];
    };
    auto update_pair = [&](const int kb, const int Ra, const int Ja, const bool two, const int Rb, const int Jb) {
        Tile t0, t1;
        if (two && Ja < Ra && Jb < Rb && 16 * max(Ra, Rb) + 15 < n1) {
            load_full(kb, Ra, Ja, t0);
            load_full(kb, Rb, Jb, t1);
        } else {
            load_tile(kb, Ra, Ja, t0);
            load_tile(kb, two ? Rb : Ra, two ? Jb : Ja, t1);       // (no second tile: the first again, not stored)
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            t0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.a[kk], t0.b[kk], t0.c, 0, 0, 0);
            t1.c = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.a[kk], t1.b[kk], t1.c, 0, 0, 0);
        }
        store_tile(t0);
        if (two) store_tile(t1);
    };
    // one tile, two accumulator chains of two MFMAs (on the critical path: wave 1's hand-off tile, wave 0's diagonal tile)
    auto update_one = [&](const int kb, const int R, const int J) {
        Tile t;
        load_tile(kb, R, J, t);
        f64x4 c2 = {0.0, 0.0, 0.0, 0.0};
        t.c = __builtin_amdgcn_mfma_f64_16x16x4f64(t.a[0], t.b[0], t.c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t.a[2], t.b[2], c2, 0, 0, 0);
        t.c = __builtin_amdgcn_mfma_f64_16x16x4f64(t.a[1], t.b[1], t.c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t.a[3], t.b[3], c2, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) t.c[r] += c2[r];
        store_tile(t);
    };
    // the tiles (R, kcol), R >= kcol + 2, of this worker wave, two at a time (tile (kcol + 1, kcol) is wave 0's: substituted inside the
    // diagonal tile's elimination): update by panel kb (kb < 0: none, column block 0), then the rows' substitution against the diagonal
    // tile of panel `kcol`.  Wave 1's first tile is (kcol + 2, kcol): with it final, wave 1 applies panel kcol to tile (kcol + 2, kcol + 1) -
    // the tile wave 0 carries through the NEXT diagonal tile's elimination - and tells wave 0.
    auto column_tiles = [&](const int kb, const int kcol) {
        bool told = false;
        for (int R0 = kcol + 1 + wave; R0 < RT; R0 += 2 * kWorkers) {             // (wave >= 1: R0 starts at kcol + 2)
            const int R1 = R0 + kWorkers < RT ? R0 + kWorkers : -1;
            if (kb >= 0) {
                update_pair(kb, R0, kcol, R1 >= 0, R1, kcol);
                wave_lds_sync();
            }
            FPROF_MARK(3);                            // 3: tiles of the next column block (update)
            if (wave == 1 && !told) {                 // the hand-off tile alone first: wave 0 is waiting for it
                told = true;
                finish_rows(16 * kcol, kcol, R0, -1);
                if (kcol + 1 < P) {                   // (a next diagonal tile exists; R0 = kcol + 2 < RT here)
                    double keep[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) { keep[kk] = ninv[kk]; ninv[kk] = -vinv[16 * kcol + 4 * kk + kq]; }
                    wave_lds_sync();
                    update_one(16 * kcol, kcol + 2, kcol + 1);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) ninv[kk] = keep[kk];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&sh.lead, kcol + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (R1 >= 0) finish_rows(16 * kcol, kcol, R1, -1);
            } else {
                finish_rows(16 * kcol, kcol, R0, R1);
            }
            FPROF_MARK(4);                            // 4: rows of the next column block (substitution behind the diagonal block)
        }
        if (wave == 1 && !told && lane == 0) __hip_atomic_store(&sh.lead, kcol + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (no tile below: nothing to wait for)
    };
    // bounded spin on an LDS word (a miscount must not hang the GPU: the step is rejected instead)
    auto spin_until = [&](int *word, const int need) {
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { if (lane == 0) { sh.flag = 1; atomicAdd(spin_timeouts, 1u); } break; }     // (counted: lfr_batch_spin_timeouts; the GPU tests assert 0)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // the worker waves' own barrier (wave 0 is busy with a diagonal tile and does not take part)
    auto worker_barrier = [&](const int target) {
        if constexpr (kWorkers > 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&sh.wbar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            spin_until(&sh.wbar, target);
        }
    };

    if (tid == 0) { sh.flag = 0; sh.ready = -1; sh.lead = 0; sh.lead_t = 0; sh.wbar = 0; }
    __syncthreads();
    FPROF_RESET();
    // ---- column block 0 ----
    if (wave == 0) { factor_diag(0, 0); FPROF_MARK(1); }                            // 0: diagonal blocks (wave 0)
    else { column_tiles(-1, 0); worker_barrier(kWorkers); }
    FPROF_MARK(5);                                    // 5: barrier
    // ---- phases: panel k updates what is right of it, column block k+1 comes out final ----
    for (int k = 0; k + 1 < P; ++k) {
        const int kb = 16 * k;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ninv[kk] = -vinv[kb + 4 * kk + kq];
        if (wave == 0) {
            // Wave 0 runs AHEAD of the others: no barrier at the end of a phase.  The next diagonal tile needs panel k - its rows k+1 are
            // wave 0's own (substituted inside the previous elimination) - on top of the earlier panels (wave 1's first trailing pair of
            // the phase before: lead_t); the tile below it, carried through the elimination, needs wave 1's special update (lead).
            spin_until(&sh.lead_t, k);
            update_one(kb, k + 1, k + 1);
            FPROF_MARK(3);                            // 3 (wave 0): the diagonal tile's update
            spin_until(&sh.lead, k + 1);
            FPROF_MARK(2);                            // 2: wave 0 waiting for wave 1
            wave_lds_sync();
            factor_diag(kb + 16, k + 1);
            FPROF_MARK(1);
        } else {
            // tiles (R, J), k + 2 <= J <= R < RT, J a column block that exists: t-th tile of the row-major lower triangle.  Tiles 0 and 1,
            // (k+2, k+2) and (k+3, k+2), are what wave 0 takes next: wave 1 does them first and says so; the others are dealt round-robin.
            const int m = RT - (k + 2);
            int T = m > 0 ? (m * (m + 1)) >> 1 : 0;
            if (T > 0 && RT > P) --T;                                 // (the last tile would be columns >= n of the right-hand-side row)
            if (wave == 1) {
                if (T > 0) update_pair(kb, k + 2, k + 2, T > 1, k + 3, k + 2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&sh.lead_t, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const int Tw = T > 2 ? T - 2 : 0;
            int I = 0, J = 0, tcur = 0;
            auto seek = [&](const int tt) { J += tt - tcur; tcur = tt; while (J > I) { J -= I + 1; ++I; } };
            // (wave 1 has the lead tiles, the special update and wave 0 waiting for it: with three or more workers it takes no other
            // trailing tile)
            constexpr int kDeal = kWorkers >= 3 ? kWorkers - 1 : kWorkers;
            const int first = kWorkers >= 3 ? wave - 2 : wave - 1;
            for (int u = first; u >= 0 && u < Tw; u += 2 * kDeal) {   // wave-uniform
                seek(u + 2);
                const int Ra = k + 2 + I, Ja = k + 2 + J;
                const bool two = u + kDeal < Tw;
                int Rb = Ra, Jb = Ja;
                if (two) { seek(u + kDeal + 2); Rb = k + 2 + I; Jb = k + 2 + J; }
                update_pair(kb, Ra, Ja, two, Rb, Jb);
            }
            FPROF_MARK(1);                            // 1: trailing tiles
            column_tiles(kb, k + 1);
            worker_barrier(kWorkers * (k + 2));       // column block k+1 is final for every worker
        }
        FPROF_MARK(5);
    }
    __syncthreads();
    FPROF_FLUSH();
}

// vectors live in LDS for both variants: 8 vectors of n doubles
// (two waves per SIMD = 256 registers each, VGPRs + the MFMA accumulators: what the 4 / 2 workgroups per CU of the two smaller
// LDS classes need; without the attribute the allocator takes 264)
template <int kBlockThreads>
__device__ __forceinline__ void solve_component(const KernelArgs &a, const int max_rows, const int ci, double *dyn, BlockShared &sh) {
#ifdef LFR_PROFILE_WGTIME
    const unsigned long long wg_t0_ = __builtin_amdgcn_s_memtime();
    const unsigned long long wg_r0_ = wall_clock64();          // 100 MHz, the same counter on every XCD
#endif
    const int tid = threadIdx.x;
    const CompDesc d = a.descs[ci];
    const int n_var = d.n_var, n = 2 * n_var, E = (int)d.n_edges;
    const int tv = a.tukey_variant;
    const EdgeRec *edges = a.edges + d.edge_off;

    // LDS variant: vectors + packed matrix in dynamic LDS.  HBM variant: packed matrix, then the
    // vectors, in this component's workspace (L2-resident); no dynamic LDS at all, so the row count
    // is only bounded by the 32767-node limit of the batch format.
    double *Mat = dyn + 2 * (size_t)(max_rows + 2) + 7 * (size_t)max_rows;
    double *vx = dyn;                 // x (n + 2, zero slot at n)
    double *vxc = vx + max_rows + 2;  // trial point
    double *vg = vxc + max_rows + 2;  // gradient at x
    double *vgn = vg + max_rows;      // gradient at trial point
    double *vscale = vgn + max_rows;
    double *vdiag = vscale + max_rows;
    // rhs -> step.  LDS variant: row n of the matrix - the right-hand side rides through the factorization as one more row
    // (LDL^T of [[A, g], [g^T, .]]: the unscaled entries of that row come out as L^-1 g), so the forward substitution, n
    // dependent steps per solve, is gone; the (n, n) entry is never used.
    double *vstep = Mat + tri(n, 0);
    double *vD = vdiag + max_rows;
    const int n1 = n + 1;             // rows the factorization carries
    double *vadiag = vD + max_rows;   // diagonal of unscaled J^T J at x
    double *vdelta = vadiag + max_rows;

    for (int i = tid; i < n + 2; i += kBlockThreads) { vx[i] = 0.0; vxc[i] = 0.0; }
    __syncthreads();

    // One sweep over the edges at xv: returns the cost, fills gout = J^T r and (want_matrix) Mat =
    // unscaled J^T J.  Deterministic owner-computes assembly: phase 1 evaluates every edge (one
    // thread per edge, 64 B of corrected jacobian/residual to the scratch); phase 2 gives every row
    // of the normal matrix to ONE thread, which walks the node's out-edges then in-edges in a
    // fixed order - no atomics, bitwise reproducible.
    const lfr::NodeInc *inc = a.node_inc + d.node_off;
    const uint32_t *in_idx = a.in_idx + d.edge_off;
    double *es = a.workspace + a.es_off[ci];
    const int lane = tid;     // PROF_FLUSH uses `lane == 0`
    (void)lane;
    // Two threads per matrix row in the assembly walk (out-edges / in-edges) keep the sums bitwise reproducible only while every
    // off-diagonal entry gets ONE term from each of them, i.e. while no node pair is matched twice inside the component (the
    // reference keeps duplicated matches, solve.cc:476-478; they are rare).  Duplicates sit next to each other in a node's in-edge
    // list (sorted by record index = by source node): one scan per solve decides, components with duplicates take the one-thread walk.
#ifndef LFR_SPLIT_WALK
#define LFR_SPLIT_WALK 1
#endif
    bool split_walk = LFR_SPLIT_WALK != 0;
    if (split_walk) {
        int dup = 0;
        for (int i = tid; i + 1 < E; i += kBlockThreads) {
            const uint32_t e0 = in_idx[i], e1 = in_idx[i + 1];
            const uint32_t k0 = *reinterpret_cast<const uint32_t *>(&edges[e0].src), k1 = *reinterpret_cast<const uint32_t *>(&edges[e1].src);
            dup |= ((k0 ^ k1) & 0x7fffffffu) == 0u;                 // same source, same destination (the kind bit aside)
        }
        split_walk = block_max<kBlockThreads>((double)dup, sh) == 0.0;
    }
    PROF_DECL
    auto sweep_scratch = [&](const double *xv, double *gout, bool want_matrix) -> double {
        double cost = 0.0;
        PROF_SWEEP_MARK(4);
        // the matrix starts at zero: cleared by everybody, coalesced, under the edge evaluation (the walk's barrier orders it)
        if (want_matrix) for (uint32_t i = tid; i < tri(n, 0); i += kBlockThreads) Mat[i] = 0.0;
        uint4 q[5], qn[5];                           // the next record is in flight while this one is evaluated
        if (tid < E) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(edges + tid);
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = rp[i];
        }
        for (int e = tid; e < E; e += kBlockThreads) {
            if (e + kBlockThreads < E) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(edges + e + kBlockThreads);
#pragma unroll
                for (int i = 0; i < 5; ++i) qn[i] = rp[i];
            }
            float flow[18];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                flow[4 * i] = __uint_as_float(q[i].x); flow[4 * i + 1] = __uint_as_float(q[i].y);
                flow[4 * i + 2] = __uint_as_float(q[i].z); flow[4 * i + 3] = __uint_as_float(q[i].w);
            }
            flow[16] = __uint_as_float(q[4].x); flow[17] = __uint_as_float(q[4].y);
            const float sim = __uint_as_float(q[4].z);
            const int s = (int)(q[4].w & 0xffffu), dk = (int)(q[4].w >> 16);
            const int dn = dk & 0x7fff, kind = dk >> 15;
            const int xa = s < n_var ? 2 * s : n, xb = dn < n_var ? 2 * dn : n;
            EdgeOut o;
            eval_edge<true>(flow, sim, kind, tv, xv[xa], xv[xa + 1], xv[xb], xv[xb + 1], o);
            cost += o.cost;
            double2 *w = reinterpret_cast<double2 *>(es + 8 * (size_t)e);
            w[0] = make_double2(o.j00, o.j01); w[1] = make_double2(o.j10, o.j11);
            w[2] = make_double2(o.sq, o.r0);   w[3] = make_double2(o.r1, 0.0);
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = qn[i];
        }
        const double total = block_sum<kBlockThreads>(cost, sh);          // (barriers inside: scratch is complete, Mat is zero)
        PROF_SWEEP_MARK(3);
        // The scratch and the edge records live in HBM/L2: the loads of kAhead edges are issued together
        // (one latency per batch instead of one per edge), then consumed in the fixed order.
#ifndef LFR_AHEAD
#define LFR_AHEAD 8
#endif
        constexpr int kAhead = LFR_AHEAD;
        auto walk_out = [&](const int row, const lfr::NodeInc &ni, double &gacc, double &dsame, double &dlow) {   // edges v -> w : J1 = d r / d x_v
            const int v = row >> 1, c = row & 1;
            for (uint32_t k0 = 0; k0 < ni.out_count; k0 += kAhead) {
                double2 q0[kAhead], q1[kAhead], q2[kAhead], q3[kAhead];
                int wn_[kAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const uint32_t e = ni.out_begin + min(k0 + u, ni.out_count - 1);
                    const double2 *w = reinterpret_cast<const double2 *>(es + 8 * (size_t)e);
                    q0[u] = w[0]; q1[u] = w[1]; q2[u] = w[2]; q3[u] = w[3];
                    wn_[u] = (int)(edges[e].dst_kind & 0x7fff);
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    if (k0 + u >= ni.out_count) break;
                    const double2 a0 = q0[u], a1 = q1[u], a2 = q2[u], a3 = q3[u];
                    const double jc0 = c ? a0.y : a0.x, jc1 = c ? a1.y : a1.x;    // column c of J1
                    gacc += jc0 * a2.y + jc1 * a3.x;
                    if (want_matrix) {
                        dsame += jc0 * jc0 + jc1 * jc1;
                        if (c) dlow += a0.y * a0.x + a1.y * a1.x;
                        const int wn = wn_[u];
                        if (wn < v) {                           // block (v, w) += J1^T * sq
                            mat_add(&Mat[tri(row, 2 * wn)], jc0 * a2.x);
                            mat_add(&Mat[tri(row, 2 * wn + 1)], jc1 * a2.x);
                        }
                    }
                }
            }
        };
        auto walk_in = [&](const int row, const lfr::NodeInc &ni, double &gacc, double &dsame) {                 // edges w -> v : d r / d x_v = sq * I
            const int v = row >> 1, c = row & 1;
            for (uint32_t k0 = 0; k0 < ni.in_count; k0 += kAhead) {
                double2 q0[kAhead], q1[kAhead], q2[kAhead], q3[kAhead];
                int wn_[kAhead];
                uint32_t e_[kAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) e_[u] = in_idx[ni.in_begin + min(k0 + u, ni.in_count - 1)];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const double2 *w = reinterpret_cast<const double2 *>(es + 8 * (size_t)e_[u]);
                    q0[u] = w[0]; q1[u] = w[1]; q2[u] = w[2]; q3[u] = w[3];
                    wn_[u] = (int)edges[e_[u]].src;
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    if (k0 + u >= ni.in_count) break;
                    const double2 a0 = q0[u], a1 = q1[u], a2 = q2[u], a3 = q3[u];
                    const double sq = a2.x, rc = c ? a3.x : a2.y;
                    gacc += sq * rc;
                    if (want_matrix) {
                        dsame += sq * sq;
                        const int wn = wn_[u];
                        if (wn < v) {                           // block (v, w) += sq * J1'
                            mat_add(&Mat[tri(row, 2 * wn)], sq * (c ? a1.x : a0.x));
                            mat_add(&Mat[tri(row, 2 * wn + 1)], sq * (c ? a1.y : a0.y));
                        }
                    }
                }
            }
        };
        if (!split_walk) {
            // one thread per matrix row: its out-edges, then its in-edges (components with duplicated matches: see split_walk)
            for (int row = tid; row < n; row += kBlockThreads) {
                const lfr::NodeInc ni = inc[row >> 1];
                double gacc = 0.0, dsame = 0.0, dlow = 0.0;     // A[row][row], A[2v+1][2v] (c == 1 only)
                walk_out(row, ni, gacc, dsame, dlow);
                walk_in(row, ni, gacc, dsame);
                gout[row] = gacc;
                if (want_matrix) {
                    Mat[tri(row, row)] = dsame;
                    if (row & 1) Mat[tri(row, row - 1)] = dlow;
                    vadiag[row] = dsame;
                }
            }
            __syncthreads();
        } else {
            // two threads per matrix row, in different waves (rows padded to whole waves: a wave runs ONE of the two loops): the
            // first takes the node's out-edges, the second its in-edges.  An off-diagonal entry receives one term from each
            // (x + y = y + x on top of an exact zero: the order of the two atomics is immaterial); diagonal and gradient partial
            // sums meet in a fixed order after the barrier.
            const int n_pad = (n + 63) & ~63;
            double *part_g = vstep, *part_d = vD;              // free during a sweep (rewritten at the next iteration's start)
            for (int item = tid; item < 2 * n_pad; item += kBlockThreads) {
                const bool in_half = item >= n_pad;              // wave-uniform
                const int row = item - (in_half ? n_pad : 0);
                if (row >= n) continue;
                const lfr::NodeInc ni = inc[row >> 1];
                double gacc = 0.0, dsame = 0.0, dlow = 0.0;
                if (!in_half) {
                    walk_out(row, ni, gacc, dsame, dlow);
                    gout[row] = gacc;
                    if (want_matrix) {
                        vadiag[row] = dsame;
                        if (row & 1) Mat[tri(row, row - 1)] = dlow;
                    }
                } else {
                    walk_in(row, ni, gacc, dsame);
                    part_g[row] = gacc;
                    part_d[row] = dsame;
                }
            }
            __syncthreads();
            for (int row = tid; row < n; row += kBlockThreads) {
                gout[row] += part_g[row];
                if (want_matrix) {
                    const double dd = vadiag[row] + part_d[row];
                    vadiag[row] = dd;
                    Mat[tri(row, row)] = dd;
                }
            }
            __syncthreads();
        }
        return total;
    };

    // ---- the fused sweep (round 3): evaluation and assembly in ONE pass, no scratch in HBM ----
    // The scratch sweep above moves 272 B per edge (80-B record, 64 B of corrected jacobian written, read by the out-walk and by
    // the in-walk): 21.9 GB per solve of config 5, 0.42 of the HBM roof with the waves waiting 59 % of their cycles.  Here four
    // neighbouring lanes own a node: they evaluate its out-edges (lane p takes edges p, p + 4, ...), keep the node's own diagonal
    // block and gradient in registers (summed over the four lanes in a fixed tree), send the cross block to LDS with no-return
    // atomics (an entry gets one term from each direction of a match: the order of two additions on an exact zero is immaterial)
    // and the terms that belong to the DESTINATION node - sq^2 on its diagonal, sq * r on its gradient, ~#in-edges contributors
    // in any order - as 64-bit FIXED-POINT integers (2^-40 units, |term| < 128, so 2^15 in-edges cannot overflow): integer
    // addition is associative, the sums are bitwise reproducible whatever the order.  Resolution 9e-13 on entries of O(1-100).
    // Components with duplicated matches (three or more terms on a cross entry) or a term outside the range take the scratch sweep.
#ifndef LFR_FUSED_SWEEP
#define LFR_FUSED_SWEEP 1
#endif
    bool fused_sweep = LFR_FUSED_SWEEP != 0 && split_walk && !a.scratch_sweep;
    auto sweep_fused = [&](const double *xv, double *gout, bool want_matrix, bool &overflow) -> double {
        constexpr double kFx = 0x1p40, kFxInv = 0x1p-40;
        unsigned long long *fx_g = reinterpret_cast<unsigned long long *>(vstep), *fx_d = reinterpret_cast<unsigned long long *>(vD);
        double cost = 0.0;
        PROF_SWEEP_MARK(4);
        if (want_matrix) for (uint32_t i = tid; i < tri(n, 0); i += kBlockThreads) Mat[i] = 0.0;
        for (int i = tid; i < n; i += kBlockThreads) { fx_g[i] = 0ull; if (i < n_var) fx_d[i] = 0ull; }
        if (tid == 0) sh.ovf = 0;
        __syncthreads();
        bool ovf = false;
        auto fx_add = [&](unsigned long long *p, double v) {
            ovf = ovf || !(fabs(v) < 128.0);
            __hip_atomic_fetch_add(p, (unsigned long long)__double2ll_rn(v * kFx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // lanes per node: as many as let every node of the component have its lanes in one pass (a wave holds 64 / P whole nodes),
        // eight at most; bigger components take four lanes per node and several passes
        const int n_nodes = (int)d.n_nodes;
        constexpr int kWavesT = kBlockThreads / 64;
        int P = 4;
#pragma unroll
        for (int q = 8; q >= 5; --q) if (P == 4 && (64 / q) * kWavesT >= n_nodes) P = q;
        const int npw = 64 / P, wl = tid & 63;
        const int slot = wl / P, part = wl - slot * P;
        const bool lane_on = slot < npw;
        for (int v0 = 0; v0 < n_nodes; v0 += kWavesT * npw) {                 // (uniform trip count: the reduction below is a wave operation)
            const int v = v0 + (tid >> 6) * npw + slot;
            const bool have_v = lane_on && v < n_nodes;
            const lfr::NodeInc ni = inc[have_v ? v : 0];
            const uint32_t cnt = have_v ? ni.out_count : 0u;
            const bool v_var = have_v && v < n_var;
            double d00 = 0.0, d10 = 0.0, d11 = 0.0, g0 = 0.0, g1 = 0.0;
            uint4 q[5], qn[5];
            if ((uint32_t)part < cnt) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(edges + ni.out_begin + part);
#pragma unroll
                for (int i = 0; i < 5; ++i) q[i] = rp[i];
            }
            for (uint32_t k = part; k < cnt; k += P) {
                if (k + P < cnt) {
                    const uint4 *rp = reinterpret_cast<const uint4 *>(edges + ni.out_begin + k + P);
#pragma unroll
                    for (int i = 0; i < 5; ++i) qn[i] = rp[i];
                }
                float flow[18];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    flow[4 * i] = __uint_as_float(q[i].x); flow[4 * i + 1] = __uint_as_float(q[i].y);
                    flow[4 * i + 2] = __uint_as_float(q[i].z); flow[4 * i + 3] = __uint_as_float(q[i].w);
                }
                flow[16] = __uint_as_float(q[4].x); flow[17] = __uint_as_float(q[4].y);
                const float sim = __uint_as_float(q[4].z);
                const int dk = (int)(q[4].w >> 16);
                const int dn = dk & 0x7fff, kind = dk >> 15;
                const bool w_var = dn < n_var;
                const int xa = v_var ? 2 * v : n, xb = w_var ? 2 * dn : n;
                EdgeOut o;
                eval_edge<true>(flow, sim, kind, tv, xv[xa], xv[xa + 1], xv[xb], xv[xb + 1], o);
                cost += o.cost;
                if (v_var) {
                    g0 += o.j00 * o.r0 + o.j10 * o.r1;
                    g1 += o.j01 * o.r0 + o.j11 * o.r1;
                    if (want_matrix) {
                        d00 += o.j00 * o.j00 + o.j10 * o.j10;
                        d10 += o.j01 * o.j00 + o.j11 * o.j10;
                        d11 += o.j01 * o.j01 + o.j11 * o.j11;
                    }
                }
                if (w_var) {
                    fx_add(&fx_g[2 * dn], o.sq * o.r0);
                    fx_add(&fx_g[2 * dn + 1], o.sq * o.r1);
                    if (want_matrix) {
                        fx_add(&fx_d[dn], o.sq * o.sq);
                        if (v_var && dn != v) {               // entry (2v + c, 2w + j) of the symmetric matrix += sq * J[j][c]
                            const bool below = dn < v;
                            const int rv = 2 * v, rw = 2 * dn;
                            mat_add(&Mat[below ? tri(rv, rw) : tri(rw, rv)], o.sq * o.j00);               // c = 0, j = 0
                            mat_add(&Mat[below ? tri(rv, rw + 1) : tri(rw + 1, rv)], o.sq * o.j10);       // c = 0, j = 1
                            mat_add(&Mat[below ? tri(rv + 1, rw) : tri(rw, rv + 1)], o.sq * o.j01);       // c = 1, j = 0
                            mat_add(&Mat[below ? tri(rv + 1, rw + 1) : tri(rw + 1, rv + 1)], o.sq * o.j11);   // c = 1, j = 1
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) q[i] = qn[i];
            }
            // the P lanes of the node, in a fixed tree: ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)), absent parts left out; the sum
            // lands in part 0
            auto quad = [&](double x) {
#pragma unroll
                for (int off = 1; off < 8; off <<= 1) {
                    const double y = __shfl_down(x, off, 64);
                    if (off < P && (part & (2 * off - 1)) == 0 && part + off < P) x += y;
                }
                return x;
            };
            g0 = quad(g0); g1 = quad(g1);
            if (want_matrix) { d00 = quad(d00); d10 = quad(d10); d11 = quad(d11); }
            if (v_var && part == 0) {
                gout[2 * v] = g0; gout[2 * v + 1] = g1;
                if (want_matrix) { vadiag[2 * v] = d00; vadiag[2 * v + 1] = d11; Mat[tri(2 * v + 1, 2 * v)] = d10; }
            }
        }
        if (ovf) sh.ovf = 1;
        const double total = block_sum<kBlockThreads>(cost, sh);          // (barriers inside: every accumulator is complete)
        overflow = sh.ovf != 0;
        PROF_SWEEP_MARK(3);
        if (overflow) { __syncthreads(); return total; }
        for (int row = tid; row < n; row += kBlockThreads) {
            gout[row] += (double)(long long)fx_g[row] * kFxInv;
            if (want_matrix) {
                const double dd = vadiag[row] + (double)(long long)fx_d[row >> 1] * kFxInv;
                vadiag[row] = dd;
                Mat[tri(row, row)] = dd;
            }
        }
        __syncthreads();
        return total;
    };
    auto sweep = [&](const double *xv, double *gout, bool want_matrix) -> double {
        if (fused_sweep) {
            bool overflow = false;
            const double c = sweep_fused(xv, gout, want_matrix, overflow);
            if (!overflow) return c;
            fused_sweep = false;                         // (uniform) this component stays with the scratch sweep
        }
        return sweep_scratch(xv, gout, want_matrix);
    };

    int exec_passes = 1;
    double cost = uni(sweep(vx, vg, true));          // (uni: the scalars of this loop are wave-uniform - SGPRs, not spilled VGPRs)
    PROF_MARK(0);                                     // 0: sweeps (evaluate + owner-computes assembly)
    for (int i = tid; i < n; i += kBlockThreads) vscale[i] = 1.0 / (1.0 + sqrt(vadiag[i]));
    __syncthreads();
    auto grad_max = [&](const double *xv, const double *gv) {
        double m = 0.0;
        for (int i = tid; i < n; i += kBlockThreads) m = fmax(m, fabs(xv[i] - clampb(xv[i] - gv[i])));
        return block_max<kBlockThreads>(m, sh);
    };
    double gmax = grad_max(vx, vg);
    double x_norm = 0.0, radius = kInitialRadius, decrease_factor = 2.0;
    bool reuse_diagonal = false, step_successful = true, matrix_valid = true;
    int n_invalid = 0, iteration = 0, term = LFR_TERM_CONVERGENCE;
    int n_successful = 0, n_ls_evals = 0, n_cand = 0;

    for (;;) {
        if (iteration >= kMaxIterations) { term = LFR_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= kGradientTol) break;
        if (radius <= kMinRadius) break;
        ++iteration;
        step_successful = false;

        PROF_MARK(4);                                 // 4: bookkeeping
        if (!matrix_valid) {           // the factorization of a rejected step overwrote J^T J: re-assemble
            sweep(vx, vg, true);
            ++exec_passes;
            matrix_valid = true;
            PROF_MARK(0);
        }
        // ---- the damped system.  Ceres solves (S A S + D^2) y = S g with the Jacobi scaling S; that is (A + S^-1 D^2 S^-1) (S y) = g: the
        // UNSCALED matrix with the same scaled LM diagonal gives the step S y directly - n diagonal entries to touch instead of n^2 / 2
        // (the packed kernel has always done this; rounds 1-2 scaled every entry here: 3 % of a 190-row component, 19 % of a sparse
        // 1000-row one whose tiles live in HBM) ----
        for (int i = tid; i < n; i += kBlockThreads) {
            if (!reuse_diagonal) vdiag[i] = fmin(fmax(vscale[i] * vscale[i] * vadiag[i], kMinLmDiag), kMaxLmDiag);
            const double di = sqrt(vdiag[i] / radius) / vscale[i];           // D / s
            vD[i] = di;
            vstep[i] = vg[i];
            Mat[tri(i, i)] += di * di;
        }
        reuse_diagonal = true;
        matrix_valid = false;
        __syncthreads();
        PROF_MARK(5);                                 // 5: damping
        double *vinv = vgn;            // free until the line search
        {
            factor_lds<kBlockThreads>(Mat, vinv, n, sh, a.prof ? a.prof + 8 * lfr::KC_COUNT + 8 + 16 * (a.cls - lfr::KC_BLOCK) : nullptr, a.queue + 15);        // 16-column panels, one barrier per panel (see factor_lds)
        }
        __syncthreads();
        PROF_MARK(1);
        bool valid = sh.flag == 0;
        if (valid) {
            {
                // Back substitution by ONE wave with the vector in registers (n <= 192: three values per lane).
                if (tid < 64) {
                    // L D L^T y = g with w = L^-1 g already in row n of the factored matrix (it rode through the factorization): z = w, then
                    // 16-row tile by tile from the last: inside the tile y = M (D^-1 z) - M = (I + D^-1 U^T)^-1 sits where the tile's strictly
                    // lower part was (factor_diag) - a 16 x 16 triangular matrix-vector product whose terms do not depend on one another
                    // (until round 5: 16 steps, each waiting for the one before: y_k = z_k / d_k, z_j -= a_kj y_k); then z_j -= a_kj y_k for
                    // the rows j above the tile (a_kj: the UNSCALED row k, so no per-entry scaling), again independent terms.  Lane = row:
                    // a term is a v_readlane broadcast and one multiply-add per register with a one-instruction address (row base + lane);
                    // the kernel is VALU-issue bound (one instruction per ~4.8 cycles and wave, scripts/probes/diag16_probe.hip).
                    constexpr int kR = 3;
                    double z[kR], inv[kR], yo[kR];
#pragma unroll
                    for (int r = 0; r < kR; ++r) {
                        const int i = tid + 64 * r;
                        z[r] = i < n ? vstep[i] : 0.0;
                        inv[r] = i < n ? vinv[i] : 0.0;
                        yo[r] = 0.0;
                    }
                    const double *mlane = Mat + tid;                               // + row base (wave-uniform) + 64 r (immediate)
                    auto back_block = [&](auto r0_tag) {
                        constexpr int r0 = decltype(r0_tag)::value;
                        const int lo = 64 * r0;
                        if (lo >= n) return;
                        const int lrow = tid + lo;                                 // the row this lane holds in register r0
                        for (int kb = min(n - 1, lo + 63) & ~15; kb >= lo; kb -= 16) {
                            double m[16][r0 + 1];
#pragma unroll
                            for (int u = 15; u >= 0; --u) m[u][r0] = (mlane + tri(min(kb + u, n), 0))[64 * r0];     // (rows above n - 1 only pad the last tile)
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const double *row = mlane + tri(min(kb + u, n), 0);
#pragma unroll
                                for (int r = 0; r < r0; ++r) m[u][r] = row[64 * r];
                            }
                            const double t = z[r0] * inv[r0];
                            const unsigned rel = (unsigned)(lrow - kb);            // 0..15 inside the tile
                            double y0 = t, y1 = 0.0;
#pragma unroll
                            for (int u = 15; u >= 1; --u) {
                                if (kb + u < n) {                                  // wave-uniform
                                    const double tc = readlane_f64(t, (kb + u) & 63);
                                    const double mv = rel < (unsigned)u ? m[u][r0] : 0.0;      // M[rel][u]
                                    if (u & 1) y1 = fma(mv, tc, y1); else y0 = fma(mv, tc, y0);
                                }
                            }
                            const double y = y0 + y1;
                            yo[r0] = rel < 16u ? y : yo[r0];
                            const bool above = lrow < kb;
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                if (kb + u < n) {                                  // wave-uniform
                                    const double yk = readlane_f64(y, (kb + u) & 63);
#pragma unroll
                                    for (int r = 0; r < r0; ++r) z[r] = fma(-m[u][r], yk, z[r]);
                                    z[r0] = fma(-(above ? m[u][r0] : 0.0), yk, z[r0]);
                                }
                            }
                        }
                    };
                    back_block(std::integral_constant<int, 2>{});
                    back_block(std::integral_constant<int, 1>{});
                    back_block(std::integral_constant<int, 0>{});
#pragma unroll
                    for (int r = 0; r < kR; ++r) if (tid + 64 * r < n) vstep[tid + 64 * r] = yo[r];
                }
                __syncthreads();
            }
        }
        PROF_MARK(6);                                 // 6: triangular solves
        double model_cost_change = 0.0;
        if (valid) {
            double part = 0.0, bad = 0.0;
            for (int i = tid; i < n; i += kBlockThreads) {
                const double rhs0 = vg[i];
                const double st = -vstep[i];
                if (!isfinite(st)) bad = 1.0;
                part += -rhs0 * st + vD[i] * vD[i] * st * st;
            }
            model_cost_change = uni(0.5 * block_sum<kBlockThreads>(part, sh));
            bad = block_max<kBlockThreads>(bad, sh);
            valid = bad == 0.0 && model_cost_change > 0.0;
        }
        if (!valid) {
            if (++n_invalid >= kMaxInvalid) { term = LFR_TERM_FAILURE; break; }
            radius = uni(radius / decrease_factor);
            decrease_factor = uni(decrease_factor * 2.0);
            continue;
        }
        n_invalid = 0;
        double gd_part = 0.0, dm_part = 0.0;
        for (int i = tid; i < n; i += kBlockThreads) {
            const double dl = -vstep[i];
            vdelta[i] = dl;
            gd_part += vg[i] * dl;
            dm_part = fmax(dm_part, fabs(dl));
        }
        const double g_dot_delta = block_sum<kBlockThreads>(gd_part, sh);
        const double dir_max = block_max<kBlockThreads>(dm_part, sh);

        // ---- projected Armijo line search ----
        double alpha = 1.0, cost_c = 0.0;
        bool ls_ok = false;
        {
            LsSample initial{0.0, cost, g_dot_delta, true, true}, previous{0, 0, 0, false, false}, current;
            int n_iter = 0;
            for (;;) {
                for (int i = tid; i < n; i += kBlockThreads) vxc[i] = clampb(__dadd_rn(vx[i], __dmul_rn(alpha, vdelta[i])));
                __syncthreads();
                PROF_MARK(4);
                cost_c = uni(sweep(vxc, vgn, true));  // also assembles J^T J at the trial point into Mat
                PROF_MARK(2);                         // 2: line-search sweeps
                ++exec_passes; ++n_ls_evals;
                current.x = alpha; current.value = cost_c; current.value_valid = isfinite(cost_c);
                current.gradient = 0.0; current.gradient_valid = false;
                if (current.value_valid && !(cost_c > cost + kLsSufficientDecrease * g_dot_delta * alpha)) { ls_ok = true; break; }
                if (current.value_valid) {
                    double p = 0.0;
                    for (int i = tid; i < n; i += kBlockThreads) p += vdelta[i] * vgn[i];
                    current.gradient = block_sum<kBlockThreads>(p, sh);
                    current.gradient_valid = isfinite(current.gradient);
                }
                const double nstep = ls_next_step_regs(initial, previous, current, dir_max, n_iter);
                if (nstep < 0.0) break;
                previous = current;
                alpha = uni(nstep);
            }
        }
        if (!ls_ok) {
            for (int i = tid; i < n; i += kBlockThreads) vxc[i] = clampb(__dadd_rn(vx[i], vdelta[i]));
            __syncthreads();
            cost_c = uni(sweep(vxc, vgn, true));
            ++exec_passes;
        }
        ++n_cand;
        const double cost_cand = uni(isfinite(cost_c) ? cost_c : DBL_MAX);
        double sn = 0.0;
        for (int i = tid; i < n; i += kBlockThreads) sn += (vx[i] - vxc[i]) * (vx[i] - vxc[i]);
        const double step_norm = sqrt(block_sum<kBlockThreads>(sn, sh));
        if (step_norm <= kParameterTol * (x_norm + kParameterTol)) break;
        const double cost_change = cost - cost_cand;
        if (fabs(cost_change) <= kFunctionTol * cost) break;
        const double rel = cost_change / model_cost_change;
        if (rel > kMinRelDecrease) {
            double xn = 0.0;
            __syncthreads();
            for (int i = tid; i < n; i += kBlockThreads) { vx[i] = vxc[i]; xn += vxc[i] * vxc[i]; }
            x_norm = uni(sqrt(block_sum<kBlockThreads>(xn, sh)));
            // the accepted candidate is the last evaluated point: its cost, gradient and J^T J (in Mat,
            // the factorization was no longer needed) are already there - no extra sweep
            for (int i = tid; i < n; i += kBlockThreads) vg[i] = vgn[i];
            __syncthreads();
            cost = cost_cand;
            matrix_valid = true;
            gmax = grad_max(vx, vg);
            step_successful = true;
            ++n_successful;
            const double t = 2.0 * rel - 1.0;
            radius = uni(fmin(kMaxRadius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t)));
            decrease_factor = 2.0;
            reuse_diagonal = false;
        } else {
            radius = uni(radius / decrease_factor);
            decrease_factor = uni(decrease_factor * 2.0);
        }
    }
    __syncthreads();
    PROF_MARK(4);
#ifdef LFR_PROFILE_ONLY_ITER       // diagnostic builds: phase profile of the slow components only
    if (iteration >= LFR_PROFILE_ONLY_ITER)
#endif
    PROF_FLUSH();
    for (int i = tid; i < n; i += kBlockThreads)
        a.positions[2 * (size_t)a.node_ids[d.node_off + (i >> 1)] + (i & 1)] = term != LFR_TERM_FAILURE ? vx[i] : 0.0;
    if (tid == 0) {
        CompInfoDev inf;
        inf.iterations = iteration; inf.termination = term; inf.n_successful = n_successful;
        inf.n_ls_evals = n_ls_evals; inf.n_cand_evals = n_cand; inf.exec_passes = exec_passes;
        inf.final_cost = cost;
#ifdef LFR_PROFILE_WGTIME          // diagnostic builds: the workgroup's lifetime in s_memtime ticks instead of the cost
        inf.final_cost = (double)(__builtin_amdgcn_s_memtime() - wg_t0_);
        if (LFR_PROFILE_WGTIME == 2) inf.final_cost = exec_passes + 1e3 * n_ls_evals + 1e6 * n_cand + 1e9 * n_successful + 1e12 * n_invalid;
        if (LFR_PROFILE_WGTIME == 3) {           // start time, lifetime in kiloticks, hardware id (XCC, SE, CU)
            inf.final_cost = (double)wg_r0_;
            inf.iterations = (int)(wall_clock64() - wg_r0_);
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));          // HW_ID: cu [11:8], sh [12], se [15:13]
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));         // XCC_ID [3:0]
            inf.termination = (int)(((hw >> 8) & 0xffu) | ((xcc & 0xfu) << 8));
        }
#endif
        a.infos[ci] = inf;
    }
}

// =============================================================================================
// Components whose normal matrix does not fit LDS (> 192 rows): LEVEL-SCHEDULED sparse LDL^T along the elimination tree
// =============================================================================================
// The reference gives these systems to Ceres' SPARSE_NORMAL_CHOLESKY (solve.cc:147).  The plan (lfr_treeplan.cpp, made when the batch
// is created) renumbers the variable nodes by nested dissection, packs them into blocks of <= 8 nodes = 16 rows, factors the block
// structure symbolically and lists the columns by LEVEL of the block elimination tree.  Columns of one level are independent, so the
// factorization is a handful of levels (6-8 for a 2.5 k-row cap-sized component) whose column tasks the workgroup's waves take side by
// side - round 3's block-envelope kernel walked ~170 dependent panels with one wave working and three waiting.
//
// Workspace of a component (a.workspace + a.ws_off[ci]): the plan's words (TreePlan::blob), the 16x16 tiles column by column (the
// diagonal tile first), 6 doubles of partial sums per sweep item, kTreeVectors vectors of 16 NB + 16 doubles in MATRIX order (row of
// node at position p, coordinate c: 2 p + c; index 16 NB is a zero slot the constants read).
//
//   sweep      one lane per (node, neighbour) ITEM: it evaluates the records between the two nodes (both directions, duplicates) and
//              STORES the pair's 2x2 cross block; the node's diagonal block and gradient are summed over its items in list order by
//              one lane.  No atomics, no scratch per edge: bitwise reproducible whatever the input (a record is evaluated by the
//              owner of either end: two evaluations per record, nothing else to exchange).
//   factor     THIN plans (every column's rows fit one wave's carried tiles - every cap-sized component measured): one wave per COLUMN
//              TASK, no workgroup barrier between levels.  A task streams through the column's left-looking update entries
//              sum_k U(I,k) D_k^-1 U(J,k)^T on the fp64 matrix cores (the right-hand side w_J rides in the diagonal tile) - entry by
//              entry as the state word of entry column k turns 1 -, then factors the diagonal tile with lane = row in lanes 0-15
//              (v_readlane broadcasts) while lane 16 carries w_J and lanes 17-63 the rows of the tiles below it through the same steps
//              (a_ic -= (a_ik / d_k) a_ck IS their substitution), stores, and publishes its own state word.  Waves take the columns of
//              a level round-robin, rotated by level; with TEAMS the waves of up to 8 workgroups of one XCD share the walk (state words
//              in the workspace, sc1 loads).  Plans that are not thin keep the level-by-level phases (updates / diagonal tiles / extra
//              rows) with a barrier after each phase that had work.
//   substitute levels top down, one wave per column (thin plans: gated by the parent's state word turning 2 instead of a barrier per
//              level): z_J = w_J - sum_I U(I,J)^T y_I, then the 16 steps inside the diagonal tile.
// Column k keeps the UNSCALED entries a_ik (L_ik = a_ik / d_k), d_k on the diagonal, 1 / d_k in vinv.
#ifndef LFR_THREADS_G
#define LFR_THREADS_G 512
#endif
// -DLFR_PROFILE_PHASES -DLFR_PROFILE_TREE: the slots of the phase profile split the tree kernel's sweep and factorization -
// 0 node pass + reductions of the sweeps, 1 column tasks (+ extra rows), 2 zeroing the tiles, 3 sweep items, 4 bookkeeping and the LM
// diagonal, 5 left-looking updates, 6 back substitution
#if defined(LFR_PROFILE_PHASES) && defined(LFR_PROFILE_TREE)
#define TPROF_MARK(i) PROF_MARK(i)
#define LFR_TREE_SLOT_SCALE 4
#else
#define TPROF_MARK(i)
#define LFR_TREE_SLOT_SCALE 5
#endif
struct TreeShared {
    // per wave: four tiles in rows of 18 doubles (the column task turns its accumulators from the matrix cores' layout into
    // lane = row through them; the extra-row tasks stage a diagonal tile there) + 16 doubles (the right-hand side / 1/d)
    double x[LFR_THREADS_G / 64][4 * 288 + 16];
    double red3[LFR_THREADS_G / 64][5];
    int pend[lfr::kTreeMaxFlagColumns];        // "thin" plans: children a column still waits for (factorization) / column solved (back substitution)
};
// ---- TEAMS: several workgroups on ONE component (round 5) ----
// A launch of this class lasts as long as its slowest component, and a component used to be one workgroup on one CU whatever its size
// (130 components on 256 CUs: half the chip idle while a 2258-row component ran its 36 LM iterations of ~250 us alone).  A component whose
// expected work (the hand-out key of k_wg_order_keys) is above a threshold is now solved by a TEAM of 2, 4 or 8 workgroups: the sweep's
// items, the node pass and the vector passes are split over the team's threads, the column tasks of the factorization and of the back
// substitution over the team's waves (same dependency counters, in HBM instead of LDS), the reductions meet in a fixed order (every
// member computes the same bits, so the replicated trust-region state stays identical) - results are a function of the team size, which
// is a function of the component, hence bitwise repeatable.
//
// Visibility (MI355X: per-CU vector L1 never refreshed by other CUs' stores, one L2 per XCD): a team is formed from workgroups that
// registered with the SAME XCC id, so its members share one L2.  Payload (tiles, vectors, partial sums): plain stores, `s_waitcnt
// vmcnt(0)` before the hand-off (the store has reached L2), and every load of mutable workspace data with `sc1` (bypasses L1, served
// by the shared L2).  Synchronisation words (barrier counters, dependency counters, mailboxes, flags): agent-scope atomics only.
// No `buffer_wbl2` / `buffer_inv` anywhere: nothing has to leave or re-enter the XCD.  Teams never span XCDs by construction
// (registration below); workgroups that cannot complete a unit of their XCD work alone.
//
// Control words of a launch (KernelArgs::team_ctl, zeroed per solve): [0..7] workgroups registered per XCC, [8] registered in total,
// [9] abort (a bounded spin ran out somewhere: every wait gives up, the components involved fail), [10] components solved by a team,
// [11] SOLO mode (below), [12] components being solved right now, [13] components solved off their team size, [16 + 16 u + m] mailbox
// of member m of unit u (a one-slot channel: the leader stores when it reads 0, the member clears), [16 + 16 u + 8 + L] arrival counter
// of the team led by rank L, [kTeamUnitState + u] state of unit u (0 forming, 1 complete, 2 dissolved: decided ONCE by compare-and-swap,
// so that the members of a unit agree).  Reduction slots (KernelArgs::team_red): per unit and leader, two parities x members x 4 doubles.
//
// Residency.  Units form from workgroups that are resident at the same time, and nothing guarantees that: another
// kernel may hold CUs (the other classes of the same solve do, by design; so may another process or a smaller partition).  Waiting is
// harmless while SOMEBODY works - a team at work ends, takes the next component, exits when the queue is empty, and the CUs it frees
// bring the missing registrations.  The one state that would never end by itself: nobody at work ([12] = 0), the queue not empty, and
// everyone resident waiting - in registration for members that cannot arrive, or at the head of the queue for a team size that does not
// exist.  A waiter that sees [12] = 0 and neither a registration nor the queue's head move for team_patience_us (wall clock, 50 ms by
// default) raises [11]: units still forming dissolve, and from then on ANY leader takes the head of the queue and solves it alone -
// the launch degrades to one workgroup per component (the plain solve_tree_kernel's way; that component's bits are those of a team of
// one, counted in [13] = lfr_batch_team_fallbacks), no wait runs out, nothing fails.
#ifndef LFR_TEAM_MAX
#define LFR_TEAM_MAX 8
#endif
constexpr int kTeamMax = LFR_TEAM_MAX;                     // workgroups per unit (a power of two <= 8)
constexpr int kTeamUnitsPerXcc = 256 / kTeamMax;           // a launch has at most 256 workgroups
constexpr int kTeamUnitState = 16 + 16 * 8 * kTeamUnitsPerXcc;
constexpr int kTeamCtlWords = kTeamUnitState + 8 * kTeamUnitsPerXcc;
constexpr int kTeamRedPerUnit = (kTeamMax / 2) * 2 * kTeamMax * 16;
constexpr unsigned kTeamMsgEnd = 0xf0000000u;
static_assert(kTeamMax == 2 || kTeamMax == 4 || kTeamMax == 8, "team size");
struct TeamCtx {
    int S = 1, r = 0;                 // workgroups in the team, this workgroup's index in it
    unsigned int *bar = nullptr;      // arrival counter of the team (monotonic)
    unsigned int target = 0;          // its value when everyone has arrived at the latest barrier
    double *red = nullptr;            // reduction slots [2][kTeamMax][16]: five {value, tag} granules per member
    unsigned gen = 0, epoch = 0;      // reductions of this team so far; the launch (tags of earlier launches linger in the slots)
    unsigned int *ctl = nullptr;      // control words of the launch ([9] = abort)
    bool dead = false;                // this workgroup has seen the abort word: waits return at once
};
__device__ __forceinline__ unsigned team_ld(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void team_st(unsigned int *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one thread: spin until *p - target >= 0 (monotonic counters) / until pred; false = gave up (abort raised)
__device__ __noinline__ bool team_wait_counter(unsigned int *p, unsigned target, unsigned int *ctl, unsigned int *timeouts, int limit_log2) {
    int spins = 0;
    while ((int)(team_ld(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255) == 0 && (team_ld(ctl + 9) != 0u || (spins >> limit_log2) != 0)) {
            if (team_ld(ctl + 9) == 0u) atomicAdd(timeouts, 1u);
            team_st(ctl + 9, 1u);
            return false;
        }
    }
    return true;
}

// The plan's words are written by the host before the launch and never by the kernel: read through the constant address space they
// are scalar loads (s_load, one per wave, through the scalar cache) instead of vector loads + v_readfirstlane behind ~1 us of latency.
typedef const __attribute__((address_space(4))) uint32_t *PlanWords;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// four sums and a maximum over the workgroup with one pair of barriers
template <int kBlockThreads>
__device__ __forceinline__ void block_reduce5(double &s0, double &s1, double &s2, double &s3, double &m0, TreeShared &ts) {
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); m0 = wave_max(m0);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { double *r = ts.red3[threadIdx.x >> 6]; r[0] = s0; r[1] = s1; r[2] = s2; r[3] = s3; r[4] = m0; }
    __syncthreads();
    double a = 0.0, b = 0.0, e = 0.0, f = 0.0, c = ts.red3[0][4];
#pragma unroll
    for (int w = 0; w < kBlockThreads / 64; ++w) { a += ts.red3[w][0]; b += ts.red3[w][1]; e += ts.red3[w][2]; f += ts.red3[w][3]; c = fmax(c, ts.red3[w][4]); }
    s0 = a; s1 = b; s2 = e; s3 = f; m0 = c;
}

template <int kBlockThreads, bool TEAM>
__device__ __forceinline__ void solve_tree_component(const KernelArgs &a, const int ci, BlockShared &sh, TreeShared &ts, TeamCtx &tm) {
    constexpr int kWaves = kBlockThreads / 64;
    constexpr uint32_t kNone = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
#if defined(LFR_PROFILE_WGTIME) && LFR_PROFILE_WGTIME == 3
    const unsigned long long tree_r0_ = wall_clock64();
#endif
    // the team's threads / waves (TEAM = false: this workgroup alone, and every helper below is what it was before teams existed)
    const int tS = TEAM ? tm.S : 1;
    const int gt = TEAM ? tm.r * kBlockThreads + tid : tid, GT = tS * kBlockThreads;
    const int gw = TEAM ? tm.r * kWaves + wave : wave, GW = tS * kWaves;
    constexpr int kAux = TEAM ? 16 : 0;                   // cache policy of the workspace loads: sc1 = past this CU's L1
    auto ldd = [](const double *p) -> double {
        if constexpr (TEAM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
    };
#ifdef LFR_TRACE_TREE      // (scripts/tree_trace.py) events of the FIRST component of the hand-out order, LM iterations 2-4
#ifndef LFR_TRACE_IT0
#define LFR_TRACE_IT0 2
#define LFR_TRACE_IT1 4
#endif
    int trace_it = 0;
    const bool traced = a.trace != nullptr && ci == (int)a.wg_order[a.desc_begin - a.wg_begin];
    auto tr = [&](const unsigned type, const unsigned col) {
        if (traced && trace_it >= LFR_TRACE_IT0 && trace_it <= LFR_TRACE_IT1 && lane == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            const unsigned long long i = atomicAdd(a.trace, 2ull);
            if (i + 2 < (1ull << 20)) { a.trace[2 + i] = t; a.trace[3 + i] = ((unsigned long long)type << 56) | ((unsigned long long)gw << 48) | ((unsigned long long)trace_it << 32) | col; }
        }
    };
#define TR(type, col) tr(type, col)
#define TR_IT(it) trace_it = (it)
#else
#define TR(type, col)
#define TR_IT(it)
#endif
    // barrier over the team: every wave's stores have reached L2 before its workgroup arrives
    auto tsync = [&]() {
        if constexpr (TEAM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            tm.target += (unsigned)tS;
            if (tid == 0 && !tm.dead) {
                __hip_atomic_fetch_add(tm.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!team_wait_counter(tm.bar, tm.target, tm.ctl, a.queue + 15, 22)) tm.dead = true;
            }
            __syncthreads();
        } else {
            __syncthreads();
        }
    };
    const CompDesc d = a.descs[ci];
    const int tv = a.tukey_variant;
    const EdgeRec *edges = a.edges + d.edge_off;
    double *wsb = a.workspace + a.ws_off[ci];
    const PlanWords pl = (PlanWords)(uintptr_t)wsb;
    const int NB = (int)pl[0], n_tiles = (int)pl[1], n_pad = (int)pl[4], n_levels = (int)pl[5], n_items = (int)pl[6];
    const PlanWords colptr = pl + pl[8], rowsof = pl + pl[9], nreal = pl + pl[10], level_ptr = pl + pl[11], level_cols = pl + pl[12],
                    p1_ptr = pl + pl[13], p1_tasks = pl + pl[14], upd = pl + pl[15], x_ptr = pl + pl[16], x_tasks = pl + pl[17],
                    ncarry = pl + pl[18], col_upd_ptr = pl + pl[25], col_upd = pl + pl[26];
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(wsb);            // (per-lane reads: vector loads)
    const uint32_t *items = hdr + pl[19], *item_edges = hdr + pl[20], *node_items = hdr + pl[21], *ipos = hdr + pl[22];
    const PlanWords col_desc = pl + pl[27];
    double *atiles = wsb + pl[2], *tiles = atiles + ((size_t)n_tiles << 8), *part = wsb + pl[23], *vec = wsb + pl[3];     // A (the sweeps' J^T J), the factor
    const size_t vs = pl[24];
    double *vx = vec, *vxc = vec + vs, *vg = vec + 2 * vs, *vgn = vec + 3 * vs, *vscale = vec + 4 * vs, *vdiag = vec + 5 * vs, *vstep = vec + 6 * vs,
           *vD = vec + 7 * vs, *vadiag = vec + 8 * vs, *vdelta = vec + 9 * vs, *vinv = vec + 10 * vs, *vw = vec + 11 * vs;
    const int n = n_pad;                                 // rows incl. padding (inert: x = g = step = 0)

    unsigned int *gteam = reinterpret_cast<unsigned int *>(wsb + pl[29]);      // TEAM: [0] bad-pivot flag, [16 + J] dependency counter of column J
    for (int i = gt; i < (int)vs; i += GT) {
        vx[i] = 0.0; vxc[i] = 0.0; vscale[i] = 1.0; vD[i] = 0.0; vg[i] = 0.0; vgn[i] = 0.0; vstep[i] = 0.0; vinv[i] = 0.0; vw[i] = 0.0;
        vadiag[i] = 0.0; vdiag[i] = 0.0; vdelta[i] = 0.0;
    }
    {   // A's tiles: zero once per solve - a sweep stores the same entries every time (one 2x2 block per matched pair, one per node).
        // The factor's tiles too: the column task writes rows 0-14 of the third tile it carries, and row 15 (a padding row) must not
        // hold whatever the workspace held before (a NaN times the zero of a padding row's solution is a NaN).
        double2 *t2 = reinterpret_cast<double2 *>(atiles);
        const double2 z = make_double2(0.0, 0.0);
        for (size_t i = gt; i < ((size_t)n_tiles << 8); i += GT) t2[i] = z;
    }
    tsync();
    // sums and maxima over the team: the workgroup's value first (every thread holds it), then the members' values in a fixed order
    // A reduction over the team IS a barrier: every workgroup publishes its five values as 16-byte {value, tag} granules (one lane,
    // one write-through store each; tag = launch << 32 | reduction number) once all its waves' stores have reached L2, and gathers
    // the members' granules by polling them, a lane per granule - one store and one round of polls instead of an arrival counter,
    // its polls, and a separate read of the values.  Two sets of slots alternate: a member can only be one reduction ahead.
    auto treduce5 = [&](double &s0, double &s1, double &s2, double &s3, double &m0) {
        if constexpr (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_reduce5<kBlockThreads>(s0, s1, s2, s3, m0, ts);
        if constexpr (TEAM) {
            ++tm.gen;
            if (wave == 0) {
                const unsigned long long tag = ((unsigned long long)tm.epoch << 32) | tm.gen;
                const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(tm.red + (tm.gen & 1u) * (kTeamMax * 16), 0, (int)0xfffffffe, 0x00020000);
                if (lane < 5) {
                    const double v = lane == 0 ? s0 : lane == 1 ? s1 : lane == 2 ? s2 : lane == 3 ? s3 : m0;
                    const u32x2_t vb = __builtin_bit_cast(u32x2_t, v);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{vb[0], vb[1], (unsigned)tag, (unsigned)(tag >> 32)}, rR, (unsigned)(128 * tm.r + 16 * lane), 0, 16);
                }
                const int m = lane / 5, j = lane - 5 * m;
                const bool mine = lane < 5 * tS;
                double val = 0.0;
                bool got = !mine || tm.dead;
                int spins = 0;
                while (true) {
                    if (!got) {
                        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rR, (unsigned)(128 * m + 16 * j), 0, 16);
                        if ((((unsigned long long)q[3] << 32) | q[2]) == tag) { got = true; val = __builtin_bit_cast(double, u32x2_t{q[0], q[1]}); }
                    }
                    if (__ballot(!got) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255) == 0 && (__builtin_amdgcn_readfirstlane((int)team_ld(tm.ctl + 9)) != 0 || (spins >> 22) != 0)) {
                        if (lane == 0) { if (team_ld(tm.ctl + 9) == 0u) atomicAdd(a.queue + 15, 1u); team_st(tm.ctl + 9, 1u); }
                        tm.dead = true;
                        break;
                    }
                }
                double x[5];
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) {
                    double acc = readlane_f64(val, jj);
                    for (int mm = 1; mm < tS; ++mm) { const double t = readlane_f64(val, 5 * mm + jj); acc = jj == 4 ? fmax(acc, t) : acc + t; }
                    x[jj] = acc;
                }
                if (lane == 0) { sh.red[0] = x[0]; sh.red[1] = x[1]; sh.red[2] = x[2]; sh.red[3] = x[3]; sh.red[4] = x[4]; }     // (not ts.red3: slower waves may still be reading it)
            }
            __syncthreads();
            s0 = sh.red[0]; s1 = sh.red[1]; s2 = sh.red[2]; s3 = sh.red[3]; m0 = sh.red[4];
            __syncthreads();
        }
    };
    auto treduce4 = [&](double &s0, double &s1, double &s2, double &m0) { double z = 0.0; treduce5(s0, s1, s2, z, m0); };

    PROF_DECL
    // ---- one sweep at x = xb (trial = false) or at the trial point clamp(xb + alpha dl): the cost, gout = J^T r and the unscaled
    //      J^T J in the tiles.  The trial point is formed on the fly by whoever needs a coordinate (separately rounded product and
    //      sum, like Ceres forms its candidate) and STORED by the node pass (xout) - no vector pass, no barrier in front of the
    //      items: what they read (xb, dl) was published by the barrier of an earlier reduction, what they write (cross blocks,
    //      partial sums) was last read before one.  Besides the cost the sweep returns what the trust-region loop wants to know
    //      about the point: sn = |x - xc|^2, xn = |xc|^2, gd = dl . g(xc), gm = max |xc - clamp(xc - g(xc))|. ----
    struct SweepOut { double cost, sn, xn, gd, gm; };
    auto sweep = [&](const double *xb, const double *dl, const double alpha, const bool trial, double *xout, double *gout) -> SweepOut {
        TPROF_MARK(2);
        double cost = 0.0;
        auto point = [&](const uint32_t row) -> double {
            const double x0 = ldd(xb + row);
            return trial ? clampb(__dadd_rn(x0, __dmul_rn(alpha, ldd(dl + row)))) : x0;
        };
        // One item = 8 words (lfr_treeplan.cpp): rows of the node and of the neighbour, the pair's block in A, the record count, the first
        // two records inline.  The item of the NEXT pass is fetched before this one is evaluated, and both records of the usual pair
        // are loaded before the first evaluation: one exposed round trip per pass instead of three dependent ones.
        auto load_record = [&](const uint32_t ew, uint4 (&qv)[5]) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(edges + (ew >> 2));
#pragma unroll
            for (int k = 0; k < 5; ++k) qv[k] = rp[k];
        };
        int i = gt;
        uint4 ia = make_uint4(0u, 0u, 0u, 0u), ib = ia;
        if (i < n_items) { ia = reinterpret_cast<const uint4 *>(items)[2 * i]; ib = reinterpret_cast<const uint4 *>(items)[2 * i + 1]; }
        while (i < n_items) {
            const uint4 it = ia, it2 = ib;                                       // {row, row of the neighbour, cross block, records} {record 0, record 1, further, -}
            const int in = i + GT;
            if (in < n_items) { ia = reinterpret_cast<const uint4 *>(items)[2 * in]; ib = reinterpret_cast<const uint4 *>(items)[2 * in + 1]; }
            const double xv0 = point(it.x), xv1 = point(it.x + 1), xu0 = point(it.y), xu1 = point(it.y + 1);
            uint4 q0[5], q1[5];
            load_record(it2.x, q0);
            load_record(it.w > 1u ? it2.y : it2.x, q1);
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0, d00 = 0.0, d10 = 0.0, d11 = 0.0, g0 = 0.0, g1 = 0.0;
            auto one = [&](const uint32_t ew, const uint4 (&qv)[5]) {
                float flow[18];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    flow[4 * k] = __uint_as_float(qv[k].x); flow[4 * k + 1] = __uint_as_float(qv[k].y);
                    flow[4 * k + 2] = __uint_as_float(qv[k].z); flow[4 * k + 3] = __uint_as_float(qv[k].w);
                }
                flow[16] = __uint_as_float(qv[4].x); flow[17] = __uint_as_float(qv[4].y);
                const float sim = __uint_as_float(qv[4].z);
                const int kind = (int)(qv[4].w >> 31);
                const bool rev = (ew & 2u) != 0u;                                // the record runs neighbour -> node
                EdgeOut o;
                eval_edge<true>(flow, sim, kind, tv, rev ? xu0 : xv0, rev ? xu1 : xv1, rev ? xv0 : xu0, rev ? xv1 : xu1, o);
                if (ew & 1u) cost += o.cost;
                if (!rev) {                                                      // d r / d x_node = J1, d r / d x_neighbour = sq I
                    d00 += o.j00 * o.j00 + o.j10 * o.j10; d10 += o.j01 * o.j00 + o.j11 * o.j10; d11 += o.j01 * o.j01 + o.j11 * o.j11;
                    g0 += o.j00 * o.r0 + o.j10 * o.r1; g1 += o.j01 * o.r0 + o.j11 * o.r1;
                    c00 += o.j00 * o.sq; c01 += o.j10 * o.sq; c10 += o.j01 * o.sq; c11 += o.j11 * o.sq;       // J1^T (sq I)
                } else {                                                         // d r / d x_node = sq I, d r / d x_neighbour = J1
                    d00 += o.sq * o.sq; d11 += o.sq * o.sq;
                    g0 += o.sq * o.r0; g1 += o.sq * o.r1;
                    c00 += o.sq * o.j00; c01 += o.sq * o.j01; c10 += o.sq * o.j10; c11 += o.sq * o.j11;       // (sq I) J1
                }
            };
            if (it.w > 0u) one(it2.x, q0);
            if (it.w > 1u) one(it2.y, q1);
            for (uint32_t q = 2u; q < it.w; ++q) {                               // duplicated matches
                const uint32_t ew = item_edges[it2.z + q - 2u];
                uint4 qx[5];
                load_record(ew, qx);
                one(ew, qx);
            }
            if (it.z != kNone) {                                                 // the neighbour sits earlier in the order: the pair's block is stored here
                double2 *t = reinterpret_cast<double2 *>(atiles + it.z);
                t[0] = make_double2(c00, c01); t[8] = make_double2(c10, c11);
            }
            double2 *pp = reinterpret_cast<double2 *>(part + 6 * (size_t)i);
            pp[0] = make_double2(d00, d10); pp[1] = make_double2(d11, g0); pp[2] = make_double2(g1, 0.0);
            i = in;
        }
        TPROF_MARK(3);
        TR(20, 0);
        SweepOut so;
        // The node pass: one lane per node sums the node's items in list order.  What it needs that does not depend on the items
        // (its item range, its point, the diagonal tile) is requested BEFORE the barrier that publishes the items' partial sums; the
        // cost rides in the reduction behind the node pass: one barrier and one reduction per sweep.
        struct NodePre { uint32_t ip, i0, i1, dt; double xo0, xo1, dl0, dl1; };
        auto node_pre = [&](const int p) -> NodePre {
            NodePre q;
            q.ip = p < 8 * NB ? ipos[p] : kNone;
            q.i0 = q.i1 = 0u; q.dt = 0u; q.xo0 = q.xo1 = q.dl0 = q.dl1 = 0.0;
            if (q.ip != kNone) {
                q.i0 = node_items[p]; q.i1 = node_items[p + 1]; q.dt = hdr[pl[8] + (p >> 3)];
                q.xo0 = ldd(xb + 2 * p); q.xo1 = ldd(xb + 2 * p + 1);
                if (trial) { q.dl0 = ldd(dl + 2 * p); q.dl1 = ldd(dl + 2 * p + 1); }
            }
            return q;
        };
#ifndef LFR_T_SWEEP_MERGE
#define LFR_T_SWEEP_MERGE 1
#endif
#if LFR_T_SWEEP_MERGE
        NodePre np = node_pre(gt);
        tsync();                                                            // cross blocks and partial sums are out
#else
        { double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0; treduce5(cost, z1, z2, z3, z4); }
        const double cost_total = cost;
        cost = 0.0;
        NodePre np = node_pre(gt);
#endif
        TR(21, 0);
        double sn = 0.0, xn = 0.0, gd = 0.0, gm = 0.0;
        for (int p = gt; p < 8 * NB; p += GT) {
#if LFR_T_SWEEP_MERGE
            const NodePre nq = np;
            if (p + GT < 8 * NB) np = node_pre(p + GT);
#else
            const NodePre nq = p == gt ? np : node_pre(p);
#endif
            if (nq.ip == kNone) continue;
            const uint32_t i0 = nq.i0, i1 = nq.i1;
            const double xo0 = nq.xo0, xo1 = nq.xo1, dl0 = nq.dl0, dl1 = nq.dl1;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
#ifndef LFR_T_NODE_BATCH
#define LFR_T_NODE_BATCH 1
#endif
            for (uint32_t ib0 = i0; ib0 < i1; ib0 += (uint32_t)LFR_T_NODE_BATCH) {
                double u[LFR_T_NODE_BATCH][5];
#pragma unroll
                for (int k = 0; k < LFR_T_NODE_BATCH; ++k) {
                    const double *pp = part + 6 * (size_t)min(ib0 + (uint32_t)k, i1 - 1u);
#pragma unroll
                    for (int c = 0; c < 5; ++c) u[k][c] = ldd(pp + c);
                }
#pragma unroll
                for (int k = 0; k < LFR_T_NODE_BATCH; ++k) {
                    if (ib0 + (uint32_t)k < i1) { s0 += u[k][0]; s1 += u[k][1]; s2 += u[k][2]; s3 += u[k][3]; s4 += u[k][4]; }
                }
            }
            double *T = atiles + ((size_t)nq.dt << 8) + 34 * (p & 7);  // entry (2 slot, 2 slot) of the diagonal tile
            T[0] = s0; T[16] = s1; T[17] = s2;
            gout[2 * p] = s3; gout[2 * p + 1] = s4;
            vadiag[2 * p] = s0; vadiag[2 * p + 1] = s2;
            const double xc0 = trial ? clampb(__dadd_rn(xo0, __dmul_rn(alpha, dl0))) : xo0, xc1 = trial ? clampb(__dadd_rn(xo1, __dmul_rn(alpha, dl1))) : xo1;
            if (trial) { xout[2 * p] = xc0; xout[2 * p + 1] = xc1; }
            sn += (xo0 - xc0) * (xo0 - xc0) + (xo1 - xc1) * (xo1 - xc1);
            xn += xc0 * xc0 + xc1 * xc1;
            gd += dl0 * s3 + dl1 * s4;
            gm = fmax(gm, fmax(fabs(xc0 - clampb(xc0 - s3)), fabs(xc1 - clampb(xc1 - s4))));
        }
        TR(22, 0);
        treduce5(cost, sn, xn, gd, gm);                                     // (its barrier also publishes the node pass)
        TR(23, 0);
#if !LFR_T_SWEEP_MERGE
        cost = cost_total;
#endif
        so.cost = cost; so.sn = sn; so.xn = xn; so.gd = gd; so.gm = gm;
        return so;
    };

    // descriptor of a column task (lfr_treeplan.cpp: col_desc, columns in level order): one pair of scalar loads
    struct ColDesc { uint32_t J, t0, nc, nbp, ne, e_rest, k0, tb0, a00, a01, a02, k1, tb1, a10, a11, a12, nsub, i0, i1, i2, i3, parent, p1_first; };
    auto load_desc = [&](const int q) -> ColDesc {
        const PlanWords w = col_desc + 32 * (size_t)q;
        ColDesc c;
        c.J = w[0]; c.t0 = w[1]; c.nc = w[2]; c.nbp = w[3]; c.ne = w[4]; c.e_rest = w[5];
        c.k0 = w[6]; c.tb0 = w[7]; c.a00 = w[8]; c.a01 = w[9]; c.a02 = w[10];
        c.k1 = w[11]; c.tb1 = w[12]; c.a10 = w[13]; c.a11 = w[14]; c.a12 = w[15];
        c.nsub = w[16]; c.i0 = w[17]; c.i1 = w[18]; c.i2 = w[19]; c.i3 = w[20]; c.parent = w[21]; c.p1_first = w[23];
        return c;
    };
    // The factorization reads and writes the workspace through BUFFER instructions: a resource descriptor per array (four SGPRs), a
    // scalar byte offset per tile / vector block and ONE 32-bit VGPR offset per lane layout - instead of a 64-bit VGPR address per
    // access.  (With flat addresses the column loop ran out of registers, and a single address spilled and reloaded between the
    // prefetch and the elimination made the wave wait for the prefetch it was meant to overlap.)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(atiles, 0, (int)0xfffffffe, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(tiles, 0, (int)0xfffffffe, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(vec, 0, (int)0xfffffffe, 0x00020000);
    auto bld = [](const __amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAux));
    };
    auto bst = [](const double x, const __amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), r, voff, soff, 0);
    };
    const unsigned bo_ab = 8u * (unsigned)((r16 << 4) + kq);   // bytes: operand layout of the matrix cores, element kk at + 32 kk
    const unsigned bo_c = 8u * (unsigned)((kq << 4) + r16);    // accumulator layout: element r at + 512 r
    const unsigned bo_k = 8u * (unsigned)kq, bo_r = 8u * (unsigned)r16;
    const unsigned so_inv = (unsigned)(10 * vs) * 8u, so_w = (unsigned)(11 * vs) * 8u, so_D = (unsigned)(7 * vs) * 8u, so_step = (unsigned)(6 * vs) * 8u;   // byte offsets of the vectors in rV
    // operands of one update entry in the matrix cores' layouts: B = U(J,k) (also the A operand of the diagonal tile), 1/d, w_k and
    // the tiles (I_i, k) of the carried rows
    struct UpdB { double raw[4], iv[4], wk[4]; };
    struct UpdY { double y0[4], y1[4], y2[4]; };
    auto load_b = [&](const uint32_t k, const uint32_t tb, UpdB &o) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            o.raw[kk] = bld(rU, bo_ab + 32u * kk, tb << 11);
            o.iv[kk] = bld(rV, bo_k + 32u * kk, so_inv + (k << 7));
            o.wk[kk] = bld(rV, bo_k + 32u * kk, so_w + (k << 7));
        }
    };
    auto load_y = [&](const uint32_t a0, const uint32_t a1, const uint32_t a2, UpdY &o) {
        if (a0 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) o.y0[kk] = bld(rU, bo_ab + 32u * kk, a0 << 11);
        }
        if (a1 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) o.y1[kk] = bld(rU, bo_ab + 32u * kk, a1 << 11);
        }
        if (a2 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) o.y2[kk] = bld(rU, bo_ab + 32u * kk, a2 << 11);
        }
    };
    struct ColAcc { f64x4 cD, cS0, cS1, cS2; double wacc; };
    auto apply_ops = [&](const uint32_t a0, const uint32_t a1, const uint32_t a2, const UpdB &o, const UpdY &y, ColAcc &c) {
        double bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { bv[kk] = -(o.raw[kk] * o.iv[kk]); c.wacc = fma(bv[kk], o.wk[kk], c.wacc); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c.cD = __builtin_amdgcn_mfma_f64_16x16x4f64(o.raw[kk], bv[kk], c.cD, 0, 0, 0);
        if (a0 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c.cS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y0[kk], bv[kk], c.cS0, 0, 0, 0);
        }
        if (a1 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c.cS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y1[kk], bv[kk], c.cS1, 0, 0, 0);
        }
        if (a2 != kNone) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c.cS2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y2[kk], bv[kk], c.cS2, 0, 0, 0);
        }
    };
    // what a column task loads before it can start: A's tiles of the column (accumulator layout: row 4 r + kq, column r16), w_J, the
    // LM diagonal of this lane's row, the operands of its first update entry
    // (two halves: A's tiles, w_J and the LM diagonal do not depend on the column's children - they are requested BEFORE the wave
    // waits for them; the operands of the first update entry are the children's rows)
    struct ColPre { f64x4 cD, cS0, cS1, cS2; double wj, dd; UpdB o0; UpdY y0; };
    auto issue_static = [&](const ColDesc &c, ColPre &p) {
        const unsigned so = c.t0 << 11;
#pragma unroll
        for (int r = 0; r < 4; ++r) p.cD[r] = bld(rA, bo_c + 512u * r, so);
        p.cS0 = f64x4{0.0, 0.0, 0.0, 0.0}; p.cS1 = p.cS0; p.cS2 = p.cS0;
        if (c.nc > 0u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.cS0[r] = bld(rA, bo_c + 512u * r, so + 2048u);
        }
        if (c.nc > 1u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.cS1[r] = bld(rA, bo_c + 512u * r, so + 4096u);
        }
        if (c.nc > 2u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.cS2[r] = bld(rA, bo_c + 512u * r, so + 6144u);
        }
        p.wj = bld(rV, bo_r, so_w + (c.J << 7));
        p.dd = bld(rV, bo_r, so_D + (c.J << 7));
    };
    auto issue_ops = [&](const ColDesc &c, ColPre &p) {
        if (c.ne > 0u) { load_b(c.k0, c.tb0, p.o0); load_y(c.a00, c.a01, c.a02, p.y0); }
    };

    // ---- "thin" plans (every column carries all its tiles): no barrier per level.  Every column has a STATE word (ts.pend[J] in LDS; a
    //      team's words in HBM): 0 = not factored yet, 1 = factored (its rows are in the workspace), 2 = solved (back substitution).  The
    //      waves walk their columns in level order - dependencies always point to lower levels, so some wave can always proceed.
    //      Round 5: a column does not wait for "all my children are done" any more.  Its left-looking update has one ENTRY per
    //      descendant that touches it - 5 to 20 entries for the columns near the root, applied in a fixed order (k ascending) - and
    //      almost all of those descendants were finished long before the last child is: the column task streams through its entries,
    //      each gated by the state of ITS column k only (one poll refreshes the states of up to 64 entries, a lane each), so that
    //      when the last child reports, what is left is that child's entry, the turn and the elimination.  Before, the whole chain
    //      "wait for the children, then one dependent round trip to the workspace per entry" sat on the critical path: half of the
    //      factorization's time for a 2.4 k-row component. ----
    const bool thin = pl[28] != 0u;
    // (a miscount must not hang the GPU: the step is rejected, the counter says so; a team gives up as a whole - the abort word ends
    // every wait of the launch's teams and fails their components)
    bool wdead = false;                                   // this wave has run out of patience once (TEAM: or has seen the abort word)
    auto spin_timeout = [&]() {
        if (lane == 0) {
            sh.flag = 1;
            if constexpr (TEAM) { if (team_ld(tm.ctl + 9) == 0u) atomicAdd(a.queue + 15, 1u); team_st(tm.ctl + 9, 1u); team_st(gteam, 1u); }
            else atomicAdd(a.queue + 15, 1u);
        }
        wdead = true;                                     // (this wave gives up waiting for the rest of the solve: one timeout must not cascade into seconds)
    };
    auto state_lane = [&](const uint32_t J) -> int {       // per lane (J may differ between lanes)
        if constexpr (TEAM) return (int)team_ld(gteam + 16 + J);
        else return __hip_atomic_load(&ts.pend[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto pend_load = [&](const uint32_t J) -> int { return __builtin_amdgcn_readfirstlane(state_lane(J)); };
    auto state_store = [&](const uint32_t J, const int v) {
        if constexpr (TEAM) team_st(gteam + 16 + J, (unsigned)v);
        else __hip_atomic_store(&ts.pend[J], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // wait until `ready()`; TEAM: the abort word is looked at every 256 polls
    auto pend_wait = [&](auto ready) {
        if (wdead) return;
        int spins = 0;
        while (!ready()) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if constexpr (TEAM) { if ((spins & 255) == 0 && __builtin_amdgcn_readfirstlane((int)team_ld(tm.ctl + 9)) != 0) { wdead = true; sh.flag = 1; break; } }
            if (spins > (1 << 22)) { spin_timeout(); break; }
        }
    };
    // what a finished column / a waiting column needs around a state word: the column's rows have left the wave (TEAM: have reached
    // L2 - the reader's loads bypass its own L1), and the reader's loads are issued after the state was seen
    auto publish_fence = [&]() {
        if constexpr (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };
    auto observe_fence = [&]() {
        if constexpr (TEAM) asm volatile("" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // the gate of a column's update entries: entry i of the column (col_upd[e_first + i], column k_i) may be loaded once k_i is factored
    struct EntryGate { uint32_t e_first, ne, chunk, kl; unsigned long long ok; };
    auto gate_open = [&](const ColDesc &c) -> EntryGate { return EntryGate{c.e_rest - 2u, c.ne, kNone, kNone, 0ull}; };
    auto gate_wait = [&](EntryGate &g, const uint32_t i) {
        const uint32_t c = i >> 6;
        if (c != g.chunk) {                                // the columns of entries 64 c .. 64 c + 63, one per lane
            g.chunk = c; g.ok = 0ull;
            g.kl = (c << 6) + (uint32_t)lane < g.ne ? hdr[pl[26] + 5u * (g.e_first + (c << 6) + (uint32_t)lane)] : kNone;
        }
        if ((g.ok >> (i & 63u)) & 1ull) return;
        pend_wait([&]() {
            g.ok |= __ballot(g.kl == kNone || state_lane(g.kl) != 0);
            return ((g.ok >> (i & 63u)) & 1ull) != 0ull;
        });
        observe_fence();
    };

    // One column task: the updates of the diagonal tile, of w_J and of the carried tiles with the operands in `pn` (accumulators in the
    // matrix cores' layout), a turn through LDS into lane = row, the elimination, whole rows back to the workspace.  `next_ready`: the
    // loads of the wave's next column (descriptor `dn`) go out into `pn` once the accumulators have left their registers.
    auto run_column = [&](const ColDesc &dc, ColPre &pn, EntryGate &gate, const bool gated, const bool next_static, const bool next_ready, const ColDesc &dn, const bool finish_extra, double *xd, unsigned long long *fprof, unsigned long long (&ft_)[6], unsigned long long &ft0_) {
        (void)fprof; (void)ft_; (void)ft0_;
        ColAcc acc;
        acc.cD = pn.cD; acc.cS0 = pn.cS0; acc.cS1 = pn.cS1; acc.cS2 = pn.cS2; acc.wacc = 0.0;
        const double wj = pn.wj;
        {   // the scaled LM diagonal (A + S^-1 D^2 S^-1): this lane holds the diagonal entry of row r16 if r16 = 4 r + kq
            const double dd2 = (r16 & 3) == kq ? pn.dd * pn.dd : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.cD[r] += (r16 >> 2) == r ? dd2 : 0.0;
        }
#ifndef LFR_T_PINGPONG
#define LFR_T_PINGPONG 1
#endif
#if LFR_T_PINGPONG
        // Further entries (a column near the root of the tree has 5-20 of them: one per descendant that touches it).  Their operands are
        // all there once the children are done, and a load from the workspace is 1-2 us: with one entry at a time the root chain of a
        // 2.4 k-row component spent half of the factorization's critical path waiting for operands.  Two operand sets alternate - the
        // loop's own and the prefetch set, which is free until the next column's loads go out - so an entry's loads are in flight
        // while the previous entry goes through the matrix cores.
        {
            const uint32_t e_end = dc.e_rest - 2u + dc.ne;                   // (entries 1 .. ne - 1 are col_upd[e_rest - 1 .. e_end))
            uint32_t e = dc.e_rest - 1u;
            UpdB oB;
            UpdY yB;
            uint32_t b0 = kNone, b1 = kNone, b2 = kNone, c0 = kNone, c1 = kNone, c2 = kNone;
            const bool more = dc.ne > 1u;
            if (more) { if (gated) gate_wait(gate, 1u); b0 = dc.a10; b1 = dc.a11; b2 = dc.a12; load_b(dc.k1, dc.tb1, oB); load_y(b0, b1, b2, yB); }       // entry 1: its words came with the descriptor
            uint32_t ei = 1u;                                                // (index of the entry in set B within the column)
            if (dc.ne > 0u) apply_ops(dc.a00, dc.a01, dc.a02, pn.o0, pn.y0, acc);
            if (more) for (;;) {
                const bool n1 = e + 1u < e_end;                             // set B holds entry e
                if (n1) {
                    if (gated) gate_wait(gate, ei + 1u);
                    const uint32_t k = col_upd[5 * (e + 1u)], tb = col_upd[5 * (e + 1u) + 1];
                    c0 = col_upd[5 * (e + 1u) + 2]; c1 = col_upd[5 * (e + 1u) + 3]; c2 = col_upd[5 * (e + 1u) + 4];
                    load_b(k, tb, pn.o0); load_y(c0, c1, c2, pn.y0);
                }
                apply_ops(b0, b1, b2, oB, yB, acc);
                if (!n1) break;
                ++e; ++ei;                                                  // the prefetch set holds entry e
                const bool n2 = e + 1u < e_end;
                if (n2) {
                    if (gated) gate_wait(gate, ei + 1u);
                    const uint32_t k = col_upd[5 * (e + 1u)], tb = col_upd[5 * (e + 1u) + 1];
                    b0 = col_upd[5 * (e + 1u) + 2]; b1 = col_upd[5 * (e + 1u) + 3]; b2 = col_upd[5 * (e + 1u) + 4];
                    load_b(k, tb, oB); load_y(b0, b1, b2, yB);
                }
                apply_ops(c0, c1, c2, pn.o0, pn.y0, acc);
                if (!n2) break;
                ++e; ++ei;
            }
        }
#else
        if (dc.ne > 0u) apply_ops(dc.a00, dc.a01, dc.a02, pn.o0, pn.y0, acc);
        for (uint32_t e = dc.e_rest - 1u; e < dc.e_rest - 2u + dc.ne && dc.ne > 1u; ++e) {    // further entries (a separator's column has one per child)
            if (gated) gate_wait(gate, e - (dc.e_rest - 2u));
            const uint32_t k = col_upd[5 * e], tb = col_upd[5 * e + 1], a0 = col_upd[5 * e + 2], a1 = col_upd[5 * e + 3], a2 = col_upd[5 * e + 4];
            UpdB o;
            UpdY y;
            load_b(k, tb, o);
            load_y(a0, a1, a2, y);
            apply_ops(a0, a1, a2, o, y, acc);
        }
#endif
        double wacc = acc.wacc;
        wacc += __shfl_xor(wacc, 16, 64);
        wacc += __shfl_xor(wacc, 32, 64);
        FPROF_MARK(0);                            // 0: loads + left-looking updates
        TR(3, dc.J);
        // matrix-core layout (row 4 r + kq, column r16) -> lane = row
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *o = xd + (4 * r + kq) * 18 + r16;
            o[0] = acc.cD[r]; o[288] = acc.cS0[r]; o[576] = acc.cS1[r]; o[864] = acc.cS2[r];
        }
        if (kq == 0) xd[1152 + r16] = wj + wacc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (next_static) issue_static(dn, pn);    // the next column's loads (the accumulators' registers are free now): in flight during the elimination
        if (next_ready) issue_ops(dn, pn);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nc = (int)dc.nc, nbp = (int)dc.nbp;
        const int s = lane - 17;
        const bool is_diag = lane < 16, is_rhs = lane == 16, on = lane >= 17 && (s >> 4) < nc;
        double av[16];
        {
            const double *rs = is_diag ? xd + r16 * 18 : is_rhs ? xd + 1152 : xd + 288 * (1 + (s >> 4)) + (s & 15) * 18;
            const double2 *rowp = reinterpret_cast<const double2 *>(rs);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double2 v = rowp[j]; av[2 * j] = v.x; av[2 * j + 1] = v.y; }
#pragma unroll
            for (int j = 0; j < 16; ++j) av[j] = is_diag ? (j <= r16 ? av[j] : 0.0) : ((is_rhs || on) ? av[j] : 0.0);
        }
        FPROF_MARK(1);                            // 1: the turn through LDS
        bool bad = false;
        double my_inv = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (kk < nbp) {
                const double dk = readlane_f64(av[kk], kk);
                bad = bad || !(dk > 0.0);
                const double ik = fast_rcp(dk);
                my_inv = lane == kk ? ik : my_inv;
                const double lik = av[kk] * ik;
#pragma unroll
                for (int j = kk + 1; j < 16; ++j) av[j] = fma(-lik, readlane_f64(av[kk], j), av[j]);
            }
        }
        FPROF_MARK(2);                            // 2: elimination
        TR(4, dc.J);
        if (lane < nbp) bst(my_inv, rV, 8u * (unsigned)lane, so_inv + (dc.J << 7));
        if (is_diag || on) {                      // rows of the diagonal tile and of the carried tiles: whole rows, 16 bytes per store
            const unsigned ro = is_diag ? 8u * (unsigned)(r16 << 4) : 8u * (unsigned)(((1 + (s >> 4)) << 8) + ((s & 15) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__builtin_bit_cast(u32x2_t, av[2 * j])[0], __builtin_bit_cast(u32x2_t, av[2 * j])[1],
                                                                __builtin_bit_cast(u32x2_t, av[2 * j + 1])[0], __builtin_bit_cast(u32x2_t, av[2 * j + 1])[1]},
                                                       rU, ro + 16u * j, dc.t0 << 11, 0);
        }
        if (is_rhs) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__builtin_bit_cast(u32x2_t, av[2 * j])[0], __builtin_bit_cast(u32x2_t, av[2 * j])[1],
                                                                __builtin_bit_cast(u32x2_t, av[2 * j + 1])[0], __builtin_bit_cast(u32x2_t, av[2 * j + 1])[1]},
                                                       rV, 16u * j, so_w + (dc.J << 7), 0);
        }
        if (bad && lane == 0) { sh.flag = 1; if constexpr (TEAM) team_st(gteam, 1u); }
        FPROF_MARK(3);                            // 3: stores
        if (finish_extra && dc.nsub > dc.nc) {
            // Tiles below the diagonal beyond the carried ones (thin plans: a few per component): this wave finishes them itself, three at
            // a time - left-looking updates on the matrix cores, a turn through LDS, then the substitution against the factored diagonal
            // tile (staged in LDS with 1 / d), lane = row in lanes 16-63.
            __builtin_amdgcn_wave_barrier();
            if (is_diag) {
#pragma unroll
                for (int j = 0; j < 16; ++j) xd[(r16 << 4) + j] = av[j];
            }
            if (lane < 16) xd[1152 + lane] = my_inv;
            for (uint32_t i = dc.nc; i < dc.nsub; i += 3u) {
                const uint32_t cnt = min(3u, dc.nsub - i);
                __builtin_amdgcn_wave_barrier();
                for (uint32_t u = 0; u < cnt; ++u) {
                    const uint32_t tk = dc.p1_first + (i - dc.nc) + u;
                    const uint32_t tt = p1_tasks[4 * tk], ub = p1_tasks[4 * tk + 1], ue = p1_tasks[4 * tk + 2];
                    f64x4 c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = bld(rA, bo_c + 512u * r, tt << 11);
                    for (uint32_t e = ub; e < ue; ++e) {
                        const uint32_t ia = upd[3 * e], ib = upd[3 * e + 1], k = upd[3 * e + 2];
                        double aa[4], bb[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            aa[kk] = bld(rU, bo_ab + 32u * kk, ia << 11);
                            bb[kk] = -(bld(rU, bo_ab + 32u * kk, ib << 11) * bld(rV, bo_k + 32u * kk, so_inv + (k << 7)));
                        }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[kk], bb[kk], c, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) xd[288 * (1 + u) + (4 * r + kq) * 18 + r16] = c[r];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int s2 = lane - 16;
                const bool act = lane >= 16 && (uint32_t)(s2 >> 4) < cnt;
                double rr[16];
                {
                    const double2 *rowp = reinterpret_cast<const double2 *>(xd + 288 * (1 + (act ? (s2 >> 4) : 0)) + (s2 & 15) * 18);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const double2 v = rowp[j]; rr[2 * j] = v.x; rr[2 * j + 1] = v.y; }
                }
#pragma unroll
                for (int j = 0; j < 15; ++j) {
                    const double tj = rr[j] * xd[1152 + j];
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) rr[c] = fma(-tj, xd[(c << 4) + j], rr[c]);
                }
                if (act) {
                    const unsigned ro = 8u * (unsigned)((s2 & 15) << 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__builtin_bit_cast(u32x2_t, rr[2 * j])[0], __builtin_bit_cast(u32x2_t, rr[2 * j])[1],
                                                                        __builtin_bit_cast(u32x2_t, rr[2 * j + 1])[0], __builtin_bit_cast(u32x2_t, rr[2 * j + 1])[1]},
                                                               rU, ro + 16u * j, (dc.t0 + 1u + i + (uint32_t)(s2 >> 4)) << 11, 0);
                }
            }
        }
    };

    // this wave's next column after position q of level l (levels ascending / descending); false: none left
    // (the first column of level l goes to wave l mod GW: a chain of single-column levels - the blocks of a top separator - then
    // alternates between waves, and the parent's wave streams through its finished entries while the child is being eliminated)
#ifndef LFR_T_ROTATE
#define LFR_T_ROTATE 1
#endif
    auto first_of = [&](const int l) -> int { return LFR_T_ROTATE ? (gw - l % GW + GW) % GW : gw; };
    auto next_up = [&](int &l, int &q) -> bool {
        q += GW;
        while (q >= (int)level_ptr[l + 1]) { if (++l >= n_levels) return false; q = (int)level_ptr[l] + first_of(l); }
        return true;
    };
    auto next_down = [&](int &l, int &q) -> bool {
        q += GW;
        while (q >= (int)level_ptr[l + 1]) { if (--l < 0) return false; q = (int)level_ptr[l] + first_of(l); }
        return true;
    };
    auto factor_thin = [&]() -> bool {
        double *xd = ts.x[wave];
        unsigned long long *fprof = a.prof ? a.prof + 8 * lfr::KC_COUNT + 8 + 16 * (a.cls - lfr::KC_BLOCK) : nullptr;      // (-DLFR_PROFILE_FACTOR)
        (void)fprof;
        FPROF_DECL
        // (the columns' state words are zeroed by the pass that builds the LM diagonal, in front of its barrier)
        int l = 0, q = (int)level_ptr[0] + first_of(0) - GW;
        bool have = next_up(l, q), pre = false, pre_static = false;
        ColDesc dn;
        ColPre pn;
        if (have) dn = load_desc(q);
        while (have) {
            const ColDesc dc = dn;
#ifndef LFR_T_SPLIT_PREFETCH
#define LFR_T_SPLIT_PREFETCH 2      // 0: a column's loads go out together once its children are done (round 4); 1: the half that does not depend
                                    // on the children goes out early, for the wave's next column inside the current one; 2: ... AFTER the current
                                    // column has been published when the next one is not ready yet (the publishing `s_waitcnt vmcnt(0)` waits for
                                    // every load in flight: a prefetch in front of it delayed the hand-off along the critical path)
#endif
            EntryGate gate = gate_open(dc);
            TR(1, dc.J | (pre ? 0x10000u : 0u) | (dc.ne << 20));
            if (!pre) {                                   // not prefetched: wait for the column of the first entry, then load its rows
                if (LFR_T_SPLIT_PREFETCH && !pre_static) issue_static(dc, pn);
                if (dc.ne > 0u) gate_wait(gate, 0u);
                if (!LFR_T_SPLIT_PREFETCH) issue_static(dc, pn);
                issue_ops(dc, pn);
            }
            TR(2, dc.J);
            FPROF_MARK(5);                                // 5: waiting for children
            have = next_up(l, q);
            bool next_ready = false;
            if (have) {
                dn = load_desc(q);
                next_ready = dn.ne == 0u || pend_load(dn.k0) != 0;          // (its first entry's column is factored: that entry's rows can be requested)
                if (next_ready) observe_fence();
            }
            run_column(dc, pn, gate, true, LFR_T_SPLIT_PREFETCH == 1 ? have : next_ready, next_ready, dn, true, xd, fprof, ft_, ft0_);
            pre = next_ready; pre_static = LFR_T_SPLIT_PREFETCH == 1 ? have : next_ready;
            publish_fence();                                               // the column's rows are out before anyone hears of it
            if (lane == 0) state_store(dc.J, 1);
            TR(5, dc.J);
            if (LFR_T_SPLIT_PREFETCH == 2 && have && !next_ready) { issue_static(dn, pn); pre_static = true; }
        }
        tsync();
        if constexpr (TEAM) {                             // a bad pivot anywhere in the team rejects the step for everyone
            if (tid == 0 && team_ld(gteam) != 0u) sh.flag = 1;
            __syncthreads();
        }
        FPROF_FLUSH();
        return sh.flag == 0;
    };

    auto factor = [&]() -> bool {
        if (thin || TEAM) return factor_thin();           // (a team only ever takes thin plans: solve_tree_kernel)
        double *xd = ts.x[wave];                          // (sh.flag was reset before the caller's last barrier)
        unsigned long long *fprof = a.prof ? a.prof + 8 * lfr::KC_COUNT + 8 + 16 * (a.cls - lfr::KC_BLOCK) : nullptr;      // (-DLFR_PROFILE_FACTOR)
        (void)fprof;
        FPROF_DECL
        for (int l = 0; l < n_levels; ++l) {
            // (a) tiles their columns do not carry (columns with more than three tiles below the diagonal): left-looking update by
            // one wave per tile, A(I,J) - sum_k U(I,k) D_k^-1 U(J,k)^T on the fp64 matrix cores
            const int u0 = (int)p1_ptr[l], u1 = (int)p1_ptr[l + 1];
            for (int t = u0 + wave; t < u1; t += kWaves) {
                const uint32_t tt = p1_tasks[4 * t], ub = p1_tasks[4 * t + 1], ue = p1_tasks[4 * t + 2];
                f64x4 c;
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = bld(rA, bo_c + 512u * r, tt << 11);
                for (uint32_t u = ub; u < ue; ++u) {
                    const uint32_t ia = upd[3 * u], ib = upd[3 * u + 1], k = upd[3 * u + 2];
                    double av[4], bv[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        av[kk] = bld(rU, bo_ab + 32u * kk, ia << 11);
                        bv[kk] = -(bld(rU, bo_ab + 32u * kk, ib << 11) * bld(rV, bo_k + 32u * kk, so_inv + (k << 7)));
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) bst(c[r], rU, bo_c + 512u * r, tt << 11);
            }
            FPROF_MARK(4);                                // 4: tile tasks
            // (b) the level's columns, one wave each, software pipelined: while a column is turned and eliminated the loads of the
            // wave's NEXT column are in flight (a load from the workspace is ~1000-2000 cycles: the components of a launch do not fit L2).
            //   the updates of the diagonal tile, of w_J and of the carried tiles (accumulators in the matrix cores' layout), a turn
            //   through LDS into lane = row, the elimination, whole rows back to the workspace
            const int q1 = (int)level_ptr[l + 1];
            int q = (int)level_ptr[l] + wave;
            ColDesc dn;
            ColPre pn;
            if (q < q1) { dn = load_desc(q); issue_static(dn, pn); issue_ops(dn, pn); }
            while (q < q1) {
                const ColDesc dc = dn;
                const int qn = q + kWaves;
                if (qn < q1) dn = load_desc(qn);          // (scalar loads: on their way during the updates below)
                EntryGate gate = gate_open(dc);
                run_column(dc, pn, gate, false, qn < q1, qn < q1, dn, false, xd, fprof, ft_, ft0_);
                q = qn;
            }
            __syncthreads();
            FPROF_MARK(5);                                // 5: waiting at the level's barrier
            TPROF_MARK(1);
            // (c) rows the column tasks did not carry: substituted against the finished diagonal tile, four tiles per wave
            const int x0 = (int)x_ptr[l], x1 = (int)x_ptr[l + 1];
            if (x1 > x0) {
                for (int t = x0 + wave; t < x1; t += kWaves) {
                    const int J = (int)x_tasks[3 * t], i0 = (int)x_tasks[3 * t + 1], cnt = (int)x_tasks[3 * t + 2];
                    const uint32_t t0 = colptr[J];
                    const double *Td = tiles + ((size_t)t0 << 8);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 4; ++i) xd[lane + 64 * i] = Td[lane + 64 * i];
                    if (lane < 16) xd[1152 + lane] = vinv[16 * (size_t)J + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const bool act = kq < cnt;
                    double *rowp = tiles + ((size_t)(t0 + 1u + (uint32_t)i0 + (uint32_t)(act ? kq : 0)) << 8) + (r16 << 4);
                    double r[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) r[c] = rowp[c];
#pragma unroll
                    for (int j = 0; j < 15; ++j) {
                        const double tj = r[j] * xd[1152 + j];
#pragma unroll
                        for (int c = j + 1; c < 16; ++c) r[c] = fma(-tj, xd[(c << 4) + j], r[c]);
                    }
                    if (act) {
#pragma unroll
                        for (int c = 1; c < 16; ++c) rowp[c] = r[c];
                    }
                }
                __syncthreads();
                TPROF_MARK(5);
            }
        }
        FPROF_FLUSH();
        return sh.flag == 0;
    };

    // ---- L D L^T y = g with w = L^-1 g: levels top down; y_J = D^-1 (w_J - sum_I U(I,J)^T y_I) through the diagonal tile.  Pipelined
    //      like the factorization: the next column's tiles are on their way while the 16 dependent steps of this one run. ----
    struct BackPre { double m[16], iv, z, tvv[4][4], yv[4][4]; };
    // (two halves again: the column's own tiles, 1/d and w_J are final since its factorization - requested before the wave waits for
    // the parent; y of the rows below is what the parent's solution releases)
    auto issue_back_static = [&](const ColDesc &c, BackPre &p) {
        const unsigned so = c.t0 << 11;
#pragma unroll
        for (int k = 0; k < 16; ++k) p.m[k] = bld(rU, bo_r + 128u * k, so);
        p.iv = r16 < (int)c.nbp ? bld(rV, bo_r, so_inv + (c.J << 7)) : 0.0;
        p.z = bld(rV, bo_r, so_w + (c.J << 7));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((uint32_t)i < c.nsub) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.tvv[i][r] = bld(rU, bo_c + 512u * r, so + 2048u * (1 + i));
            }
        }
    };
    auto issue_back_y = [&](const ColDesc &c, BackPre &p) {
        const uint32_t rows[4] = {c.i0, c.i1, c.i2, c.i3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((uint32_t)i < c.nsub) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.yv[i][r] = bld(rV, bo_k + 32u * r, so_step + (rows[i] << 7));
            }
        }
    };
    // one column of the back substitution with its operands in `pn`; the next column's loads go out before the 16 dependent steps
    auto run_back = [&](const ColDesc &dc, BackPre &pn, const bool next_static, const bool next_ready, const ColDesc &dn) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((uint32_t)i < dc.nsub) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = fma(pn.tvv[i][r], pn.yv[i][r], acc);
            }
        }
        for (uint32_t t = dc.t0 + 5u; t < dc.t0 + 1u + dc.nsub; ++t) {          // further tiles (a column with more than four rows below the diagonal)
            const uint32_t I = rowsof[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fma(bld(rU, bo_c + 512u * r, t << 11), bld(rV, bo_k + 32u * r, so_step + (I << 7)), acc);
        }
        double m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = pn.m[k];
        const double iv = pn.iv;
        double z = pn.z;
        if (next_static) issue_back_static(dn, pn);
        if (next_ready) issue_back_y(dn, pn);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        z -= acc;
        double yo = 0.0;
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            const double yvk = z * iv;
            const double yk = readlane_f64(yvk, k);
            yo = (r16 == k) ? yvk : yo;
            z = fma(-((r16 < k) ? m[k] : 0.0), yk, z);
        }
        if (lane < 16) bst(yo, rV, bo_r, so_step + (dc.J << 7));
    };
    auto back_substitute = [&]() {
        if (thin || TEAM) {       // no barrier per level: a column starts when its parent is solved (state 2; every column is in state 1 after the factorization)
            int l = n_levels - 1, q = (int)level_ptr[l] + first_of(l) - GW;
            bool have = next_down(l, q), pre = false, pre_static = false;
            ColDesc dn;
            BackPre pn;
            if (have) dn = load_desc(q);
            while (have) {
                const ColDesc dc = dn;
                TR(6, dc.J | (pre ? 0x10000u : 0u));
                if (!pre) {
                    if (LFR_T_SPLIT_PREFETCH && !pre_static) issue_back_static(dc, pn);
                    pend_wait([&]() { return dc.parent == kNone || pend_load(dc.parent) == 2; });
                    observe_fence();
                    if (!LFR_T_SPLIT_PREFETCH) issue_back_static(dc, pn);
                    issue_back_y(dc, pn);
                }
                TR(7, dc.J);
                have = next_down(l, q);
                bool next_ready = false;
                if (have) {
                    dn = load_desc(q);
                    next_ready = dn.parent == kNone || pend_load(dn.parent) == 2;
                    if (next_ready) observe_fence();
                }
                run_back(dc, pn, LFR_T_SPLIT_PREFETCH == 1 ? have : next_ready, next_ready, dn);
                pre = next_ready; pre_static = LFR_T_SPLIT_PREFETCH == 1 ? have : next_ready;
                publish_fence();
                if (lane == 0) state_store(dc.J, 2);
                TR(8, dc.J);
                if (LFR_T_SPLIT_PREFETCH == 2 && have && !next_ready) { issue_back_static(dn, pn); pre_static = true; }
            }
            tsync();
            return;
        }
        for (int l = n_levels - 1; l >= 0; --l) {
            const int q1 = (int)level_ptr[l + 1];
            int q = (int)level_ptr[l] + wave;
            ColDesc dn;
            BackPre pn;
            if (q < q1) { dn = load_desc(q); issue_back_static(dn, pn); issue_back_y(dn, pn); }
            while (q < q1) {
                const ColDesc dc = dn;
                const int qn = q + kWaves;
                if (qn < q1) dn = load_desc(qn);
                run_back(dc, pn, qn < q1, qn < q1, dn);
                q = qn;
            }
            __syncthreads();
        }
    };

    // ---- the trust-region loop of solve_component, over vectors in matrix order.  Round 5: an iteration of a large component is a
    //      chain of round trips to L2 and of barriers, not arithmetic, so the vector passes are gone except two: the trial point, the
    //      step / point norms, the projected gradient and the line search's directional derivative come out of the sweep (above),
    //      an accepted candidate is a swap of pointers, the Jacobi scaling and the dependency counters ride in the pass that builds
    //      the LM diagonal.  Per iteration: that pass, the factorization, the back substitution, one pass + reduction for the
    //      model's cost change, and two reductions per sweep. ----
    // (-DLFR_PROFILE_PHASES: 0 sweeps, 1 factorization, 5 scaling, 6 back substitution, 4 everything else - the slots of the LDS kernels)
    int exec_passes = 1;
    SweepOut sw = sweep(vx, vdelta, 0.0, false, vxc, vg);
    double cost = uni(sw.cost);                       // (uni: the loop's scalars are replicated - SGPRs, like the LDS kernel's)
    PROF_MARK(0);
    double gmax = uni(sw.gm);
    double x_norm = 0.0, radius = kInitialRadius, decrease_factor = 2.0;
    bool reuse_diagonal = false, step_successful = true, matrix_valid = true;
    int n_invalid = 0, iteration = 0, term = LFR_TERM_CONVERGENCE;
    int n_successful = 0, n_ls_evals = 0, n_cand = 0;
    for (;;) {
        if (iteration >= kMaxIterations) { term = LFR_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= kGradientTol) break;
        if (radius <= kMinRadius) break;
        ++iteration;
        TR_IT(iteration);
        TR(10, 0);
        step_successful = false;
        PROF_MARK(4);
        if (!matrix_valid) {           // A holds J^T J of a rejected trial point: re-assemble at x
            sweep(vx, vdelta, 0.0, false, vxc, vg);
            ++exec_passes;
            matrix_valid = true;
            PROF_MARK(0);
        }
        // the damped system (A + S^-1 D^2 S^-1) (S y) = g: the scaled LM diagonal onto the diagonal tiles, g into w.
        // (The vector passes of this loop take FOUR elements per thread and step, every load of the step issued before the first use: a
        // load from the workspace is ~1-2 k cycles, and an element at a time each pass paid that once per element - index n is the
        // vectors' zero slot, so an index past the end reads zeros and stores nothing.)
        for (int i0 = gt; i0 < n; i0 += 4 * GT) {
            double sc[4], dg[4], ad[4], gg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * GT, n); sc[u] = ldd(vscale + i); dg[u] = ldd(vdiag + i); ad[u] = ldd(vadiag + i); gg[u] = ldd(vg + i); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * GT;
                if (i >= n) break;
                if (iteration == 1) { sc[u] = 1.0 / (1.0 + sqrt(ad[u])); vscale[i] = sc[u]; }       // the Jacobi scaling: fixed at the initial point
                if (!reuse_diagonal) { dg[u] = fmin(fmax(sc[u] * sc[u] * ad[u], kMinLmDiag), kMaxLmDiag); vdiag[i] = dg[u]; }
                vD[i] = sqrt(dg[u] / radius) / sc[u];                        // D / s (the column tasks add D^2 to their diagonal tiles)
                vw[i] = gg[u];
            }
        }
        reuse_diagonal = true;
        if (thin || TEAM) {            // the columns' state words of the barrier-free schedule: nothing factored yet
            if constexpr (TEAM) {
                for (int q = gt; q < NB; q += GT) team_st(gteam + 16 + q, 0u);
            } else {
                for (int q = tid; q < NB; q += kBlockThreads) ts.pend[q] = 0;
            }
        }
        if (tid == 0) { sh.flag = 0; if constexpr (TEAM) { if (tm.r == 0) team_st(gteam, 0u); } }     // raised by a non-positive pivot
        tsync();
        TR(11, 0);
        PROF_MARK(LFR_TREE_SLOT_SCALE);
        bool valid = factor();                            // (reads A, writes the factor: J^T J at x stays in A until a trial point is swept)
        TR(12, 0);
        PROF_MARK(1);
        if (valid) back_substitute();
        TR(13, 0);
        PROF_MARK(6);
        double model_cost_change = 0.0, g_dot_delta = 0.0, dir_max = 0.0;
        if (valid) {
            double partial = 0.0, gd_part = 0.0, zero = 0.0, dm_part = 0.0;
            for (int i0 = gt; i0 < n; i0 += 4 * GT) {
                double gi[4], Di[4], st[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * GT, n); gi[u] = ldd(vg + i); Di[u] = ldd(vD + i); st[u] = ldd(vstep + i); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * GT;
                    if (i >= n) break;
                    const double dl = -st[u];
                    vdelta[i] = dl;
                    partial += -gi[u] * dl + Di[u] * Di[u] * dl * dl;
                    gd_part += gi[u] * dl;
                    dm_part = isfinite(dl) ? fmax(dm_part, fabs(dl)) : INFINITY;
                }
            }
            treduce4(partial, gd_part, zero, dm_part);
            TR(14, 0);
            model_cost_change = uni(0.5 * partial); g_dot_delta = uni(gd_part); dir_max = uni(dm_part);
            valid = isfinite(dir_max) && model_cost_change > 0.0;
        }
        if (!valid) {
            // (a team's bad-pivot word is reset by the leader at the top of the next iteration; every member must have
            // read it first - no sweep or reduction, hence no team barrier, lies on this path otherwise)
            if constexpr (TEAM) tsync();
            if (++n_invalid >= kMaxInvalid) { term = LFR_TERM_FAILURE; break; }
            radius = uni(radius / decrease_factor);
            decrease_factor = uni(decrease_factor * 2.0);
            continue;
        }
        n_invalid = 0;
        // ---- projected Armijo line search ----
        double alpha = 1.0, cost_c = 0.0;
        bool ls_ok = false;
        {
            LsSample initial{0.0, cost, g_dot_delta, true, true}, previous{0, 0, 0, false, false}, current;
            int n_iter = 0;
            for (;;) {
                PROF_MARK(4);
                sw = sweep(vx, vdelta, alpha, true, vxc, vgn);             // also assembles J^T J at the trial point
                cost_c = uni(sw.cost);
                matrix_valid = false;
                PROF_MARK(0);
                ++exec_passes; ++n_ls_evals;
                current.x = alpha; current.value = cost_c; current.value_valid = isfinite(cost_c);
                current.gradient = 0.0; current.gradient_valid = false;
                if (current.value_valid && !(cost_c > cost + kLsSufficientDecrease * g_dot_delta * alpha)) { ls_ok = true; break; }
                if (current.value_valid) {
                    current.gradient = uni(sw.gd);                 // delta . g(trial point), summed by the sweep's node pass
                    current.gradient_valid = isfinite(current.gradient);
                }
                TR(15, 0);
                // The next trial step is a pure function of replicated scalars - 10-40 k cycles of polynomial fitting and root
                // isolation (9-18 us when all eight waves of the CU run it side by side: two waves per SIMD share the issue port).
                // One wave computes it, the others wait at the barrier; every workgroup of a team does the same and gets the same bits.
                double nstep;
                {
                    int n_iter_w0 = n_iter;
                    if (wave == 0) {
                        const double v = ls_next_step_wave(initial, previous, current, dir_max, n_iter_w0);
                        if (lane == 0) sh.bcast[0] = v;
                    }
                    __syncthreads();
                    nstep = uni(sh.bcast[0]);
                    ++n_iter;                                    // (= what ls_next_step_regs did to wave 0's copy)
                    __syncthreads();
                }
                TR(16, 0);
                if (nstep < 0.0) break;
                previous = current;
                alpha = nstep;
            }
        }
        if (!ls_ok) {
            sw = sweep(vx, vdelta, 1.0, true, vxc, vgn);
            cost_c = uni(sw.cost);
            matrix_valid = false;
            ++exec_passes;
        }
        ++n_cand;
        const double cost_cand = uni(isfinite(cost_c) ? cost_c : DBL_MAX);
        // step norm, and - should the candidate be accepted - its norm and projected gradient: from the sweep
        const double step_norm = sqrt(sw.sn);
        if (step_norm <= kParameterTol * (x_norm + kParameterTol)) break;
        const double cost_change = cost - cost_cand;
        if (fabs(cost_change) <= kFunctionTol * cost) break;
        const double rel = cost_change / model_cost_change;
        if (rel > kMinRelDecrease) {
            { double *t = vx; vx = vxc; vxc = t; t = vg; vg = vgn; vgn = t; }      // the candidate becomes x, its gradient g
            x_norm = uni(sqrt(sw.xn));
            cost = cost_cand;
            matrix_valid = true;              // the accepted candidate is the last evaluated point: its J^T J is in the tiles
            gmax = uni(sw.gm);
            step_successful = true;
            ++n_successful;
            const double t = 2.0 * rel - 1.0;
            radius = uni(fmin(kMaxRadius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t)));
            decrease_factor = 2.0;
            reuse_diagonal = false;
        } else {
            radius = uni(radius / decrease_factor);
            decrease_factor = uni(decrease_factor * 2.0);
        }
    }
    PROF_MARK(4);
    PROF_FLUSH();
    tsync();
    if constexpr (TEAM) { if (team_ld(tm.ctl + 9) != 0u) term = LFR_TERM_FAILURE; }      // a wait of this launch's teams gave up: nothing computed since can be trusted
    for (int p = gt; p < 8 * NB; p += GT) {
        const uint32_t v = ipos[p];
        if (v == kNone) continue;
        double *out = a.positions + 2 * (size_t)a.node_ids[d.node_off + v];
        out[0] = term != LFR_TERM_FAILURE ? ldd(vx + 2 * p) : 0.0;
        out[1] = term != LFR_TERM_FAILURE ? ldd(vx + 2 * p + 1) : 0.0;
    }
    if (tid == 0 && (!TEAM || tm.r == 0)) {
        CompInfoDev inf;
        inf.iterations = iteration; inf.termination = term; inf.n_successful = n_successful;
        inf.n_ls_evals = n_ls_evals; inf.n_cand_evals = n_cand; inf.exec_passes = exec_passes;
        inf.final_cost = cost;
#if defined(LFR_PROFILE_WGTIME) && LFR_PROFILE_WGTIME == 3      // diagnostic builds (scripts/sparse_timeline.py): start (100 MHz ticks) in the cost, lifetime << 4 | team size in the termination
        inf.final_cost = (double)tree_r0_;
        inf.termination = (int)(((wall_clock64() - tree_r0_) << 4) | (unsigned long long)tS);
        inf.iterations = iteration | (min(exec_passes, 1023) << 8) | (min(n_successful, 255) << 18);
#endif
        a.infos[ci] = inf;
    }
}

// persistent workgroups over the class's queue, like solve_block_kernel
template <int kBlockThreads>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu((kBlockThreads + 255) / 256, (kBlockThreads + 255) / 256))) void solve_tree_kernel(const KernelArgs a) {
    __shared__ BlockShared sh;
    __shared__ TreeShared ts;
    __shared__ int next_ci;
    for (;;) {
        if (threadIdx.x == 0) {
            const int k = a.desc_begin + (int)atomicAdd(a.queue + a.cls, 1u);
            next_ci = k < a.desc_end ? (int)a.wg_order[k - a.wg_begin] : -1;
        }
        __syncthreads();
        const int ci = __builtin_amdgcn_readfirstlane(next_ci);
        __syncthreads();
        if (ci < 0) break;
        TeamCtx tm;
        solve_tree_component<kBlockThreads, false>(a, ci, sh, ts, tm);
        __syncthreads();
    }
}

// The same with TEAMS (the comment above TeamCtx).  Workgroups register with their XCC id; kTeamMax consecutive registrations of one XCD
// form a UNIT whose members share an L2.  A unit starts as one team led by its rank 0.  The leader of a team takes the next component
// from the class's queue - handed out by expected work, descending, and the wanted team size is a monotone function of that key - and
// tells its members {component, team size} through their mailboxes; when the component wants a smaller team than the current one the
// team splits into equal sub-teams for good (the first keeps the component, the leaders of the others go to the queue themselves).
// A plan that is not "thin" (dense components: barrier schedule) is solved by the leader alone while its members wait.
// Workgroups that cannot complete a unit (their XCD received no multiple of kTeamMax of them) work as teams of one.
// One thread, every 256 polls of a wait that only somebody else can end: has the launch stalled for good?  Nobody at work and neither a
// registration nor the queue's head moved for the launch's patience (wall clock: s_memrealtime counts at 100 MHz) => SOLO mode.
struct TeamWatch {
    unsigned long long t_last = 0ull, sig_last = ~0ull;
    __device__ __forceinline__ void look(const KernelArgs &a, unsigned int *ctl) {
        const unsigned long long sig = (unsigned long long)team_ld(ctl + 8) << 32 | team_ld(a.queue + a.cls);
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (sig != sig_last || team_ld(ctl + 12) != 0u) { sig_last = sig; t_last = now; return; }
        if (now - t_last > 100ull * a.team_patience_us) team_st(ctl + 11, 1u);
    }
};
__device__ __forceinline__ int tree_team_size(const KernelArgs &a, const int ci) {
    const CompDesc d = a.descs[ci];
    const uint32_t work = (uint32_t)d.n_var * (1u + ((uint32_t)d.n_nodes - d.n_var > 1u ? 1u : 0u));      // = k_wg_order_keys
    int t = 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) if ((2 << k) <= kTeamMax && work >= a.team_work[k]) t = 2 << k;
    return t;
}
template <int kBlockThreads>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu((kBlockThreads + 255) / 256, (kBlockThreads + 255) / 256))) void solve_tree_team_kernel(const KernelArgs a) {
    __shared__ BlockShared sh;
    __shared__ TreeShared ts;
    __shared__ unsigned int bc[4];
    const int tid = threadIdx.x;
    unsigned int *const ctl = a.team_ctl;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;             // XCC_ID [3:0]
        const unsigned slot = __hip_atomic_fetch_add(ctl + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the XCD's count is out before the total says "everyone has registered")
        __hip_atomic_fetch_add(ctl + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned unit = slot / kTeamMax;
        // A unit is complete when kTeamMax workgroups of this XCD hold its slots, and dissolved when that cannot happen (the whole grid
        // has registered and the XCD's count is short) or the launch has gone SOLO.  Its members must agree, so the unit's state word
        // is decided once, by compare-and-swap: "complete" by the holder of the unit's last slot (slots are handed out in order: the
        // other kTeamMax - 1 registered before it), "dissolved" by whichever waiter finds a reason first.
        unsigned state = 2u;
        if (unit < (unsigned)kTeamUnitsPerXcc) {
            unsigned int *const us = ctl + kTeamUnitState + xcc * kTeamUnitsPerXcc + unit;
            auto decide = [&](unsigned v) { unsigned expect = 0u; __hip_atomic_compare_exchange_strong(us, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            if (slot % kTeamMax == kTeamMax - 1) decide(1u);
            TeamWatch watch;
            int spins = 0;
            while ((state = team_ld(us)) == 0u) {
                if (team_ld(ctl + 11) != 0u) { decide(2u); continue; }
                if (team_ld(ctl + 8) >= gridDim.x && team_ld(ctl + xcc) < (unit + 1u) * kTeamMax) { decide(2u); continue; }
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 255) == 0) {
                    if (team_ld(ctl + 9) != 0u) { state = 2u; break; }
                    watch.look(a, ctl);
                }
            }
        }
        const bool complete = state == 1u;
        bc[0] = xcc * kTeamUnitsPerXcc + (complete ? unit : 0u); bc[1] = slot % kTeamMax; bc[2] = complete ? (unsigned)kTeamMax : 1u;
    }
    __syncthreads();
    const unsigned unit_slot = __builtin_amdgcn_readfirstlane(bc[0]);
    const int rank = (int)__builtin_amdgcn_readfirstlane(bc[1]);
    int S = (int)__builtin_amdgcn_readfirstlane(bc[2]);
    int L = S == 1 ? rank : 0;                            // leader of my team: ranks [L, L + S) of the unit
    unsigned int *const mbox = ctl + 16 + 16 * unit_slot, *const bars = mbox + 8;
    TeamCtx tm;
    tm.ctl = ctl;
    tm.epoch = a.team_epoch;
    __syncthreads();
    for (;;) {
        if (tid == 0) {
            unsigned msg = kTeamMsgEnd;
            if (rank == L) {
                // A component is solved by EXACTLY the team size its key asks for (its results are a function of that size): a team
                // takes the head of the queue only if it is large enough - it splits when it is larger - and waits otherwise for a
                // larger team to take it (the queue hands out by key, descending: whoever has split has seen the last component of
                // its former size leave).  Look, then pop by compare-and-swap.
                int spins = 0;
                TeamWatch watch;
                for (;;) {
                    const unsigned head = team_ld(a.queue + a.cls);
                    const int k = a.desc_begin + (int)head;
                    if (k >= a.desc_end || team_ld(ctl + 9) != 0u) break;
                    const int ci = (int)a.wg_order[k - a.wg_begin];
                    const bool thin = reinterpret_cast<const uint32_t *>(a.workspace + a.ws_off[ci])[28] != 0u;
                    const int want = thin ? tree_team_size(a, ci) : 1;
                    // SOLO mode (the comment above TeamCtx: Residency): the team this component asks for cannot form - its leader alone
                    const bool off_size = want > S && team_ld(ctl + 11) != 0u;
                    if (want <= S || off_size) {
                        unsigned expect = head;
                        if (__hip_atomic_compare_exchange_strong(a.queue + a.cls, &expect, head + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            atomicAdd(ctl + 12, 1u);               // somebody is at work (until the matching decrement below)
                            if (off_size) atomicAdd(ctl + 13, 1u);
                            // (a plan that keeps the barrier schedule, or an off-size component: the leader alone - message 0 -, the team stays as it is)
                            msg = (thin && !off_size ? (unsigned)(32 - __builtin_clz((unsigned)want)) << 28 : 0u) | (unsigned)(ci + 1);
                            break;
                        }
                        continue;                                  // someone else took it: look again
                    }
                    __builtin_amdgcn_s_sleep(16);                  // too large for this team: a larger one will take it
                    if ((++spins & 255) == 0) {
                        watch.look(a, ctl);
                        if ((spins >> 25) != 0) { atomicAdd(a.queue + 15, 1u); team_st(ctl + 9, 1u); break; }
                    }
                }
                if (msg == kTeamMsgEnd || (msg >> 28) != 0u) {
                    // From the highest rank DOWN, each store acknowledged before the next: when the team splits, the leader of a sub-team
                    // (the lowest rank of its members) goes to the queue as soon as it has read this message, and whatever it then sends its
                    // members must queue up BEHIND this message in their one-slot mailboxes.  Until round 6 the order was ascending: a
                    // sub-leader that was through the queue before this loop reached its members could overtake, and a member that read
                    // the two messages in the wrong order joined the wrong component (scripts/tree_sync_model.py run_team_formation found
                    // it; a window of a few L2 round trips against a dozen on the sub-leader's side - never seen on the hardware).
                    for (int m = L + S - 1; m > L; --m) {                                  // (a member clears its mailbox when it has read it)
                        int spins = 0;
                        while (team_ld(mbox + m) != 0u && (++spins >> 22) == 0) __builtin_amdgcn_s_sleep(1);
                        team_st(mbox + m, msg);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
            } else {
                int spins = 0;
                for (;;) {
                    msg = team_ld(mbox + rank);
                    if (msg != 0u) { team_st(mbox + rank, 0u); break; }
                    __builtin_amdgcn_s_sleep(8);
                    if ((++spins & 255) == 0 && (team_ld(ctl + 9) != 0u || (spins >> 25) != 0)) {
                        if (team_ld(ctl + 9) == 0u) atomicAdd(a.queue + 15, 1u);
                        team_st(ctl + 9, 1u);
                        msg = kTeamMsgEnd;
                        break;
                    }
                }
            }
            bc[3] = msg;
        }
        __syncthreads();
        const unsigned msg = __builtin_amdgcn_readfirstlane(bc[3]);
        __syncthreads();
        if (msg == kTeamMsgEnd) break;
        const int ci = (int)(msg & 0x0fffffffu) - 1;
        bool alone = (msg >> 28) == 0u;                   // (leader only: a component whose plan keeps the barrier schedule; the team stays)
        if (!alone) {
            const int s_new = 1 << ((int)(msg >> 28) - 1);
            const int L_new = L + ((rank - L) / s_new) * s_new;
            const bool mine = L_new == L;                 // the component stays with the first sub-team
            if (L_new != L) tm.target = 0;                // (a counter nobody has used yet)
            S = s_new; L = L_new;
            if (!mine) continue;                          // detached: my sub-team's leader goes to the queue itself
            alone = S == 1;
        }
        if (alone) {
            TeamCtx solo;
            solve_tree_component<kBlockThreads, false>(a, ci, sh, ts, solo);
        } else {
            tm.S = S; tm.r = rank - L; tm.bar = bars + L;
            tm.red = a.team_red + (size_t)unit_slot * kTeamRedPerUnit + (size_t)(L / 2) * (2 * kTeamMax * 16);
            if (tid == 0 && tm.r == 0) atomicAdd(ctl + 10, 1u);           // (statistics: components solved by a team)
            solve_tree_component<kBlockThreads, true>(a, ci, sh, ts, tm);
        }
        if (tid == 0 && rank == L) atomicSub(ctl + 12, 1u);              // (the workgroup that took the component from the queue)
        __syncthreads();
    }
}

// A fused batch solved a SECOND time: its packed-class records are materialised once (one thread per 16-byte chunk) and every
// later solve reads them - contiguous 80-byte records cost the packed kernel 10 % less than the gather (0.49 against 0.54 ms on
// config 4), while a one-shot pipeline (one solve per batch) never pays for writing and re-reading 400 MB.
__global__ void k_materialize_records(uint32_t n_records, const uint32_t *edge_ref, const uint32_t *edge_word, const uint32_t *f_row,
                                      const float *disp1, const float *disp2, const float *sim, uint4 *records) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p = t / 5;
    const int chunk = (int)(t - 5 * p);
    if (p >= n_records) return;
    const uint32_t eid = edge_ref[p], m = eid >> 1;
    const size_t row = f_row ? (size_t)f_row[m] : (size_t)m;
    const float *fl = ((eid & 1u) ? disp1 : disp2) + 18 * row;
    uint4 q;
    if (chunk < 4) {
        const uint2 a = reinterpret_cast<const uint2 *>(fl)[2 * chunk], b = reinterpret_cast<const uint2 *>(fl)[2 * chunk + 1];
        q.x = a.x; q.y = a.y; q.z = b.x; q.w = b.y;
    } else {
        q.x = __float_as_uint(fl[16]); q.y = __float_as_uint(fl[17]); q.z = __float_as_uint(sim[m]); q.w = edge_word[p];
    }
    records[5 * p + chunk] = q;
}

// Order in which a class hands out its components: by expected duration, longest first.  The batch order inside a class is by
// edge count, which predicts a workgroup's lifetime hardly better than a random order (list-scheduling the measured lifetimes of
// the config-5 class of 131-192 rows on 256 CUs: 6.9 ms by edges, 6.6 random, 4.9 with the lifetimes known).  Rows (the
// factorization is cubic in them) and whether the component joins several tracks (its Tukey edges cost iterations: 6.3 against
// 4.2 on average, Spearman 0.58) give 5.8 ms.  key = class, then rows x (1 + [more than one track]) descending; a component
// has one constant node (the root) per track.
__global__ void k_wg_order_keys(const CompDesc *descs, int n, int b1, int b2, int b3, uint32_t *keys, uint32_t *vals, int wg_begin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CompDesc d = descs[i];
    const uint32_t cls = (i >= b1) + (i >= b2) + (i >= b3);                   // boundaries of the four classes, relative
    const uint32_t work = (uint32_t)d.n_var * (1u + ((uint32_t)d.n_nodes - d.n_var > 1u ? 1u : 0u));
    keys[i] = (cls << 24) | (0xffffffu - min(work, 0xffffffu));
    vals[i] = (uint32_t)(wg_begin + i);
}

// Persistent workgroups: the launch holds as many workgroups as the chip can keep resident for the class and each of them
// takes the next component of the class from an atomic queue until the class is empty.  A workgroup per component left the
// order to the hardware dispatcher, which deals workgroups to the XCDs round-robin whatever they cost: a CU sat idle 0.16 ms on
// average (up to 1.3 ms) before its next 160-KB workgroup while the queue was still full, and CUs ended up with one to seven
// components each (`scripts/c5_timeline.py`).  Here a free CU always takes the largest component left.
template <int kBlockThreads>
__device__ __forceinline__ void block_kernel_body(const KernelArgs &a, int max_rows) {
    extern __shared__ double dyn[];
    __shared__ BlockShared sh;
    __shared__ int next_ci;
    for (;;) {
        if (threadIdx.x == 0) {
            const int k = a.desc_begin + (int)atomicAdd(a.queue + a.cls, 1u);
            next_ci = k < a.desc_end ? (int)a.wg_order[k - a.wg_begin] : -1;
        }
        __syncthreads();
        // (readfirstlane: the compiler cannot know that an LDS load is wave-uniform; with the index in an SGPR the descriptor, the
        // sizes and every pointer derived from it are scalar loads / SALU arithmetic instead of ~40 VGPRs carried through the solve)
        const int ci = __builtin_amdgcn_readfirstlane(next_ci);
        __syncthreads();                              // everyone has read it before the next round overwrites it
        if (ci < 0) break;
        solve_component<kBlockThreads>(a, max_rows, ci, dyn, sh);
        __syncthreads();                              // the component's last LDS reads are done
    }
}
// (LFR_BLOCK_WPE_256=1 with LFR_THREADS_L=256: the experiment of round 6 - the 192-row class on 256 threads with 512 registers and no
// spill at all ran 5.92 ms per config-5 solve against 5.80 with the spills and 4.55 for 512 threads: the kernel lives on its threads, the
// spills are not what holds it; profiles/r06_ab/block_kernel_spills.txt)
#ifndef LFR_BLOCK_WPE_256
#define LFR_BLOCK_WPE_256 2
#endif
template <int kBlockThreads>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(kBlockThreads == 256 ? LFR_BLOCK_WPE_256 : 2, kBlockThreads == 256 ? LFR_BLOCK_WPE_256 : 2))) void solve_block_kernel(const KernelArgs a, int max_rows) {
    block_kernel_body<kBlockThreads>(a, max_rows);
}

#ifndef LFR_THREADS_S
#define LFR_THREADS_S 128
#endif
#ifndef LFR_THREADS_M
#define LFR_THREADS_M 256
#endif
#ifndef LFR_THREADS_L
#define LFR_THREADS_L 512
#endif
constexpr int kThreadsS = LFR_THREADS_S, kThreadsM = LFR_THREADS_M, kThreadsL = LFR_THREADS_L, kThreadsG = LFR_THREADS_G;

size_t block_vector_doubles(int max_rows) { return 2 * (size_t)(max_rows + 2) + 8 * (size_t)max_rows; }
size_t block_lds_bytes(int max_rows, bool global_matrix) {
    if (global_matrix) return 0;                       // matrix and vectors live in the HBM workspace
    // nine vectors (the step is row max_rows of the matrix) + the packed triangle of max_rows + 1 rows
    return (block_vector_doubles(max_rows) - (size_t)max_rows + (size_t)(max_rows + 1) * (max_rows + 2) / 2) * sizeof(double);
}

// LFR_HOST_TRACE=1: host-side time stamps (us, steady clock) of the calls that make the Solver span, to stderr
static inline void host_trace(const char *what) {
    static const bool on = [] { const char *e = getenv("LFR_HOST_TRACE"); return e && e[0] == '1'; }();
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "lfr-host %10.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            lfr::set_error("%s failed: %s", #expr, hipGetErrorString(_e));                    \
            return LFR_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

}  // namespace

// =============================================================================================
// batch management + C ABI
// =============================================================================================
constexpr size_t kProfWords = 8 * lfr::KC_COUNT + 8 + 64;  // phase counters of -DLFR_PROFILE_PHASES + 16 32-bit class queues + -DLFR_PROFILE_FACTOR (16 per workgroup class)

constexpr uint32_t kPackedEventsAliased = 1u << 31;     // ev_recorded: the packed launch is timed by the solve's own pair of events
struct lfr_batch {
    int device = 0;
    lfr::DevCtx *ctx = nullptr;
    int tukey_variant = LFR_TUKEY_CERES1;
    int64_t n_graph_nodes = 0;
    int shard_world = 1;
    // launch geometry (device-assembled batches: read back once as AsmSummary)
    int n_desc = 0;
    int class_begin[lfr::KC_COUNT + 1] = {0};
    int64_t class_edges[lfr::KC_COUNT] = {0};
    int class_max_rows[lfr::KC_COUNT] = {0};          // largest system of every workgroup class (sizes its launch's LDS)
    int64_t n_edges = 0, n_nodes = 0, n_tracks = 0;
    // device: everything lives in `slab` (+ the workgroup kernels' workspace in `ws_slab`)
    lfr::DevArena slab, ws_slab;
    CompDesc *d_descs = nullptr;
    EdgeRec *d_edges = nullptr;
    uint32_t *d_node_ids = nullptr;
    double *d_positions = nullptr;
    CompInfoDev *d_infos = nullptr;
    double *d_workspace = nullptr;
    uint64_t es_doubles = 0;                             // per-edge scratch of the workgroup classes (8 doubles per edge), the head of the workspace
    int tree_levels_max = 0;                             // KC_GLOBAL: levels of the deepest elimination tree
    int64_t tree_blocks = 0, tree_updates = 0;           // KC_GLOBAL: 16-row columns / left-looking tile updates per factorization, summed over the class
    int tree_begin = 0;                                  // first descriptor of the class; per component of the class: columns, tiles, 16x16x16 updates, levels, sweep items
    // teams of workgroups per component (solve_tree_team_kernel): control words + reduction slots at the tail of the workspace,
    // the work thresholds of teams of 2 / 4 / 8 (LFR_TREE_TEAM), the workgroups the class's components ask for together
    unsigned int *d_team_ctl = nullptr;
    double *d_team_red = nullptr;
    uint32_t team_work[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    int team_wgs = 0;
    uint32_t team_patience_us = 50000u;                  // LFR_TEAM_PATIENCE_MS: the comment above TeamCtx (Residency)
    std::vector<int64_t> tree_comp_stats;                // 5 per component
    int64_t tree_tiles = 0, tree_dense_tiles = 0;          // KC_GLOBAL: 16x16 tiles stored / tiles of the dense lower triangles
    uint64_t *d_ws_off = nullptr, *d_es_off = nullptr;
    // fused gather: the packed kernel reads the graph's own flow arrays (kept alive through dev_hold)
    bool fused = false;                                  // the NEXT solve gathers (true until the records have been materialised)
    uint32_t packed_edges = 0;                           // records of the packed classes (the head of the edge array)
    uint32_t *d_edge_ref = nullptr, *d_edge_word = nullptr;
    std::shared_ptr<lfr::DevProblem> dev_hold;
    unsigned long long *d_prof = nullptr;
    lfr::NodeInc *d_node_inc = nullptr;
    uint32_t *d_in_idx = nullptr;
    uint32_t *d_desc_component = nullptr, *d_desc_class = nullptr, *d_desc_tracks = nullptr;   // device-assembled: behind the host mirrors
    // host mirrors (host-assembled: filled at creation; device-assembled: fetched on first use)
    bool mirrors_valid = false;
    std::vector<CompDesc> descs;
    std::vector<int64_t> desc_component;
    std::vector<int32_t> desc_class, desc_tracks;
    std::vector<uint32_t> node_ids;
    // pinned staging of the positions (downloads, zero-copy view)
    double *h_positions = nullptr;
    size_t h_positions_bytes = 0;
    float *h_positions_f32 = nullptr, *d_positions_f32 = nullptr;      // lfr_batch_positions_view_f32: converted on the device, half the copy
    size_t h_positions_f32_bytes = 0, d_positions_f32_bytes = 0;
    // events / streams
    static constexpr int kSlots = 64;                    // event ring: timings of the last 64 solves
    static constexpr int kEvPerSlot = 2 * (lfr::KC_COUNT + 1);
    hipEvent_t ev_ring[kSlots * kEvPerSlot];
    hipEvent_t *ev = ev_ring;                            // slot of the current solve
    uint32_t ev_recorded[kSlots] = {};                   // per slot: classes whose start/end events were recorded
    int64_t n_solves = 0;
    bool serial = false;                               // LFR_SERIAL_CLASSES=1: all classes on the caller's stream
    hipEvent_t ev_fork = nullptr;
    lfr::DevArena order_slab;                          // hand-out order of the workgroup classes + the sort's temporaries
    uint32_t *d_wg_order = nullptr;
    hipEvent_t ev_order = nullptr;                     // the order is sorted on the context's stream: solves wait for it
    hipStream_t side_stream = nullptr;                 // the packed launch runs beside the workgroup-per-component kernels
    hipStream_t wg_stream[lfr::KC_COUNT] = {nullptr};  // one stream per further workgroup class (all owned by the device context)
    hipStream_t last_stream = nullptr;                 // stream of the latest solve (downloads wait for it)
    int packed_slot = 0;                               // class slot that carries the packed launch's events
    double h2d_ms = 0.0;             // upload (host-assembled) or device assembly incl. waiting for the flows
    std::vector<CompInfoDev> infos;      // last downloaded
    bool infos_valid = false;

    lfr_batch() { for (auto &e : ev_ring) e = nullptr; }
    ~lfr_batch() {
        if (ctx) {
            (void)hipSetDevice(device);
            if (n_solves > 0) (void)hipStreamSynchronize(last_stream);      // nothing may still use the slab
            if (side_stream) (void)hipStreamSynchronize(side_stream);
            for (auto &w : wg_stream) if (w) (void)hipStreamSynchronize(w);
            (void)hipStreamSynchronize(ctx->s_main);
            if (h_positions) ctx->pinned_release(h_positions, h_positions_bytes);
            if (h_positions_f32) ctx->pinned_release(h_positions_f32, h_positions_f32_bytes);
            if (d_positions_f32) ctx->dev_release(d_positions_f32, d_positions_f32_bytes);
        }
        if (ctx) {                                      // (every stream this batch used has been waited for above: the events are idle)
            for (auto &e : ev_ring) ctx->event_release(e, true);
            ctx->event_release(ev_fork, false);
            ctx->event_release(ev_order, false);
        } else {
            for (auto &e : ev_ring) if (e) (void)hipEventDestroy(e);
        }
        // slab / ws_slab return to the context's cache in their destructors
    }
};

namespace {

#define TAKE_B(dst, T, count)                                                                                 \
    b->dst = b->slab.take_n<T>((size_t)(count));                                                              \
    if (!b->dst) { lfr::set_error("batch slab exhausted (%s)", #dst); return LFR_ERR_NOMEM; }

// host mirrors of a device-assembled batch (descriptors, component ids, classes, node ids): 2-6 MB, fetched once
int ensure_mirrors(lfr_batch *b) {
    if (b->mirrors_valid) return LFR_OK;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = b->ctx->s_main;
    const size_t nd = (size_t)b->n_desc, nn = (size_t)b->n_nodes;
    b->descs.resize(nd); b->desc_component.resize(nd); b->desc_class.resize(nd); b->desc_tracks.resize(nd);
    b->node_ids.resize(nn);
    std::vector<uint32_t> comp(nd), cls(nd), trk(nd);
    if (nd) {
        HIP_TRY(hipMemcpyAsync(b->descs.data(), b->d_descs, nd * sizeof(CompDesc), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(comp.data(), b->d_desc_component, 4 * nd, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(cls.data(), b->d_desc_class, 4 * nd, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(trk.data(), b->d_desc_tracks, 4 * nd, hipMemcpyDeviceToHost, st));
    }
    if (nn) HIP_TRY(hipMemcpyAsync(b->node_ids.data(), b->d_node_ids, 4 * nn, hipMemcpyDeviceToHost, st));
    HIP_TRY(lfr::stream_wait(st));
    for (size_t i = 0; i < nd; ++i) { b->desc_component[i] = comp[i]; b->desc_class[i] = (int32_t)cls[i]; b->desc_tracks[i] = (int32_t)trk[i]; }
    b->mirrors_valid = true;
    return LFR_OK;
}

// device-assembled batch: labels (device graph stage) -> batch layout, all in HBM
int create_on_device(lfr_batch *b, const lfr::Problem &p, int shard_rank, int shard_world) {
    std::shared_ptr<lfr::DevProblem> dp = p.dev_get(b->device);
    const bool stage_flows = shard_world == 1;          // a shard gathers its rows zero-copy from pinned host memory
    if (!dp) {                                           // labels came from the host stage, or live on another GPU
        int rc = p.ensure_host_labels();
        if (rc != LFR_OK) return rc;
        rc = lfr::upload_labels(p, b->device, stage_flows, dp);
        if (rc != LFR_OK) return rc;
        p.dev_set(b->device, dp);
    }
    const lfr::DevGraph &dg = *dp->graph;
    const int64_t N = dg.N, M = dg.M, C = p.stats.n_components;
    hipStream_t st = b->ctx->s_main;
    // (the assembly's duration for the statistics is host wall clock: the assembly ends with the stage's one synchronisation anyway, and
    // a pair of timing events needed another blocking wait on an idle stream - tens of microseconds of a 1.7-ms Solver span)
    const auto t_asm0 = std::chrono::steady_clock::now();
    const size_t fixed = sizeof(double) * 2 * (size_t)std::max<int64_t>(N, 1) + sizeof(CompInfoDev) * (size_t)(C + 1) +
                         kProfWords * sizeof(unsigned long long) + ((size_t)1 << 16);
    if (!b->slab.init(b->ctx, lfr::assembly_output_bytes(N, M, C) + fixed)) return LFR_ERR_NOMEM;
    TAKE_B(d_positions, double, 2 * std::max<int64_t>(N, 1));
    TAKE_B(d_infos, CompInfoDev, C + 1);
    TAKE_B(d_prof, unsigned long long, kProfWords);
    // Roots, constants and nodes outside every solved component stay at 0 for the life of the batch
    // (solve.cc:609-612); the kernels overwrite every variable on every solve, so no per-solve memset.
    HIP_TRY(hipMemsetAsync(b->d_positions, 0, sizeof(double) * 2 * (size_t)std::max<int64_t>(N, 1), st));
    HIP_TRY(hipMemsetAsync(b->d_prof, 0, kProfWords * sizeof(unsigned long long), st));
    lfr::DeviceAssembly dev;
    host_trace("assembly: launches begin");
    const int rc = lfr::assemble_on_device(p, *dp, shard_rank, shard_world, b->slab, dev);     // ends with the one synchronisation
    host_trace("assembly: summary is back");
    if (rc != LFR_OK) return rc;
    b->h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_asm0).count();
    b->d_descs = dev.d_descs; b->d_edges = dev.d_edges; b->d_node_ids = dev.d_node_ids; b->d_node_inc = dev.d_node_inc;
    b->d_in_idx = dev.d_in_idx; b->d_ws_off = dev.d_ws_off; b->d_es_off = dev.d_es_off;
    b->d_desc_component = dev.d_desc_component; b->d_desc_class = dev.d_desc_class; b->d_desc_tracks = dev.d_desc_tracks;
    b->fused = dev.fused; b->d_edge_ref = dev.d_edge_ref; b->d_edge_word = dev.d_edge_word; b->packed_edges = dev.summary.packed_edges;
    if (dev.fused) b->dev_hold = dp;
    const lfr::AsmSummary &s = dev.summary;
    b->n_desc = (int)s.n_desc; b->n_edges = s.total_edges; b->n_nodes = s.total_nodes; b->n_tracks = s.n_tracks;
    for (int c = 0; c <= lfr::KC_COUNT; ++c) b->class_begin[c] = (int)s.class_begin[c];
    for (int c = 0; c < lfr::KC_COUNT; ++c) b->class_edges[c] = (int64_t)s.class_edges[c];
    for (int c = 0; c < lfr::KC_COUNT; ++c) b->class_max_rows[c] = (int)s.class_max_rows[c];
    b->es_doubles = s.es_doubles;                        // (the workspace itself: finish_workspace)
    return LFR_OK;
}

// host-assembled batch (lfr_problem_build): shard the problem's arrays, upload
int create_from_host(lfr_batch *b, const lfr::Problem &p, int shard_rank, int shard_world) {
    const std::vector<int32_t> shard = lfr::assign_shards(p, shard_world);
    const bool whole = shard_world == 1;
    std::vector<EdgeRec> edges_copy;
    std::vector<uint32_t> in_idx_copy;
    std::vector<lfr::NodeInc> node_inc_copy;
    std::vector<uint64_t> ws_off, es_off;
    uint64_t ws = 0;
    for (size_t i = 0; i < p.descs.size(); ++i) {
        if (shard[i] != shard_rank) continue;
        CompDesc d = p.descs[i];
        if (!whole) {
            const uint32_t eo = (uint32_t)edges_copy.size(), no = (uint32_t)b->node_ids.size();
            edges_copy.insert(edges_copy.end(), p.edges.begin() + d.edge_off, p.edges.begin() + d.edge_off + d.n_edges);
            in_idx_copy.insert(in_idx_copy.end(), p.in_idx.begin() + d.edge_off, p.in_idx.begin() + d.edge_off + d.n_edges);
            b->node_ids.insert(b->node_ids.end(), p.node_ids.begin() + d.node_off, p.node_ids.begin() + d.node_off + d.n_nodes);
            node_inc_copy.insert(node_inc_copy.end(), p.node_inc.begin() + d.node_off, p.node_inc.begin() + d.node_off + d.n_nodes);
            d.edge_off = eo; d.node_off = no;
        }
        b->descs.push_back(d); b->desc_component.push_back(p.desc_component[i]);
        b->desc_class.push_back(p.desc_class[i]); b->desc_tracks.push_back(p.desc_tracks[i]);
        const int cls = p.desc_class[i], rows = 2 * d.n_var;
        es_off.push_back(ws); ws_off.push_back(0);
        if (cls >= lfr::KC_BLOCK) {
            ws += 8 * (uint64_t)d.n_edges;                                                     // per-edge scratch
            b->class_max_rows[cls] = std::max(b->class_max_rows[cls], rows);
        }
        b->class_edges[cls] += d.n_edges;
        b->n_edges += d.n_edges; b->n_nodes += d.n_nodes; b->n_tracks += p.desc_tracks[i];
    }
    if (whole) b->node_ids = p.node_ids;
    const std::vector<EdgeRec> &edges = whole ? p.edges : edges_copy;
    const std::vector<uint32_t> &in_idx = whole ? p.in_idx : in_idx_copy;
    const std::vector<lfr::NodeInc> &node_inc = whole ? p.node_inc : node_inc_copy;
    b->es_doubles = ws;                                  // (KC_GLOBAL: the matrices' workspace is planned in finish_workspace)
    b->n_desc = (int)b->descs.size();
    {   // class ranges (descs are sorted by class)
        int c = 0;
        b->class_begin[0] = 0;
        for (int i = 0; i <= b->n_desc; ++i) {
            const int cls = i < b->n_desc ? b->desc_class[i] : lfr::KC_COUNT;
            while (c < cls) b->class_begin[++c] = i;
        }
    }
    b->mirrors_valid = true;
    hipStream_t st = b->ctx->s_main;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct EvGuard { hipEvent_t &x, &y; ~EvGuard() { if (x) (void)hipEventDestroy(x); if (y) (void)hipEventDestroy(y); } } guard{e0, e1};
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    const size_t nd = std::max<size_t>(b->descs.size(), 1), ne = std::max<size_t>(edges.size(), 1), nn = std::max<size_t>(b->node_ids.size(), 1);
    const size_t npos = 2 * (size_t)std::max<int64_t>(b->n_graph_nodes, 1);
    const size_t bytes = nd * (sizeof(CompDesc) + sizeof(CompInfoDev) + 16) + ne * (sizeof(EdgeRec) + 4) + nn * (4 + sizeof(lfr::NodeInc)) +
                         npos * sizeof(double) + kProfWords * sizeof(unsigned long long) + ((size_t)1 << 16);
    if (!b->slab.init(b->ctx, bytes)) return LFR_ERR_NOMEM;
    TAKE_B(d_descs, CompDesc, nd); TAKE_B(d_edges, EdgeRec, ne); TAKE_B(d_node_ids, uint32_t, nn);
    TAKE_B(d_node_inc, lfr::NodeInc, nn); TAKE_B(d_in_idx, uint32_t, ne);
    TAKE_B(d_positions, double, npos); TAKE_B(d_infos, CompInfoDev, nd);
    TAKE_B(d_ws_off, uint64_t, nd); TAKE_B(d_es_off, uint64_t, nd); TAKE_B(d_prof, unsigned long long, kProfWords);
    HIP_TRY(hipMemsetAsync(b->d_positions, 0, npos * sizeof(double), st));        // solve.cc:609-612, see create_on_device
    HIP_TRY(hipMemsetAsync(b->d_prof, 0, kProfWords * sizeof(unsigned long long), st));
    if (!b->descs.empty()) {
        HIP_TRY(hipMemcpyAsync(b->d_ws_off, ws_off.data(), ws_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_es_off, es_off.data(), es_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_descs, b->descs.data(), b->descs.size() * sizeof(CompDesc), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_edges, edges.data(), edges.size() * sizeof(EdgeRec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_node_ids, b->node_ids.data(), b->node_ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_node_inc, node_inc.data(), node_inc.size() * sizeof(lfr::NodeInc), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_in_idx, in_idx.data(), in_idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));               // the staging vectors above die at return
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    b->h2d_ms = ms;
    return LFR_OK;
}

// plan i of the elimination-tree class: tab[3 i + 2] doubles from ws[tab[3 i]] (where one copy landed all plans) to ws[tab[3 i + 1]] (its component's workspace)
__global__ void k_place_plans(double *ws, const uint64_t *tab) {
    const uint64_t from = tab[3 * blockIdx.x], to = tab[3 * blockIdx.x + 1], n = tab[3 * blockIdx.x + 2];
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) ws[to + i] = ws[from + i];
}

// The workgroup kernels' workspace: per-edge scratch, then - for the components of the HBM class - the elimination-tree plans
// (lfr_treeplan.cpp) with their tiles, vectors and team areas.  The plans need the components' (source, destination, kind) lists on the host: the
// last word of every edge record, fetched with one strided copy when the batch was assembled on the device.
int finish_workspace(lfr_batch *b, const lfr::Problem &p) {
    const int g0 = b->class_begin[lfr::KC_GLOBAL], g1 = b->class_begin[lfr::KC_COUNT];
    hipStream_t st = b->ctx->s_main;
    if (g1 <= g0) {
        if (b->es_doubles) {
            if (!b->ws_slab.init(b->ctx, b->es_doubles * sizeof(double))) return LFR_ERR_NOMEM;
            b->d_workspace = (double *)b->ws_slab.base;
        }
        return LFR_OK;
    }
    host_trace("finish_workspace: enters (components above the LDS classes)");
    { const int rc = ensure_mirrors(b); if (rc != LFR_OK) return rc; }
    host_trace("finish_workspace: host mirrors of the descriptors");
    const int ng = g1 - g0;
    const uint32_t e0 = b->descs[g0].edge_off;
    uint64_t ne = 0;
    for (int i = g0; i < g1; ++i) ne = std::max<uint64_t>(ne, (uint64_t)b->descs[i].edge_off + b->descs[i].n_edges - e0);
    // The record words and the plans are megabytes in ~130 allocations made by the pool's workers: handing them back to the system (munmap with
    // TLB shootdowns on every core a worker ran on) took 1.2 ms at the end of this function - a thread of its own does it while the solve starts.
    struct PlanTemps { std::vector<uint32_t> words; std::vector<lfr::TreePlan> plans; };
    struct Later { PlanTemps *p; ~Later() { PlanTemps *q = p; std::thread([q] { delete q; }).detach(); } } later{new PlanTemps()};
    std::vector<uint32_t> &words = later.p->words;
    words.resize(ne);
    if (p.host_batch && b->shard_world == 1) {
        for (uint64_t e = 0; e < ne; ++e) words[e] = (uint32_t)p.edges[e0 + e].src | ((uint32_t)p.edges[e0 + e].dst_kind << 16);
    } else if (ne) {
        // (a device-assembled whole batch kept one word per record beside the records: one linear copy.  The strided copy of the records' last
        // words - 4 bytes of every 80 - took 1.1 ms for 0.3 M records)
        if (b->d_edge_word) HIP_TRY(hipMemcpyAsync(words.data(), b->d_edge_word + e0, 4 * ne, hipMemcpyDeviceToHost, st));
        else HIP_TRY(hipMemcpy2DAsync(words.data(), 4, reinterpret_cast<const char *>(b->d_edges + e0) + 76, sizeof(EdgeRec), 4, ne, hipMemcpyDeviceToHost, st));
        HIP_TRY(lfr::stream_wait(st));
    }
    host_trace("finish_workspace: record words on the host");
    std::vector<lfr::TreePlan> &plans = later.p->plans;
    plans.resize(ng);
    {
        std::atomic<int> next{0};
        // (a plan is ~1 ms of one core for a cap-sized component and the plans are independent: as many threads as components, up to half
        // the machine - 32 threads left the 130 plans of the sparse bench workload at 6.5 ms of a 10-ms Solver span)
        const int T = std::max(1, std::min(ng, (int)std::min(128u, std::max(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), std::thread::hardware_concurrency() / 2))));
        auto work = [&] {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= ng) break;
                const CompDesc &d = b->descs[g0 + i];
                lfr::tree_plan(d.n_var, d.n_edges, words.data() + (d.edge_off - e0), plans[i]);
            }
        };
        lfr::run_on_pool(T, work);                 // (persistent workers: creating 127 threads per batch was most of the 4 ms this step took)
    }
    host_trace("finish_workspace: plans made");
    std::vector<uint64_t> off(ng);
    uint64_t ws = (b->es_doubles + 31) / 32 * 32, hdr_total = 0;
    b->tree_tiles = b->tree_dense_tiles = 0;
    b->tree_levels_max = 0; b->tree_blocks = 0; b->tree_updates = 0;
    for (int i = 0; i < ng; ++i) {
        if (plans[i].blob.empty()) { lfr::set_error("a component is too large for the elimination-tree plan's 32-bit offsets"); return LFR_ERR_UNSUPPORTED; }
        off[i] = ws;
        ws += (plans[i].doubles() + 31) / 32 * 32;
        hdr_total += plans[i].header_doubles();
        b->tree_tiles += plans[i].n_tiles;
        b->tree_dense_tiles += (int64_t)plans[i].NB * (plans[i].NB + 1) / 2;
        b->tree_levels_max = std::max(b->tree_levels_max, plans[i].n_levels);
        b->tree_blocks += plans[i].NB; b->tree_updates += (int64_t)plans[i].n_updates;
        const int64_t cs[5] = {plans[i].NB, plans[i].n_tiles, (int64_t)plans[i].n_updates, plans[i].n_levels, plans[i].n_items};
        b->tree_comp_stats.insert(b->tree_comp_stats.end(), cs, cs + 5);
    }
    b->tree_begin = g0;
    // Teams (the comment above TeamCtx): a component whose hand-out key reaches team_work[k] is solved by 2 << k workgroups.
    // LFR_TREE_TEAM="w2,w4[,w8]" sets the thresholds, "0" switches teams off.  Defaults from the cap-sized sparse workload: one
    // workgroup runs ~0.11 us per row and LM iteration, iteration counts vary 10-40 whatever the size, so everything above ~700 rows
    // can end a launch on its own.
    {
        uint32_t w[3] = {700u, 1500u, 2000u};
        if (const char *e = getenv("LFR_TREE_TEAM")) {
            unsigned v[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
            const int got = sscanf(e, "%u,%u,%u", &v[0], &v[1], &v[2]);
            if (got >= 1 && v[0] == 0u) w[0] = w[1] = w[2] = 0xffffffffu;
            else if (got >= 2) { w[0] = v[0]; w[1] = std::max(v[0], v[1]); w[2] = got >= 3 ? std::max(w[1], v[2]) : 0xffffffffu; }
        }
        int64_t wgs = 0;
        bool any = false;
        for (int i = 0; i < ng; ++i) {
            const CompDesc &d = b->descs[g0 + i];
            const uint32_t work = (uint32_t)d.n_var * (1u + ((uint32_t)d.n_nodes - d.n_var > 1u ? 1u : 0u));
            int t = 1;
            for (int k = 0; k < 3; ++k) if ((2 << k) <= kTeamMax && work >= w[k]) t = 2 << k;
            if (plans[i].blob[28] == 0u) t = 1;
            any = any || t > 1;
            wgs += t;
        }
        b->team_wgs = 0;
        if (const char *e = getenv("LFR_TEAM_PATIENCE_MS")) b->team_patience_us = (uint32_t)(std::min(60000.0, std::max(0.01, atof(e))) * 1000.0);
        if (any && b->n_desc < (1 << 28)) {
            for (int k = 0; k < 3; ++k) b->team_work[k] = w[k];
            b->team_wgs = (int)std::min<int64_t>(256, (wgs + 31) / 32 * 32);
        }
    }
    const uint64_t team_off = ws;
    if (b->team_wgs) ws += (kTeamCtlWords + 1) / 2 + 8ull * kTeamUnitsPerXcc * kTeamRedPerUnit;
    // behind everything: where the plans' words land in ONE copy before a kernel moves each to its component's workspace, + {from, to, doubles} per plan
    const uint64_t land_off = ws;
    ws += hdr_total + 3ull * (uint64_t)ng;
    if (!b->ws_slab.init(b->ctx, ws * sizeof(double))) return LFR_ERR_NOMEM;
    b->d_workspace = (double *)b->ws_slab.base;
    if (b->team_wgs) {
        b->d_team_ctl = reinterpret_cast<unsigned int *>(b->d_workspace + team_off);
        b->d_team_red = b->d_workspace + team_off + (kTeamCtlWords + 1) / 2;
        // (the reduction slots carry {value, launch << 32 | reduction} granules: tags of an earlier owner of this memory must not match)
        HIP_TRY(hipMemsetAsync(b->d_team_red, 0, 8ull * kTeamUnitsPerXcc * kTeamRedPerUnit * sizeof(double), st));
    }
    // the plans' words: staged in one pinned buffer (it must outlive the asynchronous copies: waited for below)
    host_trace("finish_workspace: workspace allocated, team area cleared");
    size_t got = 0;
    double *stage = (double *)b->ctx->pinned_acquire((hdr_total + 3ull * (uint64_t)ng) * sizeof(double), &got);
    if (!stage) return LFR_ERR_NOMEM;
    {
        // (one copy + one kernel instead of a copy per plan: 130 small copies took 1.6 ms)
        uint64_t so = 0;
        uint64_t *tab = reinterpret_cast<uint64_t *>(stage + hdr_total);
        for (int i = 0; i < ng; ++i) {
            tab[3 * i] = land_off + so; tab[3 * i + 1] = off[i]; tab[3 * i + 2] = plans[i].header_doubles();
            so += plans[i].header_doubles();
        }
        std::atomic<int> next{0};
        lfr::run_on_pool(std::min(ng, 16), [&] {                    // (megabytes of words into the pinned buffer: a few workers, not one)
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= ng) break;
                memcpy(stage + (tab[3 * i] - land_off), plans[i].blob.data(), plans[i].blob.size() * 4);
            }
        });
        if (hipMemcpyAsync(b->d_workspace + land_off, stage, (hdr_total + 3ull * (uint64_t)ng) * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
            b->ctx->pinned_release(stage, got); lfr::set_error("hipMemcpyAsync of the plans failed"); return LFR_ERR_HIP;
        }
        hipLaunchKernelGGL(k_place_plans, dim3((unsigned)ng), dim3(256), 0, st, b->d_workspace, reinterpret_cast<const uint64_t *>(b->d_workspace + land_off + hdr_total));
    }
    hipError_t e1 = hipMemcpyAsync(b->d_ws_off + g0, off.data(), ng * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    hipError_t e2 = hipStreamSynchronize(st);
    host_trace("finish_workspace: plans uploaded");
    b->ctx->pinned_release(stage, got);
    host_trace("finish_workspace: staging buffer back");
    if (e1 != hipSuccess || e2 != hipSuccess) { lfr::set_error("upload of the elimination-tree plans failed"); return LFR_ERR_HIP; }
    if (getenv("LFR_VERBOSE"))
        fprintf(stderr, "lfr: %d component(s) above %d rows: elimination-tree plans keep %lld of %lld tiles (%.1f %%) in %lld columns, at most %d levels, workspace %.1f MB\n", ng,
                lfr::block_max_rows(), (long long)b->tree_tiles, (long long)b->tree_dense_tiles, 100.0 * b->tree_tiles / std::max<int64_t>(1, b->tree_dense_tiles),
                (long long)b->tree_blocks, b->tree_levels_max, ws * 8e-6);
    return LFR_OK;
}

}  // namespace

extern "C" {

int lfr_problem_build_hip_ex(const lfr_graph *g, int device, int64_t max_nodes_in_component, const int64_t *component_override,
                             int flags, lfr_problem **out) {
    if (!g || !out) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    const bool stage_flows = !(flags & LFR_BUILD_FLOWS_STAY_ON_HOST);
    if (!component_override) {
        lfr_problem *h = new lfr_problem();
        const int rc = lfr::graph_stage_on_device(g->g, max_nodes_in_component, device, stage_flows, h->p);
        if (rc == LFR_OK) { *out = h; return LFR_OK; }
        delete h;
        if (rc != lfr::LFR_GRAPHSTAGE_USE_HOST) { *out = nullptr; return rc; }
    }
    return lfr_problem_build_labels(g, max_nodes_in_component, component_override, out);   // host graph stage
}

// Multi-GPU, one process per GPU: the graph stage of ONE rank.  The constrained spanning forest of solve.cc:489-541 is a global greedy, but
// it decomposes exactly by connected component of the match graph (no union ever crosses one), and so does everything behind it (roots,
// components, the size cap, the solve): rank r takes the connected components k = r (mod world) in node order - found by the same
// union-find pass on every rank - and runs tracks / roots / components over their matches only.  lfr_batch_create(p, device, 0, 1) then
// assembles exactly this rank's components; positions of the other ranks' nodes stay 0.  Returns with lfr_problem_cc_sharded(p) = 0
// when one connected component dominates (real data: wrong matches link everything) - the problem then covers the whole graph and the
// caller shards its components at assembly (lfr_batch_create(p, device, rank, world)) as before.
int lfr_problem_build_hip_shard(const lfr_graph *g, int device, int64_t max_nodes_in_component, int flags, int shard_rank, int shard_world,
                                lfr_problem **out) {
    if (!g || !out || shard_world < 1 || shard_world > 64 || shard_rank < 0 || shard_rank >= shard_world) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    const bool stage_flows = !(flags & LFR_BUILD_FLOWS_STAY_ON_HOST);
    lfr_problem *h = new lfr_problem();
    const int rc = lfr::graph_stage_on_device(g->g, max_nodes_in_component, device, stage_flows, h->p, shard_rank, shard_world);
    if (rc == LFR_OK) { *out = h; return LFR_OK; }
    delete h;
    if (rc != lfr::LFR_GRAPHSTAGE_USE_HOST) { *out = nullptr; return rc; }
    return lfr_problem_build_labels(g, max_nodes_in_component, nullptr, out);      // host graph stage over the whole graph (not sharded)
}
int lfr_problem_cc_sharded(const lfr_problem *p) {
    if (!p) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    return p->p.cc_sharded ? 1 : 0;
}

int lfr_problem_build_hip(const lfr_graph *g, int device, int64_t max_nodes_in_component, const int64_t *component_override,
                          lfr_problem **out) {
    return lfr_problem_build_hip_ex(g, device, max_nodes_in_component, component_override, 0, out);
}

namespace { __global__ void lfr_warmup_kernel(int *p) { if (p) *p = 0; } }

int lfr_hip_warmup(int device) {
    // Creating the HIP context costs a few hundred ms; a host program can call this from a side
    // thread while it parses its input (the `solve` launcher does).
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));
    hipLaunchKernelGGL(lfr_warmup_kernel, dim3(1), dim3(64), 0, nullptr, (int *)nullptr);   // loads this unit's code object
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    int level = 2;                                   // LFR_WARMUP_LEVEL: 0 = context only, 1 = + toy graph, 2 = + a million matches
    if (const char *e = getenv("LFR_WARMUP_LEVEL")) level = atoi(e);
    if (level < 1) return LFR_OK;
    // HIP also resolves every kernel on its first launch (~1 ms each; the graph stage and the assembly launch
    // about forty different ones, rocPRIM's included): push a toy graph - one 18-node track (workgroup kernel)
    // and one 3-node track (packed kernel) - through the whole device pipeline once.  Best effort.
    {
        constexpr int kImg = 18;
        std::vector<std::string> names(kImg);
        std::vector<const char *> name_ptrs(kImg);
        std::vector<float> facts(kImg, 1.0f);
        for (int i = 0; i < kImg; ++i) { names[i] = "warmup" + std::to_string(i); name_ptrs[i] = names[i].c_str(); }
        std::vector<int32_t> p1, p2;
        std::vector<int64_t> off{0};
        std::vector<uint32_t> f1, f2;
        for (int a = 0; a < kImg; ++a)
            for (int b = a + 1; b < kImg; ++b) {
                p1.push_back(a); p2.push_back(b);
                f1.push_back(0); f2.push_back(0);                                  // the 18-node track
                if (b < 3) { f1.push_back(1); f2.push_back(1); }                   // the 3-node track
                off.push_back((int64_t)f1.size());
            }
        const size_t M = f1.size();
        std::vector<float> sim(M, 0.9f), flows(18 * M, 0.01f);
        lfr_graph *g = nullptr; lfr_problem *pr = nullptr; lfr_batch *bt = nullptr;
        if (lfr_graph_from_arrays(kImg, name_ptrs.data(), facts.data(), (int64_t)p1.size(), p1.data(), p2.data(), off.data(),
                                  f1.data(), f2.data(), sim.data(), flows.data(), flows.data(), nullptr, 0, &g) == LFR_OK &&
            lfr_problem_build_hip(g, device, 0, nullptr, &pr) == LFR_OK &&
            lfr_batch_create(pr, device, 0, 1, LFR_TUKEY_CERES1, &bt) == LFR_OK) {
            lfr_solve_stats st;
            (void)lfr_batch_solve(bt, nullptr, &st);
            const double *view = nullptr;
            (void)lfr_batch_positions_view(bt, &view);
        }
        lfr_batch_free(bt); lfr_problem_free(pr); lfr_graph_free(g);
    }
    // the large-input sort / scan kernels of the graph stage and of the assembly (the toy graph only reached the small-input ones)
    if (lfr::warm_graphstage_primitives(ctx) != LFR_OK || lfr::warm_assembly_primitives(ctx) != LFR_OK) return LFR_OK;     // best effort
    // The first LARGE device-to-host copy of a process takes ~7 ms longer than the next one (the toy graph's 300 bytes take another
    // path; measured with the kernel trace of the CLI: the positions' copy started 7.3 ms after the solve kernel had ended): one 8-MB
    // copy between scratch buffers here.
    {
        size_t db = 0, hb = 0;
        const size_t bytes = (size_t)8 << 20;
        void *d = ctx->dev_acquire(bytes, &db);
        void *h = ctx->pinned_acquire(bytes, &hb);
        if (d && h) {
            (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->s_main);
            (void)hipStreamSynchronize(ctx->s_main);
        }
        if (d) ctx->dev_release(d, db);
        if (h) ctx->pinned_release(h, hb);
    }
    if (level < 2) return LFR_OK;
    // rocPRIM picks other kernels (one-sweep radix sort, look-back scans) once the inputs are large: a second pass with a
    // million matches (170 k four-node tracks, zero flows) resolves those as well, so that a one-shot caller's
    // "Total time" is not spent loading code.  Best effort, ~30 ms beside the caller's parse.
    {
        constexpr int kImg = 64, kLen = 4;
        constexpr int64_t kTracks = 170000;
        std::vector<std::string> names(kImg);
        std::vector<const char *> name_ptrs(kImg);
        std::vector<float> facts(kImg, 1.0f);
        for (int i = 0; i < kImg; ++i) { names[i] = "warmup" + std::to_string(i); name_ptrs[i] = names[i].c_str(); }
        // one ImagePair per (image a, image b = a + d): its matches are the tracks whose window covers both
        std::vector<int32_t> p1, p2;
        std::vector<int64_t> off{0};
        std::vector<uint32_t> f1, f2;
        for (int a = 0; a < kImg; ++a)
            for (int d = 1; d < kLen; ++d) {
                const int b = a + d;
                if (b >= kImg) continue;
                p1.push_back(a); p2.push_back(b);
                for (int64_t t = 0; t < kTracks; ++t) {              // track t sits on images s .. s + kLen - 1, s = t % (kImg - kLen + 1)
                    const int s0 = (int)(t % (kImg - kLen + 1));
                    if (a >= s0 && b < s0 + kLen) { f1.push_back((uint32_t)t); f2.push_back((uint32_t)t); }
                }
                off.push_back((int64_t)f1.size());
            }
        const size_t M = f1.size();
        std::vector<float> sim(M), flows(18 * M, 0.0f);
        for (size_t m = 0; m < M; ++m) sim[m] = 0.5f + 0.4f * (float)((m * 2654435761u) & 0xffff) / 65536.0f;
        lfr_graph *g = nullptr; lfr_problem *pr = nullptr; lfr_batch *bt = nullptr;
        if (lfr_graph_from_arrays(kImg, name_ptrs.data(), facts.data(), (int64_t)p1.size(), p1.data(), p2.data(), off.data(),
                                  f1.data(), f2.data(), sim.data(), flows.data(), flows.data(), nullptr, 0, &g) == LFR_OK &&
            lfr_problem_build_hip(g, device, 0, nullptr, &pr) == LFR_OK &&
            lfr_batch_create(pr, device, 0, 1, LFR_TUKEY_CERES1, &bt) == LFR_OK) {
            (void)lfr_batch_solve(bt, ctx->s_main, nullptr);
            const double *view = nullptr;
            (void)lfr_batch_positions_view(bt, &view);
        }
        lfr_batch_free(bt); lfr_problem_free(pr); lfr_graph_free(g);
    }
    for (int i = 0; i <= lfr::KC_COUNT; ++i) (void)ctx->side_stream(i);     // last: a few ms each, and only long-track inputs need them
    return LFR_OK;
}

int lfr_hip_reserve(int device, int64_t n_nodes, int64_t n_matches) {
    // Pre-populate the slab caches with what a pipeline run over a graph of this size will ask for, so that the
    // timed span of a one-shot caller (the `solve` launcher knows the sizes once the file is parsed... or guesses
    // them from the file size while it is still parsing) pays no hipMalloc / hipHostMalloc.
    if (n_nodes < 0 || n_matches < 0) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    const int64_t N = n_nodes, M = n_matches;
    const size_t want[5] = {
        (size_t)16 * M + (size_t)4 * N + (size_t)144 * M + ((size_t)1 << 16),                              // DevGraph with staged flows
        (size_t)9 * N + 4096,                                                                             // DevProblem
        (size_t)96 * M + (size_t)112 * N + ((size_t)32 << 20),                                            // graph-stage temporaries
        (size_t)96 * M + (size_t)48 * N + (size_t)128 * (N + 1) + ((size_t)32 << 20),                     // assembly temporaries (C <= N)
        lfr::assembly_output_bytes(N, M, N) + (size_t)(16 + 32) * (size_t)(N + 1) + ((size_t)1 << 17)};   // batch slab
    void *p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t got[5] = {0, 0, 0, 0, 0};
    // (the graph-stage and assembly temporaries are never alive together: the larger of the two serves both)
    for (int i = 0; i < 5; ++i) {
        if (i == 2 && want[3] >= want[2]) continue;
        if (i == 3 && want[2] > want[3]) continue;
        p[i] = ctx->dev_acquire(want[i], &got[i]);
    }
    for (int i = 0; i < 5; ++i) if (p[i]) ctx->dev_release(p[i], got[i]);
    size_t hb = 0;
    if (void *h = ctx->pinned_acquire(sizeof(double) * 2 * (size_t)std::max<int64_t>(N, 1), &hb)) ctx->pinned_release(h, hb);
    return LFR_OK;
}

namespace {
__global__ void eval_edges_kernel(int64_t n, const float *flows, const float *sim, const int32_t *kind, const double *x1,
                                  const double *x2, int tukey_variant, double *out8, double *cost_only) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float fl[18];
    for (int k = 0; k < 18; ++k) fl[k] = flows[18 * i + k];
    EdgeOut o;
    eval_edge<true>(fl, sim[i], kind[i], tukey_variant, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], o);
    double *r = out8 + 8 * i;
    r[0] = o.cost; r[1] = o.r0; r[2] = o.r1; r[3] = o.j00; r[4] = o.j01; r[5] = o.j10; r[6] = o.j11; r[7] = o.sq;
    EdgeOut c;
    eval_edge<false>(fl, sim[i], kind[i], tukey_variant, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], c);
    cost_only[i] = c.cost;
}
}  // namespace

int lfr_debug_eval_edges(int device, int64_t n, const float *flows, const float *sim, const int32_t *kind, const double *x1,
                         const double *x2, int tukey_variant, double *out8, double *cost_only) {
    if (n < 0 || (n > 0 && (!flows || !sim || !kind || !x1 || !x2 || !out8 || !cost_only))) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    if (n == 0) return LFR_OK;
    HIP_TRY(hipSetDevice(device));
    lfr::DevArena ar;
    if (!ar.init(ctx, (size_t)n * (72 + 4 + 4 + 16 + 16 + 64 + 8) + 4096)) return LFR_ERR_NOMEM;
    float *d_fl = ar.take_n<float>(18 * n), *d_sim = ar.take_n<float>(n);
    int32_t *d_kind = ar.take_n<int32_t>(n);
    double *d_x1 = ar.take_n<double>(2 * n), *d_x2 = ar.take_n<double>(2 * n), *d_out = ar.take_n<double>(8 * n), *d_c = ar.take_n<double>(n);
    if (!d_c) { lfr::set_error("arena exhausted"); return LFR_ERR_NOMEM; }
    hipStream_t st = ctx->s_main;
    HIP_TRY(hipMemcpyAsync(d_fl, flows, 72 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_sim, sim, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_kind, kind, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_x1, x1, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_x2, x2, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(eval_edges_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, d_fl, d_sim, d_kind, d_x1, d_x2, tukey_variant, d_out, d_c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out8, d_out, 64 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cost_only, d_c, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(lfr::stream_wait(st));
    return LFR_OK;
}

namespace {
// samples: 15 doubles per case = (x, value, gradient, value_valid, gradient_valid) of the initial, previous and current sample
__global__ void ls_next_step_kernel(int64_t n, const double *samples, const double *dir_max, int register_version, double *step) {
    // (version 2 - the wave-cooperative form - takes one case per 64-thread workgroup, every lane with the same arguments)
    const int64_t i = register_version == 2 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LsSample s[3];
    for (int k = 0; k < 3; ++k) {
        const double *q = samples + 15 * i + 5 * k;
        s[k].x = q[0]; s[k].value = q[1]; s[k].gradient = q[2]; s[k].value_valid = q[3] != 0.0; s[k].gradient_valid = q[4] != 0.0;
    }
    int it0 = 0;
    if (register_version == 2) { const double v = ls_next_step_wave(s[0], s[1], s[2], dir_max[i], it0); if (threadIdx.x == 0) step[i] = v; }
    else step[i] = register_version ? ls_next_step_regs(s[0], s[1], s[2], dir_max[i], it0) : ls_next_step(s[0], s[1], s[2], dir_max[i], it0);
}
}  // namespace

int lfr_debug_ls_next_step(int device, int64_t n, const double *samples, const double *dir_max, int register_version, double *step) {
    if (n < 0 || (n > 0 && (!samples || !dir_max || !step))) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    if (n == 0) return LFR_OK;
    HIP_TRY(hipSetDevice(device));
    lfr::DevArena ar;
    if (!ar.init(ctx, (size_t)n * 8 * 18 + 4096)) return LFR_ERR_NOMEM;
    double *d_s = ar.take_n<double>(15 * n), *d_d = ar.take_n<double>(n), *d_a = ar.take_n<double>(n);
    if (!d_a) { lfr::set_error("arena exhausted"); return LFR_ERR_NOMEM; }
    hipStream_t st = ctx->s_main;
    HIP_TRY(hipMemcpyAsync(d_s, samples, 120 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_d, dir_max, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ls_next_step_kernel, dim3((unsigned)(register_version == 2 ? n : (n + 63) / 64)), dim3(64), 0, st, n, d_s, d_d, register_version, d_a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(step, d_a, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(lfr::stream_wait(st));
    return LFR_OK;
}

int lfr_hip_synchronize(int device) {
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(ctx->s_copy));
    HIP_TRY(hipStreamSynchronize(ctx->s_main));
    return LFR_OK;
}

int lfr_hip_trim(int device) {
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    ctx->trim();
    return LFR_OK;
}

void lfr_batch_free(lfr_batch *b) { delete b; }

int lfr_batch_create(const lfr_problem *ph, int device, int shard_rank, int shard_world, int tukey_variant,
                     lfr_batch **out) {
    if (!ph || !out || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world ||
        (tukey_variant != LFR_TUKEY_CERES1 && tukey_variant != LFR_TUKEY_CERES2)) {
        lfr::set_error("bad argument"); return LFR_ERR_ARG;
    }
    *out = nullptr;
    const lfr::Problem &p = ph->p;
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<lfr_batch> b(new lfr_batch());            // every error path below releases what was acquired
    b->device = device; b->ctx = ctx; b->tukey_variant = tukey_variant; b->shard_world = shard_world;
    { const char *e = getenv("LFR_SERIAL_CLASSES"); b->serial = e && e[0] == '1'; }
    b->n_graph_nodes = p.g->n_nodes();
    int rc = p.host_batch ? create_from_host(b.get(), p, shard_rank, shard_world) : create_on_device(b.get(), p, shard_rank, shard_world);
    if (rc != LFR_OK) return rc;
    if ((rc = finish_workspace(b.get(), p)) != LFR_OK) return rc;
    host_trace("lfr_batch_create: workspace done");
    if (!(b->ev_fork = ctx->event_acquire(false))) return LFR_ERR_HIP;
    if (b->class_begin[lfr::KC_COUNT] > b->class_begin[lfr::KC_BLOCK]) {       // workgroup classes run beside the packed launch
        if (!(b->side_stream = ctx->side_stream(0))) return LFR_ERR_HIP;
        for (int cls = lfr::KC_BLOCK; cls < lfr::KC_COUNT; ++cls)
            if (b->class_begin[cls + 1] > b->class_begin[cls] && !(b->wg_stream[cls] = ctx->side_stream(1 + cls))) return LFR_ERR_HIP;
        {   // hand-out order of the persistent launches
            const int wg_begin = b->class_begin[lfr::KC_BLOCK], n_wg = b->class_begin[lfr::KC_COUNT] - wg_begin;
            hipStream_t so = ctx->s_main;
            size_t tmp_bytes = 0;
            HIP_TRY(rocprim::radix_sort_pairs<LfrRadixSortConfig>(nullptr, tmp_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                                                   (uint32_t *)nullptr, (size_t)n_wg, 0u, 27u, so));
            if (!b->order_slab.init(ctx, 16 * (size_t)n_wg + tmp_bytes + 4096)) return LFR_ERR_NOMEM;
            b->d_wg_order = b->order_slab.take_n<uint32_t>(n_wg);
            uint32_t *keys = b->order_slab.take_n<uint32_t>(n_wg), *keys_sorted = b->order_slab.take_n<uint32_t>(n_wg), *vals = b->order_slab.take_n<uint32_t>(n_wg);
            void *tmp = b->order_slab.take(tmp_bytes);
            if (!b->d_wg_order || !keys || !keys_sorted || !vals || !tmp) { lfr::set_error("order slab exhausted"); return LFR_ERR_NOMEM; }
            hipLaunchKernelGGL(k_wg_order_keys, dim3((n_wg + 255) / 256), dim3(256), 0, so, b->d_descs + wg_begin, n_wg,
                               b->class_begin[lfr::KC_BLOCK_M] - wg_begin, b->class_begin[lfr::KC_BLOCK_L] - wg_begin, b->class_begin[lfr::KC_GLOBAL] - wg_begin,
                               keys, vals, wg_begin);
            HIP_TRY(hipGetLastError());
            HIP_TRY(rocprim::radix_sort_pairs<LfrRadixSortConfig>(tmp, tmp_bytes, keys, keys_sorted, vals, b->d_wg_order, (size_t)n_wg, 0u, 27u, so));
            if (!(b->ev_order = ctx->event_acquire(false))) return LFR_ERR_HIP;
            HIP_TRY(hipEventRecord(b->ev_order, so));
        }
        host_trace("lfr_batch_create: hand-out order enqueued");
        const int lds_s = (int)block_lds_bytes(std::max(b->class_max_rows[lfr::KC_BLOCK], 2), false);
        const int lds_m = (int)block_lds_bytes(std::max(b->class_max_rows[lfr::KC_BLOCK_M], 2), false);
        const int lds_l = (int)block_lds_bytes(std::max(b->class_max_rows[lfr::KC_BLOCK_L], 2), false);
        HIP_TRY(hipFuncSetAttribute((const void *)solve_block_kernel<kThreadsS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
        HIP_TRY(hipFuncSetAttribute((const void *)solve_block_kernel<kThreadsM>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds_m, kThreadsM == kThreadsS ? lds_s : 0)));
        HIP_TRY(hipFuncSetAttribute((const void *)solve_block_kernel<kThreadsL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    std::max(lds_l, std::max(kThreadsL == kThreadsM ? lds_m : 0, kThreadsL == kThreadsS ? lds_s : 0))));
    }
    {   // the packed launch is reported in the slot of its largest class (by edges)
        int64_t best = -1;
        for (int cls = 0; cls < lfr::KC_BLOCK; ++cls) if (b->class_edges[cls] > best) { best = b->class_edges[cls]; b->packed_slot = cls; }
    }
    *out = b.release();
    host_trace("lfr_batch_create returns");
    return LFR_OK;
}

int lfr_batch_solve(lfr_batch *b, void *hip_stream, lfr_solve_stats *stats) {
    if (!b) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    host_trace("lfr_batch_solve enters");
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)hip_stream;
    KernelArgs a;
    a.descs = b->d_descs; a.edges = b->d_edges; a.node_ids = b->d_node_ids; a.positions = b->d_positions;
    a.infos = b->d_infos; a.workspace = b->d_workspace; a.ws_off = b->d_ws_off; a.es_off = b->d_es_off;
    a.node_inc = b->d_node_inc; a.in_idx = b->d_in_idx; a.tukey_variant = b->tukey_variant; a.prof = b->d_prof;
    { const char *e = getenv("LFR_SCRATCH_SWEEP"); a.scratch_sweep = (e && e[0] == '1') ? 1 : 0; }
    a.queue = reinterpret_cast<unsigned int *>(b->d_prof + 8 * lfr::KC_COUNT);
    a.wg_order = b->d_wg_order; a.wg_begin = b->class_begin[lfr::KC_BLOCK];
    a.edge_ref = b->d_edge_ref; a.edge_word = b->d_edge_word;
    a.f_row = nullptr; a.f_disp1 = a.f_disp2 = a.f_sim = nullptr;
    a.team_ctl = b->d_team_ctl; a.team_red = b->d_team_red;
    a.trace = nullptr;
#ifdef LFR_TRACE_TREE
    static unsigned long long *d_trace = nullptr;
    if (!d_trace) { HIP_TRY(hipMalloc(&d_trace, ((size_t)1 << 20) * 8 + 64)); }
    HIP_TRY(hipMemsetAsync(d_trace, 0, 64, st));
    a.trace = d_trace;
#endif
    for (int k = 0; k < 3; ++k) a.team_work[k] = b->team_work[k];
    a.team_epoch = (uint32_t)(b->n_solves + 1);
    a.team_patience_us = b->team_patience_us;
    bool materialised = false;
    if (b->fused) {
        const lfr::DevGraph &dgr = *b->dev_hold->graph;
        a.f_row = dgr.flow_row; a.f_disp1 = dgr.disp1; a.f_disp2 = dgr.disp2; a.f_sim = dgr.sim;
        if (b->n_solves > 0) {           // solved before: this batch is being re-used - write the records once, read them from now on
            if (b->packed_edges)
                hipLaunchKernelGGL(k_materialize_records, dim3((unsigned)(((uint64_t)5 * b->packed_edges + 255) / 256)), dim3(256), 0, st, b->packed_edges,
                                   b->d_edge_ref, b->d_edge_word, dgr.flow_row, dgr.disp1, dgr.disp2, dgr.sim, reinterpret_cast<uint4 *>(b->d_edges));
            HIP_TRY(hipGetLastError());
            b->fused = false;
            materialised = true;
        }
    }
    b->ev = b->ev_ring + (b->n_solves % lfr_batch::kSlots) * lfr_batch::kEvPerSlot;
    uint32_t &recorded = b->ev_recorded[b->n_solves % lfr_batch::kSlots];
    recorded = 0;
    ++b->n_solves;
    b->last_stream = st;
    if (!b->ev[0]) for (int i = 0; i < lfr_batch::kEvPerSlot; ++i) if (!(b->ev[i] = b->ctx->event_acquire(true))) return LFR_ERR_HIP;
    HIP_TRY(hipEventRecord(b->ev[0], st));
    if (b->class_begin[lfr::KC_COUNT] > b->class_begin[lfr::KC_BLOCK]) {
        HIP_TRY(hipStreamWaitEvent(st, b->ev_order, 0));
        HIP_TRY(hipMemsetAsync(a.queue, 0, 64, st));                       // the classes' component queues
        if (b->d_team_ctl) HIP_TRY(hipMemsetAsync(b->d_team_ctl, 0, kTeamCtlWords * sizeof(unsigned int), st));   // registrations, mailboxes, barrier counters of the teams
    }

    // The packed classes go out as ONE launch on the caller's stream (solve_packed_kernel); the few
    // workgroup-per-component problems run beside it on a side stream.  LFR_SERIAL_CLASSES=1
    // launches every class separately on the caller's stream (per-class timings for diagnostics).
    static const int kPackedOrder[5] = {lfr::KC_G64_4, lfr::KC_G64_2, lfr::KC_G32, lfr::KC_G16, lfr::KC_G8};
    static const int kCompsPerBlock[lfr::KC_COUNT] = {8 * kPackedWaves, 4 * kPackedWaves, 2 * kPackedWaves, 2 * kPackedWaves, kPackedWaves, 1, 1, 1, 1};
    const dim3 blk(64 * kPackedWaves);
    auto launch_block = [&](int cls, hipStream_t cs) -> int {
        a.desc_begin = b->class_begin[cls]; a.desc_end = b->class_begin[cls + 1]; a.cls = cls;
        const int n = a.desc_end - a.desc_begin;
        if (n <= 0) return LFR_OK;
        const int rows = b->class_max_rows[cls];
        const size_t lds = block_lds_bytes(rows, cls == lfr::KC_GLOBAL);
        // persistent workgroups: what the chip keeps resident for the class (by LDS, and 8 waves of 256 registers per CU)
        const int threads = cls == lfr::KC_BLOCK ? kThreadsS : cls == lfr::KC_BLOCK_M ? kThreadsM : cls == lfr::KC_BLOCK_L ? kThreadsL : kThreadsG;
        const int by_waves = std::max(1, 512 / threads), by_lds = lds ? std::max(1, (int)((size_t)160 * 1024 / (lds + 256))) : by_waves;
        const int wgs = std::min(n, b->ctx->n_cu * std::min(by_waves, by_lds));
        switch (cls) {
            case lfr::KC_BLOCK:   hipLaunchKernelGGL((solve_block_kernel<kThreadsS>), dim3(wgs), dim3(kThreadsS), lds, cs, a, rows); break;
            case lfr::KC_BLOCK_M: hipLaunchKernelGGL((solve_block_kernel<kThreadsM>), dim3(wgs), dim3(kThreadsM), lds, cs, a, rows); break;
            case lfr::KC_BLOCK_L: hipLaunchKernelGGL((solve_block_kernel<kThreadsL>), dim3(wgs), dim3(kThreadsL), lds, cs, a, rows); break;
            default: {
                if (const char *e = getenv("LFR_DEBUG_TREE_FIRST")) a.desc_end = std::min(a.desc_end, a.desc_begin + std::max(1, atoi(e)));   // (experiments: only the first k of the hand-out order)
                // elimination-tree kernel: one 512-thread workgroup per CU (two waves per SIMD), static LDS only
                // (at least 8 kTeamMax workgroups: whatever the placement, one of the eight XCDs then holds a complete unit)
                if (b->d_team_ctl) hipLaunchKernelGGL((solve_tree_team_kernel<kThreadsG>), dim3(std::max(8 * kTeamMax, std::min(b->team_wgs, b->ctx->n_cu / 32 * 32))), dim3(kThreadsG), 0, cs, a);
                else hipLaunchKernelGGL((solve_tree_kernel<kThreadsG>), dim3(std::min(n, b->ctx->n_cu)), dim3(kThreadsG), 0, cs, a);
                break;
            }
        }
        HIP_TRY(hipGetLastError());
        return LFR_OK;
    };
    const bool have_side = b->class_begin[lfr::KC_COUNT] > b->class_begin[lfr::KC_BLOCK];
    if (b->serial) {
        for (int cls = 0; cls < lfr::KC_COUNT; ++cls) {
            a.desc_begin = b->class_begin[cls]; a.desc_end = b->class_begin[cls + 1]; a.cls = cls;
            const int n = a.desc_end - a.desc_begin;
            if (n <= 0) continue;
            recorded |= 1u << cls;
            HIP_TRY(hipEventRecord(b->ev[2 + 2 * cls], st));
            {
                const dim3 grid((n + kCompsPerBlock[cls] - 1) / kCompsPerBlock[cls]);
                switch (cls) {
                    case lfr::KC_G8:    if (b->fused) hipLaunchKernelGGL((solve_group_kernel<8, 1, 3, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((solve_group_kernel<8, 1, 3, false>), grid, blk, 0, st, a); break;
                    case lfr::KC_G16:   if (b->fused) hipLaunchKernelGGL((solve_group_kernel<16, 1, 6, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((solve_group_kernel<16, 1, 6, false>), grid, blk, 0, st, a); break;
                    case lfr::KC_G32:   break;     // retired class, never assigned
                    case lfr::KC_G64_2: if (b->fused) hipLaunchKernelGGL((solve_group_kernel<32, 1, 6, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((solve_group_kernel<32, 1, 6, false>), grid, blk, 0, st, a); break;
                    case lfr::KC_G64_4: if (b->fused) hipLaunchKernelGGL((solve_group_kernel<32, 2, 5, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((solve_group_kernel<32, 2, 5, false>), grid, blk, 0, st, a); break;
                    default: { const int rc = launch_block(cls, st); if (rc != LFR_OK) return rc; }
                }
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipEventRecord(b->ev[3 + 2 * cls], st));
        }
    } else {
        // Launch plan.  A workgroup-per-component kernel needs a (nearly) empty CU for each of its
        // 512-thread workgroups; once the packed launch has flooded the chip such a workgroup only gets a
        // CU at the packed launch's tail, i.e. the two kernels would run back to back.  So when big-workgroup
        // classes exist THEY go first, on the caller's stream, and the packed launch follows from a side
        // stream (its cross-queue wait makes it the later dispatch) and fills the remaining CUs.
        // Events are barrier packets on their stream: only the launches that exist are bracketed.
        PackedRanges r;
        int nb = 0;
        for (int i = 0; i < 5; ++i) {
            const int cls = kPackedOrder[i];
            r.blk_begin[i] = nb;
            r.desc_begin[i] = b->class_begin[cls]; r.desc_end[i] = b->class_begin[cls + 1];
            const int n = r.desc_end[i] - r.desc_begin[i];
            nb += (n + kCompsPerBlock[cls] - 1) / kCompsPerBlock[cls];
        }
        r.blk_begin[5] = nb;
        // the packed launch is timed as one unit: its events sit in the slot of the largest class
        // (the only launch of the solve - no workgroup class, no records to materialise in front: the solve's own pair of events brackets
        // exactly this kernel, a second pair would be two more barrier packets per solve for the same two time stamps)
        const bool alias = !have_side && !materialised;
        auto launch_packed = [&](hipStream_t cs) -> int {
            if (!alias) HIP_TRY(hipEventRecord(b->ev[2 + 2 * b->packed_slot], cs));
            if (b->fused) hipLaunchKernelGGL(solve_packed_kernel<true>, dim3(nb), blk, 0, cs, a, r);
            else hipLaunchKernelGGL(solve_packed_kernel<false>, dim3(nb), blk, 0, cs, a, r);
            HIP_TRY(hipGetLastError());
            if (!alias) HIP_TRY(hipEventRecord(b->ev[3 + 2 * b->packed_slot], cs));
            recorded |= 1u << b->packed_slot;
            if (alias) recorded |= kPackedEventsAliased;
            return LFR_OK;
        };
        auto launch_big = [&](int cls, hipStream_t cs) -> int {
            HIP_TRY(hipEventRecord(b->ev[2 + 2 * cls], cs));
            const int rc = launch_block(cls, cs);
            if (rc != LFR_OK) return rc;
            HIP_TRY(hipEventRecord(b->ev[3 + 2 * cls], cs));
            recorded |= 1u << cls;
            return LFR_OK;
        };
        if (!have_side) {
            if (nb > 0) { const int rc = launch_packed(st); if (rc != LFR_OK) return rc; }
        } else {
            // Every workgroup class on its own stream (the first on the caller's).  Dispatch order = issue order: the HBM-matrix
            // class first (its components run longest), then the 130-row class, the 192-row class, the 88-row class.  A 160 KB
            // workgroup only starts on an empty CU: issued first, the largest class kept the others out until its queue drained,
            // and the one slow component among THEM (iteration counts vary 8x) then ended the solve alone (config 5: 17.0 ms
            // against 13.8 at the time).  With persistent workgroups the middle class takes the whole chip for its 1.5 ms, the
            // large class follows, and the small class fills the CUs the large one's tail leaves (measured 9.15 ms against
            // 9.4-9.5 smallest-first and 12.4 largest-first); LFR_WG_ORDER overrides for experiments.  Round 3, with the fused sweep
            // (5.1 ms): largest-first 7.9 ms; largest-first with only its share of the CUs (by rows x threads x components) and full
            // grids behind it 5.8-7.2 ms - the pending workgroups of the later launches do not take the CUs the first one frees, and
            // the packed launch starves; the order stays.
            static const std::array<int, 4> kBigOrder = [] {
                std::array<int, 4> o = {lfr::KC_GLOBAL, lfr::KC_BLOCK_M, lfr::KC_BLOCK_L, lfr::KC_BLOCK};
                if (const char *e = getenv("LFR_WG_ORDER")) {
                    int v[4];
                    if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) {
                        unsigned seen = 0;
                        for (int i = 0; i < 4; ++i) if (v[i] >= lfr::KC_BLOCK && v[i] < lfr::KC_COUNT) seen |= 1u << v[i];
                        if (seen == (0xfu << lfr::KC_BLOCK)) for (int i = 0; i < 4; ++i) o[i] = v[i];
                    }
                }
                return o;
            }();
            int first = -1, n_big = 0;
            for (int i = 0; i < 4; ++i) if (b->class_begin[kBigOrder[i] + 1] > b->class_begin[kBigOrder[i]]) { if (first < 0) first = kBigOrder[i]; ++n_big; }
            if (n_big > 1 || nb > 0) HIP_TRY(hipEventRecord(b->ev_fork, st));
            { const int rc = launch_big(first, st); if (rc != LFR_OK) return rc; }
            for (int i = 0; i < 4; ++i) {
                const int cls = kBigOrder[i];
                if (cls == first || b->class_begin[cls + 1] <= b->class_begin[cls]) continue;
                HIP_TRY(hipStreamWaitEvent(b->wg_stream[cls], b->ev_fork, 0));
                const int rc = launch_big(cls, b->wg_stream[cls]);
                if (rc != LFR_OK) return rc;
            }
            if (nb > 0) {
                HIP_TRY(hipStreamWaitEvent(b->side_stream, b->ev_fork, 0));
                const int rc = launch_packed(b->side_stream);
                if (rc != LFR_OK) return rc;
                HIP_TRY(hipStreamWaitEvent(st, b->ev[3 + 2 * b->packed_slot], 0));
            }
            for (int i = 0; i < 4; ++i) {
                const int cls = kBigOrder[i];
                if (cls == first || b->class_begin[cls + 1] <= b->class_begin[cls]) continue;
                HIP_TRY(hipStreamWaitEvent(st, b->ev[3 + 2 * cls], 0));
            }
        }
    }
    HIP_TRY(hipEventRecord(b->ev[1], st));
    b->infos_valid = false;
    host_trace("lfr_batch_solve: launched");
    if (!stats) return LFR_OK;

    HIP_TRY(lfr::stream_wait(st));
#ifdef LFR_TRACE_TREE
    if (const char *tf = getenv("LFR_TREE_TRACE_FILE")) {
        unsigned long long n = 0;
        HIP_TRY(hipMemcpy(&n, a.trace, 8, hipMemcpyDeviceToHost));
        n = std::min<unsigned long long>(n, (1ull << 20) - 4);
        std::vector<unsigned long long> h(n + 2);
        HIP_TRY(hipMemcpy(h.data(), a.trace, (n + 2) * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(tf, "wb")) { fwrite(h.data() + 2, 8, n, f); fclose(f); }
    }
#endif
#ifdef LFR_PROFILE_FACTOR
    {
        unsigned long long h[64];
        HIP_TRY(hipMemcpy(h, b->d_prof + 8 * lfr::KC_COUNT + 8, sizeof h, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(b->d_prof + 8 * lfr::KC_COUNT + 8, 0, sizeof h));
        for (int c = 0; c < 4; ++c) for (int w = 0; w < 2; ++w) if (h[16 * c + 8 * w + 7])
            fprintf(stderr, "lfr-fprof class %d wave %d: factorizations %llu  cycles each: diag %.0f  trailing %.0f  wait %.0f  col-update %.0f  col-finish %.0f  barrier %.0f   (wave 0: diag = loads + elimination, trailing = its stores)\n",
                    lfr::KC_BLOCK + c, w, h[16 * c + 8 * w + 7], (double)h[16 * c + 8 * w] / h[16 * c + 8 * w + 7], (double)h[16 * c + 8 * w + 1] / h[16 * c + 8 * w + 7],
                    (double)h[16 * c + 8 * w + 2] / h[16 * c + 8 * w + 7], (double)h[16 * c + 8 * w + 3] / h[16 * c + 8 * w + 7],
                    (double)h[16 * c + 8 * w + 4] / h[16 * c + 8 * w + 7], (double)h[16 * c + 8 * w + 5] / h[16 * c + 8 * w + 7]);
    }
#endif
#ifdef LFR_PROFILE_PHASES
    {
        unsigned long long h[8 * lfr::KC_COUNT];
        HIP_TRY(hipMemcpy(h, b->d_prof, sizeof h, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(b->d_prof, 0, sizeof h));
        for (int c = 0; c < lfr::KC_COUNT; ++c) if (h[c * 8 + 7])
            fprintf(stderr, "lfr-prof class %d: waves %llu  per-wave cycles: [0] %.0f  [1] %.0f  [2] %.0f  [3] %.0f  [4] %.0f  [5] %.0f  [6] %.0f   (packed: prologue/elim/sweep/reduce/transitions/setup/zero; block: jac-sweeps/factor/ls-sweeps/-/bookkeeping/scaling/trisolve)\n", c,
                    h[c * 8 + 7], (double)h[c * 8] / h[c * 8 + 7], (double)h[c * 8 + 1] / h[c * 8 + 7], (double)h[c * 8 + 2] / h[c * 8 + 7],
                    (double)h[c * 8 + 3] / h[c * 8 + 7], (double)h[c * 8 + 4] / h[c * 8 + 7], (double)h[c * 8 + 5] / h[c * 8 + 7], (double)h[c * 8 + 6] / h[c * 8 + 7]);
    }
#endif
    {   // bounded spins that ran out (wave hand-offs of the factorizations, the teams' barriers): a rejected LM step or a failed
        // component instead of a hung GPU - visible under LFR_VERBOSE, an ERROR under LFR_SPIN_TIMEOUT_FATAL=1 (the GPU tests set it)
        static const bool verbose = getenv("LFR_VERBOSE") != nullptr;
        static const bool fatal = [] { const char *e = getenv("LFR_SPIN_TIMEOUT_FATAL"); return e && e[0] == '1'; }();
        if ((verbose || fatal) && b->class_begin[lfr::KC_COUNT] > b->class_begin[lfr::KC_BLOCK]) {
            unsigned int v = 0;
            HIP_TRY(hipMemcpy(&v, a.queue + 15, sizeof v, hipMemcpyDeviceToHost));
            if (v && verbose) fprintf(stderr, "lfr: %u bounded spin-wait(s) of the workgroup kernels ran out during this solve (rejected LM steps / failed components)\n", v);
            if (v && fatal) { lfr::set_error("%u bounded spin-wait(s) of the workgroup kernels ran out (LFR_SPIN_TIMEOUT_FATAL=1)", v); return LFR_ERR_HIP; }
        }
    }
    memset(stats, 0, sizeof *stats);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
    stats->kernel_ms = ms;
    stats->h2d_ms = b->h2d_ms;
    { const int rc = ensure_mirrors(b); if (rc != LFR_OK) return rc; }
    b->infos.resize(b->descs.size());
    if (!b->descs.empty()) {
        HIP_TRY(hipMemcpyAsync(b->infos.data(), b->d_infos, b->descs.size() * sizeof(CompInfoDev), hipMemcpyDeviceToHost, st));
        HIP_TRY(lfr::stream_wait(st));
    }
    b->infos_valid = true;
    stats->n_components = (int64_t)b->descs.size();
    stats->n_edges = b->n_edges; stats->n_nodes = b->n_nodes; stats->n_tracks = b->n_tracks;
    double best_ms = -1.0;
    int64_t edges = 0, nodes = 0, refp_e = 0, refp_n = 0;
    for (int cls = 0; cls < lfr::KC_COUNT; ++cls) {
        // one accounting unit per kernel LAUNCH: the packed classes are one launch unless serial
        const bool merged = !b->serial && cls < lfr::KC_BLOCK;
        if (!merged || cls == 0) { edges = 0; nodes = 0; refp_e = 0; refp_n = 0; }
        for (int i = b->class_begin[cls]; i < b->class_begin[cls + 1]; ++i) {
            const CompInfoDev &f = b->infos[i];
            const int64_t E = b->descs[i].n_edges, N = b->descs[i].n_nodes;
            const int64_t jac = 1 + f.n_ls_evals + f.n_successful, cst = f.n_cand_evals;
            stats->ref_jacobian_passes_edges += E * jac;
            stats->ref_cost_passes_edges += E * cst;
            stats->ref_passes_nodes += N * (jac + cst);
            stats->exec_passes_edges += E * f.exec_passes;
            stats->sum_iterations += f.iterations;
            stats->sum_final_cost += f.final_cost;
            if (f.termination == LFR_TERM_CONVERGENCE) ++stats->n_converged;
            else if (f.termination == LFR_TERM_NO_CONVERGENCE) ++stats->n_no_convergence;
            else ++stats->n_failed;
            edges += E; nodes += N; refp_e += E * (jac + cst); refp_n += N * (jac + cst);
        }
        if (merged && cls != lfr::KC_BLOCK - 1) continue;
        const int slot = merged ? b->packed_slot : cls;
        ms = 0.f;
        if (recorded >> slot & 1u) {
            const bool al = merged && (recorded & kPackedEventsAliased);
            HIP_TRY(hipEventElapsedTime(&ms, al ? b->ev[0] : b->ev[2 + 2 * slot], al ? b->ev[1] : b->ev[3 + 2 * slot]));
        }
        if (edges > 0 && ms > best_ms) {
            best_ms = ms;
            stats->dominant_kernel_ms = ms; stats->dominant_kernel_edges = edges; stats->dominant_kernel_nodes = nodes;
            stats->dominant_ref_passes_edges = refp_e; stats->dominant_ref_passes_nodes = refp_n;
        }
    }
    return LFR_OK;
}

// Bounded spins that ran out during the batch's latest solve (the hand-off flags of the LDS factorization, the dependency counters of
// the elimination-tree kernel): a timeout rejects an LM step instead of hanging the GPU, so it must be visible - 0 in every test.
int64_t lfr_batch_spin_timeouts(lfr_batch *b) {
    if (!b) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    if (b->n_solves == 0 || !b->d_prof) return 0;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    unsigned int v = 0;
    HIP_TRY(hipMemcpy(&v, reinterpret_cast<unsigned int *>(b->d_prof + 8 * lfr::KC_COUNT) + 15, sizeof v, hipMemcpyDeviceToHost));
    return (int64_t)v;
}

// Components the latest solve handed to a TEAM of two or more workgroups (solve_tree_team_kernel); 0 when the batch has none above the
// thresholds (LFR_TREE_TEAM) or no elimination-tree class at all.
int64_t lfr_batch_team_runs(lfr_batch *b) {
    if (!b) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    if (b->n_solves == 0 || !b->d_team_ctl) return 0;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    unsigned int v = 0;
    HIP_TRY(hipMemcpy(&v, b->d_team_ctl + 10, sizeof v, hipMemcpyDeviceToHost));
    return (int64_t)v;
}

// Components the latest solve's teams could not serve at their size and a single workgroup solved instead (SOLO mode: the comment above
// TeamCtx, Residency) - 0 whenever the launch's workgroups were resident together.
int64_t lfr_batch_team_fallbacks(lfr_batch *b) {
    if (!b) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    if (b->n_solves == 0 || !b->d_team_ctl) return 0;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    unsigned int v = 0;
    HIP_TRY(hipMemcpy(&v, b->d_team_ctl + 13, sizeof v, hipMemcpyDeviceToHost));
    return (int64_t)v;
}

// Test infrastructure: `workgroups` 512-thread workgroups of 256 registers per lane (a whole CU each, like the elimination-tree kernel's)
// that do nothing but stay resident for `milliseconds`, on a stream of their own; returns once they have started.
namespace {
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_occupy(unsigned long long ticks, unsigned int *started) {
    extern __shared__ unsigned int occ_lds[];                 // (100 KB of dynamic LDS: no second large workgroup fits the CU either way)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { occ_lds[0] = 1u; atomicAdd(started, occ_lds[0]); }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}
int lfr_debug_occupy(int device, int workgroups, double milliseconds) {
    if (workgroups < 1 || workgroups > 4096 || !(milliseconds > 0.0) || milliseconds > 2000.0) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr::DevCtx *ctx = lfr::dev_ctx(device);
    if (!ctx) return LFR_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    static hipStream_t s_occ[16] = {};
    static unsigned int *h_started[16] = {};
    if (device < 0 || device >= 16) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    if (!s_occ[device]) {
        HIP_TRY(hipStreamCreateWithFlags(&s_occ[device], hipStreamNonBlocking));
        HIP_TRY(hipHostMalloc((void **)&h_started[device], 64, hipHostMallocMapped));
    }
    HIP_TRY(hipStreamSynchronize(s_occ[device]));            // (an earlier occupation has ended)
    volatile unsigned int *seen = h_started[device];
    *seen = 0u;
    unsigned int *d_started = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&d_started, h_started[device], 0));
    HIP_TRY(hipFuncSetAttribute((const void *)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(k_occupy, dim3((unsigned)workgroups), dim3(512), 100 * 1024, s_occ[device], (unsigned long long)(milliseconds * 1e5), d_started);
    HIP_TRY(hipGetLastError());
    // the workgroups that fit are resident once the count stops growing: at most n_cu of them at a time
    const unsigned want = (unsigned)std::min(workgroups, ctx->n_cu);
    const auto t0 = std::chrono::steady_clock::now();
    while (*seen < want && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < std::min(milliseconds, 100.0)) std::this_thread::yield();
    return LFR_OK;
}

int lfr_batch_timing(lfr_batch *b, int solves_back, double *total_ms, double *class_ms, int64_t *class_edges) {
    if (!b || solves_back < 0 || solves_back >= lfr_batch::kSlots || solves_back >= b->n_solves) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    HIP_TRY(hipSetDevice(b->device));
    const int ring_slot = (b->n_solves - 1 - solves_back) % lfr_batch::kSlots;
    hipEvent_t *ev = b->ev_ring + ring_slot * lfr_batch::kEvPerSlot;
    const uint32_t recorded = b->ev_recorded[ring_slot];
    HIP_TRY(hipEventSynchronize(ev[1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
    if (total_ms) *total_ms = ms;
    for (int cls = 0; cls < lfr::KC_COUNT; ++cls) {
        if (class_ms) {
            ms = 0.f;
            if (recorded >> cls & 1u) {
                const bool al = cls == b->packed_slot && !b->serial && (recorded & kPackedEventsAliased);
                HIP_TRY(hipEventElapsedTime(&ms, al ? ev[0] : ev[2 + 2 * cls], al ? ev[1] : ev[3 + 2 * cls]));
            }
            class_ms[cls] = ms;
        }
        if (class_edges) {      // edges of the LAUNCH timed in this slot (the packed launch carries all packed classes)
            int64_t e = 0;
            const bool packed = !b->serial && cls < lfr::KC_BLOCK;
            const int lo = packed ? (cls == b->packed_slot ? 0 : cls + 1) : cls, hi = packed ? (cls == b->packed_slot ? lfr::KC_BLOCK : cls + 1) : cls + 1;
            for (int c = lo; c < hi; ++c) e += b->class_edges[c];
            class_edges[cls] = e;
        }
    }
    return LFR_OK;
}

// The positions of the whole graph in pinned host memory, valid until the next solve / download / free of this
// batch: waits for the latest solve, one D2H at link speed, no further copy.  Nodes outside this batch's shard
// read 0.
int lfr_batch_positions_view(lfr_batch *b, const double **positions) {
    if (!b || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    HIP_TRY(hipSetDevice(b->device));
    const size_t bytes = sizeof(double) * 2 * (size_t)std::max<int64_t>(b->n_graph_nodes, 1);
    if (!b->h_positions) {
        b->h_positions = (double *)b->ctx->pinned_acquire(bytes, &b->h_positions_bytes);
        if (!b->h_positions) return LFR_ERR_NOMEM;
    }
    hipStream_t st = b->ctx->s_main;
    host_trace("positions_view enters");
    if (b->n_solves > 0) HIP_TRY(hipStreamWaitEvent(st, b->ev[1], 0));        // end of the latest solve, whatever stream it ran on
    HIP_TRY(hipMemcpyAsync(b->h_positions, b->d_positions, bytes, hipMemcpyDeviceToHost, st));
    host_trace("positions_view: copy issued");
    HIP_TRY(lfr::stream_wait(st));
    host_trace("positions_view: copy done");
    *positions = b->h_positions;
    return LFR_OK;
}

namespace { __global__ void k_positions_to_f32(int64_t n, const double *in, float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {            // n is even (two coordinates per node): one 16-byte load, one 8-byte store per thread
        const double2 v = reinterpret_cast<const double2 *>(in)[i];
        reinterpret_cast<float2 *>(out)[i] = make_float2((float)v.x, (float)v.y);
    }
} }

int lfr_batch_positions_view_f32(lfr_batch *b, const float **positions) {
    if (!b || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    HIP_TRY(hipSetDevice(b->device));
    const int64_t n = 2 * std::max<int64_t>(b->n_graph_nodes, 1);
    const size_t bytes = sizeof(float) * (size_t)n;
    if (!b->h_positions_f32) {
        b->h_positions_f32 = (float *)b->ctx->pinned_acquire(bytes, &b->h_positions_f32_bytes);
        if (!b->h_positions_f32) return LFR_ERR_NOMEM;
    }
    if (!b->d_positions_f32) {
        b->d_positions_f32 = (float *)b->ctx->dev_acquire(bytes, &b->d_positions_f32_bytes);
        if (!b->d_positions_f32) return LFR_ERR_NOMEM;
    }
    hipStream_t st = b->ctx->s_main;
    if (b->n_solves > 0) HIP_TRY(hipStreamWaitEvent(st, b->ev[1], 0));        // end of the latest solve, whatever stream it ran on
    hipLaunchKernelGGL(k_positions_to_f32, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, n, b->d_positions, b->d_positions_f32);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_positions_f32, b->d_positions_f32, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(lfr::stream_wait(st));
    *positions = b->h_positions_f32;
    return LFR_OK;
}

int lfr_batch_download(lfr_batch *b, double *positions) {
    if (!b || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    const double *view = nullptr;
    int rc = lfr_batch_positions_view(b, &view);
    if (rc != LFR_OK) return rc;
    if (b->shard_world == 1) {             // whole problem: every node the batch does not solve is 0 in the view as well
        const size_t n = 2 * (size_t)b->n_graph_nodes;
        const int T = n >= ((size_t)1 << 20) ? 4 : 1;
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back([=] { memcpy(positions + n * t / T, view + n * t / T, sizeof(double) * (n * (t + 1) / T - n * t / T)); });
        memcpy(positions, view, sizeof(double) * (n / T));
        for (auto &x : th) x.join();
        return LFR_OK;
    }
    if ((rc = ensure_mirrors(b)) != LFR_OK) return rc;
    for (size_t i = 0; i < b->node_ids.size(); ++i) {      // a shard writes its own nodes only
        const size_t n = b->node_ids[i];
        positions[2 * n] = view[2 * n]; positions[2 * n + 1] = view[2 * n + 1];
    }
    return LFR_OK;
}

int64_t lfr_batch_component_info(lfr_batch *b, int64_t *component, int32_t *iterations, int32_t *termination,
                                 double *final_cost, int32_t *n_var_nodes, int32_t *n_edges) {
    if (!b) return LFR_ERR_ARG;
    if (ensure_mirrors(b) != LFR_OK) return LFR_ERR_HIP;
    if (!b->infos_valid) {
        if (hipSetDevice(b->device) != hipSuccess) return LFR_ERR_HIP;
        b->infos.resize(b->descs.size());
        hipStream_t st = b->ctx->s_main;
        if (b->n_solves > 0 && hipStreamWaitEvent(st, b->ev[1], 0) != hipSuccess) return LFR_ERR_HIP;
        if (!b->descs.empty() &&
            (hipMemcpyAsync(b->infos.data(), b->d_infos, b->descs.size() * sizeof(CompInfoDev), hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipStreamSynchronize(st) != hipSuccess))
            return LFR_ERR_HIP;
        b->infos_valid = true;
    }
    for (size_t i = 0; i < b->descs.size(); ++i) {
        if (component) component[i] = b->desc_component[i];
        if (iterations) iterations[i] = b->infos[i].iterations;
        if (termination) termination[i] = b->infos[i].termination;
        if (final_cost) final_cost[i] = b->infos[i].final_cost;
        if (n_var_nodes) n_var_nodes[i] = b->descs[i].n_var;
        if (n_edges) n_edges[i] = (int32_t)b->descs[i].n_edges;
    }
    return (int64_t)b->descs.size();
}

// Per component (the order of lfr_batch_component_info): what the elimination-tree plan of a component above 192 rows holds - 16-row
// columns, 16x16 tiles of the factor, 16x16x16 left-looking updates per factorization, levels of the elimination tree, sweep items; zeros
// for the components of the other kernel classes.  Lets a checker count the flops and bytes a solve executed (bench.py's roofline).
int64_t lfr_batch_tree_stats(lfr_batch *b, int64_t *columns, int64_t *tiles, int64_t *updates, int64_t *levels, int64_t *items) {
    if (!b) return LFR_ERR_ARG;
    if (ensure_mirrors(b) != LFR_OK) return LFR_ERR_HIP;
    const size_t n = b->descs.size();
    for (size_t i = 0; i < n; ++i) {
        const int64_t k = (int64_t)i - b->tree_begin;
        const bool in = k >= 0 && (size_t)(5 * k + 4) < b->tree_comp_stats.size();
        const int64_t *cs = in ? &b->tree_comp_stats[5 * k] : nullptr;
        if (columns) columns[i] = in ? cs[0] : 0;
        if (tiles) tiles[i] = in ? cs[1] : 0;
        if (updates) updates[i] = in ? cs[2] : 0;
        if (levels) levels[i] = in ? cs[3] : 0;
        if (items) items[i] = in ? cs[4] : 0;
    }
    return (int64_t)n;
}

int lfr_solve_hip(const lfr_problem *p, int device, int tukey_variant, double *positions, lfr_solve_stats *stats) {
    if (!p || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    lfr_batch *b = nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    int rc = lfr_batch_create(p, device, 0, 1, tukey_variant, &b);
    if (rc != LFR_OK) return rc;
    std::unique_ptr<lfr_batch> guard(b);
    const auto tc1 = std::chrono::steady_clock::now();
    rc = lfr_batch_solve(b, b->ctx->s_main, stats);         // (no statistics asked for: nothing but the positions leaves the device)
    if (rc != LFR_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    rc = lfr_batch_download(b, positions);
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) stats->d2h_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (getenv("LFR_VERBOSE") || getenv("LFR_TIMING"))
        fprintf(stderr, "lfr: lfr_solve_hip wall: batch creation %.3f ms, solve (+ statistics) %.3f ms, download %.3f ms\n",
                std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(t0 - tc1).count(),
                std::chrono::duration<double, std::milli>(t1 - t0).count());
    return rc;
}

int lfr_solve_hip_multi(const lfr_problem *p, const int *devices, int n_devices, int tukey_variant, double *positions,
                        lfr_solve_stats *stats) {
    if (!p || !devices || n_devices < 1 || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    if (n_devices == 1) return lfr_solve_hip(p, devices[0], tukey_variant, positions, stats);
    // one host thread per device (components are independent: no exchange between shards)
    std::vector<int> rcs(n_devices, LFR_OK);
    std::vector<lfr_solve_stats> sts(n_devices);
    std::vector<std::string> errs(n_devices);
    const size_t n = (size_t)p->p.g->n_nodes();
    memset(positions, 0, sizeof(double) * 2 * n);
    // labels-only problems: the labels of the other GPUs come from the GPU that ran the graph stage (peer copies in upload_labels);
    // without such a GPU the host copies are fetched once, before the threads fork
    {
        bool on_a_device = false;
        { std::lock_guard<std::mutex> lk(p->p.label_mu); for (auto &d : p->p.devs) on_a_device = on_a_device || (d && d->track); }
        if (!p->p.host_batch && !on_a_device) { const int rc = p->p.ensure_host_labels(); if (rc != LFR_OK) return rc; }
    }
    auto work = [&](int k) {
        lfr_batch *bt = nullptr;
        rcs[k] = lfr_batch_create(p, devices[k], k, n_devices, tukey_variant, &bt);
        std::unique_ptr<lfr_batch> guard(bt);
        if (rcs[k] == LFR_OK) rcs[k] = lfr_batch_solve(bt, bt->ctx->s_main, &sts[k]);
        if (rcs[k] == LFR_OK) rcs[k] = lfr_batch_download(bt, positions);       // disjoint node sets per shard
        if (rcs[k] != LFR_OK) errs[k] = lfr_last_error();
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_devices; ++k) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    for (int k = 0; k < n_devices; ++k) if (rcs[k] != LFR_OK) { lfr::set_error("device %d: %s", devices[k], errs[k].c_str()); return rcs[k]; }
    if (stats) {
        *stats = sts[0];
        for (int k = 1; k < n_devices; ++k) {
            const lfr_solve_stats &s = sts[k];
            stats->n_components += s.n_components; stats->n_edges += s.n_edges; stats->n_nodes += s.n_nodes; stats->n_tracks += s.n_tracks;
            stats->n_converged += s.n_converged; stats->n_no_convergence += s.n_no_convergence; stats->n_failed += s.n_failed;
            stats->sum_iterations += s.sum_iterations; stats->ref_jacobian_passes_edges += s.ref_jacobian_passes_edges;
            stats->ref_cost_passes_edges += s.ref_cost_passes_edges; stats->exec_passes_edges += s.exec_passes_edges;
            stats->ref_passes_nodes += s.ref_passes_nodes; stats->sum_final_cost += s.sum_final_cost;
            stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms); stats->h2d_ms = std::max(stats->h2d_ms, s.h2d_ms);
        }
    }
    return LFR_OK;
}

// solve.cc:487-641 over several GPUs from one process, the GRAPH STAGE INCLUDED (lfr_solve_hip_multi shards the solve of
// a problem whose tracks / roots / components one GPU computed): device k builds the problem of the connected components of the match
// graph dealt to shard k (lfr_problem_build_hip_shard), assembles, solves and downloads it; when the graph cannot be dealt out (one giant
// connected component) every device holds the whole problem and takes its share of the components, as lfr_solve_hip_multi does.
int lfr_solve_graph_hip_multi(const lfr_graph *g, const int *devices, int n_devices, int64_t max_nodes_in_component, int tukey_variant,
                              double *positions, lfr_problem_stats *problem_stats, lfr_solve_stats *stats) {
    if (!g || !devices || n_devices < 1 || n_devices > 64 || !positions) { lfr::set_error("bad argument"); return LFR_ERR_ARG; }
    const size_t n = (size_t)g->g.n_nodes();
    memset(positions, 0, sizeof(double) * 2 * n);
    std::vector<int> rcs(n_devices, LFR_OK);
    std::vector<lfr_solve_stats> sts(n_devices);
    std::vector<lfr_problem_stats> pst(n_devices);
    std::vector<int> sharded(n_devices, 0);
    std::vector<std::string> errs(n_devices);
    // entries of the device list that name the same GPU take turns at the graph stage and the assembly (one context, one pair of
    // streams: the stages are written for one caller per device); their solves overlap
    static std::mutex dev_mu[64];
    auto work = [&](int k) {
        lfr_problem *pr = nullptr;
        lfr_batch *bt = nullptr;
        {
            std::lock_guard<std::mutex> turn(dev_mu[devices[k] & 63]);
            rcs[k] = n_devices == 1 ? lfr_problem_build_hip_ex(g, devices[k], max_nodes_in_component, nullptr, 0, &pr)
                                    : lfr_problem_build_hip_shard(g, devices[k], max_nodes_in_component, LFR_BUILD_FLOWS_STAY_ON_HOST, k, n_devices, &pr);
            if (rcs[k] == LFR_OK) {
                sharded[k] = pr->p.cc_sharded ? 1 : 0;
                pst[k] = pr->p.stats;
                rcs[k] = sharded[k] || n_devices == 1 ? lfr_batch_create(pr, devices[k], 0, 1, tukey_variant, &bt)
                                                      : lfr_batch_create(pr, devices[k], k, n_devices, tukey_variant, &bt);
            }
        }
        std::unique_ptr<lfr_batch> guard(bt);
        if (rcs[k] == LFR_OK) rcs[k] = lfr_batch_solve(bt, bt->ctx->s_main, &sts[k]);
        if (rcs[k] == LFR_OK) {
            // disjoint node sets per shard.  A connected-component shard is a WHOLE problem of its own (every other node reads 0 in its
            // view): only its nonzero entries may be written, the other shards' nodes live in the same array
            if (sharded[k] && n_devices > 1) {
                const double *view = nullptr;
                rcs[k] = lfr_batch_positions_view(bt, &view);
                if (rcs[k] == LFR_OK) for (size_t i = 0; i < 2 * n; ++i) if (view[i] != 0.0) positions[i] = view[i];
            } else rcs[k] = lfr_batch_download(bt, positions);
        }
        if (rcs[k] != LFR_OK) errs[k] = lfr_last_error();
        guard.reset();
        if (pr) lfr_problem_free(pr);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_devices; ++k) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    for (int k = 0; k < n_devices; ++k) if (rcs[k] != LFR_OK) { lfr::set_error("device %d: %s", devices[k], errs[k].c_str()); return rcs[k]; }
    if (problem_stats) {
        lfr_problem_stats a = pst[0];
        for (int k = 1; k < n_devices; ++k) {
            const lfr_problem_stats &b = pst[k];
            if (sharded[0] && sharded[k]) {                 // per-shard counts add up; a whole-graph problem (no shard) is the same on every device
                a.n_tracks += b.n_tracks; a.n_components += b.n_components; a.n_cut_components += b.n_cut_components;
                a.n_solved_components += b.n_solved_components; a.n_solved_tracks += b.n_solved_tracks;
                a.n_solved_edges += b.n_solved_edges; a.n_solved_nodes += b.n_solved_nodes;
                a.max_track_size = std::max(a.max_track_size, b.max_track_size);
                a.max_component_size = std::max(a.max_component_size, b.max_component_size);
                a.kruskal_rounds = std::max(a.kruskal_rounds, b.kruskal_rounds); a.tie_resorts = std::max(a.tie_resorts, b.tie_resorts);
            }
            a.tracks_ms = std::max(a.tracks_ms, b.tracks_ms); a.roots_ms = std::max(a.roots_ms, b.roots_ms);
            a.graph_cut_ms = std::max(a.graph_cut_ms, b.graph_cut_ms); a.assemble_ms = std::max(a.assemble_ms, b.assemble_ms);
        }
        // a shard keeps the node numbering of the whole graph: the nodes of the OTHER shards are isolated in it, each a track and a
        // component of its own - counted once per foreign shard in the sums above
        int n_sh = 0;
        for (int k = 0; k < n_devices; ++k) n_sh += sharded[k];
        if (sharded[0] && n_sh > 1) { a.n_tracks -= (int64_t)(n_sh - 1) * (int64_t)n; a.n_components -= (int64_t)(n_sh - 1) * (int64_t)n; }
        *problem_stats = a;
    }
    if (stats) {
        *stats = sts[0];
        for (int k = 1; k < n_devices; ++k) {
            const lfr_solve_stats &s2 = sts[k];
            stats->n_components += s2.n_components; stats->n_edges += s2.n_edges; stats->n_nodes += s2.n_nodes; stats->n_tracks += s2.n_tracks;
            stats->n_converged += s2.n_converged; stats->n_no_convergence += s2.n_no_convergence; stats->n_failed += s2.n_failed;
            stats->sum_iterations += s2.sum_iterations; stats->ref_jacobian_passes_edges += s2.ref_jacobian_passes_edges;
            stats->ref_cost_passes_edges += s2.ref_cost_passes_edges; stats->exec_passes_edges += s2.exec_passes_edges;
            stats->ref_passes_nodes += s2.ref_passes_nodes; stats->sum_final_cost += s2.sum_final_cost;
            stats->kernel_ms = std::max(stats->kernel_ms, s2.kernel_ms); stats->h2d_ms = std::max(stats->h2d_ms, s2.h2d_ms);
        }
    }
    return LFR_OK;
}

}  // extern "C"

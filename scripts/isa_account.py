"""Instruction-level account of solve_packed_kernel: compiles lfr_solve.hip with -DLFR_ISA_MARKS (comment marks at the
phase boundaries, no GPU needed), walks the ISA of the kernel in textual order and counts the instructions of every phase by kind, per
packed class.  The counts are STATIC (one pass over the unrolled code: every edge slot, every elimination step, every instantiation of
the elimination once); next to the s_memtime phase profile (-DLFR_PROFILE_PHASES) they say what a phase's cycles are spent on.
usage: python scripts/isa_account.py [kernel-substring] > profiles/r04_isa_account_packed_kernel.txt"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "local-feature-refinement_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "solve_packed_kernelILb0E"
out = "/tmp/lfr_solve_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DLFR_ISA_MARKS", "-I", os.path.join(ROOT, "include"), "-I", C,
                       "-S", "--cuda-device-only", os.path.join(C, "lfr_solve.hip"), "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]


def kind(op):
    if op.startswith("ds_add") or op.startswith("ds_pk_add") or op.startswith("ds_sub") or op.startswith("ds_max") or op.startswith("ds_min"): return "lds_atomic"
    if op.startswith("ds_swizzle") or op.startswith("ds_bpermute") or op.startswith("ds_permute"): return "lds_xbar"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_mfma"): return "mfma"
    if re.match(r"v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|cmp\w*|cvt_f64|div|ldexp|frexp|trunc|floor|rndne|fract)_?\w*f64", op) or op.endswith("_f64") or "f64" in op: return "valu_f64"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_add") or op.startswith("ds_pk_add") or "ds_add" in op: return "lds_atomic"
    if op.startswith("ds_swizzle") or op.startswith("ds_bpermute") or op.startswith("ds_permute"): return "lds_xbar"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    return "other"


NAMES = {"0": "prologue: edges -> registers", "loop_top": "iteration entry: convergence tests, LM diagonal, rows of A -> registers (all instantiations)", "1": "LM step tail (model change, transitions of A)", "5": "step setup: diagonal, build rows of A in registers",
         "gauss_jordan": "Gauss-Jordan elimination (all instantiations)", "step_reductions": "step: reductions, validity, trial point",
         "6": "zero J^T J in LDS", "sweep_setup": "sweep: slot setup", "eval": "sweep: edge evaluation (cost.cc model: interpolant, loss, corrector)",
         "assemble": "sweep: assembly (pair exchange through DPP, LDS atomics)", "eval_cost_only": "cost-only sweep: evaluation", "cost_only_loop": "cost-only sweep: loop",
         "2": "after the sweep: LDS sync", "3": "post-sweep reductions", "4": "transitions (convergence tests, line search bookkeeping)", "entry": "kernel entry / class dispatch"}
cls, mark = "entry", "entry"
acc = collections.defaultdict(lambda: collections.Counter())
dpp = collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r";\s*LFR_CLASS (\d+) (\d+) (\d+)", t)
    if m:
        cls = "<%s,%s,%s>" % m.groups(); mark = "0"; continue
    m = re.match(r";\s*LFR_MARK (\w+)", t)
    if m:
        mark = m.group(1); continue
    if not t or t[0] in ";." or t.endswith(":"): continue
    op = t.split()[0]
    acc[(cls, mark)][kind(op)] += 1
    if "dpp" in op or "row_" in t or "quad_perm" in t: dpp[(cls, mark)] += 1
kinds = ["valu_f64", "valu_other", "lane", "lds", "lds_atomic", "lds_xbar", "vmem", "scratch", "smem", "salu", "waitcnt", "branch", "other"]
print("solve_packed_kernel<false>: static instruction counts per phase (compile-time unrolled code, one pass), from the ISA of this tree\n")
order = ["0", "loop_top", "5", "gauss_jordan", "step_reductions", "1", "6", "sweep_setup", "eval", "assemble", "eval_cost_only", "cost_only_loop", "2", "3", "4"]
for c in sorted({k[0] for k in acc}):
    print("class %s" % c)
    print("  %-62s %6s | %s | dpp" % ("phase", "total", " ".join("%9s" % k[:9] for k in kinds)))
    tot = collections.Counter()
    for mk in order + sorted({k[1] for k in acc if k[0] == c} - set(order)):
        a = acc.get((c, mk))
        if not a: continue
        n = sum(a.values())
        tot.update(a)
        print("  %-62s %6d | %s | %4d" % (NAMES.get(mk, mk)[:62], n, " ".join("%9d" % a[k] for k in kinds), dpp[(c, mk)]))
    print("  %-62s %6d | %s" % ("sum", sum(tot.values()), " ".join("%9d" % tot[k] for k in kinds)))
    print()

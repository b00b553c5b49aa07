"""Post-process the rocprofv3 output of scripts/profile_round4.sh (gpurun_out/prof_r04) into the summaries under profiles/ (runs here).
Every summary carries kernel_source_sha256 - the hash bench.py compares with the tree it runs from before it reports their numbers.
usage: python scripts/pmc_summarize.py [tag, default r04] [directory, default gpurun_out/prof_<tag>]"""
import collections, csv, glob, hashlib, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
prof = os.path.join(ROOT, "profiles")


def kernel_source_sha256():      # (the same function as bench.py's)
    h = hashlib.sha256()
    for f in ("lfr_solve.hip", "lfr_device.hpp"):
        h.update(open(os.path.join(ROOT, "local-feature-refinement_amd", "csrc", f), "rb").read())
    return h.hexdigest()


SHA = kernel_source_sha256()


def short(name):
    m = re.search(r"(solve_block_kernel<[^>]*>|solve_packed_kernel(<[^>]*>)?|solve_tree_team_kernel<[^>]*>|solve_tree_kernel<[^>]*>|solve_group_kernel<[^>]*>)", name)
    return m.group(1).replace(" ", "") if m else None


def collect(prefix):
    pm = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(out, prefix + "_*/"))):
        fs = glob.glob(d + "*counter_collection.csv")
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(fs[0])):
            kn = short(row["Kernel_Name"])
            if kn:
                agg[kn][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for kn, cs in agg.items():
            for k, v in cs.items():
                big = [x for x in v if x > 0.2 * max(v)] if max(v) > 0 else v      # drop the warm-up's toy launches
                pm[kn][k] = {"dispatches": len(big), "mean_per_dispatch": sum(big) / max(1, len(big))}
    return pm


p4 = collect("pmc4")
json.dump(dict(p4, kernel_source_sha256=SHA), open(os.path.join(prof, TAG + "_pmc_solve_packed_kernel.json"), "w"), indent=1)
p5 = collect("pmc5")
blocks = {k: v for k, v in p5.items() if k.startswith("solve_block_kernel")}
summary = {"kernel_source_sha256": SHA,
           "workload": "config5 stand-in (scripts/prof_c5.py): one solve = the three LDS classes of solve_block_kernel, concurrent (+ a tiny packed launch)",
           "source": "scripts/profile_round%s.sh: separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU; "
                     "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT; SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) over python scripts/prof_c5.py; "
                     "per-dispatch means per kernel, summed over the three kernels for the per-solve figures" % TAG[-1],
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); units KB; WRITE_SIZE uncorrected",
           "kernels": blocks}
summary["hbm_bytes_per_solve"] = sum((2 * c["FETCH_SIZE"]["mean_per_dispatch"] + c["WRITE_SIZE"]["mean_per_dispatch"]) * 1024 for c in blocks.values() if "FETCH_SIZE" in c and "WRITE_SIZE" in c)
# the kernel's own output is a few MB (positions, per-component info): WRITE_SIZE is the register spills going out to scratch memory, and as much comes back
summary["hbm_write_bytes_per_solve"] = sum(c["WRITE_SIZE"]["mean_per_dispatch"] * 1024 for c in blocks.values() if "WRITE_SIZE" in c)
summary["hbm_read_bytes_per_solve"] = sum(2 * c["FETCH_SIZE"]["mean_per_dispatch"] * 1024 for c in blocks.values() if "FETCH_SIZE" in c)
for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES"):
    s = sum(cs[c]["mean_per_dispatch"] for cs in blocks.values() if c in cs)
    if s:
        summary[c] = s
if summary.get("SQ_WAVE_CYCLES"):
    summary["valu_busy"] = summary.get("SQ_ACTIVE_INST_VALU", 0) / summary["SQ_WAVE_CYCLES"]
    summary["wait_fraction"] = summary.get("SQ_WAIT_ANY", 0) / summary["SQ_WAVE_CYCLES"]
json.dump(summary, open(os.path.join(prof, TAG + "_pmc_config5.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k not in ("kernels", "source", "correction", "workload")}, indent=1))
# cap-sized sparse workload: solve_tree_kernel (one launch per solve, beside three tiny LDS-class launches and a packed one)
pS = collect("pmcS")
trees = {k: v for k, v in pS.items() if k.startswith("solve_tree_kernel") or k.startswith("solve_tree_team_kernel")}
if trees:
    ssum = {"kernel_source_sha256": SHA,
            "workload": "cap-sized sparse components (scripts/prof_sparse.py 12000 = bench.py's sparse_capsized_workload): solve_tree_team_kernel (teams of up to 8 workgroups per component; round 4: solve_tree_kernel), one launch per solve",
            "source": "scripts/profile_round%s.sh: separate rocprofv3 --pmc passes over python scripts/prof_sparse.py 12000; per-dispatch means" % TAG[-1],
            "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); units KB; WRITE_SIZE uncorrected",
            "kernels": trees}
    ssum["hbm_bytes_per_solve"] = sum((2 * c["FETCH_SIZE"]["mean_per_dispatch"] + c["WRITE_SIZE"]["mean_per_dispatch"]) * 1024 for c in trees.values() if "FETCH_SIZE" in c and "WRITE_SIZE" in c)
    for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES"):
        v = sum(cs[c]["mean_per_dispatch"] for cs in trees.values() if c in cs)
        if v:
            ssum[c] = v
    if ssum.get("SQ_WAVE_CYCLES"):
        ssum["valu_busy"] = ssum.get("SQ_ACTIVE_INST_VALU", 0) / ssum["SQ_WAVE_CYCLES"]
        ssum["wait_fraction"] = ssum.get("SQ_WAIT_ANY", 0) / ssum["SQ_WAVE_CYCLES"]
    json.dump(ssum, open(os.path.join(prof, TAG + "_pmc_sparse.json"), "w"), indent=1)
    print("sparse:", json.dumps({k: v for k, v in ssum.items() if k not in ("kernels", "source", "correction", "workload")}, indent=1))
# config 4: what bench.py reads for roofline.traffic
# (solve_packed_kernel<false> reads the batch's 80-byte records: the timed steps; <true> gathers from the graph's arrays: the FIRST solve of a batch)
pk = p4.get("solve_packed_kernel<false>") or next((v for k, v in p4.items() if k.startswith("solve_packed_kernel")), None)
pk_gather = p4.get("solve_packed_kernel<true>")
dom = None
tr = glob.glob(os.path.join(out, "trace", "*kernel_trace.csv"))
if tr:
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(tr[0])):
        if short(row["Kernel_Name"]) == "solve_packed_kernel<false>":
            by[int(row["Grid_Size_X"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if by:
        dom = {"kernel": "solve_packed_kernel<false>", "by_grid_size_x": {str(g): {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
                                                                         for g, v in sorted(by.items(), key=lambda kv: -len(kv[1]))},
               "note": "the config-4 batch is the grid with the most launches (timed steps + warm-up); the first solve of every batch runs the gathering variant <true>"}
        json.dump(dom, open(os.path.join(out, TAG + "_dominant_kernel_launches.json"), "w"), indent=1)
if pk and "FETCH_SIZE" in pk:
    old = json.load(open(os.path.join(prof, "pmc_traffic.json")))
    full = max(dom["by_grid_size_x"].items(), key=lambda kv: kv[1]["launches"]) if dom else None
    new = {"kernel": "solve_packed_kernel", "kernel_source_sha256": SHA, "edges_per_launch": old["edges_per_launch"],
           "FETCH_SIZE_KB": pk["FETCH_SIZE"]["mean_per_dispatch"], "WRITE_SIZE_KB": pk["WRITE_SIZE"]["mean_per_dispatch"],
           "hbm_bytes_per_launch": int((2 * pk["FETCH_SIZE"]["mean_per_dispatch"] + pk["WRITE_SIZE"]["mean_per_dispatch"]) * 1024),
           "correction": old["correction"],
           "valu_busy": pk["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / pk["SQ_WAVE_CYCLES"]["mean_per_dispatch"] if "SQ_WAVE_CYCLES" in pk else None,
           "sq": {k: v["mean_per_dispatch"] for k, v in pk.items() if k.startswith("SQ_")},
           "rocprof_avg_launch_us": full[1]["avg_us"] if full else None, "rocprof_launches": full[1]["launches"] if full else None,
           "gather_variant_first_solve": ({"FETCH_SIZE_KB": pk_gather["FETCH_SIZE"]["mean_per_dispatch"], "WRITE_SIZE_KB": pk_gather["WRITE_SIZE"]["mean_per_dispatch"],
                                           "note": "solve_packed_kernel<true>: 8-byte loads of isolated 72-byte flow rows (FETCH_SIZE not doubled: the correction is for 16 B/lane reads); "
                                                   "a row straddles cache lines, ~2x the bytes it needs - the price of not writing and re-reading 400 MB of records in a one-shot run"}
                                          if pk_gather and "FETCH_SIZE" in pk_gather else None),
           "source": "round %s: scripts/profile_round%s.sh (rocprofv3 --kernel-trace --stats of python bench.py --steps 20 --warmup 3 --no-cpu-baseline; separate --pmc passes of "
                     "python bench.py --steps 5 --warmup 1 --span-reps 1 --no-cpu-baseline --no-long-tracks --no-sparse), summarised by scripts/pmc_summarize.py" % (TAG[-1], TAG[-1])}
    json.dump(new, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    print("pmc_traffic.json:", {k: new[k] for k in ("hbm_bytes_per_launch", "valu_busy", "rocprof_avg_launch_us")})
# kernel stats of the bench command: the 60 kernels with the largest total duration
try:
    rows = list(csv.DictReader(open(os.path.join(out, "trace", "bench_kernel_stats.csv"))))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(os.path.join(out, TAG + "_bench_kernel_stats.csv"), "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows[:60]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r.get("Percentage", "")])
except (OSError, KeyError) as e:
    print("kernel stats:", e)
for f in (TAG + "_bench_kernel_stats.csv", TAG + "_dominant_kernel_launches.json", TAG + "_bench_line_under_rocprof.json", TAG + "_bench_line.json",
          TAG + "_phase_profile_packed_kernel.txt", TAG + "_phase_profile_tree_kernel.txt", TAG + "_tree_timeline.txt", TAG + "_tree_trace.txt", TAG + "_config5_tail.txt",
          TAG + "_gpu_tests.txt", TAG + "_pipeline_trace_config4.txt", TAG + "_pipeline_trace_config5.txt", TAG + "_pmc_packed_instruction_mix.txt",
          TAG + "_ablation_packed_kernel.txt", TAG + "_smoke.txt"):
    if os.path.exists(os.path.join(out, f)):
        shutil.copy(os.path.join(out, f), os.path.join(prof, f))

#!/bin/bash
# Runs on the GPU box (through gpurun): round-3 evidence for profiles/.
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2. separate --pmc passes of the short bench (config 4, solve_packed_kernel)
#   3. separate --pmc passes over the config-5 workload (scripts/prof_c5.py: the three LDS classes of solve_block_kernel)
#   4. the bench line of an un-profiled run
# Every step under its own timeout: a faulting run must not eat the lease.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r03; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace pass failed"
SHORT="python $R/bench.py --steps 5 --warmup 1 --span-reps 1 --no-cpu-baseline --no-long-tracks --no-sparse"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc4_$tag -o pmc -- $SHORT > $OUT/pmc4_$tag.log 2>&1 || echo "pmc pass (config 4) $tag failed"
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc5_$tag -o pmc -- python $R/scripts/prof_c5.py > $OUT/pmc5_$tag.log 2>&1 || echo "pmc pass (config 5) $tag failed"
done
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc5_mfma -o pmc -- python $R/scripts/prof_c5.py > $OUT/pmc5_mfma.log 2>&1 || echo "pmc pass (config 5) mfma failed"
python - <<PY
import csv, collections, glob, json, os
out = "$OUT"
try:
    rows = list(csv.DictReader(open(out + "/trace/bench_kernel_stats.csv")))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out + "/r03_bench_kernel_stats.csv", "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows[:60]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r.get("Percentage", "")])
    for r in rows[:14]:
        print("%-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    tr = glob.glob(out + "/trace/*kernel_trace.csv")
    if tr:
        by = collections.defaultdict(list)
        for row in csv.DictReader(open(tr[0])):
            if "solve_packed_kernel" in row["Kernel_Name"]:
                by[int(row["Grid_Size_X"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        dom = {str(g): {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3} for g, v in sorted(by.items(), key=lambda kv: -len(kv[1]))}
        json.dump({"kernel": "solve_packed_kernel", "by_grid_size_x": dom,
                   "note": "the config-4 batch is the grid with the most launches (timed steps + warm-up + one-shot spans)"}, open(out + "/r03_dominant_kernel_launches.json", "w"), indent=1)
        print(json.dumps(dom))
except Exception as e:
    print("kernel stats:", e)

def collect(prefix, match):
    pm = collections.defaultdict(dict)
    for d in glob.glob(out + "/" + prefix + "_*/"):
        fs = glob.glob(d + "*counter_collection.csv")
        if not fs: continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(fs[0])):
            name = row["Kernel_Name"]
            if match in name:
                short = name.split("(")[0].replace("void (anonymous namespace)::", "")
                agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for kn, cs in agg.items():
            for k, v in cs.items():
                big = [x for x in v if x > 0.2 * max(v)] if max(v) > 0 else v     # drop the warm-up's toy launches
                pm[kn][k] = {"n": len(big), "mean": sum(big) / max(1, len(big))}
    return pm
p4 = collect("pmc4", "solve_packed_kernel")
json.dump(p4, open(out + "/r03_pmc_solve_packed_kernel.json", "w"), indent=1)
p5 = collect("pmc5", "solve_block_kernel")
p5.update(collect("pmc5", "solve_packed_kernel"))
tot = 0.0
for kn, cs in p5.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        tot += (2 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024
summary = {"workload": "config5 stand-in (scripts/prof_c5.py): one solve = the three LDS classes of solve_block_kernel (+ a tiny packed launch)",
           "hbm_bytes_per_solve": tot,
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); units KB; WRITE_SIZE uncorrected",
           "kernels": p5}
for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"):
    s = sum(cs[c]["mean"] for cs in p5.values() if c in cs and "block" in "".join(k for k in p5 if p5[k] is cs))
    if s: summary[c] = s
json.dump(summary, open(out + "/r03_pmc_config5.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
PY
grep -o '{"metric.*' $OUT/bench_under_rocprof.log | tail -1 > $OUT/r03_bench_line_under_rocprof.json
timeout -k 5 600 python $R/bench.py --steps 20 --warmup 3 > $OUT/r03_bench_line.json 2> $OUT/r03_bench.err || echo "bench failed"
ls -la $OUT | head -40

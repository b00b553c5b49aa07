#!/bin/bash
# Runs on the GPU box (through gpurun): round-2 evidence for profiles/.
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ activity) of the same command (short variant)
# Every step under its own timeout: a faulting run must not eat the lease.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r02; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace pass failed"
SHORT="python $R/bench.py --steps 5 --warmup 1 --span-reps 1 --no-cpu-baseline --no-long-tracks"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- $SHORT > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
done
python - <<PY
import csv, collections, glob, json, os
out = "$OUT"
rows = list(csv.DictReader(open(out + "/trace/bench_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(out + "/r02_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows[:60]:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r.get("Percentage", "")])
for r in rows[:12]:
    print("%-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
# the stats file averages every launch of a kernel name (warm-up toys, the config-5 leg's small packed part): the dominant kernel's
# full-size launches on their own, from the kernel trace
tr = glob.glob(out + "/trace/*kernel_trace.csv")
if tr:
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(tr[0])):
        if "solve_packed_kernel" in row["Kernel_Name"]:
            by[int(row["Grid_Size_X"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    dom = {str(g): {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3} for g, v in sorted(by.items(), key=lambda kv: -len(kv[1]))}
    json.dump({"kernel": "solve_packed_kernel", "by_grid_size_x": dom,
               "note": "the config-4 batch is the grid with the most launches (timed steps + warm-up + one-shot spans)"}, open(out + "/r02_dominant_kernel_launches.json", "w"), indent=1)
    print(json.dumps(dom))
pm = {}
for d in glob.glob(out + "/pmc_*/"):
    fs = glob.glob(d + "*counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(fs[0])):
        if "solve_packed_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        big = [x for x in v if x > 0.2 * max(v)]          # drop the warm-up's toy launches
        pm[k] = {"n": len(big), "mean": sum(big) / max(1, len(big))}
json.dump(pm, open(out + "/r02_pmc_solve_packed_kernel.json", "w"), indent=1)
print(json.dumps(pm, indent=1))
PY
grep -o '{"metric.*' $OUT/bench_under_rocprof.log | tail -1 > $OUT/r02_bench_line_under_rocprof.json

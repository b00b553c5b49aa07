#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel time of the whole device pipeline (graph stage, assembly, solve).
# usage: scripts/prof_pipeline.sh <tag>     (every step under its own timeout: a faulting run must not eat the lease)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pipe_$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 5 --warmup 1 --span-reps 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || { echo "profiled bench failed"; tail -5 $OUT/bench_under_rocprof.log; exit 1; }
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace/bench_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:32]:
    print("%-80s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for HBM traffic.
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
head -8 $OUT/trace/bench_kernel_stats.csv
python - <<PY
import csv, collections
for tag in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open("$OUT/%s/pmc_counter_collection.csv" % tag)):
        agg[(row["Kernel_Name"][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if "solve" in k[0]: print(tag, k, "n=%d mean=%.1f" % (len(v), sum(v) / len(v)))
PY

#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>"   (run on the GPU box through gpurun; one --pmc pass per call)
set -e
cd /tmp; export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1 || true
python - <<PY
import csv, collections, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    if "solve" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f per-dispatch" % (c, v / cnt[(k, c)]))
PY

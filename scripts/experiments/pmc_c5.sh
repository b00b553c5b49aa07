#!/bin/bash
# usage (GPU box): scripts/pmc_c5.sh "<counters>"  - one --pmc pass over the config-5 workload (workgroup kernel)
cd /tmp; export TMPDIR=/tmp
tag=$(echo $1 | tr ' ' '_' | cut -c1-40)
timeout -k 5 150 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc5_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/prof_c5.py > $GRAFT_REPO_ROOT/gpurun_out/pmc5_$tag.log 2>&1 || echo "pass failed"
python - <<PY
import csv, collections, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc5_$tag/*counter_collection.csv")
agg = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if "solve_block_kernel" in row["Kernel_Name"]:
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("   %-28s %16.0f per dispatch (n=%d)" % (c, sum(v) / len(v), len(v)))
PY

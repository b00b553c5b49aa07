#!/bin/bash
# kernel + copy timeline of ONE `solve` CLI run over the config-4 file: what the one-shot "Total time" is made of on the device
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_cli; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python - <<PY
import sys, os
sys.path.insert(0, "$R/local-feature-refinement_amd")
from lfr_amd import capi, synthetic
capi.write_matching_file("/tmp/config4.pb", synthetic.config4())
PY
cat > /tmp/run_cli.py <<PY
import sys
sys.path.insert(0, "$R/local-feature-refinement_amd")
from lfr_amd.solve_cli import main
rc = main(["--matches_file", "/tmp/config4.pb", "--output_file", "/tmp/sol.pb"])
sys.exit(rc)
PY
LFR_TIMING=1 python /tmp/run_cli.py > $OUT/plain.log 2>&1; grep "Total\|one-shot" $OUT/plain.log
LFR_TIMING=1 timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python /tmp/run_cli.py > $OUT/run.log 2>&1
grep "Total\|one-shot" $OUT/run.log
python - <<PY
import csv, glob, re
rows = []
for f in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("$OUT/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
last_solve = max(i for i, r in enumerate(rows) if "solve_packed" in r[2])
t_end = rows[min(len(rows) - 1, last_solve + 3)][1]
grp = [r for r in rows if r[0] > t_end - 25e6 and r[0] <= t_end]
t0 = grp[0][0]; prev = t0
out = open("$OUT/last_25ms.txt", "w")
for s, e, n in grp:
    n = re.sub(r"\(anonymous namespace\)::|lfr::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", n)[:90]
    out.write("%9.1f us  +%7.1f us gap  %8.1f us  %s\n" % ((s - t0) / 1e3, max(0, s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
PY
awk '$3+0 > 100 || $6+0 > 100' $OUT/last_25ms.txt | head -80

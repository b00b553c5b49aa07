#!/bin/bash
# usage: scripts/pipeline_trace.sh c4|c5  -> kernels of the LAST pipeline repetition, in launch order, with durations and gaps
which=${1:-c4}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$which; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $R/scripts/pipeline_trace.py $which 4 > $OUT/run.log 2>&1
grep "^rep" $OUT/run.log
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("$OUT/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
# the last repetition: everything that starts inside the host-timed span of the last "rep" line before the final read-back ends (config 5's
# host cut is a 5 ms hole in the launches: a gap threshold split that repetition in two)
import re as _re
tot = [float(m.group(1)) for m in (_re.search(r"total ([0-9.]+) ms", l) for l in open("$OUT/run.log") if l.startswith("rep")) if m]
last_end = max(r[1] for r in rows)
grp = [r for r in rows if r[0] >= last_end - (tot[-1] * 1e6 + 30e3)]
t0 = grp[0][0]
import re
busy = 0
out = open("$OUT/last_rep.txt", "w")
prev_end = t0
for s, e, n in grp:
    n = re.sub(r"\(anonymous namespace\)::|lfr::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", n)[:90]
    line = "%9.1f us  +%7.1f us gap  %8.1f us  %s" % ((s - t0) / 1e3, max(0, s - prev_end) / 1e3, (e - s) / 1e3, n)
    out.write(line + "\n")
    busy += e - s
    prev_end = max(prev_end, e)
print("last repetition: %d launches/copies, span %.3f ms, sum of durations %.3f ms" % (len(grp), (grp[-1][1] - t0) / 1e6, busy / 1e6))
PY
head -150 $OUT/last_rep.txt

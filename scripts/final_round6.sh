#!/bin/bash
# Runs on the GPU box (through gpurun) after scripts/profile_round6.sh + `python scripts/pmc_summarize.py r06`: the un-profiled bench line (its
# roofline.traffic now comes from PMC summaries of THIS tree), smoke(), the whole GPU suite.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r06; mkdir -p $OUT; cd $R
timeout -k 5 900 python bench.py --steps 20 --warmup 3 > $OUT/r06_bench_line.json 2> $OUT/r06_bench.err || echo "bench failed"
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r06_smoke.txt 2>&1 || echo "smoke failed"
timeout -k 5 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/r06_gpu_tests.txt
tail -2 $OUT/r06_smoke.txt; cat $OUT/r06_gpu_tests.txt; python - <<PY
import json
d = json.loads([l for l in open("$OUT/r06_bench_line.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "traffic", d["roofline"].get("traffic"), "solver", d["solver_span"]["ms"], "total_res", d["total_span_resident_graph"]["ms"])
for k in ("long_tracks_workload", "sparse_capsized_workload"):
    w = d.get(k, {}); print(k, w.get("ms_per_step"), (w.get("roofline") or {}).get("frac"), (w.get("roofline") or {}).get("traffic"))
PY

#!/bin/bash
# usage: scripts/r3_bench.sh TAG [pytest -k expression]
tag=$1; mkdir -p gpurun_out
if [ -n "$2" ]; then timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "$2" > gpurun_out/${tag}_tests.log 2>&1; tail -6 gpurun_out/${tag}_tests.log; fi
timeout -k 5 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -3 gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().split("\n")[-1])
print("value %.3e edges/s  ms/step %.4f  roofline frac %.3f  solver_span %.2f ms  total_span %.2f  total_resident %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["solver_span"]["ms"], d["total_span"]["ms"], d["total_span_resident_graph"]["ms"]))
cb = d.get("cpu_baseline", {})
print("cpu_baseline:", {k: cb.get(k) for k in ("value", "cores")}, cb.get("at_reference_default_8_threads"), cb.get("at_all_cores"))
lt = d.get("long_tracks_workload", {})
print("long tracks: ms/step %.3f  total_resident %.2f  graph stage %s  roofline %s" % (lt.get("ms_per_step", 0), lt.get("total_span_resident_graph_ms", 0), lt.get("graph_stage"), {k: lt["roofline"][k] for k in ("achieved", "frac")} if "roofline" in lt else None))
sp = d.get("sparse_capsized_workload", {})
print("sparse:", {k: sp.get(k) for k in ("ms_per_step", "edges_per_s", "mean_iterations_large", "max_iterations_large", "batch_creation_ms", "setup_s")})
print(sp.get("workload"))
PY

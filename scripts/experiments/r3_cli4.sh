#!/bin/bash
mkdir -p gpurun_out
python - > gpurun_out/r3_cli4.log 2>&1 <<PY
import sys, os, time
sys.path.insert(0, "local-feature-refinement_amd")
from lfr_amd import capi, synthetic
capi.write_matching_file("/tmp/config4.pb", synthetic.config4())
os.environ["LFR_VERBOSE"] = "1"
for th in (32, 64):
    os.environ["LFR_HOST_THREADS"] = str(th)
    for rep in range(2):
        t = time.perf_counter(); g = capi.Graph.from_matches_file("/tmp/config4.pb"); print("threads %d: parse %.1f ms" % (th, (time.perf_counter() - t) * 1e3), flush=True); del g
PY
grep "scanner\|parse" gpurun_out/r3_cli4.log | cut -c1-300
for i in 1 2 3; do
  LFR_VERBOSE=1 multi-view-refinement/build/solve --matches_file /tmp/config4.pb --output_file /tmp/sol.pb 2>&1 | grep "scanner\|wall inside" | cut -c1-330
done
LFR_TIMING=1 python scripts/cli_e2e.py 2>&1 | grep "CLI wall\|Total time\|back to back"

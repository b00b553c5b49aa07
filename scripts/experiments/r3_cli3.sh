#!/bin/bash
mkdir -p gpurun_out
python - > gpurun_out/r3_cli3.log 2>&1 <<PY
import sys, os
sys.path.insert(0, "local-feature-refinement_amd")
from lfr_amd import capi, synthetic
capi.write_matching_file("/tmp/config4.pb", synthetic.config4())
print("written", os.path.getsize("/tmp/config4.pb"))
PY
for i in 1 2 3; do
  s=$(date +%s.%N)
  LFR_VERBOSE=1 multi-view-refinement/build/solve --matches_file /tmp/config4.pb --output_file /tmp/sol.pb >> gpurun_out/r3_cli3.log 2>&1
  e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc) s rc=$?" >> gpurun_out/r3_cli3.log
done
grep -v "^#\|^max \|time:" gpurun_out/r3_cli3.log | cut -c1-420

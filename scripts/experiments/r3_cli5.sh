#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 300 python -m pytest tests/test_gpu_pipeline.py -q -x -k "ingest_straight or resident_graph" 2>&1 | tail -2
python - > gpurun_out/r3_cli5.log 2>&1 <<PY
import sys, os
sys.path.insert(0, "local-feature-refinement_amd")
from lfr_amd import capi, synthetic
capi.write_matching_file("/tmp/config4.pb", synthetic.config4())
PY
for pin in 0 1; do
  echo "== LFR_PIN_FLOWS=$pin"
  for i in 1 2 3; do
    LFR_PIN_FLOWS=$pin LFR_VERBOSE=1 multi-view-refinement/build/solve --matches_file /tmp/config4.pb --output_file /tmp/sol$pin.pb 2>&1 | grep "scanner\|wall inside\|Total" | cut -c1-330
  done
  LFR_PIN_FLOWS=$pin LFR_TIMING=1 python scripts/cli_e2e.py 2>&1 | grep "CLI wall\|Total time\|back to back"
done
cmp /tmp/sol0.pb /tmp/sol1.pb && echo "outputs identical"

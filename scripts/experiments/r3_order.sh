#!/bin/bash
# share of the CUs the first workgroup class gets (LFR_WG_SHARE_BIAS) on config 5
mkdir -p gpurun_out
for bias in 1.0 0.9 0.8 1.1; do
  echo "== share bias $bias"
  LFR_WG_SHARE_BIAS=$bias timeout -k 5 200 python scripts/prof_c5.py 2>&1 | grep "config5:\|per launch" | tail -3
done
echo "== round-2 order, no cap"; LFR_WG_ORDER=8,6,7,5 timeout -k 5 200 python scripts/prof_c5.py 2>&1 | grep "config5:\|per launch" | tail -2

#!/bin/bash
# round-3 iteration loop on the GPU box: config-5 timing of the default build and of named variants, phase profile, parity subset
# usage: scripts/r3_check.sh TAG [variant ...]     (variants: lfr_amd/_variants/NAME.so built here with scripts/mkvariant.sh)
tag=$1; shift
mkdir -p gpurun_out
V=$PWD/local-feature-refinement_amd/lfr_amd/_variants
echo "== default build" | tee gpurun_out/${tag}_c5.log
timeout -k 5 200 python scripts/prof_c5.py >> gpurun_out/${tag}_c5.log 2>&1
grep -v "^rows\|histogram" gpurun_out/${tag}_c5.log | tail -5
for v in "$@"; do
  echo "== variant $v" | tee -a gpurun_out/${tag}_c5.log
  LFR_LIB_OVERRIDE=$V/$v.so timeout -k 5 200 python scripts/prof_c5.py > gpurun_out/${tag}_c5_$v.log 2>&1
  grep "config5:\|per launch\|lfr-prof class [5678]\|lfr-fprof" gpurun_out/${tag}_c5_$v.log | tail -12 | cut -c1-330
done
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "long_tracks or config5 or huge or kernel_class or standin or fuzz or units" > gpurun_out/${tag}_tests.log 2>&1
tail -5 gpurun_out/${tag}_tests.log

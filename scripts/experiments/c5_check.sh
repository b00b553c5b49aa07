#!/bin/bash
# config-5 iteration loop on the GPU box: timing, phase profile (prof variant), parity subset
mkdir -p gpurun_out
timeout -k 5 200 python scripts/prof_c5.py > gpurun_out/c5_a.log 2>&1
LFR_LIB_OVERRIDE=$PWD/local-feature-refinement_amd/lfr_amd/_variants/prof.so timeout -k 5 200 python scripts/prof_c5.py > gpurun_out/c5_b.log 2>&1
grep -v "^rows" gpurun_out/c5_a.log | tail -4
grep "lfr-prof class [5678]" gpurun_out/c5_b.log | tail -4 | cut -c1-170
timeout -k 5 500 python -m pytest tests -m gpu -x -q -k "long_tracks or config5 or huge or kernel_class or standin or fuzz" > gpurun_out/split_tests.log 2>&1
tail -4 gpurun_out/split_tests.log

#!/bin/bash
# cooperative union-find rounds: label tests (bit-identical with the host stage), then config-5 lap times old/new
mkdir -p gpurun_out
timeout -k 5 400 python -m pytest tests/test_gpu_pipeline.py -q -x -k "giant or round_based or real_shaped or labels_are" > gpurun_out/r3_rounds_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_rounds_tests.log
for mode in coop1 coop2 coop4 hostloop; do
    echo "== $mode"
    unset LFR_ROUNDS_HOST_LOOP LFR_ROUNDS_BLOCKS_PER_CU
    case $mode in hostloop) export LFR_ROUNDS_HOST_LOOP=1;; coop1) export LFR_ROUNDS_BLOCKS_PER_CU=1;; coop2) export LFR_ROUNDS_BLOCKS_PER_CU=2;; coop4) export LFR_ROUNDS_BLOCKS_PER_CU=4;; esac
    LFR_VERBOSE=3 timeout -k 5 200 python scripts/pipeline_trace.py c5 3 > gpurun_out/r3_c5_laps_${mode}.log 2>&1
    tail -13 gpurun_out/r3_c5_laps_${mode}.log | grep -v "MARK\|memsets\|slabs\|device graph ready"
done

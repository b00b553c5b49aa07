#!/bin/bash
# block-envelope kernel after a change: the sparse + huge-component tests, then timings (1000 and 12000 tracks)
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -q -x -k "sparse or sky or huge or envelope or capsized" > gpurun_out/r3_sky3_tests.log 2>&1; tail -3 gpurun_out/r3_sky3_tests.log
timeout -k 5 300 python scripts/sky_check.py 1000 > gpurun_out/r3_sky3_1k.log 2>&1; tail -4 gpurun_out/r3_sky3_1k.log
timeout -k 5 300 python scripts/sky_check.py 12000 > gpurun_out/r3_sky3_12k.log 2>&1; tail -4 gpurun_out/r3_sky3_12k.log

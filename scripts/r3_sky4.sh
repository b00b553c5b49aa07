#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -q -x -k "sparse or sky or huge or envelope or capsized" > gpurun_out/r3_sky4_tests.log 2>&1; tail -3 gpurun_out/r3_sky4_tests.log
timeout -k 5 300 python scripts/sky_check.py 12000 2>&1 | grep "sparse:\|components above" | tail -2

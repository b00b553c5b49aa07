#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 200 python scripts/prof_c5.py 2>&1 | grep "config5:\|per launch" | tail -2
timeout -k 5 300 python scripts/sky_check.py 12000 2>&1 | grep "sparse:\|components above" | tail -2
timeout -k 5 1400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_damp_tests.log 2>&1; tail -3 gpurun_out/r3_damp_tests.log

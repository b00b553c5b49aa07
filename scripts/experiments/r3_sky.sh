#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 300 python scripts/sky_check.py 3000 > gpurun_out/r3_sky.log 2>&1; tail -8 gpurun_out/r3_sky.log
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "sparse or huge or global or kernel_class or very_long" > gpurun_out/r3_sky_tests.log 2>&1; tail -15 gpurun_out/r3_sky_tests.log

#!/bin/bash
mkdir -p gpurun_out
V=$PWD/local-feature-refinement_amd/lfr_amd/_variants
echo "== round-2 dense HBM kernel (base_r2.so), 1000 tracks"
LFR_LIB_OVERRIDE=$V/base_r2.so timeout -k 5 700 python scripts/sky_check.py 1000 > gpurun_out/r3_sky_dense.log 2>&1; tail -5 gpurun_out/r3_sky_dense.log
echo "== block-envelope kernel, 1000 tracks"
timeout -k 5 300 python scripts/sky_check.py 1000 > gpurun_out/r3_sky_1k.log 2>&1; tail -5 gpurun_out/r3_sky_1k.log

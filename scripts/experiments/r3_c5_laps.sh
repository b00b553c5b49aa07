#!/bin/bash
mkdir -p gpurun_out
LFR_VERBOSE=2 python scripts/pipeline_trace.py c5 3 > gpurun_out/r3_c5_laps.log 2>&1
tail -80 gpurun_out/r3_c5_laps.log

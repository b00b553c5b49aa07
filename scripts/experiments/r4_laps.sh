#!/bin/bash
# config-5 lap trace of the graph stage (LFR_VERBOSE=2) + launch-ordered trace of the last config-4 pipeline repetition
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out


bash scripts/pipeline_trace.sh c5 > gpurun_out/r4_c5_trace.txt 2>&1
bash scripts/pipeline_trace.sh c4 > gpurun_out/r4_c4_trace.txt 2>&1
head -5 gpurun_out/r4_c4_trace.txt

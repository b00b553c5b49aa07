#!/bin/bash
# end-of-round evidence: the whole -m gpu suite, the profile passes, the CLI end to end
mkdir -p gpurun_out
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests.log 2>&1; tail -4 gpurun_out/r3_final_tests.log
scripts/profile_round3.sh > gpurun_out/profile_round3.log 2>&1; tail -3 gpurun_out/profile_round3.log
LFR_TIMING=1 python scripts/cli_e2e.py > gpurun_out/r3_cli_e2e.txt 2>&1; grep "CLI wall\|Total time\|back to back" gpurun_out/r3_cli_e2e.txt

#!/bin/bash
mkdir -p gpurun_out
LFR_TIMING=1 python scripts/cli_e2e.py > gpurun_out/r3_cli_e2e.txt 2>&1
grep -h "Total time\|Solver time\|one-shot wall\|CLI wall\|back to back" gpurun_out/r3_cli_e2e.txt

#!/bin/bash
# one-shot CLI: new ingest-to-device test, then the end-to-end timing with and without the early flow upload
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py -q -x -k "ingest_straight or resident_graph" > gpurun_out/r3_cli_tests.log 2>&1; tail -3 gpurun_out/r3_cli_tests.log
LFR_TIMING=1 python scripts/cli_e2e.py > gpurun_out/r3_cli_e2e.txt 2>&1
LFR_TIMING=1 LFR_INGEST_TO_DEVICE=0 python scripts/cli_e2e.py > gpurun_out/r3_cli_e2e_late_upload.txt 2>&1
grep -h "Total time\|Solver time\|one-shot wall\|CLI wall" gpurun_out/r3_cli_e2e.txt gpurun_out/r3_cli_e2e_late_upload.txt

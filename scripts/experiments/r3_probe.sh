#!/bin/bash
mkdir -p gpurun_out
timeout 300 scripts/probes/diag16_probe.bin > gpurun_out/r3_diag16_probe.txt 2>&1
cat gpurun_out/r3_diag16_probe.txt

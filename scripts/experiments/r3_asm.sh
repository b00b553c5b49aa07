#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -q -x -k "late_workgroup or device_assembly or sharded or every_kernel_class or standins or golden or fuzz" > gpurun_out/r3_asm_tests.log 2>&1; tail -3 gpurun_out/r3_asm_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-long-tracks --no-sparse 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.4f solver_span %.3f total_res %.3f total %.3f' % (d['value'], d['ms_per_step'], d['solver_span']['ms'], d['total_span_resident_graph']['ms'], d['total_span']['ms']))"

#!/bin/bash
# Round-4 evidence that does not depend on the solve kernels' sources (the PMC passes of scripts/profile_round4.sh stay valid while
# lfr_solve.hip / lfr_device.hpp are unchanged - bench.py checks their hash): kernel-trace statistics of the bench command, the bench line
# of an un-profiled run, the launch-ordered traces of one config-4 and one config-5 pipeline run, the GPU tests, the CLI end to end.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace pass failed"
grep -o '{"metric.*' $OUT/bench_under_rocprof.log | tail -1 > $OUT/r04_bench_line_under_rocprof.json
cd $R
bash scripts/pipeline_trace.sh c4 > $OUT/r04_pipeline_trace_config4.txt 2>&1
bash scripts/pipeline_trace.sh c5 > $OUT/r04_pipeline_trace_config5.txt 2>&1
timeout -k 5 900 python bench.py --steps 20 --warmup 3 > $OUT/r04_bench_line.json 2> $OUT/r04_bench.err || echo "bench failed"
timeout -k 5 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/r04_gpu_tests.txt
timeout -k 5 300 python scripts/cli_e2e.py > $OUT/r04_cli_e2e.txt 2>&1 || echo "cli e2e failed"
tail -3 $OUT/r04_gpu_tests.txt; tail -5 $OUT/r04_cli_e2e.txt

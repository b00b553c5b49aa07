cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_units.py -x -q -m gpu -k "not bench" 2>&1 | tail -4
bash scripts/pipeline_trace.sh c4 > /dev/null 2>&1; grep "^rep" gpurun_out/trace_c4/run.log

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "giant or real_shaped or fuzz or equal_sim or labels or empty" 2>&1 | tail -4
bash scripts/pipeline_trace.sh c4 > gpurun_out/r4_c4_trace.txt 2>&1; grep "^rep" gpurun_out/r4_c4_trace.txt

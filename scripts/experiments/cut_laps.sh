cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "giant or real_shaped" 2>&1 | tail -2
LFR_VERBOSE=2 timeout 300 python scripts/pipeline_trace.py c5 6 2>&1 | grep "recursive bisection\|meta edges of\|re-labelled\|^rep" | tail -8

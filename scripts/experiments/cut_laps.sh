cd $GRAFT_REPO_ROOT
LFR_VERBOSE=2 timeout 300 python scripts/pipeline_trace.py c5 6 2>&1 | grep "recursive bisection\|meta edges of\|^rep" | tail -12

cd $GRAFT_REPO_ROOT
for v in 16384 6000 2500 1000 300; do
  echo "== spawn_min $v"
  LFR_CUT_SPAWN_MIN_X=$v LFR_VERBOSE=2 timeout 300 python scripts/pipeline_trace.py c5 4 2>&1 | grep "recursive bisection\|meta edges of\|^rep" | tail -6
done

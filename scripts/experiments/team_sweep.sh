#!/bin/bash
# LFR_TREE_TEAM thresholds on the cap-sized sparse workload (prof_sparse.py 12000): kernel ms of three solves per setting
for s in "700,1500,2000" "700,1300,2000" "700,1100,2000" "600,1300,1900" "800,1500,2000" "700,1500,1800" "500,1200,2000" "700,1300,1800" "900,1600,2100" "700,1000,1600"; do
  echo "== LFR_TREE_TEAM=$s"
  LFR_TREE_TEAM=$s LFR_VERBOSE=0 timeout 120 python scripts/prof_sparse.py 12000 2>&1 | grep "^sparse:" | awk '{print $7}' | tr '\n' ' '
  echo
done

#!/bin/bash
# A/B of the workgroup kernel on the GPU box: lfr_amd/_variants/base.so (built from an older lfr_solve.hip) against the tree's library
R=$GRAFT_REPO_ROOT; cd $R; V=local-feature-refinement_amd/lfr_amd/_variants
for v in base cur base cur; do
  if [ $v = base ]; then export LFR_LIB_OVERRIDE=$V/base.so; else unset LFR_LIB_OVERRIDE; fi
  echo "== $v"; timeout 200 python scripts/prof_c5.py 2>&1 | grep -E "config5:|per launch" | tail -3
done

#!/bin/bash
# usage: scripts/ab2.sh "<flags A>" "<flags B>" : same-box A/B/A/B of the short bench (merged launch only)
run() { LFR_HIPCC_FLAGS="$1" python -c "
import sys; sys.path.insert(0,'local-feature-refinement_amd')
from lfr_amd import build; build.build(force=True)" >/dev/null 2>&1
for i in 1 2; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('flags[$1] ms/step %.4f launch_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac']))"; done; }
run "$1"; run "$2"; run "$1"; run "$2"

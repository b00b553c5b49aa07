#!/bin/bash
# usage (on the GPU box): scripts/abv.sh A B [C...]  -> alternates the short bench over prebuilt variants (scripts/mkvariant.sh)
R=$(cd $(dirname $0)/.. && pwd)
for round in 1 2; do for v in "$@"; do
so=${v%+serial}; ser=0; [ "$so" != "$v" ] && ser=1
LFR_SERIAL_CLASSES=$ser LFR_LIB_OVERRIDE=$R/local-feature-refinement_amd/lfr_amd/_variants/$so.so python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-14s ms/step %.4f launch_ms %.4f frac %.3f' % ('$v', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac']))"
done; done

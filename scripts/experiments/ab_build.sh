#!/bin/bash
# usage: scripts/ab_build.sh "<extra hipcc flags>"  -> rebuilds the library with the flags and runs the short bench
LFR_HIPCC_FLAGS="$1" python -c "
import sys; sys.path.insert(0,'local-feature-refinement_amd')
from lfr_amd import build; build.build(force=True)"
echo "### flags: $1"; scripts/bench_short.sh 2>&1 | head -8

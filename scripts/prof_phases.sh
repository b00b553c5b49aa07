#!/bin/bash
LFR_HIPCC_FLAGS="-DLFR_PROFILE_PHASES $1" python -c "
import sys; sys.path.insert(0,'local-feature-refinement_amd')
from lfr_amd import build; build.build(force=True)"
if [ "$2" != "" ]; then LFR_SERIAL_CLASSES=1 python scripts/gpu_check.py $2 2>&1 | grep -E "lfr-prof|$2|config" | tail -4
else LFR_SERIAL_CLASSES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep lfr-prof; fi

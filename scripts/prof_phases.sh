#!/bin/bash
LFR_HIPCC_FLAGS="-DLFR_PROFILE_PHASES $1" python -c "
import sys; sys.path.insert(0,'local-feature-refinement_amd')
from lfr_amd import build; build.build(force=True)"
if [ "$2" == "long" ]; then LFR_SERIAL_CLASSES=1 python scripts/gpu_check.py long 2>&1 | grep -E "lfr-prof|long" | tail -8
else LFR_SERIAL_CLASSES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep lfr-prof; fi

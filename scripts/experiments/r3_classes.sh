#!/bin/bash
V=$PWD/local-feature-refinement_amd/lfr_amd/_variants
echo "== default"; timeout -k 5 200 python scripts/prof_c5.py 2>&1 | grep "config5:\|per launch" | tail -2
for v in noM m112 s104m150 s72m130; do echo "== $v"; LFR_LIB_OVERRIDE=$V/$v.so timeout -k 5 200 python scripts/prof_c5.py 2>&1 | grep "config5:\|per launch" | tail -2; done

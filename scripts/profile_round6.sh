#!/bin/bash
# Runs on the GPU box (through gpurun): round-6 evidence for profiles/ (summarised HERE afterwards by `python scripts/pmc_summarize.py r06`).
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2. separate --pmc passes (never combined with other trace domains) over config 4 (short bench: solve_packed_kernel), config 5
#      (scripts/prof_c5.py: solve_block_kernel), the cap-sized sparse workload (scripts/prof_sparse.py: solve_tree_team_kernel)
#   3. phase profile (s_memtime) of the tree kernel, its per-component timeline and the event trace of its critical component, from the
#      diagnostic builds under lfr_amd/_variants/ (tprof.so, wgtime.so, trace.so); the tail of config 5's launch (wgtime.so)
#   4. launch-ordered traces of both one-shot pipelines
# Every step under its own timeout: a faulting run must not eat the lease.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r06; mkdir -p $OUT
V=$R/local-feature-refinement_amd/lfr_amd/_variants
cd /tmp; export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace pass failed"
SHORT="python $R/bench.py --steps 5 --warmup 1 --span-reps 1 --no-cpu-baseline --no-long-tracks --no-sparse"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc4_$tag -o pmc -- $SHORT > $OUT/pmc4_$tag.log 2>&1 || echo "pmc pass (config 4) $tag failed"
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc5_$tag -o pmc -- python $R/scripts/prof_c5.py > $OUT/pmc5_$tag.log 2>&1 || echo "pmc pass (config 5) $tag failed"
    timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcS_$tag -o pmc -- python $R/scripts/prof_sparse.py 12000 > $OUT/pmcS_$tag.log 2>&1 || echo "pmc pass (sparse) $tag failed"
done
for w in 5 S; do
    prog=$([ $w = 5 ] && echo "$R/scripts/prof_c5.py" || echo "$R/scripts/prof_sparse.py 12000")
    timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc${w}_mfma -o pmc -- python $prog > $OUT/pmc${w}_mfma.log 2>&1 || echo "pmc pass mfma ($w) failed"
done
cd $R
[ -f $V/tprof.so ] && LFR_LIB_OVERRIDE=$V/tprof.so timeout -k 5 200 python scripts/prof_sparse.py 12000 2>&1 | grep -E "lfr-prof class 8|sparse:|components above" > $OUT/r06_phase_profile_tree_kernel.txt
[ -f $V/wgtime.so ] && LFR_LIB_OVERRIDE=$V/wgtime.so timeout -k 5 200 python scripts/sparse_timeline.py > $OUT/r06_tree_timeline.txt 2>&1
[ -f $V/trace.so ] && LFR_LIB_OVERRIDE=$V/trace.so timeout -k 5 200 python scripts/tree_trace.py > $OUT/r06_tree_trace.txt 2>&1
[ -f $V/wgtime.so ] && LFR_TIMING_LIB=$V/wgtime.so timeout -k 5 300 python scripts/c5_tail.py > $OUT/r06_config5_tail.txt 2>&1
# round 6: the packed kernel's instruction mix (SQ passes of scripts/pmc_packed.sh) and the phase-twice ablations (scripts/ab_packed.py over
# the variants under lfr_amd/_variants/, built here beforehand with scripts/mkvariant_solve.sh)
timeout -k 5 400 bash scripts/pmc_packed.sh prof_r06/packed_mix > /dev/null 2>&1; cp $OUT/packed_mix/pmc.txt $OUT/r06_pmc_packed_instruction_mix.txt 2>/dev/null
VARS=""; for v in dbl ablev ablgj ablred ablzero; do [ -f $V/$v.so ] && VARS="$VARS $v"; done
[ -n "$VARS" ] && timeout -k 5 600 python scripts/ab_packed.py --rounds 2 main $VARS > $OUT/r06_ablation_packed_kernel.txt 2>&1
bash scripts/pipeline_trace.sh c4 > $OUT/r06_pipeline_trace_config4.txt 2>&1
bash scripts/pipeline_trace.sh c5 > $OUT/r06_pipeline_trace_config5.txt 2>&1
grep -o '{"metric.*' $OUT/bench_under_rocprof.log | tail -1 > $OUT/r06_bench_line_under_rocprof.json
# (the un-profiled bench line, the GPU tests and smoke(): scripts/final_round5.sh, AFTER `python scripts/pmc_summarize.py r06` here has written
# profiles/pmc_traffic.json for this tree - bench.py reports roofline.traffic only from a summary whose kernel-source hash matches)
ls $OUT | head -80

#!/bin/bash
# Runs on the GPU box (through gpurun): round-4 evidence for profiles/ (summarised HERE afterwards by scripts/pmc_summarize.py r04).
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2. separate --pmc passes (never combined with other trace domains) over
#        config 4 (short bench: solve_packed_kernel), config 5 (scripts/prof_c5.py: solve_block_kernel), the cap-sized sparse
#        workload (scripts/prof_sparse.py: solve_tree_kernel)
#   3. phase profiles (s_memtime) of the packed kernel and of the tree kernel from -DLFR_PROFILE_PHASES builds (lfr_amd/_variants/)
#   4. the bench line of an un-profiled run
# Every step under its own timeout: a faulting run must not eat the lease.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace pass failed"
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
V=$R/local-feature-refinement_amd/lfr_amd/_variants
if [ -f $V/prof.so ]; then
    LFR_LIB_OVERRIDE=$V/prof.so LFR_SERIAL_CLASSES=1 timeout -k 5 200 python $R/bench.py --steps 3 --warmup 1 --span-reps 1 --no-cpu-baseline --no-long-tracks --no-sparse 2>&1 | grep "lfr-prof" > $OUT/r04_phase_profile_packed_kernel.txt
fi
if [ -f $V/tprof.so ]; then
    LFR_LIB_OVERRIDE=$V/tprof.so LFR_SERIAL_CLASSES=1 timeout -k 5 200 python $R/scripts/prof_sparse.py 12000 2>&1 | grep -E "lfr-prof class 8|sparse:|components above" > $OUT/r04_phase_profile_tree_kernel.txt
fi
if [ -f $V/fprof.so ]; then
    LFR_LIB_OVERRIDE=$V/fprof.so LFR_SERIAL_CLASSES=1 timeout -k 5 200 python $R/scripts/prof_sparse.py 12000 2>&1 | grep -E "lfr-fprof class 8" >> $OUT/r04_phase_profile_tree_kernel.txt
fi
grep -o '{"metric.*' $OUT/bench_under_rocprof.log | tail -1 > $OUT/r04_bench_line_under_rocprof.json
timeout -k 5 900 python $R/bench.py --steps 20 --warmup 3 > $OUT/r04_bench_line.json 2> $OUT/r04_bench.err || echo "bench failed"
ls $OUT | head -60

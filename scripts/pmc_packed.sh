#!/bin/bash
# usage (GPU box, through gpurun): scripts/pmc_packed.sh <tag> [lib.so]  -> gpurun_out/<tag>/pmc.txt
# rocprofv3 --pmc passes (SQ block, 8 counters a pass) over the config-4 solve of scripts/ab_packed.py's child, per-dispatch means of the packed kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
[ -n "$2" ] && export LFR_LIB_OVERRIDE=$R/$2
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/p$i -o pmc -- python $R/scripts/ab_packed.py --child 147000 4 /tmp/pmc_ref_$$.npy c4 > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/pmc.txt
  rm -f /tmp/pmc_ref_$$.npy
done
python - <<PY > $O/pmc.txt
import csv, collections, glob
for f in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if "solve_packed" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            big = [x for x in v if x > 0.2 * max(v)] if max(v) > 0 else v
            print("   %-28s %16.0f per dispatch (%d dispatches)" % (c, sum(big) / max(1, len(big)), len(big)))
PY
cat $O/pmc.txt

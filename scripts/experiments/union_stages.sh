cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "giant or real_shaped or fuzz or equal_sim or labels" 2>&1 | tail -3
for c in c5 c4; do
  bash scripts/pipeline_trace.sh $c > /dev/null 2>&1
  echo "== $c"; grep "^rep 3" gpurun_out/trace_$c/run.log
  python - $c <<'P'
import csv,sys
rows=[]
for r in csv.DictReader(open('gpurun_out/trace_%s/t_kernel_trace.csv'%sys.argv[1])):
    if 'k_cc_union' in r["Kernel_Name"] or 'k_uf_flatten' in r["Kernel_Name"] or 'k_meta_union_cut' in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, 'U' if 'cc_union' in r["Kernel_Name"] else 'C' if 'cut' in r["Kernel_Name"] else 'f'))
rows.sort()
print(' '.join('%s%.0f'%(k,d) for _,d,k in rows[-14:]))
P
done

cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu -k "not bench" 2>&1 | tail -8
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_sort.json 2> gpurun_out/bench_sort.err; tail -3 gpurun_out/bench_sort.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_sort.json'))
for k in ('value','ms_per_step','solver_span','total_span','total_span_resident_graph'): print(k, json.dumps(d[k])[:100])
print(d['setup_ms'])
l=d['long_tracks_workload']
for k in ('ms_per_step','total_span_resident_graph_ms','graph_stage'): print(k, l[k])
P
bash scripts/pipeline_trace.sh c5 > gpurun_out/r4_c5_trace.txt 2>&1

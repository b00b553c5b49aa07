cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "not bench" 2>&1 | tail -15
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_sort.json 2> gpurun_out/bench_sort.err; tail -3 gpurun_out/bench_sort.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_sort.json'))
for k in ('value','ms_per_step','solver_span','total_span','total_span_resident_graph'): print(k, json.dumps(d[k])[:160])
print(d['setup_ms'])
print(json.dumps(d['long_tracks_workload'])[:1500])
P

"""One-shot pipeline (graph stage -> assembly -> solve -> positions) of config 4 or config 5, a few repetitions: run it under
`rocprofv3 --kernel-trace` (scripts/pipeline_trace.sh) to see what the Total span is made of.  usage: pipeline_trace.py [c4|c5] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ma = synthetic.config4() if which == "c4" else synthetic.config5()
g = capi.Graph.from_arrays(ma)
L = capi.lib()
L.lfr_hip_warmup(0)
L.lfr_hip_reserve(0, g.n_nodes, g.n_edges // 2)
g.to_device(0)
p = b = pos = None
for r in range(reps):
    del p, b, pos                      # (the previous repetition's batch: its destructor waits for s_main - not inside the timed part)
    L.lfr_hip_synchronize(0)
    time.sleep(0.01)                   # (an idle gap in the trace: pipeline_trace.sh takes the launches behind the last one as the last repetition)
    t0 = time.perf_counter()
    p = capi.Problem(g, device_graph_stage=0)
    t1 = time.perf_counter()
    b = capi.Batch(p, 0)
    t2 = time.perf_counter()
    b.solve(None, want_stats=False)
    pos = b.positions_view_f32()
    t3 = time.perf_counter()
    print("rep %d: graph stage %.3f ms, batch %.3f ms, solve + positions %.3f ms, total %.3f ms" % (r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
    print("MARK rep %d end" % r, flush=True)

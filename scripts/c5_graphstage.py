"""config 5: what the device graph stage spends (LFR_VERBOSE=2 lap traces) - tracks (rounds), cut continuation, assembly."""
import os, sys, time
os.environ.setdefault("LFR_VERBOSE", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
g = capi.Graph.from_arrays(ma)
capi.lib().lfr_hip_reserve(0, g.n_nodes, g.n_edges // 2)
g.to_device(0)
for rep in range(3):
    capi.lib().lfr_hip_synchronize(0)
    t0 = time.perf_counter()
    p = capi.Problem(g, device_graph_stage=0)
    t1 = time.perf_counter()
    b = capi.Batch(p, 0)
    capi.lib().lfr_hip_synchronize(0)
    ta = time.perf_counter()
    b.solve(want_stats=False)
    tb = time.perf_counter()
    b.positions_view()
    t2 = time.perf_counter()
    print("rep %d: graph stage %.2f ms, batch create (to device idle) %.2f ms, solve enqueue %.2f ms, view (waits for the solve) %.2f ms; stats %s" % (rep, (t1 - t0) * 1e3, (ta - t1) * 1e3, (tb - ta) * 1e3, (t2 - tb) * 1e3, {k: round(v, 2) if isinstance(v, float) else v for k, v in p.stats().items()}), flush=True)

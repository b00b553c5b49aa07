"""graph stage + batch creation on the GPU, twice in one process (one-time vs steady-state cost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
ma = synthetic.config4()
g = capi.Graph.from_arrays(ma)
capi.lib().lfr_hip_warmup(0)
for rep in range(3):
    t0 = time.perf_counter(); p = capi.Problem(g, device_graph_stage=0); t1 = time.perf_counter()
    b = capi.Batch(p, device=0); t2 = time.perf_counter()
    st = p.stats()
    print("rep %d: graph stage %.1f ms (tracks %.1f roots %.2f comps %.2f)  batch_create %.1f ms" % (rep, (t1 - t0) * 1e3, st["tracks_ms"], st["roots_ms"], st["graph_cut_ms"], (t2 - t1) * 1e3), flush=True)
    del b, p

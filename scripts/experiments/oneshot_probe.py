"""Where does a ONE-SHOT pipeline run (the `solve` launcher) lose time against the steady state of bench.py?
usage: oneshot_probe.py [file|arrays] [thread|main] [est|exact|none]"""
import os, sys, time, ctypes, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
src, where, res = (sys.argv + ["arrays", "main", "exact"])[1:4]
import numpy as np
from lfr_amd import capi, synthetic
L = capi.lib()
pb = "/tmp/config4.pb"
ma = synthetic.config4()
if src == "file" and not os.path.exists(pb):
    capi.write_matching_file(pb, ma)
nbytes = os.path.getsize(pb) if os.path.exists(pb) else 605434861
def warm():
    L.lfr_hip_warmup(0)
    if res == "est":
        L.lfr_hip_reserve(0, int(nbytes / 230 / 2.8 * 1.1), int(nbytes / 230 * 1.1))
if where == "thread":
    th = threading.Thread(target=warm); th.start()
else:
    warm()
g = capi.Graph.from_matches_file(pb) if src == "file" else capi.Graph.from_arrays(ma)
if where == "thread":
    th.join()
if res == "exact":
    L.lfr_hip_reserve(0, g.n_nodes, g.n_edges // 2)
for rep in range(2):
    g.evict_device()
    L.lfr_hip_synchronize(0)
    t0 = time.perf_counter(); g.to_device(0); L.lfr_hip_synchronize(0); t2 = time.perf_counter()
    p = capi.Problem(g, device_graph_stage=0); t3 = time.perf_counter()
    b = capi.Batch(p, 0); t4 = time.perf_counter()
    b.solve(None, want_stats=False); v = b.positions_view(); t5 = time.perf_counter()
    print("%s %s %s rep %d: upload %.2f ms, graph stage %.2f ms, batch create %.2f ms, solve+view %.2f ms" %
          (src, where, res, rep, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))

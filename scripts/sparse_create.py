"""Where does the creation of a batch with elimination-tree components go?  LFR_HOST_TRACE=1 python scripts/sparse_create.py [tracks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
ma = synthetic.capsized_sparse(n_tracks=int(sys.argv[1]) if len(sys.argv) > 1 else 12000)
g = capi.Graph.from_arrays(ma)
L = capi.lib(); L.lfr_hip_warmup(0); g.to_device(0)
for r in range(4):
    p = capi.Problem(g, device_graph_stage=0)
    L.lfr_hip_synchronize(0)
    print("== repetition %d" % r, file=sys.stderr, flush=True)
    t0 = time.perf_counter(); b = capi.Batch(p, 0); t1 = time.perf_counter()
    b.solve(None, want_stats=False); pos = b.positions_view(); t2 = time.perf_counter()
    print("rep %d: batch %.3f ms, solve + positions %.3f ms" % (r, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    del b, p, pos

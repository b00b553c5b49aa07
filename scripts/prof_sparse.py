"""Cap-sized sparse components (solve_tree_kernel): timing on the GPU box; with LFR_LIB_OVERRIDE=<-DLFR_PROFILE_PHASES -DLFR_PROFILE_TREE / -DLFR_PROFILE_FACTOR build> the phase profiles.
usage: python scripts/prof_sparse.py [n_tracks] [n_images]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 1344
os.environ.setdefault("LFR_VERBOSE", "1")
ma = synthetic.capsized_sparse(n_tracks=nt, n_images=ni)
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g)
t0 = time.perf_counter()
b = capi.Batch(p, 0)
t1 = time.perf_counter()
for i in range(3):
    st = b.solve()
    print("sparse: edges %d comps %d kernel %.3f ms dominant %.3f ms  noconv %d fail %d  (batch creation %.1f ms)" % (
        st["n_edges"], st["n_components"], st["kernel_ms"], st["dominant_kernel_ms"], st["n_no_convergence"], st["n_failed"], (t1 - t0) * 1e3), flush=True)
info = b.component_info()
rows = 2 * info["n_var_nodes"]
tot, cms, ced = b.timing()
print("per launch ms:", {i: round(float(cms[i]), 3) for i in range(len(cms)) if ced[i] > 0}, "edges:", {i: int(ced[i]) for i in range(len(cms)) if ced[i] > 0})
big = rows > 192
print("components above 192 rows: %d, rows max %d mean %.0f, iterations mean %.2f max %d, edges in them %d" % (big.sum(), rows.max(), rows[big].mean(), info["iterations"][big].mean(), info["iterations"][big].max(), info["n_edges"][big].sum()))

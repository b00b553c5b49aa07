"""config 5 (long tracks, workgroup kernels): timing + optional phase profile (LFR_LIB_OVERRIDE=<-DLFR_PROFILE_PHASES build>)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g)
b = capi.Batch(p, 0)
for i in range(3):
    st = b.solve()
    print("config5: edges %d comps %d kernel %.3f ms dominant %.3f ms  noconv %d fail %d" % (st["n_edges"], st["n_components"], st["kernel_ms"], st["dominant_kernel_ms"], st["n_no_convergence"], st["n_failed"]), flush=True)
info = b.component_info()
print("rows: min %d max %d mean %.1f; iterations mean %.2f max %d" % (2 * info["n_var_nodes"].min(), 2 * info["n_var_nodes"].max(), 2 * info["n_var_nodes"].mean(), info["iterations"].mean(), info["iterations"].max()))
tot, cms, ced = b.timing()
print("per launch ms:", {i: round(float(cms[i]), 3) for i in range(len(cms)) if ced[i] > 0}, "edges:", {i: int(ced[i]) for i in range(len(cms)) if ced[i] > 0})
cb = np.bincount(np.minimum(2 * info["n_var_nodes"], 400) // 10)
print("rows histogram (bins of 10):", cb.tolist())
rows = 2 * info["n_var_nodes"]; it = info["iterations"]
top = np.argsort(-it)[:8]
print("most iterations:", [(int(it[i]), int(rows[i]), int(info["n_edges"][i])) for i in top])
for lo, hi in ((33, 88), (89, 130), (131, 192)):
    m = (rows >= lo) & (rows <= hi)
    if m.any(): print("rows %d-%d: comps %d iterations mean %.2f max %d, sum(iter*rows^2) %.3g" % (lo, hi, m.sum(), it[m].mean(), it[m].max(), float((it[m] * rows[m].astype(float) ** 2).sum())))

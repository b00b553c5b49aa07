"""config 5 with a -DLFR_PROFILE_WGTIME=3 build: when and where every workgroup of the workgroup-per-component launches ran."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
b.solve()
st = b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"]
big = info["n_edges"] > 320
t0 = info["final_cost"][big]; life = info["iterations"][big].astype(np.float64); hw = info["termination"][big]; r = rows[big]
t0 = (t0 - t0.min()) / 1e5; t1 = t0 + life / 1e5             # wall_clock64: 100 MHz
print("kernel %.3f ms; workgroups %d; span %.3f ms" % (st["kernel_ms"], big.sum(), t1.max()))
cls = np.where(r <= 88, 0, np.where(r <= 130, 1, 2))
for c, name in enumerate(("S", "M", "L")):
    m = cls == c
    if m.any(): print(" class %s: %4d wgs, first start %.3f last start %.3f last end %.3f, sum life %.1f ms" % (name, m.sum(), t0[m].min(), t0[m].max(), t1[m].max(), (t1[m] - t0[m]).sum()))
print(" distinct (xcc, se, sh, cu) ids: %d; per xcc: %s" % (len(np.unique(hw)), np.bincount(hw >> 8).tolist()))
edges = np.linspace(0, t1.max(), 25)
for i in range(24):
    mid = 0.5 * (edges[i] + edges[i + 1])
    run = (t0 <= mid) & (t1 > mid)
    print("  t=%6.2f ms: running S %3d M %3d L %3d  -> CUs busy %3d" % (mid, (run & (cls == 0)).sum(), (run & (cls == 1)).sum(), (run & (cls == 2)).sum(), len(np.unique(hw[run]))))
# per-CU view: how many L workgroups each CU ran, and how long it sat idle between two of them while the queue was not empty
L = cls == 2
last_start = t0[L].max()
gaps = []
per_cu = {}
for cu in np.unique(hw):
    m = (hw == cu)
    order = np.argsort(t0[m]); a = t0[m][order]; b = t1[m][order]; c = cls[m][order]
    per_cu[cu] = int((c == 2).sum())
    for k in range(1, len(a)):
        if c[k] == 2 and a[k] <= last_start: gaps.append(a[k] - b[:k].max())
gaps = np.array(gaps)
print(" L workgroups per CU: min %d median %d max %d; idle gap before an L workgroup starts (ms): median %.3f mean %.3f p90 %.3f max %.3f, sum over CUs %.1f ms" %
      (min(per_cu.values()), int(np.median(list(per_cu.values()))), max(per_cu.values()), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), gaps.max(), gaps.clip(0).sum()))

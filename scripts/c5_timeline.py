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

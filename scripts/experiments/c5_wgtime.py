"""config 5 with a -DLFR_PROFILE_WGTIME build (LFR_LIB_OVERRIDE): distribution of workgroup lifetimes per class."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
for i in range(2):
    st = b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"]; t = info["final_cost"] / 2.4e6; it = info["iterations"]
print("kernel %.3f ms" % st["kernel_ms"])
for lo, hi in ((33, 88), (89, 130), (131, 192)):
    m = (rows >= lo) & (rows <= hi) & (info["n_edges"] > 320)
    if not m.any(): continue
    tt = t[m]
    print("rows %3d-%3d: %4d workgroups, lifetime ms: mean %.3f median %.3f p90 %.3f p99 %.3f max %.3f sum %.1f; per iteration mean %.3f" %
          (lo, hi, m.sum(), tt.mean(), np.median(tt), np.percentile(tt, 90), np.percentile(tt, 99), tt.max(), tt.sum(), (tt / np.maximum(1, it[m])).mean()))
    top = np.argsort(-tt)[:5]
    print("   slowest:", [(round(float(tt[i]), 2), int(it[m][i]), int(rows[m][i]), int(info["n_edges"][m][i])) for i in top])

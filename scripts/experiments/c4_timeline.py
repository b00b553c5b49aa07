"""config 4 with a -DLFR_PROFILE_WGTIME=3 build: when the waves of the packed launch ran (start, lifetime per component's wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config4()
p = capi.Problem(capi.Graph.from_arrays(ma), device_graph_stage=0)
b = capi.Batch(p, 0)
b.solve(); st = b.solve()
info = b.component_info()
t0 = info["final_cost"]; life = info["iterations"].astype(np.float64); hw = info["termination"]
t0 = (t0 - t0.min()) / 100.0; t1 = t0 + life / 100.0                       # microseconds (wall_clock64: 100 MHz)
# one record per wave: components of a wave share start and (to within the info write) end
key = np.round(t0 * 100).astype(np.int64) * (1 << 20) + hw
_, first = np.unique(key, return_index=True)
w0, w1 = t0[first], t1[first]
print("kernel %.1f us; %d components in ~%d waves; wave lifetime us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f; sum %.0f us = %.1f us x 2048 slots" %
      (st["kernel_ms"] * 1e3, len(t0), len(w0), (w1 - w0).mean(), np.median(w1 - w0), np.percentile(w1 - w0, 90), np.percentile(w1 - w0, 99), (w1 - w0).max(), (w1 - w0).sum(), (w1 - w0).sum() / 2048))
print("last wave start %.1f us, last end %.1f us" % (w0.max(), w1.max()))
edges = np.linspace(0, w1.max(), 21)
for i in range(20):
    mid = 0.5 * (edges[i] + edges[i + 1])
    print("  t=%6.1f us: waves running %5d" % (mid, ((w0 <= mid) & (w1 > mid)).sum()))
xcc = (hw[first] >> 16) & 15
print("per XCC: first wave start us / waves started in the first 10 us / in the first 60 us / total waves")
for x in range(8):
    m = xcc == x
    print("  xcc %d: %7.2f  %5d  %5d  %5d" % (x, w0[m].min(), (w0[m] < 10).sum(), (w0[m] < 60).sum(), m.sum()))
order = np.argsort(w0)
print("start times of the first 4096 waves (us), every 256th:", np.round(w0[order][:4096:256], 1).tolist())
rows_ = 2 * info["n_var_nodes"][first]
print("rows of the waves started in the first 30 us: mean %.1f; of the rest: %.1f" % (rows_[w0 < 30].mean(), rows_[w0 >= 30].mean()))
print("lifetime of the waves started in the first 30 us: mean %.1f us max %.1f" % ((w1 - w0)[w0 < 30].mean(), (w1 - w0)[w0 < 30].max()))

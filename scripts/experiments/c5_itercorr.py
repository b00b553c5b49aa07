"""config 5: does anything known before the solve predict a component's iteration count (for longest-first dispatch)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from scipy.stats import spearmanr
from lfr_amd import capi, synthetic
ma = synthetic.config5()
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g, device_graph_stage=0)
b = capi.Batch(p, 0)
b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"]; it = info["iterations"]; e = info["n_edges"]; fc = info["final_cost"]
track, root, comp = p.labels()
m = rows > 130
print("L class: %d comps; iterations mean %.2f sd %.2f" % (m.sum(), it[m].mean(), it[m].std()))
for name, x in (("rows", rows), ("edges", e), ("edges/rows", e / np.maximum(1, rows)), ("final_cost", fc), ("final_cost/edges", fc / np.maximum(1, e))):
    print("  spearman(iterations, %-16s) = %+.3f" % (name, spearmanr(it[m], x[m]).correlation))
# tracks per component (a component made of several tracks has inter-track / Tukey edges)
cid = info["component"]
tracks_per_comp = np.zeros(comp.max() + 1, np.int64)
ut = np.unique(np.stack([comp, track], 1), axis=0)
np.add.at(tracks_per_comp, ut[:, 0], 1)
tpc = tracks_per_comp[cid]
print("  spearman(iterations, tracks per component) = %+.3f ; comps with >1 track: %d, their mean iterations %.2f vs %.2f" %
      (spearmanr(it[m], tpc[m]).correlation, (tpc[m] > 1).sum(), it[m & (tpc > 1)].mean() if (m & (tpc > 1)).any() else 0, it[m & (tpc == 1)].mean()))

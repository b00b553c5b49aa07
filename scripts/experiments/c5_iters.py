import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic
ma = synthetic.config5()
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g, device_graph_stage=0)
b = capi.Batch(p, 0)
st = b.solve()
info = b.component_info()
it = info["iterations"]
print("components", len(it), "kernel_ms", st["kernel_ms"], "hist", np.bincount(it).tolist())
track, root, comp = p.labels()
ntr = np.array([len(set(track[comp == c])) for c in info["component"][:50]])
order = np.argsort(-it)[:25]
print("top by iterations: (batch index, iters, n_var_nodes, n_edges, tracks in comp)")
for i in order:
    c = info["component"][i]
    print(int(i), int(it[i]), int(info["n_var_nodes"][i]), int(info["n_edges"][i]), len(set(track[comp == c])))

"""Which components of a sparse test configuration fail / differ between two solves (diagnostics for the tree kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.generate(seed=23, n_images=260, n_tracks=2500, track_degree=6, eps_out=0.02, chain_links=2, dup_frac=0.02, ratio_sims=True)
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g)
b = capi.Batch(p, 0)
res = []
for rep in range(3):
    st = b.solve()
    info = b.component_info()
    pos = b.download().copy()
    res.append((info, pos))
    bad = np.flatnonzero(info["termination"] == 2)
    print("solve %d: failed %d:" % (rep, st["n_failed"]), [(int(i), int(info["n_var_nodes"][i]), int(info["n_edges"][i]), int(info["iterations"][i])) for i in bad])
print("positions identical across solves:", (res[0][1] == res[1][1]).all(), (res[1][1] == res[2][1]).all())
print("iterations identical:", (res[0][0]["iterations"] == res[1][0]["iterations"]).all())

import os, sys
sys.path.insert(0, "local-feature-refinement_amd")
import numpy as np
from lfr_amd import capi, synthetic
for name, kw in [("W-only", dict(seed=80, n_images=64, n_tracks=4000)), ("with-block", dict(seed=80, n_images=64, n_tracks=4000, eps_out=0.001)),
                 ("long", dict(seed=76, n_images=96, n_tracks=60, len_dist="uniform", len_lo=20, len_hi=80))]:
    p = capi.Problem(capi.Graph.from_arrays(synthetic.generate(**kw)))
    b = capi.Batch(p, 0); xs = []
    for i in range(4):
        b.solve(); xs.append(b.download().copy())
    print(name, [int((xs[0] != x).sum()) for x in xs[1:]], [float(np.abs(xs[0]-x).max()) for x in xs[1:]])

"""config 5 with a -DLFR_PROFILE_WGTIME build: per-component lifetime + features -> gpurun_out/c5_comps.npz (offline scheduling studies)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g, device_graph_stage=0)
b = capi.Batch(p, 0)
b.solve(); b.solve()
info = b.component_info()
track, root, comp = p.labels()
ut = np.unique(np.stack([comp, track], 1), axis=0)
tpc = np.zeros(comp.max() + 1, np.int64); np.add.at(tpc, ut[:, 0], 1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "c5_comps.npz"), life_ms=info["final_cost"] / 2.4e6, rows=2 * info["n_var_nodes"], edges=info["n_edges"],
         tracks=tpc[info["component"]], component=info["component"])
print("saved", len(info["component"]))

"""config 5 with a -DLFR_PROFILE_WGTIME=2 build: evaluation counts of the slowest components."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config5()
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
st = b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"]; it = info["iterations"]; c = info["final_cost"]
big = np.where(info["n_edges"] > 320)[0]
passes = (c % 1e3); ls = (c // 1e3) % 1e3; cand = (c // 1e6) % 1e3; succ = (c // 1e9) % 1e3; inval = c // 1e12
order = big[np.argsort(-passes[big])[:12]]
print("comp: rows edges iterations | sweeps ls_evals candidates successful invalid")
for i in order:
    print("  %4d %6d %3d | %3d %3d %3d %3d %3d" % (rows[i], info["n_edges"][i], it[i], passes[i], ls[i], cand[i], succ[i], inval[i]))
print("all workgroup comps: sweeps/iteration mean %.2f; ls evals total %d over %d iterations" % ((passes[big] / np.maximum(1, it[big])).mean(), ls[big].sum(), it[big].sum()))

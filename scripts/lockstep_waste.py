"""How much of the packed kernel's wave time is lockstep waste - groups of one wave waiting for the group with the most LM rounds / slot rows?
Config 4 on the GPU box: per-component iterations and sizes from the solve, waves formed as the kernel forms them (batch order, 64/S components
per wave), cost model = rounds x slot rows (sweep) + rounds x rows^2 (solve, small).  Compares the batch order with an order that knew the
iteration counts (an upper bound for any predictor).  usage: python scripts/lockstep_waste.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
ma = synthetic.config4()
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
st = b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"].astype(int); E = info["n_edges"].astype(int); it = info["iterations"].astype(int) + 1     # sweeps ~ iterations + 1
cls = np.where((rows <= 8) & (E <= 24), 0, np.where((rows <= 16) & (E <= 96), 1, np.where((rows <= 24) & (E <= 192), 2, 3)))
S = {0: 8, 1: 16, 2: 32, 3: 64}
tot = {}
for c in range(4):
    m = np.nonzero(cls == c)[0]                      # batch order inside the class
    if not len(m): continue
    G = 64 // S[c]
    def cost(order):
        r = it[order]; sl = -(-E[order] // S[c])
        n = len(order); pad = (-n) % G
        r = np.concatenate([r, np.zeros(pad, int)]).reshape(-1, G); sl = np.concatenate([sl, np.zeros(pad, int)]).reshape(-1, G)
        # a wave runs max(rounds) rounds; in a round it sweeps max(slot rows) of the groups still active: approximate by max r x max sl
        return float((r.max(1) * sl.max(1)).sum()), float((r * sl).sum() / G)
    lock, ideal = cost(m)
    srt = m[np.lexsort((-it[m], -(-(-E[m] // S[c]))))]           # by slot rows, then iterations: what a perfect predictor could do
    lock2, _ = cost(srt)
    print("class S=%2d: %6d components, iterations mean %.2f (min %d max %d); wave cost (rounds x slot rows): batch order %.0f, packed ideal %.0f -> waste x%.3f; "
          "order that knows the iteration counts: waste x%.3f" % (S[c], len(m), it[m].mean() - 1, it[m].min() - 1, it[m].max() - 1, lock, ideal, lock / ideal, lock2 / ideal))
    tot[c] = (lock, ideal, lock2)
print("all classes: waste x%.3f now, x%.3f with known iteration counts" % (sum(v[0] for v in tot.values()) / sum(v[1] for v in tot.values()), sum(v[2] for v in tot.values()) / sum(v[1] for v in tot.values())))

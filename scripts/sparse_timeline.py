"""Cap-sized sparse workload with a -DLFR_PROFILE_WGTIME=3 build (LFR_LIB_OVERRIDE): when every component of the elimination-tree class
ran, for how long, on how many workgroups.   usage: LFR_LIB_OVERRIDE=.../wgtime.so python scripts/sparse_timeline.py [n_tracks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
ma = synthetic.capsized_sparse(n_tracks=nt)
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
b.solve()
st = b.solve()
info = b.component_info()
rows = 2 * info["n_var_nodes"]
big = rows > 192
t0 = info["final_cost"][big]; code = info["termination"][big].astype(np.int64); itw = info["iterations"][big].astype(np.int64); r = rows[big]
it = itw & 255; sweeps = (itw >> 8) & 1023; succ = (itw >> 18) & 255
ran = code != 0
t0, code, it, sweeps, succ, r = t0[ran], code[ran], it[ran], sweeps[ran], succ[ran], r[ran]
life = (code >> 4) / 100.0; T = code & 15          # us
t0 = (t0 - t0.min()) / 100.0
end = t0 + life
print("kernel %.3f ms; %d components; span %.3f ms; sum of lifetimes x team %.1f ms" % (st["kernel_ms"], int(ran.sum()), end.max() / 1e3, (life * T).sum() / 1e3))
o = np.argsort(-end)
print("last to end:  rows  T  iters sweeps accepted   start us   life us   us/iter   end us")
for k in o[:24]:
    print("            %5d  %d  %4d  %4d  %4d  %9.1f %9.1f %8.1f %9.1f" % (r[k], T[k], it[k], sweeps[k], succ[k], t0[k], life[k], life[k] / max(1, it[k]), end[k]))
for t in sorted(set(T.tolist())):
    m = T == t
    print("team size %d: %3d components, us per iteration and row: median %.4f  (rows %d-%d); us/iter median %.1f" % (t, m.sum(), np.median(life[m] / it[m] / r[m]), r[m].min(), r[m].max(), np.median(life[m] / it[m])))
edges = np.linspace(0, end.max(), 21)
for i in range(20):
    mid = 0.5 * (edges[i] + edges[i + 1])
    run = (t0 <= mid) & (end > mid)
    print("  t=%7.1f us: components running %3d, workgroups busy %3d" % (mid, run.sum(), T[run].sum()))

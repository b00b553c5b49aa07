"""Cap-sized sparse components: the elimination-tree kernel with and without TEAMS of workgroups, on the GPU box.
usage: python scripts/sparse_ab.py [n_tracks] [setting ...]      (a setting is a value of LFR_TREE_TEAM: "0" = one workgroup per component)
Prints per setting: ms per solve (HIP events, median of 7), components solved by teams, spin timeouts, and against the first setting
the largest position difference and whether iteration counts / terminations agree; then the bitwise-repeat check."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
settings = sys.argv[2:] or ["0", "700,1500", "350,750"]
ma = synthetic.capsized_sparse(n_tracks=nt)
g = capi.Graph.from_arrays(ma)
p = capi.Problem(g)
ref = None
for sset in settings:
    os.environ["LFR_TREE_TEAM"] = sset
    t0 = time.perf_counter()
    b = capi.Batch(p, 0)
    t_create = (time.perf_counter() - t0) * 1e3
    ms = []
    for i in range(8):
        st = b.solve()
        ms.append(st["kernel_ms"])
    pos = b.download().copy()
    info = b.component_info()
    rep_ok = True
    for _ in range(3):
        b.solve()
        rep_ok = rep_ok and bool((b.download() == pos).all())
    rows = 2 * info["n_var_nodes"]
    big = rows > 192
    line = "LFR_TREE_TEAM=%-10s kernel ms median %.3f min %.3f (first %.3f)  team components %d  spin timeouts %d  failed %d  bitwise repeat %s  create %.1f ms" % (
        sset, statistics.median(ms[1:]), min(ms[1:]), ms[0], b.team_runs(), b.spin_timeouts(), st["n_failed"], rep_ok, t_create)
    if ref is None:
        ref = (pos, info)
    else:
        d = np.abs(pos - ref[0]).max()
        line += "  | vs first: max |dx| %.3e, iterations equal %.4f, terminations equal %s" % (
            d, float((info["iterations"][big] == ref[1]["iterations"][big]).mean()), bool((info["termination"] == ref[1]["termination"]).all()))
    print(line, flush=True)
    del b
print("components above 192 rows: %d; rows max %d; iterations max %d" % (big.sum(), rows.max(), info["iterations"][big].max()))

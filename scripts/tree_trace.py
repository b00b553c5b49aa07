"""Event trace of ONE component in the elimination-tree kernel (a -DLFR_TRACE_TREE build, LFR_DEBUG_TREE_FIRST=1): where the wall time of an
LM iteration goes - phases, and per column task of the factorization / back substitution when it started, when its first entry's column
was ready, when its updates, its elimination and its publication were done.
usage: LFR_LIB_OVERRIDE=.../trace.so LFR_DEBUG_TREE_FIRST=1 python scripts/tree_trace.py [n_tracks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
path = "/tmp/lfr_tree_trace.bin"
os.environ["LFR_TREE_TRACE_FILE"] = path
os.environ.setdefault("LFR_DEBUG_TREE_FIRST", "1")
ma = synthetic.capsized_sparse(n_tracks=nt)
p = capi.Problem(capi.Graph.from_arrays(ma))
b = capi.Batch(p, 0)
b.solve()
st = b.solve()
print("kernel %.3f ms" % st["kernel_ms"])
ev = np.fromfile(path, dtype=np.uint64).reshape(-1, 2)
t = ev[:, 0].astype(np.int64); w = ev[:, 1]
typ = (w >> np.uint64(56)).astype(int); gw = ((w >> np.uint64(48)) & np.uint64(0xff)).astype(int); it = ((w >> np.uint64(32)) & np.uint64(0xffff)).astype(int); col = (w & np.uint64(0xffffffff)).astype(np.int64)
print("%d events, iterations %s, waves %d" % (len(t), sorted(set(it.tolist())), len(set(gw.tolist()))))
CYC = 1.0 / 2400.0     # us per cycle (s_memtime counts core cycles, ~2.4 GHz)
names = {10: "iteration top", 11: "LM diagonal pass + barrier", 12: "factorization", 13: "back substitution", 14: "step pass + reduction", 20: "sweep items", 21: "cost reduction",
         22: "node pass", 23: "point reduction", 15: "(to line search step)", 16: "ls_next_step"}
for I in sorted(set(it.tolist())):
    m = (it == I) & (gw == 0) & (typ >= 10)
    o = np.argsort(t[m]); tt = t[m][o]; ty = typ[m][o]
    print("--- iteration %d, wave 0 of the team: phase boundaries (us since the iteration's top)" % I)
    for k in range(len(tt)):
        print("   %8.1f  +%6.1f  %s" % ((tt[k] - tt[0]) * CYC, (tt[k] - tt[k - 1]) * CYC if k else 0.0, names.get(ty[k], str(ty[k]))))
    # factorization: per column
    f0 = t[(it == I) & (typ == 11) & (gw == 0)]
    if len(f0) == 0: continue
    f0 = f0[0]
    cols = {}
    for k in np.flatnonzero((it == I) & (typ >= 1) & (typ <= 5)):
        J = int(col[k] & 0xffff)
        d = cols.setdefault(J, {})
        d[typ[k]] = (t[k] - f0) * CYC
        if typ[k] == 1: d["pre"] = bool(col[k] & 0x10000); d["ne"] = int(col[k] >> 20); d["gw"] = gw[k]
    done = sorted(cols.items(), key=lambda kv: -kv[1].get(5, 0.0))
    print("   factorization: the 16 columns published last (us since the factorization's start): column wave ne | start, first entry ready, updates done, eliminated, published")
    for J, d in done[:16]:
        print("      col %4d wave %2d ne %2d %s| %7.1f %7.1f %7.1f %7.1f %7.1f   (wait %5.1f, updates %5.1f, turn+elim %5.1f, store+publish %5.1f)" % (
            J, d.get("gw", -1), d.get("ne", -1), "P" if d.get("pre") else " ", d.get(1, 0), d.get(2, 0), d.get(3, 0), d.get(4, 0), d.get(5, 0),
            d.get(2, 0) - d.get(1, 0), d.get(3, 0) - d.get(2, 0), d.get(4, 0) - d.get(3, 0), d.get(5, 0) - d.get(4, 0)))
    allc = list(cols.values())
    print("   all %d columns: mean wait %.1f updates %.1f turn+elim %.1f store+publish %.1f us" % (len(allc), np.mean([d.get(2, 0) - d.get(1, 0) for d in allc]),
          np.mean([d.get(3, 0) - d.get(2, 0) for d in allc]), np.mean([d.get(4, 0) - d.get(3, 0) for d in allc]), np.mean([d.get(5, 0) - d.get(4, 0) for d in allc])))
    b0 = t[(it == I) & (typ == 12) & (gw == 0)]
    if len(b0):
        b0 = b0[0]
        bc = {}
        for k in np.flatnonzero((it == I) & (typ >= 6) & (typ <= 8)):
            bc.setdefault(int(col[k] & 0xffff), {})[typ[k]] = (t[k] - b0) * CYC
        last = sorted(bc.items(), key=lambda kv: -kv[1].get(8, 0.0))
        print("   back substitution: first 8 and last 8 columns solved: column | start, parent ready, published")
        for J, d in sorted(bc.items(), key=lambda kv: kv[1].get(8, 0.0))[:8] + last[:8][::-1]:
            print("      col %4d | %7.1f %7.1f %7.1f" % (J, d.get(6, 0), d.get(7, 0), d.get(8, 0)))

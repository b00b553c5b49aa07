"""cap-sized sparse workload: resident solve time of the tree's library against variants (child process each), two rounds on one box;
positions compared bitwise with the first.  usage: python scripts/experiments/sparse_ab.py main <variant> ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LFR_SP_CHILD"):
    sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.capsized_sparse(n_tracks=12000) if hasattr(synthetic, "capsized_sparse") else None
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    b = capi.Batch(p, 0)
    ts = []
    for i in range(8):
        st = b.solve()
        ts.append(st["kernel_ms"])
    np.save(os.environ["LFR_SP_CHILD"], b.download())
    print("kernel ms: median %.3f min %.3f  (noconv %d fail %d)" % (float(np.median(ts[2:])), min(ts), st["n_no_convergence"], st["n_failed"]), flush=True)
    sys.exit(0)
import numpy as np
ref = None
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, LFR_SP_CHILD="/tmp/sp_%s.npy" % name)
        if name != "main":
            env["LFR_LIB_OVERRIDE"] = os.path.join(ROOT, "local-feature-refinement_amd", "lfr_amd", "_variants", name + ".so")
        print("round %d %-8s" % (rnd, name), end=" ", flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
        pos = np.load("/tmp/sp_%s.npy" % name)
        if ref is None: ref = pos
        print("           max|dx| vs first: %.3g" % float(np.abs(pos - ref).max()), flush=True)

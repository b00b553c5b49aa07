"""A/B of packed-kernel variants on config 4 (GPU box): for every library given (path or name under lfr_amd/_variants/, "main" = the tree's
liblfr_hip.so) a child process solves the same resident batch N times and reports the median kernel time; positions are compared with the
first library's (max |dx|, should be ~1e-16: the variants differ in instruction selection only).
usage: python scripts/ab_packed.py [--tracks 147000] [--reps 15] main gj0 gj1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = os.path.join(ROOT, "local-feature-refinement_amd", "lfr_amd", "_variants")

def child(tracks, reps, ref_path, cfg):
    sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.config4(n_tracks=tracks, seed=2) if cfg == "c4" else synthetic.config2()
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g, device_graph_stage=0)
    b = capi.Batch(p, 0)
    ts = []
    for i in range(reps + 3):
        st = b.solve(None, want_stats=(i == reps + 2))
        t, c, _ = b.timing(0)
        if i >= 2: ts.append(float(t))
    pos = np.array(b.positions_view(), copy=True)
    ts.sort()
    d = -1.0
    if os.path.exists(ref_path):
        d = float(np.abs(np.load(ref_path) - pos).max())
    else:
        np.save(ref_path, pos)
    print("%-28s kernel ms: median %.4f min %.4f max %.4f | mean iters %.4f conv %d fail %d | max|dx| vs first %.3e"
          % (os.environ.get("LFR_AB_NAME"), ts[len(ts) // 2], ts[0], ts[-1], st["sum_iterations"] / max(1, st["n_components"]), st["n_converged"], st["n_failed"], d), flush=True)

if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(int(a[1]), int(a[2]), a[3], a[4]); sys.exit(0)
    tracks, reps, cfg, rounds = 147000, 15, "c4", 1
    while a and a[0].startswith("--"):
        if a[0] == "--tracks": tracks = int(a[1])
        elif a[0] == "--reps": reps = int(a[1])
        elif a[0] == "--cfg": cfg = a[1]
        elif a[0] == "--rounds": rounds = int(a[1])
        a = a[2:]
    ref = "/tmp/ab_packed_ref_%d.npy" % os.getpid()
    for r in range(rounds):
        for name in a:
            env = dict(os.environ, LFR_AB_NAME=name)
            if name != "main":
                env["LFR_LIB_OVERRIDE"] = name if os.path.sep in name else os.path.join(V, name + ".so")
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(tracks), str(reps), ref, cfg], env=env)
    if os.path.exists(ref): os.remove(ref)

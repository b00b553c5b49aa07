"""config 5 (LDS workgroup kernels): which components end the launch?  Lifetimes from a -DLFR_PROFILE_WGTIME=3 build (LFR_TIMING_LIB), iteration /
evaluation counts from the product build (deterministic, same batch).   usage: LFR_TIMING_LIB=.../wgtime.so python scripts/c5_tail.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from lfr_amd import capi, synthetic
    ma = synthetic.config5()
    p = capi.Problem(capi.Graph.from_arrays(ma))
    b = capi.Batch(p, 0)
    b.solve(); st = b.solve()
    info = b.component_info()
    np.savez(sys.argv[2], kernel_ms=st["kernel_ms"], **info)
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "child", "/tmp/c5_prod.npz"], env=env)
env["LFR_LIB_OVERRIDE"] = os.environ["LFR_TIMING_LIB"]
subprocess.check_call([sys.executable, __file__, "child", "/tmp/c5_time.npz"], env=env)
a = np.load("/tmp/c5_prod.npz"); t = np.load("/tmp/c5_time.npz")
assert (a["component"] == t["component"]).all()
rows = 2 * a["n_var_nodes"]; big = a["n_edges"] > 320
t0 = t["final_cost"][big]; life = t["iterations"][big].astype(np.float64) / 100.0; t0 = (t0 - t0.min()) / 100.0   # us
it = a["iterations"][big]; r = rows[big]; ne = a["n_edges"][big]
end = t0 + life
print("kernel %.3f ms (timing build %.3f); %d workgroup components; span %.3f ms; sum of lifetimes %.1f ms = %.3f ms on 256 CUs" % (
    float(a["kernel_ms"]), float(t["kernel_ms"]), big.sum(), end.max() / 1e3, life.sum() / 1e3, life.sum() / 256e3))
o = np.argsort(-end)
print("last to end: rows edges iters  start us  life us  us/iter  end us")
for k in o[:20]:
    print("           %5d %6d %4d %9.1f %8.1f %8.1f %8.1f" % (r[k], ne[k], it[k], t0[k], life[k], life[k] / max(1, it[k]), end[k]))
print("longest lifetimes:")
for k in np.argsort(-life)[:12]:
    print("           %5d %6d %4d %9.1f %8.1f %8.1f %8.1f" % (r[k], ne[k], it[k], t0[k], life[k], life[k] / max(1, it[k]), end[k]))
for lo, hi in ((33, 88), (89, 130), (131, 192)):
    m = (r >= lo) & (r <= hi)
    if m.any(): print("rows %3d-%3d: %4d comps, iterations mean %.1f max %d, us/iter median %.1f, life mean %.1f max %.1f us" % (lo, hi, m.sum(), it[m].mean(), it[m].max(), np.median(life[m] / it[m]), life[m].mean(), life[m].max()))

"""Ad-hoc GPU parity/timing probe (development aid; the real tests are tests/test_gpu_*.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from lfr_amd import capi, synthetic
import lfr_oracle as O

def run(name, ma, threads=8):
    g = capi.Graph.from_arrays(ma); p = capi.Problem(g, device_assembly=bool(os.environ.get('LFR_CHECK_DEVICE')))
    t = time.time(); pos, st = p.solve_hip(0); t_gpu = time.time() - t
    ref = O.run(ma, n_threads=threads)
    if ref["rc"] != 0:      # oversized components: hand the product's cut to the oracle (Graclus is not restatable)
        ref = O.run(ma, n_threads=threads, comp_override=p.labels()[2])
        print("  (oracle used the product's component assignment: %d cut components)" % p.stats()["n_cut_components"])
    err = np.abs(pos - ref["positions"]).max()
    b = capi.Batch(p, 0)
    st2 = b.solve(); st3 = b.solve()
    info = b.component_info()
    oi = ref["infos"][info["component"]]
    it_eq = (oi["iterations"] == info["iterations"]).mean()
    bad = np.nonzero(np.abs(pos - ref["positions"]).max(axis=1) > 6.25e-6)[0]
    print("%-10s nodes %7d edges %8d comps %7d | err %.3e (bad nodes %d) iters-equal %.6f | kernel %.3f ms (2nd %.3f) dominant %.3f ms | oracle solver %.1f ms | fail %d noconv %d"
          % (name, g.n_nodes, st["n_edges"], st["n_components"], err, bad.size, it_eq, st2["kernel_ms"], st3["kernel_ms"], st3["dominant_kernel_ms"], ref["solver_ms"], st["n_failed"], st["n_no_convergence"]), flush=True)
    return err

which = sys.argv[1:] or ["small", "mid", "long", "c2"]
if "small" in which: run("small", synthetic.generate(seed=11, n_images=48, n_tracks=400, eps_out=0.002))
if "mid" in which: run("mid", synthetic.generate(seed=12, n_images=200, n_tracks=20000, eps_out=0.001))
if "long" in which: run("long", synthetic.generate(seed=3, n_images=96, n_tracks=200, len_dist="uniform", len_lo=20, len_hi=60, eps_out=0.0005))
if "c2" in which: run("config2", synthetic.config2())
if "c4" in which: run("config4", synthetic.config4())
if "c5" in which: run("config5", synthetic.config5(), threads=64)

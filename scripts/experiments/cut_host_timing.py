"""Host part of the size cap on config 5's meta graph: time of lfr_debug_recursive_cut by spawn threshold (child processes: the threshold is read
once per process).  usage: python scripts/experiments/cut_host_timing.py"""
import os, pickle, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from lfr_amd import capi, synthetic
PKL = "/tmp/meta5.pkl"
if os.environ.get("LFR_CUT_TIMING_CHILD"):
    m = pickle.load(open(PKL, "rb"))[0]
    e, w = m["edges"], m["w"]
    nw = np.zeros(int(e.max()) + 1, np.int64)
    for t, s in m["node_w"].items(): nw[t] = s
    ts = []
    for r in range(12):
        t = time.perf_counter(); nodes, sub = capi.recursive_cut(e, w, nw, m["cap"]); ts.append((time.perf_counter() - t) * 1e3)
    print("spawn_min %s: recursive_cut ms median %.2f min %.2f (subsets %d)" % (os.environ.get("LFR_CUT_SPAWN_MIN", "default"), np.median(ts[2:]), min(ts), sub.max() + 1), flush=True)
    sys.exit(0)
import cut_quality as cq
if not os.path.exists(PKL):
    pickle.dump(cq.meta_graphs(synthetic.config5()), open(PKL, "wb"))
print("cpus", len(os.sched_getaffinity(0)), flush=True)
for v in (sys.argv[1:] or ["1000000000", "30000", "10000", "2500", "800", "200"]):
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, LFR_CUT_SPAWN_MIN=v, LFR_CUT_TIMING_CHILD="1"))

"""config 5: the graph stage with the block's short pending lists finished on one XCD (LFR_ROUNDS_TAIL="pending,workgroups") - child process
per setting (the switch is read once): median / min graph-stage ms over 8 warm repetitions, tracks_ms, and a checksum of the labels."""
import os, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
if os.environ.get("LFR_RT_CHILD"):
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.config5()
    g = capi.Graph.from_arrays(ma)
    capi.lib().lfr_hip_reserve(0, g.n_nodes, g.n_edges // 2)
    g.to_device(0)
    ts, tr = [], []
    for rep in range(10):
        capi.lib().lfr_hip_synchronize(0)
        t0 = time.perf_counter()
        p = capi.Problem(g, device_graph_stage=0)
        ts.append((time.perf_counter() - t0) * 1e3); tr.append(p.stats()["tracks_ms"])
    lab = p.labels()
    crc = zlib.crc32(np.ascontiguousarray(lab[0]).tobytes()) ^ zlib.crc32(np.ascontiguousarray(lab[2]).tobytes())
    print("LFR_ROUNDS_TAIL=%-10s graph stage ms median %.2f min %.2f | tracks_ms median %.2f min %.2f | rounds %d | labels crc %08x" % (
        os.environ.get("LFR_ROUNDS_TAIL", "default"), np.median(ts[2:]), min(ts), np.median(tr[2:]), min(tr), p.stats()["kruskal_rounds"], crc), flush=True)
    sys.exit(0)
for v in (sys.argv[1:] or ["0", "1024,1", "1024,4", "4096,1", "4096,4", "4096,8", "16384,4", "16384,8", "16384,16", "65536,16", "0"]):
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, LFR_ROUNDS_TAIL=v, LFR_RT_CHILD="1"))

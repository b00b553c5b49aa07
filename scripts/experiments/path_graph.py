"""A path-shaped match graph (node k matched to node k+1, ids increasing along the path): the worst case for a union-find that hooks by index -
how long do the staged unions, the flattening and the label walk of the device graph stage take?  """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic

def path(n, n_images=600, seed=1, shuffle=False):
    rng = np.random.default_rng(seed)
    k = np.arange(n - 1)
    a, b = k, k + 1
    if shuffle:
        perm = rng.permutation(n - 1); a, b = a[perm], b[perm]
    names = ["im%05d.png" % i for i in range(n_images)]
    return synthetic.MatchArrays(image_names=names, facts=np.ones(n_images, np.float32),
                                 pair_img1=(a % n_images).astype(np.int32), pair_img2=(b % n_images).astype(np.int32),
                                 pair_off=np.arange(n, dtype=np.int64), feat1=(a // n_images).astype(np.uint32), feat2=(b // n_images).astype(np.uint32),
                                 sim=rng.uniform(0.5, 1.0, n - 1).astype(np.float32),
                                 disp1=np.zeros((n - 1, 9, 2), np.float32), disp2=np.zeros((n - 1, 9, 2), np.float32))

for n, sh in ((20000, False), (300000, False), (300000, True)):
    ma = path(n, shuffle=sh)
    g = capi.Graph.from_arrays(ma)
    os.environ["LFR_VERBOSE"] = "2"
    t = time.perf_counter()
    try:
        p = capi.Problem(g, device_graph_stage=0)
        st = p.stats()
        print("path of %d nodes (shuffled input %s): %.1f ms; tracks_ms %.2f rounds %d tracks %d components %d" % (n, sh, (time.perf_counter() - t) * 1e3, st["tracks_ms"], st["kruskal_rounds"], st["n_tracks"], st["n_components"]), flush=True)
    except Exception as e:
        print("path of %d nodes: %s after %.1f ms" % (n, e, (time.perf_counter() - t) * 1e3), flush=True)

"""Elimination-tree plans of a workload's large components, on the CPU (no GPU needed): blocks, tiles, levels, column rounds.
usage: python scripts/tree_plan_stats.py [n_tracks] [n_images]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np
from lfr_amd import capi, synthetic


def component_words(ma, p):
    """per component above 96 variable nodes: (n_var, words) in the batch's conventions (variables first, records by source)"""
    track, is_root, comp = p.labels()
    M = ma.sim.shape[0]
    pair_of = np.repeat(np.arange(len(ma.pair_off) - 1), np.diff(ma.pair_off))
    k1 = ma.pair_img1[pair_of].astype(np.int64) << 32 | ma.feat1.astype(np.int64)
    k2 = ma.pair_img2[pair_of].astype(np.int64) << 32 | ma.feat2.astype(np.int64)
    seq = np.empty(2 * M, np.int64); seq[0::2] = k1; seq[1::2] = k2
    uniq, first, inv = np.unique(seq, return_index=True, return_inverse=True)
    rank = np.empty(len(uniq), np.int64); rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
    node = rank[inv]
    n1, n2 = node[0::2], node[1::2]
    N = len(uniq)
    assert N == len(comp)
    keep = comp[n1] == comp[n2]
    src = np.concatenate([n1[keep], n2[keep]]); dst = np.concatenate([n2[keep], n1[keep]])
    kind = (track[src] != track[dst]).astype(np.uint32)
    has_out = np.zeros(N, bool); has_out[src] = True
    var = has_out & (is_root == 0)
    out = []
    order = np.argsort(comp[src], kind="stable")
    src, dst, kind = src[order], dst[order], kind[order]
    cs = comp[src]
    bounds = np.flatnonzero(np.r_[True, cs[1:] != cs[:-1], True])
    for i in range(len(bounds) - 1):
        s, d, k = src[bounds[i]:bounds[i + 1]], dst[bounds[i]:bounds[i + 1]], kind[bounds[i]:bounds[i + 1]]
        nodes = np.unique(np.r_[s, d])
        v = var[nodes]
        if v.sum() <= 96:
            continue
        loc = np.empty(len(nodes), np.int64)
        loc[v] = np.arange(v.sum()); loc[~v] = v.sum() + np.arange((~v).sum())
        ls, ld = loc[np.searchsorted(nodes, s)], loc[np.searchsorted(nodes, d)]
        both_const = (ls >= v.sum()) & (ld >= v.sum())
        ls, ld, k = ls[~both_const], ld[~both_const], k[~both_const]
        w = (ls | ((ld | (k.astype(np.int64) << 15)) << 16)).astype(np.uint32)
        w = w[np.argsort(ls, kind="stable")]
        out.append((int(v.sum()), w))
    return out


if __name__ == "__main__":
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    ni = int(sys.argv[2]) if len(sys.argv) > 2 else 1344
    ma = synthetic.capsized_sparse(n_tracks=nt, n_images=ni)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    comps = component_words(ma, p)
    print("%d components above 96 variable nodes" % len(comps))
    t0 = time.perf_counter()
    rows = []
    for nv, w in comps:
        blob, info = capi.tree_plan(nv, w)
        rows.append((nv, len(w), info))
    dt = time.perf_counter() - t0
    print("plans: %.1f ms (one thread)" % (dt * 1e3))
    rows.sort(key=lambda r: -r[0])
    for nv, ne, info in rows[:12]:
        dense = info["blocks"] * (info["blocks"] + 1) // 2
        print("n_var %5d edges %6d  tracks %4d segments %4d blocks %4d (%.2f nodes/block) tiles %5d (%.1f/col, %.1f %% of dense)  levels %3d  rounds %3d updates %6d"
              % (nv, ne, info["tracks"], info["segments"], info["blocks"], nv / info["blocks"], info["tiles"], info["tiles"] / info["blocks"],
                 100.0 * info["tiles"] / dense, info["levels"], info["column_rounds"], info["updates"]))
    lv = np.array([r[2]["levels"] for r in rows]); rd = np.array([r[2]["column_rounds"] for r in rows]); bl = np.array([r[2]["blocks"] for r in rows])
    print("levels mean %.1f max %d; column rounds mean %.1f max %d; blocks mean %.0f max %d" % (lv.mean(), lv.max(), rd.mean(), rd.max(), bl.mean(), bl.max()))

"""A7 (solve.cc:185-250, 311-364): how good is the built-in two-way cut that stands in for COLMAP/Graclus'
ComputeNormalizedMinGraphCut?  Runs here (CPU only).

For the meta graphs (tracks = nodes, weight = int(100 * sum of similarities), solve.cc:329) of the oversized components of a few
workloads it compares the NORMALIZED CUT VALUE  cut(A,B) (1/vol A + 1/vol B)  of
    builtin   lfr_bisect_graph (csrc/lfr_graph.cpp)
    spectral  Fiedler vector of the normalized Laplacian (scipy eigsh / lobpcg) + the best sweep cut
    kl        networkx Kernighan-Lin refinement started from the built-in partition
on the top-level bisection, and the fraction of inter-track similarity the FULL recursion (down to <= #images nodes per part,
orphans -> singletons) drops with the built-in and with the spectral two-way cut.

    python scripts/cut_quality.py [--json profiles/r03_cut_quality.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
from lfr_amd import capi, synthetic  # noqa: E402


def match_nodes(ma, g):
    img, feat = g.nodes()
    key = img.astype(np.int64) << 32 | feat.astype(np.int64)
    order = np.argsort(key)
    ks = key[order]
    gi = {n: i for i, n in enumerate(g.image_names())}
    remap = np.array([gi.get(n, -1) for n in ma.image_names])
    cnt = np.diff(ma.pair_off)
    k1 = remap[np.repeat(ma.pair_img1, cnt)].astype(np.int64) << 32 | ma.feat1.astype(np.int64)
    k2 = remap[np.repeat(ma.pair_img2, cnt)].astype(np.int64) << 32 | ma.feat2.astype(np.int64)
    return order[np.searchsorted(ks, k1)], order[np.searchsorted(ks, k2)]


def meta_graphs(ma):
    """[(edges (E,2) int, weights int, node_weights {track: size}, cap, sim per meta edge)] of the oversized meta components"""
    g = capi.Graph.from_arrays(ma)
    p_nocut = capi.Problem(g, max_nodes_in_component=1 << 40, device_assembly=True)      # labels only (no batch: components may exceed its 32767-node limit)
    track, root, comp0 = p_nocut.labels()
    p = capi.Problem(g, device_assembly=True)
    comp = p.labels()[2]
    n1, n2 = match_nodes(ma, g)
    cap = g.n_images
    sizes = np.bincount(comp0)
    tsize = np.bincount(track)
    out = []
    inter = track[n1] != track[n2]
    for c in np.nonzero(sizes > cap)[0]:
        m = inter & (comp0[n1] == c)
        ta, tb = track[n1[m]], track[n2[m]]
        lo, hi = np.minimum(ta, tb), np.maximum(ta, tb)
        key = lo.astype(np.int64) * (track.max() + 1) + hi
        uk, inv = np.unique(key, return_inverse=True)
        wsum = np.bincount(inv, weights=ma.sim[m].astype(np.float64))
        edges = np.stack([uk // (track.max() + 1), uk % (track.max() + 1)], 1).astype(np.int64)
        w = (100.0 * wsum).astype(np.int64)                          # static_cast<int>(100 * sum), solve.cc:329
        nodes = np.unique(edges)
        dropped = float(ma.sim[m][comp[n1[m]] != comp[n2[m]]].sum())
        out.append(dict(edges=edges, w=w, sim=wsum, node_w={int(t): int(tsize[t]) for t in nodes}, cap=cap,
                        product_dropped=dropped, total_inter=float(ma.sim[m].sum())))
    return out


def compact(edges):
    nodes, inv = np.unique(edges, return_inverse=True)
    return nodes, inv.reshape(-1, 2)


def ncut_value(e, w, side):
    deg = np.bincount(e[:, 0], weights=w, minlength=len(side)) + np.bincount(e[:, 1], weights=w, minlength=len(side))
    cut = float(w[side[e[:, 0]] != side[e[:, 1]]].sum())
    v0, v1 = float(deg[side == 0].sum()), float(deg[side == 1].sum())
    return cut / v0 + cut / v1 if v0 > 0 and v1 > 0 else float("inf")


def builtin(edges, w):
    part = capi.bisect_graph(edges, np.maximum(w, 1))
    nodes, e = compact(edges)
    return np.array([part[int(t)] for t in nodes], np.int64)


def spectral(edges, w):
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    nodes, e = compact(edges)
    n = len(nodes)
    ww = np.maximum(w, 1).astype(np.float64)
    A = sp.coo_matrix((np.r_[ww, ww], (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n)).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    dis = 1.0 / np.sqrt(deg)
    Lap = sp.identity(n) - sp.diags(dis) @ A @ sp.diags(dis)
    if n <= 3:
        side = np.zeros(n, np.int64); side[n // 2:] = 1
        return side
    try:
        if n < 3000:
            vals, vecs = np.linalg.eigh(Lap.toarray())
            f = vecs[:, 1]
        else:
            rng = np.random.default_rng(0)
            X = rng.standard_normal((n, 4))
            X[:, 0] = np.sqrt(deg)
            vals, vecs = sla.lobpcg(Lap, X, largest=False, tol=1e-6, maxiter=400, M=None)
            f = vecs[:, np.argsort(vals)[1]]
    except Exception:
        vals, vecs = sla.eigsh(Lap, k=2, sigma=-1e-3, which="LM")
        f = vecs[:, np.argsort(vals)[1]]
    f = f * dis
    order = np.argsort(f, kind="stable")
    rank = np.empty(n, np.int64); rank[order] = np.arange(n)
    # sweep: prefix sets of the sorted order; cut changes when an edge's earlier endpoint enters
    lo = np.minimum(rank[e[:, 0]], rank[e[:, 1]]); hi = np.maximum(rank[e[:, 0]], rank[e[:, 1]])
    delta = np.zeros(n + 1)
    np.add.at(delta, lo, ww); np.add.at(delta, hi, -ww)
    cut = np.cumsum(delta)[:n - 1]                                     # cut after taking order[0..i]
    vol0 = np.cumsum(deg[order])[:n - 1]
    vol = deg.sum()
    val = cut / vol0 + cut / (vol - vol0)
    k = int(np.argmin(val))
    side = np.ones(n, np.int64)
    side[order[:k + 1]] = 0
    return side


def kl(edges, w, start):
    import networkx as nx
    from networkx.algorithms.community import kernighan_lin_bisection
    nodes, e = compact(edges)
    G = nx.Graph()
    G.add_nodes_from(range(len(nodes)))
    for (a, b), wi in zip(e, np.maximum(w, 1)):
        if G.has_edge(a, b):
            G[a][b]["weight"] += int(wi)
        else:
            G.add_edge(int(a), int(b), weight=int(wi))
    A, B = kernighan_lin_bisection(G, partition=(set(np.nonzero(start == 0)[0].tolist()), set(np.nonzero(start == 1)[0].tolist())),
                                   weight="weight", max_iter=10, seed=0)
    side = np.ones(len(nodes), np.int64)
    side[list(A)] = 0
    return side


def recursion_dropped(edges, w, sim, node_w, cap, two_way):
    """solve.cc:185-250 around a two-way cut: returns the similarity of the meta edges that end up between different parts"""
    label = {}
    counter = [0]

    def rec(e_idx, nodes):
        weight = sum(node_w[t] for t in nodes)
        if weight <= cap or len(e_idx) == 0:
            for t in nodes:
                label[t] = counter[0]
            counter[0] += 1
            return
        sub = edges[e_idx]
        ns, ce = compact(sub)
        side = two_way(sub, w[e_idx])
        smap = {int(t): int(s) for t, s in zip(ns, side)}
        es = side[ce]
        for s in (0, 1):
            keep = (es[:, 0] == s) & (es[:, 1] == s)
            part_nodes = [t for t in nodes if smap.get(t, -1) == s]
            if sum(node_w[t] for t in part_nodes) <= cap:
                for t in part_nodes:
                    label[t] = counter[0]
                counter[0] += 1
            else:
                in_edges = set(np.unique(sub[keep]).tolist()) if keep.any() else set()
                for t in part_nodes:                                  # orphans -> singletons (solve.cc:240-246)
                    if t not in in_edges:
                        label[t] = counter[0]; counter[0] += 1
                rec(e_idx[keep], [t for t in part_nodes if t in in_edges])

    rec(np.arange(len(edges)), sorted(node_w))
    lab = np.zeros(int(edges.max()) + 1, np.int64)
    for t, v in label.items():
        lab[t] = v
    la, lb = lab[edges[:, 0]], lab[edges[:, 1]]
    return float(sim[la != lb].sum())


def chain_graph(n=100000, seed=1):
    rng = np.random.default_rng(seed)
    e = np.stack([np.arange(n - 1), np.arange(1, n)], 1)
    extra = rng.integers(0, n, size=(n // 50, 2))
    extra = extra[np.abs(extra[:, 0] - extra[:, 1]) < 40]
    extra = extra[extra[:, 0] != extra[:, 1]]
    edges = np.concatenate([e, extra]).astype(np.int64)
    w = rng.integers(20, 100, size=len(edges)).astype(np.int64)
    return edges, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--builtin-only", action="store_true", help="skip the (slow) spectral / Kernighan-Lin legs")
    args = ap.parse_args()
    capi.lib()
    rows = []
    workloads = [("config1_standin", synthetic.config1_standin), ("config3_standin", synthetic.config3_standin), ("config5", synthetic.config5),
                 ("capsized_sparse_8000", lambda: synthetic.capsized_sparse(n_tracks=8000))]
    for name, make in workloads:
        t0 = time.time()
        for k, mg in enumerate(meta_graphs(make())):
            e, w = mg["edges"], mg["w"]
            nodes, ce = compact(e)
            ww = np.maximum(w, 1).astype(np.float64)
            tb = time.time(); sb = builtin(e, w); tb = time.time() - tb
            row = dict(workload=name, component=k, meta_nodes=int(len(nodes)), meta_edges=int(len(e)), cap=int(mg["cap"]),
                       ncut_builtin=ncut_value(ce, ww, sb), ms_builtin=tb * 1e3)
            if args.builtin_only:
                if len(nodes) <= 20000:
                    row["dropped_similarity_fraction_product"] = mg["product_dropped"] / max(mg["total_inter"], 1e-30)
                rows.append(row)
                print(json.dumps(row), flush=True)
                continue
            ts = time.time(); ss = spectral(e, w); ts = time.time() - ts
            row.update(ncut_spectral=ncut_value(ce, ww, ss), ms_spectral=ts * 1e3)
            if len(e) <= 200000:
                row["ncut_kl_from_builtin"] = ncut_value(ce, ww, kl(e, w, sb))
            if len(nodes) <= 20000:
                row["dropped_similarity_fraction_product"] = mg["product_dropped"] / max(mg["total_inter"], 1e-30)
                row["dropped_similarity_fraction_builtin_recursion"] = recursion_dropped(e, w, mg["sim"], mg["node_w"], mg["cap"], builtin) / max(mg["total_inter"], 1e-30)
                row["dropped_similarity_fraction_spectral_recursion"] = recursion_dropped(e, w, mg["sim"], mg["node_w"], mg["cap"], spectral) / max(mg["total_inter"], 1e-30)
            rows.append(row)
            print(json.dumps(row), flush=True)
        print("# %s: %.1f s" % (name, time.time() - t0), flush=True)
    e, w = chain_graph()
    nodes, ce = compact(e)
    ww = w.astype(np.float64)
    tb = time.time(); sb = builtin(e, w); tb = time.time() - tb
    row = dict(workload="chain_1e5_tracks", component=0, meta_nodes=int(len(nodes)), meta_edges=int(len(e)),
               ncut_builtin=ncut_value(ce, ww, sb), ms_builtin=tb * 1e3)
    if not args.builtin_only:
        ts = time.time(); ss = spectral(e, w); ts = time.time() - ts
        row.update(ncut_spectral=ncut_value(ce, ww, ss), ms_spectral=ts * 1e3)
    rows.append(row)
    print(json.dumps(row), flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Graph stage: tracks (solve.cc:489-549), roots (552-582), components (252-373), assembly
(79-143).  The literal Python restatement (oracle/lfr_ref.py: std::set<std::string> semantics,
std::sort+reverse tie-breaks) is the spec; the C oracle and the native host code of
liblfr_hip.so (interned images, linked-list image sets) must agree with it exactly."""
import numpy as np
import pytest

import lfr_oracle as O
import lfr_ref as R
from lfr_amd import capi, synthetic


def bisect_ptr():
    """address of the product's two-way cut (include/lfr.h: lfr_bisect_graph) for the C oracle's recursion"""
    import ctypes
    return ctypes.cast(capi.lib().lfr_bisect_graph, ctypes.c_void_p).value


def fuzz_pairs(seed):
    rng = np.random.default_rng(seed)
    n_images = int(rng.integers(3, 7))
    n_feat = int(rng.integers(2, 6))
    pairs = []
    for _ in range(int(rng.integers(3, 12))):
        a, b = rng.choice(n_images, size=2, replace=False)
        ms = []
        for _ in range(int(rng.integers(0, 6))):
            ms.append({"feature_idx1": int(rng.integers(0, n_feat)), "feature_idx2": int(rng.integers(0, n_feat)),
                       "similarity": float(np.float32(rng.choice([0.5, 0.75, 0.9]))),     # many exact ties
                       "disp1": [tuple(float(np.float32(v)) for v in rng.normal(0, 0.1, 2)) for _ in range(9)],
                       "disp2": [tuple(float(np.float32(v)) for v in rng.normal(0, 0.1, 2)) for _ in range(9)]})
        pairs.append({"image_name1": "im%d" % a, "fact1": 1.0, "image_name2": "im%d" % b, "fact2": 1.0, "matches": ms})
    return pairs


@pytest.mark.parametrize("seed", range(60))
def test_tracks_and_roots_fuzz(lfr_lib, seed):
    pairs = fuzz_pairs(seed)
    g = R.MatchGraph(pairs)
    if g.n_nodes == 0:
        pytest.skip("empty graph")
    track, n_tracks = R.build_tracks(g)
    roots = R.select_roots(g, track, n_tracks)
    ma = synthetic.pairs_to_arrays(pairs)
    native_g = capi.Graph.from_arrays(ma)
    prob = capi.Problem(native_g)
    t2, r2, c2 = prob.labels()
    assert native_g.n_nodes == g.n_nodes
    assert (t2 == np.asarray(track)).all()
    assert (r2 == np.asarray(roots)).all()
    o = O.run(ma, solve=False)
    assert (o["track"] == np.asarray(track)).all() and (o["is_root"] == np.asarray(roots)).all()
    # every track holds at most one node per image (the merge rule of solve.cc:506-511)
    for t in range(n_tracks):
        imgs = [g.node_key[i][0] for i in range(g.n_nodes) if track[i] == t]
        assert len(imgs) == len(set(imgs))
    # components (solve.cc:252-373).  Above the cap the reference calls Graclus, which cannot be restated: the
    # literal restatement runs the reference's recursion / orphan rule / re-labelling around the product's own
    # two-way cut (lfr_bisect_graph), so everything but that primitive is checked independently
    cap = len(g.images_set)
    st = prob.stats()
    comp, n_comp, n_over = R.split_components(g, track, n_tracks, cap, bisect_fn=capi.bisect_graph)
    assert (c2 == np.asarray(comp)).all() and st["n_components"] == n_comp and st["n_cut_components"] == n_over
    assert o["rc"] == (0 if n_over == 0 else -2)
    o2 = O.run(ma, solve=False, bisect=bisect_ptr())
    assert o2["rc"] == 0 and (o2["comp"] == np.asarray(comp)).all() and o2["n_oversized"] == n_over
    # a component may exceed the cap only if it is a single track (tracks are never split)
    for c in np.unique(c2):
        members = np.nonzero(c2 == c)[0]
        assert len(members) <= cap or len(set(t2[members])) == 1


def test_component_equal_to_cap_is_not_cut(lfr_lib):
    """solve.cc:314 uses <=: a component with exactly #images nodes stays whole."""
    def m(f1, f2, s):
        z = [(0.0, 0.0)] * 9
        return {"feature_idx1": f1, "feature_idx2": f2, "similarity": s, "disp1": z, "disp2": z}
    # images a,b,c,d (cap 4).  track {a0,b0}; track {a1,c0}; a1-b0 is rejected (image a twice) -> inter-track edge
    pairs = [{"image_name1": "a", "fact1": 1.0, "image_name2": "b", "fact2": 1.0, "matches": [m(0, 0, 0.9), m(1, 0, 0.5)]},
             {"image_name1": "a", "fact1": 1.0, "image_name2": "c", "fact2": 1.0, "matches": [m(1, 0, 0.8)]},
             {"image_name1": "c", "fact1": 1.0, "image_name2": "d", "fact2": 1.0, "matches": []}]
    res = R.solve_pairs(pairs)
    assert res["n_tracks"] == 2 and res["n_components"] == 1 and res["max_component_size"] == 4 == len({"a", "b", "c", "d"})
    g = capi.Graph.from_arrays(synthetic.pairs_to_arrays(pairs))
    st = capi.Problem(g).stats()
    assert st["n_components"] == 1 and st["n_cut_components"] == 0 and st["max_component_size"] == 4
    # one image fewer (drop the empty c-d pair): cap 3 < 4 nodes -> must be cut
    st2 = capi.Problem(capi.Graph.from_arrays(synthetic.pairs_to_arrays(pairs[:2]))).stats()
    assert st2["n_cut_components"] == 1 and st2["n_components"] == 2


def test_synthetic_graphs_match_oracle_at_scale(lfr_lib):
    ma = synthetic.generate(seed=51, n_images=300, n_tracks=4000, eps_out=0.001)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    o = O.run(ma, solve=False)
    assert o["rc"] == 0
    t, r, c = p.labels()
    assert (t == o["track"]).all() and (r == o["is_root"]).all() and (c == o["comp"]).all()
    st = p.stats()
    for k in ("n_tracks", "max_track_size", "n_components", "max_component_size"):
        assert st[k] == o[k]
    assert st["n_solved_edges"] <= g.n_edges and st["n_solved_components"] <= st["n_components"]


@pytest.mark.parametrize("seed,cap", [(s, c) for s in range(8) for c in (14, 17, 25)])
def test_size_cap_recursion_against_literal_restatement(lfr_lib, seed, cap):
    """Oversized components on purpose (small caps): the product's recursive cut + re-split (lfr_graph.cpp) vs the
    literal restatements of solve.cc:185-250,311-364 in oracle/lfr_ref.py and oracle/lfr_oracle.c, all three around
    the same two-way primitive.  Checks the recursion, the orphan rule, the integer weights and the re-labelling."""
    ma = synthetic.generate(seed=600 + seed, n_images=14, n_tracks=60, eps_out=0.03)
    pairs = ma.to_pairs()
    g = R.MatchGraph(pairs)
    track, n_tracks = R.build_tracks(g)
    comp, n_comp, n_over = R.split_components(g, track, n_tracks, cap, bisect_fn=capi.bisect_graph)
    p = capi.Problem(capi.Graph.from_arrays(ma), cap)
    t2, _, c2 = p.labels()
    assert (t2 == np.asarray(track)).all()
    assert n_over >= 1 and p.stats()["n_cut_components"] == n_over
    assert (c2 == np.asarray(comp)).all() and p.stats()["n_components"] == n_comp
    sizes = np.bincount(c2)
    for c in np.nonzero(sizes > cap)[0]:                       # only a single (uncuttable) track may stay above the cap
        assert len(set(t2[c2 == c])) == 1


@pytest.mark.parametrize("seed", range(6))
def test_product_recursion_equals_literal_recursion_on_random_meta_graphs(lfr_lib, seed):
    """lfr_debug_recursive_cut (the product's recursion: compact sub-graphs, per-thread scratch, halves on threads) against the literal
    recursive_graph_cut of oracle/lfr_ref.py around the same primitive, on random weighted graphs with clusters, isolated edges and
    uneven node weights; the subset NUMBERS must agree too (solve.cc:205-246 numbers subset 0's parts first)."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(20, 160))
    cl = rng.integers(0, max(2, n // 12), n)
    a = rng.integers(0, n, 6 * n); b = rng.integers(0, n, 6 * n)
    keep = (a < b) & ((cl[a] == cl[b]) | (rng.random(6 * n) < 0.08))
    e = np.unique(np.stack([a[keep], b[keep]], 1), axis=0)
    w = rng.integers(0, 400, len(e))                                  # zero weights included (the primitive clamps them to 1)
    nw = rng.integers(1, 30, n)
    cap = int(rng.integers(30, 120))
    nodes, sub = capi.recursive_cut(e, w, nw, cap)
    want = R.recursive_graph_cut([tuple(map(int, x)) for x in e], [int(x) for x in w], {i: int(nw[i]) for i in range(n)}, cap, capi.bisect_graph)
    assert sorted(want) == [int(x) for x in nodes]
    assert [want[int(x)] for x in nodes] == [int(x) for x in sub]
    weights = np.bincount(sub, weights=nw[nodes])
    members = np.bincount(sub)
    assert ((weights <= cap) | (members == 1)).all()                  # only a single node may stay above the cap


def test_graph_cut_of_a_huge_component_is_fast_and_respects_the_cap(lfr_lib):
    """the region growing scanned all nodes per absorbed node (O(n^2) per bisection: 357 ms at 40 k tracks,
    minutes at 10^6).  One meta-component of ~10^5 two-node tracks chained by rejected matches, cap 8."""
    import time
    n_tr, cap = 100_000, 8
    # track t = (image 0, feature t) - (image 1, feature t), similarity 0.9; the match (image 0, feature t) -
    # (image 1, feature t+1), similarity 0.5, is rejected (both tracks already hold both images) and stays as an
    # inter-track edge: ONE meta-component, a chain of 10^5 tracks / 2*10^5 nodes
    f1 = np.concatenate([np.arange(n_tr), np.arange(n_tr - 1)]).astype(np.uint32)
    f2 = np.concatenate([np.arange(n_tr), np.arange(1, n_tr)]).astype(np.uint32)
    sim = np.concatenate([np.full(n_tr, 0.9, np.float32), np.full(n_tr - 1, 0.5, np.float32)])
    M = len(f1)
    ma = synthetic.MatchArrays(["0.png", "1.png"], np.ones(2, np.float32), np.zeros(1, np.int32), np.ones(1, np.int32),
                               np.array([0, M], np.int64), f1, f2, sim, np.zeros((M, 9, 2), np.float32), np.zeros((M, 9, 2), np.float32))
    g = capi.Graph.from_arrays(ma)
    t0 = time.perf_counter()
    p = capi.Problem(g, cap, device_assembly=True)             # labels only: this test is about the cut
    dt = time.perf_counter() - t0
    st = p.stats()
    assert st["n_tracks"] == n_tr and st["n_cut_components"] == 1
    track, _, comp = p.labels()
    sizes = np.bincount(comp)
    assert sizes.max() <= cap and sizes.min() >= 2             # tracks are never split
    assert (np.diff(comp[np.argsort(track, kind="stable")].reshape(n_tr, 2), axis=1) == 0).all()
    assert st["graph_cut_ms"] < 20_000 and dt < 40, (st["graph_cut_ms"], dt)
    print("graph cut of a %d-track chain: %.0f ms" % (n_tr, st["graph_cut_ms"]))


def test_component_override_sidecar(lfr_lib):
    ma = synthetic.generate(seed=52, n_images=10, n_tracks=40, eps_out=0.05)      # forces oversized components
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    assert p.stats()["n_cut_components"] >= 1
    _, _, comp = p.labels()
    p2 = capi.Problem(g, 0, comp)
    assert (p2.labels()[2] == comp).all() and p2.stats()["n_cut_components"] == 0
    o = O.run(ma, solve=False, comp_override=comp)
    assert o["rc"] == 0 and o["n_components"] == p2.stats()["n_components"]


def test_shards_partition_the_solvable_components(lfr_lib):
    ma = synthetic.generate(seed=53, n_images=100, n_tracks=2000)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    all_c, all_e = p.shard_components(0, 1)
    assert len(all_c) == p.stats()["n_solved_components"] and all_e.sum() == p.stats()["n_solved_edges"]
    for world in (2, 3, 8):
        parts = [p.shard_components(r, world) for r in range(world)]
        merged = np.concatenate([c for c, _ in parts])
        assert len(merged) == len(all_c) and set(merged) == set(all_c)          # disjoint cover
        loads = np.array([e.sum() for _, e in parts], float)
        assert loads.max() / loads.mean() < 1.02                                  # LPT balance by edges


def _bisect_spec(edges, weights):
    """The product's two-way cut as DESIGN.md section 3 / lfr_graph.cpp describe it, restated naively (O(n^2)):
    nodes = sorted endpoints; weights max(w, 1) as doubles; maximum-adjacency region growing from attachment 0 (ties ->
    smallest node) until three quarters of the edge-weight volume are inside (or one node is left outside); the prefix of the
    growth order with the smallest normalized cut among those holding between a quarter and three quarters of the volume wins
    (ties -> the earliest) provided its normalized cut is below 0.6 x that of the first prefix holding half of the volume - otherwise, or
    without a prefix in the window, that balanced prefix (or the whole growth); then up to eight sweeps
    in node order moving a node when that lowers cut/vol0 + cut/vol1 (never emptying a side), ending with the first idle sweep."""
    ids = sorted({a for a, _ in edges} | {b for _, b in edges})
    n = len(ids)
    loc = {v: i for i, v in enumerate(ids)}
    adj = [[] for _ in range(n)]
    deg = [0.0] * n
    volume = 0.0
    for (a, b), w in zip(edges, weights):
        w = float(max(int(w), 1)); a = loc[a]; b = loc[b]
        adj[a].append((b, w)); adj[b].append((a, w)); deg[a] += w; deg[b] += w; volume += 2 * w

    def ncut(c, v0):
        v1 = volume - v0
        return c / v0 + c / v1 if v0 > 0 and v1 > 0 else 1e300
    attach = [0.0] * n; inside = [False] * n
    vol0, n0, cut = 0.0, 0, 0.0
    order = []
    best, half = None, None
    while vol0 * 4 < 3 * volume and n0 < n - 1:
        pick = max((i for i in range(n) if not inside[i]), key=lambda i: (attach[i], -i))
        inside[pick] = True; vol0 += deg[pick]; n0 += 1; order.append(pick)
        cut += deg[pick] - 2.0 * attach[pick]
        for v, w in adj[pick]:
            attach[v] += w
        if half is None and vol0 * 2 >= volume:
            half = (n0, cut, vol0)
        if vol0 * 4 >= volume and vol0 * 4 <= 3 * volume:
            val = ncut(cut, vol0)
            if best is None or val < best[0]:
                best = (val, n0, cut, vol0)
    if best is None:
        if half is None:
            half = (n0, cut, vol0)
        n0, cut, vol0 = half
    elif half is not None and not (best[0] < 0.6 * ncut(half[1], half[2])):
        n0, cut, vol0 = half            # a lopsided prefix must beat the balanced one by 0.6x
    else:
        _, n0, cut, vol0 = best
    side = [1] * n
    for i in order[:n0]:
        side[i] = 0
    for _ in range(8):
        moved = False
        for i in range(n):
            same = sum(w for v, w in adj[i] if side[v] == side[i]); other = sum(w for v, w in adj[i] if side[v] != side[i])
            c2 = cut + same - other
            v2 = vol0 - deg[i] if side[i] == 0 else vol0 + deg[i]
            cnt0 = n0 - 1 if side[i] == 0 else n0 + 1
            if cnt0 <= 0 or cnt0 >= n:
                continue
            if ncut(c2, v2) < ncut(cut, vol0):
                side[i] ^= 1; cut, vol0, n0 = c2, v2, cnt0; moved = True
        if not moved:
            break
    return {ids[i]: side[i] for i in range(n)}


def test_bisection_equals_its_naive_restatement(lfr_lib):
    """The C++ cut (indexed heap, CSR adjacency) against the O(n^2) restatement of its own specification: random multigraphs
    with duplicate edges, zero / equal weights (ties everywhere), disconnected pieces, sparse node ids."""
    rng = np.random.default_rng(77)
    for case in range(300):
        n = int(rng.integers(2, 40))
        ids = np.sort(rng.choice(1000, n, replace=False))
        m = int(rng.integers(1, 4 * n))
        a = ids[rng.integers(0, n, m)]; b = ids[rng.integers(0, n, m)]
        keep = a != b
        if not keep.any():
            continue
        e = np.stack([a[keep], b[keep]], 1).astype(np.int32)
        w = rng.choice([0, 1, 1, 5, 5, 50, 100], len(e)).astype(np.int32)
        got = capi.bisect_graph(e, w)
        want = _bisect_spec([tuple(map(int, x)) for x in e], [int(x) for x in w])
        assert got == want, (case, e.tolist(), w.tolist())


_ARGMAX_CHILD = """
import json, sys
import numpy as np
sys.path.insert(0, %r)
from lfr_amd import capi
rng = np.random.default_rng(78)
out = []
for case in range(120):
    n = int(rng.integers(2, 300))
    m = int(rng.integers(n, 40 * n))
    a = rng.integers(0, n, m); b = rng.integers(0, n, m)
    keep = a != b
    e = np.stack([a[keep], b[keep]], 1).astype(np.int32)
    w = rng.choice([0, 1, 1, 5, 5, 50, 100, 700], len(e)).astype(np.int32)
    side = capi.bisect_graph(e, w)
    out.append(sorted(side.items()))
print(json.dumps(out))
"""


def test_bisection_heap_and_vector_scan_grow_the_same_region(lfr_lib):
    """Dense graphs take the arg-max of the region growing with an AVX2 scan over a plain array, sparse ones (and hosts without AVX2) with
    the indexed heap: the same (largest attachment, smallest id), so the same partition.  The same 120 random multigraphs (ties
    everywhere, up to 300 nodes of mean degree up to 80) through both, each in its own process (the choice is read once)."""
    import json, os, subprocess, sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "local-feature-refinement_amd")
    runs = []
    for heap_only in (False, True):
        env = dict(os.environ)
        env.pop("LFR_BISECT_HEAP_ONLY", None)
        if heap_only:
            env["LFR_BISECT_HEAP_ONLY"] = "1"
        r = subprocess.run([sys.executable, "-c", _ARGMAX_CHILD % pkg], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert runs[0] == runs[1]


def test_parallel_cut_is_schedule_independent(lfr_lib):
    """The two halves of a bisection are cut on two threads (big halves only): the labels must not depend on the schedule -
    the same dense meta graph (one giant component, thousands of edges per half) cut eight times gives the same components,
    and they equal the C oracle's sequential recursion around the same two-way cut."""
    ma = synthetic.generate(seed=61, n_images=24, n_tracks=700, len_dist="uniform", len_lo=8, len_hi=20, eps_out=0.03)
    g = capi.Graph.from_arrays(ma)
    labels = []
    for _ in range(8):
        p = capi.Problem(g, device_assembly=True)
        labels.append(p.labels()[2].copy())
    assert p.stats()["n_cut_components"] >= 1
    for c in labels[1:]:
        assert (c == labels[0]).all()
    o = O.run(ma, solve=False, bisect=bisect_ptr())
    assert (o["comp"] == labels[0]).all() and o["n_oversized"] == p.stats()["n_cut_components"]

"""Components above 192 rows (kernel class KC_GLOBAL): the level-scheduled sparse LDL^T along the elimination tree against the C oracle's dense
solver (the reference: Ceres SPARSE_NORMAL_CHOLESKY, solve.cc:147).  Tolerance 1e-4 px = 6.25e-6 units, same trajectory."""
import numpy as np
import pytest

from lfr_amd import capi, synthetic
import lfr_oracle as O
from test_gpu_parity import TOL_UNITS, bisect_ptr

pytestmark = pytest.mark.gpu


def _check(ma, min_rows, oracle_threads=8):
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    b = capi.Batch(p, 0)
    st = b.solve()
    pos = b.download()
    ref = O.run(ma, n_threads=oracle_threads, bisect=bisect_ptr())         # (one task per component: the big ones dominate)
    assert ref["rc"] == 0 and (ref["comp"] == p.labels()[2]).all()
    info = b.component_info()
    rows = 2 * info["n_var_nodes"]
    assert rows.max() >= min_rows and (rows > 192).sum() >= 3
    assert st["n_failed"] == 0
    assert b.spin_timeouts() == 0                                            # no wave gave up waiting for a dependency
    err = np.abs(pos - ref["positions"]).max(axis=1)
    assert err.max() <= TOL_UNITS, "max |dx| = %.3e units on %d nodes" % (err.max(), (err > TOL_UNITS).sum())
    oi = ref["infos"][info["component"]]
    assert (oi["termination"] == info["termination"]).all()
    big = rows > 192
    assert (oi["iterations"][big] == info["iterations"][big]).mean() >= 0.99       # same trajectory, decision for decision
    return g, p, b, st


def test_sparse_components_of_a_few_hundred_rows(lfr_lib):
    """300 images: the size cap leaves components of up to 300 nodes made of short sparse tracks (600-row systems, ~4 % of the tiles)."""
    _check(synthetic.capsized_sparse(n_images=300, n_tracks=3000, seed=17), 400)


def test_sparse_components_with_duplicates_and_random_links(lfr_lib):
    """bushier meta graphs (random wrong matches instead of a chain), duplicated matches, all-pairs tracks mixed in"""
    ma = synthetic.generate(seed=23, n_images=260, n_tracks=2500, track_degree=6, eps_out=0.02, chain_links=2, dup_frac=0.02, ratio_sims=True)
    _check(ma, 300)


def test_cap_sized_sparse_components_at_full_size(lfr_lib):
    """config-4-scale image count: components at the 1344-node cap, ~2.5 k-row systems."""
    g, p, b, st = _check(synthetic.capsized_sparse(n_tracks=2500, seed=7), 2000, oracle_threads=40)
    x1 = b.download().copy()
    for _ in range(5):                                                            # the dependency-counter schedule hands columns to whichever wave is ready:
        b.solve()                                                                 # every tile is still written by one wave in a fixed order
        assert (b.download() == x1).all() and b.spin_timeouts() == 0             # bitwise repeatable


def test_teams_of_workgroups_per_component(lfr_lib, monkeypatch):
    """Round 5: a component above a work threshold is solved by a TEAM of 2 / 4 workgroups on one XCD (LFR_TREE_TEAM).  Same trajectory and
    positions (to rounding: the reductions meet in another order) as one workgroup per component, bitwise repeatable, no wait gave up;
    with tiny thresholds more teams than the chip has units queue up and split on the way down the queue."""
    ma = synthetic.capsized_sparse(n_tracks=2500, seed=7)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    out = {}
    for setting in ("0", "700,1500", "100,200", "100,100"):
        monkeypatch.setenv("LFR_TREE_TEAM", setting)
        b = capi.Batch(p, 0)
        st = b.solve()
        pos = b.download().copy()
        info = b.component_info()
        assert st["n_failed"] == 0 and b.spin_timeouts() == 0, setting
        assert (b.team_runs() > 0) == (setting != "0"), setting
        for _ in range(3):
            b.solve()
            assert (b.download() == pos).all() and b.spin_timeouts() == 0, setting
        out[setting] = (pos, info)
    rows = 2 * out["0"][1]["n_var_nodes"]
    assert (rows >= 1500).sum() >= 3
    for setting in ("700,1500", "100,200", "100,100"):
        assert np.abs(out[setting][0] - out["0"][0]).max() <= 1e-9, setting
        assert (out[setting][1]["termination"] == out["0"][1]["termination"]).all()
        assert (out[setting][1]["iterations"] == out["0"][1]["iterations"]).mean() >= 0.99


def _explicit(n_images, matches, seed):
    """MatchArrays from explicit (image a, image b) matches between feature 0 of the images (one node per image); flows = consistent
    offsets + noise"""
    rng = np.random.default_rng(seed)
    p = np.clip(rng.normal(0.0, 0.15, size=(n_images, 2)), -0.45, 0.45)
    pairs = []
    for a, b in sorted(matches):
        assert a < b

        def flow(src, dst):
            base = p[dst] - p[src]
            return [(float(base[0] + rng.normal(0, 0.02)), float(base[1] + rng.normal(0, 0.02))) for _ in range(9)]
        pairs.append({"image_name1": "%06d.png" % a, "fact1": 1.0, "image_name2": "%06d.png" % b, "fact2": 1.0,
                      "matches": [{"feature_idx1": 0, "feature_idx2": 0, "similarity": float(rng.uniform(0.5, 1.0)),
                                   "disp1": flow(b, a), "disp2": flow(a, b)}]})
    return synthetic.pairs_to_arrays(pairs)


def test_star_and_comb_shaped_tracks(lfr_lib):
    """Shapes the synthetic generator does not make.  A STAR: one image matched against 170 others and nothing else - its centre is the track's
    root (largest score), so the variable nodes are linked only through a constant: 170 independent one-node pieces of the elimination
    tree.  A COMB: a 120-node path with a tooth on every node (240 nodes, a tree).  And a path with a few long-range chords (cycles)."""
    star = [(0, i) for i in range(1, 171)]
    comb = [(200 + i, 201 + i) for i in range(119)] + [(200 + i, 400 + i) for i in range(120)]
    chords = [(700 + i, 701 + i) for i in range(149)] + [(700 + i, 700 + i + 37) for i in range(0, 110, 11)]
    ma = _explicit(900, star + comb + chords, seed=5)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    b = capi.Batch(p, 0)
    st = b.solve()
    pos = b.download()
    ref = O.run(ma, n_threads=8)
    assert ref["rc"] == 0 and (ref["comp"] == p.labels()[2]).all()
    info = b.component_info()
    rows = 2 * info["n_var_nodes"]
    assert sorted(rows.tolist()) == [298, 340, 478]                              # all three above 192 rows: the elimination-tree kernel
    assert st["n_failed"] == 0 and b.spin_timeouts() == 0
    assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS
    oi = ref["infos"][info["component"]]
    assert (oi["termination"] == info["termination"]).all() and (oi["iterations"] == info["iterations"]).all()
    ts = b.tree_stats()
    assert (ts["columns"] > 0).all() and (ts["levels"] >= 1).all()


@pytest.mark.parametrize("occupied", [128, 248])
def test_teams_without_their_cus(lfr_lib, monkeypatch, occupied):
    """the teams of a launch form from workgroups that are resident on one XCD at the same time.  Another kernel holds
    `occupied` of the 256 CUs (lfr_debug_occupy: 512-thread workgroups at the tree kernel's register budget, dealt round-robin to the XCDs)
    while the batch is solved.  128: sixteen CUs per XCD are left - two units each, the rest of the grid arrives when those have emptied
    the queue: same bits as the undisturbed solve.  248: ONE CU per XCD is left, no unit can ever form: after LFR_TEAM_PATIENCE_MS with
    nobody at work the launch goes on with one workgroup per component - nothing fails, no wait runs out, the positions are those of
    one-workgroup solves (equal to the teams' to rounding) - instead of 2^25 polls per wait and failed components."""
    import time
    monkeypatch.setenv("LFR_TEAM_PATIENCE_MS", "2")
    ma = synthetic.capsized_sparse(n_tracks=2500, seed=7)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    b = capi.Batch(p, 0)
    st = b.solve()
    pos0 = b.download().copy()
    info0 = b.component_info()
    assert st["n_failed"] == 0 and b.team_runs() > 0 and b.team_fallbacks() == 0 and b.spin_timeouts() == 0
    hold_ms = 300.0
    capi.occupy_hip(occupied, hold_ms)
    t0 = time.perf_counter()
    st = b.solve()
    pos1 = b.download().copy()
    dt = time.perf_counter() - t0
    info1 = b.component_info()
    assert st["n_failed"] == 0 and b.spin_timeouts() == 0
    assert dt < 3.0, dt                                                           # (a wait that ran out used to take tens of seconds)
    if occupied == 128:
        assert b.team_fallbacks() == 0 and (pos1 == pos0).all()
    else:
        assert b.team_fallbacks() > 0
        assert np.abs(pos1 - pos0).max() <= 1e-9
        assert (info1["termination"] == info0["termination"]).all()
        assert (info1["iterations"] == info0["iterations"]).mean() >= 0.99
    time.sleep(hold_ms / 1e3)                                                     # the CUs are back: teams again, the same bits as before
    st = b.solve()
    assert st["n_failed"] == 0 and b.team_fallbacks() == 0 and b.spin_timeouts() == 0 and b.team_runs() > 0
    assert (b.download() == pos0).all()

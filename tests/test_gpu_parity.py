"""GPU parity tests proper: the HIP path (through the C ABI) against the C oracle on the same
seeded inputs.  Tolerance: 1e-4 px = 6.25e-6 solver units at fact = 1 (north_star; units per
colmap_utils.py:135-136).  Also size-independent properties at BASELINE.json's full size."""
import numpy as np
import pytest

import lfr_oracle as O
from lfr_amd import capi, synthetic

pytestmark = pytest.mark.gpu
TOL_UNITS = 6.25e-6

CASES = {
    # name: (generator kwargs, kernel classes expected to be exercised)
    "tracks_only": dict(seed=71, n_images=64, n_tracks=3000),
    "outliers_tukey": dict(seed=72, n_images=400, n_tracks=3000, eps_out=0.004),
    "noisy": dict(seed=73, n_images=30, n_tracks=1500, sigma_noise=0.25),
    "active_bounds": dict(seed=74, n_images=30, n_tracks=1500, sigma_p=0.7, sigma_noise=0.15),
    "steep_flows": dict(seed=75, n_images=30, n_tracks=1500, sigma_A=0.8, sigma_noise=0.1),
    "long_tracks_block": dict(seed=76, n_images=96, n_tracks=60, len_dist="uniform", len_lo=20, len_hi=80),
    "very_long_tracks_global": dict(seed=77, n_images=128, n_tracks=12, len_dist="uniform", len_lo=92, len_hi=120),
    "config5_like_cut": dict(seed=3, n_images=96, n_tracks=150, len_dist="uniform", len_lo=48, len_hi=96, eps_out=0.002),
}


def bisect_ptr():
    import ctypes
    return ctypes.cast(capi.lib().lfr_bisect_graph, ctypes.c_void_p).value


def solve_both(ma, variant="ceres1", banned=()):
    g = capi.Graph.from_arrays(ma, banned)
    p = capi.Problem(g)
    b = capi.Batch(p, 0, tukey_variant=variant)
    st = b.solve()
    pos = b.download()
    # oversized components: Graclus is not restatable -> the oracle runs the reference's recursion (solve.cc:185-250,
    # 311-364) around the product's two-way primitive and must arrive at the product's components by itself
    ref = O.run(ma, banned=banned, n_threads=8, tukey_variant=variant, bisect=bisect_ptr())
    assert ref["rc"] == 0 and (ref["comp"] == p.labels()[2]).all()
    return g, p, b, st, pos, ref


@pytest.mark.parametrize("name", sorted(CASES))
def test_parity_with_oracle(lfr_lib, name):
    ma = synthetic.generate(**CASES[name])
    g, p, b, st, pos, ref = solve_both(ma)
    assert st["n_components"] == p.stats()["n_solved_components"] > 0
    err = np.abs(pos - ref["positions"]).max(axis=1)
    assert err.max() <= TOL_UNITS, "max |dx| = %.3e units on %d nodes" % (err.max(), (err > TOL_UNITS).sum())
    info = b.component_info()
    oi = ref["infos"][info["component"]]
    assert (oi["termination"] == info["termination"]).all()
    assert (oi["iterations"] == info["iterations"]).mean() >= 0.999      # same trajectory, decision for decision
    # Ceres-equivalent evaluation counts reported by the kernels == the oracle's counters
    ne = info["n_edges"].astype(np.int64)
    # exact unless a rounding-level decision flipped inside a long line search (agree to 0.1 % then)
    assert st["ref_jacobian_passes_edges"] == pytest.approx(int((oi["n_jac_evals"] * ne).sum()), rel=1e-3)
    assert st["ref_cost_passes_edges"] == pytest.approx(int((oi["n_cost_evals"] * ne).sum()), rel=1e-3)


@pytest.mark.parametrize("variant", ["ceres1", "ceres2"])
def test_tukey_variants(lfr_lib, variant):
    ma = synthetic.generate(seed=78, n_images=300, n_tracks=1200, eps_out=0.006)
    _, _, _, st, pos, ref = solve_both(ma, variant)
    assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS


def test_variants_differ(lfr_lib):
    ma = synthetic.generate(seed=78, n_images=300, n_tracks=1200, eps_out=0.006)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    a, _ = p.solve_hip(0, "ceres1")
    b, _ = p.solve_hip(0, "ceres2")
    assert np.abs(a - b).max() > 1e-5        # the factor 2 on inter-track weights is visible


def test_banned_images(lfr_lib):
    ma = synthetic.generate(seed=79, n_images=40, n_tracks=800)
    banned = [ma.image_names[1], ma.image_names[17]]
    g, p, b, st, pos, ref = solve_both(ma, banned=banned)
    assert g.n_nodes == ref["n_nodes"] and g.n_images == 38
    assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS


def test_deterministic_and_roots_fixed(lfr_lib):
    ma = synthetic.generate(seed=80, n_images=64, n_tracks=4000, eps_out=0.001)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    b = capi.Batch(p, 0)
    b.solve()
    x1 = b.download()
    b.solve()
    x2 = b.download()
    assert (x1 == x2).all()                                   # bitwise reproducible
    _, roots, _ = p.labels()
    assert (x1[roots] == 0).all()                             # solve.cc:134-135
    assert np.abs(x1).max() <= 1.0                            # bounds, solve.cc:137-140


def test_shards_reassemble_the_full_solution(lfr_lib):
    ma = synthetic.generate(seed=81, n_images=64, n_tracks=3000)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g)
    full, st = p.solve_hip(0)
    acc = np.zeros_like(full)
    n_comp = 0
    for r in range(3):
        b = capi.Batch(p, 0, shard_rank=r, shard_world=3)
        s = b.solve()
        b.download(acc)
        n_comp += s["n_components"]
    assert n_comp == st["n_components"]
    assert (acc == full).all()


def test_headline_size_properties(lfr_lib):
    """BASELINE.json configs[3] at full size: size-independent properties + a sampled oracle check."""
    ma = synthetic.config4()
    g = capi.Graph.from_arrays(ma)
    assert abs(g.n_edges - 5.0e6) <= 0.01 * 5.0e6            # 5.0M +- 1% directed edges
    p = capi.Problem(g)
    st0 = p.stats()
    assert st0["n_cut_components"] == 0 and st0["n_solved_edges"] == g.n_edges
    b = capi.Batch(p, 0)
    st = b.solve()
    x = b.download()
    assert st["n_failed"] == 0 and st["n_no_convergence"] == 0 and st["n_converged"] == st["n_components"]
    assert st["n_tracks"] == st0["n_tracks"] and st["n_edges"] == g.n_edges
    assert np.isfinite(x).all() and np.abs(x).max() <= 1.0
    _, roots, _ = p.labels()
    assert (x[roots] == 0).all() and int(roots.sum()) == st0["n_tracks"]
    assert st["exec_passes_edges"] <= st["ref_jacobian_passes_edges"] + st["ref_cost_passes_edges"]
    ref = O.run(ma, n_threads=8)
    assert np.abs(x - ref["positions"]).max() <= TOL_UNITS


def test_huge_component_hbm_matrix(lfr_lib):
    """A single 1100-node track (2198 rows, 1.2 M edges): the HBM-matrix workgroup kernel with its
    vectors in the workspace (no LDS row limit)."""
    ma = synthetic.generate(seed=91, n_images=1200, n_tracks=1, len_dist="uniform", len_lo=1100, len_hi=1100,
                            sigma_noise=0.01)
    g, p, b, st, pos, ref = solve_both(ma)
    assert st["n_components"] == 1 and p.stats()["max_component_size"] == 1100
    assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS
    info = b.component_info()
    assert info["n_var_nodes"][0] == 1099 and info["termination"][0] == ref["infos"]["termination"][info["component"][0]]


@pytest.mark.parametrize("maker", ["config1_standin", "config3_standin"])
def test_small_image_count_standins(lfr_lib, maker, tmp_path):
    """BASELINE configs 1 and 3 (Fountain / Herzjesu) as synthetic stand-ins: few images, so
    multi-track components sit at the #images cap; some exceed it and go through the (shared) cut.
    Also run end to end through the drop-in CLI with the component side-car."""
    import os
    import subprocess
    from lfr_amd import wire
    ma = getattr(synthetic, maker)()
    g, p, b, st, pos, ref = solve_both(ma)
    assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS
    assert p.stats()["max_component_size"] <= g.n_images          # solve.cc:205-238: every part ends at or below the cap
    assert p.stats()["n_cut_components"] > 0                      # ... and these stand-ins do go through the cut
    pb, out, side = str(tmp_path / "m.pb"), str(tmp_path / "s.pb"), str(tmp_path / "comp.i64")
    capi.write_matching_file(pb, ma)
    p.labels()[2].astype("<i8").tofile(side)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "multi-view-refinement", "build", "solve"), "--matches_file", pb, "--output_file", out],
                       capture_output=True, text=True, env=dict(os.environ, LFR_COMPONENTS_FILE=side))
    assert r.returncode == 0, r.stderr
    sol = wire.decode_solution_file(open(out, "rb").read())
    got = np.array([d[1:] for im in sol for d in im["displacements"]])
    names = g.image_names()
    img, feat = g.nodes()
    order = {}
    for im in sol:
        for k, d in enumerate(im["displacements"]):
            order[(im["image_name"], d[0])] = (d[1], d[2])
    want = np.array([order[(names[i], int(f))] for i, f in zip(img, feat)])
    assert np.abs(want - ref["positions"].astype(np.float32)).max() <= TOL_UNITS + 1e-7


def test_every_kernel_class_is_exercised(lfr_lib):
    """One graph with track lengths 2..17 plus long tracks: all five packed classes and all four
    workgroup launches (three LDS footprints, HBM matrix) run, and each agrees with the oracle."""
    parts = [synthetic.generate(seed=95, n_images=400, n_tracks=3000),                                   # packed classes
             synthetic.generate(seed=96, n_images=400, n_tracks=40, len_dist="uniform", len_lo=18, len_hi=90),   # LDS matrix
             synthetic.generate(seed=97, n_images=400, n_tracks=3, len_dist="uniform", len_lo=100, len_hi=130)]  # HBM matrix
    seen = set()
    for ma in parts:
        g, p, b, st, pos, ref = solve_both(ma)
        assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS
        info = b.component_info()
        rows, edges = 2 * info["n_var_nodes"], info["n_edges"]
        for r, e in zip(rows, edges):
            cls = ("G8" if r <= 8 and e <= 24 else "G16" if r <= 16 and e <= 48 else "G16_streamed" if r <= 16 and e <= 96 else
                   "G64_2" if r <= 24 and e <= 192 else "G64_4" if r <= 32 and e <= 320 else "BLOCK_S" if r <= 88 else "BLOCK_M" if r <= 130 else "BLOCK_L" if r <= 192 else "GLOBAL")
            seen.add(cls)
    assert seen == {"G8", "G16", "G16_streamed", "G64_2", "G64_4", "BLOCK_S", "BLOCK_M", "BLOCK_L", "GLOBAL"}, seen   # every kernel path, resident and re-read slots


def test_fuzz_small_irregular_graphs(lfr_lib):
    """Hand-style irregular inputs (duplicate matches, similarity ties, image conflicts, singleton
    tracks inside multi-track components, zero flows): HIP vs the C oracle on 150 tiny graphs."""
    from test_graph_stage import fuzz_pairs
    checked = 0
    for seed in range(1000, 1150):
        pairs = fuzz_pairs(seed)
        ma = synthetic.pairs_to_arrays(pairs)
        if ma.n_matches == 0:
            continue
        g = capi.Graph.from_arrays(ma)
        p = capi.Problem(g)
        pos, st = p.solve_hip(0)
        ref = O.run(ma, n_threads=1, bisect=bisect_ptr())
        assert ref["rc"] == 0 and (ref["comp"] == p.labels()[2]).all(), seed
        assert np.abs(pos - ref["positions"]).max() <= TOL_UNITS, seed
        assert st["n_failed"] == int((ref["infos"]["termination"][ref["comp_nvar"] > 0] == 2).sum())
        checked += 1
    assert checked >= 120


@pytest.mark.parametrize("name", ["tracks_only", "outliers_tukey", "long_tracks_block", "config5_like_cut", "very_long_tracks_global"])
def test_device_assembly_equals_host_assembly(lfr_lib, name):
    """lfr_problem_build_labels + GPU-side batch assembly (lfr_assemble.hip) must produce the same
    batch as the host assembly: same component order, sizes, and bit-identical solutions."""
    ma = synthetic.generate(**CASES[name])
    g = capi.Graph.from_arrays(ma)
    ph = capi.Problem(g)
    pd = capi.Problem(g, device_assembly=True)
    assert (ph.labels()[2] == pd.labels()[2]).all()
    bh, bd = capi.Batch(ph, 0), capi.Batch(pd, 0)
    sh, sd = bh.solve(), bd.solve()
    ih, idv = bh.component_info(), bd.component_info()
    for k in ("component", "n_var_nodes", "n_edges", "iterations", "termination"):
        assert (ih[k] == idv[k]).all(), k
    for k in ("n_components", "n_edges", "n_nodes", "n_tracks", "ref_jacobian_passes_edges", "exec_passes_edges"):
        assert sh[k] == sd[k], k
    assert (bh.download() == bd.download()).all()


@pytest.mark.parametrize("kw", [
    dict(seed=71, n_images=64, n_tracks=3000),
    dict(seed=72, n_images=400, n_tracks=3000, eps_out=0.004),
    dict(seed=98, n_images=30, n_tracks=2000, eps_out=0.0005, sim_lo=0.5),
    dict(seed=76, n_images=96, n_tracks=60, len_dist="uniform", len_lo=20, len_hi=80),
])
def test_device_graph_stage_equals_host(lfr_lib, kw):
    """lfr_problem_build_hip: GPU tracks / roots / components vs the host stage, label for label."""
    ma = synthetic.generate(**kw)
    g = capi.Graph.from_arrays(ma)
    ph = capi.Problem(g)
    pd = capi.Problem(g, device_graph_stage=0)
    th, rh, ch = ph.labels()
    td, rd, cd = pd.labels()
    assert (th == td).all() and (rh == rd).all() and (ch == cd).all()
    for k in ("n_tracks", "max_track_size", "n_components", "max_component_size"):
        assert ph.stats()[k] == pd.stats()[k], k
    a, _ = ph.solve_hip(0)
    b, _ = pd.solve_hip(0)
    assert (a == b).all()


def test_device_graph_stage_fuzz_and_fallback(lfr_lib):
    """Ties, duplicates and image conflicts (the order-dependent part) + the host fallback when a
    component exceeds the cap."""
    from test_graph_stage import fuzz_pairs
    n_ok = 0
    for seed in range(2000, 2120):
        ma = synthetic.pairs_to_arrays(fuzz_pairs(seed))
        if ma.n_matches == 0:
            continue
        g = capi.Graph.from_arrays(ma)
        ph, pd = capi.Problem(g), capi.Problem(g, device_graph_stage=0)
        for x, y in zip(ph.labels(), pd.labels()):
            assert (x == y).all(), seed
        assert ph.stats()["n_cut_components"] == pd.stats()["n_cut_components"]
        n_ok += 1
    assert n_ok >= 100


def test_multi_device_entry_point_on_one_gpu(lfr_lib):
    """lfr_solve_hip_multi with the device list [0, 0, 0]: three host threads, three shards, one GPU -
    the sharded result must be bit-identical to the single-batch solve."""
    ma = synthetic.generate(seed=99, n_images=64, n_tracks=3000, eps_out=0.001)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    full, st1 = p.solve_hip(0)
    multi, stm = capi.solve_hip_multi(p, [0, 0, 0])
    assert (full == multi).all()
    for k in ("n_components", "n_edges", "n_tracks", "n_converged", "sum_iterations", "ref_jacobian_passes_edges"):
        assert st1[k] == stm[k], k


def test_device_resident_flows_producer_contract(lfr_lib):
    """lfr_graph_from_arrays_device_flows: the flow grids stay on the GPU (torch tensors standing in
    for the two-view network's output, compute_match_graph.py:163-187); results are bit-identical to
    the host-array path, also with banned images (row indirection)."""
    import ctypes
    ma = synthetic.generate(seed=93, n_images=50, n_tracks=2500, eps_out=0.001)
    # device buffers through the same HIP runtime the library is linked against (a producer such as the
    # two-view network would hand over tensor.data_ptr(); see INTEGRATION.md for the torch import order)
    hip = ctypes.CDLL("libamdhip64.so")
    ptrs = []
    for arr in (ma.disp1, ma.disp2):
        host = np.ascontiguousarray(arr, np.float32).reshape(-1, 18)
        dptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(host.nbytes)) == 0
        assert hip.hipMemcpy(dptr, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(host.nbytes), 1) == 0   # H2D
        ptrs.append(dptr)

    class _P:      # mimic tensor.data_ptr()
        def __init__(self, v):
            self.v = v

        def data_ptr(self):
            return self.v.value
    d1, d2 = _P(ptrs[0]), _P(ptrs[1])
    for banned in ((), (ma.image_names[2], ma.image_names[31])):
        g_host = capi.Graph.from_arrays(ma, banned)
        g_dev = capi.Graph.from_device_flows(ma, d1.data_ptr(), d2.data_ptr(), 0, banned)
        assert g_host.n_nodes == g_dev.n_nodes and g_host.n_edges == g_dev.n_edges
        want, _ = capi.Problem(g_host).solve_hip(0)
        got, _ = capi.Problem(g_dev, device_graph_stage=0).solve_hip(0)
        assert (want == got).all()
        with pytest.raises(capi.LfrError):
            capi.Problem(g_dev)          # host assembly needs host flows
    # the contract of lfr.h: the caller's flows only have to live until the batch exists (the fused gather
    # read them during the first solve).  Overwrite them between Batch() and solve(): the result must not change.
    g_dev = capi.Graph.from_device_flows(ma, d1.data_ptr(), d2.data_ptr(), 0, ())
    want, _ = capi.Problem(capi.Graph.from_arrays(ma)).solve_hip(0)
    b = capi.Batch(capi.Problem(g_dev, device_graph_stage=0), 0)
    for p_ in ptrs:
        assert hip.hipMemset(p_, 0x7f, ctypes.c_size_t(ma.disp1.size * 4)) == 0
    assert hip.hipDeviceSynchronize() == 0
    b.solve()
    assert (want == b.download()).all()
    b.solve()                               # (a second solve takes the record path of a fused batch)
    assert (want == b.download()).all()
    b.close()
    for p_ in ptrs:
        hip.hipFree(p_)


def test_workgroup_classes_bitwise_repeatable(lfr_lib):
    """Long tracks (all three LDS classes): the persistent launches hand components to whichever workgroup is free, the
    assembly accumulates with LDS atomics and the factorization runs on the matrix cores - the result must not depend on any
    of that: four solves of one batch and a second batch agree bit for bit."""
    ma = synthetic.generate(seed=76, n_images=96, n_tracks=120, len_dist="uniform", len_lo=20, len_hi=96, eps_out=0.002)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    b = capi.Batch(p, 0)
    xs = []
    for _ in range(12):                      # (a stress on the factorization's hand-rolled wave hand-offs: a lost or late flag would show as a
        b.solve()                            # timing-dependent result or as a spin that ran out - ADVICE r3)
        xs.append(b.download().copy())
        assert b.spin_timeouts() == 0
    b2 = capi.Batch(p, 0)
    b2.solve()
    xs.append(b2.download().copy())
    for x in xs[1:]:
        assert (x == xs[0]).all()
    rows = 2 * b.component_info()["n_var_nodes"]
    assert (rows <= 88).any() and ((rows > 88) & (rows <= 130)).any() and (rows > 130).any()


def test_fused_sweep_equals_scratch_sweep(lfr_lib, monkeypatch):
    """The workgroup kernels' fused sweep (four lanes per node, destination-side sums in 2^-40 fixed point) against the scratch sweep of
    rounds 1-2 (LFR_SCRATCH_SWEEP=1 at batch creation): same iteration counts and terminations for every component, positions within
    1e-9 units (the fixed-point resolution is 9e-13 per term); duplicated matches (three terms on a cross entry: those components take
    the scratch sweep by themselves) included."""
    ma = synthetic.generate(seed=79, n_images=96, n_tracks=150, len_dist="uniform", len_lo=18, len_hi=96, eps_out=0.002, dup_frac=0.01)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    b = capi.Batch(p, 0)
    b.solve()
    x_fused, info_fused = b.download().copy(), {k: v.copy() for k, v in b.component_info().items()}
    monkeypatch.setenv("LFR_SCRATCH_SWEEP", "1")
    b2 = capi.Batch(p, 0)
    b2.solve()
    x_scr, info_scr = b2.download().copy(), b2.component_info()
    assert (info_fused["iterations"] == info_scr["iterations"]).all()
    assert (info_fused["termination"] == info_scr["termination"]).all()
    assert np.abs(x_fused - x_scr).max() < 1e-9
    assert (2 * info_fused["n_var_nodes"] > 32).sum() >= 50            # workgroup classes are what this is about
    assert not (x_fused == x_scr).all()                                # (the two sweeps do differ in the last bits: the switch works)


def test_small_systems_in_the_workgroup_kernel(lfr_lib):
    """Every row count from 2 to 32 in solve_block_kernel (duplicated matches - solve.cc:476-478 keeps them - push small tracks beyond the
    packed classes' 320 edges): one partial panel, exactly one panel with the right-hand side in a tile of its own, two panels - the shapes the
    blocked factorization (diagonal tile + substitution matrix M, factor_lds) and the tile-wise back substitution special-case."""
    ma = synthetic.generate(seed=58, n_images=17, n_tracks=64, len_dist="uniform", len_lo=2, len_hi=17)
    pairs = ma.to_pairs()
    for pr in pairs:
        pr["matches"] = [m for m in pr["matches"] for _ in range(170)]
    ma2 = synthetic.pairs_to_arrays(pairs)
    g, p, b, st, pos, ref = solve_both(ma2)
    info = b.component_info()
    rows = 2 * info["n_var_nodes"]
    assert (info["n_edges"] > 320).all() and rows.max() <= 32
    assert set(range(2, 33, 2)) <= set(rows.tolist()), sorted(set(rows.tolist()))
    err = np.abs(pos - ref["positions"]).max(axis=1)
    assert err.max() <= TOL_UNITS, "max |dx| = %.3e units" % err.max()
    oi = ref["infos"][info["component"]]
    assert (oi["termination"] == info["termination"]).all() and (oi["iterations"] == info["iterations"]).all()
    assert b.spin_timeouts() == 0

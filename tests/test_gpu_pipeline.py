"""GPU tests of the device pipeline around the solve kernels: device-resident graph and labels,
sharded device assembly, zero-copy flow gather, pinned download view, slab caches.  Everything is
compared bit for bit with the host-assembled path (itself checked against the oracle in
test_gpu_parity.py)."""
import numpy as np
import pytest
import torch  # noqa: F401  (before liblfr_hip.so is loaded: one HIP runtime for both)

from lfr_amd import capi, synthetic

pytestmark = pytest.mark.gpu

MIXED = dict(seed=99, n_images=64, n_tracks=3000, eps_out=0.001)
BLOCKY = dict(seed=76, n_images=96, n_tracks=60, len_dist="uniform", len_lo=20, len_hi=80)
GLOBAL = dict(seed=77, n_images=128, n_tracks=12, len_dist="uniform", len_lo=92, len_hi=120)


def _info_tuple(b):
    i = b.component_info()
    return [i[k].copy() for k in ("component", "n_var_nodes", "n_edges", "iterations", "termination", "final_cost")]


@pytest.mark.parametrize("kw", [MIXED, BLOCKY, GLOBAL])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_device_assembly_equals_host_shards(lfr_lib, kw, world):
    """lfr_batch_create(rank, world) of a labels-only problem (shard filtered on the GPU) == the shard cut
    from the host-assembled batch: same components in the same order, bit-identical solutions; the shards
    partition the solvable components and reassemble the unsharded solution."""
    ma = synthetic.generate(**kw)
    g = capi.Graph.from_arrays(ma)
    ph = capi.Problem(g)
    pd = capi.Problem(g, device_graph_stage=0, flags=capi.FLOWS_STAY_ON_HOST)
    full, _ = ph.solve_hip(0)
    merged = np.zeros_like(full)
    seen = []
    for r in range(world):
        bh, bd = capi.Batch(ph, 0, r, world), capi.Batch(pd, 0, r, world)
        sh, sd = bh.solve(), bd.solve()
        for a, b in zip(_info_tuple(bh), _info_tuple(bd)):
            assert (a == b).all()
        for k in ("n_components", "n_edges", "n_nodes", "n_tracks", "ref_jacobian_passes_edges", "exec_passes_edges"):
            assert sh[k] == sd[k], k
        want = np.zeros_like(full)
        got = np.zeros_like(full)
        bh.download(want)
        bd.download(got)
        assert (want == got).all()
        comps, _ = ph.shard_components(r, world)
        assert (np.sort(comps) == np.sort(bd.component_info()["component"])).all()
        seen.append(comps)
        bd.download(merged)
    allc = np.concatenate(seen)
    assert len(np.unique(allc)) == len(allc) == ph.stats()["n_solved_components"]
    assert (merged == full).all()


def test_multi_device_entry_point_with_device_assembly(lfr_lib):
    """lfr_solve_hip_multi on a labels-only problem: every shard is assembled on "its" GPU ([0, 0, 0] here)."""
    ma = synthetic.generate(**MIXED)
    g = capi.Graph.from_arrays(ma)
    full, st1 = capi.Problem(g).solve_hip(0)
    for flags in (0, capi.FLOWS_STAY_ON_HOST):
        p = capi.Problem(g, device_graph_stage=0, flags=flags)
        multi, stm = capi.solve_hip_multi(p, [0, 0, 0])
        assert (full == multi).all()
        for k in ("n_components", "n_edges", "n_tracks", "n_converged", "sum_iterations", "ref_jacobian_passes_edges"):
            assert st1[k] == stm[k], k
        g.evict_device()


def test_resident_graph_and_eviction(lfr_lib):
    """to_device / evict_device change where the pipeline starts from, never the result."""
    ma = synthetic.generate(**MIXED)
    g = capi.Graph.from_arrays(ma)
    want, _ = capi.Problem(g).solve_hip(0)
    g.to_device(0)                                   # ingest-time upload (streamed-ingest contract)
    a, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
    g.evict_device()                                 # cold again: the pipeline uploads inside its own span
    b, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
    p = capi.Problem(g, device_graph_stage=0)
    g.evict_device()                                 # the problem keeps its own reference to the device copy
    c, _ = p.solve_hip(0)
    for x in (a, b, c):
        assert (x == want).all()


def test_ingest_straight_to_the_device_equals_the_two_call_form(lfr_lib, tmp_path):
    """lfr_graph_from_matches_file_device: the scanner sends the flows ahead of the node numbering; same graph, same positions, also after
    an eviction (the cold path rebuilds the device copy) and for a file whose every pair is banned (nothing to send)."""
    ma = synthetic.generate(**MIXED)
    pb = str(tmp_path / "m.pb")
    capi.write_matching_file(pb, ma)
    g0 = capi.Graph.from_matches_file(pb)
    want, _ = capi.Problem(g0).solve_hip(0)
    for rep in range(2):
        g = capi.Graph.from_matches_file(pb, device=0)
        assert g.n_nodes == g0.n_nodes and g.n_edges == g0.n_edges
        a, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
        assert (a == want).all()
        g.evict_device()
        b, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
        assert (b == want).all()
    empty = capi.Graph.from_matches_file(pb, banned=[n for n in ma.image_names], device=0)
    assert empty.n_nodes == 0
    with pytest.raises(capi.LfrError):
        capi.Graph.from_matches_file(pb, device=-1)


def test_positions_view_matches_download_and_waits_for_the_solve(lfr_lib):
    ma = synthetic.generate(**MIXED)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g, device_graph_stage=0)
    b = capi.Batch(p, 0)
    side = torch.cuda.Stream()                       # a non-blocking stream: the download must order itself after it
    for _ in range(3):
        b.solve(side.cuda_stream, want_stats=False)
    view = b.positions_view()
    copy = b.download()
    assert view.shape == copy.shape and (view == copy).all()
    want, _ = capi.Problem(g).solve_hip(0)
    assert (copy == want).all()


def test_labels_are_fetched_lazily_and_match_the_host_stage(lfr_lib):
    ma = synthetic.generate(**MIXED)
    g = capi.Graph.from_arrays(ma)
    pd = capi.Problem(g, device_graph_stage=0)
    pos, _ = pd.solve_hip(0)                         # whole pipeline without the labels ever visiting the host
    ph = capi.Problem(g)
    for x, y in zip(ph.labels(), pd.labels()):
        assert (x == y).all()
    assert (pos == ph.solve_hip(0)[0]).all()


def test_reserve_and_trim_do_not_change_results(lfr_lib):
    ma = synthetic.generate(**BLOCKY)
    g = capi.Graph.from_arrays(ma)
    want, _ = capi.Problem(g).solve_hip(0)
    assert lfr_lib.lfr_hip_reserve(0, g.n_nodes, g.n_edges // 2) == 0
    a, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
    assert lfr_lib.lfr_hip_trim(0) == 0
    g.evict_device()
    b, _ = capi.Problem(g, device_graph_stage=0).solve_hip(0)
    assert (a == want).all() and (b == want).all()


def test_empty_and_trivial_graphs_through_the_device_pipeline(lfr_lib):
    """No matches; a single match (one 2-node track)."""
    from lfr_amd.synthetic import pairs_to_arrays
    one = pairs_to_arrays([dict(image_name1="a.png", fact1=1.0, image_name2="b.png", fact2=1.0, matches=[
        dict(feature_idx1=0, feature_idx2=0, similarity=0.9, disp1=[(-0.1, -0.1)] * 9, disp2=[(0.1, 0.1)] * 9)])])
    g = capi.Graph.from_arrays(one)
    want, _ = capi.Problem(g).solve_hip(0)
    got, st = capi.Problem(g, device_graph_stage=0).solve_hip(0)
    assert (want == got).all() and st["n_components"] == 1
    for world in (2, 5):                             # more shards than components: empty shards are fine
        p = capi.Problem(g, device_graph_stage=0, flags=capi.FLOWS_STAY_ON_HOST)
        acc = np.zeros_like(want)
        for r in range(world):
            b = capi.Batch(p, 0, r, world)
            b.solve()
            b.download(acc)
        assert (acc == want).all()


# ---- device graph stage on "real" graph shapes: giant connected components, components above the size cap ----
def _labels_equal(ma, **kw):
    g = capi.Graph.from_arrays(ma)
    ph, pd = capi.Problem(g), capi.Problem(g, device_graph_stage=0)
    for x, y in zip(ph.labels(), pd.labels()):
        assert (x == y).all()
    for k in ("n_tracks", "max_track_size", "n_components", "max_component_size", "n_cut_components"):
        assert ph.stats()[k] == pd.stats()[k], k
    return ph, pd


@pytest.mark.parametrize("kw", [
    dict(seed=201, n_images=40, n_tracks=3000, eps_out=0.05),                 # wrong matches link almost every track
    dict(seed=202, n_images=200, n_tracks=6000, eps_out=0.03, sim_lo=0.3),
    dict(seed=203, n_images=96, n_tracks=80, len_dist="uniform", len_lo=40, len_hi=96, eps_out=0.02),   # long tracks, dense
])
def test_giant_connected_component_runs_in_parallel_rounds(lfr_lib, kw, monkeypatch):
    """One connected component holds most matches (what real match graphs look like): the device stage runs the
    constrained union-find in rounds (solve.cc:499-523 semantics, union for union) - labels bit-identical to the
    host stage, also where components exceed the cap (host bisection fed with the device's tracks).  The ordered list
    goes through the rounds in prefix blocks (5000 positions, then doubling: several blocks at this size)."""
    monkeypatch.setenv("LFR_ROUNDS_FIRST_BLOCK", "5000")
    ma = synthetic.generate(**kw)
    ph, pd = _labels_equal(ma)
    assert pd.stats()["kruskal_rounds"] > 0
    a, _ = ph.solve_hip(0)
    b, _ = pd.solve_hip(0)
    assert (a == b).all()


@pytest.mark.parametrize("maker", ["config1_standin", "config3_standin", "config5"])
def test_real_shaped_configs_stay_on_the_device_stage(lfr_lib, maker):
    ma = getattr(synthetic, maker)()
    ph, pd = _labels_equal(ma)
    assert pd.stats()["tracks_ms"] > 0 and pd.stats()["assemble_ms"] == 0    # device stage ran (the host stage fills assemble_ms only when it assembles)


@pytest.mark.parametrize("cooperative", ["0", "1", "tail"])
def test_round_based_union_find_fuzz(lfr_lib, monkeypatch, cooperative):
    """Every connected component through the rounds (LFR_SERIAL_SEGMENT_EDGES=0): ties, duplicated matches,
    image conflicts, same-image matches - the order-dependent corner cases of solve.cc:489-523.  Three schedules of the rounds: one
    launch per round (default), the single cooperative launch with grid barriers (LFR_ROUNDS_COOPERATIVE=1), and every block's pending
    list finished by two workgroups on one XCD (LFR_ROUNDS_TAIL, round 6: measured, off by default)."""
    from test_graph_stage import fuzz_pairs
    if cooperative == "tail":
        monkeypatch.setenv("LFR_ROUNDS_TAIL", "1000000,2")
    else:
        monkeypatch.setenv("LFR_ROUNDS_COOPERATIVE", cooperative)
    monkeypatch.setenv("LFR_SERIAL_SEGMENT_EDGES", "0")
    monkeypatch.setenv("LFR_ROUNDS_FIRST_BLOCK", "3")          # block boundaries inside every component, ties across them
    n_ok = 0
    for seed in range(3000, 3150):
        ma = synthetic.pairs_to_arrays(fuzz_pairs(seed))
        if ma.n_matches == 0:
            continue
        _, pd = _labels_equal(ma)
        assert pd.stats()["kruskal_rounds"] > 0
        n_ok += 1
    for seed in (71, 72):
        _labels_equal(synthetic.generate(seed=seed, n_images=64, n_tracks=3000, eps_out=0.004 * (seed - 71)))
    assert n_ok >= 120


@pytest.mark.parametrize("road,max_run", [("count", None), ("one_sort", None), ("one_sort", "2"), ("one_sort", "0")])
def test_equal_similarities_keep_the_reference_order(lfr_lib, monkeypatch, road, max_run):
    """Three roads to the reference's order of the matches inside a connected component (descending (sim, n1, n2), solve.cc:489).  Since
    round 6 small components COUNT: grouped by component, every match takes the number of matches that precede it (k_rank_sort; the full
    comparison, no ties left).  A giant component - or LFR_ONE_SORT_ORDER=1 - takes ONE sort by (component, similarity) and puts runs of
    equal similarities into (n1, n2) descending order in place; a run above kMaxTieRun (LFR_MAX_TIE_RUN) sends that to the three stable
    sorts.  Quantized similarities in small components (short runs), in a giant component (runs of thousands) and the fuzz cases (three
    similarity values): labels bit-identical to the host stage whichever way the order was made."""
    from test_graph_stage import fuzz_pairs
    if road == "one_sort":
        monkeypatch.setenv("LFR_ONE_SORT_ORDER", "1")
    if max_run is not None:
        monkeypatch.setenv("LFR_MAX_TIE_RUN", max_run)
    small = synthetic.generate(seed=311, n_images=64, n_tracks=3000, eps_out=0.001)
    small.sim[:] = np.round(small.sim * 64.0) / 64.0                               # ~17 matches per component over a few dozen values
    _, pd = _labels_equal(small)
    assert pd.stats()["tie_resorts"] == (1 if max_run == "2" else 0)               # counted / fixed in place ("0": three sorts from the start)
    giant = synthetic.generate(seed=312, n_images=40, n_tracks=3000, eps_out=0.05)
    giant.sim[:] = np.round(giant.sim * 8.0) / 8.0
    ph, pd = _labels_equal(giant)
    assert pd.stats()["kruskal_rounds"] > 0
    assert pd.stats()["tie_resorts"] == (0 if max_run == "0" else 1)               # runs of thousands: the three sorts
    a, _ = ph.solve_hip(0)
    b, _ = pd.solve_hip(0)
    assert (a == b).all()
    for seed in range(3200, 3260):
        ma = synthetic.pairs_to_arrays(fuzz_pairs(seed))
        if ma.n_matches:
            _labels_equal(ma)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(lfr_lib, scaling):
    """bench.py's N>1 path end to end: two ranks (gloo, both on GPU 0) started by bench.py itself; weak = one graph per
    rank, strong = one graph sharded on the device; the weak line carries the strong-scaling object as well."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFR_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--devices", "0,0", "--steps", "2", "--warmup", "1",
                        "--tracks", "20000", "--span-reps", "2", "--no-cpu-baseline", "--scaling", scaling],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    mg = out["multi_gpu"]                       # the line verifies itself: ranks counted by an all-reduce, per-rank records
    assert mg["ranks_seen"] == 2 and [r["rank"] for r in mg["per_rank"]] == [0, 1] and mg["backend"] == "gloo"
    assert len({r["pid"] for r in mg["per_rank"]}) == 2 and mg["edges_sum_over_ranks"] == mg["edges_headline"]
    assert mg["ms_per_step_max"] <= out["ms_per_step"] * 1.001
    one_graph = out["config"]["edges_per_gpu"] if scaling == "weak" else None
    if scaling == "weak":
        assert "strong_scaling" in out and out["strong_scaling"]["edges"] > 0 and "weak_scaling" not in out
        assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * one_graph) / (2 * one_graph) < 0.02      # two graphs
    else:
        total = out["value"] * out["ms_per_step"] * 1e-3
        assert abs(out["config"]["edges_per_gpu"] * 2 - total) / total < 0.05                                # one graph, two shards
        assert abs(out["weak_scaling"]["edges"] - 2 * total) / total < 0.1                                   # beside it: a graph per rank
    for k in ("solver_span", "total_span", "total_span_resident_graph"):
        assert out[k]["ms"] > 0


def test_bench_four_ranks_on_one_gpu_at_full_size(lfr_lib):
    """Readiness for the first 8-GPU lease: four ranks (gloo) on GPU 0 at the HEADLINE size with bench.py's
    defaults - ONE config-4 graph sharded four ways on the device (snake deal, zero-copy gather of each shard's flows), the edges of
    the shards add up to the graph; the weak leg (a 5 M-edge graph per rank) rides along as a secondary object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFR_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--devices", "0,0,0,0", "--steps", "2", "--warmup", "1",
                        "--span-reps", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # the default for N > 1 is STRONG scaling: `value` is ONE config-4 graph whose components are sharded four ways
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["solve"]["failed"] == 0
    total = out["value"] * out["ms_per_step"] * 1e-3
    assert 4.9e6 < total < 5.1e6                                                                  # one graph per step ...
    assert abs(4 * out["config"]["edges_per_gpu"] - total) / total < 0.02                          # ... in four shards of equal size
    ws = out["weak_scaling"]
    assert abs(ws["edges"] - 4 * total) / (4 * total) < 0.02 and ws["failed"] == 0 and ws["value"] > 0      # four graphs, one per rank


def test_bench_two_gpus_over_rccl(lfr_lib):
    """The first multi-GPU lease should exercise RCCL, not discover it: bench.py --gpus 2 with the default backend (nccl = RCCL on
    ROCm), one rank per GPU.  Skipped on one-GPU boxes."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LFR_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--tracks", "20000",
                        "--span-reps", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["weak_scaling"]["edges"] > 0
    assert out["solve"]["failed"] == 0
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["backend"] == "nccl" and mg["distinct_devices"] == 2


@pytest.mark.parametrize("world", [2, 4, 8])
def test_graph_stage_sharded_by_connected_component(lfr_lib, world):
    """with one process per GPU every rank used to repeat the whole graph stage.  lfr_problem_build_hip_shard runs tracks /
    roots / components over the connected components of the match graph dealt to ONE rank (solve.cc:489-541 never joins two of them); the
    batch of that problem is the rank's share.  The union of the ranks' positions is BITWISE the unsharded solve, the shards' edges and
    tracks add up, no node is solved twice - for short tracks, with wrong matches (components of several tracks) and with components above
    the size cap.  A graph that is one giant connected component cannot be dealt out: the problem then covers everything (cc_sharded False)."""
    for kw in (dict(seed=31, n_images=64, n_tracks=6000),
               dict(seed=32, n_images=48, n_tracks=3000, eps_out=0.002),
               dict(seed=33, n_images=24, n_tracks=1500, eps_out=0.004, len_dist="uniform", len_lo=6, len_hi=24)):
        ma = synthetic.generate(**kw)
        g = capi.Graph.from_arrays(ma)
        pw = capi.Problem(g, device_graph_stage=0)
        bw = capi.Batch(pw, 0)
        stw = bw.solve()
        want = bw.download().copy()
        got = np.zeros_like(want)
        edges = tracks = comps = 0
        for r in range(world):
            pr = capi.Problem(g, device_graph_stage=0, shard=(r, world))
            assert pr.cc_sharded or kw["seed"] != 31                         # (small graphs with large components may not balance: whole graph + snake deal)
            br = capi.Batch(pr, 0) if pr.cc_sharded else capi.Batch(pr, 0, r, world)
            st = br.solve()
            pos = br.download()
            assert st["n_failed"] == stw["n_failed"] == 0
            touched = (pos != 0).any(axis=1)
            assert not (touched & (got != 0).any(axis=1)).any()             # no node solved by two ranks
            got[touched] = pos[touched]
            edges += st["n_edges"]; tracks += st["n_tracks"]; comps += st["n_components"]
        assert (got == want).all()
        assert (edges, tracks, comps) == (stw["n_edges"], stw["n_tracks"], stw["n_components"])
    giant = synthetic.generate(seed=34, n_images=40, n_tracks=800, eps_out=0.05)       # wrong matches link (nearly) everything
    gg = capi.Graph.from_arrays(giant)
    pg = capi.Problem(gg, device_graph_stage=0, shard=(0, world))
    assert not pg.cc_sharded
    pw = capi.Problem(gg, device_graph_stage=0)
    assert (pg.labels()[2] == pw.labels()[2]).all()


def test_sharded_problem_first_on_a_cold_graph(lfr_lib):
    """a connected-component shard carries its parent's upload events and chunk bounds, but its match indices are
    compacted - the record gather of a shard built FIRST on a graph whose flows are still on their way to HBM (flags = 0: staged, chunked
    upload on the copy stream) filtered on the wrong index and could read rows that had not landed.  Long tracks: workgroup classes, so the
    assembly writes records.  Every rank's problem is built on a fresh Graph (cold), before any unsharded problem uploaded the flows."""
    ma = synthetic.generate(seed=3, n_images=96, n_tracks=600, len_dist="uniform", len_lo=20, len_hi=60, eps_out=0.0005)
    world = 2
    got = None
    edges = 0
    for r in range(world):
        g = capi.Graph.from_arrays(ma)                                       # cold: nothing of it is in HBM yet
        pr = capi.Problem(g, device_graph_stage=0, shard=(r, world))         # flags = 0: the flows are staged by this very call
        br = capi.Batch(pr, 0) if pr.cc_sharded else capi.Batch(pr, 0, r, world)
        st = br.solve()
        assert st["n_failed"] == 0
        pos = br.download().copy()
        if got is None:
            got = np.zeros_like(pos)
        touched = (pos != 0).any(axis=1)
        assert not (touched & (got != 0).any(axis=1)).any()
        got[touched] = pos[touched]
        edges += st["n_edges"]
    gw = capi.Graph.from_arrays(ma)
    pw = capi.Problem(gw, device_graph_stage=0)
    bw = capi.Batch(pw, 0)
    stw = bw.solve()
    want = bw.download()
    assert edges == stw["n_edges"]
    assert (got == want).all()


def test_small_component_with_many_edges_takes_the_late_workgroup_path(lfr_lib):
    """No component above 17 nodes, so the device assembly expects packed classes only (three-pass edge sort, no incidence lists, no
    records) - but duplicated matches push one 17-node track beyond 320 edges, into a workgroup class: the assembly's summary has the last
    word and redoes the sort by source node, the words, the incidence lists and the records.  Same positions as the host-assembled batch."""
    ma = synthetic.generate(seed=41, n_images=17, n_tracks=40, len_dist="uniform", len_lo=3, len_hi=17)
    pairs = ma.to_pairs()
    for pr in pairs:                                              # every match three times (solve.cc:476-478 keeps duplicates)
        pr["matches"] = pr["matches"] + [m for m in pr["matches"] for _ in range(2)]
    ma2 = synthetic.pairs_to_arrays(pairs)
    g = capi.Graph.from_arrays(ma2)
    ph = capi.Problem(g)                                          # host assembly
    want, _ = ph.solve_hip(0)
    pd = capi.Problem(g, device_graph_stage=0)
    assert pd.stats()["max_component_size"] <= 17
    bd = capi.Batch(pd, 0)
    bd.solve()
    got = bd.download()
    info = bd.component_info()
    assert (info["n_edges"] > 320).any() and (2 * info["n_var_nodes"] <= 32).all()      # the case this test is about
    assert (got == want).all()


def test_positions_view_f32_is_the_rounded_f64_view(lfr_lib):
    """lfr_batch_positions_view_f32: the displacements as the reference's SolutionFile holds them (float, solve.cc:661-664), converted
    on the device - every value the float nearest to the double of lfr_batch_positions_view, also for a shard (other nodes read 0)."""
    ma = synthetic.generate(seed=51, n_images=32, n_tracks=5000, eps_out=0.001)
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g, device_graph_stage=0)
    b = capi.Batch(p, 0)
    b.solve()
    v64 = np.array(b.positions_view(), copy=True)
    v32 = np.array(b.positions_view_f32(), copy=True)
    assert v32.dtype == np.float32 and v32.shape == v64.shape
    assert (v32 == v64.astype(np.float32)).all() and np.abs(v32).max() > 0
    bs = capi.Batch(p, 0, 1, 3)
    bs.solve()
    assert (np.array(bs.positions_view_f32()) == np.array(bs.positions_view()).astype(np.float32)).all()



def test_graph_stage_sharded_in_the_one_process_multi_gpu_entry(lfr_lib, tmp_path):
    """lfr_solve_graph_hip_multi - the drop-in's LFR_GPUS path - runs tracks / roots / components, the assembly and the
    solve per device over the connected components dealt to it (here four shards on GPU 0): positions BITWISE the one-GPU pipeline's,
    the stdout counts (# tracks, max track size, # components, max component size) merged from the shards; a graph that is one connected
    component falls back to sharing out the components of the whole problem.  The CLI with LFR_GPUS=0,0 writes the same SolutionFile
    and the same stdout counts as with one GPU."""
    import os
    import subprocess
    import sys
    for kw in (dict(seed=61, n_images=64, n_tracks=6000), dict(seed=62, n_images=48, n_tracks=3000, eps_out=0.002),
               dict(seed=34, n_images=40, n_tracks=800, eps_out=0.05)):
        ma = synthetic.generate(**kw)
        g = capi.Graph.from_arrays(ma)
        pw = capi.Problem(g, device_graph_stage=0)
        bw = capi.Batch(pw, 0)
        stw = bw.solve()
        want = bw.download().copy()
        pst = pw.stats()
        g2 = capi.Graph.from_arrays(ma)                               # cold graph: the multi entry uploads what it needs itself
        pos, mst, sst = capi.solve_graph_hip_multi(g2, [0, 0, 0, 0])
        assert (pos == want).all()
        for k in ("n_tracks", "max_track_size", "n_components", "max_component_size", "n_cut_components", "n_solved_edges"):
            assert mst[k] == pst[k], k
        for k in ("n_components", "n_edges", "n_tracks", "n_converged", "sum_iterations"):
            assert sst[k] == stw[k], k
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    solve = os.path.join(root, "multi-view-refinement", "build", "solve")
    pb = str(tmp_path / "m.pb")
    capi.write_matching_file(pb, synthetic.generate(seed=63, n_images=32, n_tracks=2000, eps_out=0.001))
    outs = {}
    for name, gpus in (("one", None), ("two", "0,0")):
        env = dict(os.environ)
        env.pop("LFR_GPUS", None)
        if gpus:
            env["LFR_GPUS"] = gpus
        out = str(tmp_path / (name + ".pb"))
        r = subprocess.run([sys.executable, solve, "--matches_file", pb, "--output_file", out], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if not l.startswith(("Graph-cut time", "Solver time", "Total time"))]
        outs[name] = (lines, open(out, "rb").read())
    assert outs["one"][0] == outs["two"][0]
    assert outs["one"][1] == outs["two"][1]

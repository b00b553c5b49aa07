"""Committed golden vectors (tests/golden/, produced by oracle/lfr_ref.py — see make_golden.py).
CPU: the C oracle and the native graph stage reproduce them.  GPU: the HIP path does."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import lfr_oracle as O
from lfr_amd import capi, synthetic, wire

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
NAMES = ["clean", "outliers", "noisy", "bounds", "linesearch", "linesearch2", "sparse_ratio", "sparse_long", "sparse_tree", "cut"]
TOL_UNITS = 6.25e-6          # 1e-4 px at fact = 1 (colmap_utils.py:135-136); north_star tolerance


def load(name):
    pairs = wire.decode_matching_file(open(os.path.join(GOLD, name + ".pb"), "rb").read())
    return pairs, np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("variant", ["ceres1", "ceres2"])
def test_c_oracle_reproduces_golden(name, variant):
    pairs, z = load(name)
    ma = synthetic.pairs_to_arrays(pairs)
    worst = int(z["trace_component_" + variant])
    # ("cut": components above the size cap - the reference's recursion around the product's two-way primitive, as in make_golden.py)
    bisect = ctypes.cast(capi.lib().lfr_bisect_graph, ctypes.c_void_p).value if name == "cut" else None
    o = O.run(ma, tukey_variant=variant, trace_comp=worst, bisect=bisect)
    assert (o["track"] == z["track"]).all() and (o["is_root"] == z["is_root"]).all() and (o["comp"] == z["comp"]).all()
    # 1e-9: the quintic line-search interpolation uses different (equally valid) root finders /
    # linear solvers in the two restatements; everything else agrees to ~1e-16
    np.testing.assert_allclose(o["positions"], z["positions_" + variant], rtol=0, atol=1e-9)
    assert (o["infos"]["iterations"] == z["iterations_" + variant]).all()
    assert (o["infos"]["termination"] == z["termination_" + variant]).all()
    assert (o["infos"]["n_ls_evals"] == z["n_ls_evals_" + variant]).all()
    tr = z["trace_" + variant]
    ct = np.array([r for r in o["trace"] if int(r[7]) in (1, 2)])
    assert ct.shape[0] == tr.shape[0]
    np.testing.assert_allclose(ct[:, 1], tr[:, 1], rtol=1e-8)       # cost per iteration
    np.testing.assert_allclose(ct[:, 4], tr[:, 4], rtol=1e-6)        # trust-region radius per iteration


@pytest.mark.parametrize("name", NAMES)
def test_native_graph_stage_reproduces_golden(lfr_lib, name):
    _, z = load(name)
    g = capi.Graph.from_matches_file(os.path.join(GOLD, name + ".pb"))
    t, r, c = capi.Problem(g).labels()
    assert (t == z["track"]).all() and (r == z["is_root"]).all() and (c == z["comp"]).all()
    img, feat = g.nodes()
    names = g.image_names()
    assert [names[i] for i in img] == list(z["node_image"]) and (feat == z["node_feat"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("variant", ["ceres1", "ceres2"])
def test_hip_reproduces_golden(lfr_lib, name, variant):
    _, z = load(name)
    g = capi.Graph.from_matches_file(os.path.join(GOLD, name + ".pb"))
    p = capi.Problem(g)
    b = capi.Batch(p, 0, tukey_variant=variant)
    st = b.solve()
    pos = b.download()
    assert st["n_failed"] == 0
    err = np.abs(pos - z["positions_" + variant]).max()
    assert err <= TOL_UNITS, err
    info = b.component_info()
    assert (info["iterations"] == z["iterations_" + variant][info["component"]]).all()
    assert (info["termination"] == z["termination_" + variant][info["component"]]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_solve_cli_end_to_end(lfr_lib, name, tmp_path):
    """The drop-in boundary: `solve --matches_file M --output_file S` (benchmark.py:99-104)."""
    out = str(tmp_path / "solution.pb")
    solve = os.path.join(os.path.dirname(HERE), "multi-view-refinement", "build", "solve")
    r = subprocess.run([solve, "--matches_file", os.path.join(GOLD, name + ".pb"), "--output_file", out],
                       capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    prefixes = ["# graph nodes: ", "# graph edges: ", "# tracks: ", "max track size: ", "Graph-cut time: ",
                "# components: ", "max component size: ", "Solver time: ", "Total time: ",
                "# points with at least one coordinate > 0.5: "]
    assert len(lines) == len(prefixes) and all(l.startswith(p) for l, p in zip(lines, prefixes)), r.stdout
    got = wire.decode_solution_file(open(out, "rb").read())
    want = wire.decode_solution_file(open(os.path.join(GOLD, name + ".solution.pb"), "rb").read())
    assert [(im["image_name"], im["fact"], [d[0] for d in im["displacements"]]) for im in got] == \
           [(im["image_name"], im["fact"], [d[0] for d in im["displacements"]]) for im in want]
    a = np.array([d[1:] for im in got for d in im["displacements"]])
    b = np.array([d[1:] for im in want for d in im["displacements"]])
    assert np.abs(a - b).max() <= TOL_UNITS

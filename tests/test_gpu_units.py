"""Unit-level GPU parity: the kernels' per-edge arithmetic - biquadratic interpolant with the
clamp / zero-partial rule (cost.cc:13-48), residual and Jacobian (cost.cc:78-90), Cauchy / Tukey losses and the
corrector (solve.cc:111,120 + Ceres) - against the C oracle's eval_edge at 1e-12, edge by edge; and the BASELINE
configurations 2 and 5 at FULL size against the oracle end to end."""
import os

import numpy as np
import pytest

import lfr_oracle as O
from lfr_amd import capi, synthetic

pytestmark = pytest.mark.gpu
TOL_UNITS = 6.25e-6          # 1e-4 px at fact = 1 (colmap_utils.py:135-136)


def _edges(rng, n):
    flows = rng.normal(0, 0.2, (n, 18)).astype(np.float32)
    sim = rng.uniform(0.05, 1.0, n).astype(np.float32)
    kind = rng.integers(0, 2, n).astype(np.int32)
    x1 = rng.uniform(-1, 1, (n, 2))                     # half of these lie outside the +-0.5 patch: clamped, zero partials
    x2 = rng.uniform(-1, 1, (n, 2))
    # exactly on the patch border (the partial is RETAINED there, cost.cc:38,41), exactly at grid nodes, at the bounds
    x1[0::17, 0] = 0.5
    x1[1::17, 1] = -0.5
    x1[2::17] = 0.0
    x1[3::17] = [1.0, -1.0]
    # Tukey inliers: residual norm^2 <= 0.0625^2 needs x2 ~ x1 + flow(x1); make a third of the edges near-consistent
    near = np.arange(n) % 3 == 0
    flows[near] *= 0.01
    x2[near] = x1[near] + rng.normal(0, 0.02, (int(near.sum()), 2))
    return flows, sim, kind, x1, x2


@pytest.mark.parametrize("variant", ["ceres1", "ceres2"])
def test_eval_edge_matches_oracle(lfr_lib, variant):
    rng = np.random.default_rng(20260927)
    n = 6000
    flows, sim, kind, x1, x2 = _edges(rng, n)
    out, cost_only = capi.eval_edges_hip(flows, sim, kind, x1, x2, variant)
    n_tukey_in = n_clamped = 0
    worst = 0.0
    for i in range(n):
        c, r, J, sq = O.eval_edge(flows[i], float(sim[i]), int(kind[i]), x1[i], x2[i], variant)
        want = np.array([c, r[0], r[1], J[0, 0], J[0, 1], J[1, 0], J[1, 1], sq])
        err = np.abs(out[i] - want).max()
        worst = max(worst, err, abs(cost_only[i] - c))
        n_tukey_in += int(kind[i] == 1 and sq > 0)
        n_clamped += int((np.abs(x1[i]) > 0.5).any())
    assert worst <= 1e-12, worst
    assert n_tukey_in > 300 and n_clamped > 1000                 # the interesting branches were exercised


def test_eval_edge_known_answers(lfr_lib):
    """Interpolator identities through the GPU path: at the grid nodes the flow is reproduced exactly; outside the
    patch the value is that of the clamped point and the partials vanish (cost.cc:17-18,38-43)."""
    rng = np.random.default_rng(7)
    grid = rng.normal(0, 0.2, (3, 3, 2)).astype(np.float32)
    flows = np.repeat(grid.reshape(1, 18), 11, axis=0)
    nodes = [(-0.5 + 0.5 * i, -0.5 + 0.5 * j) for i in range(3) for j in range(3)]
    x1 = np.array(nodes + [(0.9, 0.2), (0.5, 0.2)])
    x2 = np.zeros_like(x1)
    out, _ = capi.eval_edges_hip(flows, np.ones(11, np.float32), np.zeros(11, np.int32), x1, x2)
    for k, (i, j) in enumerate((i, j) for i in range(3) for j in range(3)):
        r_uncorrected = (x2[k] - x1[k] - grid[i, j].astype(np.float64))
        assert np.allclose(out[k, 1:3] / out[k, 7], r_uncorrected, rtol=0, atol=1e-15)
    # row 0.9 is clamped to 0.5: same interpolated flow as at row 0.5, d/drow = 0 -> J1[:,0] = -sq * (1, 0)
    f_clamped = x2[9] - x1[9] - out[9, 1:3] / out[9, 7]
    f_border = x2[10] - x1[10] - out[10, 1:3] / out[10, 7]
    assert np.allclose(f_clamped, f_border, atol=1e-15)
    assert out[9, 3] == -out[9, 7] and out[9, 5] == 0.0
    assert out[10, 3] != -out[10, 7]                             # at exactly 0.5 the partial is retained


@pytest.mark.parametrize("name", ["config2", "config5"])
def test_baseline_configs_at_full_size(lfr_lib, name):
    """BASELINE.json configs[1] (100 k tracks, 64 images, ~3.4 M edges) and the config-5 stand-in (long tracks,
    > 64-node components, ~9 M edges, workgroup kernels) at their full size: device pipeline vs the C oracle."""
    ma = getattr(synthetic, name)()
    g = capi.Graph.from_arrays(ma)
    p = capi.Problem(g, device_graph_stage=0)
    b = capi.Batch(p, 0)
    st = b.solve()
    pos = b.download()
    import ctypes
    bis = ctypes.cast(capi.lib().lfr_bisect_graph, ctypes.c_void_p).value
    ref = O.run(ma, n_threads=min(64, os.cpu_count() or 1), bisect=bis)
    assert ref["rc"] == 0
    t, r, c = p.labels()
    assert (t == ref["track"]).all() and (r == ref["is_root"]).all() and (c == ref["comp"]).all()
    assert st["n_failed"] == int((ref["infos"]["termination"][ref["comp_nvar"] > 0] == 2).sum())
    err = np.abs(pos - ref["positions"]).max()
    assert err <= TOL_UNITS, err
    info = b.component_info()
    solved = ref["comp_nvar"] > 0
    assert (np.sort(info["component"]) == np.nonzero(solved)[0]).all()
    assert (info["iterations"] == ref["infos"]["iterations"][info["component"]]).all()


@pytest.mark.parametrize("register_version", [0, 1, 2])     # loop version / register version / wave-cooperative version
def test_line_search_contraction_matches_numpy_roots(lfr_lib, register_version):
    """ArmijoLineSearch::DoSearch's step contraction (MinimizeInterpolatingPolynomial over 3..6 value / gradient constraints)
    as the kernels compute it (the packed kernel's loop version and the workgroup kernel's register version), against the numpy restatement (np.roots), including interpolants whose leading coefficients vanish."""
    import ls_cases
    S, dir_max, want = ls_cases.make(4000)
    got = capi.ls_next_step_hip(S, dir_max, register_version)
    if register_version == 2:                                   # same arithmetic per piece as the register version: the same bits
        assert (got == capi.ls_next_step_hip(S, dir_max, 1)).all()
    gave_up = want < 0
    assert ((got < 0) == gave_up).all()
    ok = ~gave_up
    rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
    # where the abscissae differ the two candidates must be a tie in VALUE (a flat interpolant): compare the interpolant there
    for k in np.flatnonzero(ok)[rel > 1e-8]:
        v = ls_cases.interpolant_values(S[k], np.array([got[k], want[k]]))
        assert v[0] <= v[1] + 1e-9 * max(1.0, abs(v[1])), (k, got[k], want[k], v)
    assert (rel > 1e-8).sum() <= 0.05 * rel.size, (rel > 1e-8).sum()
    ncons = 2 + S[:, 2, 3] + S[:, 2, 4] + S[:, 1, 3] + S[:, 1, 4]
    assert set(np.unique(ncons[S[:, 2, 3] > 0]).astype(int)) == {3, 4, 5, 6}

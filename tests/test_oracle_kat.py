"""Known-answer tests that pin the oracle (oracle/lfr_ref.py and oracle/lfr_oracle.c).

The reference has no tests or golden files and its arithmetic lives in un-vendored Ceres, so the
pins are derived analytically from the code being restated:
interpolator identities (cost.cc:13-48), closed-form losses, the two-node LM case whose answer
is c/(1+1e-4) (Ceres stops early: function/parameter tolerance 1e-4, solve.cc:152-154), polynomial
minimisation of the Armijo search, an objective-level cross-check with scipy, and the C
restatement against the readable Python one, trace for trace.
"""
import math

import numpy as np
import pytest

import lfr_oracle as O
import lfr_ref as R
from lfr_amd import synthetic

GRID = [(-0.5 + 0.5 * i, -0.5 + 0.5 * j) for i in range(3) for j in range(3)]


def both_interp(flow, r, c):
    a = R.interpolate(flow, r, c)
    b = O.interpolate(flow, r, c)
    np.testing.assert_allclose(np.concatenate(a), np.concatenate(b), rtol=0, atol=1e-15)
    return [np.asarray(v) for v in a]


def test_interpolator_reproduces_grid_nodes():
    rng = np.random.default_rng(1)
    flow = rng.normal(size=18).astype(np.float32)
    for k, (r, c) in enumerate(GRID):
        f, _, _ = both_interp(flow, r, c)
        np.testing.assert_allclose(f, flow[2 * k:2 * k + 2], atol=1e-15)


def test_interpolator_constant_and_linear_fields():
    const = np.tile(np.float32([0.3, -0.2]), 9)
    f, dr, dc = both_interp(const, 0.123, -0.321)
    np.testing.assert_allclose(f, [np.float32(0.3), np.float32(-0.2)], atol=1e-15)
    np.testing.assert_allclose(np.r_[dr, dc], 0, atol=1e-15)
    A = np.array([[0.5, -0.25], [0.125, 0.75]])
    b = np.array([0.0625, -0.5])
    lin = np.concatenate([A @ np.array(g) + b for g in GRID]).astype(np.float32)     # exactly representable
    f, dr, dc = both_interp(lin, 0.2, -0.4)
    np.testing.assert_allclose(f, A @ [0.2, -0.4] + b, atol=1e-15)
    np.testing.assert_allclose(dr, A[:, 0], atol=1e-15)
    np.testing.assert_allclose(dc, A[:, 1], atol=1e-15)


def test_interpolator_clamping_rules():
    rng = np.random.default_rng(2)
    flow = rng.normal(size=18).astype(np.float32)
    f_in, dr_in, dc_in = both_interp(flow, 0.5, 0.1)          # exactly on the border: derivative kept (cost.cc:38)
    f_out, dr_out, dc_out = both_interp(flow, 0.7, 0.1)       # outside: value of the border, zero row-derivative
    np.testing.assert_allclose(f_out, f_in, atol=0)
    assert np.all(dr_out == 0) and np.any(dr_in != 0)
    np.testing.assert_allclose(dc_out, dc_in, atol=0)
    _, dr2, dc2 = both_interp(flow, -0.9, 0.95)
    assert np.all(dr2 == 0) and np.all(dc2 == 0)


def test_interpolator_against_independent_lagrange_form():
    rng = np.random.default_rng(3)
    nodes = np.array([-0.5, 0.0, 0.5])

    def basis(t):
        return np.array([np.prod([(t - nodes[m]) / (nodes[k] - nodes[m]) for m in range(3) if m != k]) for k in range(3)])

    for _ in range(20):
        flow = rng.normal(size=18).astype(np.float32)
        r, c = rng.uniform(-0.5, 0.5, size=2)
        f, dr, dc = both_interp(flow, r, c)
        D = flow.astype(float).reshape(3, 3, 2)
        np.testing.assert_allclose(f, np.einsum("i,j,ijk->k", basis(r), basis(c), D), atol=1e-14)
        h = 1e-6
        fd = (np.einsum("i,j,ijk->k", basis(r + h), basis(c), D) - np.einsum("i,j,ijk->k", basis(r - h), basis(c), D)) / (2 * h)
        np.testing.assert_allclose(dr, fd, atol=1e-8)


def test_loss_closed_forms():
    for s in (0.0, 1e-3, 0.0625, 0.5, 10.0):
        w = 0.875
        rho = O.loss(R.KIND_INTRA, s, w)
        np.testing.assert_allclose(rho, [w * 0.0625 * math.log1p(16 * s), w / (1 + 16 * s), -w * 16 / (1 + 16 * s) ** 2], rtol=1e-14)
        np.testing.assert_allclose(R.scaled_loss(R.KIND_INTRA, s, w), rho, rtol=1e-15)
    a2 = 0.0625 ** 2
    for variant, k in (("ceres1", 1.0), ("ceres2", 2.0)):
        for s in (0.0, a2 / 2, a2, 2 * a2):
            rho = O.loss(R.KIND_INTER, s, 1.0, variant)
            np.testing.assert_allclose(R.scaled_loss(R.KIND_INTER, s, 1.0, variant), rho, rtol=1e-15)
            if s <= a2:
                v = 1 - s / a2
                np.testing.assert_allclose(rho, [k * a2 / 6 * (1 - v ** 3), k * 0.5 * v * v, -k * v / a2], rtol=1e-14, atol=1e-18)
            else:
                np.testing.assert_allclose(rho, [k * a2 / 6, 0, 0], rtol=1e-14)
    # the version switch is a factor 2 on every inter-track edge weight
    np.testing.assert_allclose(O.loss(1, a2 / 3, 1.0, "ceres2"), 2 * O.loss(1, a2 / 3, 1.0, "ceres1"), rtol=1e-15)


def test_edge_jacobian_is_the_derivative_of_the_residual():
    rng = np.random.default_rng(4)
    flow = (0.3 * rng.normal(size=18)).astype(np.float32)
    x1, x2 = np.array([0.1, -0.2]), np.array([0.25, 0.05])
    # with weight 1 and tiny residuals the Cauchy correction is ~1: compare uncorrected pieces
    c, r, J1, j2 = O.eval_edge(flow, 1.0, 0, x1, x2)
    f, dr, dc = R.interpolate(flow, *x1)
    r_raw = x2 - x1 - np.array(f)
    s = float(r_raw @ r_raw)
    sq = math.sqrt(1.0 / (1.0 + 16.0 * s))
    np.testing.assert_allclose(r, r_raw * sq, rtol=1e-14)
    np.testing.assert_allclose(J1, sq * (-np.eye(2) - np.column_stack([dr, dc])), rtol=1e-14)
    assert j2 == pytest.approx(sq, rel=1e-15)
    assert c == pytest.approx(0.5 * 0.0625 * math.log1p(16 * s), rel=1e-14)


def two_node_pairs(c):
    f_fwd = [(c[0], c[1])] * 9           # root -> x : flow  c  (rides node1->node2 = disp2)
    f_bwd = [(-c[0], -c[1])] * 9         # x -> root : flow -c
    return [{"image_name1": "a.png", "fact1": 1.0, "image_name2": "b.png", "fact2": 1.0,
             "matches": [{"feature_idx1": 0, "feature_idx2": 0, "similarity": 0.9, "disp1": f_bwd, "disp2": f_fwd}]}]


def test_two_node_known_answer_is_c_over_1_plus_1e_minus_4():
    """one accepted LM step with radius 1e4, then the parameter/function
    tolerance stops the solve and discards the (better) second candidate."""
    c = np.array([np.float32(0.3), np.float32(-0.125)], float)
    res = R.solve_pairs(two_node_pairs(c), want_trace=True)
    ma = synthetic.pairs_to_arrays(two_node_pairs(c))
    o = O.run(ma)
    roots = np.asarray(res["is_root"])
    assert roots.sum() == 1
    # the root is the node with the larger index on a score tie (solve.cc:567-568): node 1
    assert roots[1] and not roots[0]
    x = res["positions"][0]
    # node 0 is the variable; edge node0->node1 carries +c so r = x_root - x - c  => x* = -c
    np.testing.assert_allclose(x, -c / (1 + 1e-4), rtol=1e-13)
    info = list(res["infos"].values())[0]
    assert info["iterations"] == 2 and info["n_successful"] == 1
    np.testing.assert_allclose(o["positions"], res["positions"], atol=1e-16)
    assert o["infos"]["iterations"].max() == 2
    # the gap to the true minimiser (-c) is above the 1e-4 px bar: trajectory parity matters
    assert np.abs(x + c).max() > 6.25e-6


def test_polynomial_minimisation_matches_numpy_roots():
    rng = np.random.default_rng(5)
    for _ in range(200):
        f0, g0 = rng.uniform(0.5, 2), -rng.uniform(0.1, 3)
        a1 = rng.uniform(0.2, 1.0)
        s = [(0.0, f0, g0), (a1, f0 + rng.uniform(0, 2), rng.uniform(-1, 4))]
        if rng.random() < 0.5:
            s.append((a1 / rng.uniform(0.1, 0.55), f0 + rng.uniform(0, 5), rng.uniform(-1, 6)))
        lo, hi = 1e-3 * a1, 0.6 * a1
        want = R.minimize_interpolating_polynomial(s, lo, hi)
        got = O.minimize_poly([[x, v, g, 1, 1] for (x, v, g) in s], lo, hi)
        assert got == pytest.approx(want, rel=1e-7, abs=1e-10)
    # cubic through (0, 1, -1), (1, 1, 1): symmetric parabola-like, minimum at 0.5
    assert O.minimize_poly([[0, 1, -1, 1, 1], [1, 1, 1, 1, 1]], 1e-3, 0.6) == pytest.approx(0.5, abs=1e-12)


def test_polynomial_minimisation_with_vanishing_leading_coefficients():
    """Samples of exact quadratics / cubics / quartics: the degree-5 interpolant's derivative has roots near infinity
    (the first version of the oracle, a simultaneous complex iteration from a coefficient bound, lost the small roots there)."""
    import ls_cases
    S, dir_max, want = ls_cases.make(1500, seed=12)
    n_diff = 0
    for k in range(S.shape[0]):
        if not S[k, 2, 3]:
            continue
        xc = S[k, 2, 0]
        lo, hi = R.LS_MAX_STEP_CONTRACTION * xc, R.LS_MIN_STEP_CONTRACTION * xc
        rows = [S[k, 0], S[k, 2]] + ([S[k, 1]] if S[k, 1, 3] else [])
        got = O.minimize_poly(rows, lo, hi)
        ref = R.minimize_interpolating_polynomial(ls_cases.reference_samples(S[k]), lo, hi)
        if abs(got - ref) > 1e-8 * abs(ref):                      # must be a tie in value (or np.roots lost accuracy: better)
            v = ls_cases.interpolant_values(S[k], np.array([got, ref]))
            assert v[0] <= v[1] + 1e-9 * max(1.0, abs(v[1])), (k, got, ref, v)
            n_diff += 1
    assert n_diff <= 0.05 * S.shape[0], n_diff


CASES = {
    "clean": dict(seed=41, n_images=30, n_tracks=60),
    "outliers": dict(seed=42, n_images=200, n_tracks=60, eps_out=0.03),
    "noisy": dict(seed=43, n_images=24, n_tracks=40, sigma_noise=0.25),
    "bounds": dict(seed=44, n_images=24, n_tracks=40, sigma_p=0.7, sigma_noise=0.15),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("variant", ["ceres1", "ceres2"])
def test_c_oracle_equals_python_reference(name, variant):
    ma = synthetic.generate(**CASES[name])
    res = R.solve_pairs(ma.to_pairs(), tukey_variant=variant, want_trace=True)
    worst = max(res["infos"], key=lambda c: res["infos"][c]["iterations"])
    o = O.run(ma, n_threads=2, tukey_variant=variant, trace_comp=worst)
    assert o["rc"] == 0
    for k in ("n_nodes", "n_edges", "n_tracks", "max_track_size", "n_components", "max_component_size"):
        assert res[k] == o[k], k
    assert (np.asarray(res["track"]) == o["track"]).all()
    assert (np.asarray(res["is_root"]) == o["is_root"]).all()
    assert (np.asarray(res["comp"]) == o["comp"]).all()
    np.testing.assert_allclose(o["positions"], res["positions"], rtol=0, atol=1e-12)
    for c, info in res["infos"].items():
        oi = o["infos"][c]
        assert (oi["iterations"], oi["termination"], oi["n_successful"], oi["n_ls_evals"]) == \
               (info["iterations"], info["termination"], info["n_successful"], info["n_ls_evals"]), c
        assert oi["n_jac_evals"] == info["n_jac_evals"] and oi["n_cost_evals"] == info["n_cost_evals"]
        assert oi["final_cost"] == pytest.approx(info["final_cost"], rel=1e-11)
    # iteration-by-iteration trace of the hardest component
    tr = [t for t in res["infos"][worst]["trace"] if "stop" not in t and not t.get("invalid")]
    ct = [row for row in o["trace"] if int(row[7]) in (1, 2)]
    assert len(tr) == len(ct)
    for t, row in zip(tr, ct):
        assert int(row[0]) == t["it"]
        assert row[1] == pytest.approx(t["cost"], rel=1e-10)
        assert row[4] == pytest.approx(t["radius"], rel=1e-9)


def test_converged_cost_close_to_true_minimum_scipy():
    """Objective-level sanity (not trajectory): the robust cost at the returned point is within
    the loose Ceres tolerances of the minimum found by an independent bounded optimiser."""
    from scipy.optimize import minimize
    ma = synthetic.generate(seed=45, n_images=30, n_tracks=25, eps_out=0.0)
    pairs = ma.to_pairs()
    res = R.solve_pairs(pairs)
    g = R.MatchGraph(pairs)
    nodes_in = {}
    for n, c in enumerate(res["comp"]):
        nodes_in.setdefault(c, []).append(n)
    checked = 0
    for c, nodes in nodes_in.items():
        if len(nodes) < 2:
            continue
        var_nodes, edges = R.assemble_component(g, res["track"], res["is_root"], res["comp"], nodes)
        prob = R.Problem(len(var_nodes), edges)
        x_ref = np.concatenate([res["positions"][n] for n in var_nodes])
        sol = minimize(lambda x: prob.evaluate(x, False)[0], x_ref, method="L-BFGS-B",
                       bounds=[(-1, 1)] * x_ref.size, options={"ftol": 1e-15, "gtol": 1e-12})
        c_ref = prob.evaluate(x_ref, False)[0]
        assert c_ref >= sol.fun - 1e-12
        assert c_ref - sol.fun <= 2e-3 * max(sol.fun, 1e-9), (c_ref, sol.fun)
        checked += 1
    assert checked >= 20

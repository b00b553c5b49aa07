"""Line-search contraction cases shared by the oracle (CPU) and kernel (GPU) tests: samples of smooth functions that are
EXACTLY low-degree polynomials, so that the interpolant Ceres fits (degree = #constraints - 1, up to 5) has vanishing leading
coefficients and its derivative a root near infinity - the configuration a root finder started from a coefficient bound
gets wrong - next to generic ones.  Expected values: the numpy restatement (np.roots, like Ceres' companion-matrix eigenvalues)."""
import numpy as np

import lfr_ref as R


def make(n, seed=11):
    rng = np.random.default_rng(seed)
    S = np.zeros((n, 3, 5))
    dir_max = rng.uniform(1e-3, 1.0, n)
    want = np.zeros(n)
    for i in range(n):
        f0 = rng.uniform(0.1, 10.0); g0 = -rng.uniform(0.01, 5.0)
        xc = rng.choice([1.0, 0.6, 0.3, 0.05, 1e-3])
        xp = xc / rng.uniform(0.1, 0.6) if rng.random() < 0.7 else 0.0
        shape = rng.uniform(0.5, 50.0)
        k3 = rng.normal(0, 5.0) if rng.random() < 0.7 else 0.0                 # cubic, or exactly quadratic
        k4 = rng.normal(0, 20.0) if rng.random() < 0.3 else 0.0
        wig = rng.uniform(0.0, 0.5) if rng.random() < 0.3 else 0.0             # a non-polynomial part
        phi = lambda a: f0 + g0 * a + shape * a * a + k3 * a ** 3 + k4 * a ** 4 + wig * (np.cos(7 * a) - 1.0)
        dphi = lambda a: g0 + 2 * shape * a + 3 * k3 * a * a + 4 * k4 * a ** 3 - 7 * wig * np.sin(7 * a)
        cur_valid = rng.random() < 0.95
        cur_grad = cur_valid and rng.random() < 0.8
        prev_valid = xp > 0 and rng.random() < 0.85
        prev_grad = prev_valid and rng.random() < 0.8
        S[i, 0] = [0.0, f0, g0, 1, 1]
        S[i, 1] = [xp, phi(xp) if prev_valid else 0.0, dphi(xp) if prev_grad else 0.0, prev_valid, prev_grad]
        S[i, 2] = [xc, phi(xc) if cur_valid else 0.0, dphi(xc) if cur_grad else 0.0, cur_valid, cur_grad]
        lo, hi = R.LS_MAX_STEP_CONTRACTION * xc, R.LS_MIN_STEP_CONTRACTION * xc
        if not cur_valid:
            step = min(max(xc * 0.5, lo), hi)
        else:
            step = R.minimize_interpolating_polynomial(reference_samples(S[i]), lo, hi)
        want[i] = step if step * dir_max[i] >= R.LS_MIN_STEP_SIZE else -1.0
    return S, dir_max, want


def reference_samples(s):
    """(x, value, gradient or None) in Ceres' order initial, current, previous (line_search.cc ArmijoLineSearch::DoSearch)"""
    out = [(s[0, 0], s[0, 1], s[0, 2]), (s[2, 0], s[2, 1], s[2, 2] if s[2, 4] else None)]
    if s[1, 3]:
        out.append((s[1, 0], s[1, 1], s[1, 2] if s[1, 4] else None))
    return out


def interpolant_values(s, xs):
    """value of the interpolating polynomial of the samples at xs (numpy solve: judges near-ties by VALUE, not by abscissa)"""
    samples = reference_samples(s)
    ncons = sum(1 + (q[2] is not None) for q in samples)
    deg = ncons - 1
    A = []; b = []
    for (x, v, g) in samples:
        A.append([x ** (deg - j) for j in range(deg + 1)]); b.append(v)
        if g is not None:
            A.append([(deg - j) * x ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)]); b.append(g)
    p = np.linalg.solve(np.array(A), np.array(b))
    return np.polyval(p, xs)

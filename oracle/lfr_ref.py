"""TEST INFRASTRUCTURE — readable CPU restatement of the reference's multi-view solver path.

    *** parity unpinned ***  

The arithmetic of the path lives in third-party code that is neither vendored nor
version-pinned by the reference (``multi-view-refinement/CMakeLists.txt:9,18``):
Ceres Solver (Levenberg-Marquardt, loss correction, bounds/line search) and
COLMAP->Graclus (normalized cut).  None of them exist in this image, and the
reference ships no tests or golden files, so this restatement cannot be checked
against the real thing here.  It follows

  * ``multi-view-refinement/solve.cc:426-670`` literally for the graph / track /
    root / component / assembly / output semantics,
  * ``multi-view-refinement/cost.cc:13-48,78-90`` literally for the interpolator and
    the residual,
  * upstream Ceres Solver's published algorithm (trust_region_minimizer.cc,
    levenberg_marquardt_strategy.cc, line_search.cc, polynomial.cc, loss_function.cc,
    corrector.cc, residual_block.cc of the 1.12-2.x series, restated from their
    documented behaviour) for everything ``ceres::Solve`` does at
    ``solve.cc:146-159``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  It is pure Python + numpy, written for clarity and used on
small cases; ``oracle/lfr_oracle.c`` is the fast C restatement of the same thing
and is cross-checked against this file.
"""
import math

import numpy as np

# ---------------------------------------------------------------------------
# constants of the numerical contract
# ---------------------------------------------------------------------------
BOUND = 1.0                      # solve.cc:89
CAUCHY_A = 0.25                  # solve.cc:111
TUKEY_A = 0.0625                 # solve.cc:120
MAX_NUM_ITERATIONS = 100         # solve.cc:149
MAX_CONSECUTIVE_INVALID = 10     # solve.cc:151
FUNCTION_TOLERANCE = 1e-4        # solve.cc:152
GRADIENT_TOLERANCE = 1e-8        # solve.cc:153
PARAMETER_TOLERANCE = 1e-4       # solve.cc:154
# Ceres defaults (Solver::Options) that the reference does not override
INITIAL_RADIUS = 1e4
MAX_RADIUS = 1e16
MIN_RADIUS = 1e-32
MIN_RELATIVE_DECREASE = 1e-3
MIN_LM_DIAGONAL = 1e-6
MAX_LM_DIAGONAL = 1e32
LS_SUFFICIENT_DECREASE = 1e-4
LS_MAX_STEP_CONTRACTION = 1e-3
LS_MIN_STEP_CONTRACTION = 0.6
LS_MAX_ITERATIONS = 20
LS_MIN_STEP_SIZE = 1e-9
DBL_MIN = 2.2250738585072014e-308
DBL_MAX = 1.7976931348623157e308

TERM_CONVERGENCE = 0
TERM_NO_CONVERGENCE = 1
TERM_FAILURE = 2

KIND_INTRA = 0    # same track      -> ScaledLoss(CauchyLoss(0.25), sim)   solve.cc:105-113
KIND_INTER = 1    # same component  -> ScaledLoss(TukeyLoss(0.0625), sim)  solve.cc:114-122


# ---------------------------------------------------------------------------
# A1: BiquadraticInterpolator::Evaluate (cost.cc:13-48)
# ---------------------------------------------------------------------------
def interpolate(flow, row, col, want_deriv=True):
    """flow: 18 values, index 2*(3*i+j)+k.  Returns (f[2], dfdrow[2], dfdcol[2])."""
    row_, col_ = row, col
    row = max(min(row, 0.5), -0.5)
    col = max(min(col, 0.5), -0.5)
    lr = (2.0 * row * (row - 0.5), (-4.0) * (row - 0.5) * (row + 0.5), 2.0 * row * (row + 0.5))
    dlr = (2.0 * row + 2.0 * (row - 0.5), (-4.0) * (row - 0.5) + (-4.0) * (row + 0.5),
           2.0 * row + 2.0 * (row + 0.5))
    lc = (2.0 * col * (col - 0.5), (-4.0) * (col - 0.5) * (col + 0.5), 2.0 * col * (col + 0.5))
    dlc = (2.0 * col + 2.0 * (col - 0.5), (-4.0) * (col - 0.5) + (-4.0) * (col + 0.5),
           2.0 * col + 2.0 * (col + 0.5))
    f = [0.0, 0.0]
    dr = [0.0, 0.0]
    dc = [0.0, 0.0]
    for k in range(2):
        for i in range(3):
            for j in range(3):
                d = float(flow[2 * (i * 3 + j) + k])
                f[k] += lr[i] * lc[j] * d
                if want_deriv:
                    if row_ == row:
                        dr[k] += dlr[i] * lc[j] * d
                    if col_ == col:
                        dc[k] += lr[i] * dlc[j] * d
    return f, dr, dc


# ---------------------------------------------------------------------------
# A8: loss functions (Ceres loss_function.cc)
# ---------------------------------------------------------------------------
def cauchy_loss(s, a=CAUCHY_A):
    b = a * a
    c = 1.0 / b
    sm = 1.0 + s * c
    inv = 1.0 / sm
    return b * math.log(sm), max(DBL_MIN, inv), -c * (inv * inv)


def tukey_loss(s, a=TUKEY_A, variant="ceres1"):
    a2 = a * a
    if s <= a2:
        v = 1.0 - s / a2
        v2 = v * v
        if variant == "ceres1":      # Ceres <= 1.14
            return a2 / 6.0 * (1.0 - v2 * v), 0.5 * v2, -1.0 / a2 * v
        return a2 / 3.0 * (1.0 - v2 * v), v2, -2.0 / a2 * v      # Ceres >= 2.0
    if variant == "ceres1":
        return a2 / 6.0, 0.0, 0.0
    return a2 / 3.0, 0.0, 0.0


def scaled_loss(kind, s, w, tukey_variant="ceres1"):
    r0, r1, r2 = cauchy_loss(s) if kind == KIND_INTRA else tukey_loss(s, variant=tukey_variant)
    return r0 * w, r1 * w, r2 * w


# ---------------------------------------------------------------------------
# A2 + A3(eval) : one residual block (cost.cc:78-90 + Ceres residual_block.cc/corrector.cc)
# ---------------------------------------------------------------------------
def eval_edge(flow, sim, kind, x1, x2, want_jac, tukey_variant="ceres1"):
    """Returns (cost, r[2] corrected, J1[2][2] corrected, j2 scalar (J2 = j2*I))."""
    f, dr, dc = interpolate(flow, x1[0], x1[1], want_jac)
    r0 = x2[0] - x1[0] - f[0]
    r1 = x2[1] - x1[1] - f[1]
    s = r0 * r0 + r1 * r1
    rho0, rho1, rho2 = scaled_loss(kind, s, sim, tukey_variant)
    cost = 0.5 * rho0
    # Corrector: rho'' <= 0 for both losses -> the simple branch sqrt(rho')
    assert s == 0.0 or rho2 <= 0.0
    sq = math.sqrt(rho1)
    J1 = None
    if want_jac:
        J1 = [[(-1.0 - dr[0]) * sq, (-dc[0]) * sq],
              [(-dr[1]) * sq, (-1.0 - dc[1]) * sq]]
    return cost, (r0 * sq, r1 * sq), J1, sq


class Problem:
    """The reduced Ceres program of one component (solve.cc:79-143).

    edges: list of (src, dst, sim, kind, flow18) with src/dst = index into the
    variable-node list, or -1 for a constant node (a track root, solve.cc:134-135,
    whose position stays (0,0)).  Residual blocks with both ends constant are not
    part of the reduced program (Ceres removes them; their cost is 'fixed cost').
    """

    def __init__(self, n_var_nodes, edges, tukey_variant="ceres1"):
        self.nv = n_var_nodes
        self.edges = [e for e in edges if not (e[0] < 0 and e[1] < 0)]
        self.tukey_variant = tukey_variant
        self.n_cost_evals = 0
        self.n_jac_evals = 0

    @staticmethod
    def project(x):
        return np.minimum(np.maximum(x, -BOUND), BOUND)

    def evaluate(self, x, want_jac):
        """Returns cost, r (2E), J (2E x 2nv dense), g = J^T r."""
        if want_jac:
            self.n_jac_evals += 1
        else:
            self.n_cost_evals += 1
        E = len(self.edges)
        cost = 0.0
        r = np.zeros(2 * E)
        J = np.zeros((2 * E, 2 * self.nv)) if want_jac else None
        for e, (src, dst, sim, kind, flow) in enumerate(self.edges):
            x1 = x[2 * src:2 * src + 2] if src >= 0 else (0.0, 0.0)
            x2 = x[2 * dst:2 * dst + 2] if dst >= 0 else (0.0, 0.0)
            c, re, J1, j2 = eval_edge(flow, sim, kind, x1, x2, want_jac, self.tukey_variant)
            cost += c
            r[2 * e], r[2 * e + 1] = re
            if want_jac:
                if src >= 0:
                    J[2 * e, 2 * src], J[2 * e, 2 * src + 1] = J1[0]
                    J[2 * e + 1, 2 * src], J[2 * e + 1, 2 * src + 1] = J1[1]
                if dst >= 0:
                    J[2 * e, 2 * dst] += j2
                    J[2 * e + 1, 2 * dst + 1] += j2
        g = J.T @ r if want_jac else None
        return cost, r, J, g


# ---------------------------------------------------------------------------
# Ceres polynomial.cc / line_search.cc (Armijo, cubic interpolation)
# ---------------------------------------------------------------------------
def _polyval(p, x):
    v = 0.0
    for c in p:
        v = v * x + c
    return v


def _poly_roots_real_parts(p):
    """Real parts of all roots (Ceres FindPolynomialRoots returns them and the
    caller tests every one of them, complex or not)."""
    p = list(p)
    while p and p[0] == 0.0:
        p.pop(0)
    deg = len(p) - 1
    if deg <= 0:
        return []
    if deg == 1:
        return [-p[1] / p[0]]
    if deg == 2:
        a, b, c = p
        D = b * b - 4 * a * c
        sD = math.sqrt(abs(D))
        if D >= 0:
            if b >= 0:
                return [(-b - sD) / (2.0 * a), (2.0 * c) / (-b - sD)]
            return [(2.0 * c) / (-b + sD), (-b + sD) / (2.0 * a)]
        return [-b / (2.0 * a), -b / (2.0 * a)]
    return [float(z.real) for z in np.roots(np.asarray(p, float))]


def minimize_interpolating_polynomial(samples, x_min, x_max):
    """samples: list of (x, value, gradient or None).  Ceres polynomial.cc."""
    ncons = sum(1 + (s[2] is not None) for s in samples)
    deg = ncons - 1
    lhs = np.zeros((ncons, ncons))
    rhs = np.zeros(ncons)
    row = 0
    for (x, v, gr) in samples:
        for j in range(deg + 1):
            lhs[row, j] = x ** (deg - j)
        rhs[row] = v
        row += 1
        if gr is not None:
            for j in range(deg):
                lhs[row, j] = (deg - j) * x ** (deg - j - 1)
            rhs[row] = gr
            row += 1
    poly = np.linalg.solve(lhs, rhs)
    best_x = (x_min + x_max) / 2.0
    best_v = _polyval(poly, best_x)
    for cand in (x_min, x_max):
        v = _polyval(poly, cand)
        if v < best_v:
            best_x, best_v = cand, v
    if len(poly) > 2:
        deriv = [poly[i] * (deg - i) for i in range(deg)]
        for root in _poly_roots_real_parts(deriv):
            if root < x_min or root > x_max:
                continue
            v = _polyval(poly, root)
            if v < best_v:
                best_x, best_v = root, v
    for (x, _, _) in samples:
        if x < x_min or x > x_max:
            continue
        v = _polyval(poly, x)
        if v < best_v:
            best_x, best_v = x, v
    return best_x


def armijo_line_search(problem, x, delta, cost0, g0_dot_delta):
    """Projected Armijo search of TrustRegionMinimizer::DoLineSearch.
    Returns (success, step_size, n_evals)."""
    dir_max = float(np.max(np.abs(delta))) if delta.size else 0.0
    initial = (0.0, cost0, g0_dot_delta)
    previous = None

    def _eval(alpha):
        xs = problem.project(x + alpha * delta)
        c, _, _, g = problem.evaluate(xs, True)      # CUBIC interpolation needs the gradient
        valid = math.isfinite(c)
        gd = float(delta @ g) if valid else None
        if gd is not None and not math.isfinite(gd):
            gd = None
        return (alpha, c if valid else None, gd)

    n_evals = 1
    current = _eval(1.0)
    n_iter = 0
    while current[1] is None or current[1] > cost0 + LS_SUFFICIENT_DECREASE * g0_dot_delta * current[0]:
        n_iter += 1
        if n_iter >= LS_MAX_ITERATIONS:
            return False, 0.0, n_evals
        lo, hi = LS_MAX_STEP_CONTRACTION * current[0], LS_MIN_STEP_CONTRACTION * current[0]
        if current[1] is None:
            step = min(max(current[0] * 0.5, lo), hi)
        else:
            samples = [initial, current]
            if previous is not None and previous[1] is not None:
                samples.append(previous)
            step = minimize_interpolating_polynomial(samples, lo, hi)
        if step * dir_max < LS_MIN_STEP_SIZE:
            return False, 0.0, n_evals
        previous = current
        n_evals += 1
        current = _eval(step)
    return True, current[0], n_evals


# ---------------------------------------------------------------------------
# A10: ceres::Solve — trust-region Levenberg-Marquardt with bounds
# ---------------------------------------------------------------------------
def solve_problem(problem, trace=None):
    """Returns (x, info).  x: 2*nv solution (all zeros when the solve FAILS: Ceres
    does not copy an unusable solution back to the user's parameter blocks)."""
    nv2 = 2 * problem.nv
    info = {"iterations": 0, "termination": TERM_CONVERGENCE, "final_cost": 0.0,
            "n_ls_evals": 0, "n_successful": 0}
    if nv2 == 0:
        return np.zeros(0), info          # "No non-constant parameter blocks found."
    x = problem.project(np.zeros(nv2))    # IterationZero: project onto the bounds
    x_norm = float(np.linalg.norm(x))
    cost, r, J, g = problem.evaluate(x, True)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))      # jacobi scaling, computed once
    Js = J * scale[None, :]

    def _gmax(x_, g_):
        return float(np.max(np.abs(x_ - problem.project(x_ - g_))))

    gmax = _gmax(x, g)
    best_x = x.copy()                     # x_cost < minimum_cost(=DBL_MAX) at iteration 0
    min_cost = cost
    radius = INITIAL_RADIUS
    decrease_factor = 2.0
    reuse_diagonal = False
    diagonal = None
    n_invalid = 0
    step_successful = True
    iteration = 0
    if trace is not None:
        trace.append({"it": 0, "cost": cost, "gmax": gmax, "radius": radius, "ok": True})

    while True:
        # FinalizeIterationAndCheckIfMinimizerCanContinue
        if iteration >= MAX_NUM_ITERATIONS:
            info["termination"] = TERM_NO_CONVERGENCE
            break
        if step_successful and gmax <= GRADIENT_TOLERANCE:
            info["termination"] = TERM_CONVERGENCE
            break
        if radius <= MIN_RADIUS:
            info["termination"] = TERM_CONVERGENCE
            break
        iteration += 1
        step_successful = False

        # LevenbergMarquardtStrategy::ComputeStep
        if not reuse_diagonal:
            diagonal = np.clip((Js * Js).sum(axis=0), MIN_LM_DIAGONAL, MAX_LM_DIAGONAL)
        D = np.sqrt(diagonal / radius)
        reuse_diagonal = True
        H = Js.T @ Js + np.diag(D * D)
        rhs = Js.T @ r
        valid = True
        try:
            L = np.linalg.cholesky(H)
            y = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            step = -y
            if not np.all(np.isfinite(step)):
                valid = False
        except np.linalg.LinAlgError:
            valid = False
        model_cost_change = 0.0
        if valid:
            mr = Js @ step
            model_cost_change = float(-mr @ (r + mr / 2.0))
            valid = model_cost_change > 0.0
        if not valid:
            n_invalid += 1
            if n_invalid >= MAX_CONSECUTIVE_INVALID:
                info["termination"] = TERM_FAILURE
                break
            radius = radius / decrease_factor      # StepIsInvalid -> StepRejected(0)
            decrease_factor *= 2.0
            reuse_diagonal = True
            if trace is not None:
                trace.append({"it": iteration, "invalid": True, "radius": radius})
            continue
        n_invalid = 0
        delta = step * scale

        # bounds-constrained problem: projected Armijo line search along delta
        ok, alpha, n_ev = armijo_line_search(problem, x, delta, cost, float(g @ delta))
        info["n_ls_evals"] += n_ev
        if ok:
            delta = delta * alpha

        x_cand = problem.project(x + delta)
        cost_cand, _, _, _ = problem.evaluate(x_cand, False)
        if not math.isfinite(cost_cand):
            cost_cand = DBL_MAX

        step_norm = float(np.linalg.norm(x - x_cand))
        if step_norm <= PARAMETER_TOLERANCE * (x_norm + PARAMETER_TOLERANCE):
            info["termination"] = TERM_CONVERGENCE
            if trace is not None:
                trace.append({"it": iteration, "stop": "parameter", "step_norm": step_norm})
            break
        cost_change = cost - cost_cand
        if abs(cost_change) <= FUNCTION_TOLERANCE * cost:
            info["termination"] = TERM_CONVERGENCE
            if trace is not None:
                trace.append({"it": iteration, "stop": "function", "cost_change": cost_change})
            break

        rel = cost_change / model_cost_change       # monotonic TrustRegionStepEvaluator
        if rel > MIN_RELATIVE_DECREASE:
            x = x_cand
            x_norm = float(np.linalg.norm(x))
            cost, r, J, g = problem.evaluate(x, True)
            Js = J * scale[None, :]
            gmax = _gmax(x, g)
            step_successful = True
            info["n_successful"] += 1
            radius = radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3)
            radius = min(MAX_RADIUS, radius)
            decrease_factor = 2.0
            reuse_diagonal = False
            if cost < min_cost:
                min_cost = cost
                best_x = x.copy()
        else:
            radius = radius / decrease_factor
            decrease_factor *= 2.0
            reuse_diagonal = True
        if trace is not None:
            trace.append({"it": iteration, "cost": cost, "cost_cand": cost_cand, "rel": rel,
                          "ok": step_successful, "radius": radius, "alpha": alpha if ok else None,
                          "gmax": gmax, "model_cost_change": model_cost_change})

    info["iterations"] = iteration
    info["final_cost"] = min_cost
    info["n_cost_evals"] = problem.n_cost_evals
    info["n_jac_evals"] = problem.n_jac_evals
    if info["termination"] == TERM_FAILURE:
        return np.zeros(nv2), info
    return best_x, info


# ---------------------------------------------------------------------------
# A0, A3-A7, A9, A11, A12: the graph stage of solve.cc, literally
# ---------------------------------------------------------------------------
class MatchGraph:
    """solve.cc:405-481 — nodes in order of first appearance, node1 before node2."""

    def __init__(self, pairs, banned=()):
        banned = set(banned)
        self.node_key = []            # (image_name, feature_idx)
        self.out_edges = []           # per node: list of (dst, sim, flow18)
        self.edges = []               # (sim, n1, n2) undirected, for Kruskal
        self.images_set = set()
        self.images_facts = {}
        lookup = {}

        def _node(img, f):
            k = (img, f)
            if k not in lookup:
                lookup[k] = len(self.node_key)
                self.node_key.append(k)
                self.out_edges.append([])
            return lookup[k]

        for p in pairs:
            n1, n2 = p["image_name1"], p["image_name2"]
            if n1 in banned or n2 in banned:
                continue
            self.images_set.add(n1)
            self.images_facts.setdefault(n1, p["fact1"])
            self.images_set.add(n2)
            self.images_facts.setdefault(n2, p["fact2"])
            for m in p["matches"]:
                if len(m["disp1"]) > 9 or len(m["disp2"]) > 9:
                    raise ValueError("more than 9 grid points (reference overflows its buffer)")
                f1 = np.zeros(18)
                for k, d in enumerate(m["disp1"]):
                    f1[2 * k], f1[2 * k + 1] = d
                f2 = np.zeros(18)
                for k, d in enumerate(m["disp2"]):
                    f2[2 * k], f2[2 * k + 1] = d
                a = _node(n1, m["feature_idx1"])
                b = _node(n2, m["feature_idx2"])
                sim = float(np.float32(m["similarity"]))
                self.edges.append((sim, a, b))
                self.out_edges[a].append((b, sim, f2))     # solve.cc:477
                self.out_edges[b].append((a, sim, f1))     # solve.cc:478
        self.n_nodes = len(self.node_key)


def build_tracks(g):
    """solve.cc:489-549: constrained maximum spanning forest."""
    edges = sorted(g.edges)
    edges.reverse()
    n = g.n_nodes
    parent = [-1] * n
    images_in_track = [{g.node_key[i][0]} for i in range(n)]

    def root(i):
        path = []
        while parent[i] != -1:
            path.append(i)
            i = parent[i]
        for p in path:
            parent[p] = i
        return i

    for (_, a, b) in edges:
        ra, rb = root(a), root(b)
        if ra == rb:
            continue
        if images_in_track[ra] & images_in_track[rb]:
            continue
        if len(images_in_track[ra]) < len(images_in_track[rb]):
            parent[ra] = rb
            images_in_track[rb] |= images_in_track[ra]
            images_in_track[ra] = set()
        else:
            parent[rb] = ra
            images_in_track[ra] |= images_in_track[rb]
            images_in_track[rb] = set()
    track = [-1] * n
    n_tracks = 0
    for i in range(n):
        if parent[i] == -1:
            track[i] = n_tracks
            n_tracks += 1
    for i in range(n):
        if track[i] == -1:
            track[i] = track[root(i)]
    return track, n_tracks


def select_roots(g, track, n_tracks):
    """solve.cc:552-582."""
    scores = []
    for i in range(g.n_nodes):
        s = 0.0
        for (dst, sim, _) in g.out_edges[i]:
            if track[i] == track[dst]:
                s += sim
        scores.append((s, i))
    scores.sort()
    scores.reverse()
    is_root = [False] * g.n_nodes
    has_root = [False] * n_tracks
    for (_, i) in scores:
        if has_root[track[i]]:
            continue
        is_root[i] = True
        has_root[track[i]] = True
    return is_root


def _bfs_components(n, adj):
    comp = [-1] * n
    nc = 0
    for s in range(n):
        if comp[s] != -1:
            continue
        comp[s] = nc
        queue = [s]
        while queue:
            u = queue.pop(0)
            for v in adj[u]:
                if comp[v] == -1:
                    comp[v] = nc
                    queue.append(v)
        nc += 1
    return comp, nc


def recursive_graph_cut(edges, weights, node_weights, max_subset_weight, bisect_fn):
    """solve.cc:185-250, literally, around ``bisect_fn(edges, weights) -> {node: 0/1}`` standing in for
    colmap::ComputeNormalizedMinGraphCut(edges, weights, 2) (solve.cc:192).  Returns {node: subset}.  The
    reference iterates std::unordered_maps here; their order only permutes the subset NUMBERS (labels), never the
    partition, and the caller re-labels by BFS (solve.cc:356-364)."""
    n_subsets = 2
    nodes_to_subsets = bisect_fn(edges, weights)
    subset_weights = [0] * n_subsets
    nodes_in_subset = [[] for _ in range(n_subsets)]
    for node in sorted(nodes_to_subsets):
        sub = nodes_to_subsets[node]
        subset_weights[sub] += node_weights[node]
        nodes_in_subset[sub].append(node)
    max_subset_idx = 0
    final = {}
    for subset_idx in range(n_subsets):
        if subset_weights[subset_idx] <= max_subset_weight:                       # solve.cc:205-211
            for node in nodes_in_subset[subset_idx]:
                final.setdefault(node, max_subset_idx)
            max_subset_idx += 1
            continue
        sub_e, sub_w = [], []
        for (a, b), w in zip(edges, weights):                                      # solve.cc:213-227
            if nodes_to_subsets[a] == subset_idx and nodes_to_subsets[b] == subset_idx:
                sub_e.append((a, b))
                sub_w.append(w)
        if sub_e:                                                                  # solve.cc:229-238
            sub = recursive_graph_cut(sub_e, sub_w, node_weights, max_subset_weight, bisect_fn)
            new_max = max_subset_idx
            for node, part in sub.items():
                final.setdefault(node, max_subset_idx + part)
                new_max = max(new_max, max_subset_idx + part)
            max_subset_idx = new_max + 1
        for node in nodes_in_subset[subset_idx]:                                   # solve.cc:240-246: orphans -> singletons
            if node in final:
                continue
            final[node] = max_subset_idx
            max_subset_idx += 1
    return final


def split_components(g, track, n_tracks, max_nodes, bisect_fn=None):
    """solve.cc:252-373 (separate_meta_graph).  Components above ``max_nodes`` go through recursive_graph_cut
    (solve.cc:311-343); its two-way cut is Graclus in the reference, which cannot be restated, so
    ``bisect_fn(edges, weights) -> {meta_node: 0/1}`` supplies the primitive (tests pass the product's own
    deterministic bisection, lfr_bisect_graph: everything AROUND the primitive - meta edges, integer weights,
    recursion, orphans, dropping cut edges, re-labelling - is then checked independently).  Returns
    (component per node, n_components, n_oversized)."""
    size = [0] * n_tracks
    for i in range(g.n_nodes):
        size[track[i]] += 1
    meta = [dict() for _ in range(n_tracks)]
    for i in range(g.n_nodes):
        for (dst, sim, _) in g.out_edges[i]:
            if track[i] != track[dst]:
                meta[track[i]][track[dst]] = meta[track[i]].get(track[dst], 0.0) + sim
    comp, nc = _bfs_components(n_tracks, meta)
    csize = [0] * nc
    members = [[] for _ in range(nc)]
    for t in range(n_tracks):
        csize[comp[t]] += size[t]
        members[comp[t]].append(t)
    gc = [0] * n_tracks
    ngc = 0
    n_over = 0
    for c in range(nc):
        if csize[c] <= max_nodes:                                                  # solve.cc:314 (== cap is not cut)
            for t in members[c]:
                gc[t] = ngc
            ngc += 1
            continue
        n_over += 1
        if bisect_fn is None:
            raise NotImplementedError("component above the size cap needs a graph cut")
        e, w = [], []
        for t in members[c]:
            for (u, s) in meta[t].items():
                if t < u:                                                          # solve.cc:325-331
                    e.append((t, u))
                    w.append(int(100 * s))
        split = recursive_graph_cut(e, w, size, max_nodes, bisect_fn)
        # solve.cc:335 asserts split.size() == #meta nodes: true whenever the cap is #images (a lone track never exceeds
        # it, and every track of a multi-track component has a meta edge)
        assert len(split) == len(members[c]) or not e
        top = 0
        for t, part in split.items():
            gc[t] = ngc + part
            top = max(top, gc[t])
        ngc = top + 1
    post = [dict() for _ in range(n_tracks)]
    for t in range(n_tracks):
        for (u, s) in meta[t].items():
            if gc[t] == gc[u]:                                                     # solve.cc:346-353
                post[t][u] = s
    fcomp, nfc = _bfs_components(n_tracks, post)
    return [fcomp[track[i]] for i in range(g.n_nodes)], nfc, n_over


def assemble_component(g, track, is_root, comp, nodes):
    """solve.cc:94-143: the Ceres problem of one component as a reduced program."""
    optimized = {}
    for n in nodes:
        for (dst, _, _) in g.out_edges[n]:
            if track[n] == track[dst] or comp[n] == comp[dst]:
                optimized[n] = True
    var_nodes = [n for n in nodes if optimized.get(n) and not is_root[n]]
    vidx = {n: k for k, n in enumerate(var_nodes)}
    edges = []
    for n in nodes:
        for (dst, sim, flow) in g.out_edges[n]:
            if track[n] == track[dst]:
                kind = KIND_INTRA
            elif comp[n] == comp[dst]:
                kind = KIND_INTER
            else:
                continue
            edges.append((vidx.get(n, -1), vidx.get(dst, -1), sim, kind, flow))
    return var_nodes, edges


def solve_pairs(pairs, banned=(), tukey_variant="ceres1", bisect_fn=None, want_trace=False):
    """The whole of solve.cc main() between parsing and serialisation.
    Returns dict with positions (n_nodes x 2), node keys, stats and per-component infos."""
    g = MatchGraph(pairs, banned)
    out = {"n_nodes": g.n_nodes, "n_edges": 2 * len(g.edges), "node_key": g.node_key,
           "images_facts": g.images_facts}
    positions = np.zeros((g.n_nodes, 2))
    if g.n_nodes == 0:
        out.update(positions=positions, n_tracks=0, n_components=0, infos={}, track=[], is_root=[],
                   comp=[])
        return out
    track, n_tracks = build_tracks(g)
    is_root = select_roots(g, track, n_tracks)
    comp, n_comp, n_over = split_components(g, track, n_tracks, len(g.images_set), bisect_fn)
    nodes_in = [[] for _ in range(n_comp)]
    for i in range(g.n_nodes):
        nodes_in[comp[i]].append(i)
    infos = {}
    for c in range(n_comp):
        if len(nodes_in[c]) == 1:
            continue                                     # solve.cc:619-622
        var_nodes, edges = assemble_component(g, track, is_root, comp, nodes_in[c])
        prob = Problem(len(var_nodes), edges, tukey_variant)
        tr = [] if want_trace else None
        x, info = solve_problem(prob, tr)
        if want_trace:
            info["trace"] = tr
        info["n_edges"] = len(prob.edges)
        info["n_var_nodes"] = len(var_nodes)
        infos[c] = info
        for k, n in enumerate(var_nodes):
            positions[n] = x[2 * k:2 * k + 2]
    sizes = [0] * n_tracks
    for t in track:
        sizes[t] += 1
    out.update(positions=positions, n_tracks=n_tracks, max_track_size=max(sizes),
               n_components=n_comp, max_component_size=max(len(v) for v in nodes_in),
               n_oversized=n_over, infos=infos, track=track, is_root=is_root, comp=comp)
    return out


def solution_images(result):
    """solve.cc:644-664: SolutionFile content (images in order of first node, float32)."""
    images, index = [], {}
    for n, (img, f) in enumerate(result["node_key"]):
        if img not in index:
            index[img] = len(images)
            images.append({"image_name": img, "fact": float(np.float32(result["images_facts"][img])),
                           "displacements": []})
        p = result["positions"][n]
        images[index[img]]["displacements"].append((f, float(np.float32(p[0])), float(np.float32(p[1]))))
    return images

"""Elimination-tree plans of the large components (csrc/lfr_treeplan.cpp; the reference: Ceres SPARSE_NORMAL_CHOLESKY, solve.cc:147).
The plan is executed on the CPU (tests/tree_plan_emul.py: the kernel's schedule, tile by tile) and compared with a dense solve."""
import numpy as np
import pytest

from lfr_amd import capi
from tree_plan_emul import Plan, dense_reference, NONE


def _words(a, b, kind):
    a = np.asarray(a, np.uint32); b = np.asarray(b, np.uint32); kind = np.asarray(kind, np.uint32)
    w = np.empty(2 * len(a), np.uint32)
    w[0::2] = a | ((b | (kind << 15)) << 16)          # the two directions of a match are neighbouring records
    w[1::2] = b | ((a | (kind << 15)) << 16)
    return w


def _by_source(w):
    """records in the batch's order: by source node (solve.cc:98-102), stable"""
    return w[np.argsort(w & 0xFFFF, kind="stable")]


def _run(n_var, w, seed=0, damping=1e-3):
    rng = np.random.default_rng(seed)
    blob, info = capi.tree_plan(n_var, w)
    pl = Plan(blob)
    lvl = pl.check(n_var)
    E = len(w)
    J1 = -np.eye(2)[None] - 0.3 * rng.standard_normal((E, 2, 2))
    sq = rng.uniform(0.5, 1.0, E)
    r = rng.standard_normal((E, 2))
    tiles, g, counted = pl.assemble(w, n_var, J1, sq, r)
    s, d = (w & 0xFFFF).astype(np.int64), ((w >> 16) & 0x7FFF).astype(np.int64)
    assert (counted[s != d] == 1).all()                                          # every record's cost is counted exactly once
    A, g_ref, pos = dense_reference(pl, w, n_var, J1, sq, r)
    np.testing.assert_allclose(g, g_ref, rtol=1e-12, atol=1e-12)
    # the tiles hold the lower triangle of A, nothing else
    full = np.zeros_like(A)
    for J in range(pl.NB):
        for t in range(pl.colptr[J], pl.colptr[J + 1]):
            I = pl.rowsof[t]
            full[16 * I:16 * I + 16, 16 * J:16 * J + 16] = tiles[t]
    np.testing.assert_allclose(full, np.tril(A), rtol=1e-12, atol=1e-12)
    # damped system (A + D^2) y = g
    real = np.repeat(pl.ipos != NONE, 2)
    Dd = damping * (1.0 + rng.random(pl.n_pad)) * real
    for J in range(pl.NB):
        tiles[pl.colptr[J]][np.arange(16), np.arange(16)] += Dd[16 * J:16 * J + 16]
    ft, w_out, inv = pl.factor(tiles, g)
    y = pl.back_substitute(ft, w_out, inv)
    M = A + np.diag(Dd)
    y_ref = np.linalg.solve(M[np.ix_(real, real)], g[real])
    assert np.abs(y[~real]).max(initial=0.0) == 0.0
    scale = max(1.0, np.abs(y_ref).max())
    assert np.abs(y[real] - y_ref).max() <= 1e-9 * scale * max(1.0, np.linalg.cond(M[np.ix_(real, real)]) * 1e-6)
    return pl, info, lvl


def _chain_of_tracks(T, L, rng, links=1, shuffle=True):
    n_var = T * L
    ids = rng.permutation(n_var) if shuffle else np.arange(n_var)
    a, b, k = [], [], []
    for t in range(T):
        for i in range(L):
            for j in range(i + 1, L):
                a.append(ids[t * L + i]); b.append(ids[t * L + j]); k.append(0)
        if t + 1 < T:
            for _ in range(links):
                a.append(ids[t * L + int(rng.integers(L))]); b.append(ids[(t + 1) * L + int(rng.integers(L))]); k.append(1)
    return n_var, _by_source(_words(a, b, k))


def test_chain_of_tracks_becomes_a_shallow_tree(lfr_lib):
    """200 five-node cliques chained by one inter-track match each: the postorder of round 3 was a path of ~60 dependent panels;
    nested dissection gives ~log2(200) levels with narrow fronts."""
    n_var, w = _chain_of_tracks(200, 5, np.random.default_rng(3))
    pl, info, lvl = _run(n_var, w)
    assert info["tracks"] == 200
    assert info["levels"] <= 14                                  # log2(200) = 7.6 levels of separators; a separator may take two blocks
    assert info["tiles"] <= 4 * info["blocks"]                   # diagonal + a few ancestors
    assert info["column_rounds"] <= info["blocks"] // 8 + 2 * info["levels"]


def test_random_tree_of_tracks_with_cycles(lfr_lib):
    rng = np.random.default_rng(5)
    T = 120
    sizes = rng.integers(1, 11, size=T)
    off = np.r_[0, np.cumsum(sizes)]
    n_var = int(off[-1])
    n_const = 7                                                  # constants (track roots) behind the variable nodes
    a, b, k = [], [], []
    for t in range(T):
        for i in range(sizes[t]):
            for j in range(i + 1, sizes[t]):
                if sizes[t] <= 5 or abs(i - j) <= 2:             # longer tracks are lattices
                    a.append(off[t] + i); b.append(off[t] + j); k.append(0)
        if t > 0:
            u = int(rng.integers(t))
            a.append(off[t] + int(rng.integers(sizes[t]))); b.append(off[u] + int(rng.integers(sizes[u]))); k.append(1)
        if t % 9 == 0:                                           # a root outside the variables
            a.append(off[t]); b.append(n_var + (t // 9) % n_const); k.append(0)
    for _ in range(10):                                          # cycles
        t, u = rng.integers(T, size=2)
        if t != u:
            a.append(off[t] + int(rng.integers(sizes[t]))); b.append(off[u] + int(rng.integers(sizes[u]))); k.append(1)
    w = _by_source(_words(a, b, k))
    pl, info, lvl = _run(n_var, w, seed=1)
    assert info["levels"] <= 40
    assert info["tiles"] < 0.25 * info["blocks"] * (info["blocks"] + 1) // 2


def test_duplicated_matches_and_a_long_dense_track(lfr_lib):
    """a 44-node all-pairs track (a dense chain of blocks: every column has many rows -> extra-row tasks) plus short tracks with
    duplicated matches hanging off it"""
    rng = np.random.default_rng(9)
    a, b, k = [], [], []
    L = 44
    for i in range(L):
        for j in range(i + 1, L):
            a.append(i); b.append(j); k.append(0)
    n_var = L
    for t in range(12):
        sz = int(rng.integers(2, 6))
        base = n_var
        for i in range(sz):
            for j in range(i + 1, sz):
                a.append(base + i); b.append(base + j); k.append(0)
                if rng.random() < 0.3:
                    a.append(base + i); b.append(base + j); k.append(0)          # a duplicated match
        a.append(base); b.append(int(rng.integers(L))); k.append(1)
        a.append(base); b.append(b[-1]); k.append(1)                              # the inter-track match twice
        n_var += sz
    w = _by_source(_words(a, b, k))
    pl, info, lvl = _run(n_var, w, seed=2)
    assert pl.x_ptr[-1] > 0                                      # the dense track's columns have more than three rows below the diagonal


def test_dense_meta_graph_falls_back_to_one_segment(lfr_lib):
    """tracks matched to (nearly) every other track: no separator exists; the plan must still be valid"""
    rng = np.random.default_rng(11)
    T, L = 14, 3
    a, b, k = [], [], []
    for t in range(T):
        for i in range(L):
            for j in range(i + 1, L):
                a.append(t * L + i); b.append(t * L + j); k.append(0)
    for t in range(T):
        for u in range(t + 1, T):
            if rng.random() < 0.8:
                a.append(t * L + int(rng.integers(L))); b.append(u * L + int(rng.integers(L))); k.append(1)
    w = _by_source(_words(a, b, k))
    _run(T * L, w, seed=3)


def test_variable_nodes_linked_only_through_constants(lfr_lib):
    """the variable nodes of a component may fall apart once the roots are constants: independent trees, one plan"""
    n_var, w1 = _chain_of_tracks(6, 4, np.random.default_rng(13), shuffle=False)
    a = list(range(0, 24, 4)); b = [24] * 6                      # every track's first node matched to one constant
    w = _by_source(np.concatenate([w1, _words(a, b, [0] * 6)]))
    pl, info, lvl = _run(n_var, w, seed=4)


def test_plan_is_deterministic(lfr_lib):
    n_var, w = _chain_of_tracks(60, 6, np.random.default_rng(17), links=2)
    b1, i1 = capi.tree_plan(n_var, w)
    b2, i2 = capi.tree_plan(n_var, w)
    assert (b1 == b2).all() and i1 == i2


@pytest.mark.parametrize("seed", range(6))
def test_random_sparse_components(lfr_lib, seed):
    rng = np.random.default_rng(100 + seed)
    T = int(rng.integers(20, 90))
    sizes = rng.integers(1, 9, size=T)
    off = np.r_[0, np.cumsum(sizes)]
    n_var = int(off[-1])
    a, b, k = [], [], []
    for t in range(T):
        for i in range(sizes[t]):
            for j in range(i + 1, sizes[t]):
                if rng.random() < 0.8 or j == i + 1:
                    a.append(off[t] + i); b.append(off[t] + j); k.append(0)
    n_links = int(T * rng.uniform(1.0, 2.0))
    for _ in range(n_links):
        t, u = rng.integers(T, size=2)
        if t != u:
            a.append(off[t] + int(rng.integers(sizes[t]))); b.append(off[u] + int(rng.integers(sizes[u]))); k.append(1)
    w = _by_source(_words(a, b, k))
    _run(n_var, w, seed=seed)

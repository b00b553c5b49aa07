"""Block-envelope plans of the large components (csrc/lfr_order.cpp; the reference: SPARSE_NORMAL_CHOLESKY, solve.cc:147)."""
import numpy as np
import pytest

from lfr_amd import capi


def _words(a, b, kind):
    a = np.asarray(a, np.uint32); b = np.asarray(b, np.uint32); kind = np.asarray(kind, np.uint32)
    return np.concatenate([a | ((b | (kind << 15)) << 16), b | ((a | (kind << 15)) << 16)]).astype(np.uint32)


def _check_plan(n_var, w, pos, fb, info):
    assert sorted(pos.tolist()) == list(range(n_var))                       # a permutation
    rt = (2 * n_var + 16) // 16
    assert info["RT"] == rt and len(fb) == rt
    assert info["tiles"] == int(sum(r - int(fb[r]) + 1 for r in range(rt)))
    assert all(int(fb[r]) <= r for r in range(rt)) and fb[(2 * n_var) >> 4] == 0   # the right-hand side's block row spans every column
    s, d = (w & 0xffff).astype(np.int64), ((w >> 16) & 0x7fff).astype(np.int64)
    m = (s < n_var) & (d < n_var)
    ps, pd = pos[s[m]].astype(np.int64), pos[d[m]].astype(np.int64)
    hi, lo = np.maximum(ps, pd), np.minimum(ps, pd)
    assert (fb[(2 * hi) >> 4].astype(np.int64) <= ((2 * lo) >> 4)).all()     # every matrix entry lies inside the envelope of its block row
    assert info["tiles"] <= rt * (rt + 1) // 2 and info["tiles"] == min(info["tiles_by_tracks"], info["tiles_rcm"])


def test_chain_of_tracks_has_a_thin_envelope(lfr_lib):
    """200 six-node cliques (intra-track edges) chained by one inter-track edge each: ~3 tiles per block row, not 75."""
    rng = np.random.default_rng(3)
    T, L = 200, 6
    n_var = T * L
    ids = rng.permutation(n_var)                       # local numbering unrelated to the structure
    a, b, k = [], [], []
    for t in range(T):
        for i in range(L):
            for j in range(i + 1, L):
                a.append(ids[t * L + i]); b.append(ids[t * L + j]); k.append(0)
        if t + 1 < T:
            a.append(ids[t * L + int(rng.integers(L))]); b.append(ids[(t + 1) * L + int(rng.integers(L))]); k.append(1)
    w = _words(a, b, k)
    pos, fb, info = capi.sky_plan(n_var, w)
    _check_plan(n_var, w, pos, fb, info)
    rt = info["RT"]
    assert info["tiles"] <= 4 * rt                      # diagonal + at most two neighbours + the right-hand side's row
    assert info["tiles"] < 0.06 * rt * (rt + 1) // 2


def test_random_tree_of_tracks_beats_dense_by_far(lfr_lib):
    """tracks hanging together as a random tree (+ a few cycles): heavy-first postorder keeps the envelope near n log n."""
    rng = np.random.default_rng(5)
    T = 220
    sizes = rng.integers(2, 11, size=T)
    off = np.r_[0, np.cumsum(sizes)]
    n_var = int(off[-1])
    a, b, k = [], [], []
    for t in range(T):
        for i in range(sizes[t]):
            for j in range(i + 1, sizes[t]):
                a.append(off[t] + i); b.append(off[t] + j); k.append(0)
        if t > 0:
            u = int(rng.integers(t))                    # random recursive tree
            a.append(off[t] + int(rng.integers(sizes[t]))); b.append(off[u] + int(rng.integers(sizes[u]))); k.append(1)
    for _ in range(6):                                  # a few cycles
        t, u = rng.integers(T, size=2)
        if t != u:
            a.append(off[t]); b.append(off[u]); k.append(1)
    w = _words(a, b, k)
    pos, fb, info = capi.sky_plan(n_var, w)
    _check_plan(n_var, w, pos, fb, info)
    rt = info["RT"]
    assert info["tiles"] < 0.25 * rt * (rt + 1) // 2


@pytest.mark.parametrize("n_var", [1, 2, 7, 8, 9, 97])
def test_small_and_degenerate_inputs(lfr_lib, n_var):
    rng = np.random.default_rng(n_var)
    m = 3 * n_var
    a = rng.integers(0, n_var + 3, size=m)              # indices >= n_var are constants: ignored by the plan
    b = rng.integers(0, n_var + 3, size=m)
    w = _words(a, b, rng.integers(0, 2, size=m))
    pos, fb, info = capi.sky_plan(n_var, w)
    _check_plan(n_var, w, pos, fb, info)
    pos2, fb2, info2 = capi.sky_plan(n_var, w)          # deterministic
    assert (pos == pos2).all() and (fb == fb2).all()
    pos3, fb3, info3 = capi.sky_plan(n_var, np.zeros(0, np.uint32))      # no edges at all: a diagonal matrix
    assert info3["tiles"] == info3["RT"] + ((2 * n_var) >> 4)

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

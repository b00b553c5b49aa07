"""The algebra of factor_lds' diagonal tiles since round 5, restated in numpy and checked against numpy's own solver (CPU; the kernel itself
is checked against the oracle by the GPU parity tests): carrying the rows of the identity through a diagonal tile's elimination yields
M = (I + D^-1 U^T)^-1; the substitution of the tiles below is tile <- tile M; the tile's part of the back substitution is y = M (D^-1 z);
U itself is never needed again.  A blocked LDL^T built ONLY from these pieces (16-row tiles, the right-hand side riding as the last row, as in
lfr_solve.hip) must solve the system."""
import numpy as np
import pytest


def eliminate(tile, below, nbp):
    """factor_diag: lane = row.  `tile`: the diagonal tile's rows (lower triangle valid), `below`: rows of the tile below (all 16 columns).
    Returns d (pivots), the substituted rows of `below` (unscaled entries) and M (nbp x nbp, unit upper triangular)."""
    a = np.tril(tile).copy()                       # group 0: entries j <= row, zeros to the right (as the kernel initialises them)
    b = below.copy()                               # group 1
    e = np.eye(tile.shape[0])                      # group 2: the rows of the identity
    d = np.zeros(nbp)
    for k in range(nbp):
        d[k] = a[k, k]
        col = a[:, k].copy()                       # readlane(a[k], j): entry (j, k) of the diagonal tile, final for j > k
        for rows in (a, b, e):
            lik = rows[:, k] / d[k]
            for j in range(k + 1, tile.shape[0]):
                rows[:, j] -= lik * col[j]
    return d, b, e[:nbp, :nbp]


def blocked_solve(A, g, T=16):
    """LDL^T of [[A, g], [g^T, .]] by tiles of T rows with the pieces above; returns y with A y = g."""
    n = A.shape[0]
    n1 = n + 1
    W = np.zeros((n1, n1)); W[:n, :n] = np.tril(A); W[n, :n] = g      # packed lower triangle + the right-hand-side row
    P = (n + T - 1) // T
    Ms, inv = {}, np.zeros(n)
    for k in range(P):
        kb = T * k
        nbp = min(T, n - kb)
        hi = min(kb + T, n1)
        tile = np.zeros((T, T)); tile[:hi - kb, :hi - kb] = W[kb:hi, kb:hi]
        for i in range(hi - kb, T): tile[i, i] = 1.0                   # padding rows: identity
        lo2, hi2 = kb + T, min(kb + 2 * T, n1)
        below = np.zeros((T, T))
        if hi2 > lo2: below[:hi2 - lo2, :] = W[lo2:hi2, kb:kb + T]
        d, bsub, M = eliminate(tile, below, nbp)
        inv[kb:kb + nbp] = 1.0 / d
        Ms[k] = M
        if hi - kb > nbp:                                              # the right-hand-side row lies in this tile: its entries are w
            # (group 0 eliminated it as an ordinary row)
            a = np.tril(tile).copy()
            r = a[nbp].copy()
            # redo the row's elimination with the final pivot rows: r_c -= (r_k / d_k) a_ck - exactly `row M` on its first nbp entries
            W[n, kb:kb + nbp] = r[:nbp] @ M
        if hi2 > lo2:
            W[lo2:hi2, kb:kb + T] = bsub[:hi2 - lo2, :]
        if lo2 < n1:                                                   # the other tiles of the column block: tile <- tile M (finish_rows)
            assert nbp == T
            W[hi2:n1, kb:kb + T] = W[hi2:n1, kb:kb + T] @ M
            # trailing update: (R, J) -= rows R of the panel (rows J of the panel / d)^T
            Lp = W[lo2:n1, kb:kb + T]
            upd = (Lp * inv[kb:kb + T]) @ Lp.T
            idx = np.arange(lo2, n1)
            W[np.ix_(idx, idx)] -= np.tril(upd)
    # back substitution, tile by tile from the last (lfr_solve.hip, solve_component): y = M (D^-1 z) inside a tile, z_j -= a_kj y_k above it
    z = W[n, :n].copy()
    y = np.zeros(n)
    for k in range(P - 1, -1, -1):
        kb = T * k
        nbp = min(T, n - kb)
        y[kb:kb + nbp] = Ms[k] @ (z[kb:kb + nbp] * inv[kb:kb + nbp])
        z[:kb] -= W[kb:kb + nbp, :kb].T @ y[kb:kb + nbp]
    return y, Ms


@pytest.mark.parametrize("n", [2, 7, 15, 16, 17, 31, 32, 33, 48, 50, 96, 190, 192])
def test_blocked_ldlt_from_m_solves_the_system(n):
    rng = np.random.default_rng(n)
    J = rng.standard_normal((3 * n + 5, n))
    A = J.T @ J + 0.1 * np.eye(n)                  # SPD, like the damped normal matrix
    g = rng.standard_normal(n)
    y, Ms = blocked_solve(A, g)
    want = np.linalg.solve(A, g)
    assert np.abs(y - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    for M in Ms.values():                          # unit upper triangular
        assert np.allclose(np.diag(M), 1.0) and np.allclose(np.tril(M, -1), 0.0)


def test_m_is_the_substitution_as_a_matrix():
    rng = np.random.default_rng(5)
    J = rng.standard_normal((40, 16))
    tile = J.T @ J + 0.5 * np.eye(16)
    rows = rng.standard_normal((16, 16))
    d, sub, M = eliminate(tile, rows, 16)
    assert np.allclose(sub, rows @ M, rtol=1e-12, atol=1e-12)                      # carried through the elimination == times M
    L = np.linalg.cholesky(tile); Lu = L / np.diag(L)                               # unit lower factor
    assert np.allclose(M, np.linalg.inv(Lu).T, rtol=1e-9, atol=1e-9)               # M = L^-T
    assert np.allclose(d, np.diag(L) ** 2, rtol=1e-10)

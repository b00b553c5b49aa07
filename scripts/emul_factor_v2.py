"""CPU emulation of the workgroup kernel's blocked LDL^T (lfr_solve.hip, factor_lds) at LANE level.

Not product code and not the oracle: a design check that runs here (no GPU).  It follows the HIP code
statement by statement - packed lower-triangular matrix with the right-hand side as row n, 16-column
panels, wave 0 factors the next diagonal block behind a `ready` flag, every wave updates its tiles with
the v_mfma_f64_16x16x4 lane layout (lane l feeds A[l&15][l>>4], B[l>>4][l&15], holds D[(l>>4)+4r][l&15])
and finishes its tiles of the next panel by a lane = row substitution - with the waves of a phase run in
random order and a race detector on every LDS access (a wave may not read what another wave wrote in
the same phase unless the `ready` flag orders the two).

    python scripts/emul_factor_v2.py            # a few sizes x wave counts, compares with numpy
"""
import sys

import numpy as np


def tri(i, j):
    return i * (i + 1) // 2 + j


class Lds:
    def __init__(self, size):
        self.v = np.zeros(size)
        self.w_phase = -np.ones(size, np.int64)      # phase of the last write
        self.w_wave = -np.ones(size, np.int64)
        self.w_pub = np.zeros(size, bool)            # written before the writer raised `ready` in that phase
        self.phase = 0
        self.ready_raised = False
        self.races = 0

    def rd(self, wave, addr, after_ready=False):
        if self.w_phase[addr] == self.phase and self.w_wave[addr] != wave:
            if not (after_ready and self.w_pub[addr]):
                self.races += 1
        return self.v[addr]

    def wr(self, wave, addr, val):
        if self.w_phase[addr] == self.phase and self.w_wave[addr] != wave:
            self.races += 1
        self.v[addr] = val
        self.w_phase[addr] = self.phase
        self.w_wave[addr] = wave
        self.w_pub[addr] = (wave == 0 and not self.ready_raised)


def factor(A, g, n_waves, rng, handicap=6):
    """A: n x n SPD, g: rhs.  Returns (Mat packed, vinv) after the emulated factorization."""
    n = A.shape[0]
    n1 = n + 1
    P = (n + 15) >> 4
    RT = (n1 + 15) >> 4
    size = tri(n1, 0) + n1
    L = Lds(size + n)                                  # + vinv
    VINV = size
    for i in range(n):
        for j in range(i + 1):
            L.v[tri(i, j)] = A[i, j]
    for j in range(n):
        L.v[tri(n, j)] = g[j]
    L.v[tri(n, n)] = 123.0                             # never used

    def factor_diag(wave, kb):
        nbp = min(16, n - kb)
        a = np.zeros((64, 16))
        for lane in range(64):
            i = lane & 15
            row = kb + i
            rv = row < n1
            for j in range(16):
                if rv and j <= i:
                    a[lane, j] = L.rd(wave, tri(row, kb + j))
                else:
                    a[lane, j] = 1.0 if j == i else 0.0
        bad = False
        for k in range(16):
            if k >= nbp:
                break
            dk = a[k, k]                                # readlane(a[k], k)
            bad = bad or not (dk > 0)
            ik = 1.0 / dk
            L.wr(wave, VINV + kb + k, ik)
            lik = a[:, k] * ik
            for j in range(k + 1, 16):
                ajk = a[j, k]                           # readlane(a[k], j)
                a[:, j] = a[:, j] - lik * ajk
        for lane in range(16):
            i = lane
            row = kb + i
            if row < n1:
                for j in range(i + 1):
                    L.wr(wave, tri(row, kb + j), a[lane, j])
        return bad

    def trsm(wave, kb, R0, R1):
        r = np.zeros((64, 16))
        act = np.zeros(64, bool)
        rows = np.zeros(64, np.int64)
        for lane in range(64):
            t, i = lane >> 4, lane & 15
            R = R0 if t == 0 else (R1 if t == 1 else -1)
            row = 16 * R + i
            act[lane] = R >= 0 and row < n1
            rows[lane] = row if act[lane] else n1 - 1
            for c in range(16):
                r[lane, c] = L.rd(wave, tri(rows[lane], kb + c), after_ready=False) if act[lane] else 0.0
        for k in range(15):
            tk = r[:, k] * L.rd(wave, VINV + kb + k, after_ready=True)
            for c in range(k + 1, 16):
                r[:, c] = r[:, c] - tk * L.rd(wave, tri(kb + c, kb + k), after_ready=True)
        for lane in range(64):
            if act[lane]:
                for c in range(1, 16):
                    L.wr(wave, tri(rows[lane], kb + c), r[lane, c])

    def update_tile(wave, R, J, kb):
        """tile (R, J) -= U(R, panel) * (U(J, panel) / d)^T with the MFMA lane layout; stores the lower part."""
        D = np.zeros((16, 16))
        Aop = np.zeros((16, 16))
        Bop = np.zeros((16, 16))
        ok = np.zeros((16, 16), bool)
        for lane in range(64):
            r16, kq = lane & 15, lane >> 4
            ia, jb = 16 * R + r16, 16 * J + r16
            iac, jbc = min(ia, n1 - 1), min(jb, n1 - 1)
            for kk in range(4):
                col = kb + 4 * kk + kq
                av = L.rd(wave, tri(iac, col))
                bv = L.rd(wave, tri(jbc, col)) * (-L.rd(wave, VINV + col))
                Aop[r16, 4 * kk + kq] = av if ia < n1 else 0.0
                Bop[4 * kk + kq, r16] = bv if jb < n1 else 0.0
            for r in range(4):
                row = 16 * R + kq + 4 * r
                rc = min(row, n1 - 1)
                okk = row < n1 and jb <= row
                cv = L.rd(wave, tri(rc, min(jbc, rc)))
                D[kq + 4 * r, r16] = cv if okk else 0.0
                ok[kq + 4 * r, r16] = okk
        D = D + Aop @ Bop
        for lane in range(64):
            r16, kq = lane & 15, lane >> 4
            for r in range(4):
                if ok[kq + 4 * r, r16]:
                    L.wr(wave, tri(16 * R + kq + 4 * r, 16 * J + r16), D[kq + 4 * r, r16])

    def column_tiles(wave, kcol):
        """tiles (R, kcol), R > kcol, of this worker wave (1..n_waves-1), two at a time"""
        out = []
        if wave == 0:
            return out
        W1 = n_waves - 1
        R = kcol + wave
        while R < RT:
            out.append((R, R + W1 if R + W1 < RT else -1))
            R += 2 * W1
        return out

    # ---- prologue: column 0 ----
    L.phase = 0
    L.ready_raised = False
    bad = factor_diag(0, 0)
    L.ready_raised = True
    for wave in rng.permutation(n_waves):
        for R0, R1 in column_tiles(wave, 0):
            trsm(wave, 0, R0, R1)
    # ---- phases ----
    for k in range(P - 1):
        L.phase += 1
        L.ready_raised = False
        kb = 16 * k
        # part B tiles: (R, J), k+2 <= J <= R <= RT-1, columns that exist (16 J < n)
        m = RT - (k + 2)
        tiles = [(k + 2 + I, k + 2 + J) for I in range(max(m, 0)) for J in range(I + 1)]
        T = len(tiles)
        W1 = n_waves - 1
        T0 = min(T, handicap * W1)
        own = {w: [] for w in range(n_waves)}
        for t in range(T):
            own[1 + t % W1].append(tiles[t])
        # wave 0 first up to `ready` (the others' part B may run before or after: random order below)
        order = list(rng.permutation(n_waves))
        did_diag = False

        def diag():
            update_tile(0, k + 1, k + 1, kb)
            b = factor_diag(0, kb + 16)
            L.ready_raised = True
            return b

        # emulate: part B of waves that come before wave 0 in `order` runs before the diagonal block is ready
        for w in order:
            if w == 0:
                bad = diag() or bad
                did_diag = True
            for (R, J) in own[w]:
                if 16 * J < n:
                    update_tile(w, R, J, kb)
        assert did_diag
        for w in rng.permutation(n_waves):
            for R0, R1 in column_tiles(w, k + 1):
                update_tile(w, R0, k + 1, kb)
                if R1 >= 0:
                    update_tile(w, R1, k + 1, kb)
                trsm(w, kb + 16, R0, R1)
    return L.v[:size].copy(), L.v[VINV:VINV + n].copy(), L.races, bad


def check(n, n_waves, rng):
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    g = rng.standard_normal(n)
    Mat, vinv, races, bad = factor(A, g, n_waves, rng)
    # reference: A = L D L^T, U = L D (unscaled columns); row n = (L^-1 g) unscaled = D L^-1... see kernel comment
    Lc = np.linalg.cholesky(A)
    d = np.diag(Lc) ** 2
    Lu = Lc / np.diag(Lc)[None, :]                  # unit lower
    U = Lu * d[None, :]
    err = 0.0
    for i in range(n):
        for j in range(i + 1):
            ref = d[i] if i == j else U[i, j]
            err = max(err, abs(Mat[tri(i, j)] - ref) / max(1.0, abs(ref)))
    z = np.linalg.solve(Lu, g)                      # L^-1 g ; stored unscaled: row n entries u_nj = z_j (as "a_nj" with l_nj = a_nj / d_j)
    zrow = np.array([Mat[tri(n, j)] for j in range(n)])
    err_rhs = np.abs(zrow - z).max()
    err_inv = np.abs(vinv - 1.0 / d).max()
    # the step the kernel derives: x = L^-T D^-1 (L^-1 g)
    x = np.linalg.solve(Lu.T, zrow * vinv)
    err_x = np.abs(x - np.linalg.solve(A, g)).max()
    print("n=%3d waves=%d: factor err %.2e rhs err %.2e inv err %.2e solve err %.2e races %d bad %s" % (n, n_waves, err, err_rhs, err_inv, err_x, races, bad))
    return err < 1e-9 and err_rhs < 1e-9 and err_inv < 1e-9 and err_x < 1e-9 and races == 0 and not bad


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    ok = True
    for n, w in [(4, 2), (15, 2), (16, 2), (17, 2), (31, 2), (32, 4), (33, 2), (48, 2), (64, 4), (88, 2), (100, 4), (128, 4), (130, 4), (144, 8), (176, 8), (190, 8), (191, 8), (192, 8)]:
        ok = check(n, w, rng) and ok
    print("OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)

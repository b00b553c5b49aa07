"""CPU execution of an elimination-tree plan (csrc/lfr_treeplan.cpp), tile by tile, the way solve_tree_component
(csrc/lfr_solve.hip) walks it: sweep items -> tiles, level-scheduled left-looking LDL^T, level-scheduled back substitution.
Test infrastructure: numpy, no product code."""
import numpy as np

NONE = 0xFFFFFFFF


class Plan:
    def __init__(self, blob):
        b = np.asarray(blob, np.uint32)
        self.blob = b
        self.NB, self.n_tiles, self.off_tiles, self.off_vec, self.n_pad, self.n_levels, self.n_items, self.n_p1 = (int(x) for x in b[:8])
        o = [int(x) for x in b[8:23]] + [int(b[25]), int(b[26]), int(b[27])]
        NB, L = self.NB, self.n_levels

        def arr(i, n):
            return b[o[i]:o[i] + n].astype(np.int64)
        self.colptr = arr(0, NB + 1)
        self.rowsof = arr(1, self.n_tiles)
        self.nreal = arr(2, NB)
        self.level_ptr = arr(3, L + 1)
        self.level_cols = arr(4, NB)
        self.p1_ptr = arr(5, L + 1)
        self.p1_tasks = arr(6, 4 * self.n_p1).reshape(-1, 4)
        n_upd = int(self.p1_tasks[:, 2].max()) if self.n_p1 else 0
        self.upd = arr(7, 3 * n_upd).reshape(-1, 3)
        self.x_ptr = arr(8, L + 1)
        self.x_tasks = arr(9, 3 * int(self.x_ptr[-1])).reshape(-1, 3)
        self.ncarry = arr(10, NB)
        self.items = arr(11, 8 * (self.n_items + 1)).reshape(-1, 8)
        n_ext = int(np.maximum(self.items[:-1, 3] - 2, 0).sum())
        self.item_edges = arr(12, n_ext)
        self.node_items = arr(13, 8 * NB + 1)
        self.ipos = arr(14, 8 * NB)
        self.off_part, self.vec_stride = int(b[23]), int(b[24])
        self.col_upd_ptr = arr(15, NB + 1)
        self.col_upd = arr(16, 5 * int(self.col_upd_ptr[-1])).reshape(-1, 5)
        self.col_desc = arr(17, 32 * NB).reshape(-1, 32)

    # ---- structural invariants ----
    def check(self, n_var):
        NB = self.NB
        assert self.n_pad == 16 * NB and self.vec_stride == 16 * NB + 16
        assert self.off_tiles % 32 == 0 and self.off_vec % 32 == 0 and self.off_tiles * 2 >= 32
        assert self.off_part == self.off_tiles + 512 * self.n_tiles and self.off_vec >= self.off_part + 6 * self.n_items
        real = self.ipos[self.ipos != NONE]
        assert sorted(real.tolist()) == list(range(n_var))                       # every variable node has exactly one position
        for J in range(NB):                                                      # real slots first, padding behind
            sl = self.ipos[8 * J:8 * J + 8]
            assert ((sl != NONE).sum() == self.nreal[J]) and (sl[:self.nreal[J]] != NONE).all() and self.nreal[J] >= 1
        assert self.colptr[0] == 0 and self.colptr[-1] == self.n_tiles
        lvl = np.full(NB, -1)
        for l in range(self.n_levels):
            for J in self.level_cols[self.level_ptr[l]:self.level_ptr[l + 1]]:
                assert lvl[J] < 0
                lvl[J] = l
        assert (lvl >= 0).all()
        for J in range(NB):
            rows = self.rowsof[self.colptr[J]:self.colptr[J + 1]]
            assert rows[0] == J and (np.diff(rows) > 0).all()
            assert (lvl[rows[1:]] > lvl[J]).all()                                # a column's rows are its ancestors
            if len(rows) > 1:
                assert lvl[rows[1]] >= lvl[J] + 1
        # update lists: every tile below a diagonal is updated by exactly one party (its column's task if carried, a tile task
        # otherwise), sources at lower levels, tiles of the right rows / columns, ascending k = a fixed summation order
        colof = np.repeat(np.arange(NB), np.diff(self.colptr))
        seen = set()
        for l in range(self.n_levels):
            for t in range(self.p1_ptr[l], self.p1_ptr[l + 1]):
                tt, ub, ue, J = (int(x) for x in self.p1_tasks[t])
                assert tt not in seen and ue >= ub
                seen.add(tt)
                assert colof[tt] == J and lvl[J] == l and tt - self.colptr[J] - 1 >= self.ncarry[J]
                u = self.upd[ub:ue]
                assert (np.diff(u[:, 2]) > 0).all()
                assert (colof[u[:, 0]] == u[:, 2]).all() and (colof[u[:, 1]] == u[:, 2]).all()
                assert (self.rowsof[u[:, 0]] == self.rowsof[tt]).all() and (self.rowsof[u[:, 1]] == J).all()
                assert (lvl[u[:, 2]] < l).all()
        for J in range(NB):
            ce = self.col_upd[self.col_upd_ptr[J]:self.col_upd_ptr[J + 1]]
            assert (np.diff(ce[:, 0]) > 0).all() and (lvl[ce[:, 0]] < lvl[J]).all()
            assert (colof[ce[:, 1]] == ce[:, 0]).all() and (self.rowsof[ce[:, 1]] == J).all()
            for i in range(3):
                m = ce[:, 2 + i] != NONE
                assert not m.any() or i < self.ncarry[J]
                if m.any():
                    assert (colof[ce[m, 2 + i]] == ce[m, 0]).all() and (self.rowsof[ce[m, 2 + i]] == self.rowsof[self.colptr[J] + 1 + i]).all()
        for q in range(NB):                                                      # the descriptors restate the tables, column by column in level order
            dsc = self.col_desc[q]
            J = int(self.level_cols[q])
            e0, e1 = self.col_upd_ptr[J], self.col_upd_ptr[J + 1]
            ns = self.colptr[J + 1] - self.colptr[J] - 1
            assert dsc[0] == J and dsc[1] == self.colptr[J] and dsc[2] == self.ncarry[J] and dsc[3] == 2 * self.nreal[J]
            assert dsc[4] == e1 - e0 and dsc[5] == e0 + 2 and dsc[16] == ns
            for i in range(min(2, e1 - e0)):
                assert (dsc[6 + 5 * i:11 + 5 * i] == self.col_upd[e0 + i]).all()
            for i in range(min(4, ns)):
                assert dsc[17 + i] == self.rowsof[self.colptr[J] + 1 + i]
            assert dsc[21] == (self.rowsof[self.colptr[J] + 1] if ns else NONE)      # parent in the elimination tree
            assert dsc[22] == sum(1 for K in range(NB) if self.colptr[K + 1] - self.colptr[K] > 1 and self.rowsof[self.colptr[K] + 1] == J)
            if ns > self.ncarry[J]:                                                # its tile tasks: one per tile it does not carry, in tile order
                for i in range(self.ncarry[J], ns):
                    assert self.p1_tasks[dsc[23] + i - self.ncarry[J], 0] == self.colptr[J] + 1 + i
        max_extra = max(int(self.colptr[J + 1] - self.colptr[J] - 1 - self.ncarry[J]) for J in range(NB))
        assert int(self.blob[28]) == (1 if NB <= 4096 and max_extra <= 6 and self.n_p1 <= NB // 8 + 4 else 0)
        for l in range(self.n_levels):
            for t in range(self.x_ptr[l], self.x_ptr[l + 1]):
                J, i0, cnt = self.x_tasks[t]
                assert lvl[J] == l and 1 <= cnt <= 4 and i0 >= self.ncarry[J] and i0 + cnt <= self.colptr[J + 1] - self.colptr[J] - 1
        for J in range(NB):
            ns = self.colptr[J + 1] - self.colptr[J] - 1
            nc = self.ncarry[J]
            assert nc <= min(ns, 3)
            if nc == 3:
                assert self.nreal[self.rowsof[self.colptr[J] + 3]] <= 7          # lanes 49-63 hold rows 0-14 of the third tile
            covered = nc + sum(int(c) for (j, i0, c) in self.x_tasks if j == J)
            assert covered == ns
            for i in range(nc, ns):                                                # every tile its column does not carry has a tile task
                assert int(self.colptr[J] + 1 + i) in seen
        return lvl

    # ---- the sweep: per-record quantities -> tiles, gradient, diagonal ----
    def assemble(self, words, n_var, J1, sq, r):
        """words[e] = src | (dst | kind << 15) << 16; J1[e] (2x2, d r / d x_src, corrected), sq[e], r[e] (2, corrected).
        Returns (tiles[n_tiles,16,16], g[n_pad], n_cost_counts[e])."""
        tiles = np.zeros((self.n_tiles * 256,))
        g = np.zeros(self.n_pad)
        counted = np.zeros(len(words), np.int64)
        part = np.zeros((self.n_items, 5))
        for i in range(self.n_items):
            xv, xu, cross, ne, ew0, ew1, ext, _ = self.items[i]
            cb = np.zeros((2, 2))
            d = np.zeros((2, 2))
            gg = np.zeros(2)
            for q in [ew0, ew1][:min(ne, 2)] + self.item_edges[ext:ext + max(ne - 2, 0)].tolist():
                e, dr, cf = int(q) >> 2, (int(q) >> 1) & 1, int(q) & 1
                counted[e] += cf
                if dr == 0:            # record v -> u : d r / d x_v = J1, d r / d x_u = sq I
                    d += J1[e].T @ J1[e]
                    gg += J1[e].T @ r[e]
                    cb += J1[e].T * sq[e]
                else:                  # record u -> v : d r / d x_v = sq I, d r / d x_u = J1
                    d += sq[e] * sq[e] * np.eye(2)
                    gg += sq[e] * r[e]
                    cb += sq[e] * J1[e]
            if cross != NONE:
                for c in range(2):
                    tiles[cross + 16 * c:cross + 16 * c + 2] = cb[c]
            part[i] = (d[0, 0], d[1, 0], d[1, 1], gg[0], gg[1])
        for p in range(8 * self.NB):
            if self.ipos[p] == NONE:
                assert self.node_items[p] == self.node_items[p + 1]
                continue
            s = part[self.node_items[p]:self.node_items[p + 1]].sum(axis=0)
            J, sl = p >> 3, p & 7
            T = self.colptr[J] * 256
            tiles[T + (2 * sl) * 16 + 2 * sl] = s[0]
            tiles[T + (2 * sl + 1) * 16 + 2 * sl] = s[1]
            tiles[T + (2 * sl + 1) * 16 + 2 * sl + 1] = s[2]
            g[2 * p], g[2 * p + 1] = s[3], s[4]
        return tiles.reshape(self.n_tiles, 16, 16), g, counted

    # ---- level-scheduled left-looking LDL^T (unscaled columns, d on the diagonal, the right-hand side riding along) ----
    def factor(self, tiles, w):
        tiles = tiles.copy()
        w = w.copy().reshape(self.NB, 16)
        inv = np.zeros((self.NB, 16))
        for l in range(self.n_levels):
            for t in range(self.p1_ptr[l], self.p1_ptr[l + 1]):      # tile tasks: the tiles their columns do not carry
                tt, ub, ue, J = self.p1_tasks[t]
                for (ta, tb, k) in self.upd[ub:ue]:
                    tiles[tt] -= tiles[ta] @ np.diag(inv[k]) @ tiles[tb].T
            for J in self.level_cols[self.level_ptr[l]:self.level_ptr[l + 1]]:      # column tasks: updates, then the elimination
                t0 = self.colptr[J]
                nbp = 2 * self.nreal[J]
                for (k, tb, c0, c1, c2) in self.col_upd[self.col_upd_ptr[J]:self.col_upd_ptr[J + 1]]:
                    B = np.diag(inv[k]) @ tiles[tb].T
                    tiles[t0] -= tiles[tb] @ B
                    w[J] -= tiles[tb] @ (inv[k] * w[k])
                    for i, c in enumerate((c0, c1, c2)):
                        if c != NONE:
                            tiles[t0 + 1 + i] -= tiles[c] @ B
                D = np.tril(tiles[t0])
                rows = [tiles[t0 + 1 + i] for i in range(self.ncarry[J])] + [w[J][None, :]]
                for kk in range(nbp):
                    dk = D[kk, kk]
                    assert dk > 0, "non-positive pivot"
                    inv[J, kk] = 1.0 / dk
                    colk = D[:, kk].copy()
                    for M in [D] + rows:
                        lik = M[:, kk] * inv[J, kk]
                        lo = kk + 1
                        if M is D:
                            M[lo:, lo:] -= np.outer(lik[lo:], colk[lo:])
                        else:
                            M[:, lo:] -= np.outer(lik, colk[lo:])
                tiles[t0] = np.tril(D)
            for t in range(self.x_ptr[l], self.x_ptr[l + 1]):                        # phase 2b: rows the column task did not carry
                J, i0, cnt = self.x_tasks[t]
                D = tiles[self.colptr[J]]
                for i in range(i0, i0 + cnt):
                    R = tiles[self.colptr[J] + 1 + i]
                    for j in range(15):
                        tj = R[:, j] * inv[J, j]
                        R[:, j + 1:] -= np.outer(tj, D[j + 1:, j])
        return tiles, w.reshape(-1), inv.reshape(-1)

    def back_substitute(self, tiles, w, inv):
        w = w.reshape(self.NB, 16)
        inv = inv.reshape(self.NB, 16)
        y = np.zeros((self.NB, 16))
        for l in range(self.n_levels - 1, -1, -1):
            for J in self.level_cols[self.level_ptr[l]:self.level_ptr[l + 1]]:
                z = w[J].copy()
                for t in range(self.colptr[J] + 1, self.colptr[J + 1]):
                    z -= tiles[t].T @ y[self.rowsof[t]]
                D = tiles[self.colptr[J]]
                for k in range(15, -1, -1):
                    y[J, k] = z[k] * inv[J, k]
                    z[:k] -= D[k, :k] * y[J, k]
        return y.reshape(-1)


def dense_reference(plan, words, n_var, J1, sq, r):
    """A = J^T J and g = J^T r in the plan's (padded) matrix order, assembled record by record."""
    pos = np.full(n_var, -1, np.int64)
    for p, v in enumerate(plan.ipos):
        if v != NONE:
            pos[v] = p
    A = np.zeros((plan.n_pad, plan.n_pad))
    g = np.zeros(plan.n_pad)
    for e, wd in enumerate(words):
        s, d = int(wd) & 0xFFFF, (int(wd) >> 16) & 0x7FFF
        if s == d:
            continue
        blocks = []
        if s < n_var:
            blocks.append((2 * pos[s], J1[e]))
        if d < n_var:
            blocks.append((2 * pos[d], sq[e] * np.eye(2)))
        for (ra, Ja) in blocks:
            g[ra:ra + 2] += Ja.T @ r[e]
            for (rb, Jb) in blocks:
                A[ra:ra + 2, rb:rb + 2] += Ja.T @ Jb
    return A, g, pos

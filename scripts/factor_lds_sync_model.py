"""CPU model of the barrier-free schedule of factor_lds (lfr_solve.hip: the blocked LDL^T of solve_block_kernel in LDS) - VERDICT r4 #7.

scripts/emul_factor_v2.py is a lane-level model of round 3's FIRST schedule (one barrier per panel).  The schedule that shipped has no workgroup
barrier inside the factorization: wave 0 owns the diagonal tiles and runs ahead, wave 1 feeds it through two LDS words, the other waves meet at a
counter of their own.  This file restates THAT schedule at TILE level - every wave as the sequence of tile / vector accesses and of accesses to the
four synchronisation words (`ready`, `lead`, `lead_t`, `wbar`) the kernel performs, in the kernel's order, with the kernel's dealing of tiles to
waves - and runs the waves under random interleavings with a vector-clock race detector:

  * a read of a tile must be ordered after the last write of it by another wave;
  * a write must be ordered after every earlier read and write of the tile by other waves (read-modify-write tiles included);
  * ordering comes only from the words: a wave that sees `ready >= need` / `lead >= need` / `lead_t >= need` joins the clock the writer released
    with that value (values only grow), a wave that leaves the workers' barrier joins every arrival's clock;
  * some wave can always proceed (deadlock detector), and at the end every tile has gone through the updates and the substitution it is due.

Not modelled: loads whose values are discarded (clamped loads of load_tile; a row of the diagonal tile reads past its diagonal in factor_diag),
the scratch words lanes without an entry store to (sh.red), the `flag` word (written with the same value by anyone).  Since round 5 the
substitution of a tile is tile <- tile M with M stored by wave 0 inside the diagonal tile: at tile level the accesses are what they were.
Tiles are (R, J), 16 x 16 blocks of the packed lower triangle with the right-hand side as row n (n1 = n + 1 rows, RT row tiles, P panels).
usage: python scripts/factor_lds_sync_model.py"""
import random


class VC:
    def __init__(self, n): self.c = [[0] * n for _ in range(n)]
    def tick(self, w): self.c[w][w] += 1
    def snap(self, w): return list(self.c[w])
    def join(self, w, o): self.c[w] = [max(a, b) for a, b in zip(self.c[w], o)]
    def after(self, stamp, w): return all(s <= c for s, c in zip(stamp, self.c[w]))


def programs(n, n_waves, drop_lead_wait=False, drop_ready_wait=False, drop_lead_t_wait=False, drop_worker_barrier=False):
    """The access sequences of factor_lds for an n-row system on n_waves waves (n_waves - 1 workers).  Steps: ("r", loc), ("w", loc),
    ("wait", word, need), ("set", word, value), ("arrive",), ("barrier", target)."""
    n1 = n + 1
    P = (n + 15) >> 4
    RT = (n1 + 15) >> 4
    K = n_waves - 1
    prog = [[] for _ in range(n_waves)]

    def factor_diag(w, panel):                          # wave 0: diagonal tile + the tile below through the same elimination, then `ready`
        p = prog[w]
        p.append(("r", ("t", panel, panel)))
        below = panel + 1 < RT
        if below: p.append(("r", ("t", panel + 1, panel)))
        p.append(("w", ("t", panel, panel))); p.append(("w", ("v", panel)))
        if below: p.append(("w", ("t", panel + 1, panel)))
        p.append(("set", "ready", 16 * panel + 16))

    def update_pair(w, k, tiles):                       # tile (R, J) -= rows R of panel k x (rows J of panel k / d)^T
        p = prog[w]
        for (R, J) in tiles:
            p.append(("r", ("t", R, k))); p.append(("r", ("t", J, k))); p.append(("r", ("t", R, J)))
        for (R, J) in tiles: p.append(("w", ("t", R, J)))

    def finish_rows(w, panel, rows):                    # substitution of tiles (R, panel) against the diagonal tile once it is published
        p = prog[w]
        for R in rows: p.append(("r", ("t", R, panel)))
        if not drop_ready_wait: p.append(("wait", "ready", 16 * panel + 16))
        p.append(("r", ("t", panel, panel))); p.append(("r", ("v", panel)))
        for R in rows: p.append(("w", ("t", R, panel)))

    def column_tiles(w, k, kcol):                       # k < 0: column block 0 (no update)
        p = prog[w]
        told = False
        R0 = kcol + 1 + w
        while R0 < RT:
            R1 = R0 + K if R0 + K < RT else -1
            rows = [R0] + ([R1] if R1 >= 0 else [])
            if k >= 0: update_pair(w, k, [(R, kcol) for R in rows])      # (1 / d of panel k: the registers loaded at the start of the phase)
            if w == 1 and not told:                     # the hand-off tile alone first, the pair's other tile behind the `lead` word
                told = True
                finish_rows(w, kcol, rows[:1])
                if kcol + 1 < P:                        # the tile wave 0 carries through the NEXT diagonal tile's elimination
                    p.append(("r", ("v", kcol)))
                    update_pair(w, kcol, [(kcol + 2, kcol + 1)])
                p.append(("set", "lead", kcol + 1))
                if len(rows) > 1: finish_rows(w, kcol, rows[1:])
            else:
                finish_rows(w, kcol, rows)
            R0 += 2 * K
        if w == 1 and not told: p.append(("set", "lead", kcol + 1))

    # ---- column block 0 ----
    factor_diag(0, 0)
    for w in range(1, n_waves):
        column_tiles(w, -1, 0)
        if K > 1 and not drop_worker_barrier: prog[w].append(("arrive",)); prog[w].append(("barrier", K))
    # ---- phases ----
    for k in range(P - 1):
        # wave 0
        p0 = prog[0]
        p0.append(("r", ("v", k)))
        if not drop_lead_t_wait: p0.append(("wait", "lead_t", k))
        p0 += [("r", ("t", k + 1, k)), ("r", ("t", k + 1, k + 1)), ("w", ("t", k + 1, k + 1))]
        if not drop_lead_wait: p0.append(("wait", "lead", k + 1))
        factor_diag(0, k + 1)
        # workers
        m = RT - (k + 2)
        T = (m * (m + 1)) >> 1 if m > 0 else 0
        if T > 0 and RT > P: T -= 1
        Tw = T - 2 if T > 2 else 0
        kdeal = K - 1 if K >= 3 else K

        def tile_of(t):                                 # t-th tile of the row-major lower triangle below / right of (k+2, k+2)
            I = 0
            while t > I: t -= I + 1; I += 1
            return (k + 2 + I, k + 2 + t)
        for w in range(1, n_waves):
            p = prog[w]
            start = len(p)
            if w == 1:
                if T > 0: update_pair(1, k, [(k + 2, k + 2)] + ([(k + 3, k + 2)] if T > 1 else []))
                p.append(("set", "lead_t", k + 1))
            first = w - 2 if K >= 3 else w - 1
            u = first
            while 0 <= u < Tw:
                tiles = [tile_of(u + 2)]
                if u + kdeal < Tw: tiles.append(tile_of(u + kdeal + 2))
                update_pair(w, k, tiles)
                u += 2 * kdeal
            column_tiles(w, k, k + 1)
            # every wave loads 1 / d of panel k into registers at the start of the phase; the load matters (and is modelled) when the wave has a
            # tile to update in the phase - a wave without one never uses the registers
            if any(st[0] == "w" for st in p[start:]): p.insert(start, ("r", ("v", k)))
            if K > 1 and not drop_worker_barrier: p.append(("arrive",)); p.append(("barrier", K * (k + 2)))
    return prog, P, RT


def run(n, n_waves, rng, **broken):
    prog, P, RT = programs(n, n_waves, **broken)
    vc = VC(n_waves)
    words = {"ready": (-1, None), "lead": (0, None), "lead_t": (0, None)}          # value, released clock
    history = {"ready": [], "lead": [], "lead_t": []}                                # (value, clock) as released
    wbar = 0; wbar_clocks = []
    last_w = {}                                         # loc -> (wave, clock)
    reads = {}                                          # loc -> list of (wave, clock) since the last write
    writes = {}
    pc = [0] * n_waves
    steps = 0
    while any(pc[w] < len(prog[w]) for w in range(n_waves)):
        runnable = []
        for w in range(n_waves):
            if pc[w] >= len(prog[w]): continue
            s = prog[w][pc[w]]
            if s[0] == "wait" and words[s[1]][0] < s[2]: continue
            if s[0] == "barrier" and wbar < s[1]: continue
            runnable.append(w)
        assert runnable, "deadlock: %s" % [(w, prog[w][pc[w]]) for w in range(n_waves) if pc[w] < len(prog[w])]
        w = rng.choice(runnable)
        s = prog[w][pc[w]]
        vc.tick(w)
        if s[0] == "wait":
            for val, clk in history[s[1]]:                  # every release with a value that satisfies the wait may be the one seen: join the first
                if val >= s[2]: vc.join(w, clk); break
        elif s[0] == "set":
            assert s[2] >= words[s[1]][0], "a synchronisation word goes backwards"
            words[s[1]] = (s[2], vc.snap(w)); history[s[1]].append((s[2], vc.snap(w)))
        elif s[0] == "arrive":
            wbar += 1; wbar_clocks.append(vc.snap(w))
        elif s[0] == "barrier":
            for clk in wbar_clocks[:s[1]]: vc.join(w, clk)
        elif s[0] == "r":
            lw = last_w.get(s[1])
            assert lw is None or lw[0] == w or vc.after(lw[1], w), "race: wave %d reads %s before wave %d's write is ordered (n %d, %d waves)" % (w, s[1], lw[0], n, n_waves)
            reads.setdefault(s[1], []).append((w, vc.snap(w)))
        elif s[0] == "w":
            lw = last_w.get(s[1])
            assert lw is None or lw[0] == w or vc.after(lw[1], w), "race: wave %d overwrites %s before wave %d's write is ordered" % (w, s[1], lw[0])
            for (rw, rc) in reads.get(s[1], []):
                assert rw == w or vc.after(rc, w), "race: wave %d overwrites %s while wave %d may still read it (n %d, %d waves)" % (w, s[1], rw, n, n_waves)
            last_w[s[1]] = (w, vc.snap(w)); reads[s[1]] = []
            writes[s[1]] = writes.get(s[1], 0) + 1
        pc[w] += 1
        steps += 1
    # every tile below / on the diagonal of a panel column went through: J updates (one per earlier panel) + its factorization / substitution
    for J in range(P):
        for R in range(J, RT):
            assert writes.get(("t", R, J), 0) == J + 1, "tile (%d, %d): %d writes, expected %d" % (R, J, writes.get(("t", R, J), 0), J + 1)
    return steps


if __name__ == "__main__":
    rng = random.Random(1)
    for n_waves in (2, 4, 8):
        for n in (16, 33, 80, 88, 129, 130, 191, 192):
            steps = [run(n, n_waves, rng) for _ in range(20)]
            print("n %3d, %d waves: %d..%d steps x 20 interleavings: no race, no deadlock, every tile complete" % (n, n_waves, min(steps), max(steps)))

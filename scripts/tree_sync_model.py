"""CPU model of the hand-rolled synchronisation of the elimination-tree kernel (solve_tree_component in lfr_solve.hip) - VERDICT r4 #7.

What the kernel does without workgroup barriers, restated as small sequential programs per wave and run under RANDOM interleavings with a
vector-clock race detector (a read of data another wave wrote must be ordered after that write through a release -> acquire chain on a
synchronisation word) and a deadlock detector (some wave can always proceed):

  * factorization, "thin" schedule: a state word per column (0 not factored, 1 factored, 2 solved).  Wave gw of GW takes the columns at
    positions q of every level l with (q - level_ptr[l] + l) % GW == gw, in level order.  A column task streams through its update entries
    in order; entry i may be loaded once state[k_i] != 0 (acquire); then it writes its rows, releases, and sets state[J] = 1.
  * back substitution: the same walk top down; column J waits for state[parent] == 2, reads y of the rows of struct(J), writes y_J, sets 2.
  * team reduction: every member publishes {value, tag = reduction number} granules into the slot set (reduction & 1), polls every member's
    granule until its tag equals the reduction number, sums.  Checked: a slot is never overwritten before every member has read it (a
    member would then poll for a tag that is gone), and every member reads exactly the values of that reduction.
  * team barrier: a monotonic arrival counter; member i leaves barrier b when the counter has reached b * S.
  * team formation and the queue (solve_tree_team_kernel): workgroups become resident as CUs allow, register with their XCD, form units of T
    (the unit's state word decided once by compare-and-swap: complete by the holder of its last slot, dissolved by a waiter that finds the
    grid exhausted or the launch in SOLO mode), leaders pop the queue's head when their team is large enough and split permanently when it
    is larger, members follow their mailboxes; a waiter that sees nobody at work and nothing moving for its patience raises SOLO mode:
    any leader then takes the head alone.  Checked under random interleavings and ANY residency (CUs taken away per XCD): every component
    solved exactly once by a team of the size it asks for or, in SOLO mode, by one workgroup; the members of a unit agree on its state;
    every workgroup exits.  The variant without the state word (completeness read off the registration count) is caught.

The model takes the plan the kernel runs (capi.tree_plan: the same words).  usage: python scripts/tree_sync_model.py [n_tracks] [seed]"""
import os, random, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
import numpy as np


class Plan:
    def __init__(self, blob):
        b = np.asarray(blob, dtype=np.int64)
        self.NB = int(b[0]); self.n_levels = int(b[5])
        self.level_ptr = b[b[11]:b[11] + self.n_levels + 1].tolist()
        cd = b[b[27]:b[27] + 32 * self.NB].reshape(self.NB, 32)
        self.col_of_q = cd[:, 0].tolist()                      # column at position q of the level order
        self.ne = cd[:, 4].tolist()
        self.e_first = (cd[:, 5] - 2).tolist()
        self.parent = [int(x) if x != 0xffffffff else -1 for x in cd[:, 21]]
        self.nsub = cd[:, 16].tolist()
        colptr = b[b[8]:b[8] + self.NB + 1]
        rowsof = b[b[9]:b[9] + int(b[1])]
        self.struct = {}                                       # column J -> blocks below its diagonal
        cu = b[b[26]:]
        self.entries = []                                      # per position q: the columns k of its update entries, in order
        for q in range(self.NB):
            J = self.col_of_q[q]
            self.struct[J] = [int(x) for x in rowsof[colptr[J] + 1:colptr[J + 1]]]
            self.entries.append([int(cu[5 * (self.e_first[q] + i)]) for i in range(self.ne[q])])
        self.q_of_col = {J: q for q, J in enumerate(self.col_of_q)}

    def columns_of_wave(self, gw, GW, down=False):
        out = []
        levels = range(self.n_levels - 1, -1, -1) if down else range(self.n_levels)
        for l in levels:
            first = (gw - l % GW + GW) % GW
            out.extend(range(self.level_ptr[l] + first, self.level_ptr[l + 1], GW))
        return out


class VC:
    """vector clocks: one component per wave"""
    def __init__(self, n): self.c = [[0] * n for _ in range(n)]
    def tick(self, w): self.c[w][w] += 1
    def snapshot(self, w): return list(self.c[w])
    def join(self, w, other): self.c[w] = [max(a, b) for a, b in zip(self.c[w], other)]
    def ordered(self, stamp, w): return all(s <= c for s, c in zip(stamp, self.c[w]))


def run_factor_and_solve(plan, GW, rng, streaming=True, forget_gate=False):
    """One factorization + back substitution under a random interleaving.  Returns the number of scheduling steps; raises AssertionError on a
    race (unordered read) or a deadlock."""
    NB = plan.NB
    vc = VC(GW)
    state = [0] * NB
    state_clock = [None] * NB                                 # clock released with the state word
    rows_clock = [None] * NB                                  # clock of the write of column J's rows (factor) / of y_J (solve)
    y_clock = [None] * NB
    # programs: list of (kind, args) micro-steps per wave, generated lazily
    def program(gw):
        for q in plan.columns_of_wave(gw, GW):
            J = plan.col_of_q[q]
            ks = plan.entries[q]
            if streaming:
                for i, k in enumerate(ks):
                    if not (forget_gate and i == len(ks) - 1): yield ("wait", k, 1)      # (forget_gate: a deliberately broken protocol for the model's own test)
                    yield ("read_rows", k)
            else:                                              # round 4's rule: every child done first (kept for comparison)
                for k in ks: yield ("wait", k, 1)
                for k in ks: yield ("read_rows", k)
            yield ("write_rows", J)
            yield ("publish", J, 1)
        yield ("barrier",)
        for q in plan.columns_of_wave(gw, GW, down=True):
            J = plan.col_of_q[q]
            if plan.parent[q] >= 0: yield ("wait", plan.parent[q], 2)
            yield ("read_rows", J)                             # its own factored rows (another wave may have factored it)
            for I in plan.struct[J]: yield ("read_y", I)
            yield ("write_y", J)
            yield ("publish", J, 2)
    progs = [program(g) for g in range(GW)]
    cur = [next(p, None) for p in progs]
    at_barrier = [False] * GW
    steps = 0
    while any(c is not None for c in cur):
        runnable = []
        for g, c in enumerate(cur):
            if c is None: continue
            if c[0] == "wait" and state[c[1]] < c[2]: continue
            if c[0] == "barrier":
                at_barrier[g] = True
                if not all(at_barrier[h] or cur[h] is None for h in range(GW)): continue
            runnable.append(g)
        assert runnable, "deadlock: no wave can proceed (%s)" % [c for c in cur if c is not None][:4]
        if all(cur[g][0] == "barrier" for g in runnable) and len(runnable) == sum(c is not None for c in cur):
            for g in runnable:                                 # the workgroup / team barrier: everyone joins everyone
                for h in runnable: vc.join(g, vc.snapshot(h))
            for g in runnable:
                at_barrier[g] = False; cur[g] = next(progs[g], None)
            steps += 1
            continue
        g = rng.choice([g for g in runnable if cur[g][0] != "barrier"] or runnable)
        c = cur[g]
        vc.tick(g)
        if c[0] == "wait":
            vc.join(g, state_clock[c[1]])                      # acquire: the state word was seen
        elif c[0] == "read_rows":
            assert rows_clock[c[1]] is not None and vc.ordered(rows_clock[c[1]], g), "race: rows of column %d read by wave %d before their write is visible" % (c[1], g)
        elif c[0] == "read_y":
            assert y_clock[c[1]] is not None and vc.ordered(y_clock[c[1]], g), "race: y of column %d read by wave %d before its write is visible" % (c[1], g)
        elif c[0] == "write_rows": rows_clock[c[1]] = vc.snapshot(g)
        elif c[0] == "write_y": y_clock[c[1]] = vc.snapshot(g)
        elif c[0] == "publish":
            assert state[c[1]] == c[2] - 1, "state word of column %d published twice / out of order" % c[1]
            state[c[1]] = c[2]; state_clock[c[1]] = vc.snapshot(g)     # release
        cur[g] = next(progs[g], None)
        steps += 1
    assert all(s == 2 for s in state)
    return steps


def run_team_reductions(S, rounds, rng, parities=2):
    """`rounds` granule reductions of a team of S workgroups under a random interleaving (micro-steps: publish, one poll per member, done)."""
    slots = [[None] * S for _ in range(2)]                    # [parity][member] = (tag, value); parities = 1: the broken one-set variant
    gen = [0] * S; phase = ["publish"] * S; seen = [dict() for _ in range(S)]
    sums = [[] for _ in range(S)]
    done = [0] * S
    while min(done) < rounds:
        m = rng.choice([i for i in range(S) if done[i] < rounds])
        if phase[m] == "publish":
            gen[m] += 1
            par = gen[m] & (parities - 1)
            old = slots[par][m]
            if old is not None:                                # the granule about to be overwritten: has everyone read it?
                assert all(done[i] >= old[0] for i in range(S)), "member %d overwrites reduction %d before member(s) %s read it" % (
                    m, old[0], [i for i in range(S) if done[i] < old[0]])
            slots[par][m] = (gen[m], 1000 * gen[m] + m)
            phase[m] = "gather"; seen[m] = {}
        else:
            for j in range(S):
                g = slots[gen[m] & (parities - 1)][j]
                if j not in seen[m] and g is not None:
                    assert g[0] <= gen[m], "member %d polls for reduction %d of member %d and finds %d: it would wait forever" % (m, gen[m], j, g[0])
                    if g[0] == gen[m]: seen[m][j] = g[1]
            if len(seen[m]) == S:
                sums[m].append(sum(seen[m][j] for j in range(S)))
                done[m] = gen[m]; phase[m] = "publish"
    want = [sum(1000 * (r + 1) + j for j in range(S)) for r in range(rounds)]
    assert all(s == want for s in sums)


def run_team_barriers(S, rounds, rng):
    """the arrival-counter barrier: nobody leaves barrier b before everybody has arrived at it"""
    counter = 0; arrived = [0] * S; left = [0] * S
    while min(left) < rounds:
        m = rng.choice([i for i in range(S) if left[i] < rounds])
        if arrived[m] == left[m]:
            arrived[m] += 1; counter += 1
        elif counter >= arrived[m] * S:
            assert all(a >= arrived[m] for a in arrived), "member %d leaves barrier %d early" % (m, arrived[m])
            left[m] = arrived[m]


def run_team_formation(X, T, grid, capacity, wants, rng, patience=400, consensus=True, descending_sends=True, max_steps=2_000_000):
    """Registration, unit states, queue, splitting, SOLO mode of solve_tree_team_kernel under a random interleaving.
    X XCDs, units of T workgroups, `grid` workgroups dealt round-robin to the XCDs, capacity[x] = workgroups XCD x holds at a time
    (None: the whole grid), wants = team size each queued component asks for (descending).  A leader fills its members' mailboxes from the
    highest rank down: the leader of a sub-team hears of a split only after its members' mailboxes hold that message, so what it sends
    them next queues up behind it (descending_sends=False: the round-5 order - a fast sub-leader's message could overtake, and its member
    joined the wrong component).  Returns (components solved off-size, steps)."""
    cap = [c if c is not None else grid for c in capacity]
    fifo = [[i for i in range(grid) if i % X == x] for x in range(X)]           # not yet dispatched, in grid order
    resident = [0] * X
    count = [0] * X; total = 0; ustate = {}; solo = False; active = 0; head = 0
    mbox = {}                                                                     # (x, unit, rank) -> message
    solved = [0] * len(wants); off_size = 0
    wg = {}                                                                       # id -> dict
    team_arrivals = {}
    live = []
    now = 0

    def sig(): return (total, head)

    def members(w):
        r = range(w["L"] + 1, w["L"] + w["S"])
        return list(reversed(r)) if descending_sends else list(r)

    def watch(w):                                                                 # TeamWatch::look
        nonlocal solo
        if w["sig"] != sig() or active != 0: w["sig"] = sig(); w["t"] = now; return
        if now - w["t"] > patience: solo = True

    exited = 0
    while exited < grid:
        assert now < max_steps, "no end in sight: %d of %d workgroups left, head %d of %d, solo %s, active %d, states %s" % (
            grid - exited, grid, head, len(wants), solo, active, sorted((v["st"], v.get("S")) for v in wg.values() if v["st"] != "exit")[:12])
        now += 1
        for x in range(X):                                                         # the dispatcher: a free CU takes the next workgroup of its XCD
            while fifo[x] and resident[x] < cap[x]:
                i = fifo[x].pop(0); resident[x] += 1
                wg[i] = {"x": x, "st": "register", "t": now, "sig": None}; live.append(i)
        i = rng.choice(live)
        w = wg[i]; x = w["x"]; st = w["st"]
        if st == "register":
            w["slot"] = count[x]; count[x] += 1; total += 1
            w["unit"] = w["slot"] // T; w["rank"] = w["slot"] % T
            if consensus and w["rank"] == T - 1: ustate.setdefault((x, w["unit"]), 1)
            w["st"] = "forming"
        elif st == "forming":
            key = (x, w["unit"])
            if consensus:
                state = ustate.get(key, 0)
                if state == 0:
                    if solo or (total >= grid and count[x] < (w["unit"] + 1) * T): ustate.setdefault(key, 2)
                    else: watch(w)
                    continue
            else:                                                                  # the round-5 rule + a solo flag: no agreement
                if count[x] >= (w["unit"] + 1) * T: state = 1
                elif solo or total >= grid: state = 2
                else: watch(w); continue
            w["state"] = state
            if state == 1: w["S"] = T; w["L"] = 0
            else: w["S"] = 1; w["L"] = w["rank"]
            w["st"] = "leader" if w["rank"] == w["L"] else "member"
        elif st == "leader":
            if head >= len(wants):                                                 # queue empty: tell the members, leave
                w["send"] = [("end", None, m) for m in members(w)]; w["st"] = "sending"; w["then"] = "exit"
                continue
            want = wants[head]
            off = want > w["S"] and solo
            if want <= w["S"] or off:
                ci = head; head += 1; active += 1
                if off: off_size += 1
                if off or want == 1 and w["S"] == 1:
                    w["job"] = (ci, 1); w["st"] = "solve_alone"
                else:
                    w["send"] = [("go", (ci, want), m) for m in members(w)]; w["st"] = "sending"; w["then"] = "team"; w["job"] = (ci, want)
            else: watch(w)
        elif st == "sending":                                                      # one mailbox per step, only into an empty one
            if not w["send"]:
                if w["then"] == "exit": w["st"] = "exit"
                else:
                    ci, want = w["job"]; w["S"] = want; w["st"] = "team_solve"; w["arrive"] = (x, w["unit"], w["L"], ci)
                continue
            kind, payload, m = w["send"][0]
            if mbox.get((x, w["unit"], m)) is None:
                tgt = [v for v in wg.values() if v["x"] == x and v.get("unit") == w["unit"] and v.get("rank") == m]
                assert tgt and tgt[0].get("state", 1) == 1 and tgt[0]["st"] != "exit", "message for a workgroup that is not in the unit (unit %s rank %d)" % ((x, w["unit"]), m)
                mbox[(x, w["unit"], m)] = (kind, payload); w["send"].pop(0)
        elif st == "member":
            msg = mbox.get((x, w["unit"], w["rank"]))
            if msg is None: continue
            mbox[(x, w["unit"], w["rank"])] = None
            if msg[0] == "end": w["st"] = "exit"
            else:
                ci, s_new = msg[1]
                L_new = w["L"] + ((w["rank"] - w["L"]) // s_new) * s_new
                mine = L_new == w["L"]
                w["S"] = s_new; w["L"] = L_new
                if not mine: w["st"] = "leader" if w["rank"] == L_new else "member"
                else:
                    lead = [v for v in wg.values() if v["x"] == x and v.get("unit") == w["unit"] and v.get("rank") == L_new][0]
                    assert lead.get("job", (None,))[0] == ci, "member %d of unit %s joins component %d, its leader %d works on %s" % (w["rank"], (x, w["unit"]), ci, L_new, lead.get("job"))
                    w["st"] = "team_solve"; w["arrive"] = (x, w["unit"], w["L"], ci)
        elif st == "solve_alone":
            ci, _ = w["job"]; solved[ci] += 1; active -= 1; w["st"] = "leader"
        elif st == "team_solve":                                                   # the team's barrier: everybody arrives, then the leader reports
            key = w["arrive"]
            arrived = w.setdefault("barrier", None)
            if arrived is None:
                team_arrivals[key] = team_arrivals.get(key, 0) + 1; w["barrier"] = True
            elif team_arrivals[key] >= w["S"]:
                w["barrier"] = None
                if w["rank"] == w["L"]:
                    ci = key[3]; solved[ci] += 1; active -= 1
                    assert w["S"] == wants[ci], "component %d asks for %d workgroups, solved by %d" % (ci, wants[ci], w["S"])
                    w["st"] = "leader"
                else: w["st"] = "member"
        if w["st"] == "exit":
            live.remove(i); resident[x] -= 1; exited += 1
    assert solved == [1] * len(wants), "components solved %s times" % solved
    return off_size, now



if __name__ == "__main__":
    from lfr_amd import capi, synthetic
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from tree_plan_stats import component_words
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ma = synthetic.capsized_sparse(n_tracks=nt, seed=7)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    comps = sorted(component_words(ma, p), key=lambda c: -c[0])[:3]
    for nv, w in comps:
        blob, info = capi.tree_plan(nv, w)
        plan = Plan(blob)
        for GW in (8, 16, 32, 64):
            steps = [run_factor_and_solve(plan, GW, rng) for _ in range(3)]
            print("n_var %d, %d columns, %d levels, %2d waves: %s scheduling steps, no race, no deadlock" % (nv, plan.NB, plan.n_levels, GW, steps))
    for S in (2, 4, 8):
        for _ in range(20):
            run_team_reductions(S, 50, rng); run_team_barriers(S, 50, rng)
    print("team reductions / barriers: 20 random interleavings each for teams of 2, 4, 8: ok")
    wants = [8] * 3 + [4] * 6 + [2] * 10 + [1] * 12
    for capacity in ([None] * 4, [8, 8, 8, 8], [7, 7, 7, 7], [1, 1, 1, 1], [8, 3, 1, 5]):
        res = [run_team_formation(4, 8, 64, capacity, wants, rng) for _ in range(5)]
        print("team formation, 64 workgroups on 4 XCDs, residency %s: components solved off-size %s" % (capacity, [r[0] for r in res]))

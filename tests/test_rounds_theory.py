"""The device graph stage's parallel rule, simulated on the CPU (numpy): "retire what can never be accepted, accept the match
that is the earliest bidder at both of its roots", run over PREFIX BLOCKS of the ordered match list, against the sequential
greedy rule of solve.cc:499-523 - same unions, same root identities (union by size, ties: root2 under root1), for any block
schedule.  The HIP kernels (lfr_graphstage.hip: k_round_eval / k_round_accept) are checked against the host stage on the GPU;
this pins the argument they rest on where no GPU is needed."""
import numpy as np
import pytest


def sequential(n, img, a, b):
    parent = np.arange(n); size = np.ones(n, np.int64)
    images = [{int(img[i])} for i in range(n)]

    def root(x):
        while parent[x] != x:
            x = parent[x]
        return x
    for x, y in zip(a, b):
        r1, r2 = root(x), root(y)
        if r1 == r2 or images[r1] & images[r2]:
            continue
        big, small = (r2, r1) if size[r1] < size[r2] else (r1, r2)         # ties: root2 under root1
        parent[small] = big; size[big] += size[small]; images[big] |= images[small]
    return np.array([root(i) for i in range(n)])


def rounds_in_blocks(n, img, a, b, blocks):
    parent = np.arange(n); size = np.ones(n, np.int64)
    n_img = int(img.max()) + 1
    bits = np.zeros((n, n_img), bool); bits[np.arange(n), img] = True
    n_rounds = 0
    for lo, hi in blocks:
        pend = np.arange(lo, hi)
        while len(pend):
            while True:                                                    # full path compression = find for everybody
                gp = parent[parent]
                if (gp == parent).all():
                    break
                parent = gp
            ra, rb = parent[a[pend]], parent[b[pend]]
            keep = (ra != rb) & ~(bits[ra] & bits[rb]).any(axis=1)         # monotone: these verdicts never change
            pend, ra, rb = pend[keep], ra[keep], rb[keep]
            if not len(pend):
                break
            n_rounds += 1
            first = np.full(n, np.iinfo(np.int64).max)
            np.minimum.at(first, ra, pend); np.minimum.at(first, rb, pend)
            win = (first[ra] == pend) & (first[rb] == pend)
            wa, wb = ra[win], rb[win]
            big = np.where(size[wa] < size[wb], wb, wa); small = np.where(size[wa] < size[wb], wa, wb)
            parent[small] = big; size[big] += size[small]; bits[big] |= bits[small]
            pend = pend[~win]
    while True:
        gp = parent[parent]
        if (gp == parent).all():
            return gp, n_rounds
        parent = gp


@pytest.mark.parametrize("seed", range(40))
def test_rounds_over_prefix_blocks_equal_the_sequential_rule(seed):
    rng = np.random.default_rng(900 + seed)
    n_img = int(rng.integers(2, 9))
    n = int(rng.integers(4, 120))
    img = rng.integers(0, n_img, n)
    m = int(rng.integers(1, 6 * n))
    a = rng.integers(0, n, m); b = rng.integers(0, n, m)                   # duplicates, same-image pairs, self matches included
    want = sequential(n, img, a, b)
    schedules = [[(0, m)], [(i, min(m, i + 3)) for i in range(0, m, 3)]]
    cuts = sorted(set(rng.integers(0, m + 1, 4).tolist()) | {0, m})
    schedules.append(list(zip(cuts[:-1], cuts[1:])))
    for blocks in schedules:
        got, _ = rounds_in_blocks(n, img, a, b, blocks)
        assert (got == want).all(), (seed, blocks)


def test_prefix_blocks_on_long_tracks_need_only_a_few_more_rounds():
    """Dense long tracks (every pair of a track matched, a few wrong matches) in the doubling block schedule of the device
    stage: same result, and the number of rounds (= launches + read-backs) stays within a small factor of the single list's."""
    rng = np.random.default_rng(5)
    n_img, n_tracks = 24, 30
    img = []; a = []; b = []
    for t in range(n_tracks):
        ids = len(img) + np.arange(n_img)
        img += list(range(n_img))
        i, j = np.triu_indices(n_img, 1)
        a += ids[i].tolist(); b += ids[j].tolist()
    n = len(img)
    wrong = rng.integers(0, n, (200, 2))
    a += wrong[:, 0].tolist(); b += wrong[:, 1].tolist()
    a = np.array(a); b = np.array(b); img = np.array(img)
    order = rng.permutation(len(a)); a = a[order]; b = b[order]
    m = len(a)
    want = sequential(n, img, a, b)

    def n_rounds(blocks):
        got, rounds = rounds_in_blocks(n, img, a, b, blocks)
        assert (got == want).all()
        return rounds
    blocks = []; lo, size = 0, 2 * n
    while lo < m:
        blocks.append((lo, min(m, lo + size))); lo += size; size *= 2
    r_single, r_blocks = n_rounds([(0, m)]), n_rounds(blocks)
    assert r_blocks <= 2 * r_single + 8                                    # a few rounds more, not a different order of magnitude


# ---- round 4: the two shortcuts of the device graph stage, restated on the CPU ----
def _sim_key(s):
    """order-preserving float32 -> uint32 (lfr_graphstage.hip: sim_key)"""
    s = np.where(s == 0, np.float32(0), s).astype(np.float32)
    b = s.view(np.uint32)
    return np.where(b & 0x80000000, ~b, b | 0x80000000).astype(np.uint32)


def _components(n, a, b):
    lab = np.arange(n)
    while True:
        m = np.minimum(lab[a], lab[b])
        new = lab.copy()
        np.minimum.at(new, a, m); np.minimum.at(new, b, m)
        new = new[new]
        if (new == lab).all():
            return lab
        lab = new


@pytest.mark.parametrize("seed", range(8))
def test_one_sort_plus_tie_fix_equals_the_three_stable_sorts(seed):
    """k_cc_sim_keys + one radix sort + k_tie_fix against the round-3 scheme (stable sorts by n2, by (sim, n1), by connected component),
    which is the reference's order (descending (sim, n1, n2), solve.cc:489) inside every component: heavy ties, duplicated matches."""
    rng = np.random.default_rng(400 + seed)
    n, M = 60, 900
    a = rng.integers(0, n, M).astype(np.uint32); b = rng.integers(0, n, M).astype(np.uint32)
    keep = a != b
    a, b = a[keep], b[keep]
    dup = rng.integers(0, len(a), 60)                                   # duplicated matches: equal (sim, n1, n2) triples
    sim = rng.choice(np.array([0.25, 0.5, 0.5, 0.75, 0.9], np.float32), len(a))
    a, b, sim = np.r_[a, a[dup]], np.r_[b, b[dup]], np.r_[sim, sim[dup]]
    M = len(a)
    cc = _components(n, a, b)[a]
    ids = np.arange(M)
    # three stable sorts (ascending complemented keys = descending order)
    o = ids[np.argsort((n - 1 - b)[ids], kind="stable")]
    hi = ((~_sim_key(sim)).astype(np.uint64) << 8) | (n - 1 - a).astype(np.uint64)
    o = o[np.argsort(hi[o], kind="stable")]
    three = o[np.argsort(cc[o], kind="stable")]
    # one sort by (component | ~sim), then the runs of equal keys ordered by (n1 desc, n2 desc, id asc)
    key = (cc.astype(np.uint64) << 32) | (~_sim_key(sim)).astype(np.uint64)
    one = ids[np.argsort(key, kind="stable")]
    ks = key[one]
    lo = 0
    while lo < M:
        hi_ = lo
        while hi_ < M and ks[hi_] == ks[lo]:
            hi_ += 1
        run = sorted(one[lo:hi_], key=lambda m: (-int(a[m]), -int(b[m]), int(m)))
        one[lo:hi_] = run
        lo = hi_
    assert (one == three).all()
    # and that IS the reference's order inside a component
    ref = sorted(ids, key=lambda m: (int(cc[m]), -float(sim[m]), -int(a[m]), -int(b[m]), int(m)))
    assert (three == np.array(ref)).all()


@pytest.mark.parametrize("seed", range(6))
def test_meta_components_are_the_connected_components_of_the_match_graph(seed):
    """k_cc_min_track: whatever partition of the nodes into tracks the greedy rule produces (every track is connected through its accepted
    matches, every match joins two nodes of one connected component), the components of the track meta-graph - tracks joined by the
    matches between them, solve.cc:262-290 - are the match graph's connected components, named by their smallest track."""
    rng = np.random.default_rng(500 + seed)
    n, M, n_img = 120, 260, 9
    img = rng.integers(0, n_img, n)
    a = rng.integers(0, n, M); b = rng.integers(0, n, M)
    keep = a != b
    a, b = a[keep], b[keep]
    root = sequential(n, img, a, b)                                      # tracks of the sequential rule (roots as ids)
    _, track = np.unique(root, return_inverse=True)
    T = track.max() + 1
    meta = _components(T, track[a], track[b])                            # union-find over the inter-track matches (intra-track ones are no-ops)
    cc = _components(n, a, b)
    min_track = np.full(n, T, np.int64)
    np.minimum.at(min_track, cc, track)                                  # smallest track per connected component ...
    label = np.empty(T, np.int64)
    label[track] = min_track[cc]                                         # ... is the label of every track in it
    assert (label == meta).all()

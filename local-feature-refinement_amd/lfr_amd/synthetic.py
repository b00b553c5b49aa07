"""Synthetic match-graph generator G(seed, n_images, n_tracks, ...) of SURVEY.md §8(d).

The reference ships no data and no generator; real match graphs come from its
two-view network (``two-view-refinement/compute_match_graph.py:163-187``), whose
output layout this generator reproduces: per match two 3x3 grids of (di, dj)
flow vectors, ``disp2`` = flow image1->image2 and ``disp1`` = flow image2->image1
(``compute_match_graph.py:179-187``), one ImagePair per ordered image pair.

Everything stored is float32 (wire precision, ``types.proto:13,16-17``); the
random stream is ``numpy.random.Generator(PCG64(seed))`` so the same arguments
give the same graph on every machine.
"""
from dataclasses import dataclass

import numpy as np

GRID = np.array([(-0.5 + 0.5 * i, -0.5 + 0.5 * j) for i in range(3) for j in range(3)],
                dtype=np.float64)  # grid index = i*3+j  (solve.cc:461-465)


@dataclass
class MatchArrays:
    """Flat (SoA) form of a MatchingFile: what ``lfr_graph_from_arrays`` takes."""
    image_names: list          # n_images strings
    facts: np.ndarray          # float32[n_images]
    pair_img1: np.ndarray      # int32[P]   image index of image_name1
    pair_img2: np.ndarray      # int32[P]
    pair_off: np.ndarray       # int64[P+1] matches of pair p are [pair_off[p], pair_off[p+1])
    feat1: np.ndarray          # uint32[M]
    feat2: np.ndarray          # uint32[M]
    sim: np.ndarray            # float32[M]
    disp1: np.ndarray          # float32[M,9,2]  flow 2->1 (rides edge node2->node1, solve.cc:478)
    disp2: np.ndarray          # float32[M,9,2]  flow 1->2 (rides edge node1->node2, solve.cc:477)

    @property
    def n_matches(self):
        return int(self.feat1.shape[0])

    def to_pairs(self):
        """Expand into the dict form of :mod:`lfr_amd.wire` (small inputs only)."""
        pairs = []
        for p in range(len(self.pair_img1)):
            lo, hi = int(self.pair_off[p]), int(self.pair_off[p + 1])
            ms = []
            for m in range(lo, hi):
                ms.append({
                    "feature_idx1": int(self.feat1[m]), "feature_idx2": int(self.feat2[m]),
                    "similarity": float(self.sim[m]),
                    "disp1": [(float(a), float(b)) for a, b in self.disp1[m]],
                    "disp2": [(float(a), float(b)) for a, b in self.disp2[m]],
                })
            i1, i2 = int(self.pair_img1[p]), int(self.pair_img2[p])
            pairs.append({"image_name1": self.image_names[i1], "fact1": float(self.facts[i1]),
                          "image_name2": self.image_names[i2], "fact2": float(self.facts[i2]),
                          "matches": ms})
        return pairs


def pairs_to_arrays(pairs):
    """Inverse of :meth:`MatchArrays.to_pairs` for well-formed inputs (9 grid points)."""
    names, idx, facts = [], {}, []

    def _img(n, f):
        if n not in idx:
            idx[n] = len(names)
            names.append(n)
            facts.append(f)
        return idx[n]

    p1, p2, off, f1, f2, sim, d1, d2 = [], [], [0], [], [], [], [], []
    for p in pairs:
        p1.append(_img(p["image_name1"], p["fact1"]))
        p2.append(_img(p["image_name2"], p["fact2"]))
        for m in p["matches"]:
            f1.append(m["feature_idx1"])
            f2.append(m["feature_idx2"])
            sim.append(m["similarity"])
            a = np.zeros((9, 2), np.float32)
            b = np.zeros((9, 2), np.float32)
            for k, d in enumerate(m["disp1"]):
                a[k] = d
            for k, d in enumerate(m["disp2"]):
                b[k] = d
            d1.append(a)
            d2.append(b)
        off.append(len(f1))
    M = len(f1)
    return MatchArrays(names, np.asarray(facts, np.float32), np.asarray(p1, np.int32),
                       np.asarray(p2, np.int32), np.asarray(off, np.int64),
                       np.asarray(f1, np.uint32), np.asarray(f2, np.uint32),
                       np.asarray(sim, np.float32),
                       np.asarray(d1, np.float32).reshape(M, 9, 2),
                       np.asarray(d2, np.float32).reshape(M, 9, 2))


def _distinct_images(rng, n_rows, L, n_images):
    """n_rows x L matrix of distinct image ids per row, uniform."""
    if n_rows == 0:
        return np.zeros((0, L), np.int64)
    if 4 * L * L > n_images:            # collisions likely: random keys + argsort
        keys = rng.random((n_rows, n_images), dtype=np.float32)
        return np.argsort(keys, axis=1, kind="stable")[:, :L].astype(np.int64)
    out = rng.integers(0, n_images, size=(n_rows, L))       # rare collisions: redraw those rows
    while True:
        s = np.sort(out, axis=1)
        bad = np.nonzero((s[:, 1:] == s[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            return out.astype(np.int64)
        out[bad] = rng.integers(0, n_images, size=(bad.size, L))


def generate(seed, n_images, n_tracks, len_dist="poisson", len_lo=None, len_hi=None,
             eps_out=0.0, sigma_p=0.15, sigma_noise=0.02, sigma_A=0.05, sim_lo=0.8, sim_hi=1.0,
             fact=1.0, track_degree=None, chain_links=0, ratio_sims=False, dup_frac=0.0):
    """Generate a synthetic match graph.

    len_dist = "poisson": L = min(n_images, 2 + Poisson(4)) (mean 6);
    len_dist = "uniform": L ~ U{len_lo..len_hi}.
    All L(L-1)/2 pairs inside a track are matched; ``eps_out`` adds that fraction
    of extra wrong matches between nodes of different tracks (different images).

    Real match graphs are sparser than that: a feature is only matched along the image pairs
    the match list holds.  ``track_degree=k``: inside a track node i is matched to its next k/2 track neighbours
    (a ring lattice: every node has ~k partners, the track stays connected) instead of to all of them.
    ``chain_links=c``: beside the ``eps_out`` random wrong matches, track t gets c wrong matches to track t+1 - the
    tracks then form one long chain-like connected component that the size cap cuts into cap-sized pieces
    (components of #images nodes made of many short tracks: the sparse systems real cap-sized components are).
    ``ratio_sims``: similarities like the ratio-test matcher's (two-view-refinement/feature_matchers.py:42:
    1 - d1/d2, spread over (0, 1) instead of U(sim_lo, sim_hi)).  ``dup_frac``: that fraction of the matches
    is emitted twice (the reference keeps duplicates, solve.cc:476-478).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if len_dist == "poisson":
        L = np.minimum(n_images, 2 + rng.poisson(4.0, size=n_tracks)).astype(np.int64)
    elif len_dist == "uniform":
        L = rng.integers(len_lo, len_hi + 1, size=n_tracks).astype(np.int64)
        L = np.minimum(L, n_images)
    else:
        raise ValueError(len_dist)

    # nodes, in track order
    node_off = np.zeros(n_tracks + 1, np.int64)
    np.cumsum(L, out=node_off[1:])
    N = int(node_off[-1])
    node_img = np.empty(N, np.int64)
    for ell in np.unique(L):
        rows = np.nonzero(L == ell)[0]
        imgs = _distinct_images(rng, rows.size, int(ell), n_images)
        dst = node_off[rows][:, None] + np.arange(ell)[None, :]
        node_img[dst.ravel()] = imgs.ravel()
    # feature_idx = running counter per image, in node order
    order = np.argsort(node_img, kind="stable")
    sorted_img = node_img[order]
    first = np.r_[True, sorted_img[1:] != sorted_img[:-1]]
    start = np.maximum.accumulate(np.where(first, np.arange(N), 0))
    node_feat = np.empty(N, np.int64)
    node_feat[order] = np.arange(N) - start
    node_p = np.clip(rng.normal(0.0, sigma_p, size=(N, 2)), -0.45, 0.45)
    node_track = np.repeat(np.arange(n_tracks), L)

    # true matches: all pairs inside a track (or a ring lattice of degree track_degree)
    a_list, b_list = [], []
    for ell in np.unique(L):
        rows = np.nonzero(L == ell)[0]
        iu, ju = np.triu_indices(int(ell), k=1)
        if track_degree is not None and ell > track_degree + 1:
            half = max(1, int(track_degree) // 2)
            d = ju - iu
            keep = (d <= half) | (d >= ell - half)           # ring distance <= half
            iu, ju = iu[keep], ju[keep]
        base = node_off[rows][:, None]
        a_list.append((base + iu[None, :]).ravel())
        b_list.append((base + ju[None, :]).ravel())
    a = np.concatenate(a_list) if a_list else np.zeros(0, np.int64)
    b = np.concatenate(b_list) if b_list else np.zeros(0, np.int64)
    o = np.lexsort((b, a))
    a, b = a[o], b[o]
    n_true = a.size
    wrong = np.zeros(n_true, bool)

    # wrong matches between different tracks
    n_out = int(round(eps_out * n_true))
    wa = wb = np.zeros(0, np.int64)
    if n_out > 0:
        wa = rng.integers(0, N, size=n_out)
        wb = rng.integers(0, N, size=n_out)
        while True:
            bad = np.nonzero((node_track[wa] == node_track[wb]) | (node_img[wa] == node_img[wb]))[0]
            if bad.size == 0:
                break
            wa[bad] = rng.integers(0, N, size=bad.size)
            wb[bad] = rng.integers(0, N, size=bad.size)
    if chain_links > 0 and n_tracks > 1:
        # track t -> track t + 1: a random node of each (different images)
        t0 = np.repeat(np.arange(n_tracks - 1), chain_links)
        ca = node_off[t0] + rng.integers(0, 1 << 30, size=t0.size) % L[t0]
        cb = node_off[t0 + 1] + rng.integers(0, 1 << 30, size=t0.size) % L[t0 + 1]
        for _ in range(64):
            bad = np.nonzero(node_img[ca] == node_img[cb])[0]
            if bad.size == 0:
                break
            cb[bad] = node_off[t0[bad] + 1] + rng.integers(0, 1 << 30, size=bad.size) % L[t0[bad] + 1]
        ok = node_img[ca] != node_img[cb]
        wa = np.concatenate([wa, ca[ok]])
        wb = np.concatenate([wb, cb[ok]])
    n_out = int(wa.size)
    if n_out > 0:
        a = np.concatenate([a, wa])
        b = np.concatenate([b, wb])
        wrong = np.concatenate([wrong, np.ones(n_out, bool)])
    if dup_frac > 0.0 and a.size:
        pick = np.nonzero(rng.random(a.size) < dup_frac)[0]
        a = np.concatenate([a, a[pick]])
        b = np.concatenate([b, b[pick]])
        wrong = np.concatenate([wrong, wrong[pick]])

    # orient so image1 < image2 (names are "%06d.png": index order == name order)
    swap = node_img[a] > node_img[b]
    a, b = np.where(swap, b, a), np.where(swap, a, b)
    M = a.size

    grid32 = GRID.astype(np.float32)

    def _flow(src, dst):
        base = (node_p[dst] - node_p[src]).astype(np.float32)                    # (M,2)
        if n_out > 0:
            base = np.where(wrong[:, None], rng.normal(0.0, 0.3, size=(M, 2)).astype(np.float32), base)
        A = rng.standard_normal(size=(M, 2, 2), dtype=np.float32) * np.float32(sigma_A)
        out = rng.standard_normal(size=(M, 9, 2), dtype=np.float32)
        out *= np.float32(sigma_noise)
        out += base[:, None, :]
        out += A[:, None, :, 0] * grid32[None, :, 0, None]                       # A . u, u = grid point
        out += A[:, None, :, 1] * grid32[None, :, 1, None]
        return out

    disp2 = _flow(a, b)   # flow 1 -> 2
    disp1 = _flow(b, a)   # flow 2 -> 1
    if ratio_sims:       # 1 - d1/d2 of a ratio test: most mass at small values, correct matches higher than wrong ones
        sim = np.where(wrong, rng.beta(1.2, 6.0, size=M), rng.beta(2.0, 3.0, size=M))
        sim = np.clip(sim, 1e-3, 0.999).astype(np.float32)
    else:
        sim = rng.uniform(sim_lo, sim_hi, size=M).astype(np.float32)

    # group by image pair, lexicographic, stable
    i1, i2 = node_img[a], node_img[b]
    o = np.lexsort((np.arange(M), i2, i1))
    a, b, i1, i2, sim, disp1, disp2 = a[o], b[o], i1[o], i2[o], sim[o], disp1[o], disp2[o]
    key = i1 * n_images + i2
    first = np.r_[True, key[1:] != key[:-1]] if M else np.zeros(0, bool)
    starts = np.nonzero(first)[0]
    pair_off = np.r_[starts, M].astype(np.int64)
    names = ["%06d.png" % i for i in range(n_images)]
    return MatchArrays(names, np.full(n_images, fact, np.float32),
                       i1[starts].astype(np.int32), i2[starts].astype(np.int32), pair_off,
                       node_feat[a].astype(np.uint32), node_feat[b].astype(np.uint32),
                       sim, disp1, disp2)


# named configurations of BASELINE.json / SURVEY.md §8(d)
def config2():
    """100k tracks, 64 images, mean length 6 (~3.4M directed edges)."""
    return generate(seed=1, n_images=64, n_tracks=100_000)


def config4(n_tracks=147_000, seed=2):
    """Headline: 1344 images, ~5.0M directed edges."""
    return generate(seed=seed, n_images=1344, n_tracks=n_tracks)


def config1_standin():
    """Stand-in for BASELINE config 1 (Fountain, 11 images, SIFT matches): the real match graph needs the
    dataset + the two-view network, neither available here.  11 images, 6000 tracks of length 2..11,
    1 % wrong matches (-> multi-track components at the 11-node cap, inter-track Tukey edges)."""
    return generate(seed=11, n_images=11, n_tracks=6000, len_dist="uniform", len_lo=2, len_hi=11, eps_out=0.01,
                    sigma_noise=0.04)


def config3_standin():
    """Stand-in for BASELINE config 3 (Herzjesu, 8 images, SuperPoint matches): 8 images, 3000 tracks."""
    return generate(seed=13, n_images=8, n_tracks=3000, len_dist="uniform", len_lo=2, len_hi=8, eps_out=0.005,
                    sigma_noise=0.03, sim_lo=0.6)


def capsized_sparse(n_images=1344, n_tracks=40_000, seed=7, track_degree=4, chain_links=1, eps_out=0.0005):
    """Cap-sized SPARSE components: 1344 images, short tracks (mean 6) matched along a ring lattice of
    degree 4, every track linked to the next by a wrong match plus a few random wrong matches: the tracks form giant connected
    meta components which the size cap (#images nodes, solve.cc:586) cuts into pieces of up to 1344 nodes - ~2700-row systems
    whose matrices are tree-plus-few-cycles sparse (what real config-4-scale data gives the solver)."""
    return generate(seed=seed, n_images=n_images, n_tracks=n_tracks, track_degree=track_degree, chain_links=chain_links,
                    eps_out=eps_out, ratio_sims=True)


def config5():
    """Long-track stress: 96 images, L~U{48..96}, 2000 tracks, 2% wrong matches."""
    return generate(seed=3, n_images=96, n_tracks=2000, len_dist="uniform", len_lo=48, len_hi=96,
                    eps_out=0.02)

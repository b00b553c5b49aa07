"""Generates tests/golden/*.pb + *.npz.

The reference ships no fixtures and cannot be built or imported here (Ceres/COLMAP/protoc
absent), so these vectors are produced by the readable restatement oracle/lfr_ref.py
(pure Python + numpy) from seeded synthetic match graphs.  They pin (a) the wire format,
(b) the graph-stage labels, (c) the per-component LM outcome (positions, iteration counts,
termination, one full per-iteration trace) for the C oracle and for the HIP kernels.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from lfr_amd import synthetic, wire  # noqa: E402
import lfr_ref  # noqa: E402

CASES = {
    "clean": dict(seed=101, n_images=16, n_tracks=14),
    "outliers": dict(seed=102, n_images=120, n_tracks=18, eps_out=0.03),
    "noisy": dict(seed=103, n_images=20, n_tracks=12, sigma_noise=0.25),
    "bounds": dict(seed=104, n_images=20, n_tracks=12, sigma_p=0.6, sigma_noise=0.1),
    # many line-search contractions (cubic and quintic interpolation), rejected steps, active bounds
    "linesearch": dict(seed=341, n_images=20, n_tracks=8, sigma_p=0.7, sigma_noise=0.15),
    "linesearch2": dict(seed=289, n_images=20, n_tracks=8, sigma_noise=0.3),
    # round 3: real-shaped inputs - every node matched to ~4-5 track neighbours instead of all pairs,
    # ratio-test-like similarities (feature_matchers.py:42), duplicated matches (solve.cc:476-478 keeps them)
    "sparse_ratio": dict(seed=511, n_images=120, n_tracks=10, track_degree=4, ratio_sims=True, dup_frac=0.05, eps_out=0.02),
    "sparse_long": dict(seed=512, n_images=160, n_tracks=4, len_dist="uniform", len_lo=20, len_hi=40, track_degree=5, eps_out=0.004,
                        ratio_sims=True, dup_frac=0.03),        # long sparse tracks (workgroup classes), a few joined by wrong matches; no component above the cap
    # round 4: the control flow of the large-component paths, iteration by iteration -
    # three ~105-node ring-lattice tracks: ~208-row sparse systems, the elimination-tree kernel (> 192 rows; the C oracle's envelope path)
    "sparse_tree": dict(seed=613, n_images=120, n_tracks=3, len_dist="uniform", len_lo=100, len_hi=110, track_degree=4, ratio_sims=True, dup_frac=0.02),
    # 14 images: multi-track components above the size cap, cut by the reference's recursion (solve.cc:185-250, 311-364) around the
    # product's two-way primitive (lfr_bisect_graph - Graclus cannot be restated); the pieces are then solved like any component
    "cut": dict(seed=612, n_images=14, n_tracks=40, eps_out=0.04),
}
NEEDS_BISECT = {"cut"}


def main():
    only = sys.argv[1:]
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        ma = synthetic.generate(**kw)
        pairs = ma.to_pairs()
        bisect = None
        if name in NEEDS_BISECT:
            from lfr_amd import capi
            bisect = capi.bisect_graph
        open(os.path.join(HERE, name + ".pb"), "wb").write(wire.encode_matching_file(pairs))
        out = {}
        for variant in ("ceres1", "ceres2"):
            res = lfr_ref.solve_pairs(pairs, tukey_variant=variant, want_trace=True, bisect_fn=bisect)
            nc = res["n_components"]
            its = np.zeros(nc, np.int32)
            term = np.zeros(nc, np.int32)
            cost = np.zeros(nc)
            nls = np.zeros(nc, np.int32)
            for c, info in res["infos"].items():
                its[c], term[c], cost[c], nls[c] = info["iterations"], info["termination"], info["final_cost"], info["n_ls_evals"]
            worst = max(res["infos"], key=lambda c: res["infos"][c]["iterations"])
            tr = [[t.get("it", -1), t.get("cost", np.nan), t.get("cost_cand", np.nan), t.get("rel", np.nan),
                   t.get("radius", np.nan), np.nan if t.get("alpha") is None else t.get("alpha", np.nan)]
                  for t in res["infos"][worst]["trace"] if "stop" not in t and not t.get("invalid")]
            out.update({"positions_" + variant: res["positions"], "iterations_" + variant: its,
                        "termination_" + variant: term, "final_cost_" + variant: cost, "n_ls_evals_" + variant: nls,
                        "trace_component_" + variant: np.int64(worst), "trace_" + variant: np.asarray(tr, float)})
        out.update(track=np.asarray(res["track"], np.int64), is_root=np.asarray(res["is_root"], bool),
                   comp=np.asarray(res["comp"], np.int64),
                   node_feat=np.asarray([k[1] for k in res["node_key"]], np.int64),
                   node_image=np.asarray([k[0] for k in res["node_key"]]))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        sol = lfr_ref.solution_images(lfr_ref.solve_pairs(pairs, bisect_fn=bisect))
        open(os.path.join(HERE, name + ".solution.pb"), "wb").write(wire.encode_solution_file(sol))
        print(name, "nodes", res["n_nodes"], "comps", res["n_components"], "max iters", int(out["iterations_ceres1"].max()),
              "ls evals > iters:", int((out["n_ls_evals_ceres1"] > out["iterations_ceres1"]).sum()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-command parity check against a REAL reference build of `solve` (multi-view-refinement/solve.cc), for
whoever has one: the reference's arithmetic is Ceres + COLMAP/Graclus, which this image cannot build
(DESIGN.md §3), so the oracle of this repo is unpinned until this script has been run somewhere.

    python scripts/compare_with_reference.py --reference /path/to/reference/build/solve --matches_file M.pb
    python scripts/compare_with_reference.py --reference ... --synthetic config2      # generate M.pb first

What it does (solve.cc:638,641,644-679):
  1. runs the reference binary on the MatchingFile, parses its `Solver time:` / `Total time:` lines and decodes its
     SolutionFile;
  2. runs this repo's drop-in (`multi-view-refinement/build/solve`) on the same file once per Tukey flavour
     (LFR_TUKEY_VARIANT=ceres1|ceres2: Ceres changed TukeyLoss by a factor 2 between 1.14 and 2.0);
  3. reports max |delta| in px (displacement * fact * 16, colmap_utils.py:135-136) per flavour, names the flavour
     that matches, and says whether the 1e-4 px bar of BASELINE.md holds;
  4. when the reference's components exceeded the size cap (its Graclus cut is not reproducible), pass
     --components_file (raw little-endian int64[n_nodes] component ids dumped from the reference run) and the
     drop-in uses them instead of its own bisection (LFR_COMPONENTS_FILE).
Also importable: bench.py calls compare() for its `$LFR_REFERENCE_SOLVE` leg.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))

PX_PER_UNIT = 16.0        # colmap_utils.py:136 (times the image's `fact`)
BAR_PX = 1e-4


def parse_stdout(text):
    out = {}
    for key, pat in (("solver_ms", r"Solver time:\s*([0-9.eE+-]+)\s*ms"), ("total_ms", r"Total time:\s*([0-9.eE+-]+)\s*ms"),
                     ("graph_cut_ms", r"Graph-cut time:\s*([0-9.eE+-]+)\s*ms"), ("n_nodes", r"# graph nodes:\s*(\d+)"),
                     ("n_edges", r"# graph edges:\s*(\d+)"), ("n_tracks", r"# tracks:\s*(\d+)"),
                     ("n_components", r"# components:\s*(\d+)"), ("max_component_size", r"max component size:\s*(\d+)")):
        m = re.search(pat, text)
        if m:
            out[key] = float(m.group(1)) if "ms" in key else int(m.group(1))
    return out


def load_solution(path):
    """{(image_name, feature_idx): (di, dj)} and {image_name: fact} of a SolutionFile (types.proto:30-46)."""
    from lfr_amd import wire
    images = wire.decode_solution_file(open(path, "rb").read())
    pos, fact = {}, {}
    for im in images:
        fact[im["image_name"]] = im.get("fact", 0.0)
        for fidx, di, dj in im["displacements"]:
            pos[(im["image_name"], int(fidx))] = (float(di), float(dj))
    return pos, fact


def diff_px(a, fact_a, b):
    """max / rms |delta| in px over the union of keys (a missing key counts as zero displacement, as the consumer
    treats it: colmap_utils.py:127-128)."""
    worst, sq, n, worst_key = 0.0, 0.0, 0, None
    for key in set(a) | set(b):
        da, db = a.get(key, (0.0, 0.0)), b.get(key, (0.0, 0.0))
        f = fact_a.get(key[0], 1.0) or 1.0
        for x, y in zip(da, db):
            e = abs(x - y) * f * PX_PER_UNIT
            sq += e * e
            n += 1
            if e > worst:
                worst, worst_key = e, key
    return worst, (sq / max(n, 1)) ** 0.5, worst_key


def run_solve(binary, matches, output, env=None, n_threads=None, timeout=None):
    cmd = [binary, "--matches_file", matches, "--output_file", output]
    if n_threads:
        cmd += ["--n_threads", str(n_threads)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=timeout)
    wall = time.perf_counter() - t0
    info = parse_stdout(r.stdout)
    info.update(rc=r.returncode, wall_s=wall)
    if r.returncode != 0:
        info["stderr_tail"] = r.stderr[-400:]
    return info


def compare(reference, matches_file, workdir=None, components_file=None, n_threads=None, variants=("ceres1", "ceres2"),
            timeout=None):
    """Returns a JSON-able dict (see module docstring)."""
    workdir = workdir or tempfile.mkdtemp(prefix="lfr_cmp_")
    ours = os.path.join(ROOT, "multi-view-refinement", "build", "solve")
    ref_out = os.path.join(workdir, "reference.solution.pb")
    res = {"matches_file": matches_file, "reference_binary": reference, "bar_px": BAR_PX}
    res["reference"] = run_solve(reference, matches_file, ref_out, n_threads=n_threads, timeout=timeout)
    if res["reference"]["rc"] != 0 or not os.path.exists(ref_out):
        res["error"] = "the reference binary failed"
        return res
    ref_pos, ref_fact = load_solution(ref_out)
    res["variants"] = {}
    best = None
    for v in variants:
        env = dict(os.environ, LFR_TUKEY_VARIANT=v)
        if components_file:
            env["LFR_COMPONENTS_FILE"] = components_file
        out = os.path.join(workdir, "lfr.%s.solution.pb" % v)
        info = run_solve(ours, matches_file, out, env=env, n_threads=n_threads, timeout=timeout)
        if info["rc"] == 0:
            pos, _ = load_solution(out)
            worst, rms, key = diff_px(ref_pos, ref_fact, pos)
            info.update(max_abs_diff_px=worst, rms_diff_px=rms, worst_point=list(key) if key else None,
                        within_bar=bool(worst <= BAR_PX))
            if best is None or worst < res["variants"][best]["max_abs_diff_px"]:
                best = v
        res["variants"][v] = info
    res["best_variant"] = best
    if best:
        b = res["variants"][best]
        res["parity"] = "green" if b["within_bar"] else "FAILED"
        for k in ("n_nodes", "n_edges", "n_tracks", "n_components", "max_component_size"):
            if k in res["reference"] and k in b and res["reference"][k] != b[k]:
                res.setdefault("stdout_mismatch", {})[k] = [res["reference"][k], b[k]]
        if res["reference"].get("max_component_size", 0) and "stdout_mismatch" in res and not components_file:
            res["hint"] = ("component counts differ: the reference cut oversized components with Graclus; dump its "
                           "component_idx per node and pass --components_file")
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default=os.environ.get("LFR_REFERENCE_SOLVE"), help="reference-built solve binary")
    ap.add_argument("--matches_file")
    ap.add_argument("--synthetic", choices=["config2", "config4", "config5", "config1_standin", "config3_standin"],
                    help="generate the MatchingFile first (SURVEY 8(d) generators)")
    ap.add_argument("--components_file")
    ap.add_argument("--n_threads", type=int)
    ap.add_argument("--workdir")
    args = ap.parse_args()
    if not args.reference:
        ap.error("--reference (or $LFR_REFERENCE_SOLVE) is required")
    workdir = args.workdir or tempfile.mkdtemp(prefix="lfr_cmp_")
    matches = args.matches_file
    if args.synthetic:
        from lfr_amd import capi, synthetic
        matches = os.path.join(workdir, args.synthetic + ".pb")
        capi.write_matching_file(matches, getattr(synthetic, args.synthetic)())
    if not matches:
        ap.error("--matches_file or --synthetic is required")
    res = compare(args.reference, matches, workdir, args.components_file, args.n_threads)
    print(json.dumps(res, indent=1))
    return 0 if res.get("parity") == "green" else 1


if __name__ == "__main__":
    sys.exit(main())

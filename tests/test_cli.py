"""`solve` launcher: Boost.program_options-compatible flags and exit codes (solve.cc:379-401)."""
import os
import subprocess
import sys

import pytest

from lfr_amd import solve_cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOLVE = os.path.join(ROOT, "multi-view-refinement", "build", "solve")


def test_launcher_is_at_the_reference_path_and_executable():
    assert os.path.isfile(SOLVE) and os.access(SOLVE, os.X_OK)      # benchmark.py:100-104 execs this path


def test_flag_forms():
    a = solve_cli.parse_args(["--matches_file", "m.pb", "--output_file=o.pb"])
    assert a["matches_file"] == "m.pb" and a["output_file"] == "o.pb" and a["n_threads"] == 8 and a["banned_images"] == []
    a = solve_cli.parse_args(["--matches", "m.pb", "--output", "o.pb", "--n_threads", "3",
                              "--banned_images", "x.png", "--banned_images=y.png"])       # unambiguous prefixes
    assert a["n_threads"] == 3 and a["banned_images"] == ["x.png", "y.png"]


@pytest.mark.parametrize("argv,needle", [
    (["--matches_file", "m.pb"], "the option '--output_file' is required but missing"),
    (["--output_file", "o.pb"], "the option '--matches_file' is required but missing"),
    (["--matches_file", "m", "--output_file", "o", "--bogus", "1"], "unrecognised option '--bogus'"),
    (["--matches_file", "m", "--output_file", "o", "--n_threads", "x"], "the argument ('x') for option '--n_threads' is invalid"),
    (["--matches_file", "m", "--matches_file", "n", "--output_file", "o"], "cannot be specified more than once"),
    (["--matches_file", "m", "--output_file"], "the required argument for option '--output_file' is missing"),
    (["m.pb"], "too many positional options"),
])
def test_flag_errors(argv, needle):
    with pytest.raises(solve_cli.FlagError) as e:
        solve_cli.parse_args(argv)
    assert needle in str(e.value)


def test_exit_codes_without_gpu_work():
    r = subprocess.run([sys.executable, SOLVE, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Patch Match graph problem solver\n\nOptions:")
    r = subprocess.run([sys.executable, SOLVE, "--matches_file", "m.pb"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("ERROR: the option '--output_file' is required but missing\n\nOptions:")


def test_parse_failure_exit_code(lfr_lib, tmp_path):
    bad = tmp_path / "bad.pb"
    bad.write_bytes(b"\x0a\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff")
    r = subprocess.run([sys.executable, SOLVE, "--matches_file", str(bad), "--output_file", str(tmp_path / "o.pb")],
                       capture_output=True, text=True)
    assert r.returncode == 255 and "Failed to parse proto object." in r.stderr      # solve.cc:433-436


@pytest.mark.gpu
def test_compare_with_reference_dry_run(lfr_lib, tmp_path):
    """scripts/compare_with_reference.py is the way from "parity unpinned" to "green" for whoever has a reference build of `solve`.
    Nobody here has one, so the script is kept runnable with the product's own launcher standing in for the reference binary: same
    stdout lines parsed, both SolutionFiles decoded, the default Tukey flavour must match itself exactly."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import compare_with_reference as cwr
    pb = os.path.join(ROOT, "tests", "golden", "outliers.pb")
    res = cwr.compare(SOLVE, pb, str(tmp_path), variants=("ceres1", "ceres2"))
    assert res.get("error") is None and res["reference"]["rc"] == 0
    assert res["best_variant"] == "ceres1" and res["parity"] == "green"
    assert res["variants"]["ceres1"]["max_abs_diff_px"] == 0.0
    assert res["variants"]["ceres2"]["max_abs_diff_px"] > 0.0            # inter-track (Tukey) edges exist: the flavours differ
    assert "stdout_mismatch" not in res
    for k in ("n_nodes", "n_edges", "n_tracks", "n_components"):
        assert res["reference"][k] == res["variants"]["ceres1"][k]

"""The C-ABI library loads and exports every symbol include/lfr.h declares (no compute calls)."""
import ctypes
import os
import re

from lfr_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lfr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lfr_lib):
    syms = declared_symbols()
    assert len(syms) >= 24
    raw = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "liblfr_hip.so does not export %s" % s
    assert sorted(capi.EXPORTS) == syms          # the Python binding covers the whole ABI


def test_version_and_error_channel(lfr_lib):
    assert lfr_lib.lfr_version() == 1
    import pytest
    with pytest.raises(capi.LfrError) as e:
        capi.Graph.from_files(["/nonexistent/file.pb"])
    assert e.value.code == -2 and "cannot open" in str(e.value)


def test_struct_layouts_match_header():
    # field counts/sizes of the two stats structs as declared in lfr.h
    assert ctypes.sizeof(capi.ProblemStats) == 9 * 8 + 6 * 8        # (tie_resorts: round 4)
    assert ctypes.sizeof(capi.SolveStats) == 12 * 8 + 5 * 8 + 4 * 8


def test_problem_shard_arguments_are_checked(lfr_lib):
    """Problem(shard=...) used to be ignored silently without a device graph stage, and dropped component_override."""
    import numpy as np
    import pytest
    from lfr_amd import synthetic
    g = capi.Graph.from_arrays(synthetic.generate(seed=5, n_images=8, n_tracks=20))
    with pytest.raises(ValueError):
        capi.Problem(g, shard=(0, 2))
    with pytest.raises(ValueError):
        capi.Problem(g, device_graph_stage=0, shard=(0, 2), component_override=np.zeros(g.n_nodes, np.int64))

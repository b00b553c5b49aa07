"""Build liblfr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
SOURCES = ["lfr_wire.cpp", "lfr_graph.cpp", "lfr_solve.hip", "lfr_assemble.hip", "lfr_graphstage.hip"]
HEADERS = ["lfr_internal.hpp", "lfr_device.hpp", "lfr_assemble.hpp"]
OUT = os.path.join(_HERE, "liblfr_hip.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "lfr.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> lfr_amd/liblfr_hip.so.  Returns the path."""
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-munsafe-fp-atomics",          # hardware fp64 atomic add for the HBM-matrix kernel
           "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += os.environ.get("LFR_HIPCC_FLAGS", "").split()
    cmd += [os.path.join(CSRC, f) for f in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT

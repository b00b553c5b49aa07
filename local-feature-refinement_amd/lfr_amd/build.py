"""Build liblfr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per source under csrc/_obj (rebuilt when the source or any header is newer), compiled in
parallel, then linked: a kernel edit recompiles one file instead of six.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["lfr_wire.cpp", "lfr_graph.cpp", "lfr_treeplan.cpp", "lfr_devctx.cpp", "lfr_solve.hip", "lfr_assemble.hip", "lfr_graphstage.hip"]
HEADERS = ["lfr_internal.hpp", "lfr_device.hpp", "lfr_assemble.hpp", "lfr_devctx.hpp", "lfr_sort.hpp"]
OUT = os.path.join(_HERE, "liblfr_hip.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-munsafe-fp-atomics",          # hardware fp64 atomic add for the HBM-matrix kernel
            "-I", os.path.join(ROOT, "include"), "-I", CSRC] + os.environ.get("LFR_HIPCC_FLAGS", "").split()


def _header_mtime():
    deps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(ROOT, "include", "lfr.h")]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _stale_objects(force):
    flags_tag = " ".join(_flags())
    tag_path = os.path.join(OBJ, "flags.txt")
    old_tag = open(tag_path).read() if os.path.exists(tag_path) else None
    ht = _header_mtime()
    stale = []
    for f in SOURCES:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        if force or old_tag != flags_tag or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), ht):
            stale.append((src, obj))
    return stale, flags_tag, tag_path


def needs_build():
    if not os.path.exists(OUT):
        return True
    stale, _, _ = _stale_objects(False)
    if stale:
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(OBJ, f + ".o")) > t for f in SOURCES)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> lfr_amd/liblfr_hip.so.  Returns the path."""
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    stale, flags_tag, tag_path = _stale_objects(force)

    def compile_one(pair):
        src, obj = pair
        cmd = [_hipcc()] + _flags() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(stale), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, stale))
    open(tag_path, "w").write(flags_tag)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ, f + ".o") for f in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT

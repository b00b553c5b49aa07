"""TEST INFRASTRUCTURE — ctypes loader for the C oracle (oracle/lfr_oracle.c).

*** parity unpinned *** (see the header of lfr_oracle.c / DESIGN.md §3).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "lfr_oracle.c")
_SO = os.path.join(_HERE, "_build", "liblfr_oracle.so")
def _cpu_tag():
    """model + flags of this machine's CPU: a -march=native build is only ever loaded where it was built"""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        keep = [l for l in txt.split("\n") if l.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    return hashlib.sha1("|".join(keep).encode()).hexdigest()[:10]


_SO_NATIVE = os.path.join(_HERE, "_build", "liblfr_oracle_native_%s.so" % _cpu_tag())    # -O3 -march=native: built ON the machine that times it (bench.py)
_lib = None
_libs = {}

INFO_DTYPE = np.dtype([("iterations", "<i4"), ("termination", "<i4"), ("n_successful", "<i4"),
                       ("n_ls_evals", "<i4"), ("n_cost_evals", "<i8"), ("n_jac_evals", "<i8"),
                       ("final_cost", "<f8"), ("initial_cost", "<f8")])

TUKEY = {"ceres1": 1, "ceres2": 2}


def build(force=False):
    """gcc -O2 the C restatement into oracle/_build/ (no -ffast-math: IEEE semantics)."""
    if not force and os.path.exists(_SO) and (not os.path.exists(_SRC)
                                              or os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, _SRC,
                           "-lm", "-lpthread"])
    return _SO


def build_native(force=False):
    """The same source at -O3 -march=native for the timed CPU baseline (bench.py's cpu_baseline leg).  Never shipped between
    machines: the instruction set is the build host's.  IEEE semantics as in build() (no -ffast-math, no contraction)."""
    if not force and os.path.exists(_SO_NATIVE) and os.path.getmtime(_SO_NATIVE) >= os.path.getmtime(_SRC):
        try:
            C.CDLL(_SO_NATIVE)
            return _SO_NATIVE
        except OSError:
            pass
    os.makedirs(os.path.dirname(_SO_NATIVE), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO_NATIVE, _SRC,
                           "-lm", "-lpthread"])
    return _SO_NATIVE


def lib(native=False):
    global _lib
    if native not in _libs:
        path = build_native() if native else build()
        L = C.CDLL(path)
        L.lfro_build.restype = C.c_int
        L.lfro_solve.restype = C.c_int
        for n in ("n_nodes", "n_tracks", "max_track_size", "n_components", "max_component_size", "n_oversized"):
            getattr(L, "lfro_" + n).restype = C.c_int64
            getattr(L, "lfro_" + n).argtypes = [C.c_void_p]
        for n in ("graph_ms", "solver_ms"):
            getattr(L, "lfro_" + n).restype = C.c_double
            getattr(L, "lfro_" + n).argtypes = [C.c_void_p]
        for n in ("node_image", "node_feat", "track", "comp", "is_root", "positions", "comp_nvar",
                  "comp_nedges", "infos"):
            getattr(L, "lfro_" + n).restype = C.c_void_p
            getattr(L, "lfro_" + n).argtypes = [C.c_void_p]
        L.lfro_free.argtypes = [C.c_void_p]
        L.lfro_set_bisect.argtypes = [C.c_void_p]
        L.lfro_set_bisect.restype = None
        L.lfro_eval_edge.restype = C.c_double
        L.lfro_minimize_poly.restype = C.c_double
        assert L.lfro_info_size() == INFO_DTYPE.itemsize
        _libs[native] = L
        if not native:
            _lib = L
    return _libs[native]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(addr, dtype, n):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def flatten(ma, banned=()):
    """MatchArrays -> per-match image indices (dense over the images actually seen, banned
    pairs dropped: solve.cc:444-446) + the list of seen image names in index order."""
    banned = set(banned)
    P = len(ma.pair_img1)
    counts = np.diff(ma.pair_off)
    keep_pair = np.array([not (ma.image_names[ma.pair_img1[p]] in banned or ma.image_names[ma.pair_img2[p]] in banned)
                          for p in range(P)], bool) if banned else np.ones(P, bool)
    keep = np.repeat(keep_pair, counts)
    i1 = np.repeat(ma.pair_img1, counts)[keep].astype(np.int64)
    i2 = np.repeat(ma.pair_img2, counts)[keep].astype(np.int64)
    # images "seen" include pairs without matches (solve.cc:448-451 runs per pair)
    seen = np.zeros(len(ma.image_names), bool)
    seen[ma.pair_img1[keep_pair]] = True
    seen[ma.pair_img2[keep_pair]] = True
    remap = np.cumsum(seen) - 1
    names = [n for n, s in zip(ma.image_names, seen) if s]
    return (remap[i1].astype(np.int32), remap[i2].astype(np.int32), keep, names)


BISECT_FN = C.CFUNCTYPE(C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                        C.POINTER(C.c_int32), C.POINTER(C.c_int32))


def run(ma, banned=(), n_threads=1, tukey_variant="ceres1", comp_override=None, trace_comp=None,
        solve=True, bisect=None, native=False):
    """Run graph stage (+ solve) of the C oracle on a MatchArrays.  Returns a dict.
    bisect: address (int / ctypes function pointer) of a two-way cut with the signature of lfr_bisect_graph, standing
    in for colmap::ComputeNormalizedMinGraphCut (solve.cc:192) when a component exceeds the size cap; without it such
    inputs return rc != 0 (the Graclus cut cannot be restated)."""
    L = lib(native)
    L.lfro_set_bisect(None if bisect is None else C.c_void_p(bisect) if isinstance(bisect, int) else C.cast(bisect, C.c_void_p))
    i1, i2, keep, names = flatten(ma, banned)
    f1 = np.ascontiguousarray(ma.feat1[keep], np.uint32)
    f2 = np.ascontiguousarray(ma.feat2[keep], np.uint32)
    sim = np.ascontiguousarray(ma.sim[keep], np.float32)
    d1 = np.ascontiguousarray(ma.disp1[keep].reshape(-1, 18), np.float32)
    d2 = np.ascontiguousarray(ma.disp2[keep].reshape(-1, 18), np.float32)
    M = int(f1.shape[0])
    h = C.c_void_p()
    co = None if comp_override is None else np.ascontiguousarray(comp_override, np.int64)
    rc = L.lfro_build(C.c_int(len(names)), C.c_int64(M), _ptr(i1), _ptr(i2), _ptr(f1), _ptr(f2), _ptr(sim),
                      _ptr(d1), _ptr(d2), None if co is None else _ptr(co), C.byref(h))
    try:
        n = L.lfro_n_nodes(h)
        out = {"rc": rc, "n_nodes": n, "n_edges": 2 * M, "image_names": names,
               "n_tracks": L.lfro_n_tracks(h), "max_track_size": L.lfro_max_track_size(h),
               "n_components": L.lfro_n_components(h), "max_component_size": L.lfro_max_component_size(h),
               "n_oversized": L.lfro_n_oversized(h), "graph_ms": L.lfro_graph_ms(h)}
        out["node_image"] = _view(L.lfro_node_image(h), np.int32, n)
        out["node_feat"] = _view(L.lfro_node_feat(h), np.uint32, n)
        if n:
            out["track"] = _view(L.lfro_track(h), np.int64, n)
            out["comp"] = _view(L.lfro_comp(h), np.int64, n)
            out["is_root"] = _view(L.lfro_is_root(h), np.uint8, n).astype(bool)
        if rc == 0 and solve and n:
            cap = 256
            rows = np.zeros((cap, 8))
            tn = C.c_int(0)
            L.lfro_solve(h, C.c_int(n_threads), C.c_int(TUKEY[tukey_variant]),
                         C.c_int64(-1 if trace_comp is None else trace_comp),
                         _ptr(rows) if trace_comp is not None else None, C.c_int(cap), C.byref(tn))
            nc = out["n_components"]
            out["positions"] = _view(L.lfro_positions(h), np.float64, 2 * n).reshape(n, 2)
            out["infos"] = _view(L.lfro_infos(h), INFO_DTYPE, nc)
            out["comp_nvar"] = _view(L.lfro_comp_nvar(h), np.int32, nc)
            out["comp_nedges"] = _view(L.lfro_comp_nedges(h), np.int32, nc)
            out["solver_ms"] = L.lfro_solver_ms(h)
            if trace_comp is not None:
                out["trace"] = rows[:tn.value].copy()
        # keep flow arrays alive until the handle is gone
        out["_keep"] = (d1, d2)
    finally:
        L.lfro_free(h)
    out.pop("_keep", None)
    return out


# --- unit-level entry points -------------------------------------------------
def interpolate(flow18, row, col):
    out = np.zeros(6)
    f = np.ascontiguousarray(flow18, np.float32)
    lib().lfro_interpolate(_ptr(f), C.c_double(row), C.c_double(col), _ptr(out))
    return out[0:2], out[2:4], out[4:6]


def loss(kind, s, w, tukey_variant="ceres1"):
    out = np.zeros(3)
    lib().lfro_loss(C.c_int(kind), C.c_double(s), C.c_double(w), C.c_int(TUKEY[tukey_variant]), _ptr(out))
    return out


def eval_edge(flow18, sim, kind, x1, x2, tukey_variant="ceres1"):
    out = np.zeros(7)
    f = np.ascontiguousarray(flow18, np.float32)
    a = np.ascontiguousarray(x1, np.float64)
    b = np.ascontiguousarray(x2, np.float64)
    c = lib().lfro_eval_edge(_ptr(f), C.c_float(sim), C.c_int(kind), _ptr(a), _ptr(b),
                             C.c_int(TUKEY[tukey_variant]), _ptr(out))
    return c, out[0:2], out[2:6].reshape(2, 2), out[6]


def minimize_poly(samples, x_min, x_max):
    s = np.ascontiguousarray(samples, np.float64)
    return lib().lfro_minimize_poly(_ptr(s), C.c_int(s.shape[0]), C.c_double(x_min), C.c_double(x_max))

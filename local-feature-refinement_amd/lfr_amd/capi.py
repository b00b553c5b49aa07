"""ctypes binding of liblfr_hip.so (C ABI: include/lfr.h).

This is the host-side mirror of the reference's solver stages (``solve.cc`` main():
ingest -> tracks/roots/components -> batched LM -> SolutionFile).  There is no CPU
fallback: if the HIP library is missing, importing the binding raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFR_LIB_OVERRIDE") or os.path.join(_HERE, "liblfr_hip.so")   # override: A/B of kernel variants

TUKEY = {"ceres1": 1, "ceres2": 2}
TERM_CONVERGENCE, TERM_NO_CONVERGENCE, TERM_FAILURE = 0, 1, 2


class LfrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lfr error %d: %s" % (code, msg))
        self.code = code


NUM_KERNEL_CLASSES = 9          # LFR_NUM_KERNEL_CLASSES


class ProblemStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_tracks", "max_track_size", "n_components", "max_component_size", "n_cut_components",
        "n_solved_components", "n_solved_tracks", "n_solved_edges", "n_solved_nodes")] + \
        [(n, C.c_double) for n in ("tracks_ms", "roots_ms", "graph_cut_ms", "assemble_ms", "kruskal_rounds", "tie_resorts")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SolveStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_components", "n_edges", "n_nodes", "n_tracks", "n_converged", "n_no_convergence", "n_failed",
        "sum_iterations", "ref_jacobian_passes_edges", "ref_cost_passes_edges", "exec_passes_edges",
        "ref_passes_nodes")] + \
        [("sum_final_cost", C.c_double), ("kernel_ms", C.c_double), ("h2d_ms", C.c_double),
         ("d2h_ms", C.c_double), ("dominant_kernel_ms", C.c_double),
         ("dominant_kernel_edges", C.c_int64), ("dominant_kernel_nodes", C.c_int64),
         ("dominant_ref_passes_edges", C.c_int64), ("dominant_ref_passes_nodes", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    """Load liblfr_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python __graft_entry__.py` "
                          "(there is no CPU fallback for the solver path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    pp = C.POINTER(C.c_void_p)
    cpp = C.POINTER(C.c_char_p)
    sig = {
        "lfr_version": (C.c_int, []),
        "lfr_last_error": (C.c_char_p, []),
        "lfr_graph_from_files": (C.c_int, [cpp, C.c_int, cpp, C.c_int, pp]),
        "lfr_graph_from_matches_file": (C.c_int, [C.c_char_p, cpp, C.c_int, pp]),
        "lfr_graph_from_matches_file_device": (C.c_int, [C.c_char_p, cpp, C.c_int, C.c_int, pp]),
        "lfr_graph_from_arrays": (C.c_int, [i32, cpp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, cpp, C.c_int, pp]),
        "lfr_graph_from_arrays_device_flows": (C.c_int, [i32, cpp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, cpp, C.c_int, pp]),
        "lfr_graph_to_device": (C.c_int, [vp, C.c_int]),
        "lfr_graph_evict_device": (C.c_int, [vp]),
        "lfr_graph_free": (None, [vp]),
        "lfr_graph_num_nodes": (i64, [vp]),
        "lfr_graph_num_edges": (i64, [vp]),
        "lfr_graph_num_images": (i32, [vp]),
        "lfr_graph_get_nodes": (C.c_int, [vp, vp, vp]),
        "lfr_graph_image_name": (C.c_char_p, [vp, i32]),
        "lfr_graph_image_fact": (C.c_float, [vp, i32]),
        "lfr_write_matching_file": (C.c_int, [C.c_char_p, i32, cpp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
        "lfr_problem_build": (C.c_int, [vp, i64, vp, pp]),
        "lfr_problem_build_labels": (C.c_int, [vp, i64, vp, pp]),
        "lfr_problem_build_hip": (C.c_int, [vp, C.c_int, i64, vp, pp]),
        "lfr_problem_build_hip_ex": (C.c_int, [vp, C.c_int, i64, vp, C.c_int, pp]),
        "lfr_problem_build_hip_shard": (C.c_int, [vp, C.c_int, i64, C.c_int, C.c_int, C.c_int, pp]),
        "lfr_problem_cc_sharded": (C.c_int, [vp]),
        "lfr_problem_free": (None, [vp]),
        "lfr_bisect_graph": (i64, [i64, vp, vp, vp, vp, vp]),
        "lfr_debug_recursive_cut": (i64, [i64, vp, vp, vp, i64, vp, i64, vp, vp]),
        "lfr_problem_get_stats": (C.c_int, [vp, C.POINTER(ProblemStats)]),
        "lfr_problem_get_labels": (C.c_int, [vp, vp, vp, vp]),
        "lfr_problem_shard_components": (i64, [vp, C.c_int, C.c_int, vp, vp]),
        "lfr_hip_warmup": (C.c_int, [C.c_int]),
        "lfr_hip_reserve": (C.c_int, [C.c_int, i64, i64]),
        "lfr_hip_trim": (C.c_int, [C.c_int]),
        "lfr_hip_synchronize": (C.c_int, [C.c_int]),
        "lfr_batch_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, pp]),
        "lfr_batch_free": (None, [vp]),
        "lfr_batch_solve": (C.c_int, [vp, vp, C.POINTER(SolveStats)]),
        "lfr_batch_download": (C.c_int, [vp, vp]),
        "lfr_batch_positions_view": (C.c_int, [vp, pp]),
        "lfr_batch_positions_view_f32": (C.c_int, [vp, pp]),
        "lfr_batch_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), vp, vp]),
        "lfr_batch_component_info": (i64, [vp, vp, vp, vp, vp, vp, vp]),
        "lfr_batch_spin_timeouts": (i64, [vp]),
        "lfr_batch_team_runs": (i64, [vp]),
        "lfr_batch_team_fallbacks": (i64, [vp]),
        "lfr_debug_occupy": (C.c_int, [C.c_int, C.c_int, C.c_double]),
        "lfr_batch_tree_stats": (i64, [vp, vp, vp, vp, vp, vp]),
        "lfr_debug_eval_edges": (C.c_int, [C.c_int, i64, vp, vp, vp, vp, vp, C.c_int, vp, vp]),
        "lfr_debug_ls_next_step": (C.c_int, [C.c_int, i64, vp, vp, C.c_int, vp]),
        "lfr_debug_tree_plan": (i64, [i32, i64, vp, vp, i64, vp]),
        "lfr_debug_pool_selftest": (i64, [C.c_int, i64, C.c_int]),
        "lfr_debug_sort_pairs": (C.c_int, [C.c_int, i64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
        "lfr_debug_exclusive_sum": (C.c_int, [C.c_int, i64, C.c_int, vp, vp]),
        "lfr_solve_hip": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(SolveStats)]),
        "lfr_solve_hip_multi": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.POINTER(SolveStats)]),
        "lfr_solve_graph_hip_multi": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, vp, C.POINTER(ProblemStats), C.POINTER(SolveStats)]),
        "lfr_write_solution": (C.c_int, [vp, vp, C.c_char_p, C.POINTER(i64)]),
        "lfr_apply_displacements": (C.c_int, [vp, vp, C.c_char_p, vp, i64, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)       # AttributeError here = the .so does not export the ABI
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTS = ["lfr_version", "lfr_last_error", "lfr_graph_from_files", "lfr_graph_from_matches_file", "lfr_graph_from_matches_file_device",
           "lfr_graph_from_arrays", "lfr_graph_from_arrays_device_flows", "lfr_graph_to_device", "lfr_graph_evict_device",
           "lfr_problem_build_hip_ex", "lfr_problem_build_hip_shard", "lfr_problem_cc_sharded", "lfr_hip_reserve", "lfr_hip_trim", "lfr_batch_positions_view", "lfr_batch_positions_view_f32", "lfr_bisect_graph", "lfr_debug_eval_edges", "lfr_debug_ls_next_step", "lfr_debug_tree_plan", "lfr_debug_pool_selftest", "lfr_debug_sort_pairs", "lfr_debug_exclusive_sum", "lfr_debug_recursive_cut", "lfr_hip_synchronize", "lfr_graph_free", "lfr_graph_num_nodes", "lfr_graph_num_edges",
           "lfr_graph_num_images", "lfr_graph_get_nodes", "lfr_graph_image_name", "lfr_graph_image_fact",
           "lfr_write_matching_file", "lfr_problem_build", "lfr_problem_build_labels", "lfr_problem_build_hip", "lfr_problem_free", "lfr_problem_get_stats",
           "lfr_problem_get_labels", "lfr_problem_shard_components", "lfr_hip_warmup", "lfr_batch_create", "lfr_batch_free", "lfr_batch_solve",
           "lfr_batch_download", "lfr_batch_timing", "lfr_batch_spin_timeouts", "lfr_batch_team_runs", "lfr_batch_team_fallbacks", "lfr_debug_occupy", "lfr_batch_tree_stats", "lfr_batch_component_info", "lfr_solve_hip", "lfr_solve_hip_multi", "lfr_solve_graph_hip_multi", "lfr_write_solution", "lfr_apply_displacements"]


def _check(rc):
    if rc != 0:
        raise LfrError(rc, lib().lfr_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _cstrs(strings):
    arr = (C.c_char_p * max(len(strings), 1))()
    for i, s in enumerate(strings):
        arr[i] = s.encode("utf-8") if isinstance(s, str) else s
    return arr


def hip_warmup_async(device=0, n_nodes=0, n_matches=0):
    """Start creating the HIP context on a side thread (ctypes releases the GIL); returns the thread.
    n_nodes / n_matches > 0 also pre-populates the slab caches for a graph of (about) that size."""
    import threading
    L = lib()

    def work():
        L.lfr_hip_warmup(device)
        if n_matches > 0:
            L.lfr_hip_reserve(device, n_nodes, n_matches)
    t = threading.Thread(target=work, daemon=True)
    t.start()
    return t


FLOWS_STAY_ON_HOST = 1      # LFR_BUILD_FLOWS_STAY_ON_HOST


class Graph:
    """Parsed match graph (solve.cc:405-481)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_matches_file(cls, path, banned=(), device=None):
        """device: ingest straight to that GPU (lfr_graph_from_matches_file_device: the flows' upload overlaps the node numbering)."""
        h = C.c_void_p()
        if device is None:
            _check(lib().lfr_graph_from_matches_file(os.fsencode(path), _cstrs(list(banned)), len(banned), C.byref(h)))
        else:
            _check(lib().lfr_graph_from_matches_file_device(os.fsencode(path), _cstrs(list(banned)), len(banned), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_files(cls, paths, banned=()):
        h = C.c_void_p()
        _check(lib().lfr_graph_from_files(_cstrs([os.fsencode(p) for p in paths]), len(paths),
                                          _cstrs(list(banned)), len(banned), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, ma, banned=()):
        """ma: :class:`lfr_amd.synthetic.MatchArrays`."""
        h = C.c_void_p()
        a = _contig(ma)
        _check(lib().lfr_graph_from_arrays(len(ma.image_names), _cstrs(ma.image_names), _ptr(a["facts"]),
                                           len(a["p1"]), _ptr(a["p1"]), _ptr(a["p2"]), _ptr(a["off"]),
                                           _ptr(a["f1"]), _ptr(a["f2"]), _ptr(a["sim"]), _ptr(a["d1"]),
                                           _ptr(a["d2"]), _cstrs(list(banned)), len(banned), C.byref(h)))
        return cls(h)

    @classmethod
    def from_device_flows(cls, ma, disp1_ptr, disp2_ptr, device=0, banned=()):
        """ma: MatchArrays (its disp1/disp2 are ignored); disp*_ptr: device addresses (int) of float32
        [n_matches, 18] arrays on HIP device `device`, e.g. tensor.data_ptr().  The caller keeps them alive
        until the Batch exists."""
        h = C.c_void_p()
        a = _contig(ma, flows=False)
        _check(lib().lfr_graph_from_arrays_device_flows(
            len(ma.image_names), _cstrs(ma.image_names), _ptr(a["facts"]), len(a["p1"]), _ptr(a["p1"]), _ptr(a["p2"]),
            _ptr(a["off"]), _ptr(a["f1"]), _ptr(a["f2"]), _ptr(a["sim"]), C.c_void_p(disp1_ptr), C.c_void_p(disp2_ptr),
            device, _cstrs(list(banned)), len(banned), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            lib().lfr_graph_free(self._h)
            self._h = None

    __del__ = close

    def to_device(self, device=0):
        """Start the asynchronous upload of the graph (endpoints, similarities, flows) to HBM."""
        _check(lib().lfr_graph_to_device(self._h, device))

    def evict_device(self):
        """Drop the graph's cached copy in HBM (the next device pipeline run uploads it again)."""
        _check(lib().lfr_graph_evict_device(self._h))

    @property
    def n_nodes(self):
        return lib().lfr_graph_num_nodes(self._h)

    @property
    def n_edges(self):
        return lib().lfr_graph_num_edges(self._h)

    @property
    def n_images(self):
        return lib().lfr_graph_num_images(self._h)

    def nodes(self):
        n = self.n_nodes
        img = np.zeros(n, np.int32)
        feat = np.zeros(n, np.uint32)
        _check(lib().lfr_graph_get_nodes(self._h, _ptr(img), _ptr(feat)))
        return img, feat

    def image_names(self):
        return [lib().lfr_graph_image_name(self._h, i).decode("utf-8") for i in range(self.n_images)]

    def image_facts(self):
        return [lib().lfr_graph_image_fact(self._h, i) for i in range(self.n_images)]

    def apply_displacements(self, positions, image_name, keypoints):
        """In-place consumer arithmetic of colmap_utils.py:126-137 on a float32 [num_features, >=2] array."""
        assert keypoints.dtype == np.float32 and keypoints.ndim == 2 and keypoints.flags.c_contiguous
        pos = np.ascontiguousarray(positions, np.float64)
        _check(lib().lfr_apply_displacements(self._h, _ptr(pos), image_name.encode("utf-8"), _ptr(keypoints),
                                             keypoints.shape[0], keypoints.shape[1]))
        return keypoints

    def write_solution(self, positions, path):
        """SolutionFile emit (solve.cc:644-679); returns the '> 0.5' count of solve.cc:666-670."""
        pos = np.ascontiguousarray(positions, np.float64)
        n_out = C.c_int64(0)
        _check(lib().lfr_write_solution(self._h, _ptr(pos), os.fsencode(path), C.byref(n_out)))
        return n_out.value


def _contig(ma, flows=True):
    M = ma.n_matches
    if not flows:
        return {"facts": np.ascontiguousarray(ma.facts, np.float32),
                "p1": np.ascontiguousarray(ma.pair_img1, np.int32), "p2": np.ascontiguousarray(ma.pair_img2, np.int32),
                "off": np.ascontiguousarray(ma.pair_off, np.int64),
                "f1": np.ascontiguousarray(ma.feat1, np.uint32), "f2": np.ascontiguousarray(ma.feat2, np.uint32),
                "sim": np.ascontiguousarray(ma.sim, np.float32)}
    return {"facts": np.ascontiguousarray(ma.facts, np.float32),
            "p1": np.ascontiguousarray(ma.pair_img1, np.int32), "p2": np.ascontiguousarray(ma.pair_img2, np.int32),
            "off": np.ascontiguousarray(ma.pair_off, np.int64),
            "f1": np.ascontiguousarray(ma.feat1, np.uint32), "f2": np.ascontiguousarray(ma.feat2, np.uint32),
            "sim": np.ascontiguousarray(ma.sim, np.float32),
            "d1": np.ascontiguousarray(ma.disp1, np.float32).reshape(M, 18),
            "d2": np.ascontiguousarray(ma.disp2, np.float32).reshape(M, 18)}


def eval_edges_hip(flows, sim, kind, x1, x2, tukey_variant="ceres1", device=0):
    """The kernels' per-edge arithmetic on the GPU (lfr_debug_eval_edges): returns (out[n, 8], cost_only[n])."""
    flows = np.ascontiguousarray(flows, np.float32).reshape(-1, 18)
    n = flows.shape[0]
    sim = np.ascontiguousarray(sim, np.float32)
    kind = np.ascontiguousarray(kind, np.int32)
    x1 = np.ascontiguousarray(x1, np.float64).reshape(n, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(n, 2)
    out = np.zeros((n, 8), np.float64)
    cost = np.zeros(n, np.float64)
    _check(lib().lfr_debug_eval_edges(device, n, _ptr(flows), _ptr(sim), _ptr(kind), _ptr(x1), _ptr(x2), TUKEY[tukey_variant],
                                      _ptr(out), _ptr(cost)))
    return out, cost


def ls_next_step_hip(samples, dir_max, register_version=False, device=0):
    """The kernels' line-search contraction on the GPU (lfr_debug_ls_next_step): samples[n, 3, 5] = (x, value, gradient,
    value_valid, gradient_valid) of the initial / previous / current sample.  Returns the next step sizes (negative: give up)."""
    samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 15)
    n = samples.shape[0]
    dir_max = np.ascontiguousarray(dir_max, np.float64)
    a = np.zeros(n, np.float64)
    _check(lib().lfr_debug_ls_next_step(device, n, _ptr(samples), _ptr(dir_max), int(register_version), _ptr(a)))
    return a


def sort_pairs_hip(keys, vals, begin_bit, end_bit, use_library=False, device=0):
    """The pipeline's stable device sort (lfr_debug_sort_pairs): uint32 / uint64 keys, uint32 values -> (sorted keys, values)."""
    keys = np.ascontiguousarray(keys)
    assert keys.dtype in (np.uint32, np.uint64)
    vals = np.ascontiguousarray(vals, np.uint32)
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    _check(lib().lfr_debug_sort_pairs(device, keys.size, keys.dtype.itemsize, _ptr(keys), _ptr(vals), int(begin_bit), int(end_bit),
                                      int(bool(use_library)), _ptr(ko), _ptr(vo)))
    return ko, vo


def exclusive_sum_hip(values, device=0):
    """The pipeline's one-launch exclusive prefix sum (lfr_debug_exclusive_sum) of a uint32 / uint64 array."""
    values = np.ascontiguousarray(values)
    assert values.dtype in (np.uint32, np.uint64)
    out = np.empty_like(values)
    _check(lib().lfr_debug_exclusive_sum(device, values.size, values.dtype.itemsize, _ptr(values), _ptr(out)))
    return out


def occupy_hip(workgroups, milliseconds, device=0):
    """Keeps `workgroups` CUs busy for `milliseconds` on a stream of their own (lfr_debug_occupy); returns when they have started."""
    _check(lib().lfr_debug_occupy(device, int(workgroups), float(milliseconds)))


def bisect_graph(edges, weights):
    """The library's substitute for colmap::ComputeNormalizedMinGraphCut(edges, weights, 2) (solve.cc:192):
    {node id: side}."""
    e = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    a, b = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
    w = np.ascontiguousarray(weights, np.int32)
    nodes = np.zeros(2 * len(w) + 1, np.int32)
    part = np.zeros(2 * len(w) + 1, np.int32)
    n = lib().lfr_bisect_graph(len(w), _ptr(a), _ptr(b), _ptr(w), _ptr(nodes), _ptr(part))
    if n < 0:
        _check(int(n))
    return {int(nodes[i]): int(part[i]) for i in range(n)}


def recursive_cut(edges, weights, node_weights, max_weight):
    """The product's size-cap recursion (lfr_debug_recursive_cut): (nodes ascending, subset of each)."""
    e = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    a, b = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
    w = np.ascontiguousarray(weights, np.int32)
    nw = np.ascontiguousarray(node_weights, np.int64)
    nodes = np.zeros(2 * len(w) + 1, np.int32)
    sub = np.zeros(2 * len(w) + 1, np.int32)
    n = lib().lfr_debug_recursive_cut(len(w), _ptr(a), _ptr(b), _ptr(w), len(nw), _ptr(nw), int(max_weight), _ptr(nodes), _ptr(sub))
    if n < 0:
        _check(int(n))
    return nodes[:n].copy(), sub[:n].copy()


def write_matching_file(path, ma):
    """Native MatchingFile writer (compute_match_graph.py:163-205 equivalent)."""
    a = _contig(ma)
    _check(lib().lfr_write_matching_file(os.fsencode(path), len(ma.image_names), _cstrs(ma.image_names),
                                         _ptr(a["facts"]), len(a["p1"]), _ptr(a["p1"]), _ptr(a["p2"]),
                                         _ptr(a["off"]), _ptr(a["f1"]), _ptr(a["f2"]), _ptr(a["sim"]),
                                         _ptr(a["d1"]), _ptr(a["d2"])))


class Problem:
    """Tracks, roots, components and the device batch layout (solve.cc:487-606, 79-143)."""

    def __init__(self, graph, max_nodes_in_component=0, component_override=None, device_assembly=False,
                 device_graph_stage=None, flags=0, shard=None):
        """device_assembly=True: graph stage only; the batch is assembled on the GPU by Batch / solve_hip.
        device_graph_stage=<device ordinal>: tracks/roots/components on that GPU too (implies device_assembly);
        flags=FLOWS_STAY_ON_HOST: do not stage the flows in HBM (sharded batches gather their rows zero-copy);
        shard=(rank, world): the graph stage over the connected components of the match graph dealt to `rank` only."""
        self.graph = graph
        h = C.c_void_p()
        co = None if component_override is None else np.ascontiguousarray(component_override, np.int64)
        self.cc_sharded = False
        if shard is not None and device_graph_stage is None:
            raise ValueError("Problem(shard=...) needs device_graph_stage: the host graph stage is not sharded (deal the components with Batch(p, dev, rank, world))")
        if shard is not None and component_override is not None:
            raise ValueError("Problem(shard=...) cannot be combined with component_override (lfr_problem_build_hip_shard has no override)")
        if device_graph_stage is not None and shard is not None and shard[1] > 1:
            # multi-GPU: this rank's connected components only (lfr_problem_build_hip_shard); cc_sharded False = the whole graph after all
            _check(lib().lfr_problem_build_hip_shard(graph._h, int(device_graph_stage), int(max_nodes_in_component), int(flags),
                                                     int(shard[0]), int(shard[1]), C.byref(h)))
            self.cc_sharded = bool(lib().lfr_problem_cc_sharded(h))
        elif device_graph_stage is not None:
            _check(lib().lfr_problem_build_hip_ex(graph._h, int(device_graph_stage), int(max_nodes_in_component), _ptr(co),
                                                  int(flags), C.byref(h)))
        else:
            fn = lib().lfr_problem_build_labels if device_assembly else lib().lfr_problem_build
            _check(fn(graph._h, int(max_nodes_in_component), _ptr(co), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().lfr_problem_free(self._h)
            self._h = None

    __del__ = close

    def stats(self):
        s = ProblemStats()
        _check(lib().lfr_problem_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def labels(self):
        n = self.graph.n_nodes
        track = np.zeros(n, np.int64)
        root = np.zeros(n, np.uint8)
        comp = np.zeros(n, np.int64)
        _check(lib().lfr_problem_get_labels(self._h, _ptr(track), _ptr(root), _ptr(comp)))
        return track, root.astype(bool), comp

    def shard_components(self, rank, world):
        """(component ids, edge counts) of the solvable components dealt to shard rank/world."""
        n = lib().lfr_problem_shard_components(self._h, rank, world, None, None)
        if n < 0:
            _check(int(n))
        comps = np.zeros(n, np.int64)
        edges = np.zeros(n, np.int64)
        lib().lfr_problem_shard_components(self._h, rank, world, _ptr(comps), _ptr(edges))
        return comps, edges

    def solve_hip(self, device=0, tukey_variant="ceres1", want_stats=True):
        """Upload + solve + download on one GPU.  Returns (positions[n,2], stats dict or None).  want_stats=False skips the statistics
        (they fetch every descriptor and per-component record to the host: 15-20 ms for 147 k components, several times the solve)."""
        n = self.graph.n_nodes
        pos = np.empty((n, 2), np.float64)                  # (fully written by the download: every node, zeros where nothing was solved)
        st = SolveStats()
        _check(lib().lfr_solve_hip(self._h, device, TUKEY[tukey_variant], _ptr(pos), C.byref(st) if want_stats else None))
        return pos, (st.as_dict() if want_stats else None)


def solve_graph_hip_multi(graph, devices, max_nodes_in_component=0, tukey_variant="ceres1"):
    """Graph stage, assembly and solve sharded over several GPUs from this process (lfr_solve_graph_hip_multi): device k takes the
    connected components of the match graph dealt to shard k.  Returns (positions[n, 2], problem stats, solve stats)."""
    n = graph.n_nodes
    pos = np.zeros((n, 2), np.float64)
    pst, st = ProblemStats(), SolveStats()
    dev = np.ascontiguousarray(devices, np.int32)
    _check(lib().lfr_solve_graph_hip_multi(graph._h, _ptr(dev), len(dev), int(max_nodes_in_component), TUKEY[tukey_variant], _ptr(pos),
                                          C.byref(pst), C.byref(st)))
    return pos, pst.as_dict(), st.as_dict()


def solve_hip_multi(problem, devices, tukey_variant="ceres1"):
    """Shard a host-assembled problem over several GPUs from this process (one host thread per device)."""
    n = problem.graph.n_nodes
    pos = np.zeros((n, 2), np.float64)
    st = SolveStats()
    dev = np.ascontiguousarray(devices, np.int32)
    _check(lib().lfr_solve_hip_multi(problem._h, _ptr(dev), len(dev), TUKEY[tukey_variant], _ptr(pos), C.byref(st)))
    return pos, st.as_dict()


class Batch:
    """A problem (or one LPT shard of it) resident in HBM."""

    def __init__(self, problem, device=0, shard_rank=0, shard_world=1, tukey_variant="ceres1"):
        self.problem = problem
        h = C.c_void_p()
        _check(lib().lfr_batch_create(problem._h, device, shard_rank, shard_world, TUKEY[tukey_variant], C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().lfr_batch_free(self._h)
            self._h = None

    __del__ = close

    def solve(self, stream=None, want_stats=True):
        """stream: a hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        st = SolveStats()
        _check(lib().lfr_batch_solve(self._h, C.c_void_p(stream) if stream else None,
                                     C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def timing(self, solves_back=0):
        """HIP-event times of one of the last 64 solves: (total_ms, per-kernel-class ms, per-class edges)."""
        tot = C.c_double(0.0)
        cls = np.zeros(NUM_KERNEL_CLASSES, np.float64)
        edges = np.zeros(NUM_KERNEL_CLASSES, np.int64)
        _check(lib().lfr_batch_timing(self._h, solves_back, C.byref(tot), _ptr(cls), _ptr(edges)))
        return tot.value, cls, edges

    def download(self, positions=None):
        n = self.problem.graph.n_nodes
        if positions is None:
            positions = np.zeros((n, 2), np.float64)
        _check(lib().lfr_batch_download(self._h, _ptr(positions)))
        return positions

    def positions_view(self):
        """Zero-copy [n, 2] float64 view of the batch's pinned staging buffer (valid until the next solve /
        download / close of this batch; nodes outside the shard read 0)."""
        n = self.problem.graph.n_nodes
        p = C.c_void_p()
        _check(lib().lfr_batch_positions_view(self._h, C.byref(p)))
        if n == 0:
            return np.zeros((0, 2), np.float64)
        buf = (C.c_double * (2 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.float64).reshape(n, 2)

    def positions_view_f32(self):
        """The same as [n, 2] float32 - the precision of the reference's SolutionFile (solve.cc:661-664) - converted on the device:
        half the bytes over PCIe."""
        n = self.problem.graph.n_nodes
        p = C.c_void_p()
        _check(lib().lfr_batch_positions_view_f32(self._h, C.byref(p)))
        if n == 0:
            return np.zeros((0, 2), np.float32)
        buf = (C.c_float * (2 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 2)

    def tree_stats(self):
        """Per component (order of component_info): columns / tiles / 16x16x16 updates per factorization / levels / sweep items of the
        elimination-tree plan (zeros for components of the other kernel classes)."""
        n = lib().lfr_batch_tree_stats(self._h, None, None, None, None, None)
        if n < 0:
            _check(int(n))
        out = {k: np.zeros(n, np.int64) for k in ("columns", "tiles", "updates", "levels", "items")}
        lib().lfr_batch_tree_stats(self._h, _ptr(out["columns"]), _ptr(out["tiles"]), _ptr(out["updates"]), _ptr(out["levels"]), _ptr(out["items"]))
        return out

    def spin_timeouts(self):
        """Bounded spin-waits of the workgroup kernels that ran out during the latest solve (0 on a healthy run)."""
        n = lib().lfr_batch_spin_timeouts(self._h)
        if n < 0:
            _check(int(n))
        return int(n)

    def team_runs(self):
        """Components the latest solve handed to a team of two or more workgroups (elimination-tree class)."""
        n = lib().lfr_batch_team_runs(self._h)
        if n < 0:
            _check(int(n))
        return int(n)

    def team_fallbacks(self):
        """Components the latest solve's teams could not serve at their size (CUs not resident together) and one workgroup solved."""
        n = lib().lfr_batch_team_fallbacks(self._h)
        if n < 0:
            _check(int(n))
        return int(n)

    def component_info(self):
        n = lib().lfr_batch_component_info(self._h, None, None, None, None, None, None)
        if n < 0:
            _check(int(n))
        comp = np.zeros(n, np.int64)
        it = np.zeros(n, np.int32)
        term = np.zeros(n, np.int32)
        cost = np.zeros(n, np.float64)
        nvar = np.zeros(n, np.int32)
        ne = np.zeros(n, np.int32)
        lib().lfr_batch_component_info(self._h, _ptr(comp), _ptr(it), _ptr(term), _ptr(cost), _ptr(nvar), _ptr(ne))
        return {"component": comp, "iterations": it, "termination": term, "final_cost": cost,
                "n_var_nodes": nvar, "n_edges": ne}


def tree_plan(n_var, words):
    """Elimination-tree plan of one large component (lfr_debug_tree_plan): (blob of uint32 words as the kernel reads it, info dict)."""
    w = np.ascontiguousarray(words, np.uint32)
    info = np.zeros(8, np.int64)
    n = lib().lfr_debug_tree_plan(int(n_var), int(w.shape[0]), _ptr(w), None, 0, _ptr(info))
    if n < 0:
        _check(int(n))
    blob = np.zeros(int(n), np.uint32)
    lib().lfr_debug_tree_plan(int(n_var), int(w.shape[0]), _ptr(w), _ptr(blob), int(n), _ptr(info))
    keys = ("blocks", "tiles", "levels", "items", "updates", "tracks", "segments", "column_rounds")
    return blob, {k: int(v) for k, v in zip(keys, info)}

"""Drop-in for the reference's `solve` executable (multi-view-refinement/solve.cc:375-682).

Same flags (Boost.program_options semantics: ``--flag value``, ``--flag=value``, unambiguous
prefixes; solve.cc:379-385), same stdout lines in the same order (solve.cc:484-485,534,549,589,
591,606,638,641,670), same exit codes (0; 1 for flag errors, solve.cc:397-401; 255 for
parse/write failures, solve.cc:433-436,674-677).  The work is done by liblfr_hip.so on a
MI355X; there is no CPU fallback — a missing library or GPU is a loud error (exit 2).

`--n_threads` (solve.cc:384,617: workers of the reference's CPU thread pool) has no GPU meaning; when it is given
explicitly it sets the worker count of the host-side stages of this drop-in (MatchingFile scanner, host graph
stage when it is used) exactly like LFR_HOST_THREADS; when it is left at its default the scanner uses
min(cores, 32) threads.

Environment (additions that default so the reference's scripts run unchanged):
  LFR_DEVICE            HIP device ordinal (default 0; a single entry in LFR_GPUS means the same)
  LFR_GPUS              comma-separated device ordinals: shard the components over several GPUs of the
                        node from this one process (e.g. 0,1,2,3,4,5,6,7)
  LFR_TUKEY_VARIANT     ceres1 (default; Ceres <= 1.14) | ceres2 (Ceres >= 2.0); the active flavour is printed
                        once on stderr when the graph has inter-track edges to apply it to
  LFR_UPLOAD_IN_TOTAL   1: keep the graph on the host until "Total time" starts (its PCIe upload is then inside that
                        span); default: the graph is copied to HBM as the last step of ingest, as the reference builds
                        its Graph object before its timer starts (solve.cc:405-487)
  LFR_COMPONENTS_FILE   raw little-endian int64[n_nodes] component ids replacing the size-cap
                        graph cut (side-car for exact parity with a reference run)
  LFR_HOST_GRAPH_STAGE  1: tracks/roots/components on the host (batch assembly stays on the GPU)
  LFR_HOST_ASSEMBLY     1: graph stage and batch layout on the host
  LFR_HOST_THREADS      worker threads of the host graph stage / scanner (default min(cores, 32))
  LFR_WAIT_WARMUP       0: start the pipeline as soon as the HIP context exists instead of when every kernel's code is loaded
  LFR_DETACH_TEARDOWN   (launcher) 1: return to the caller as soon as the output is written and let a detached child absorb the
                        ~0.3 s the driver needs to tear the process down; default: one process, the caller waits for all of it
"""
import os
import sys
import time

OPTIONS = [  # name, takes value, default text, help
    ("help", False, None, "print the help"),
    ("matches_file", True, None, "path to the matches file"),
    ("output_file", True, None, "path to the output file"),
    ("n_threads", True, "8", "# threads"),
    ("banned_images", True, "{}", "banned images"),
]


def usage():
    lines = ["Options:"]
    for name, takes, default, text in OPTIONS:
        left = "  --" + name
        if takes:
            left += " arg"
            if default is not None:
                left += " (=%s)" % default
        lines.append("%-26s %s" % (left, text) if len(left) < 26 else "%s %s" % (left, text))
    return "\n".join(lines) + "\n"


class FlagError(Exception):
    pass


def parse_args(argv):
    """Boost.program_options-compatible parsing of solve.cc:379-396."""
    names = [o[0] for o in OPTIONS]
    takes = {o[0]: o[1] for o in OPTIONS}
    out = {"help": False, "matches_file": None, "output_file": None, "n_threads": 8, "n_threads_given": False, "banned_images": []}
    seen = set()
    i = 0
    while i < len(argv):
        tok = argv[i]
        i += 1
        if not tok.startswith("--") or tok == "--":
            raise FlagError("too many positional options have been specified on the command line")
        body = tok[2:]
        value = None
        if "=" in body:
            body, value = body.split("=", 1)
        if body in names:
            name = body
        else:
            cands = [n for n in names if n.startswith(body)] if body else []
            if len(cands) == 1:
                name = cands[0]
            elif len(cands) > 1:
                raise FlagError("option '--%s' is ambiguous and matches %s" % (body, ", ".join("'--%s'" % c for c in cands)))
            else:
                raise FlagError("unrecognised option '--%s'" % body)
        if not takes[name]:
            if value is not None:
                raise FlagError("option '--%s' does not take any arguments" % name)
            out[name] = True
            continue
        if value is None:
            if i >= len(argv) or (argv[i].startswith("--") and len(argv[i]) > 2):
                raise FlagError("the required argument for option '--%s' is missing" % name)
            value = argv[i]
            i += 1
        if name == "banned_images":
            out[name].append(value)
            continue
        if name in seen:
            raise FlagError("option '--%s' cannot be specified more than once" % name)
        seen.add(name)
        if name == "n_threads":
            try:
                out[name] = int(value)
                out["n_threads_given"] = True
                if out[name] < 0:
                    raise ValueError
            except ValueError:
                raise FlagError("the argument ('%s') for option '--n_threads' is invalid" % value)
        else:
            out[name] = value
    if out["help"]:
        return out
    for req in ("matches_file", "output_file"):
        if out[req] is None:
            raise FlagError("the option '--%s' is required but missing" % req)
    return out


def main(argv=None, exiting=False):
    """exiting=True: the caller ends the process with os._exit right after (the launcher does): the kernel warm-up thread is
    not waited for.  Otherwise main() joins it before it returns, so no HIP call of this run outlives the call."""
    state = {}
    try:
        return _main(argv, state)
    finally:
        th = state.get("warm")
        if th is not None and not exiting:
            th.join()


def _main(argv, state):
    t_main = time.perf_counter()
    argv = sys.argv[1:] if argv is None else argv
    try:
        args = parse_args(argv)
    except FlagError as e:                                    # solve.cc:397-401
        sys.stderr.write("ERROR: %s\n\n" % e)
        sys.stderr.write(usage())
        return 1
    if args["help"]:                                          # solve.cc:391-394
        sys.stdout.write("Patch Match graph problem solver\n\n" + usage())
        return 0

    # First thing: load the library and start creating the HIP context on a side thread (a few hundred ms, the longest
    # fixed cost of a run) - before numpy is even imported; kernel resolution and the slab caches follow on that thread
    # while this one parses (sizes guessed from the file size: ~230 B per match on the wire, ~2.8 matches per node).
    import ctypes
    import threading
    gpus = [int(x) for x in os.environ.get("LFR_GPUS", "").split(",") if x.strip() != ""]
    device = gpus[0] if len(gpus) == 1 else int(os.environ.get("LFR_DEVICE", "0"))
    if args["n_threads_given"] and "LFR_HOST_THREADS" not in os.environ:
        os.environ["LFR_HOST_THREADS"] = str(max(1, args["n_threads"]))      # read by the scanner / host graph stage
    try:
        nbytes = os.path.getsize(args["matches_file"])
    except OSError:
        nbytes = 0
    warm_ms = [0.0]
    lib_path = os.environ.get("LFR_LIB_OVERRIDE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblfr_hip.so")
    try:
        raw = ctypes.CDLL(lib_path)
    except OSError as e:
        sys.stderr.write("FATAL: %s (build it with `python __graft_entry__.py`; there is no CPU fallback)\n" % e)
        return 2

    ctx_ready = threading.Event()
    # The side thread creates the HIP context (~0.2 s) and loads the code of every kernel the pipeline launches (a toy graph through
    # the whole pipeline + the large-input sort / scan kernels of rocPRIM over scratch memory: ~0.15 s, lfr_hip_warmup level 1).  The
    # timed spans start when it is done: the code has to be loaded either way, and a pipeline that starts while the warm-up is still
    # issuing work queues behind it on the same stream (measured: "Total time" 15-50 ms instead of ~5, the same wall clock).
    # LFR_WAIT_WARMUP=0 restores the round-2 behaviour (wait for the context only).
    wait_all = os.environ.get("LFR_WAIT_WARMUP", "1") != "0"
    os.environ.setdefault("LFR_WARMUP_LEVEL", "1")

    def _warm():
        t0 = time.perf_counter()
        if nbytes > 0:      # creates the context (~0.2 s) and the slab caches; everything after it is optional
            raw.lfr_hip_reserve(ctypes.c_int(device), ctypes.c_int64(int(nbytes / 230 / 2.8 * 1.1)), ctypes.c_int64(int(nbytes / 230 * 1.1)))
        ctx_ready.set()
        # kernel resolution with a toy graph (and, at level 2, a million-match graph: ~0.35 s): whatever is done when the parse
        # ends is time the pipeline does not spend loading code (LFR_WAIT_WARMUP=1: wait for all of it, the steadiest
        # "Total time"; LFR_WARMUP_LEVEL=0: context only)
        raw.lfr_hip_warmup(ctypes.c_int(device))
        warm_ms[0] = (time.perf_counter() - t0) * 1e3
    warm = threading.Thread(target=_warm, daemon=True)
    warm.start()
    state["warm"] = warm

    def warm_join():
        (warm.join if wait_all else ctx_ready.wait)()
    import numpy as np
    t_imp = time.perf_counter()
    try:
        from . import capi
        capi.lib()
    except (ImportError, OSError) as e:
        sys.stderr.write("FATAL: %s\n" % e)
        return 2
    device_pipeline = os.environ.get("LFR_HOST_ASSEMBLY") != "1" and os.environ.get("LFR_HOST_GRAPH_STAGE") != "1"
    # one GPU: the scanner sends the flows to HBM as soon as they are in place (beside the node numbering), the rest of the graph
    # follows at the end of the parse; LFR_UPLOAD_IN_TOTAL=1 keeps the graph on the host until the "Total time" span has started
    ingest_to_device = device_pipeline and len(gpus) <= 1 and os.environ.get("LFR_UPLOAD_IN_TOTAL") != "1" \
        and os.environ.get("LFR_INGEST_TO_DEVICE", "1") != "0"
    try:
        graph = capi.Graph.from_matches_file(args["matches_file"], args["banned_images"], device=device if ingest_to_device else None)
    except capi.LfrError as e:
        if e.code == -3:
            sys.stderr.write("Failed to parse proto object.\n")          # solve.cc:433-436
            return 255
        if e.code in (-4, -6):                                            # no usable GPU / out of device memory: as from to_device below
            sys.stderr.write("FATAL: %s\n" % e)
            return 2
        sys.stderr.write("%s\n" % e)
        return 255
    t_parsed = time.perf_counter()
    print("# graph nodes: %d" % graph.n_nodes)                            # solve.cc:484
    print("# graph edges: %d" % graph.n_edges)                            # solve.cc:485
    sys.stdout.flush()
    if device_pipeline and len(gpus) <= 1 and graph.n_nodes > 0 and os.environ.get("LFR_UPLOAD_IN_TOTAL") != "1":
        warm_join()
        try:
            t_up = time.perf_counter()
            graph.to_device(device)               # last step of ingest: asynchronous, the pipeline continues from it
            if os.environ.get("LFR_VERBOSE") == "2":            # diagnostics: how long does the upload itself take here?
                capi.lib().lfr_hip_synchronize(device)
                sys.stderr.write("lfr: upload of the graph %.1f ms (waited for; normally it overlaps the graph stage)\n"
                                 % ((time.perf_counter() - t_up) * 1e3))
        except capi.LfrError as e:
            sys.stderr.write("FATAL: %s\n" % e)
            return 2

    t_start = time.perf_counter()                                         # solve.cc:487
    override = None
    comp_file = os.environ.get("LFR_COMPONENTS_FILE")
    if comp_file:
        override = np.fromfile(comp_file, dtype="<i8")
        if override.shape[0] != graph.n_nodes:
            sys.stderr.write("FATAL: %s holds %d ids, the graph has %d nodes\n" % (comp_file, override.shape[0], graph.n_nodes))
            return 2
    n_nodes = graph.n_nodes
    positions = np.zeros((n_nodes, 2), np.float64)                        # solve.cc:609-612
    sharded_multi = (n_nodes > 0 and len(gpus) > 1 and override is None and os.environ.get("LFR_HOST_ASSEMBLY") != "1"
                     and os.environ.get("LFR_HOST_GRAPH_STAGE") != "1" and os.environ.get("LFR_MULTI_SHARED_STAGE") != "1")
    if sharded_multi:
        # several GPUs, one process: every GPU runs the graph stage, the assembly and the solve over ITS connected components of the
        # match graph (lfr_solve_graph_hip_multi).  The stdout lines of solve.cc:534-641 come out in the reference's order once the
        # shards' counts are merged; LFR_MULTI_SHARED_STAGE=1: the graph stage on the first GPU, the solve sharded (round 5).
        warm_join()
        try:
            variant = os.environ.get("LFR_TUKEY_VARIANT", "ceres1")
            t1 = time.perf_counter()
            positions, st, sst = capi.solve_graph_hip_multi(graph, gpus, 0, variant)
            t2 = time.perf_counter()
        except (capi.LfrError, KeyError) as e:
            sys.stderr.write("FATAL: HIP solve failed: %s\n" % e)
            return 2
        print("# tracks: %d" % st["n_tracks"])                            # solve.cc:534
        print("max track size: %d" % st["max_track_size"])                # solve.cc:549
        print("Graph-cut time: %dms" % int(st["graph_cut_ms"]))           # solve.cc:589
        print("# components: %d" % st["n_components"])                    # solve.cc:591
        print("max component size: %d" % st["max_component_size"])        # solve.cc:606
        if st["n_cut_components"]:
            sys.stderr.write("note: %d component(s) above the size cap (#images nodes) were split by the built-in deterministic "
                             "bisection, not by COLMAP/Graclus as the reference does (solve.cc:192): the refined positions of this input are "
                             "NOT reference-equivalent; supply the reference's components through LFR_COMPONENTS_FILE for that "
                             "(DESIGN.md, section 3)\n" % st["n_cut_components"])
        # (the shards' graph stages and solves are one call: "Solver time" is that call less the slowest shard's graph stage)
        stage_ms = st["tracks_ms"] + st["roots_ms"] + st["graph_cut_ms"]
        print("Solver time: %dms" % int(max(0.0, (t2 - t1) * 1e3 - stage_ms)))     # solve.cc:638
        print("Total time: %dms" % int((t2 - t_start) * 1e3))             # solve.cc:641
    elif n_nodes > 0:
        try:
            # graph stage + batch assembly on the GPU (falls back to the host stage for the graph cut /
            # huge connected components); LFR_HOST_GRAPH_STAGE=1 / LFR_HOST_ASSEMBLY=1 force the host paths
            if os.environ.get("LFR_HOST_ASSEMBLY") == "1":
                problem = capi.Problem(graph, 0, override)
            elif os.environ.get("LFR_HOST_GRAPH_STAGE") == "1":
                problem = capi.Problem(graph, 0, override, device_assembly=True)
            else:
                warm_join()
                # several GPUs: each assembles its own shard and gathers that shard's flows zero-copy
                problem = capi.Problem(graph, 0, override, device_graph_stage=device,
                                       flags=capi.FLOWS_STAY_ON_HOST if len(gpus) > 1 else 0)
        except capi.LfrError as e:
            sys.stderr.write("FATAL: %s\n" % e)
            return 2
        st = problem.stats()
        t_graph = time.perf_counter()
        print("# tracks: %d" % st["n_tracks"])                            # solve.cc:534
        print("max track size: %d" % st["max_track_size"])                # solve.cc:549
        print("Graph-cut time: %dms" % int(st["graph_cut_ms"]))           # solve.cc:589
        print("# components: %d" % st["n_components"])                    # solve.cc:591
        print("max component size: %d" % st["max_component_size"])        # solve.cc:606
        if st["n_cut_components"]:
            sys.stderr.write("note: %d component(s) above the size cap (#images nodes) were split by the built-in deterministic "
                             "bisection, not by COLMAP/Graclus as the reference does (solve.cc:192): the refined positions of this input are "
                             "NOT reference-equivalent; supply the reference's components through LFR_COMPONENTS_FILE for that "
                             "(DESIGN.md, section 3)\n" % st["n_cut_components"])
        sys.stdout.flush()
        warm_join()
        t1 = time.perf_counter()                                          # solve.cc:615
        try:
            variant = os.environ.get("LFR_TUKEY_VARIANT", "ceres1")
            if st["n_components"] < st["n_tracks"]:       # some component couples tracks through inter-track (Tukey) edges
                sys.stderr.write("note: TukeyLoss flavour %s (%s; the reference pins no Ceres version - set "
                                 "LFR_TUKEY_VARIANT to switch)\n" % (variant, "Ceres <= 1.14" if variant == "ceres1" else "Ceres >= 2.0"))
            if len(gpus) > 1:
                positions, sst = capi.solve_hip_multi(problem, gpus, variant)
            else:
                # batch on the device, solve, positions in pinned host memory: written to the SolutionFile from there (no copy into a
                # fresh array); the statistics (every descriptor and per-component record to the host, 15-20 ms for 147 k components)
                # only when somebody reads them
                batch = capi.Batch(problem, device, tukey_variant=variant)
                t_b = time.perf_counter()
                sst = batch.solve(None, want_stats=bool(os.environ.get("LFR_VERBOSE")))
                t_s = time.perf_counter()
                positions = batch.positions_view()
                if os.environ.get("LFR_TIMING"):
                    sys.stderr.write("lfr: one-shot wall: graph stage %.2f ms, prints %.2f ms, batch (assembly issued) %.2f ms, solve issued %.2f ms, "
                                     "positions on the host %.2f ms\n" % ((t_graph - t_start) * 1e3, (t1 - t_graph) * 1e3, (t_b - t1) * 1e3,
                                                                       (t_s - t_b) * 1e3, (time.perf_counter() - t_s) * 1e3))
                if sst is not None:
                    sst["d2h_ms"] = 0.0
        except (capi.LfrError, KeyError) as e:
            sys.stderr.write("FATAL: HIP solve failed: %s\n" % e)
            return 2
        t2 = time.perf_counter()
        print("Solver time: %dms" % int((t2 - t1) * 1e3))                 # solve.cc:638
        print("Total time: %dms" % int((t2 - t_start) * 1e3))             # solve.cc:641
        if os.environ.get("LFR_VERBOSE"):
            sys.stderr.write("lfr: graph stage wall %.3f ms (device: tracks %.3f roots %.3f components %.3f ms, %d union-find rounds), "
                             "solve kernels %.3f ms, assembly %.3f ms, d2h %.3f ms, %d components (%d failed, %d not converged)\n"
                             % ((t_graph - t_start) * 1e3, st["tracks_ms"], st["roots_ms"], st["graph_cut_ms"], int(st.get("kruskal_rounds", 0)),
                                sst["kernel_ms"], sst["h2d_ms"], sst["d2h_ms"], sst["n_components"], sst["n_failed"], sst["n_no_convergence"]))
    t_solved = time.perf_counter()
    try:
        n_outside = graph.write_solution(positions, args["output_file"])
    except capi.LfrError:
        print("# points with at least one coordinate > 0.5: %d" % int((abs(positions) > 0.5).any(axis=1).sum()))
        sys.stderr.write("Failed to write proto object.\n")               # solve.cc:674-677
        return 255
    print("# points with at least one coordinate > 0.5: %d" % n_outside)  # solve.cc:670
    if os.environ.get("LFR_VERBOSE"):
        t_end = time.perf_counter()
        sys.stderr.write("lfr: wall inside main %.3f s = imports %.3f + parse %.3f + upload/graph stage/solve %.3f + write %.3f "
                         "(HIP context + kernel warm-up on the side thread: %.0f ms)\n"
                         % (t_end - t_main, t_imp - t_main, t_parsed - t_imp, t_solved - t_parsed, t_end - t_solved, warm_ms[0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())

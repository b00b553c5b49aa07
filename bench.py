#!/usr/bin/env python3
"""Headline benchmark: edges/s (+ tracks/s) of the multi-view refinement solver on the 5M-edge synthetic match
graph of BASELINE.json (configs[3] / SURVEY.md §8(d) "config 4": 1344 images, ~147k tracks, ~5.0M directed edges).

`value` (the bench contract): K steps of the hot path with its inputs resident in HBM - every solve kernel over
the assembled batch; the output array is fully rewritten by each step.
Next to it, first class, the reference's own two spans, each over --span-reps one-shot repetitions:
  solver_span   solve.cc:615-638  problem construction (device assembly of the batch) + solve + results on the host
  total_span    solve.cc:487-641  + tracks / roots / components; the graph starts on the HOST, its PCIe upload is inside
  total_span_resident_graph       the same with the graph already in HBM (streamed-ingest / device-producer contract)
and the CPU baseline over the same two spans.

N > 1: one process per GPU (torch.distributed / RCCL), no data-path collective (components are independent,
solve.cc:594-597); the statistics vector is all-reduced once for reporting.  The default for N > 1 is --scaling strong
(BASELINE.json's configuration: ONE 5M-edge graph, seed 2, its components sharded over the ranks on the device - the
same graph as the N = 1 run, so a scaling curve compares like with like); the line then carries a `weak_scaling`
object (every rank its OWN 5M-edge graph, seed 2 + rank).  --scaling weak makes that the headline instead and
carries a `strong_scaling` object.  `python bench.py --gpus N` without a launcher environment starts the N ranks itself.

    python bench.py --gpus 1 --steps 20 --warmup 3
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6         # MI355X fp64 vector peak = 1/2 of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md)
FLOP_PER_EDGE_EVAL = 200        # ~200 fp64 flop per edge evaluation (interpolant + partials + loss + corrector)
EDGE_BYTES = 84                 # u32 src + u32 dst + f32 sim + 18 x f32 flow per evaluation pass
NODE_BYTES = 32                 # 16 B position read + 16 B written per variable node per pass


def kernel_source_sha256():
    """sha256 over the kernel sources a PMC profile depends on: a committed counter file carries the hash of the
    sources it was collected with; when they differ from the tree this run was built from, its numbers are not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in ("lfr_solve.hip", "lfr_device.hpp"):
        with open(os.path.join(ROOT, "local-feature-refinement_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def fresh(d):
    """the committed PMC summary `d` was collected with the kernel sources of this tree"""
    return bool(d) and d.get("kernel_source_sha256") == kernel_source_sha256()


def cpu_leg(ma, comp_override, n_threads, n_edges):
    """The C restatement (oracle/lfr_oracle.c, -O3 -march=native, built on this box; envelope Cholesky above 192 rows) on one of the
    secondary workloads at `n_threads` threads, components taken from the product (the Graclus cut cannot be restated): Solver span only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lfr_oracle
    best = None
    for _ in range(2):
        r = lfr_oracle.run(ma, n_threads=n_threads, native=True, comp_override=comp_override)
        if r["rc"] != 0:
            return {"error": "oracle rc %d" % r["rc"]}
        if best is None or r["solver_ms"] < best["solver_ms"]:
            best = r
    return {"threads": n_threads, "kind": "port", "solver_span_ms": best["solver_ms"], "edges_per_s": n_edges / (best["solver_ms"] * 1e-3),
            "what": "C restatement of the Ceres path (not Ceres), assembly + solve of all components (solve.cc:615-638), best of two runs; "
                    "systems above 192 rows are factored inside the envelope of a reverse Cuthill-McKee order (a sparse direct solver, "
                    "like the reference's SPARSE_NORMAL_CHOLESKY), smaller ones dense"}, best


def profile_numbers(kernel, n_edges):
    """HBM bytes per launch and VALU busy of the dominant kernel FROM THE COMMITTED rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE / WRITE_SIZE / SQ counters collected in separate --pmc runs of this same
    command, corrected as MI355X_MICROARCH.md prescribes) - not measured in this run.  None when the profile is for
    a different kernel/workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if d.get("kernel") != kernel or d.get("edges_per_launch") != n_edges:
        return None
    return d


def pmc_numbers(name):
    """a committed PMC summary under profiles/ (None when it is not there)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: become `torch.distributed.run` with N ranks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def spans(fn, reps, sync):
    """median / min of `reps` one-shot repetitions of fn() (wall clock, device idle before and after)."""
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return {"ms": statistics.median(ts), "min_ms": min(ts), "reps": reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tracks", type=int, default=147_000, help="tracks of the synthetic graph (147000 -> ~5.0M edges)")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="auto: strong (ONE config-4 graph sharded over the ranks) when N > 1; N = 1 is the same either way")
    ap.add_argument("--span-reps", type=int, default=7)
    ap.add_argument("--devices", default="", help="comma-separated HIP device per rank (default: the local rank); with LFR_DIST_BACKEND=gloo "
                                                  "several ranks may share one GPU (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-tracks", action="store_true", help="skip the second, workgroup-kernel dominated workload (config 5 stand-in)")
    ap.add_argument("--no-sparse", action="store_true", help="skip the third workload (cap-sized sparse components, elimination-tree kernel)")
    ap.add_argument("--sparse-tracks", type=int, default=12000)
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import numpy as np
    import torch
    from lfr_amd import capi, dist, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver path has no CPU fallback)")
    rank, world, local = dist.init(backend=os.environ.get("LFR_DIST_BACKEND") or None)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.devices:
        local = [int(x) for x in args.devices.split(",")][rank]
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    L = capi.lib()
    strong = args.scaling in ("auto", "strong") and world > 1

    # ---- workload: config 4; weak: one graph per rank, strong: one graph for all ----
    t0 = time.perf_counter()
    ma = synthetic.config4(n_tracks=args.tracks, seed=2 if strong else 2 + rank)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    graph = capi.Graph.from_arrays(ma)              # the parsed graph of solve.cc:405-481 (pinned host arrays)
    t_ingest = time.perf_counter() - t0
    L.lfr_hip_warmup(local)
    L.lfr_hip_reserve(local, graph.n_nodes, graph.n_edges // 2)
    stream = torch.cuda.current_stream().cuda_stream     # kernels + HIP events run on torch's stream
    sync = torch.cuda.synchronize
    shard = (rank, world) if strong else (0, 1)
    flags = capi.FLOWS_STAY_ON_HOST if strong else 0

    def make_batch(problem):
        """the batch of this rank: a problem whose graph stage already ran over this rank's connected components only is assembled whole;
        otherwise (one rank, or one giant connected component) the COMPONENTS are dealt out at assembly (snake deal)"""
        return capi.Batch(problem, local) if problem.cc_sharded else capi.Batch(problem, local, shard[0], shard[1])

    def pipeline(keep=None):
        """solve.cc:487-641 on the GPU: graph stage -> batch assembly -> solve -> positions on the host.  N > 1 (strong): every rank runs
        the union-find pass over the endpoints, then tracks / roots / components / assembly / solve over ITS connected components only."""
        problem = capi.Problem(graph, device_graph_stage=local, flags=flags, shard=shard if strong else None)
        batch = make_batch(problem)
        batch.solve(stream, want_stats=False)
        pos = batch.positions_view_f32()
        if keep is not None:
            keep.extend([problem, batch, pos])

    # ---- the reference's spans, one-shot repetitions (first-class numbers, never `value`) ----
    kept = []
    pipeline(kept)                                   # untimed: first-launch costs, slab caches
    problem, batch = kept[0], kept[1]
    pst = problem.stats()

    def total_cold():
        graph.evict_device()                         # the graph starts on the host: PCIe upload inside the span
        if world > 1:
            dist.barrier()
        pipeline()

    def total_resident():
        if world > 1:
            dist.barrier()
        pipeline()

    def solver_only():
        b = make_batch(problem)                                  # problem construction: device assembly of the batch
        b.solve(stream, want_stats=False)
        b.positions_view_f32()

    def d2h(fn):
        ts = []
        for _ in range(5):
            sync(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)
    positions_ms = {"f32": d2h(batch.positions_view_f32), "f64": d2h(batch.positions_view),
                    "what": "the spans read the results back as float32 (converted on the device: the precision the reference's SolutionFile "
                            "holds, solve.cc:661-664); f64 = the same read-back in the solver's own precision (until round 5 inside the spans)"}

    reps = max(1, args.span_reps)
    sp_total = spans(total_cold, reps, sync)
    graph.to_device(local)
    sp_total_res = spans(total_resident, reps, sync)
    sp_solver = spans(solver_only, reps, sync)
    for sp in (sp_total, sp_total_res, sp_solver):   # a span ends when the slowest rank is done
        sp["ms"] = dist.max_over_ranks(sp["ms"])
        sp["min_ms"] = dist.max_over_ranks(sp["min_ms"])

    # ---- the timed steps: every solve kernel over the resident batch ----
    for _ in range(args.warmup):
        batch.solve(stream, want_stats=False)
    torch.cuda.synchronize()
    dist.barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        batch.solve(stream, want_stats=False)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t_start
    elapsed_local = elapsed
    elapsed = dist.max_over_ranks(elapsed)

    # ---- per-kernel durations of the timed steps (HIP events recorded on the launch stream) ----
    n_ev = min(args.steps, 64)
    cls_ms = np.zeros(capi.NUM_KERNEL_CLASSES)
    tot_ms = 0.0
    for back in range(n_ev):
        t, c, cls_edges = batch.timing(back)
        tot_ms += t
        cls_ms += c
    cls_ms /= n_ev
    tot_ms /= n_ev
    st = batch.solve(stream, want_stats=True)            # deterministic: same pass counts as every timed step
    dom = int(np.argmax(cls_ms))
    glob = dist.allreduce_stats(st)

    edges_total = glob["n_edges"]
    tracks_total = glob["n_tracks"]
    value = edges_total * args.steps / elapsed
    ms_step = elapsed / args.steps * 1e3

    def rate(sp):
        return {"ms": sp["ms"], "min_ms": sp["min_ms"], "reps": sp["reps"], "edges_per_s": edges_total / (sp["ms"] * 1e-3),
                "tracks_per_s": tracks_total / (sp["ms"] * 1e-3)}

    res = {
        "metric": "edges_per_s", "value": value, "unit": "edges/s",
        "tracks_per_s": tracks_total * args.steps / elapsed,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config4: synthetic match graph, 1344 images, %d tracks%s (mean length 6), %d directed edges%s, "
                               "Gaussian flows, seed %s" % (args.tracks, "" if strong else "/GPU", st["n_edges"] if not strong else edges_total,
                                                            "" if strong else "/GPU", "2" if strong else "2+rank"),
                   "step": "all solve kernels over the HBM-resident batch (inputs resident when the timed region starts)",
                   "edges_per_gpu": st["n_edges"], "tracks_per_gpu": st["n_tracks"], "components_per_gpu": st["n_components"],
                   "parallelism": ("components sharded, %d rank(s), no data-path collective" % world) +
                                  (" (graph stage, assembly and solve per rank over its connected components of the match graph)" if strong and problem.cc_sharded else "")},
        # the reference's own spans, one-shot, median of `reps` (max over ranks)
        "solver_span": dict(rate(sp_solver), what="solve.cc:615-638: device assembly of the batch (problem construction) + solve + positions on the host (float32)"),
        "positions_d2h_ms": positions_ms,
        "total_span": dict(rate(sp_total), what="solve.cc:487-641: graph stage + assembly + solve + positions on the host; the graph starts in "
                                                "(pinned) host memory, its PCIe upload (%.0f MB) is inside" % (graph.n_edges / 2 * 156e-6)),
        "total_span_resident_graph": dict(rate(sp_total_res), what="solve.cc:487-641 with the match graph already in HBM (lfr_graph_to_device at ingest / "
                                                                   "device-producer contract): no PCIe upload inside"),
    }
    if rank == 0:
        serial = os.environ.get("LFR_SERIAL_CLASSES") == "1"
        kernel_names = ["solve_group_kernel<8,1,3>", "solve_group_kernel<16,1,6>", "(retired)",
                        "solve_group_kernel<32,1,6>", "solve_group_kernel<32,2,5>", "solve_block_kernel<lds,rows<=88>",
                        "solve_block_kernel<lds,rows<=130>", "solve_block_kernel<lds,rows<=192>", "solve_tree_kernel<elimination tree,hbm>"]
        if not serial:      # one launch for all packed classes; its events sit in the slot of the largest class
            kernel_names[dom if dom < 5 else 0] = "solve_packed_kernel"
        dur_s = cls_ms[dom] * 1e-3
        # physical roofs of the dominant launch.  The kernel keeps its edges in VGPRs: it reads HBM once
        # (80-B record per edge, 4-B id + 16-B position per node) and is bound by fp64 VALU issue/latency.
        b_once = st["dominant_kernel_edges"] * 80 + st["dominant_kernel_nodes"] * 20
        exec_evals = st["exec_passes_edges"] * (st["dominant_kernel_edges"] / max(1, st["n_edges"]))   # edge evaluations executed by that launch
        flops = exec_evals * FLOP_PER_EDGE_EVAL
        prof = profile_numbers(kernel_names[dom], int(st["dominant_kernel_edges"]))
        prof_stale = bool(prof) and not fresh(prof)
        if prof_stale:
            prof = None                                   # counters of other kernel sources: not this build's
        traffic = prof.get("hbm_bytes_per_launch") if prof else None
        hbm_bytes = max(b_once, traffic or 0)
        b_stream = st["dominant_ref_passes_edges"] * EDGE_BYTES + st["dominant_ref_passes_nodes"] * NODE_BYTES
        res["roofline"] = {
            "bound": "fp64-valu", "kernel": kernel_names[dom],
            "achieved": flops / dur_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / dur_s / 1e12 / FP64_PEAK_TFLOPS,
            "flops_per_launch": int(flops), "edge_evaluations_executed": int(exec_evals),
            "flop_per_edge_evaluation": FLOP_PER_EDGE_EVAL,
            "launch_ms": cls_ms[dom], "launch_edges": int(st["dominant_kernel_edges"]),
            "hbm": {"achieved": hbm_bytes / dur_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_bytes / dur_s / 1e9 / HBM_PEAK_GBPS,
                    "bytes_per_launch": int(hbm_bytes), "read_once_bytes": int(b_once)},
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_traffic.json (committed rocprofv3 PMC passes of this command with these kernel sources - kernel_source_sha256 matches; "
                               "not measured in this run)" if traffic else
                               "none: profiles/pmc_traffic.json was collected with other kernel sources (kernel_source_sha256 differs)" if prof_stale else None),
            "kernel_source_sha256": kernel_source_sha256(),
            "valu_busy": (prof or {}).get("valu_busy"),
            "streaming_equiv": {"bytes": int(b_stream), "GBps": b_stream / dur_s / 1e9,
                                "passes_per_edge_reference": st["dominant_ref_passes_edges"] / max(1, st["dominant_kernel_edges"]),
                                "note": "SURVEY 8(d) bookkeeping only: 84 B/edge + 32 B/node for every evaluation pass Ceres would perform; "
                                        "the kernel does not move these bytes, so this is not a roofline fraction"},
        }
        res["all_kernels_ms"] = tot_ms
        keep = range(9) if serial else [dom if dom < 5 else 0, 5, 6, 7, 8]
        res["class_ms"] = {kernel_names[i]: round(float(cls_ms[i]), 4) for i in keep if cls_edges[i] > 0}
        res["class_edges"] = {kernel_names[i]: int(cls_edges[i]) for i in keep if cls_edges[i] > 0}
        res["setup_ms"] = {"generate": t_gen * 1e3, "ingest_arrays": t_ingest * 1e3,
                           "graph_stage_kernels": {"tracks": pst["tracks_ms"], "roots": pst["roots_ms"], "components": pst["graph_cut_ms"]},
                           "assembly_incl_flow_wait": st["h2d_ms"]}
        res["solve"] = {"converged": st["n_converged"], "no_convergence": st["n_no_convergence"], "failed": st["n_failed"],
                        "mean_iterations": st["sum_iterations"] / max(1, st["n_components"])}
    if world > 1:
        # Self-verification of a multi-GPU line: how many ranks the process group really had (an all-reduced count of
        # ones), which device each rank ran on (name, PCI bus id / uuid where torch exposes them), what each rank solved and how long
        # its K steps took locally - the headline is total edges / the slowest rank's time.
        props = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "local_device": int(local), "name": torch.cuda.get_device_name(local),
              "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None,
              "edges": int(st["n_edges"]), "components": int(st["n_components"]), "ms_per_step_local": elapsed_local / args.steps * 1e3,
              "cc_sharded": bool(problem.cc_sharded), "pid": os.getpid()}
        ranks_seen = int(round(dist.sum_over_ranks(1.0)))
        per_rank = dist.gather_objects(me)
        if rank == 0:
            ms = [r["ms_per_step_local"] for r in per_rank]
            res["multi_gpu"] = {"ranks_seen": ranks_seen, "backend": torch.distributed.get_backend(), "per_rank": per_rank,
                                "distinct_devices": len({(r["pci_bus_id"], r["uuid"], r["local_device"]) for r in per_rank}),
                                "edges_sum_over_ranks": int(sum(r["edges"] for r in per_rank)), "edges_headline": int(edges_total),
                                "ms_per_step_min": min(ms), "ms_per_step_max": max(ms)}
            n1 = os.environ.get("LFR_BENCH_N1_VALUE")
            if n1:
                res["multi_gpu"]["efficiency_vs_n1"] = value / (world * float(n1))
                res["multi_gpu"]["n1_value"] = float(n1)
    if world > 1 and not strong:
        # strong scaling beside the weak headline: ONE graph (seed 2), components sharded on the device, Total span
        mas = ma if rank == 0 else synthetic.config4(n_tracks=args.tracks, seed=2)
        gs = graph if rank == 0 else capi.Graph.from_arrays(mas)
        gs.evict_device()

        def strong_total():
            dist.barrier()
            p = capi.Problem(gs, device_graph_stage=local, flags=capi.FLOWS_STAY_ON_HOST, shard=(rank, world))
            b = capi.Batch(p, local) if p.cc_sharded else capi.Batch(p, local, rank, world)
            b.solve(stream, want_stats=False)
            b.positions_view_f32()
            return b
        bs = strong_total()
        sst = dist.allreduce_stats(bs.solve(stream, want_stats=True))
        sp = spans(strong_total, reps, sync)
        sp["ms"] = dist.max_over_ranks(sp["ms"]); sp["min_ms"] = dist.max_over_ranks(sp["min_ms"])
        if rank == 0:
            res["strong_scaling"] = {"total_span_ms": sp["ms"], "min_ms": sp["min_ms"], "reps": reps, "edges": sst["n_edges"],
                                     "edges_per_s": sst["n_edges"] / (sp["ms"] * 1e-3), "tracks_per_s": sst["n_tracks"] / (sp["ms"] * 1e-3),
                                     "what": "solve.cc:487-641 on ONE config-4 graph (seed 2): every rank runs the integer graph stage, assembles and "
                                             "solves its shard (flows gathered zero-copy from pinned host memory), positions of the shard on the host"}
            # what the design predicts for this span: every rank repeats the serial part - endpoints / similarities / node
            # images over PCIe (20 B per match + 4 B per node) and the graph stage - and gets 1/N of the rest: the flows it gathers over its
            # own PCIe link (144 B per match of its shard) and the Solver span of rank 0's own one-GPU run
            pcie = 57e9
            n_m, n_n = sst["n_edges"] / 2.0, float(graph.n_nodes)
            serial_ms = (sp_total_res["ms"] - sp_solver["ms"]) + (20.0 * n_m + 4.0 * n_n) / pcie * 1e3
            parallel_ms = sp_solver["ms"] + 144.0 * n_m / pcie * 1e3
            res["strong_scaling"].update({"predicted_ms": serial_ms + parallel_ms / world, "predicted_serial_ms": serial_ms,
                                          "predicted_parallel_ms": parallel_ms,
                                          "model": "serial (graph stage + 20 B/match + 4 B/node over PCIe at 57 GB/s, repeated by every rank) + "
                                                   "(one-GPU Solver span + 144 B/match of flows over PCIe) / N"})
    if world > 1 and strong:
        # weak scaling beside the strong headline: every rank its OWN 5M-edge graph (seed 2 + rank), whole batches, the same K steps
        maw = ma if rank == 0 else synthetic.config4(n_tracks=args.tracks, seed=2 + rank)
        gw = graph if rank == 0 else capi.Graph.from_arrays(maw)
        gw.to_device(local)
        pw = capi.Problem(gw, device_graph_stage=local)
        bw = capi.Batch(pw, local)
        for _ in range(max(1, args.warmup)):
            bw.solve(stream, want_stats=False)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            bw.solve(stream, want_stats=False)
        torch.cuda.synchronize()
        dist.barrier()
        el_w = dist.max_over_ranks(time.perf_counter() - t0)
        wst = dist.allreduce_stats(bw.solve(stream, want_stats=True))
        if rank == 0:
            res["weak_scaling"] = {"value": wst["n_edges"] * args.steps / el_w, "unit": "edges/s", "ms_per_step": el_w / args.steps * 1e3,
                                   "edges": wst["n_edges"], "tracks_per_s": wst["n_tracks"] * args.steps / el_w, "failed": wst["n_failed"],
                                   "what": "every rank solves its OWN config-4 graph (seed 2 + rank): %d steps of all solve kernels over the "
                                           "HBM-resident batches, max over ranks; linear by construction (no data-path collective)" % args.steps}
        del bw, pw
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import lfr_oracle
            cores = args.cpu_threads or os.cpu_count() or 1
            # The C restatement of the Ceres path at -O3 -march=native (built here, on the box that times it), one component per task
            # from an atomic cursor, largest first (solve.cc:599-634), per-thread scratch reused across components.  Timed at the
            # reference's default of 8 threads (solve.cc:384) and at every core of the box; each the best of two runs.
            runs = {}
            for nt in sorted({min(8, cores), cores}):
                best = None
                for _ in range(2):
                    r = lfr_oracle.run(ma, n_threads=nt, native=True)
                    if best is None or r["solver_ms"] + r["graph_ms"] < best["solver_ms"] + best["graph_ms"]:
                        best = r
                runs[nt] = best
            nt_best = min(runs, key=lambda k: runs[k]["solver_ms"])
            ref = runs[nt_best]
            err = float(np.abs(batch.download() - ref["positions"]).max())
            evals = float(((ref["infos"]["n_jac_evals"] + ref["infos"]["n_cost_evals"]) * ref["comp_nedges"].astype(np.int64)).sum())

            def leg(nt):
                r = runs[nt]
                return {"threads": nt, "solver_span_ms": r["solver_ms"], "total_span_ms": r["solver_ms"] + r["graph_ms"],
                        "edges_per_s": st["n_edges"] / (r["solver_ms"] * 1e-3),
                        "gflops_per_thread": evals * FLOP_PER_EDGE_EVAL / (r["solver_ms"] * 1e-3) / 1e9 / nt}
            res["cpu_baseline"] = {
                "value": st["n_edges"] / (ref["solver_ms"] * 1e-3), "unit": "edges/s", "cores": nt_best, "kind": "port",
                "sample": "the whole rank-0 graph (%d edges, %d edge evaluations), best of two runs per thread count, C restatement of the Ceres path "
                          "(oracle/lfr_oracle.c, gcc -O3 -march=native -ffp-contract=off, built on this box), not Ceres; value = Solver span "
                          "(solve.cc:615-638: assembly + solve) at the best thread count, to be compared with solver_span, not with `value`"
                          % (st["n_edges"], int(evals)),
                "solver_span": {"ms": ref["solver_ms"], "edges_per_s": st["n_edges"] / (ref["solver_ms"] * 1e-3)},
                "total_span": {"ms": ref["solver_ms"] + ref["graph_ms"], "edges_per_s": st["n_edges"] / ((ref["solver_ms"] + ref["graph_ms"]) * 1e-3)},
                "at_reference_default_8_threads": leg(min(8, cores)), "at_all_cores": leg(cores),
                "max_abs_diff_vs_gpu_units": err,
            }
            res["speedup_vs_cpu_baseline"] = {"solver_span": ref["solver_ms"] / res["solver_span"]["ms"],
                                              "total_span": (ref["solver_ms"] + ref["graph_ms"]) / res["total_span"]["ms"],
                                              "solver_span_vs_8_threads": runs[min(8, cores)]["solver_ms"] / res["solver_span"]["ms"]}
        if not args.no_long_tracks and world == 1:
            # config 4 holds only components of <= 32 rows (packed kernel).  Real long-track data (BASELINE
            # configs[4], ETH3D) runs in the workgroup-per-component kernels: a second, clearly labelled workload.
            t0 = time.perf_counter()
            ma5 = synthetic.config5()
            g5 = capi.Graph.from_arrays(ma5)
            t_prep5 = time.perf_counter() - t0
            L.lfr_hip_reserve(local, g5.n_nodes, g5.n_edges // 2)

            def pipe5(keep=None):
                p5 = capi.Problem(g5, device_graph_stage=local)
                b5 = capi.Batch(p5, local)
                b5.solve(stream, want_stats=False)
                b5.positions_view_f32()
                if keep is not None:
                    keep.extend([p5, b5])
            k5 = []
            pipe5(k5)
            p5, b5 = k5
            g5.to_device(local)
            sp5 = spans(lambda: pipe5(), max(1, min(3, reps)), sync)

            def solver5():                                        # solve.cc:615-638 like for like with the CPU leg: assembly + solve + positions
                bb = capi.Batch(p5, local)
                bb.solve(stream, want_stats=False)
                bb.positions_view_f32()
            sps5 = spans(solver5, max(1, min(3, reps)), sync)
            n5 = 5
            for _ in range(2):
                b5.solve(stream, want_stats=False)
            sync()
            t0 = time.perf_counter()
            for _ in range(n5):
                b5.solve(stream, want_stats=False)
            sync()
            ms5 = (time.perf_counter() - t0) / n5 * 1e3
            st5 = b5.solve(stream, want_stats=True)
            _, c5, e5 = b5.timing(0)
            info5 = b5.component_info()
            rows5 = 2.0 * info5["n_var_nodes"].astype(np.float64)
            wg5 = rows5 > 32                                      # the workgroup classes (a dense LDL^T per LM iteration)
            fact_flops = float((rows5[wg5] ** 3 / 3.0 * info5["iterations"][wg5]).sum())
            eval_flops = float(st5["exec_passes_edges"]) * FLOP_PER_EDGE_EVAL
            prof5 = pmc_numbers("r06_pmc_config5.json")
            stale5 = bool(prof5) and not fresh(prof5)
            if stale5:
                prof5 = None
            traffic5 = (prof5 or {}).get("hbm_bytes_per_solve")
            cpu5 = None
            if not args.no_cpu_baseline:
                cpu5, _ = cpu_leg(ma5, p5.labels()[2], min(8, os.cpu_count() or 1), st5["n_edges"])
            alg_bytes5 = float(st5["exec_passes_edges"]) * 80
            res["long_tracks_workload"] = {
                # Two roofs of the solve as a whole (the three LDS classes run concurrently).  HBM: every sweep re-streams the 80-byte
                # records (the fused sweep of round 3 keeps everything else in registers and LDS; until then 64 B of corrected jacobian
                # per edge went through a scratch array, written once and read twice: 272 B per edge and sweep); bytes from the
                # committed PMC passes.  fp64: n^3/3 per factorization (one per LM iteration) on the fp64 matrix cores + 200 flop per
                # executed edge evaluation on the fp64 VALU, both 78.6 TFLOP/s peak.
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                             "traffic": traffic5,
                             "achieved": alg_bytes5 / (ms5 * 1e-3) / 1e9, "frac": alg_bytes5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "frac_what": "ALGORITHMIC bytes (below) over the step time; traffic_ratio = counter traffic / algorithmic bytes",
                             "traffic_ratio": (traffic5 / alg_bytes5) if traffic5 else None,
                             # the kernels write a few MB of results: WRITE_SIZE is spilled registers going out to scratch memory, and as much comes back
                             "traffic_written": (prof5 or {}).get("hbm_write_bytes_per_solve"),
                             "traffic_ratio_without_spills": ((traffic5 - 2.0 * prof5["hbm_write_bytes_per_solve"]) / alg_bytes5) if traffic5 and prof5.get("hbm_write_bytes_per_solve") else None,
                             "algorithmic_bytes": alg_bytes5,
                             "algorithmic_bytes_what": "per executed sweep and edge: the 80 B record (nothing else leaves the CU)",
                             "traffic_source": ("profiles/r06_pmc_config5.json (committed rocprofv3 PMC passes over this workload with these kernel sources, 2*FETCH_SIZE + WRITE_SIZE; "
                                                "not measured in this run)" if traffic5 else
                                                "none: the committed counters were collected with other kernel sources (kernel_source_sha256 differs)" if stale5 else None),
                             "fp64": {"achieved": (fact_flops + eval_flops) / (ms5 * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": (fact_flops + eval_flops) / (ms5 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                      "factorization_flops": fact_flops, "edge_evaluation_flops": eval_flops,
                                      "factorizations": int(info5["iterations"][wg5].sum()), "edge_evaluations_executed": int(st5["exec_passes_edges"])},
                             "pmc": {k: prof5[k] for k in prof5 if k.startswith("SQ_") or k in ("valu_busy", "wait_fraction")} if prof5 else None},
                "workload": "config5 stand-in: synthetic match graph, 96 images, 2000 tracks of 48-96 nodes, 2 %% wrong matches (components above the "
                            "size cap are cut), %d directed edges, %d components" % (st5["n_edges"], st5["n_components"]),
                "ms_per_step": ms5, "edges_per_s": st5["n_edges"] / (ms5 * 1e-3), "tracks_per_s": st5["n_tracks"] / (ms5 * 1e-3), "steps": n5,
                "kernel_ms": {kernel_names[i]: round(float(c5[i]), 3) for i in range(9) if e5[i] > 0},
                "kernel_edges": {kernel_names[i]: int(e5[i]) for i in range(9) if e5[i] > 0},
                "total_span_resident_graph_ms": sp5["ms"], "graph_stage": {k: p5.stats()[k] for k in ("tracks_ms", "roots_ms", "graph_cut_ms", "kruskal_rounds", "n_cut_components")},
                "solver_span_ms": sps5["ms"],
                "solver_span_what": "solve.cc:615-638 on the GPU, one-shot: device assembly of the batch + solve + positions on the host - the span the CPU leg times",
                "speedup_vs_cpu_baseline_8_threads": (cpu5["solver_span_ms"] / sps5["ms"]) if cpu5 and "solver_span_ms" in cpu5 else None,
                "mean_iterations": st5["sum_iterations"] / max(1, st5["n_components"]), "failed": st5["n_failed"], "no_convergence": st5["n_no_convergence"],
                "setup_s": t_prep5,
                "cpu_baseline": cpu5,
            }
            del b5, p5, g5, ma5
        if not args.no_sparse and world == 1:
            # cap-sized SPARSE components - 1344 images, short tracks matched along a ring lattice, chained by wrong
            # matches; the size cap leaves components of up to 1344 nodes (2.7 k-row systems, tree-plus-few-cycles sparse): the
            # elimination-tree kernel of the HBM class (round 2: a dense packed matrix; round 3: a block envelope factored as one chain of panels)
            t0 = time.perf_counter()
            mas = synthetic.capsized_sparse(n_tracks=args.sparse_tracks)
            gs = capi.Graph.from_arrays(mas)
            t_preps = time.perf_counter() - t0
            ps = capi.Problem(gs, device_graph_stage=local)
            t0 = time.perf_counter()
            bs = capi.Batch(ps, local)
            sync()
            t_batch = time.perf_counter() - t0

            def solver_s():                                       # solve.cc:615-638 like for like with the CPU leg (the elimination-tree plans are made on the host here)
                bb = capi.Batch(ps, local)
                bb.solve(stream, want_stats=False)
                bb.positions_view_f32()
            sps_s = spans(solver_s, max(1, min(3, reps)), sync)
            for _ in range(2):
                bs.solve(stream, want_stats=False)
            sync()
            ns = 3
            t0 = time.perf_counter()
            for _ in range(ns):
                bs.solve(stream, want_stats=False)
            sync()
            mss = (time.perf_counter() - t0) / ns * 1e3
            sts = bs.solve(stream, want_stats=True)
            infos = bs.component_info()
            rowss = 2 * infos["n_var_nodes"]
            bigs = rowss > 192
            _, cs, es_ = bs.timing(0)
            # What the launch executed, counted from the elimination-tree plans (lfr_batch_tree_stats) and the solver's own
            # counters: per factorization of a component 8192 flop per 16x16x16 left-looking update on the fp64 matrix cores, 3840 per
            # stored tile for the elimination / substitution of its 16 rows (240 multiply-adds each) and 512 per tile in the back
            # substitution; 200 flop per edge evaluation, every record evaluated by the owner of either end (2 x).  HBM bytes per
            # factorization + solve: each tile of A read once, each tile of the factor written once, read once by the back substitution
            # and ~1.5 times as an operand of an update (6 KB per tile + 3 KB per update); per sweep 2 x 80 B per record + ~150 B per
            # sweep item (item words, partial sums, the pair's 2x2 block).
            ts_ = bs.tree_stats()
            its = infos["iterations"].astype(np.float64)
            fact_flops_s = float(((ts_["updates"] * 8192.0 + ts_["tiles"] * (3840.0 + 512.0)) * its).sum())
            eval_flops_s = float(sts["exec_passes_edges"]) * 2.0 * FLOP_PER_EDGE_EVAL
            fact_bytes_s = float(((ts_["tiles"] * 6144.0 + ts_["updates"] * 3072.0) * its).sum())
            sweeps_per_edge = float(sts["exec_passes_edges"]) / max(1, sts["n_edges"])
            sweep_bytes_s = float(sts["exec_passes_edges"]) * 160.0 + float(ts_["items"].sum()) * 150.0 * sweeps_per_edge
            prof_s = pmc_numbers("r06_pmc_sparse.json")
            stale_s = bool(prof_s) and not fresh(prof_s)
            if stale_s:
                prof_s = None
            traffic_s = (prof_s or {}).get("hbm_bytes_per_solve")
            cpu_s = None
            if not args.no_cpu_baseline:
                cpu_s, _ = cpu_leg(mas, ps.labels()[2], min(8, os.cpu_count() or 1), sts["n_edges"])
            big_i = int(np.argmax(np.where(bigs, its * ts_["columns"], 0))) if bigs.any() else 0
            res["sparse_capsized_workload"] = {
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                             "achieved": (fact_bytes_s + sweep_bytes_s) / (mss * 1e-3) / 1e9,
                             "frac": (fact_bytes_s + sweep_bytes_s) / (mss * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "frac_what": "ALGORITHMIC bytes over the step time.  Neither roof binds: the launch lasts as long as its slowest component - "
                                          "iterations x (levels of dependent column tasks + sweeps), chains of round trips to L2 on a team of up to 8 workgroups",
                             "algorithmic_bytes": fact_bytes_s + sweep_bytes_s, "factorization_bytes": fact_bytes_s, "sweep_bytes": sweep_bytes_s,
                             "traffic": traffic_s, "traffic_ratio": (traffic_s / (fact_bytes_s + sweep_bytes_s)) if traffic_s else None,
                             "traffic_source": ("profiles/r06_pmc_sparse.json (committed rocprofv3 PMC passes over this workload with these kernel sources; not measured in this run)"
                                                if traffic_s else "none: the committed counters were collected with other kernel sources" if stale_s else None),
                             "fp64": {"achieved": (fact_flops_s + eval_flops_s) / (mss * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": (fact_flops_s + eval_flops_s) / (mss * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                      "factorization_flops": fact_flops_s, "edge_evaluation_flops": eval_flops_s,
                                      "factorizations": int(its[bigs].sum()), "tile_updates_per_solve": float((ts_["updates"] * its).sum())},
                             "pmc": {k: prof_s[k] for k in prof_s if k.startswith("SQ_") or k in ("valu_busy", "wait_fraction")} if prof_s else None},
                "plans": {"columns": int(ts_["columns"].sum()), "tiles": int(ts_["tiles"].sum()), "updates_per_factorization": int(ts_["updates"].sum()),
                          "levels_max": int(ts_["levels"].max()), "levels_mean": float(ts_["levels"][bigs].mean()) if bigs.any() else 0.0,
                          "dense_tiles": float((np.ceil(rowss[bigs] / 16.0) * (np.ceil(rowss[bigs] / 16.0) + 1) / 2).sum())},
                "critical_component": {"rows": int(rowss[big_i]), "iterations": int(infos["iterations"][big_i]), "columns": int(ts_["columns"][big_i]),
                                       "levels": int(ts_["levels"][big_i])},
                "cpu_baseline": cpu_s,
                "solver_span_ms": sps_s["ms"],
                "solver_span_what": "solve.cc:615-638 on the GPU, one-shot: device assembly of the batch incl. the elimination-tree plans (host threads) + "
                                    "solve + positions on the host - the span the CPU leg times; ms_per_step is the resident solve only",
                "speedup_vs_cpu_baseline_8_threads": (cpu_s["solver_span_ms"] / sps_s["ms"]) if cpu_s and "solver_span_ms" in cpu_s else None,
                "spin_timeouts": bs.spin_timeouts(),
                "workload": "capsized_sparse: synthetic match graph, 1344 images, %d tracks (mean length 6) matched along ring lattices of degree 4 and chained "
                            "by wrong matches, ratio-test similarities; %d directed edges, %d components, %d of them above 192 rows (max %d rows)"
                            % (args.sparse_tracks, sts["n_edges"], sts["n_components"], int(bigs.sum()), int(rowss.max())),
                "ms_per_step": mss, "edges_per_s": sts["n_edges"] / (mss * 1e-3), "tracks_per_s": sts["n_tracks"] / (mss * 1e-3), "steps": ns,
                "kernel_ms": {kernel_names[i]: round(float(cs[i]), 3) for i in range(9) if es_[i] > 0},
                "kernel_edges": {kernel_names[i]: int(es_[i]) for i in range(9) if es_[i] > 0},
                "mean_iterations_large": float(infos["iterations"][bigs].mean()) if bigs.any() else 0.0,
                "max_iterations_large": int(infos["iterations"][bigs].max()) if bigs.any() else 0,
                "dense_factorization_flops_equivalent": float((rowss[bigs].astype(np.float64) ** 3 / 3.0 * infos["iterations"][bigs]).sum()),
                "batch_creation_ms": t_batch * 1e3, "failed": sts["n_failed"], "no_convergence": sts["n_no_convergence"], "setup_s": t_preps,
                "team_components": bs.team_runs(),
                "team_fallbacks": bs.team_fallbacks(),        # components a single workgroup solved because their team could not form (CUs not resident together): 0 on a free GPU
                "note": "round 3 (block-envelope kernel: ~160 dependent 16-column panels per factorization on one wave): 35 ms; round 4: nested "
                        "dissection + columns by level of the elimination tree, the workgroup's eight waves take independent columns side by side: 9 ms; "
                        "round 5: teams of 2 / 4 / 8 workgroups on one XCD per component (LFR_TREE_TEAM), update entries streamed as their columns finish",
            }
            del bs, ps, gs, mas
        ref_bin = os.environ.get("LFR_REFERENCE_SOLVE")
        if ref_bin and world == 1:
            # a reference-built `solve`, if someone supplies one, on the same graph as a .pb
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import compare_with_reference as cwr
            wd = tempfile.mkdtemp(prefix="lfr_bench_ref_")
            pb = os.path.join(wd, "config4.pb")
            capi.write_matching_file(pb, ma)
            res["reference_solve"] = cwr.compare(ref_bin, pb, wd) if os.path.exists(ref_bin) else {"error": "%s does not exist" % ref_bin}
        print(json.dumps(res))
    dist.barrier()
    dist.shutdown()


if __name__ == "__main__":
    main()

"""N>1 path on CPU: two gloo processes shard one problem (no data-path collective) and
all-reduce the statistics vector, exactly as the RCCL ranks do on the GPUs."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.path.join(%(root)r, "local-feature-refinement_amd"))
    import numpy as np
    import torch, torch.distributed as td
    from lfr_amd import capi, dist, synthetic
    rank, world, _ = dist.init(backend="gloo")
    ma = synthetic.generate(seed=61, n_images=80, n_tracks=1500, eps_out=0.0005)     # same graph on every rank
    p = capi.Problem(capi.Graph.from_arrays(ma))
    comps, edges = p.shard_components(rank, world)
    local = {"n_components": len(comps), "n_edges": int(edges.sum()), "sum_final_cost": 0.25 * (rank + 1)}
    tot = dist.allreduce_stats(local)
    tmax = dist.max_over_ranks(10.0 + rank)
    ids = torch.full((p.stats()["n_solved_components"],), -1, dtype=torch.int64)
    ids[:len(comps)] = torch.from_numpy(comps)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    td.all_gather(gathered, ids)
    if rank == 0:
        allc = np.concatenate([g.numpy()[g.numpy() >= 0] for g in gathered])
        st = p.stats()
        print(json.dumps({"world": world, "sum_components": tot["n_components"], "sum_edges": tot["n_edges"],
                          "cost": tot["sum_final_cost"], "tmax": tmax, "unique": int(len(set(allc.tolist()))),
                          "total": int(len(allc)), "solved": st["n_solved_components"], "solved_edges": st["n_solved_edges"]}))
    dist.barrier()
    dist.shutdown()
""")


def test_two_rank_gloo_sharding(tmp_path, lfr_lib):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert out["sum_components"] == out["solved"] == out["unique"] == out["total"]      # disjoint cover
    assert out["sum_edges"] == out["solved_edges"]
    assert abs(out["cost"] - 0.75) < 1e-12 and out["tmax"] == 11.0


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher environment must become N ranks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, argv
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])

    class A:
        gpus = 4
    try:
        bench.self_launch(A())
    except SystemExit:
        pass
    argv = seen["argv"]
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in argv and argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-4:] == ["--gpus", "4", "--steps", "3"] and argv[-5].endswith("bench.py")


def test_snake_deal_balances_every_kernel_class(lfr_lib):
    """Strong scaling shards ONE graph: the snake deal over the batch order (class, edges descending) gives every shard
    the same mix - per class, shard edge totals differ by at most one component."""
    import numpy as np
    from lfr_amd import capi, synthetic
    ma = synthetic.generate(seed=62, n_images=96, n_tracks=900, len_dist="uniform", len_lo=2, len_hi=40, eps_out=0.0005)
    p = capi.Problem(capi.Graph.from_arrays(ma))
    _, all_e = p.shard_components(0, 1)
    for world in (2, 4, 8):
        loads = np.array([p.shard_components(r, world)[1].sum() for r in range(world)], float)
        assert loads.sum() == all_e.sum()
        assert loads.max() - loads.min() <= all_e.max() * 3          # a few classes, each within one component

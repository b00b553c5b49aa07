"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm, "gloo" on CPU for tests).

Components are independent (solve.cc:594-597: each task reads its own edges and writes its own
positions), so the data path needs NO collective: every rank solves its own shard.  The only
exchange is an all-reduce(sum) of a small statistics vector for reporting (the reference has no
global convergence criterion; coupling components through one would change the results).
"""
import os

STAT_KEYS = ["n_components", "n_edges", "n_nodes", "n_tracks", "n_converged", "n_no_convergence", "n_failed",
             "sum_iterations", "ref_jacobian_passes_edges", "ref_cost_passes_edges", "exec_passes_edges",
             "ref_passes_nodes", "sum_final_cost"]


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    rank, world, local = env_rank_world()
    if world > 1:
        import torch
        import torch.distributed as td
        if not td.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                torch.cuda.set_device(local)
            td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        td.barrier()


def _tensor(values):
    import torch
    import torch.distributed as td
    dev = "cuda" if (td.is_initialized() and td.get_backend() == "nccl") else "cpu"   # "nccl" is RCCL on ROCm
    return torch.tensor(values, dtype=torch.float64, device=dev)


def allreduce_stats(stats):
    """Sum the additive entries of a solve-stats dict over all ranks (one small all-reduce)."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return dict(stats)
    t = _tensor([float(stats.get(k, 0)) for k in STAT_KEYS])
    td.all_reduce(t, op=td.ReduceOp.SUM)
    out = dict(stats)
    for k, v in zip(STAT_KEYS, t.tolist()):
        out[k] = v if k == "sum_final_cost" else int(round(v))
    return out


def max_over_ranks(value):
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return float(value)
    t = _tensor([float(value)])
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return float(value)
    t = _tensor([float(value)])
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable records: the bench line's per-rank evidence)."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [obj]
    out = [None] * td.get_world_size()
    td.all_gather_object(out, obj)
    return out


def shutdown():
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        td.destroy_process_group()

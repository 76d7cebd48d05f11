"""Multi-GPU layout of the hot path: replicas only.

One SAC run does not shard (batch 256 against 1.6 MB of weights: an all-reduce per microsecond-scale
step would be pure latency), so the 8 GPUs of a node run independent seeds / tasks, one process per
GPU -- the reference's own scale-out model (/root/reference/launch_jobs.sh:15-24: one job per
config).  There is no data-path collective; the only communication is one all-gather of a small
per-GPU result vector at the end (RCCL over xGMI when the backend is "nccl"; payload O(100 B), so
latency-only)."""
from __future__ import annotations

import os

SWEEP = [("Lift", 42, 7), ("Door", 46, 7), ("Stack", 55, 7), ("Wipe", 379, 6), ("PickPlaceCan", 46, 7),
         ("NutAssemblyRound", 46, 7), ("TwoArmPegInHole", 73, 12), ("TwoArmHandoff", 86, 14)]


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def task_for_rank(rank, sweep=False):
    """(task, obs_dim, act_dim, seed) of a rank: same workload / different seed, or the 8-task sweep."""
    task, O, A = SWEEP[rank % len(SWEEP)] if sweep else SWEEP[0]
    return task, O, A, 17 + rank


def init_process_group(backend, local_rank=0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(dist, value):
    """MAX-reduce of one float (the timed region is as long as the slowest rank's)."""
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(dist, vec):
    """All-gather of a small per-rank result vector -> list (one list of floats per rank)."""
    import torch
    if dist is None:
        return [[float(x) for x in vec]]
    mine = torch.tensor([float(x) for x in vec], dtype=torch.float64, device=_device(dist))
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in o.tolist()] for o in out]


def aggregate_steps_per_second(world, steps_per_rank, elapsed_max):
    """Whole-job throughput: every rank ran `steps_per_rank` steps inside the max-over-ranks time."""
    return world * steps_per_rank / elapsed_max

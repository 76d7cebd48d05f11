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


class NodeBarrier:
    """Barrier of the ranks of ONE node through a page of shared memory (/dev/shm): every rank owns one cache line and
    writes the number of the barrier it has reached; a rank leaves when all lines show that number.  A few microseconds,
    host only -- the timed region of a 20-step run is ~0.6 ms, and a collective on the GPUs just to line the host
    processes up would be a tenth of it (north_star: "RCCL only for a final metric all-gather").  Set-up goes through the
    process group once (rank 0 creates the file, a dist.barrier() publishes it); if anything about it fails, or the job
    spans nodes, the caller keeps using dist.barrier()."""
    LINE = 8                                   # int64 per rank line (64 B)

    def __init__(self, dist, rank, world, timeout_s=900.0):
        import mmap
        import numpy as np
        self.rank, self.world, self.timeout_s, self.epoch = rank, world, timeout_s, 0
        self.path = "/dev/shm/sac_bench_barrier_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getuid())
        nbytes = 8 * self.LINE * world
        # exactly two dist.barrier() calls on every rank, whatever fails in between (a rank that skipped one would leave
        # the others waiting in it); the error, if any, is raised behind them
        err, fd = None, -1
        try:
            if rank == 0:
                fd = os.open(self.path, os.O_CREAT | os.O_TRUNC | os.O_RDWR, 0o600)
                os.ftruncate(fd, nbytes)
        except OSError as e:
            err = e
        dist.barrier()                         # the file exists and is zeroed
        try:
            if err is None:
                if rank != 0:
                    fd = os.open(self.path, os.O_RDWR)
                self._mm = mmap.mmap(fd, nbytes)
                self._a = np.frombuffer(self._mm, dtype=np.int64).reshape(world, self.LINE)
        except (OSError, ValueError) as e:
            err = e
        finally:
            if fd >= 0:
                os.close(fd)
        dist.barrier()                         # everybody has mapped it (or given up)
        if rank == 0:
            try:
                os.unlink(self.path)           # (the mappings keep it alive; nothing is left behind)
            except OSError:
                pass
        if err is not None:
            raise err

    def wait(self):
        import time
        self.epoch += 1
        e, a = self.epoch, self._a
        a[self.rank, 0] = e                    # single writer per line; aligned 8-byte stores are atomic on x86-64
        col = a[:, 0]
        t_end = None
        while True:
            for _ in range(2000):
                if (col >= e).all():
                    return
            if t_end is None:
                t_end = time.monotonic() + self.timeout_s
            elif time.monotonic() > t_end:
                raise RuntimeError("NodeBarrier: rank %d waited %.0f s at barrier %d (lines: %s)"
                                   % (self.rank, self.timeout_s, e, col.tolist()))


def node_barrier(dist):
    """A NodeBarrier for the ranks of this job if they all live on this node, else None."""
    if dist is None:
        return None
    rank, _, world = rank_info()
    same_node = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world
    ok = 1
    nb = None
    try:
        if not same_node or os.environ.get("SAC_BENCH_NODE_BARRIER", "1") == "0":
            raise RuntimeError("not used")
        nb = NodeBarrier(dist, rank, world)
    except Exception:                           # noqa: BLE001 -- any failure: fall back for everybody
        ok = 0
    # all ranks agree (a rank that failed inside __init__ has still passed the same number of dist.barrier() calls only
    # if it failed before the first one or after the last: the all-reduce below is what makes the decision common)
    return nb if min_over_ranks(dist, ok) == 1 else None


def min_over_ranks(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


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

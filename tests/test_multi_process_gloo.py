"""CPU, world_size 2, gloo: the N>1 path of bench.py -- independent replicas, max-over-ranks timing,
one all-gather of per-rank result vectors (the job's only collective; RCCL on the GPU node)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from robosuite_benchmark_amd import parallel
    assert parallel.rank_info() == (rank, rank, world)
    dist = parallel.init_process_group("gloo")
    task = parallel.task_for_rank(rank, sweep=True)
    elapsed = 1.0 + rank                      # the slower rank defines the timed region
    tmax = parallel.max_over_ranks(dist, elapsed)
    res = parallel.gather_results(dist, [1000.0 / elapsed, float(rank), task[1]])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, tmax, res, task[0]))


def test_world_size_two_gloo_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tmax, res, task in out:
        assert tmax == 2.0                                        # MAX over ranks
        assert res == [[1000.0, 0.0, 42.0], [500.0, 1.0, 46.0]]   # identical on both ranks, rank order
    assert [o[3] for o in out] == ["Lift", "Door"]
    from robosuite_benchmark_amd import parallel
    assert parallel.aggregate_steps_per_second(2, 1000, out[0][1]) == 1000.0


def test_single_process_path_needs_no_process_group():
    from robosuite_benchmark_amd import parallel
    assert parallel.max_over_ranks(None, 0.25) == 0.25
    assert parallel.gather_results(None, [1, 2.5]) == [[1.0, 2.5]]


def _barrier_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from robosuite_benchmark_amd import parallel
    dist = parallel.init_process_group("gloo")
    nb = parallel.node_barrier(dist)
    assert nb is not None and not os.path.exists(nb.path)          # shared memory in use, nothing left in /dev/shm
    # nobody leaves barrier k before everybody has reached it: a rank that sleeps in front of it holds the others back
    late = []
    for k in range(200):
        if k % 50 == rank * 10:
            time.sleep(0.02)
        t0 = time.perf_counter()
        nb.wait()
        late.append(time.perf_counter() - t0)
    waited = sum(1 for x in late if x > 0.01)
    t0 = time.perf_counter()
    for _ in range(2000):
        nb.wait()
    per = (time.perf_counter() - t0) / 2000
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, waited, per))


def test_node_barrier_holds_the_ranks_together_and_costs_microseconds():
    """The bracket of bench.py's timed region on N > 1 ranks of one node (parallel.NodeBarrier)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_barrier_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, waited, per in out:
        # each rank was held back by the sleepers of the OTHER ranks: 4 sleeps per rank in 200 barriers
        assert waited >= 4 * (world - 1) - 2, (rank, waited)
        assert per < 2e-3, per                                       # (microseconds on an idle box; generous for CI)

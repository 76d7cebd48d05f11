"""GPU: RCCL itself (torch.distributed backend "nccl" on ROCm) on the one-GPU box.

north_star: "RCCL only for a final metric all-gather".  Two ranks cannot share a GPU under RCCL, so the N > 1 tests of
this repo run on gloo; what a world-size-1 nccl group still exercises is everything that can break independently of the
peer count: librccl loading, `init_process_group(backend="nccl", device_id=...)`, collectives on DEVICE tensors through
`parallel.max_over_ranks / min_over_ranks / gather_results`, a device barrier, `destroy_process_group` -- and the nccl
branch of bench.py's main() (`--force-dist`)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import torch
from robosuite_benchmark_amd import parallel
dist = parallel.init_process_group("nccl", 0)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
assert parallel._device(dist).type == "cuda"
assert parallel.max_over_ranks(dist, 3.25) == 3.25
assert parallel.min_over_ranks(dist, -1.5) == -1.5
assert parallel.gather_results(dist, [1.0, 2.5, -3.0, 4.0]) == [[1.0, 2.5, -3.0, 4.0]]
t = torch.arange(8, dtype=torch.float32, device="cuda:0")
dist.all_reduce(t)
assert t.tolist() == list(range(8))
dist.barrier(device_ids=[0])
nb = parallel.node_barrier(dist)              # the shared-memory barrier's set-up runs over the nccl group's barrier
assert nb is not None
nb.wait(); nb.wait()
dist.destroy_process_group()
print("RCCL_SMOKE_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_world_size_one_nccl_group_runs_the_result_collectives_on_device_tensors():
    # (a child process: the process group's life cycle stays out of the test runner)
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SMOKE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_nccl_branch_runs_with_force_dist():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
                        "--steps", "20", "--warmup", "5", "--buffer", "100000", "--no-cpu-baseline", "--no-stepwise",
                        "--no-peaks", "--profile-steps", "20"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["per_gpu"]) == 1
    assert "shared memory" in line["ranks_barrier"] or "nccl" in line["ranks_barrier"]

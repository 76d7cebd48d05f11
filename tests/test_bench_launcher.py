"""CPU: `python bench.py --gpus N` from a plain interpreter starts its N ranks itself (BASELINE.json configs[4] is an
8-GPU job; the reference's scale-out is one independent job per config, /root/reference/launch_jobs.sh:15-24).
--dry-run replaces the device work by a sleep, everything else -- child processes, gloo rendezvous on 127.0.0.1,
barriers, max-over-ranks timing, the all-gather of per-rank results, rank -> task mapping, one JSON line -- is the
real code path of an N-GPU run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_plain_python_launches_its_own_ranks_and_prints_one_line():
    steps, ms = 20, 4.0
    p = run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--sweep", "--steps", str(steps), "--warmup", "2",
                  "--dry-run-ms-per-step", str(ms))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["warmup"] == 2 and d["scaling"] == "weak" and d["dry_run"] is True
    assert d["metric"].startswith("SAC grad-steps/sec") and d["unit"] == "grad-steps/s" and d["vs_baseline"] is None
    # rank -> task mapping of the sweep (parallel.SWEEP) and independent seeds
    assert [(t["rank"], t["task"], t["obs_dim"], t["act_dim"], t["seed"]) for t in d["config"]["rank_tasks"]] == [
        (0, "Lift", 42, 7, 17), (1, "Door", 46, 7, 18)]
    # every rank's result vector arrived through the all-gather; rank 1 "trains" at half the speed
    assert len(d["per_gpu"]) == 2 and d["per_gpu"][0][1] == 0.5 and d["per_gpu"][1][1] == 1.5
    # (the timed region ends in a barrier: the fast rank waits for the slow one, both clocks read the slow rank's time)
    assert all(abs(x[0] - steps / d["elapsed_max_s"]) <= 0.1 * x[0] for x in d["per_gpu"])
    # timing = MAX over ranks: at least the slow rank's sleep; value = whole-job steps / that time
    assert d["elapsed_max_s"] >= steps * ms * 2 * 1e-3
    assert abs(d["value"] - 2 * steps / d["elapsed_max_s"]) <= 0.01 * d["value"]
    assert abs(d["ms_per_step"] - d["elapsed_max_s"] / steps * 1e3) < 1e-3
    # the ranks of one node line up through shared memory (parallel.NodeBarrier)
    assert d["ranks_barrier"].startswith("shared memory")


def test_the_process_groups_barrier_is_the_fall_back():
    p = run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "5", "--warmup", "1",
                  env_extra=dict(SAC_BENCH_NODE_BARRIER="0"))
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 2 and d["ranks_barrier"] == "torch.distributed barrier (gloo)"


def test_a_failing_rank_makes_the_launcher_fail():
    p = run_bench("--gpus", "2", "--backend", "no-such-backend", "--dry-run", "--steps", "2", "--warmup", "0")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_a_rank_other_than_zero_dying_ends_the_job_at_once_and_is_named():
    """the launcher waited on rank 0 first -- which sat in the rendezvous for its 15-30 minute timeout
    when another rank had died.  All children are polled now: the job ends within seconds, the exit line names the rank
    that failed first and that rank's stderr is shown."""
    import time
    t0 = time.time()
    p = run_bench("--gpus", "3", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0", "--dry-run-fail-rank", "2",
                  timeout=120)
    assert p.returncode != 0 and time.time() - t0 < 90
    assert "rank 2 failed first" in p.stderr and "exits on purpose" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_torchrun_style_environment_is_still_honoured():
    """Under torch.distributed.run the ranks already exist (RANK / WORLD_SIZE in the environment): no self-launch."""
    p = run_bench("--gpus", "1", "--dry-run", "--steps", "3", "--warmup", "1",
                  env_extra=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip())
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "single GPU"
    p = run_bench("--gpus", "2", "--dry-run", "--steps", "3", env_extra=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert p.returncode != 0 and "does not match" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_two_ranks_share_the_one_gpu_of_this_box():
    """The N = 2 path end to end on real kernels: two self-launched ranks, both on device 0, gloo for the collective."""
    p = run_bench("--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "60", "--warmup", "10", "--buffer",
                  "50000", timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and len(d["per_gpu"]) == 2 and d["value"] > 0
    assert all(x[1] == x[1] for x in d["per_gpu"])          # finite losses from both ranks

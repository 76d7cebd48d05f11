#!/usr/bin/env python3
"""bench.py -- SAC gradient-steps/second of the MI355X-native hot loop.

A "step" = one pass of the hot path over one minibatch: random_batch index draw + row gather +
one full SAC gradient step (rlkit_custom.py:234-238 of the reference), with the replay buffer
already resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N = 1: BASELINE.json configs[1] (Lift-Panda-OSC-POSE: obs 42 / act 7, batch 256, 1e6-slot buffer, full).
N > 1: one independent replica per GPU (independent seed of the same workload; `--sweep` runs the
8-task sweep of configs[4] instead), no data-path collective, one RCCL all-gather of the per-GPU
results at the end ("weak" scaling).  Rank 0 prints ONE JSON line.

Launching N > 1: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as plain `python bench.py --gpus N`: the parent then
starts N child ranks itself (one per GPU, before it has touched any GPU API), relays rank 0's JSON line and
exits non-zero if any rank failed -- the reference's own scale-out is "one job per config"
(/root/reference/launch_jobs.sh:15-24).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from robosuite_benchmark_amd import parallel  # noqa: E402

H = 256
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)" (spec)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
EXTRA_TASKS = [("TwoArmLift", 89, 14), ("LiftModded", 64, 4)]


def flops_per_kernel(B, O, A):
    """Algorithmic FLOPs per launch of each step kernel (SURVEY.md section 8d formulas, split by kernel)."""
    P = O * H + H * H + 2 * H * A              # policy MACs / sample
    Q = (O + A) * H + H * H + H                # Q MACs / sample
    return {
        "k_fwd_a": 2 * B * (2 * P + 2 * Q),                              # pi(s), pi(s'), Q1/Q2(s,a)
        "k_fwd_b": 2 * B * (4 * Q + 2 * (H + H * H + A * H)),            # Q1/Q2(s,a_new), T1/T2(s',a') + actor dX (2 nets)
        "k_bwd": 2 * B * (2 * (H + H * H) + 2 * A * H + H * H),          # critic dX (2 nets) | policy head^T, fc1^T
        "k_dw_adam": 2 * B * (2 * Q + P),                                # dW of two critics + policy
    }


def workload_tag(task, B):
    return f"{task.lower()}_b{B}"


def pmc_traffic(kernel, tag):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary of THIS workload
    (profiles/*<tag>*pmc_traffic.json; collected offline by scripts/profile_round.sh: PMC passes cannot run inside
    this process).  Older summaries without a workload tag are the Lift batch-256 ones."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", "lift_b256") != tag:
            continue
        k = d.get("kernels", {}).get(kernel)
        if k:
            return k.get("traffic_bytes_per_launch"), os.path.relpath(f, ROOT)
    return None, None


def gather_bytes_per_step(B, O, A):
    return 2 * B * (2 * O + A + 2) * 4 + 8 * B


def fill_buffer(buf, n, O, A, seed):
    rs = np.random.RandomState(seed)
    chunk = 125_000
    done = 0
    while done < n:
        m = min(chunk, n - done)
        obs = rs.normal(0, 0.5, (m, O)).astype(np.float32)
        nobs = rs.normal(0, 0.5, (m, O)).astype(np.float32)
        act = rs.uniform(-1, 1, (m, A)).astype(np.float32)
        rew = rs.uniform(0, 1, m).astype(np.float32)
        buf.add_block(obs, act, rew, nobs, np.zeros(m, np.uint8))       # terminals always 0 (ignore_done)
        done += m
    buf.ingest_wait()                                                   # (inserts are asynchronous)


B_INIT = 0.1


def build_replica(task, O, A, B, n_buf, seed, device):
    from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, SACTrainer, TanhGaussianPolicy
    rs = np.random.RandomState(seed)
    # (initial hidden bias stated explicitly -- rlkit's Mlp default at the commit the reference pins; "parity unpinned",
    #  DESIGN.md section 2 -- so that the final losses recorded per round stay comparable whatever the holders default to)
    pol = TanhGaussianPolicy([H, H], O, A, rs=rs, b_init_value=B_INIT)
    qs = [FlattenMlp([H, H], 1, O + A, rs=rs, b_init_value=B_INIT) for _ in range(4)]
    # trainer_kwargs of every shipped variant.json (RUN17/variant.json:52-58)
    trainer = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], discount=0.99,
                         reward_scale=1.0, policy_lr=1e-3, qf_lr=5e-4, soft_target_tau=0.005,
                         target_update_period=5, use_automatic_entropy_tuning=True, batch_size=B,
                         noise_seed=seed, device=device)
    buf = EnvReplayBuffer(n_buf, obs_dim=O, action_dim=A, device=device)
    fill_buffer(buf, n_buf, O, A, 1234 + seed)
    buf.seed(seed)
    return trainer, buf


def measure_peaks(device):
    """the roofline denominators measured on this box (also the clock warm-up of the run)."""
    import ctypes as C
    from robosuite_benchmark_amd import _lib
    out = (C.c_float * 4)()
    _lib.check(_lib.load().sac_measure_peaks(int(device), out), "sac_measure_peaks")
    return dict(hbm_copy_gbs=round(float(out[0]), 1), fp32_mfma_tflops=round(float(out[1]), 2),
                how="float4 stream copy of 1 GiB (read + write; 16-KB tiles per workgroup, best of 6 passes over three grid "
                    "sizes); back-to-back v_mfma_f32_16x16x4_f32 on 8 independent accumulators, one and four waves per "
                    "SIMD (best of 4 passes)")


def cpu_baseline(O, A, B, n_host=1_000_000, budget_s=45.0):
    """The oracle (rlkit-equivalent torch restatement + the reference-shaped float64 host buffer of the full
    capacity) timed on this box's host cores, SURVEY.md 8d protocol: 50 warm-up steps, then repeats of 1000 steps,
    median -- five repeats per thread setting unless the time budget cuts them (what was cut is stated)."""
    import torch
    from oracle.sac_step_torch import HostReplayBuffer, RlkitEquivalentSAC, init_sac_params, np_to_f32_batch
    rs = np.random.RandomState(0)
    hb = HostReplayBuffer(n_host, O, A)
    chunk = 250_000
    for i in range(0, n_host, chunk):
        m = min(chunk, n_host - i)
        hb.fill_block(rs.normal(0, 0.5, (m, O)), rs.uniform(-1, 1, (m, A)), rs.uniform(0, 1, (m, 1)),
                      np.zeros((m, 1), np.uint8), rs.normal(0, 0.5, (m, O)))
    sac = RlkitEquivalentSAC(init_sac_params(O, A, seed=0), A, policy_lr=1e-3, qf_lr=5e-4, soft_target_tau=0.005,
                             target_update_period=5)
    np.random.seed(17)
    torch.manual_seed(17)
    default_threads = int(torch.get_num_threads())
    share = max(1, min(16, os.cpu_count() or 1))      # a one-GPU box's CPU share is 16 cores

    def one():
        b, _ = hb.random_batch(B)
        b = np_to_f32_batch(b)
        e1, e2 = torch.randn(B, A).numpy(), torch.randn(B, A).numpy()
        sac.step(b["observations"], b["actions"], b["rewards"], b["terminals"], b["next_observations"], e1, e2)

    def timed(threads, budget):
        torch.set_num_threads(threads)
        for _ in range(50):
            one()
        rates, t_start = [], time.perf_counter()
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(1000):
                one()
            rates.append(1000 / (time.perf_counter() - t0))
            # stop when the next repeat would not fit the budget any more
            if rep < 4 and (time.perf_counter() - t_start) * (rep + 2) / (rep + 1) > budget:
                break
        return float(np.median(rates)), len(rates)

    # eager torch on ~60 small ops per step does not scale with threads: 1 thread and the box's CPU share are both
    # timed, the faster one is reported (torch's default of one thread per host CPU is far slower)
    # (the single-thread setting gets the smaller share of the budget: it is the slower one on every box seen so far)
    r1, n1 = timed(1, budget_s * 0.2)
    rs_, ns = timed(share, budget_s * 0.8)
    torch.set_num_threads(default_threads)
    best, cores, n = (r1, 1, n1) if r1 >= rs_ else (rs_, share, ns)
    cut = "" if n == 5 else f" (time budget cut the 5 repeats to {n})"
    return dict(value=round(best, 2), unit="grad-steps/s", cores=cores, kind="port",
                sample=f"median of {n} x 1000 steps of the same workload (batch {B}, obs {O}, act {A}) after 50 warm-up"
                       f"{cut}; reference-shaped float64 host buffer of {n_host} rows, np.random.randint + fancy-index "
                       f"gather + float32 cast + torch {torch.__version__} CPU eager step; "
                       f"1 thread: {r1:.1f}/s ({n1} repeats), {share} threads: {rs_:.1f}/s ({ns} repeats)",
                host_cpus=os.cpu_count())


def build_td3_replica(O, A, B, n_buf, seed, device=0):
    from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, TanhMlpPolicy, TD3Trainer
    rs = np.random.RandomState(seed)
    pols = [TanhMlpPolicy([H, H], A, O, rs=rs, b_init_value=B_INIT) for _ in range(2)]
    qs = [FlattenMlp([H, H], 1, O + A, rs=rs, b_init_value=B_INIT) for _ in range(4)]
    tr = TD3Trainer(policy=pols[0], qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], target_policy=pols[1],
                    target_policy_noise=0.2, discount=0.99, reward_scale=1.0, policy_learning_rate=1e-3,
                    qf_learning_rate=5e-4, policy_and_target_update_period=2, tau=0.005, batch_size=B,
                    noise_seed=seed, device=device)
    buf = EnvReplayBuffer(n_buf, obs_dim=O, action_dim=A, device=device)
    fill_buffer(buf, n_buf, O, A, 1234 + seed)
    buf.seed(seed)
    return tr, buf


def cpu_baseline_td3(O, A, B, budget=12.0):
    import torch
    from oracle.sac_step_torch import HostReplayBuffer, np_to_f32_batch
    from oracle.td3_step_torch import RlkitEquivalentTD3, init_td3_params
    rs = np.random.RandomState(0)
    hb = HostReplayBuffer(100_000, O, A)
    hb.fill_block(rs.normal(0, 0.5, (100_000, O)), rs.uniform(-1, 1, (100_000, A)), rs.uniform(0, 1, (100_000, 1)),
                  np.zeros((100_000, 1), np.uint8), rs.normal(0, 0.5, (100_000, O)))
    td3 = RlkitEquivalentTD3(init_td3_params(O, A, seed=0), A, qf_learning_rate=5e-4)
    np.random.seed(17)
    best = None
    for threads in (1, max(1, min(16, os.cpu_count() or 1))):
        torch.set_num_threads(threads)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget / 2 and n < 2000:
            b, _ = hb.random_batch(B)
            b = np_to_f32_batch(b)
            td3.step(b["observations"], b["actions"], b["rewards"], b["terminals"], b["next_observations"],
                     torch.randn(B, A).numpy())
            n += 1
        r = n / (time.perf_counter() - t0)
        if best is None or r > best[0]:
            best = (r, threads, n)
    return dict(value=round(best[0], 2), unit="grad-steps/s", cores=best[1], kind="port",
                sample=f"{best[2]} TD3 steps (batch {B}), float64 host buffer of 100000 rows, torch CPU eager")


def td3_main(args):
    """`bench.py --agent TD3`: the TD3 row of SURVEY.md 8f on the headline workload shape (N = 1 only; own JSON line)."""
    O, A, B = 42, 7, args.batch
    tr, buf = build_td3_replica(O, A, B, args.buffer, 17)
    tr.train_loop(buf, max(args.warmup, 1), batch_size=B)
    tr._lib.sac_sync(tr._h)
    t0 = time.perf_counter()
    first, last = tr.train_loop(buf, args.steps, batch_size=B)
    tr._lib.sac_sync(tr._h)
    el = time.perf_counter() - t0
    for _ in range(50):
        tr.train(buf.random_batch(B))
    tr._lib.sac_sync(tr._h)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        tr.train(buf.random_batch(B))
    tr._lib.sac_sync(tr._h)
    el2 = time.perf_counter() - t1
    P, Q = O * H + H * H + H * A, (O + A) * H + H * H + H
    # algorithmic FLOPs per step: critic pass every step, actor pass every 2nd (SURVEY.md 8d formulas adapted)
    critic = 2 * B * (P + 4 * Q) + 2 * 2 * B * (Q + H * H + H)
    actor = 2 * B * (P + Q) + 2 * B * (H + H * H + A * H) + 2 * B * (P + H * H + H * A)
    gflop = (critic + 0.5 * actor) / 1e9
    out = dict(metric=f"TD3 grad-steps/sec (batch={B}, {args.buffer} buffer), 1 GPU", value=round(args.steps / el, 2),
               unit="grad-steps/s", n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=round(el / args.steps * 1e3, 5),
               higher_is_better=True, vs_baseline=None, dtype="f32", data="synthetic (as the SAC line)",
               config=dict(workload=f"Lift-Panda TD3 inner loop: obs {O} / act {A}, batch {B}, {args.buffer}-slot HBM replay "
                                    "buffer, hidden 256x256, policy_and_target_update_period 2, tau .005, noise .2 / clip .5"),
               launches_per_step=("2 (critic pass: k_abc<M_TD3_CRITIC> + k_dw_adam)" if tr.is_fused() else "4 (critic pass)")
               + " + 3 on policy steps",
               roofline=dict(bound="mfma", unit="TFLOP/s", peak=PEAK_FP32_MFMA_TFLOPS, gflop_per_step=round(gflop, 4),
                             achieved=round(gflop * args.steps / el / 1e3, 3),
                             frac=round(gflop * args.steps / el / 1e3 / PEAK_FP32_MFMA_TFLOPS, 5), traffic=None),
               stepwise_interface=dict(value=round(args.steps / el2, 2), unit="grad-steps/s"),
               final={"QF1 Loss": float(last[0]), "Policy Loss": float(last[2])},
               cpu_baseline=None if args.no_cpu_baseline else cpu_baseline_td3(O, A, B))
    print(json.dumps(out), flush=True)


def concurrent_replicas(task, O, A, B, n_replicas, steps, device, by_xcd=False, fused=False, split=False):
    """Extra data point, never `value`: R independent runs (own buffer, nets, streams) driven from R host threads
    on ONE GPU -- the reference's real workload is 5 seeds x 29 configurations of independent jobs
    (/root/reference/launch_jobs.sh:15-24), and a single batch-256 run leaves most of the chip idle."""
    import threading
    # co-tenant runs: the fused step needs the whole chip to itself, so the replicas take the four-launch step -- unless
    # `fused`: then the library serialises their fused launches, and one run's weight-gradient launch (small blocks) can
    # share the chip with the next run's fused launch
    if not fused and not split:
        os.environ["SAC_FUSED"] = "0"
    reps = [build_replica(task, O, A, B, 100_000, 100 + i, device) for i in range(n_replicas)]
    os.environ.pop("SAC_FUSED", None)
    if by_xcd:          # experiment: replica i confined to the 32 CUs of XCD i % 8 (CU-masked streams)
        from robosuite_benchmark_amd import _lib
        for i, (tr, buf) in enumerate(reps):
            _lib.check(tr._lib.sac_trainer_set_xcd(tr._h, i % 8), "sac_trainer_set_xcd")
            _lib.check(tr._lib.sac_buffer_set_xcd(buf._h, i % 8), "sac_buffer_set_xcd")
    if split:           # experiment: the chip dealt out in equal shares of XCDs; a run whose fused step fits its share keeps it
        from robosuite_benchmark_amd import _lib
        per = max(1, 8 // n_replicas)
        for i, (tr, buf) in enumerate(reps):
            mask = ((1 << per) - 1) << (per * (i % (8 // per)))
            _lib.check(tr._lib.sac_trainer_set_xcd_mask(tr._h, mask), "sac_trainer_set_xcd_mask")
            _lib.check(tr._lib.sac_buffer_set_xcd_mask(buf._h, mask), "sac_buffer_set_xcd_mask")
    for tr, buf in reps:
        tr.train_loop(buf, 100, batch_size=B)
    t0 = time.perf_counter()
    th = [threading.Thread(target=tr.train_loop, args=(buf, steps), kwargs=dict(batch_size=B)) for tr, buf in reps]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    return dict(replicas=n_replicas, steps_each=steps, value=round(n_replicas * steps / el, 2), unit="grad-steps/s",
                placement=("%d XCDs per replica, CU-masked streams" % max(1, 8 // n_replicas)) if split else
                "one XCD (32 CUs) per replica, CU-masked streams" if by_xcd else "every launch spans the chip",
                step=("fused on its own XCDs" if all(tr.is_fused() for tr, _ in reps) else "four launches") if split else
                "fused (launches serialised across runs)" if fused and not by_xcd else "four launches",
                note="aggregate of independent runs sharing one GPU (100000-slot buffers); not the headline metric")


# ---------------------------------------------------------------------------------------------------
# N > 1 from plain `python bench.py --gpus N`: start the ranks ourselves
# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank job.  Nothing here touches a GPU API (no torch.cuda, no HIP): the children
    are started first and this process only waits, so no process that has initialised the GPU is ever re-executed.
    Rank 0's stdout carries the JSON line; every rank's stderr is kept (a temporary file per rank) and shown when the
    job fails.  ALL children are polled: the moment any rank exits non-zero the others -- which would otherwise sit in a
    rendezvous or barrier until its timeout, 15-30 minutes -- are ended (exactly the PIDs started here) and the parent
    exits non-zero naming the culprit."""
    import tempfile
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(_free_port())
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, errs, outs = [], [], []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        errs.append(tempfile.TemporaryFile(mode="w+", prefix=f"bench_rank{r}_err_"))
        outs.append(tempfile.TemporaryFile(mode="w+", prefix=f"bench_rank{r}_out_"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=outs[r], stderr=errs[r], text=True))
    rcs, failed = [None] * n, None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            time.sleep(0.5)                       # (ranks that die of the same cause get to say so themselves)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                rcs[r] = p.wait()
            break
        time.sleep(0.02)

    def text_of(f):
        f.seek(0)
        return f.read()
    out0 = text_of(outs[0])
    for r in range(1, n):                         # the other ranks' stdout is not the result: pass it on as diagnostics
        t = text_of(outs[r])
        if t:
            sys.stderr.write(t)
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed} failed first; rank exit codes {rcs}\n")
        for r in range(n):
            t = text_of(errs[r])
            if t.strip():
                sys.stderr.write(f"---- rank {r} stderr (last lines) ----\n" + "\n".join(t.splitlines()[-25:]) + "\n")
        if out0:
            sys.stderr.write(out0)
        raise SystemExit(1)
    for r in range(n):
        t = text_of(errs[r])
        if t:
            sys.stderr.write(t)
    lines = [l for l in (out0 or "").splitlines() if l.strip().startswith("{")]
    if len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n{out0}\n")
        raise SystemExit(1)
    print(lines[0], flush=True)


class _DryTrainer:
    """--dry-run: the launcher / rendezvous / timing / gathering plumbing of an N-rank job without any GPU.  A rank
    "trains" by sleeping (rank r takes (r + 1) x the per-step time, so the max-over-ranks rule is visible)."""

    def __init__(self, rank, ms_per_step):
        self.rank, self.ms = rank, ms_per_step

    def train_loop(self, buf, n, batch_size=None):
        time.sleep(n * self.ms * (self.rank + 1) * 1e-3)
        d = np.zeros(32, np.float32)
        d[0], d[1], d[28] = 0.5 + self.rank, 0.25 + self.rank, 0.125
        return d, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1_000_000)
    ap.add_argument("--sweep", action="store_true", help="N>1: 8-task sweep (BASELINE configs[4]) instead of seeds")
    ap.add_argument("--task", type=str, default=None,
                    help="N=1 only: another task of the sweep table (e.g. Door, TwoArmHandoff); default Lift")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stepwise", action="store_true", help="skip the extra stepwise-interface data point")
    ap.add_argument("--no-peaks", action="store_true", help="skip the measured-peak microbenchmarks (spec peaks only)")
    ap.add_argument("--backend", type=str, default="nccl", help="collective backend for N>1 (nccl == RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank computes on device 0 (use with --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU at all: ranks sleep instead of training (tests of the launcher and the N>1 plumbing; "
                         "use with --backend gloo)")
    ap.add_argument("--dry-run-ms-per-step", type=float, default=1.0)
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1,
                    help="--dry-run only (launcher tests): this rank exits with code 3 before the rendezvous")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1: still initialise the process group (world size 1) so that the N>1 branch -- RCCL set-up, "
                         "barriers, max-over-ranks, the result all-gather -- runs on a one-GPU box")
    ap.add_argument("--profile-steps", type=int, default=500)
    ap.add_argument("--agent", type=str, default="SAC", choices=["SAC", "TD3"],
                    help="TD3: the SURVEY 8f row on the same workload shape (N=1, its own JSON line); default SAC = the headline metric")
    ap.add_argument("--xcd-replicas", action="store_true", help="with --replicas-per-gpu: confine replica i to XCD i % 8")
    ap.add_argument("--fused-replicas", action="store_true", help="with --replicas-per-gpu: the replicas keep the fused step")
    ap.add_argument("--split-replicas", action="store_true",
                    help="with --replicas-per-gpu R: replica i owns 8/R XCDs (CU-masked streams) and keeps the fused step if it fits them")
    ap.add_argument("--replicas-per-gpu", type=int, default=0,
                    help="N=1 only: also time R concurrent independent runs on the GPU (reported beside, never as, value)")
    args = ap.parse_args()

    if args.agent == "TD3":
        if args.gpus != 1:
            raise SystemExit("--agent TD3 is a single-GPU line")
        return td3_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    rank, local_rank, world = parallel.rank_info()
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    if args.dry_run and rank == args.dry_run_fail_rank:
        sys.stderr.write("bench.py: rank %d exits on purpose (--dry-run-fail-rank)\n" % rank)
        raise SystemExit(3)
    dist = None
    if world > 1 or args.force_dist:
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist = parallel.init_process_group(args.backend, local_rank)   # nccl == RCCL on ROCm
    if args.single_device:
        local_rank = 0
        # ranks sharing one GPU: the fused step's workgroups hand data to each other inside a launch and need the
        # whole chip to themselves, so co-tenant ranks take the four-launch step
        os.environ.setdefault("SAC_FUSED", "0")

    task, O, A, seed = parallel.task_for_rank(rank, sweep=args.sweep and world > 1)
    if args.task is not None and world == 1:
        task, O, A = next(t for t in parallel.SWEEP + EXTRA_TASKS if t[0] == args.task)
    B = args.batch
    peaks = None
    if args.dry_run:
        trainer, buf = _DryTrainer(rank, args.dry_run_ms_per_step), None
        sync = lambda: None                                            # noqa: E731
    else:
        import torch
        trainer, buf = build_replica(task, O, A, B, args.buffer, seed=seed, device=local_rank)
        if world == 1 and not args.no_peaks:
            peaks = measure_peaks(local_rank)                          # (before the timed region: also ramps the clocks)

        def sync():
            torch.cuda.synchronize()
            trainer._lib.sac_sync(trainer._h)

    # the ranks of one node line up through shared memory (parallel.NodeBarrier: microseconds, no GPU work inside the timed
    # region's bracket); the process group's own barrier is the fall-back
    nb = parallel.node_barrier(dist)
    ranks_barrier = (nb.wait if nb is not None else dist.barrier) if dist is not None else (lambda: None)

    def barrier():
        ranks_barrier()
        sync()

    # ---- warm-up (untimed), then EXACTLY K timed steps --------------------------------------
    if args.warmup > 0:
        # W untimed steps: all but the last four in one call, the last four as single-step calls (a call's host path --
        # draw, gather, step launches, wait -- is then warm when the timed call starts)
        singles = min(4, args.warmup - 1) if not args.dry_run else 0
        trainer.train_loop(buf, args.warmup - singles, batch_size=B)
        for _ in range(singles):
            trainer.train_loop(buf, 1, batch_size=B)
    for _ in range(3):          # (the bracket itself is warm too: its second call of a process costs 20-35 us more than its tenth)
        barrier()
    t0 = time.perf_counter()
    first, last = trainer.train_loop(buf, args.steps, batch_size=B)      # returns after the stream drained
    t_call = time.perf_counter() - t0
    # closing bracket: the ranks' barrier + torch.cuda.synchronize() (device-wide: it covers the library's streams; the
    # library's own wait has already run inside train_loop, which also checked the fused step's abort word)
    ranks_barrier()
    if not args.dry_run:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    elapsed_max = parallel.max_over_ranks(dist, elapsed)
    # the only collective of the job: per-GPU result vectors (steps/s, last losses, alpha) all-gathered
    per_gpu = parallel.gather_results(dist, [args.steps / elapsed, float(last[0]), float(last[1]), float(last[28])])

    out = None
    if rank == 0:
        value = parallel.aggregate_steps_per_second(world, args.steps, elapsed_max)
        tasks = [parallel.task_for_rank(r, sweep=args.sweep and world > 1) for r in range(world)]
        out = {
            "metric": "SAC grad-steps/sec (batch=256, 1e6 buffer) at 1/2/4/8 GPU" if B == 256 else
                      f"SAC grad-steps/sec (batch={B})",
            "value": round(value, 2), "unit": "grad-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (SURVEY.md 8d: obs~N(0,.5^2), act~U(-1,1), rew~U(0,1), terminals 0; "
                    "rlkit-style random init)",
            "config": {"workload": f"{task}-Panda-OSC-POSE SAC inner loop: obs {O} / act {A}, batch {B}, "
                                   f"{args.buffer}-slot HBM replay buffer (full), hidden 256x256, "
                                   "variant.json trainer_kwargs (lr 1e-3/5e-4, tau .005, period 5)",
                       "parallelism": ("independent task per GPU (8-task sweep)" if args.sweep and world > 1
                                       else "independent seed per GPU, no data-path collective") if world > 1
                       else "single GPU",
                       "rank_tasks": [dict(rank=r, task=t[0], obs_dim=t[1], act_dim=t[2], seed=t[3])
                                      for r, t in enumerate(tasks)],
                       "per_step": "MT19937 index draw + row gather + full SAC gradient step"},
            "elapsed_max_s": round(elapsed_max, 6),
            "timed_region_us": {"train_loop_call": round(t_call * 1e6, 1), "closing_barrier": round((elapsed - t_call) * 1e6, 1)},
            "ranks_barrier": "none (one rank)" if dist is None else ("shared memory (parallel.NodeBarrier)" if nb is not None
                                                                     else f"torch.distributed barrier ({args.backend})"),
            "per_gpu": per_gpu,
            "final": {"QF1 Loss": float(last[0]), "QF2 Loss": float(last[1]), "Alpha": float(last[28])},
        }
        if args.dry_run:
            out["data"] = "none (--dry-run: ranks sleep instead of training; plumbing test only, not a measurement)"
            out["dry_run"] = True
        else:
            out.update(device_report(args, trainer, buf, task, O, A, B, world, value, elapsed_max, peaks, local_rank))
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def device_report(args, trainer, buf, task, O, A, B, world, value, elapsed_max, peaks, device):
    """Everything of the JSON line that comes from the device after the timed region (rank 0)."""
    out = {"device_ms": {k: round(v, 3) for k, v in trainer.loop_timing_ms().items()}}
    # ---- per-kernel durations: instrumented replay of the same loop (HIP events on the
    #      launching streams) ------------------------------------------------------------------
    nprof = min(args.profile_steps, args.steps, 4096)
    prof = trainer.profile_loop(buf, nprof, batch_size=B)
    fl_all = flops_per_kernel(B, O, A)
    step_kernels = [k for k in prof if k.startswith("k_") and k not in ("k_gather", "k_mt_randint") and prof[k] > 0]
    # a fused launch carries the FLOPs of the launches it replaces ("k_fwd_abc" = a + b + c ...)
    fused_parts = {"k_fwd_abc": ("k_fwd_a", "k_fwd_b", "k_bwd"), "k_chain": ("k_fwd_a", "k_fwd_b"),
                   "k_chain_bwd": ("k_fwd_a", "k_fwd_b", "k_bwd")}
    fl = {k: (sum(fl_all[p] for p in fused_parts[k]) if k in fused_parts else fl_all[k]) for k in step_kernels}
    kern = {}
    for k, f in fl.items():
        ms = prof[k]
        kern[k] = dict(ms=round(ms, 5), gflop=round(f / 1e9, 5), tflops=round(f / (ms * 1e-3) / 1e12, 3))
    gb = gather_bytes_per_step(B, O, A) * nprof
    kern["k_gather"] = dict(ms=round(prof["k_gather"], 5), bytes=gb, steps_per_launch=nprof,
                            gbs=round(gb / (prof["k_gather"] * 1e-3) / 1e9, 2))
    kern["k_mt_randint"] = dict(ms=round(prof["k_mt_randint"], 5), indices=nprof * B)
    dom = max(fl, key=lambda k: prof[k])
    # A launch's duration as rocprofv3's kernel trace reports it runs from dispatch to completion, i.e. it
    # includes the ~2 us dispatch boundary (back-to-back launches: the trace's durations add up to the step).
    # The event intervals above (minus the cost of an empty event pair) exclude it, so the boundary is added
    # back: (timed step - sum of the intervals) / launches per step.  `achieved` uses that longer duration.
    # The step time is the DEVICE time of the timed loop (events on the trainer's stream), so the per-call fixed
    # cost of a short loop does not leak into a kernel's duration.
    step_ms = out["device_ms"]["steps"] / args.steps
    boundary_ms = max(0.0, (step_ms - sum(prof[k] for k in fl)) / len(fl))
    dom_ms = prof[dom] + boundary_ms
    achieved = fl[dom] / (dom_ms * 1e-3) / 1e12
    # (the kernel's name as rocprofv3's kernel trace prints it)
    symbol = {"k_fwd_abc": "k_abc", "k_chain": "k_chain" if os.environ.get("SAC_CHAIN8") == "0" else "k_chain8",
              "k_chain_bwd": "k_chain8"}.get(dom, dom)
    traffic, traffic_src = pmc_traffic(symbol, workload_tag(task, B))
    pk_m = peaks["fp32_mfma_tflops"] if peaks else None
    whole = sum(fl.values()) * value / world / 1e12
    out["roofline"] = dict(
        bound="mfma", kernel=dom, kernel_symbol=symbol, achieved=round(achieved, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
        frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 5), peak_spec=PEAK_FP32_MFMA_TFLOPS, peak_measured=pk_m,
        frac_measured=round(achieved / pk_m, 5) if pk_m else None, traffic=traffic, traffic_source=traffic_src,
        flops_per_launch=fl[dom], avg_launch_ms=round(dom_ms, 5), exec_ms=round(prof[dom], 5),
        dispatch_boundary_ms=round(boundary_ms, 5), achieved_exec=round(fl[dom] / (prof[dom] * 1e-3) / 1e12, 3),
        event_pair_ms=round(prof["event_pair"], 5), launches_per_step=len(fl),
        whole_step=dict(gflop=round(sum(fl.values()) / 1e9, 4), tflops=round(whole, 3),
                        frac=round(whole / PEAK_FP32_MFMA_TFLOPS, 5),
                        frac_measured=round(whole / pk_m, 5) if pk_m else None))
    # ---- gather against the HBM roof, at K = 1 and K = 1000 steps per launch ------------
    pk_h = peaks["hbm_copy_gbs"] if peaks else None
    per_step = gather_bytes_per_step(B, O, A)

    def gather_line(K, reps):
        ms = float(np.median([buf.sample_gather_device(B, K)[1] for _ in range(reps)]))
        gbs = per_step * K / (ms * 1e-3) / 1e9
        return dict(steps_per_launch=K, launch_ms=round(ms, 5), achieved=round(gbs, 2), frac=round(gbs / PEAK_HBM_GBS, 5),
                    frac_measured=round(gbs / pk_h, 5) if pk_h else None)

    out["roofline_gather"] = dict(bound="hbm", kernel="k_gather", unit="GB/s", peak=PEAK_HBM_GBS, peak_spec=PEAK_HBM_GBS,
                                  peak_measured=pk_h, algorithmic_bytes_per_step=per_step,
                                  K1=gather_line(1, 21), K1000=gather_line(1000, 5),
                                  in_loop=dict(steps_per_launch=nprof, achieved=kern["k_gather"]["gbs"],
                                               frac=round(kern["k_gather"]["gbs"] / PEAK_HBM_GBS, 5)),
                                  traffic=pmc_traffic("k_gather", workload_tag(task, B))[0])
    out["peaks_measured"] = peaks
    out["kernels"] = kern
    # ---- fixed cost of one sac_train_loop call: t(n) = fixed + n * per_step from two loop lengths -------
    if world == 1:
        def loop_s(n):
            ts = []
            for _ in range(5):
                trainer._lib.sac_sync(trainer._h)
                t0 = time.perf_counter()
                trainer.train_loop(buf, n, batch_size=B)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts))
        t_a, t_b = loop_s(20), loop_s(520)
        per = (t_b - t_a) / 500
        out["fixed_call_us"] = round((t_a - 20 * per) * 1e6, 1)
        out["short_loop"] = dict(steps=20, value=round(20 / t_a, 2), unit="grad-steps/s",
                                 note="a 20-step sac_train_loop call timed like the headline (host wall, call to return)")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(O, A, B, n_host=args.buffer)
    elif world == 1:
        out["cpu_baseline"] = None
    if world == 1 and not args.no_stepwise:
        # the reference's unmodified loop body (random_batch -> train) through the Python duck types, batches
        # staying on the device (DeviceBatch): extra data point, never `value`
        # Sampling: the construction default of the drop-in -- the buffer bound to np.random itself (train.py:112), the
        # process-wide stream right after every call, no hand-over.
        n_sw = min(args.steps, 2000)
        np.random.seed(17)
        buf.bind_numpy_global_stream()
        for _ in range(50):
            trainer.train(buf.random_batch(B))
        trainer._lib.sac_sync(trainer._h)
        t0 = time.perf_counter()
        for _ in range(n_sw):
            trainer.train(buf.random_batch(B))
        trainer._lib.sac_sync(trainer._h)
        out["stepwise_interface"] = dict(value=round(n_sw / (time.perf_counter() - t0), 2), unit="grad-steps/s", steps=n_sw,
                                         sampling="np.random global stream (construction default; state words bound: %s)" % buf._bound,
                                         note="replay_buffer.random_batch(B); trainer.train(batch) per step from Python")
        ref = np.random.RandomState(17)
        for _ in range(50 + n_sw):
            ref.randint(0, buf.num_steps_can_sample(), B)
        out["stepwise_interface"]["np_random_is_where_rlkit_would_leave_it"] = bool(
            np.array_equal(np.random.get_state()[1], ref.get_state()[1]) and np.random.get_state()[2] == ref.get_state()[2])
        buf.seed(17)
    if world == 1:
        # asynchronous ingest: one epoch's 2 500 exploration rows; host-blocking time of the
        # insert call vs the time until the rows have landed in HBM
        rs = np.random.RandomState(5)
        blk = (rs.normal(0, 0.5, (2500, O)), rs.uniform(-1, 1, (2500, A)), rs.uniform(0, 1, 2500),
               rs.normal(0, 0.5, (2500, O)), np.zeros(2500, np.uint8))        # float64, the reference's native dtype
        buf.ingest_wait()
        t0 = time.perf_counter()
        buf.add_block(*blk)
        t1 = time.perf_counter()
        buf.ingest_wait()
        t2 = time.perf_counter()
        out["ingest"] = dict(rows=2500, call_returns_us=round((t1 - t0) * 1e6, 1), landed_us=round((t2 - t0) * 1e6, 1),
                             note="sac_buffer_add_f64 of one epoch's exploration steps: pack into pinned staging + enqueue "
                                  "(call returns) vs rows resident in HBM")
    if world == 1 and not args.no_stepwise:
        # hidden_sizes the fused kernels do not carry (arguments.py:98,104): the library's general step on
        # the same buffer -- an extra data point, never `value`
        from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
        gs = {}
        for hs in ([512, 512], [256, 256, 256]):
            rs = np.random.RandomState(3)
            pol = TanhGaussianPolicy(hs, O, A, rs=rs, b_init_value=B_INIT)
            qs = [FlattenMlp(hs, 1, O + A, rs=rs, b_init_value=B_INIT) for _ in range(4)]
            tg = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], policy_lr=1e-3, qf_lr=5e-4,
                            soft_target_tau=0.005, target_update_period=5, batch_size=B, noise_seed=3, device=device)
            tg.train_loop(buf, 50, batch_size=B)
            n_g = min(args.steps, 500)
            t0 = time.perf_counter()
            _, last_g = tg.train_loop(buf, n_g, batch_size=B)
            dt = time.perf_counter() - t0
            # multiply-adds x 2 of one step: policy forward on s and s' + backward + dW (8 B Pp); each Q net forward and
            # backward on (s,a) and (s,a_new) + dW (10 B Pq); each target forward (2 B Pq)
            gflop = B * (8.0 * pol.flat().size + 24.0 * qs[0].flat().size) * 1e-9
            gs["x".join(map(str, hs))] = dict(value=round(n_g / dt, 1), unit="grad-steps/s", us_per_step=round(1e6 * dt / n_g, 2),
                                               step_kind=tg.fused_mode(), finite=bool(np.all(np.isfinite(last_g))),
                                               gflop_per_step=round(gflop, 3), tflops=round(gflop * n_g / dt * 1e-3, 2))
            del tg
        out["general_step"] = dict(shapes=gs, note="hidden_sizes beyond two layers of <= 256 units: one launch per hidden layer and pass "
                                                   "direction (csrc/sac_general.h); steps include the index draw and the gather")
        buf.seed(17)
    if world == 1 and args.replicas_per_gpu > 1:
        out["concurrent_replicas"] = concurrent_replicas(task, O, A, B, args.replicas_per_gpu, args.steps, device,
                                                         by_xcd=args.xcd_replicas, fused=args.fused_replicas, split=args.split_replicas)
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TD3 on the same workload shape as bench.py: Lift obs 42 / act 7, batch 256, 1e6-slot buffer.
One JSON line: fused-loop grad-steps/s, the stepwise interface, and the CPU oracle (torch restatement) beside it.
    python scripts/bench_td3.py [--steps 2000] [--no-cpu-baseline]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

H = 256


def build(O, A, B, n_buf, seed, device=0):
    from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, TanhMlpPolicy, TD3Trainer
    rs = np.random.RandomState(seed)
    pols = [TanhMlpPolicy([H, H], A, O, rs=rs) for _ in range(2)]
    qs = [FlattenMlp([H, H], 1, O + A, rs=rs) for _ in range(4)]
    tr = TD3Trainer(policy=pols[0], qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], target_policy=pols[1],
                    target_policy_noise=0.2, discount=0.99, reward_scale=1.0, policy_learning_rate=1e-3,
                    qf_learning_rate=5e-4, policy_and_target_update_period=2, tau=0.005, batch_size=B,
                    noise_seed=seed, device=device)
    buf = EnvReplayBuffer(n_buf, obs_dim=O, action_dim=A, device=device)
    bench.fill_buffer(buf, n_buf, O, A, 1234 + seed)
    buf.seed(seed)
    return tr, buf


def cpu_baseline(O, A, B, budget=12.0):
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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    O, A, B = 42, 7, a.batch
    tr, buf = build(O, A, B, a.buffer, 17)
    tr.train_loop(buf, 200, batch_size=B)
    tr._lib.sac_sync(tr._h)
    t0 = time.perf_counter()
    first, last = tr.train_loop(buf, a.steps, batch_size=B)
    tr._lib.sac_sync(tr._h)
    el = time.perf_counter() - t0
    for _ in range(50):
        tr.train(buf.random_batch(B))
    tr._lib.sac_sync(tr._h)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        tr.train(buf.random_batch(B))
    tr._lib.sac_sync(tr._h)
    el2 = time.perf_counter() - t1
    P, Q = O * H + H * H + H * A, (O + A) * H + H * H + H
    # algorithmic FLOPs per step: critic pass every step, actor pass every 2nd (SURVEY.md 8d formulas adapted)
    critic = 2 * B * (P + 4 * Q) + 2 * 2 * B * (Q + H * H + H)
    actor = 2 * B * (P + Q) + 2 * B * (H + H * H + A * H) + 2 * B * (P + H * H + H * A)
    gflop = (critic + 0.5 * actor) / 1e9
    out = dict(metric="TD3 grad-steps/sec (batch=256, 1e6 buffer), 1 GPU", value=round(a.steps / el, 2), unit="grad-steps/s",
               ms_per_step=round(el / a.steps * 1e3, 5), steps=a.steps, dtype="f32",
               config=dict(workload=f"Lift-Panda TD3 inner loop: obs {O} / act {A}, batch {B}, {a.buffer}-slot HBM replay buffer, "
                                    "hidden 256x256, policy_and_target_update_period 2, tau .005, noise .2 / clip .5"),
               launches_per_step="4 (critic pass) + 3 on policy steps",
               roofline=dict(bound="mfma", unit="TFLOP/s", peak=bench.PEAK_FP32_MFMA_TFLOPS, gflop_per_step=round(gflop, 4),
                             achieved=round(gflop * a.steps / el / 1e3, 3),
                             frac=round(gflop * a.steps / el / 1e3 / bench.PEAK_FP32_MFMA_TFLOPS, 5)),
               stepwise_interface=dict(value=round(a.steps / el2, 2), unit="grad-steps/s"),
               final={"QF1 Loss": float(last[0]), "Policy Loss": float(last[2])},
               cpu_baseline=None if a.no_cpu_baseline else cpu_baseline(O, A, B))
    print(json.dumps(out))

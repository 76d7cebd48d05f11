"""Step rate of the general step (csrc/sac_general.h) on a few hidden_sizes, next to the fused kernels on [256, 256].
usage: python scripts/general_shapes_rate.py [steps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, SACTrainer, TanhGaussianPolicy  # noqa: E402


def rate(hidden, hidden_q, O, A, B, steps, general=False):
    if general:
        os.environ["SAC_GENERAL"] = "1"
    rs = np.random.RandomState(0)
    pol = TanhGaussianPolicy(list(hidden), O, A, rs=rs)
    qs = [FlattenMlp(list(hidden_q), 1, O + A, rs=rs) for _ in range(4)]
    tr = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=B, noise_seed=1)
    os.environ.pop("SAC_GENERAL", None)
    n = 20000
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(rs.normal(0, .5, (n, O)).astype(np.float32), rs.uniform(-1, 1, (n, A)).astype(np.float32),
                  rs.uniform(0, 1, (n, 1)).astype(np.float32), rs.normal(0, .5, (n, O)).astype(np.float32),
                  np.zeros((n, 1), np.uint8))
    buf.seed(3)
    tr.train_loop(buf, 50, batch_size=B)
    t0 = time.perf_counter()
    tr.train_loop(buf, steps, batch_size=B)
    dt = time.perf_counter() - t0
    return dict(policy_hidden=list(hidden), qf_hidden=list(hidden_q), obs=O, act=A, batch=B, step_kind=tr.fused_mode(),
                steps_per_s=round(steps / dt, 1), us_per_step=round(1e6 * dt / steps, 2))


def rate_td3(hidden, O, A, B, steps, general=False):
    from robosuite_benchmark_amd import TanhMlpPolicy, TD3Trainer
    if general:
        os.environ["SAC_GENERAL"] = "1"
    rs = np.random.RandomState(0)
    pols = [TanhMlpPolicy(list(hidden), A, O, rs=rs) for _ in range(2)]
    qs = [FlattenMlp(list(hidden), 1, O + A, rs=rs) for _ in range(4)]
    tr = TD3Trainer(policy=pols[0], qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], target_policy=pols[1],
                    batch_size=B, noise_seed=1)
    os.environ.pop("SAC_GENERAL", None)
    n = 20000
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(rs.normal(0, .5, (n, O)).astype(np.float32), rs.uniform(-1, 1, (n, A)).astype(np.float32),
                  rs.uniform(0, 1, (n, 1)).astype(np.float32), rs.normal(0, .5, (n, O)).astype(np.float32),
                  np.zeros((n, 1), np.uint8))
    buf.seed(3)
    tr.train_loop(buf, 50, batch_size=B)
    t0 = time.perf_counter()
    tr.train_loop(buf, steps, batch_size=B)
    dt = time.perf_counter() - t0
    return dict(agent="TD3", hidden=list(hidden), obs=O, act=A, batch=B, step_kind=tr.fused_mode(),
                steps_per_s=round(steps / dt, 1), us_per_step=round(1e6 * dt / steps, 2))


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    for hp, hq, B, gen in (((256, 256), (256, 256), 256, False), ((256, 256), (256, 256), 256, True),
                           ((512, 512), (512, 512), 256, False), ((256, 256, 256), (256, 256, 256), 256, False),
                           ((1024, 1024), (1024, 1024), 256, False), ((400, 300), (400, 300), 100, False),
                           ((512, 512), (512, 512), 1024, False)):
        print(json.dumps(rate(hp, hq, 42, 7, B, steps, gen)), flush=True)
    for hs, gen in (((256, 256), False), ((256, 256), True), ((512, 512), False), ((400, 300), False)):
        print(json.dumps(rate_td3(hs, 42, 7, 256, steps, gen)), flush=True)

"""Soak of the general step (csrc/sac_general.h): two trainers with the same seeds over a long loop are bitwise equal (the
split reductions and last-workgroup hand-offs are deterministic whatever the arrival order), stay finite, and loop ==
stepwise over the same batches.   usage: python scratch/soak_general.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, SACTrainer, TanhGaussianPolicy

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
O, A, B = 42, 7, 256


def make(hs, hq):
    rs = np.random.RandomState(1)
    pol = TanhGaussianPolicy(hs, O, A, rs=rs)
    qs = [FlattenMlp(hq, 1, O + A, rs=rs) for _ in range(4)]
    tr = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=B, noise_seed=5,
                    policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, target_update_period=1)
    rs = np.random.RandomState(2)
    n = 50000
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(rs.normal(0, .5, (n, O)).astype(np.float32), rs.uniform(-1, 1, (n, A)).astype(np.float32),
                  rs.uniform(0, 1, (n, 1)).astype(np.float32), rs.normal(0, .5, (n, O)).astype(np.float32), np.zeros((n, 1), np.uint8))
    buf.seed(9)
    return tr, buf


for hs, hq in (([512, 512], [512, 512]), ([300, 200, 100], [400, 300])):
    t0 = time.time()
    (a, ba), (b, bb), (c, bc) = make(hs, hq), make(hs, hq), make(hs, hq)
    la = a.train_loop(ba, steps, batch_size=B)[1]
    lb = b.train_loop(bb, steps, batch_size=B)[1]
    for _ in range(min(steps, 3000)):
        c.train(bc.random_batch(B))
    sa, sb = a.state_dict(), b.state_dict()
    same = all(np.array_equal(sa["params"][k], sb["params"][k]) for k in sa["params"]) and np.array_equal(la, lb)
    if steps <= 3000:
        sc = c.state_dict()
        same = same and all(np.array_equal(sa["params"][k], sc["params"][k]) for k in sa["params"])
    print(f"general step {hs}/{hq}: two runs of {steps} steps bitwise equal (+ stepwise when <= 3000): {same} | finite: "
          f"{bool(np.all(np.isfinite(la)))} | QF1 loss {la[0]:.4f} alpha {la[28]:.4f} | {time.time() - t0:.1f} s", flush=True)
    assert same and np.all(np.isfinite(la))

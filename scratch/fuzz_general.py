"""Randomised sweep of the general step against the oracle (shapes, batch sizes, trainer kwargs drawn at random; SAC and TD3).
usage: python scratch/fuzz_general.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair, make_td3_pair
from tests.test_gpu_sac_step import batch_and_noise, check_diag, TOL
from tests.test_gpu_td3 import batch_and_noise as td3_batch, check_diag as td3_check

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ["SAC_GENERAL"] = "1"          # (shapes that fit the fused kernels go through the general step too)
bad = 0
for c in range(cases):
    depth_p, depth_q = rs.randint(1, 5), rs.randint(1, 5)
    width = lambda: int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(300, 700), 4 * rs.randint(1, 150)]))
    hp, hq = tuple(width() for _ in range(depth_p)), tuple(width() for _ in range(depth_q))
    O, A, B = int(rs.randint(1, 130)), int(rs.randint(1, 17)), int(rs.choice([1, rs.randint(2, 70), rs.randint(70, 400), 16 * rs.randint(1, 20)]))
    td3 = c % 3 == 2
    try:
        if td3:
            oracle, hip = make_td3_pair(O, A, B, seed=c, hidden=hp, policy_and_target_update_period=int(rs.randint(1, 4)))
            for s_ in range(3):
                nb, eps = td3_batch(B, O, A, seed=1000 * c + s_)
                want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], eps)
                td3_check(hip.train(nb, eps=eps), want)
        else:
            kw = dict(target_update_period=int(rs.randint(1, 4)), use_automatic_entropy_tuning=bool(rs.rand() < 0.8),
                      reward_scale=float(rs.choice([1.0, 0.5, 3.0])))
            oracle, hip = make_pair(O, A, B, seed=c, hidden=hp, hidden_q=hq, **kw)
            for s_ in range(3):
                nb, eps = batch_and_noise(B, O, A, seed=1000 * c + s_, term_frac=0.1)
                want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], *eps)
                check_diag(hip.train(nb, eps=eps), want, tol=2e-4 if s_ else 2 * TOL)
        assert hip.fused_mode() == 3
        print(f"case {c}: {'TD3' if td3 else 'SAC'} policy {hp} q {hq if not td3 else hp} obs {O} act {A} batch {B}: ok", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"case {c}: {'TD3' if td3 else 'SAC'} policy {hp} q {hq} obs {O} act {A} batch {B}: MISMATCH {str(e)[:200]}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

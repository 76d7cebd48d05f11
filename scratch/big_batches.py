"""Large batches against the oracle: SAC / TD3 on the fused kernels' four-launch path and the general step at 2048-8192 rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair, make_td3_pair
from tests.test_gpu_sac_step import batch_and_noise, check_diag, TOL
from tests.test_gpu_td3 import batch_and_noise as td3_batch, check_diag as td3_check
for (O, A, B, hidden, td3) in ((42, 7, 4096, (256, 256), False), (46, 7, 8192, (256, 256), False), (89, 14, 3000, (256, 256), True),
                               (42, 7, 2048, (512, 512), False), (60, 6, 5000, (300, 200, 100), False), (42, 7, 4100, (400, 300), True)):
    if td3:
        oracle, hip = make_td3_pair(O, A, B, seed=1, hidden=hidden)
        nb, eps = td3_batch(B, O, A, seed=7)
        want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], eps)
        td3_check(hip.train(nb, eps=eps), want)
    else:
        oracle, hip = make_pair(O, A, B, seed=1, hidden=hidden)
        nb, eps = batch_and_noise(B, O, A, seed=7, term_frac=0.05)
        want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], *eps)
        # (the extremes of log pi over thousands of rows sit at |a| -> 1, where log(1 - a^2 + 1e-6) amplifies the last bit of
        #  tanhf: 1e-4 there)
        check_diag(hip.train(nb, eps=eps), want, tol=1e-4)
    print(("TD3" if td3 else "SAC"), O, A, B, hidden, "step kind", hip.fused_mode(), "ok", flush=True)

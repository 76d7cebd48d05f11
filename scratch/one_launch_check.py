"""EXPERIMENT: the whole step as one launch (SAC_ONE=1: k_step1) against k_abc + k_dw_adam: the same bits; step rates."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair
from tests.test_gpu_fused_step import _buffer
shapes = ((42, 7, 256), (89, 14, 256), (46, 7, 128), (379, 6, 256), (55, 7, 240))
for (O, A, B) in shapes:
    trs = []
    for e in ("0", "1"):
        os.environ["SAC_ONE"] = e
        trs.append(make_pair(O, A, B, seed=3, noise_seed=5)[1])
    os.environ.pop("SAC_ONE")
    bufs = [_buffer(6000, O, A, 2), _buffer(6000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    outs = [t.train_loop(b, 25, batch_size=B)[1] for t, b in zip(trs, bufs)]
    sa, sb = trs[0].state_dict(), trs[1].state_dict()
    same = np.array_equal(outs[0], outs[1]) and all(np.array_equal(sa["params"][k], sb["params"][k]) for k in sa["params"]) and \
        all(np.array_equal(sa["opt"][k][j], sb["opt"][k][j]) for k in sa["opt"] for j in range(2)) and np.array_equal(sa["scalars"], sb["scalars"])
    rates = []
    for rep in range(2):
        for t, b in zip(trs, bufs):
            t.train_loop(b, 50, batch_size=B)
            t0 = time.perf_counter(); t.train_loop(b, 2000, batch_size=B); rates.append(2000 / (time.perf_counter() - t0))
    print(f"obs {O} act {A} batch {B}: kinds {trs[0].fused_mode()}/{trs[1].fused_mode()} bitwise equal {same} | two launches {rates[0]:.0f} {rates[2]:.0f} steps/s, one {rates[1]:.0f} {rates[3]:.0f}", flush=True)

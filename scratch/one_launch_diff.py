import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair
from tests.test_gpu_fused_step import _buffer
O, A, B = 42, 7, 256
for nsteps in (1, 2, 5):
    trs = []
    for e in ("0", "1"):
        os.environ["SAC_ONE"] = e
        trs.append(make_pair(O, A, B, seed=3, noise_seed=5)[1])
    bufs = [_buffer(6000, O, A, 2), _buffer(6000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    outs = [t.train_loop(b, nsteps, batch_size=B)[1] for t, b in zip(trs, bufs)]
    sa, sb = trs[0].state_dict(), trs[1].state_dict()
    print("steps", nsteps, "diag maxdiff", np.abs(np.asarray(outs[0]) - np.asarray(outs[1])).max())
    for k in sa["params"]:
        d = np.abs(sa["params"][k] - sb["params"][k]); print("  param", k, d.max(), int((d > 0).sum()), "of", d.size)
    for k in sa["opt"]:
        for j in range(2):
            d = np.abs(sa["opt"][k][j] - sb["opt"][k][j]); print("  opt", k, j, d.max(), int((d > 0).sum()))
    print("  scalars", sa["scalars"], sb["scalars"])

"""Where does the overlapped loop first differ from the four-launch loop?  (diagnostic; SAC_OVERLAP=0/1 from the environment)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.test_gpu_fused_step import _pair_of_hip, _buffer

O, A, B, steps = 42, 7, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 40
fused, plain = _pair_of_hip(O, A, B, seed=4, noise_seed=9)
bufs = [_buffer(5000, O, A, 8), _buffer(5000, O, A, 8)]
for b in bufs:
    b.seed(31)
fused.train_loop(bufs[0], steps, batch_size=B)
plain.train_loop(bufs[1], steps, batch_size=B)
ta = fused.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
tb = plain.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
bad = [i for i in range(steps) if not np.array_equal(ta[i], tb[i])]
print("SAC_OVERLAP", os.environ.get("SAC_OVERLAP"), "steps that differ:", bad[:20], "of", steps)
if bad:
    i = bad[0]
    for k in range(30):
        if ta[i, k] != tb[i, k]:
            print("  step", i, DIAG_NAMES[k] if k < len(DIAG_NAMES) else k, ta[i, k], tb[i, k])
sa, sb = fused.state_dict(), plain.state_dict()
for k in sa["params"]:
    d = np.abs(sa["params"][k] - sb["params"][k])
    if d.max() > 0:
        print("  param", k, "max diff", d.max(), "n", int((d > 0).sum()), "of", d.size)
# gradients of the LAST step (flat nn.Linear order: W0 b0 W1 b1 Wmean bmean Wstd bstd | W0 b0 W1 b1 W2 b2)
for name, shapes in (("g_policy", [(256, O), (256,), (256, 256), (256,), (A, 256), (A,), (A, 256), (A,)]),
                     ("g_qf1", [(256, O + A), (256,), (256, 256), (256,), (1, 256), (1,)])):
    n = sum(int(np.prod(s)) for s in shapes)
    ga, gb = fused.debug_fetch(name, n), plain.debug_fetch(name, n)
    off = 0
    for s in shapes:
        k = int(np.prod(s))
        a, b = ga[off:off + k].reshape(s), gb[off:off + k].reshape(s)
        off += k
        d = np.abs(a - b)
        if d.max() > 0:
            idx = np.argwhere(d > 0)
            print("  ", name, s, "max diff", d.max(), "scale", np.abs(b).max(), "n", len(idx), "first", idx[:6].tolist(),
                  "rows", sorted(set(idx[:, 0].tolist()))[:20] if idx.ndim == 2 and idx.shape[1] == 2 else "")

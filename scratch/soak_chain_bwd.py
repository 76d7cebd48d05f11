"""Soak of the batch-1024 fused launch (step kind 4): long loops stay on it (no give-up), two runs agree bit for bit, and two such
trainers (plus a batch-256 fused one) taking turns in one process give what each gives alone."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair
from tests.test_gpu_fused_step import _buffer

def state_equal(sa, sb):
    return all(np.array_equal(sa["params"][k], sb["params"][k]) for k in sa["params"]) and \
        all(np.array_equal(sa["opt"][k][j], sb["opt"][k][j]) for k in sa["opt"] for j in range(2)) and np.array_equal(sa["scalars"], sb["scalars"])

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
O, A, B = 46, 7, 1024
runs = []
for rep in range(2):
    t = make_pair(O, A, B, seed=3, noise_seed=5)[1]
    b = _buffer(20000, O, A, 2); b.seed(4)
    t0 = time.perf_counter()
    done = 0
    while done < N:
        t.train_loop(b, 5000, batch_size=B); done += 5000
    dt = time.perf_counter() - t0
    print(f"run {rep}: {N} steps, {N / dt:.0f} steps/s, step kind at the end {t.fused_mode()}", flush=True)
    runs.append((t.state_dict(), t.fused_mode()))
print("two runs bit-identical:", state_equal(runs[0][0], runs[1][0]), "| stayed on kind 4:", runs[0][1] == 4 and runs[1][1] == 4, flush=True)

# three fused trainers taking turns in one process (the library serialises their fused launches)
alone = []
shapes = ((46, 7, 1024, 3), (46, 7, 1024, 9), (42, 7, 256, 5))
for (o, a, bs, seed) in shapes:
    t = make_pair(o, a, bs, seed=seed, noise_seed=seed + 1)[1]
    b = _buffer(6000, o, a, 2); b.seed(seed)
    for _ in range(6):
        t.train_loop(b, 50, batch_size=bs)
    alone.append(t.state_dict()); del t, b
trs = [make_pair(o, a, bs, seed=seed, noise_seed=seed + 1)[1] for (o, a, bs, seed) in shapes]
bufs = []
for (o, a, bs, seed) in shapes:
    b = _buffer(6000, o, a, 2); b.seed(seed); bufs.append(b)
for _ in range(6):
    for t, b, (o, a, bs, seed) in zip(trs, bufs, shapes):
        t.train_loop(b, 50, batch_size=bs)
print("taking turns == alone:", [state_equal(t.state_dict(), s) for t, s in zip(trs, alone)], "kinds", [t.fused_mode() for t in trs], flush=True)

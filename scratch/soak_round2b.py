"""Soak of what the second half of round 2 added: (1) Wipe 379/6 -- the fused step with its first layers split over the
column parts and exchanged inside the launch -- against the four-launch step over N steps, bitwise; (2) the stepwise
interface with read-ahead (random_batch -> train from Python, an insert every 997 steps so that roll-backs happen)
against the fused loop over the same steps, bitwise."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests.helpers import synth_transitions

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000


def same(a, b):
    return all(np.array_equal(a["params"][k], b["params"][k]) for k in a["params"]) and np.array_equal(a["scalars"], b["scalars"])


t0 = time.time()
tr, buf = bench.build_replica("Wipe", 379, 6, 256, 100_000, 17, 0)
assert tr.is_fused()
_, last_f = tr.train_loop(buf, N, batch_size=256)
st_f = tr.state_dict()
os.environ["SAC_FUSED"] = "0"
tr2, buf2 = bench.build_replica("Wipe", 379, 6, 256, 100_000, 17, 0)
os.environ.pop("SAC_FUSED")
assert not tr2.is_fused()
_, last_p = tr2.train_loop(buf2, N, batch_size=256)
ok = same(st_f, tr2.state_dict()) and np.array_equal(last_f, last_p)
print("Wipe: fused (split first layers) == four-launch after", N, "steps:", ok, "QF1 loss", float(last_f[0]), "%.1f s" % (time.time() - t0), flush=True)
assert ok and np.isfinite(last_f[:28]).all()
del tr, tr2, buf, buf2

t0 = time.time()
M = N // 2
obs, act, rew, term, nobs = synth_transitions(64, 42, 7, seed=5)
runs = []
for mode in ("stepwise", "loop"):
    tr, buf = bench.build_replica("Lift", 42, 7, 256, 100_000, 17, 0)
    done = 0
    while done < M:
        n = min(997, M - done)
        if mode == "loop":
            tr.train_loop(buf, n, batch_size=256)
        else:
            for _ in range(n):
                tr.train(buf.random_batch(256))
        done += n
        buf.add_block(obs, act, rew, nobs, term)            # (a roll-back whenever batches had been read ahead)
    tr._lib.sac_sync(tr._h)
    runs.append((tr.state_dict(), buf.rng_state()))
(a, ra), (b, rb) = runs
ok = same(a, b) and np.array_equal(ra[0], rb[0]) and ra[1] == rb[1]
print("Lift: stepwise with read-ahead == fused loop after", M, "steps with inserts every 997:", ok, "%.1f s" % (time.time() - t0), flush=True)
assert ok

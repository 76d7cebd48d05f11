"""Soak of what round 3 added: (1) the stepwise interface with the buffer BOUND to np.random (host mirror, read-ahead,
an insert and a host consumer of np.random every 997 steps) against the fused loop on a private stream seeded alike, over
N steps, bitwise, np.random's own state included; (2) Door 46/7 batch 1024 on k_chain: two identically seeded runs of
N / 8 steps end bitwise equal (a rare ordering bug in the chained launch would show here) and stay within tolerance of the
four-launch step; (3) the TD3 critic pass, fused vs four launches, N / 2 steps, bitwise."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests.helpers import synth_transitions

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000


def same(a, b):
    return all(np.array_equal(a["params"][k], b["params"][k]) for k in a["params"]) and np.array_equal(a["scalars"], b["scalars"])


# (1)
t0 = time.time()
obs, act, rew, term, nobs = synth_transitions(64, 42, 7, seed=5)
runs = []
for mode in ("stepwise bound to np.random", "loop on a private stream"):
    tr, buf = bench.build_replica("Lift", 42, 7, 256, 100_000, 17, 0)
    ref = np.random.RandomState(17)
    if mode.startswith("stepwise"):
        np.random.seed(17)
        buf.bind_numpy_global_stream()
    done = 0
    while done < N:
        n = min(997, N - done)
        if mode.startswith("loop"):
            tr.train_loop(buf, n, batch_size=256)
        else:
            for _ in range(n):
                tr.train(buf.random_batch(256))
        done += n
        buf.add_block(obs, act, rew, nobs, term)
        if mode.startswith("stepwise"):
            x = np.random.uniform(size=3)                    # a host consumer between training blocks (no gauss cache: comparable)
        else:
            st = buf.rng_state()                             # the private stream makes the same three draws
            tmp = np.random.RandomState(0); tmp.set_state(("MT19937", st[0], st[1], 0, 0.0)); tmp.uniform(size=3)
            buf.seed_from_numpy(tmp)
    tr._lib.sac_sync(tr._h)
    key = np.random.get_state() if mode.startswith("stepwise") else None
    runs.append((tr.state_dict(), buf.rng_state(), key))
ok = same(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1][0], runs[1][1][0]) and runs[0][1][1] == runs[1][1][1]
ok = ok and np.array_equal(runs[0][2][1], runs[1][1][0]) and runs[0][2][2] == runs[1][1][1]
print("stepwise (bound to np.random, read-ahead, inserts, host consumers) == fused loop after", N, "steps:", ok, "%.1f s" % (time.time() - t0), flush=True)
assert ok

# (2)
t0 = time.time()
M = max(200, N // 8)
fin = []
for env in ({}, {}, {"SAC_CHAIN": "0"}):
    os.environ.update(env)
    tr, buf = bench.build_replica("Door", 46, 7, 1024, 100_000, 17, 0)
    for k in env:
        os.environ.pop(k)
    _, last = tr.train_loop(buf, M, batch_size=1024)
    fin.append((tr.state_dict(), last, tr.fused_mode()))
ok = fin[0][2] in (2, 4) and fin[2][2] == 0 and same(fin[0][0], fin[1][0]) and np.array_equal(fin[0][1], fin[1][1])
rel = float(np.max(np.abs(fin[0][1][:28] - fin[2][1][:28]) / np.maximum(1.0, np.abs(fin[2][1][:28]))))
print("chained launch (step kind %d): two runs of" % fin[0][2], M, "steps bitwise equal:", ok, "| vs four launches, max rel diff of the last diagnostics: %.2e" % rel,
      "%.1f s" % (time.time() - t0), flush=True)
assert ok and np.isfinite(fin[0][1][:28]).all() and rel < 0.5      # (chaotic divergence over thousands of steps: same regime, not same numbers)

# (3)
t0 = time.time()
M = N // 2
fin = []
for env in ({}, {"SAC_FUSED": "0"}):
    os.environ.update(env)
    tr, buf = bench.build_td3_replica(42, 7, 256, 100_000, 17)
    for k in env:
        os.environ.pop(k)
    _, last = tr.train_loop(buf, M, batch_size=256)
    fin.append((tr.state_dict(), last, tr.is_fused()))
ok = fin[0][2] and not fin[1][2] and same(fin[0][0], fin[1][0]) and np.array_equal(fin[0][1], fin[1][1])
print("TD3: fused critic pass == four launches after", M, "steps:", ok, "%.1f s" % (time.time() - t0), flush=True)
assert ok

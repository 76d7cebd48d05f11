"""Randomised sweep of the replay buffer against a host emulation (NumPy legacy RandomState + fancy indexing): random capacities
(wrap-around), insert sizes, batch sizes, seeds; random_batch (host copy and device batches), sample_indices, gather.
usage: python scratch/fuzz_buffer.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import EnvReplayBuffer

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    O, A = int(rs.randint(1, 400)), int(rs.randint(1, 17))
    cap = int(rs.choice([rs.randint(1, 50), rs.randint(50, 3000), rs.randint(3000, 60000)]))
    buf = EnvReplayBuffer(cap, obs_dim=O, action_dim=A)
    H = dict(obs=np.zeros((cap, O), np.float32), act=np.zeros((cap, A), np.float32), rew=np.zeros((cap, 1), np.float32),
             term=np.zeros((cap, 1), np.float32), nobs=np.zeros((cap, O), np.float32))
    top = size = 0
    seed = int(rs.randint(0, 2 ** 31))
    buf.seed(seed)
    ref = np.random.RandomState(seed)
    ok = True
    for op in range(int(rs.randint(3, 25))):
        kind = rs.choice(["add", "batch", "dev", "idx", "seed"], p=[0.35, 0.25, 0.2, 0.15, 0.05])
        if kind == "add" or size == 0:
            n = int(rs.choice([1, rs.randint(1, 30), rs.randint(30, 9000)]))
            blk = (rs.normal(size=(n, O)), rs.uniform(-1, 1, (n, A)), rs.uniform(0, 1, (n, 1)), rs.normal(size=(n, O)),
                   (rs.uniform(size=(n, 1)) < 0.1).astype(np.uint8))
            buf.add_block(blk[0], blk[1], blk[2], blk[3], blk[4])
            idx = (top + np.arange(n)) % cap
            keep = slice(max(0, n - cap), n)            # (a block longer than the capacity: the last `cap` rows win)
            H["obs"][idx[keep]] = blk[0][keep]; H["act"][idx[keep]] = blk[1][keep]; H["rew"][idx[keep]] = blk[2][keep]
            H["nobs"][idx[keep]] = blk[3][keep]; H["term"][idx[keep]] = blk[4][keep]
            top = (top + n) % cap; size = min(size + n, cap)
            ok = ok and buf.num_steps_can_sample() == size
        elif kind in ("batch", "dev"):
            B = int(rs.choice([1, rs.randint(1, 300), 256, rs.randint(300, 1500)]))
            want = ref.randint(0, size, B)
            if kind == "batch":
                got, idx = buf.random_batch(B, return_indices=True)
                ok = ok and np.array_equal(idx, want)
            else:
                got = buf.random_batch(B)              # a device batch: read lazily
                ok = ok and np.array_equal(got.indices(), want)
            ok = ok and all(np.array_equal(np.asarray(got[k]).reshape(B, -1), H[h][want]) for k, h in
                            (("observations", "obs"), ("actions", "act"), ("rewards", "rew"), ("terminals", "term"), ("next_observations", "nobs")))
        elif kind == "idx":
            B, K = int(rs.randint(1, 400)), int(rs.randint(1, 6))
            got = buf.sample_indices(B, K)
            want = np.stack([ref.randint(0, size, B) for _ in range(K)])
            ok = ok and np.array_equal(np.asarray(got).reshape(K, B), want)
        else:
            seed = int(rs.randint(0, 2 ** 31)); buf.seed(seed); ref = np.random.RandomState(seed)
        if not ok:
            break
    key, pos = buf.rng_state()
    st = ref.get_state()
    ok = ok and np.array_equal(key, st[1]) and pos == st[2]
    print(f"case {c}: obs {O} act {A} capacity {cap} size {size}: {'ok' if ok else 'MISMATCH at op ' + str(op) + ' ' + str(kind)}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)

"""Generates tests/golden/index_stream.npz from NumPy's legacy RandomState -- the reference's
own generator for random_batch (scripts/train.py:112 -> rlkit SimpleReplayBuffer.random_batch).
Run:  python tests/golden/make_index_fixture.py"""
import os

import numpy as np

out = {}
for seed in (17, 59, 83, 129, 251):
    for size in (3300, 5800, 10_000, 1_000_000):
        rs = np.random.RandomState(seed)
        out[f"s{seed}_n{size}"] = np.concatenate([rs.randint(0, size, 256) for _ in range(4)])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "index_stream.npz"), **out)
print("wrote", len(out), "vectors")

"""Oracle: the index stream of ``EnvReplayBuffer.random_batch``  (TEST INFRASTRUCTURE).

Follows (call sites; the arithmetic itself is NumPy's frozen legacy stream):

* /root/reference/scripts/train.py:112      ``np.random.seed(args.seed)``
* /root/reference/util/rlkit_custom.py:235  ``replay_buffer.random_batch(batch_size)``
  -> rlkit ``SimpleReplayBuffer.random_batch``: ``np.random.randint(0, self._size, batch_size)``
  (rlkit is not vendored; pinned commits b7f97b2 / d63dab7, README.md:28).

Restated algorithm: MT19937 seeded with ``init_genrand``;
``randint(0, size, B)`` with the default int64 dtype and ``size-1 <= 0xffffffff``
uses *masked rejection on 32-bit draws*: ``mask`` = smallest ``2^k - 1 >= size-1``;
per output draw ``v = next_u32() & mask`` until ``v <= size-1``.  Rejected draws
are consumed.  ``size == 1`` consumes nothing.

Pinned against ``numpy.random.RandomState`` itself in tests/test_oracle_index_stream.py
(NumPy *is* the reference implementation of this stream), including the
generator state after the call.
"""
from __future__ import annotations

import numpy as np

N = 624
M = 397
MATRIX_A = 0x9908B0DF
UPPER = 0x80000000
LOWER = 0x7FFFFFFF


class MT19937:
    """Plain MT19937 with NumPy's legacy ``seed(int)`` initialisation."""

    def __init__(self, seed: int | None = None):
        self.mt = np.zeros(N, dtype=np.uint32)
        self.pos = N
        if seed is not None:
            self.seed(seed)

    def seed(self, seed: int) -> None:
        mt = [0] * N
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt = np.array(mt, dtype=np.uint32)
        self.pos = N

    # state exchange in NumPy's ('MT19937', key, pos, ...) convention
    def get_state(self):
        return self.mt.copy(), int(self.pos)

    def set_state(self, key, pos) -> None:
        self.mt = np.asarray(key, dtype=np.uint32).copy()
        self.pos = int(pos)

    def _twist(self) -> None:
        mt = self.mt.astype(np.uint64)
        # sequential semantics: element i uses new values of i+M-N once i >= N-M.
        # Vectorised in the three dependency-free spans [0,227) [227,454) [454,623] + last.
        def mix(u, v):
            y = (u & UPPER) | (v & LOWER)
            return (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(MATRIX_A), np.uint64(0))

        a = N - M  # 227
        mt[0:a] = mt[M:N] ^ mix(mt[0:a], mt[1:a + 1])
        mt[a:2 * a] = mt[0:a] ^ mix(mt[a:2 * a], mt[a + 1:2 * a + 1])
        mt[2 * a:N - 1] = mt[a:N - 1 - a] ^ mix(mt[2 * a:N - 1], mt[2 * a + 1:N])
        mt[N - 1] = mt[M - 1] ^ mix(mt[N - 1:N], mt[0:1])[0]
        self.mt = mt.astype(np.uint32)
        self.pos = 0

    @staticmethod
    def temper(y: np.ndarray) -> np.ndarray:
        y = y.astype(np.uint32)
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9D2C5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xEFC60000))
        y = y ^ (y >> np.uint32(18))
        return y

    def next_block(self, n: int) -> np.ndarray:
        """Up to ``n`` tempered 32-bit outputs from the current block (at least 1)."""
        if self.pos >= N:
            self._twist()
        take = min(n, N - self.pos)
        out = self.temper(self.mt[self.pos:self.pos + take])
        self.pos += take
        return out

    def next_u32(self) -> int:
        return int(self.next_block(1)[0])


def bounded_mask(rng: int) -> int:
    mask = rng
    mask |= mask >> 1
    mask |= mask >> 2
    mask |= mask >> 4
    mask |= mask >> 8
    mask |= mask >> 16
    return mask


def randint_masked(gen: MT19937, size: int, count: int) -> np.ndarray:
    """``np.random.randint(0, size, count)`` on the legacy stream (int64 output).

    Sequential semantics kept exactly: draws are consumed one at a time, in order,
    and the generator stops right after the draw that produced the last output.
    """
    if size <= 0:
        raise ValueError("low >= high")
    out = np.empty(count, dtype=np.int64)
    rng = size - 1
    if rng == 0:
        out[:] = 0
        return out
    if rng > 0xFFFFFFFF:
        raise NotImplementedError("replay buffers above 2^32 slots are out of scope")
    mask = np.uint32(bounded_mask(rng))
    filled = 0
    while filled < count:
        need = count - filled
        start_pos = gen.pos if gen.pos < N else 0
        blk = gen.next_block(N)            # the rest of the current block
        v = blk & mask
        ok = v <= np.uint32(rng)
        acc_idx = np.flatnonzero(ok)
        if len(acc_idx) >= need:
            last = acc_idx[need - 1]
            out[filled:] = v[acc_idx[:need]]
            # un-consume the draws after the one that produced the last output
            gen.pos = start_pos + int(last) + 1
            filled = count
        else:
            out[filled:filled + len(acc_idx)] = v[acc_idx]
            filled += len(acc_idx)
    return out


class ReplayIndexStream:
    """The global ``np.random`` stream as the replay buffer sees it."""

    def __init__(self, seed: int):
        self.gen = MT19937(seed)

    def random_batch_indices(self, size: int, batch_size: int) -> np.ndarray:
        return randint_masked(self.gen, size, batch_size)

"""Pins oracle/replay_index_stream.{py,c} against NumPy's own legacy RandomState -- the
generator the reference seeds (scripts/train.py:112) and rlkit's random_batch draws from."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle.replay_index_stream import MT19937, ReplayIndexStream, randint_masked

SEEDS = [1, 17, 59, 83, 129, 251]
SIZES = [1, 2, 3, 3300, 5800, 10_000, 999_999, 1_000_000, 2 ** 20, 2 ** 20 + 1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", SEEDS)
def test_seed_state_matches_numpy(seed):
    rs = np.random.RandomState(seed)
    _, key, pos, _, _ = rs.get_state()
    g = MT19937(seed)
    assert np.array_equal(g.mt, key) and g.pos == pos


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("size", SIZES)
def test_randint_matches_numpy(seed, size):
    rs = np.random.RandomState(seed)
    g = MT19937(seed)
    for batch in (1, 128, 256, 1024, 7):
        want = rs.randint(0, size, batch)
        got = randint_masked(g, size, batch)
        assert got.dtype == np.int64
        assert np.array_equal(want, got)
    _, key, pos, _, _ = rs.get_state()
    # NumPy twists lazily: compare the *future stream*, and the raw state when both are mid-block
    if pos < 624 and g.pos < 624:
        assert np.array_equal(g.mt, key) and g.pos == pos
    assert np.array_equal(rs.randint(0, 1 << 31, 1000), randint_masked(g, 1 << 31, 1000))


def test_many_steps_equal_one_long_draw():
    """1000 random_batch(256) calls consume the same stream as one randint of 256000."""
    a = ReplayIndexStream(17)
    seq = np.concatenate([a.random_batch_indices(1_000_000, 256) for _ in range(50)])
    b = ReplayIndexStream(17)
    assert np.array_equal(seq, b.random_batch_indices(1_000_000, 256 * 50))
    rs = np.random.RandomState(17)
    assert np.array_equal(seq, rs.randint(0, 1_000_000, 256 * 50))


def test_reference_point_seed17_draw_count():
    """seed 17, size 1e6, B=256 consumes 265 draws (pos 624 -> 265)."""
    g = MT19937(17)
    randint_masked(g, 1_000_000, 256)
    assert g.pos == 265
    rs = np.random.RandomState(17)
    rs.randint(0, 1_000_000, 256)
    assert rs.get_state()[2] == 265


def test_choice_spelling_is_same_stream():
    """A newer rlkit spells it np.random.choice(size, B, replace=True) -- same stream."""
    for seed in SEEDS:
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        assert np.array_equal(r1.randint(0, 5800, 128), r2.choice(5800, 128, replace=True))


def test_golden_index_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "index_stream.npz"))
    for seed in (17, 59, 83, 129, 251):
        for size in (3300, 5800, 10_000, 1_000_000):
            g = ReplayIndexStream(seed)
            got = np.concatenate([g.random_batch_indices(size, 256) for _ in range(4)])
            assert np.array_equal(got, z[f"s{seed}_n{size}"])


def test_c_restatement_matches_numpy(tmp_path):
    so = tmp_path / "liboracle_index.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so),
                           os.path.join(ROOT, "oracle", "replay_index_stream.c")])
    lib = ctypes.CDLL(str(so))
    lib.oracle_randint.restype = ctypes.c_int64
    lib.oracle_randint.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
    state = ctypes.create_string_buffer(4 * 624 + 4)
    for seed in (17, 251):
        for size in (2, 5800, 1_000_000):
            lib.oracle_mt_seed(state, ctypes.c_uint32(seed))
            out = np.empty(4096, dtype=np.int64)
            lib.oracle_randint(state, size, out.size, out.ctypes.data)
            assert np.array_equal(out, np.random.RandomState(seed).randint(0, size, out.size))

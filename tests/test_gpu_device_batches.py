"""GPU: the reference's UNMODIFIED stepwise loop (random_batch -> train, /root/reference/util/rlkit_custom.py:235-238)
on device-resident batches: same numbers as host batches and as the fused loop, no copy unless somebody looks."""
import numpy as np
import pytest

from robosuite_benchmark_amd import DeviceBatch, EnvReplayBuffer
from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu


def filled(n, O, A, seed, **kw):
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed, term_frac=0.05)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A, **kw)
    buf.add_block(obs, act, rew, nobs, term)
    return buf


def test_lazy_batch_is_the_eager_batch():
    O, A, B, n = 42, 7, 256, 5000
    lazy, eager = filled(n, O, A, 3), filled(n, O, A, 3, lazy_batches=False)
    lazy.seed(17); eager.seed(17)
    rs = np.random.RandomState(17)
    for _ in range(3):
        a, b = lazy.random_batch(B), eager.random_batch(B)
        assert isinstance(a, DeviceBatch) and a.on_device and type(b) is dict
        assert np.array_equal(a.indices(), rs.randint(0, n, B))
        assert a.on_device                                   # indices() does not pull the batch
        assert set(a.keys()) == set(b.keys()) and not a.on_device
        for k in b:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    assert isinstance(a, dict) and len(a) == 5 and "rewards" in a


@pytest.mark.parametrize("O,A,B,steps", [(42, 7, 256, 40), (10, 3, 32, 100)])
def test_stepwise_on_device_batches_equals_fused_loop(O, A, B, steps):
    n = 8000
    _, fused = make_pair(O, A, B, seed=4, noise_seed=5)
    _, stepw = make_pair(O, A, B, seed=4, noise_seed=5)
    buf_a, buf_b = filled(n, O, A, 8), filled(n, O, A, 8)
    buf_a.seed(17); buf_b.seed(17)
    first, last = fused.train_loop(buf_a, steps, batch_size=B)
    outs = [stepw.train(buf_b.random_batch(B)) for _ in range(steps)]     # the reference's loop body, unchanged
    assert outs[0] is not None and all(o is None for o in outs[1:])       # diagnostics only when the epoch needs them
    assert np.array_equal(outs[0], first)
    stepw.end_epoch(0)
    sa, sb = fused.state_dict(), stepw.state_dict()
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    assert np.array_equal(sa["scalars"], sb["scalars"])
    (ka, pa), (kb, pb) = buf_a.rng_state(), buf_b.rng_state()
    assert pa == pb and np.array_equal(ka, kb)
    assert stepw.get_diagnostics()["QF1 Loss"] == float(first[0])


def test_a_batch_somebody_read_trains_the_same():
    O, A, B, n = 46, 7, 64, 3000
    _, t1 = make_pair(O, A, B, seed=2, noise_seed=9)
    _, t2 = make_pair(O, A, B, seed=2, noise_seed=9)
    b1, b2 = filled(n, O, A, 6), filled(n, O, A, 6)
    b1.seed(3); b2.seed(3)
    for _ in range(5):
        dev, host = b1.random_batch(B), b2.random_batch(B)
        _ = host["rewards"].mean()                           # looking at it moves it to the host
        assert dev.on_device and not host.on_device
        t1.train(dev)
        t2.train(host)
    s1, s2 = t1.state_dict(), t2.state_dict()
    for k in s1["params"]:
        assert np.array_equal(s1["params"][k], s2["params"][k]), k


def test_expired_device_batch_is_refused_not_wrong():
    O, A, B = 11, 2, 32
    _, tr = make_pair(O, A, B, seed=1)
    buf = filled(500, O, A, 4)
    buf.seed(1)
    old = buf.random_batch(B)
    keep = [buf.random_batch(B) for _ in range(16)]         # the ring holds 16 batches
    with pytest.raises(RuntimeError, match="expired"):
        old["observations"]
    with pytest.raises(RuntimeError, match="expired"):
        tr.train(old)
    assert keep[-1]["actions"].shape == (B, A) and keep[0]["rewards"].shape == (B, 1)
    other = EnvReplayBuffer(500, obs_dim=O + 1, action_dim=A)
    other.add_block(np.zeros((10, O + 1), np.float32), np.zeros((10, A), np.float32), np.zeros(10, np.float32),
                    np.zeros((10, O + 1), np.float32), np.zeros(10, np.uint8))
    with pytest.raises(RuntimeError, match="does not match"):
        tr.train(other.random_batch(B))

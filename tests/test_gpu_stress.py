"""GPU: longer runs and handle churn -- the loop stays finite over thousands of steps (several slot-ring
wrap-arounds), repeated create/destroy does not leak or corrupt, and odd-but-legal shapes work."""
import numpy as np
import pytest

from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu


def filled(n, O, A, seed):
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(obs, act, rew, nobs, term)
    return buf


def test_five_thousand_steps_stay_finite_and_keep_learning():
    O, A, B = 42, 7, 256
    _, hip = make_pair(O, A, B, seed=8, noise_seed=5)
    buf = filled(50_000, O, A, 1)
    buf.seed(3)
    i = DIAG_NAMES.index
    first, last = hip.train_loop(buf, 3000, batch_size=B)          # 12 chunks of the double-buffered ring
    assert np.all(np.isfinite(first)) and np.all(np.isfinite(last))
    first2, last2 = hip.train_loop(buf, 2000, batch_size=B)
    assert np.all(np.isfinite(last2))
    assert last2[i("QF1 Loss")] < first[i("QF1 Loss")] * 0.1
    sc = hip.state_dict()["scalars"]
    assert sc[3] == 5000 and sc[4] == 5000
    # the device generator consumed exactly what 5000 random_batch(256) calls consume
    key, pos = buf.rng_state()
    rs = np.random.RandomState(3)
    rs.randint(0, 50_000, 256 * 5000)
    assert np.array_equal(rs.get_state()[1], key) and rs.get_state()[2] == pos      # the generator's words and position
    tmp = np.random.RandomState(0)
    buf.sync_to_numpy(tmp)
    assert np.array_equal(tmp.randint(0, 1 << 30, 100), rs.randint(0, 1 << 30, 100))
    for name in ("policy", "qf1", "target_qf2"):
        assert np.all(np.isfinite(hip.state_dict()["params"][name]))


def test_handle_churn():
    ref = None
    for rep in range(12):
        _, hip = make_pair(10, 3, 32, seed=2, noise_seed=9)
        buf = filled(2000, 10, 3, 4)
        buf.seed(11)
        _, last = hip.train_loop(buf, 40, batch_size=32)
        if ref is None:
            ref = last
        assert np.array_equal(last, ref)                       # fresh handles reproduce bit for bit
        del hip, buf


@pytest.mark.parametrize("O,A,B", [(1, 1, 16), (3, 16, 48), (130, 2, 32), (64, 8, 80)])
def test_odd_shapes(O, A, B):
    oracle, hip = make_pair(O, A, B, seed=6)
    obs, act, rew, term, nobs = synth_transitions(B, O, A, seed=12, term_frac=0.1)
    rs = np.random.RandomState(1)
    eps = (rs.normal(size=(B, A)).astype(np.float32), rs.normal(size=(B, A)).astype(np.float32))
    want = oracle.step(obs, act, rew, term.astype(np.float32), nobs, *eps)
    got = hip.train(dict(observations=obs, actions=act, rewards=rew, terminals=term.astype(np.float32),
                         next_observations=nobs), eps=eps)
    for i, name in enumerate(DIAG_NAMES):
        assert abs(got[i] - want[name]) <= 1e-5 * max(1.0, abs(want[name])), (name, got[i], want[name])
    buf = filled(500, O, A, 2)
    buf.seed(5)
    batch, idx = buf.random_batch(B, return_indices=True)
    assert np.array_equal(idx, np.random.RandomState(5).randint(0, 500, B))


def test_act_dim_above_sixteen_is_rejected_cleanly():
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    pol = TanhGaussianPolicy([256, 256], 20, 17)
    qs = [FlattenMlp([256, 256], 1, 37) for _ in range(4)]
    with pytest.raises(RuntimeError, match="act_dim"):
        SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=32)


def test_td3_three_thousand_steps_with_an_odd_batch_stay_finite():
    from robosuite_benchmark_amd._lib import TD3_DIAG_NAMES
    from tests.helpers import make_td3_pair
    O, A, B = 46, 7, 200                                   # 200 = 12.5 row-blocks: padded slots all the way
    _, hip = make_td3_pair(O, A, B, seed=8, noise_seed=5)
    buf = filled(30_000, O, A, 1)
    buf.seed(3)
    first, last = hip.train_loop(buf, 3001, batch_size=B)
    i = TD3_DIAG_NAMES.index
    assert np.all(np.isfinite(first[:28])) and np.all(np.isfinite(last[:28]))
    assert last[i("QF1 Loss")] < first[i("QF1 Loss")]
    sc = hip.state_dict()["scalars"]
    assert (sc[0], sc[3], sc[4]) == (1501.0, 3001.0, 3001.0)      # policy steps on even step numbers
    tmp, rs = np.random.RandomState(0), np.random.RandomState(3)
    rs.randint(0, 30_000, B * 3001)
    buf.sync_to_numpy(tmp)
    assert np.array_equal(tmp.randint(0, 1 << 30, 50), rs.randint(0, 1 << 30, 50))
    for name in ("policy", "target_policy", "qf1", "target_qf2"):
        assert np.all(np.isfinite(hip.state_dict()["params"][name]))

"""GPU: batches of 1024 rows and more (BASELINE config 3: Door, batch 1024) -- the step with its two forward launches
as ONE (k_chain, csrc/sac_chain.h: column split 1, a workgroup runs the policy, takes its own head and goes on into the Q
nets) against the four-launch step of the same library and against the oracle."""
import os

import numpy as np
import pytest

from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.helpers import make_pair, rel_err, synth_transitions

pytestmark = pytest.mark.gpu


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _buffer(n, O, A, seed):
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed, term_frac=0.05)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(obs, act, rew, nobs, term)
    return buf


# Door 1024 = BASELINE config 3; 2048; a two-tile head (2A > 16); wide first layers (Wipe); a padded last row-block
SHAPES = [(46, 7, 1024), (42, 7, 2048), (89, 14, 1024), (379, 6, 1024), (46, 7, 1010)]


@pytest.mark.parametrize("O,A,B", SHAPES)
def test_chain_step_against_the_four_launch_step(O, A, B):
    chain = _with_env(dict(SAC_CHAIN=1), lambda: make_pair(O, A, B, seed=4, noise_seed=9)[1])
    four = _with_env(dict(SAC_CHAIN=0), lambda: make_pair(O, A, B, seed=4, noise_seed=9)[1])
    assert chain.fused_mode() in (2, 4) and four.fused_mode() == 0
    bufs = [_buffer(6000, O, A, 8), _buffer(6000, O, A, 8)]
    for b in bufs:
        b.seed(31)
    fa, _ = chain.train_loop(bufs[0], 1, batch_size=B)
    fb, _ = four.train_loop(bufs[1], 1, batch_size=B)
    # the policy passes, both heads, Q_i(s, a) and the target nets run the same MFMA sequences: bit for bit
    for name, n in (("a_new", B * A), ("log_pi", B), ("mu", B * A), ("log_std", B * A), ("a_next", B * A), ("log_pi_next", B),
                    ("q1", B), ("q2", B), ("q_target", B), ("g_qf1", None), ("g_qf2", None)):
        n = n or chain.state_dict()["params"][name[2:]].size
        assert np.array_equal(chain.debug_fetch(name, n), four.debug_fetch(name, n)), name
    # Q_i(s, a_new) contracts [obs | a_new] directly instead of adding W1[:, action] (a_new - a) to launch A's z: rounding
    for name, tol in (("q1_new", 2e-6), ("q2_new", 2e-6)):
        x, y = chain.debug_fetch(name, B), four.debug_fetch(name, B)
        assert np.max(np.abs(x - y)) <= tol * max(1.0, float(np.max(np.abs(y)))), name
    gx, gy = chain.debug_fetch("g_policy", chain.state_dict()["params"]["policy"].size), four.debug_fetch("g_policy", four.state_dict()["params"]["policy"].size)
    assert np.max(np.abs(gx - gy)) <= 2e-5 * float(np.max(np.abs(gy)))
    for i, name in enumerate(DIAG_NAMES):
        assert abs(fa[i] - fb[i]) <= 2e-6 * max(1.0, abs(fb[i])), name
    # a few more steps: the trajectories stay together
    _, la = chain.train_loop(bufs[0], 5, batch_size=B)
    _, lb = four.train_loop(bufs[1], 5, batch_size=B)
    assert np.all(np.isfinite(la))
    for i, name in enumerate(DIAG_NAMES):
        assert abs(la[i] - lb[i]) <= 1e-4 * max(1.0, abs(lb[i])), name
    (ka, pa), (kb, pb) = bufs[0].rng_state(), bufs[1].rng_state()
    assert pa == pb and np.array_equal(ka, kb)


@pytest.mark.parametrize("task,O,A,B", [("Door", 46, 7, 1024), ("TwoArmLift", 89, 14, 1024), ("Lift", 42, 7, 2048)])
def test_chain_step_against_the_oracle(task, O, A, B):
    oracle, hip = _with_env(dict(SAC_CHAIN=1), lambda: make_pair(O, A, B, seed=3))
    assert hip.fused_mode() in (2, 4)
    obs, act, rew, term, nobs = synth_transitions(B, O, A, seed=11, term_frac=0.05)
    rs = np.random.RandomState(2)
    eps = (rs.normal(size=(B, A)).astype(np.float32), rs.normal(size=(B, A)).astype(np.float32))
    batch = dict(observations=obs, actions=act, rewards=rew, terminals=term.astype(np.float32), next_observations=nobs)
    for _ in range(2):
        want = oracle.step(obs, act, rew, term.astype(np.float32), nobs, *eps)
        got = hip.train(batch, eps=eps)
        hip.end_epoch(0)
        for i, name in enumerate(DIAG_NAMES):
            assert abs(got[i] - want[name]) <= 1e-5 * max(1.0, abs(want[name])), (name, got[i], want[name])   # north_star: 1e-5
        L = oracle.last
        for name, ref, scale in (("a_new", L["a_new"], 1.0), ("log_pi", L["log_pi"], None), ("q1_new", L["q1_new"], None),
                                 ("q2_new", L["q2_new"], None), ("q_target", L["y"], None)):
            ref = ref.detach().numpy().reshape(-1)
            x = hip.debug_fetch(name, ref.size)
            assert np.max(np.abs(x - ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(ref)))), name
        for name, key in (("g_policy", "g_policy"), ("g_qf1", "g_qf1")):
            ref = L[key]
            # oracle gradients: [W0, W1, ..., b0, b1, ...]; the library's flat layout is W0 b0 W1 b1 ...
            nl = len(ref) // 2
            flat = np.concatenate([np.concatenate([ref[l].ravel(), ref[nl + l].ravel()]) for l in range(nl)])
            x = hip.debug_fetch(name, flat.size)
            assert np.max(np.abs(x - flat)) <= 5e-5 * float(np.max(np.abs(flat))), name


def test_which_batches_take_the_chained_launch():
    """Default selection: column split 1 (1024 rows and more, an even number of row-blocks) and first layers of at most eight
    k-chunks (the eight-wave kernel also pays beyond one round of workgroups: batch 2048); exactly one workgroup per CU and
    narrow first layers: the backward blocks inside the same launch (kind 4)."""
    kinds = {}
    for (O, A, B) in [(46, 7, 1024), (86, 14, 1024), (379, 6, 1024), (46, 7, 2048), (46, 7, 512), (42, 7, 256)]:
        kinds[(O, A, B)] = _with_env(dict(SAC_CHAIN=None, SAC_FUSED=None), lambda: make_pair(O, A, B, seed=1)[1]).fused_mode()
    assert kinds == {(46, 7, 1024): 4, (86, 14, 1024): 2, (379, 6, 1024): 0, (46, 7, 2048): 2, (46, 7, 512): 0, (42, 7, 256): 1}


@pytest.mark.parametrize("O,A,B", [(46, 7, 1024), (86, 14, 1024), (42, 7, 992), (300, 6, 1024)])
def test_eight_wave_chain_kernel_is_the_four_wave_one_bit_for_bit(O, A, B, monkeypatch):
    """k_chain8 (512 threads per workgroup: two 16-column tiles per wave, two waves per SIMD) runs every tile's MFMA sequence
    and every split-K sum in k_chain's order: same trajectory, bit for bit (SAC_CHAIN8=0 selects the four-wave kernel)."""
    from tests.test_gpu_fused_step import _buffer
    monkeypatch.setenv("SAC_CHAIN", "1")
    monkeypatch.setenv("SAC_CHAIN_BWD", "0")
    monkeypatch.setenv("SAC_CHAIN8", "0")
    _, four = make_pair(O, A, B, seed=3, noise_seed=5)
    monkeypatch.setenv("SAC_CHAIN8", "1")
    _, eight = make_pair(O, A, B, seed=3, noise_seed=5)
    assert four.fused_mode() == 2 and eight.fused_mode() == 2
    bufs = [_buffer(5000, O, A, 2), _buffer(5000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    la = four.train_loop(bufs[0], 12, batch_size=B)[1]
    lb = eight.train_loop(bufs[1], 12, batch_size=B)[1]
    assert np.array_equal(la, lb)
    sa, sb = four.state_dict(), eight.state_dict()
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    for k in sa["opt"]:
        assert np.array_equal(sa["opt"][k][0], sb["opt"][k][0]) and np.array_equal(sa["opt"][k][1], sb["opt"][k][1]), k


@pytest.mark.parametrize("O,A,B", [(46, 7, 1024), (86, 14, 1024), (46, 7, 2048), (379, 6, 1040)])
def test_eight_wave_backward_kernel_is_the_four_wave_one_bit_for_bit(O, A, B, monkeypatch):
    """k_bwd8 (column split 1, 512 threads per workgroup) against k_bwd<.., 1>: same trajectory, bit for bit, behind
    k_chain8 and behind the two forward launches (SAC_BWD8=0 selects the four-wave kernel)."""
    from tests.test_gpu_fused_step import _buffer
    monkeypatch.setenv("SAC_CHAIN_BWD", "0")
    monkeypatch.setenv("SAC_BWD8", "0")
    _, four = make_pair(O, A, B, seed=3, noise_seed=5)
    monkeypatch.setenv("SAC_BWD8", "1")
    _, eight = make_pair(O, A, B, seed=3, noise_seed=5)
    bufs = [_buffer(5000, O, A, 2), _buffer(5000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    la = four.train_loop(bufs[0], 9, batch_size=B)[1]
    lb = eight.train_loop(bufs[1], 9, batch_size=B)[1]
    assert np.array_equal(la, lb)
    sa, sb = four.state_dict(), eight.state_dict()
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    for k in sa["opt"]:
        assert np.array_equal(sa["opt"][k][0], sb["opt"][k][0]) and np.array_equal(sa["opt"][k][1], sb["opt"][k][1]), k
    for name in ("g_policy", "g_qf1", "g_qf2"):
        n = sa["params"][name[2:]].size
        assert np.array_equal(four.debug_fetch(name, n), eight.debug_fetch(name, n)), name


def _chain_pair(O, A, B, monkeypatch, **env):
    """(k_chain8 with the backward blocks inside, k_chain8 + k_bwd8) with identical parameters"""
    monkeypatch.setenv("SAC_CHAIN_BWD", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    _, one = make_pair(O, A, B, seed=3, noise_seed=5)
    for k in env:
        monkeypatch.delenv(k)
    monkeypatch.setenv("SAC_CHAIN_BWD", "0")
    _, two = make_pair(O, A, B, seed=3, noise_seed=5)
    monkeypatch.delenv("SAC_CHAIN_BWD")
    return one, two


def _same_state(sa, sb):
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    for k in sa["opt"]:
        assert np.array_equal(sa["opt"][k][0], sb["opt"][k][0]) and np.array_equal(sa["opt"][k][1], sb["opt"][k][1]), k
    assert np.array_equal(sa["scalars"], sb["scalars"])


@pytest.mark.parametrize("O,A,B", [(46, 7, 1024), (86, 14, 1024), (42, 7, 1010), (30, 16, 800), (10, 3, 544)])
def test_backward_blocks_inside_the_chained_launch_bit_for_bit(O, A, B, monkeypatch):
    """k_chain8<.., BWD>: two launches per step at batch 1024 -- the backward blocks behind in-launch hand-offs (the entropy
    coefficient's log-pi sums, the target values, the other twin's action gradient).  Same blocks, same arithmetic as
    k_chain8 + k_bwd8: the same trajectory bit for bit, through the loop and the stepwise interface."""
    from tests.test_gpu_fused_step import _buffer
    one, two = _chain_pair(O, A, B, monkeypatch)
    assert one.fused_mode() == 4 and two.fused_mode() == 2
    bufs = [_buffer(5000, O, A, 2), _buffer(5000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    fa, la = one.train_loop(bufs[0], 11, batch_size=B)
    fb, lb = two.train_loop(bufs[1], 11, batch_size=B)
    assert np.array_equal(fa, fb) and np.array_equal(la, lb)
    for _ in range(5):
        one.train(bufs[0].random_batch(B)); two.train(bufs[1].random_batch(B))
    one._lib.sac_sync(one._h); two._lib.sac_sync(two._h)
    _same_state(one.state_dict(), two.state_dict())
    for name in ("g_policy", "g_qf1", "g_qf2"):
        n = one.state_dict()["params"][name[2:]].size
        la_, lb_ = one.train_loop(bufs[0], 1, batch_size=B)[1], two.train_loop(bufs[1], 1, batch_size=B)[1]
        assert np.array_equal(one.debug_fetch(name, n), two.debug_fetch(name, n)), name


@pytest.mark.parametrize("stall_at,script", [(3, "loop"), (1, "loop"), (7, "step")])
def test_chained_launch_with_backward_giving_up_falls_back_transparently(stall_at, script, monkeypatch):
    """A hand-off of launch `stall_at` times out (test hook: item N of row-block 0 leaves without publishing): that step and
    everything queued behind it apply nothing, the trainer falls back to k_chain8 + k_bwd8 for good and re-runs the lost
    steps -- same trajectory, counters and generator as an undisturbed run."""
    from tests.test_gpu_fused_step import _buffer
    O, A, B = 46, 7, 1024
    one, two = _chain_pair(O, A, B, monkeypatch, SAC_FUSED_TEST_STALL=stall_at)
    bufs = [_buffer(5000, O, A, 2), _buffer(5000, O, A, 2)]
    for b in bufs:
        b.seed(4)
    n = 12
    if script == "loop":
        one.train_loop(bufs[0], n, batch_size=B); two.train_loop(bufs[1], n, batch_size=B)
    else:
        for _ in range(n):
            one.train(bufs[0].random_batch(B)); two.train(bufs[1].random_batch(B))
        one._lib.sac_sync(one._h); two._lib.sac_sync(two._h)
    assert one.fused_mode() == 2
    _same_state(one.state_dict(), two.state_dict())
    (ka, pa), (kb, pb) = bufs[0].rng_state(), bufs[1].rng_state()
    assert pa == pb and np.array_equal(ka, kb)

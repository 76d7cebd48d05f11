"""GPU: the fused step (k_abc: launches A + B + C as one launch with in-launch hand-offs, csrc/sac_fused.h) against the
four-launch step of the same library -- identical arithmetic in identical order, so everything must agree bit for
bit -- and the give-up path of the hand-offs (timeout -> nothing applied -> four-launch fall-back -> the lost steps re-run)."""
import os

import numpy as np
import pytest

from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu


def _pair_of_hip(O, A, B, seed, noise_seed, **env):
    """(fused, four-launch) trainers with identical parameters."""
    old = {k: os.environ.get(k) for k in ("SAC_FUSED", "SAC_FUSED_TEST_STALL")}
    try:
        os.environ.pop("SAC_FUSED", None)
        for k, v in env.items():
            os.environ[k] = str(v)
        _, fused = make_pair(O, A, B, seed=seed, noise_seed=noise_seed)
        os.environ.pop("SAC_FUSED_TEST_STALL", None)
        os.environ["SAC_FUSED"] = "0"
        _, plain = make_pair(O, A, B, seed=seed, noise_seed=noise_seed)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return fused, plain


def _buffer(n, O, A, seed):
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed, term_frac=0.05)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(obs, act, rew, nobs, term)
    return buf


def _same(sa, sb):
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    for k in sa["opt"]:
        assert np.array_equal(sa["opt"][k][0], sb["opt"][k][0]) and np.array_equal(sa["opt"][k][1], sb["opt"][k][1]), k
    assert np.array_equal(sa["scalars"], sb["scalars"])


# every sweep shape (batch <= 256), odd batch sizes (padded row-blocks), one and sixteen row-blocks, both head widths
# (2A <= 16 and > 16), narrow and wide first layers
@pytest.mark.parametrize("O,A,B,steps", [(42, 7, 256, 40), (42, 7, 128, 30), (46, 7, 256, 12), (55, 7, 256, 12), (73, 12, 256, 12),
                                         (86, 14, 256, 12), (89, 14, 256, 12), (64, 4, 128, 12), (50, 4, 100, 12),
                                         (10, 3, 16, 25), (112, 16, 48, 8), (1, 1, 17, 8), (60, 7, 250, 8),
                                         # wide first layers (obs_dim > 112: a refilling ring instead of a held one): Wipe
                                         (379, 6, 256, 10), (379, 7, 128, 6), (130, 5, 64, 8), (113, 7, 48, 6), (496, 3, 32, 4),
                                         # ... with a two-tile head (2A > 16) and at the 128-column boundary of the split
                                         (150, 12, 64, 6), (300, 14, 256, 5), (128, 9, 80, 5), (112, 8, 96, 5)])
def test_fused_step_equals_four_launch_step_bitwise(O, A, B, steps):
    fused, plain = _pair_of_hip(O, A, B, seed=4, noise_seed=9)
    assert fused.is_fused() and not plain.is_fused()
    bufs = [_buffer(5000, O, A, 8), _buffer(5000, O, A, 8)]
    for b in bufs:
        b.seed(31)
    fa, la = fused.train_loop(bufs[0], steps, batch_size=B)
    fb, lb = plain.train_loop(bufs[1], steps, batch_size=B)
    assert np.array_equal(fa, fb) and np.array_equal(la, lb)
    ta = fused.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
    tb = plain.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
    assert np.array_equal(ta, tb)
    _same(fused.state_dict(), plain.state_dict())
    for name, n in (("a_new", B * A), ("log_pi", B), ("q1", B), ("q2", B), ("q_target", B), ("q1_new", B), ("q2_new", B),
                    ("log_pi_next", B)):
        assert np.array_equal(fused.debug_fetch(name, n), plain.debug_fetch(name, n)), name
    for name in ("g_policy", "g_qf1", "g_qf2"):
        n = fused.state_dict()["params"][name[2:]].size
        assert np.array_equal(fused.debug_fetch(name, n), plain.debug_fetch(name, n)), name


def test_unsupported_shapes_are_refused_not_miscomputed():
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    pol = TanhGaussianPolicy([256, 256], 500, 3)
    qs = [FlattenMlp([256, 256], 1, 503) for _ in range(4)]
    with pytest.raises(RuntimeError, match="obs_dim 500 unsupported"):
        SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=32)


def test_which_shapes_take_the_fused_step():
    for (O, A, B), want in (((42, 7, 256), True), ((42, 7, 512), False), ((379, 6, 256), True), ((112, 7, 64), True),
                            ((113, 7, 64), True), ((42, 7, 1), True), ((42, 7, 272), False)):
        _, hip = make_pair(O, A, B, seed=1)
        assert hip.is_fused() is want, (O, A, B)


def test_stepwise_and_host_batches_on_the_fused_step():
    """sac_step (host batch, caller's noise) and sac_step_device (device batch) drive the same two launches."""
    O, A, B = 42, 7, 256
    fused, plain = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    obs, act, rew, term, nobs = synth_transitions(B, O, A, seed=5, term_frac=0.1)
    rs = np.random.RandomState(1)
    batch = dict(observations=obs, actions=act, rewards=rew, terminals=term.astype(np.float32), next_observations=nobs)
    for s in range(6):
        eps = (rs.normal(size=(B, A)).astype(np.float32), rs.normal(size=(B, A)).astype(np.float32)) if s % 2 else None
        da, db = fused.train(batch, eps=eps), plain.train(batch, eps=eps)
        assert np.array_equal(da, db), s
    bufs = [_buffer(3000, O, A, 1), _buffer(3000, O, A, 1)]
    for b in bufs:
        b.seed(7)
    for _ in range(20):
        fused.train(bufs[0].random_batch(B))
        plain.train(bufs[1].random_batch(B))
    fused._lib.sac_sync(fused._h)
    _same(fused.state_dict(), plain.state_dict())


def test_two_fused_trainers_interleaved_from_one_thread():
    """Fused launches of different trainers are serialised on the device (two half-resident grids would starve each
    other): interleaving them without any synchronisation in between must neither hang nor change a result."""
    O, A, B = 42, 7, 256
    a, ref = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    b, _ = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    bufs = [_buffer(3000, O, A, 1) for _ in range(3)]
    for x in bufs:
        x.seed(7)
    for _ in range(30):
        a.train(bufs[0].random_batch(B))          # asynchronous: device batches, no host synchronisation
        b.train(bufs[1].random_batch(B))
    for _ in range(30):
        ref.train(bufs[2].random_batch(B))
    for t in (a, b, ref):
        t._lib.sac_sync(t._h)
    _same(a.state_dict(), ref.state_dict())
    _same(b.state_dict(), ref.state_dict())


def test_two_fused_trainers_side_by_side_on_disjoint_halves_of_the_chip():
    """Batch 128 = 128 workgroups: a trainer confined to four XCDs (CU-masked stream, sac_trainer_set_xcd_mask) keeps the
    fused step, two such trainers run CONCURRENTLY from two host threads without the cross-trainer gate -- and each
    computes exactly what a run that has the chip to itself computes.  A share too small for the fused step (one XCD)
    switches the trainer to the four-launch step, same results."""
    import threading
    from robosuite_benchmark_amd import _lib
    O, A, B, steps = 42, 7, 128, 150
    a, ref = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    b, _ = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    c, _ = _pair_of_hip(O, A, B, seed=2, noise_seed=3)
    bufs = [_buffer(3000, O, A, 1) for _ in range(4)]
    for x in bufs:
        x.seed(7)
    for tr, buf, mask in ((a, bufs[0], 0x0f), (b, bufs[1], 0xf0)):
        _lib.check(tr._lib.sac_trainer_set_xcd_mask(tr._h, mask), "sac_trainer_set_xcd_mask")
        _lib.check(buf._lib.sac_buffer_set_xcd_mask(buf._h, mask), "sac_buffer_set_xcd_mask")
        assert tr.is_fused()
    _lib.check(c._lib.sac_trainer_set_xcd_mask(c._h, 0x04), "sac_trainer_set_xcd_mask")
    assert not c.is_fused()
    errs = []

    def run(tr, buf):
        try:
            tr.train_loop(buf, steps, batch_size=B)
        except Exception as e:                       # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(a, bufs[0])), threading.Thread(target=run, args=(b, bufs[1]))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    c.train_loop(bufs[2], steps, batch_size=B)
    ref.train_loop(bufs[3], steps, batch_size=B)
    assert a.is_fused() and b.is_fused()             # nobody gave up
    for t in (a, b, c):
        _same(t.state_dict(), ref.state_dict())
    _lib.check(a._lib.sac_trainer_set_xcd_mask(a._h, 0xff), "sac_trainer_set_xcd_mask")      # back to the whole chip
    assert a.is_fused()
    a.train_loop(bufs[0], 5, batch_size=B)
    ref.train_loop(bufs[3], 5, batch_size=B)
    _same(a.state_dict(), ref.state_dict())


def test_a_lost_producer_falls_back_transparently_inside_the_loop_call():
    """Launch 3 of the loop loses one producer workgroup (test hook): its consumers give up after the hand-off timeout,
    launch D of that step and of every later one applies nothing.  The call notices, puts the trainer on the four-launch
    step, takes the generator back to the first unapplied step and runs the rest of the loop again: it RETURNS NORMALLY,
    and state, diagnostics, per-step trace and the buffer's generator equal an undisturbed four-launch run bit for bit."""
    O, A, B = 42, 7, 256
    fused, plain = _pair_of_hip(O, A, B, seed=4, noise_seed=9, SAC_FUSED_TEST_STALL=3)
    bufs = [_buffer(4000, O, A, 8), _buffer(4000, O, A, 8)]
    for b in bufs:
        b.seed(31)
    assert fused.is_fused()
    fa, la = fused.train_loop(bufs[0], 10, batch_size=B)
    assert not fused.is_fused()
    fb, lb = plain.train_loop(bufs[1], 10, batch_size=B)
    assert np.array_equal(fa, fb) and np.array_equal(la, lb)
    _same(fused.state_dict(), plain.state_dict())
    assert fused.state_dict()["scalars"][3] == 10 and fused.state_dict()["scalars"][4] == 10
    ka, pa = bufs[0].rng_state()
    kb, pb = bufs[1].rng_state()
    assert np.array_equal(ka, kb) and pa == pb          # the index stream went on exactly where an undisturbed run's would
    ref = np.random.RandomState(31)
    for _ in range(10):
        ref.randint(0, 4000, B)
    assert np.array_equal(ka, ref.get_state()[1]) and pa == ref.get_state()[2]
    # both continue on the four-launch step
    _, la = fused.train_loop(bufs[0], 15, batch_size=B)
    _, lb = plain.train_loop(bufs[1], 15, batch_size=B)
    assert np.array_equal(la, lb) and np.all(np.isfinite(la))
    _same(fused.state_dict(), plain.state_dict())


def test_a_lost_producer_inside_a_call_that_started_on_a_speculative_chunk():
    """Loop calls in a row: the third call's first chunk was drawn and gathered by the second (speculation, the generator's
    mirror still in front of it).  Launch 12 -- the second step of that third call -- loses a producer: the replay must start
    from the state in front of the speculative chunk's batches."""
    O, A, B = 42, 7, 256
    fused, plain = _pair_of_hip(O, A, B, seed=4, noise_seed=9, SAC_FUSED_TEST_STALL=12)
    bufs = [_buffer(4000, O, A, 8), _buffer(4000, O, A, 8)]
    for b in bufs:
        b.seed(31)
    for n in (5, 5, 10, 7):
        _, la = fused.train_loop(bufs[0], n, batch_size=B)
        _, lb = plain.train_loop(bufs[1], n, batch_size=B)
        assert np.array_equal(la, lb), n
        (ka, pa), (kb, pb) = bufs[0].rng_state(), bufs[1].rng_state()
        assert pa == pb and np.array_equal(ka, kb), n
    assert not fused.is_fused()
    _same(fused.state_dict(), plain.state_dict())


@pytest.mark.parametrize("stall_at,steps", [(1, 5), (40, 100), (57, 60)])
def test_a_lost_producer_falls_back_transparently_on_the_stepwise_interface(stall_at, steps):
    """The reference's own loop -- random_batch(); train() on device batches, nobody waiting for a step -- with launch
    `stall_at` losing a producer: the steps launched behind it (the host runs ahead of the device) apply nothing; the
    host notices at a later call, falls back and re-runs them from their slots.  No exception, same trajectory."""
    O, A, B = 42, 7, 128
    fused, plain = _pair_of_hip(O, A, B, seed=5, noise_seed=3, SAC_FUSED_TEST_STALL=stall_at)
    bufs = [_buffer(3000, O, A, 2), _buffer(3000, O, A, 2)]
    for b in bufs:
        b.seed(8)
    for tr, b in ((fused, bufs[0]), (plain, bufs[1])):
        for _ in range(steps):
            tr.train(b.random_batch(B))
        _lib_sync(tr)
    assert not fused.is_fused()
    _same(fused.state_dict(), plain.state_dict())
    assert fused.state_dict()["scalars"][4] == steps


def test_the_epoch_driver_survives_a_fused_step_giving_up(tmp_path):
    """driver.experiment died mid-epoch when the fused step gave up.  Now the run completes and logs the
    same rows as a run on the four-launch step."""
    import json
    from robosuite_benchmark_amd import variant
    from robosuite_benchmark_amd.driver import experiment
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    v = variant.load_variant(os.path.join(root, "tests", "golden", "Lift-Panda-OSC-POSE-SEED17.variant.json"))
    v["algorithm_kwargs"].update(num_trains_per_train_loop=60, num_expl_steps_per_train_loop=300,
                                 min_num_steps_before_training=400, num_eval_steps_per_epoch=200,
                                 expl_max_path_length=100, eval_max_path_length=100)
    old = {k: os.environ.get(k) for k in ("SAC_FUSED", "SAC_FUSED_TEST_STALL")}
    try:
        os.environ.pop("SAC_FUSED", None)
        os.environ["SAC_FUSED_TEST_STALL"] = "75"                    # inside the second epoch's training block
        rows_a = experiment(v, seed=3, num_epochs=3, quiet=True)
        os.environ.pop("SAC_FUSED_TEST_STALL", None)
        os.environ["SAC_FUSED"] = "0"
        rows_b = experiment(v, seed=3, num_epochs=3, quiet=True)
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
    for ra, rb in zip(rows_a, rows_b):
        for k in ra:
            if not k.startswith("time/"):
                assert ra[k] == rb[k], k


def _lib_sync(tr):
    from robosuite_benchmark_amd import _lib
    _lib.check(tr._lib.sac_sync(tr._h), "sac_sync")

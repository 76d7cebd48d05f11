"""GPU: the weight-gradient launch's job shapes.  A first layer of 65-112 input columns is ONE job per 16-row tile of dW
(five to seven 16-column tiles; round 2: a 64-column strip + a tail strip = two jobs, which put the two-arm tasks and Stack
at 288-304 jobs for 256 CUs).  Per tile the arithmetic is unchanged, so a run must equal the two-strip table bit for bit."""
import os

import numpy as np
import pytest

from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu


def _trainer(O, A, B, merged):
    old = os.environ.get("SAC_DW_NO_MERGE")
    try:
        if merged:
            os.environ.pop("SAC_DW_NO_MERGE", None)
        else:
            os.environ["SAC_DW_NO_MERGE"] = "1"
        return make_pair(O, A, B, seed=6, noise_seed=2)[1]
    finally:
        if old is None:
            os.environ.pop("SAC_DW_NO_MERGE", None)
        else:
            os.environ["SAC_DW_NO_MERGE"] = old


def _buffer(n, O, A, seed):
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed, term_frac=0.05)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    buf.add_block(obs, act, rew, nobs, term)
    return buf


# Stack 55/7 (policy 64: untouched, Q 80: five tiles), TwoArmPegInHole 73/12 (80 / 96: five / six), TwoArmHandoff 86/14 and
# TwoArmLift 89/14 (96 / 112: six / seven), 100/4 (112 / 128: seven / two full strips), an odd batch, batch 1024 (chain)
@pytest.mark.parametrize("O,A,B", [(55, 7, 256), (73, 12, 256), (86, 14, 256), (89, 14, 128), (100, 4, 256), (89, 14, 250),
                                   (73, 12, 1024)])
def test_whole_row_jobs_equal_strip_jobs_bitwise(O, A, B):
    a, b = _trainer(O, A, B, True), _trainer(O, A, B, False)
    bufs = [_buffer(4000, O, A, 3), _buffer(4000, O, A, 3)]
    for x in bufs:
        x.seed(12)
    steps = 12                                       # (past two Polyak updates: target_update_period 5)
    _, la = a.train_loop(bufs[0], steps, batch_size=B)
    _, lb = b.train_loop(bufs[1], steps, batch_size=B)
    assert np.array_equal(la, lb)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k
    for k in sa["opt"]:
        assert np.array_equal(sa["opt"][k][0], sb["opt"][k][0]) and np.array_equal(sa["opt"][k][1], sb["opt"][k][1]), k
    for name in ("g_policy", "g_qf1", "g_qf2"):
        n = sa["params"][name[2:]].size
        assert np.array_equal(a.debug_fetch(name, n), b.debug_fetch(name, n)), name

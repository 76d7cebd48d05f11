"""GPU: checkpoint / resume.  The reference's snapshots cannot resume a run
(networks only, /root/reference/util/rlkit_custom.py:68-82); here a resumed run must continue bit for bit."""
import json
import os

import numpy as np
import pytest

from robosuite_benchmark_amd import EnvReplayBuffer, checkpoint as ck
from robosuite_benchmark_amd.driver import experiment
from robosuite_benchmark_amd.variant import default_variant
from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu


def _same_state(a, b):
    for k in a["params"]:
        assert np.array_equal(a["params"][k], b["params"][k]), k
    for k in a["opt"]:
        assert np.array_equal(a["opt"][k][0], b["opt"][k][0]) and np.array_equal(a["opt"][k][1], b["opt"][k][1]), k
    assert np.array_equal(a["scalars"], b["scalars"])


def test_buffer_rows_cursor_and_stream_round_trip():
    O, A, cap = 11, 3, 1000
    obs, act, rew, term, nobs = synth_transitions(1300, O, A, seed=5, term_frac=0.1)
    buf = EnvReplayBuffer(cap, obs_dim=O, action_dim=A)
    buf.add_block(obs, act, rew, nobs, term)              # wraps: top = 300, size = 1000
    buf.seed(9)
    buf.sample_indices(64)
    st = buf.state_dict()
    assert (st["top"], st["size"]) == (300, 1000)
    # storage order: rows 0..299 hold samples 1000..1299, rows 300..999 samples 300..999
    np.testing.assert_array_equal(st["observations"][:300], obs[1000:])
    np.testing.assert_array_equal(st["next_observations"][300:], nobs[300:1000])
    np.testing.assert_array_equal(st["terminals"].ravel()[:300], (term[1000:].ravel() != 0).astype(np.uint8))
    clone = EnvReplayBuffer(cap, obs_dim=O, action_dim=A)
    clone.load_state_dict(st)
    st2 = clone.state_dict()
    for k, v in st.items():
        assert np.array_equal(np.asarray(v), np.asarray(st2[k])), k
    assert np.array_equal(buf.sample_indices(256), clone.sample_indices(256))
    with pytest.raises(RuntimeError, match="outside the buffer"):
        buf.read_rows(900, 200)
    other = EnvReplayBuffer(cap, obs_dim=O + 1, action_dim=A)
    with pytest.raises(ValueError, match="another shape"):
        other.load_state_dict(st)


def test_resume_continues_bit_for_bit(tmp_path):
    O, A, B, n = 42, 7, 256, 20_000
    data = synth_transitions(n, O, A, seed=8)

    def fresh():
        _, tr = make_pair(O, A, B, seed=4, noise_seed=123)
        buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
        buf.add_block(data[0], data[1], data[2], data[4], data[3])
        buf.seed(17)
        return tr, buf

    tr_a, buf_a = fresh()                                   # uninterrupted: 300 + 300 steps
    tr_a.train_loop(buf_a, 300, batch_size=B)
    _, last_a = tr_a.train_loop(buf_a, 300, batch_size=B)

    tr_b, buf_b = fresh()                                   # interrupted after 300
    tr_b.train_loop(buf_b, 300, batch_size=B)
    man = ck.save_checkpoint(str(tmp_path / "ck"), tr_b, buf_b, extra=dict(epoch=0))
    assert man["buffer"]["size"] == n and "params.target_qf2" in man["arrays"]
    del tr_b, buf_b

    _, tr_c = make_pair(O, A, B, seed=99, noise_seed=123)   # different init: everything must come from the file
    buf_c = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    assert ck.load_checkpoint(str(tmp_path / "ck"), tr_c, buf_c) == dict(epoch=0)
    _, last_c = tr_c.train_loop(buf_c, 300, batch_size=B)

    assert np.array_equal(last_a, last_c)
    _same_state(tr_a.state_dict(), tr_c.state_dict())
    ka, pa = buf_a.rng_state()
    kc, pc = buf_c.rng_state()
    assert pa == pc and np.array_equal(ka, kc)
    # torn / foreign checkpoints are refused
    cur = ck.current_dir(str(tmp_path / "ck"))
    bad = json.load(open(os.path.join(cur, "manifest.json")))
    bad["format"] = "something else"
    os.makedirs(tmp_path / "bad")
    json.dump(bad, open(tmp_path / "bad" / "manifest.json", "w"))
    with pytest.raises(ValueError, match="not a"):
        ck.load_checkpoint(str(tmp_path / "bad"), tr_c, buf_c)
    # an array swapped for another save's (same dtype and shape: only the content hash can tell) is refused, and the
    # live state is untouched by the failed load
    before = tr_c.state_dict()
    a = np.load(os.path.join(cur, "adam_m.qf1.npy"))
    np.save(os.path.join(cur, "adam_m.qf1.npy"), a + 1.0)
    with pytest.raises(ValueError, match="content does not match"):
        ck.load_checkpoint(str(tmp_path / "ck"), tr_c, buf_c)
    _same_state(before, tr_c.state_dict())


def test_exported_state_dicts_hold_the_device_weights(tmp_path):
    import torch
    O, A, B = 46, 7, 64
    _, tr = make_pair(O, A, B, seed=2)
    path = str(tmp_path / "nets.pt")
    ck.export_torch_state_dicts(path, tr)
    sd = torch.load(path, weights_only=True)
    flat = tr.state_dict()["params"]
    for net in ck.NETS:
        got = np.concatenate([v.numpy().ravel() for v in sd[net].values()])
        assert np.array_equal(got, flat[net]), net
    back = ck.read_rlkit_zip_params(path, O, A)
    assert np.array_equal(back["qf2"], flat["qf2"])


def test_driver_resume_reproduces_the_uninterrupted_run(tmp_path):
    v = default_variant(env="Lift", seed=3, batch_size=128)
    v["algorithm_kwargs"].update(num_epochs=4, num_trains_per_train_loop=40, num_expl_steps_per_train_loop=100,
                                 num_eval_steps_per_epoch=100, min_num_steps_before_training=200,
                                 expl_max_path_length=50, eval_max_path_length=50)
    v["replay_buffer_size"] = 5000
    straight = experiment(v, seed=3, quiet=True)
    ckd = str(tmp_path / "ck")
    experiment(v, seed=3, quiet=True, num_epochs=2, checkpoint_dir=ckd, log_dir=str(tmp_path / "run"))
    resumed = experiment(v, seed=3, quiet=True, checkpoint_dir=ckd, resume=True, log_dir=str(tmp_path / "run"))
    assert [r["Epoch"] for r in resumed] == [2, 3]
    for want, got in zip(straight[2:], resumed):
        for k in want:
            if not k.startswith("time/"):
                assert want[k] == got[k], k
    lines = open(tmp_path / "run" / "progress.csv").read().strip().split("\n")
    assert len(lines) == 1 + 4                              # one header, epochs 0..3 appended across the resume

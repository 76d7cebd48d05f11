"""Checkpoint interop without a GPU: the torch-zip exporter and the raw-storage reader of the reference's
zip-format params.pkl (no unpickling) are inverses; layer naming/order follows rlkit's Mlp registration."""
import glob
import os
import zipfile

import numpy as np
import pytest
import torch

from robosuite_benchmark_amd import checkpoint as ck

HERE = os.path.dirname(os.path.abspath(__file__))


def _random_params(O, A, seed=3):
    rs = np.random.RandomState(seed)
    out = {}
    for net, layers in ck.rlkit_layer_shapes(O, A).items():
        out[net] = np.concatenate([rs.standard_normal(int(np.prod(s))).astype(np.float32) for _, s in layers])
    out["target_qf1"], out["target_qf2"] = out["qf1"] + 1.0, out["qf2"] - 1.0
    return out


def test_layer_shapes_follow_rlkit_registration_order():
    sh = ck.rlkit_layer_shapes(42, 7)
    assert [n for n, _ in sh["policy"]] == ["fc0.weight", "fc0.bias", "fc1.weight", "fc1.bias", "last_fc.weight",
                                            "last_fc.bias", "last_fc_log_std.weight", "last_fc_log_std.bias"]
    assert dict(sh["policy"])["fc0.weight"] == (256, 42) and dict(sh["qf1"])["fc0.weight"] == (256, 49)
    assert dict(sh["qf2"])["last_fc.weight"] == (1, 256)
    # parameter counts of SURVEY.md section 8a: policy 80 398, each Q 78 849
    assert sum(int(np.prod(s)) for _, s in sh["policy"]) == 80398
    assert sum(int(np.prod(s)) for _, s in sh["qf1"]) == 78849


def test_export_then_raw_read_round_trip(tmp_path):
    O, A = 46, 7
    params = _random_params(O, A)
    path = str(tmp_path / "nets.pt")
    ck.export_torch_state_dicts(path, params, O, A)
    assert zipfile.is_zipfile(path)
    sd = torch.load(path, weights_only=True)                    # a plain dict of tensors: safe loader accepts it
    assert list(sd.keys()) == list(ck.NETS)
    assert tuple(sd["policy"]["last_fc_log_std.weight"].shape) == (A, 256)
    assert tuple(sd["target_qf1"]["fc0.weight"].shape) == (256, O + A)
    back = ck.read_rlkit_zip_params(path, O, A)                 # storages 0..19 = policy, qf1, qf2 in order
    for net in ("policy", "qf1", "qf2"):
        np.testing.assert_array_equal(back[net], params[net])


def test_raw_read_rejects_wrong_dims_and_legacy_files(tmp_path):
    params = _random_params(42, 7)
    path = str(tmp_path / "nets.pt")
    ck.export_torch_state_dicts(path, params, 42, 7)
    with pytest.raises(ValueError, match="storage 0"):
        ck.read_rlkit_zip_params(path, 46, 7)
    legacy = tmp_path / "legacy.pkl"
    legacy.write_bytes(b"\\x80\\x02\\x8a\\nl\\xfc\\x9cF\\xf9 j\\xa8P\\x19.")     # magic of the pre-zip torch format
    with pytest.raises(ValueError, match="legacy"):
        ck.read_rlkit_zip_params(str(legacy), 42, 7)


@pytest.mark.skipif(not os.path.isdir("/root/reference/log/runs"), reason="reference tree not mounted (GPU box)")
def test_raw_read_of_a_shipped_snapshot_matches_the_committed_fixture():
    src = glob.glob("/root/reference/log/runs/Lift-Panda-OSC-POSE-SEED129/*/params.pkl")[0]
    got = ck.read_rlkit_zip_params(src, 42, 7)
    want = np.load(os.path.join(HERE, "golden", "trained_weights_lift_seed129.npz"))
    for net in ("policy", "qf1", "qf2"):
        np.testing.assert_array_equal(got[net], want[net])


# ---- save/load atomicity (no GPU: a stand-in with the trainer's checkpoint surface) -----------------------------
class _StubTrainer:
    NETS = {"policy": 0, "qf1": 1, "qf2": 2, "target_qf1": 3, "target_qf2": 4}
    obs_dim, act_dim, _batch, _num_train_steps = 5, 2, 8, 0
    discount = reward_scale = policy_lr = qf_lr = soft_target_tau = 0.5
    target_update_period, use_automatic_entropy_tuning, target_entropy, noise_seed = 1, True, -2.0, 0

    def __init__(self, fill):
        self._h = object()
        self.fill(fill)

    def fill(self, v):
        self.st = dict(params={k: np.full(32, v + i, np.float32) for i, k in enumerate(self.NETS)},
                       opt={k: (np.full(32, v, np.float32), np.full(32, -v, np.float32)) for k in ("policy", "qf1", "qf2")},
                       scalars=np.full(6, v, np.float64))

    def state_dict(self):
        return self.st

    def load_state_dict(self, st):
        self.st = st


def test_a_killed_save_leaves_the_previous_checkpoint_intact(tmp_path, monkeypatch):
    d = str(tmp_path / "ck")
    tr = _StubTrainer(1.0)
    ck.save_checkpoint(d, tr, extra=dict(epoch=0))
    tr.fill(2.0)
    real_save, calls = ck._save, []

    def dying_save(dirname, man, name, arr):
        calls.append(name)
        if len(calls) == 4:
            raise KeyboardInterrupt("killed half-way through the second save")
        real_save(dirname, man, name, arr)

    monkeypatch.setattr(ck, "_save", dying_save)
    with pytest.raises(KeyboardInterrupt):
        ck.save_checkpoint(d, tr, extra=dict(epoch=1))
    monkeypatch.setattr(ck, "_save", real_save)
    back = _StubTrainer(9.0)
    assert ck.load_checkpoint(d, back) == dict(epoch=0)          # the old generation, whole
    assert float(back.st["params"]["policy"][0]) == 1.0 and float(back.st["opt"]["qf2"][1][0]) == -1.0
    ck.save_checkpoint(d, tr, extra=dict(epoch=1))               # the next complete save cleans the torn one up
    assert ck.load_checkpoint(d, back) == dict(epoch=1) and float(back.st["params"]["policy"][0]) == 2.0
    assert sorted(x for x in os.listdir(d) if x.startswith("gen-")) == [os.path.basename(ck.current_dir(d))]


def test_truncated_or_swapped_arrays_are_refused(tmp_path):
    d = str(tmp_path / "ck")
    ck.save_checkpoint(d, _StubTrainer(1.0), extra=dict(epoch=0))
    cur = ck.current_dir(d)
    back = _StubTrainer(9.0)
    f = os.path.join(cur, "adam_v.qf1.npy")
    blob = open(f, "rb").read()
    open(f, "wb").write(blob[:len(blob) // 2])                   # a save that stopped half-way through a file
    with pytest.raises(Exception):
        ck.load_checkpoint(d, back)
    np.save(f, np.full(32, 123.0, np.float32))                   # right dtype and shape, other content
    with pytest.raises(ValueError, match="content does not match"):
        ck.load_checkpoint(d, back)
    assert float(back.st["params"]["policy"][0]) == 9.0          # nothing was loaded
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(str(tmp_path / "nothing"), back)


def test_a_dangling_latest_pointer_is_not_no_checkpoint(tmp_path):
    """`latest` naming a generation without a manifest made checkpoint_exists() False -- a resuming run
    started from scratch and its first save deleted what was left.  Now: the newest other complete generation is used,
    then the flat layout of the first format revision; with neither, the damaged directory is an error, not 'nothing'."""
    import shutil
    d = str(tmp_path / "ck")
    tr = _StubTrainer(1.0)
    ck.save_checkpoint(d, tr, extra=dict(epoch=0))
    good = ck.current_dir(d)
    with open(os.path.join(d, "latest"), "w") as f:
        f.write("gen-77\n")                                      # names a generation that is not there
    assert ck.checkpoint_exists(d) and ck.current_dir(d) == good
    back = _StubTrainer(9.0)
    assert ck.load_checkpoint(d, back) == dict(epoch=0) and float(back.st["params"]["qf1"][0]) == 2.0
    # a legacy flat manifest beside a stale pointer
    flat = str(tmp_path / "flat")
    os.makedirs(flat)
    for name in os.listdir(good):
        shutil.copy(os.path.join(good, name), os.path.join(flat, name))
    with open(os.path.join(flat, "latest"), "w") as f:
        f.write("gen-3\n")
    assert ck.current_dir(flat) == flat
    assert ck.load_checkpoint(flat, _StubTrainer(5.0)) == dict(epoch=0)
    # nothing complete left at all: refuse loudly
    os.remove(os.path.join(good, "manifest.json"))
    with pytest.raises(FileNotFoundError, match="refusing"):
        ck.checkpoint_exists(d)
    assert ck.checkpoint_exists(str(tmp_path / "never_written")) is False

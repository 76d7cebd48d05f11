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

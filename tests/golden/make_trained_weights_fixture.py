"""Trained fp32 weights of a shipped run as a realistic-weights fixture.

Reads the raw little-endian tensor blobs `params/data/0..19` inside the ZIP-format checkpoint
/root/reference/log/runs/Lift-Panda-OSC-POSE-SEED129/*/params.pkl with `zipfile` -- nothing is
unpickled, nothing from the file is executed.  Blob order (pinned by sizes):
0-7 policy (fc0.W 256x42, fc0.b, fc1.W, fc1.b, last_fc.W 7x256, last_fc.b, last_fc_log_std.W, .b),
8-13 qf1 (fc0.W 256x49, fc0.b, fc1.W, fc1.b, last_fc.W 1x256, last_fc.b), 14-19 qf2.
The snapshot aliases target_qf* to qf* (only 20 storages), so the fixture holds three nets.
Run (needs /root/reference):  python tests/golden/make_trained_weights_fixture.py"""
import glob
import os
import zipfile

import numpy as np

src = glob.glob("/root/reference/log/runs/Lift-Panda-OSC-POSE-SEED129/*/params.pkl")[0]
z = zipfile.ZipFile(src)
shapes = {"policy": [(256, 42), (256,), (256, 256), (256,), (7, 256), (7,), (7, 256), (7,)],
          "qf1": [(256, 49), (256,), (256, 256), (256,), (1, 256), (1,)],
          "qf2": [(256, 49), (256,), (256, 256), (256,), (1, 256), (1,)]}
out, k = {}, 0
for net, shp in shapes.items():
    parts = []
    for s in shp:
        raw = z.read(f"params/data/{k}")
        a = np.frombuffer(raw, dtype="<f4")
        assert a.size == int(np.prod(s)), (net, k, a.size, s)
        parts.append(a)
        k += 1
    out[net] = np.concatenate(parts).astype(np.float32)      # flat nn.Linear layout: W, b per layer
assert k == 20
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trained_weights_lift_seed129.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {n: (v.size, float(np.abs(v).max())) for n, v in out.items()}, os.path.getsize(dst), "bytes")

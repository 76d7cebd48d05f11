"""Diagnostic: in-kernel wall-clock stamps (needs the -DSAC_STAMPS build: scratch/libsac_hip_stamps_<tag>.so).
usage: python scratch/stamps.py <tag> <kernel id 0..4> <n stamps> [first block] [last block]"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
import bench
kid, ns = int(sys.argv[2]), int(sys.argv[3])
tr, buf = bench.build_replica("Lift", 42, 7, 256, 100_000, 17, 0)
tr.train_loop(buf, 200, batch_size=256)
lib = _lib.load()
out = np.zeros(5 * 512 * 16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
st = out.reshape(5, 512, 16).astype(np.int64)
w = st[kid]
blocks = [b for b in range(512) if w[b, 0] > 0]
t0 = min(w[b, 0] for b in blocks)
for sel, name in ((lambda b: (b & 7) < 4, "blocks b%8<4"), (lambda b: (b & 7) >= 4, "blocks b%8>=4")):
    bl = [b for b in blocks if sel(b)]
    if not bl:
        continue
    ww = w[bl]
    med = [np.median(ww[:, i] - t0) / 100.0 for i in range(ns)]
    print(f"kernel {kid} {name} n={len(bl)}: " + " ".join(f"s{i}={med[i]:.2f}" for i in range(ns)),
          "| max end", (ww[:, ns - 1].max() - t0) / 100.0)
if kid == 4:
    ww = w[:245]
    order = [0, 3, 4, 1, 2]; names = ["start", "table", "operands", "mfma+reduce", "adam+stores"]
    t = [np.median(ww[:, i] - t0) / 100.0 for i in order]
    print("k_dw_adam", " ".join(f"{names[k]}={t[k]:.2f}" for k in range(5)), "| max end", (ww[:, 2].max() - t0) / 100.0)
if kid == 1:
    ww = w[[b for b in blocks]]
    f = (ww[:, 15] - ww[:, 14]) / ((ww[:, 4] - ww[:, 0]) / 100.0)   # shader clocks per us
    print("shader clock during k_fwd_b: median %.0f MHz (min %.0f, max %.0f)" % (np.median(f), f.min(), f.max()))

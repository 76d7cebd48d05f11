"""Diagnostic: in-kernel stamps of k_dw_adam's tile blocks (needs a -DSAC_STAMPS build: scratch/libsac_hip_stamps_<tag>.so).
usage: python scratch/stamps_dw.py <tag> [batch]"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
import bench
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tr, buf = bench.build_replica("Lift", 42, 7, B, 100_000, 17, 0)
tr.train_loop(buf, 201, batch_size=B)       # the last step is at an even loop position: stamp slot 4 (with the inner stamps)
lib = _lib.load()
out = np.zeros(5 * 512 * 16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
st = out.reshape(5, 512, 16).astype(np.int64)
w = st[4]
bl = [b for b in range(512) if w[b, 0] > 0]
t0 = min(w[b, 0] for b in bl)
ww = w[bl]
for i, nm in ((0, "start"), (3, "table read"), (4, "operands arrived"), (1, "reduced (barrier)"), (2, "end")):
    v = (ww[:, i] - t0) / 100.0
    print(f"{nm:>18s}: median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
d3 = st[3]
b3 = [b for b in range(512) if d3[b, 0] > 0]
if b3:
    print("diagnostics block: start %.2f end %.2f" % ((d3[b3[0], 0] - t0) / 100.0, (d3[b3[0], 1] - t0) / 100.0))

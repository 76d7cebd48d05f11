"""Diagnostic: in-kernel stamps of k_chain (a -DSAC_STAMPS build).  usage: python scratch/stamps_chain.py <tag> <O> <A> <B>"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
import bench
O, A, B = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tr, buf = bench.build_replica("X", O, A, B, 100_000, 17, 0)
tr.train_loop(buf, 100, batch_size=B)
lib = _lib.load()
out = np.zeros(5 * 512 * 16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
w = out.reshape(5, 512, 16).astype(np.int64)[0]
blocks = [b for b in range(512) if w[b, 0] > 0]
t0 = min(w[b, 0] for b in blocks)
names = {0: "start", 1: "rows", 2: "L0 done", 3: "L1 done", 4: "head gemm", 5: "head math", 6: "Q L0 done", 7: "Q L1 done", 8: "2nd L0", 9: "2nd L1",
         10: "tail gemm", 12: "end"}
for item, nm in ((0, "P0"), (1, "P1"), (2, "N"), (3, "C")):
    bl = [b for b in blocks if ((b & 7) >> 1) == item]
    ww = w[bl]
    cols = [i for i in range(13) if (ww[:, i] > 0).all() and i in names]
    print(f"item {nm} (n={len(bl)}):", "  ".join(f"{names[i]}={np.median(ww[:, i] - t0) / 100.0:.2f}" for i in cols), "| last end", (ww[:, 12].max() - t0) / 100.0)

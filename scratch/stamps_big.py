"""Diagnostic: in-kernel stamps of the four-launch step kernels at a large batch (a -DSAC_STAMPS build).
usage: python scratch/stamps_big.py <tag> <O> <A> <B>    (env: SAC_RB2 / SAC_FORCE_SP as for any run)"""
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
st = out.reshape(5, 512, 16).astype(np.int64)
names = {0: ["start", "rows committed", "first layer done", "slice GEMM done", "end", "chunk0", "chunk1", "chunk2", "chunk3"],
         1: ["start", "rows committed", "head done", "first layer + barrier", "q partial (fwd end)", "tail: dq/dh2", "tail: dq/dh1", "end",
             "kernarg", "requests issued", "head inputs here"]}
for kid in (0, 1):
    w = st[kid]
    blocks = [b for b in range(512) if w[b, 0] > 0]
    if not blocks:
        continue
    t0 = min(w[b, 0] for b in blocks)
    for sel, nm in ((lambda b: (b & 7) < 4, "b%8<4"), (lambda b: (b & 7) >= 4, "b%8>=4")):
        bl = [b for b in blocks if sel(b)]
        if not bl:
            continue
        ww = w[bl]
        print(f"kernel {kid} blocks {nm} (n={len(bl)}):", "  ".join(f"{names[kid][i]}={np.median(ww[:, i] - t0) / 100.0:.2f}"
              for i in range(len(names[kid])) if (ww[:, i] > 0).all()), "| last end", (ww[:, [4, 7][kid]].max() - t0) / 100.0, flush=True)

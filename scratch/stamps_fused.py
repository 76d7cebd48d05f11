"""Diagnostic: in-kernel wall-clock stamps of the fused step k_abc (needs the -DSAC_STAMPS build:
scratch/libsac_hip_stamps_<tag>.so).   usage: python scratch/stamps_fused.py <tag> [batch [obs_dim act_dim]]"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
import bench
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
O, A = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (42, 7)
tr, buf = bench.build_replica("X", O, A, B, 100_000, 17, 0)
tr.train_loop(buf, 200, batch_size=B)
lib = _lib.load()
out = np.zeros(5 * 512 * 16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
st = out.reshape(5, 512, 16).astype(np.int64)
w = st[0]
blocks = [b for b in range(512) if w[b, 0] > 0]
t0 = min(w[b, 0] for b in blocks)
names = ["start", "A done", "A published", "B wait over", "head done", "q partial out", "B done", "B published", "C wait over", "end",
         "D wait over", "D done"]
XOR = 4     # SAC_XR_XOR of the build: role index xr = (b % 8) ^ XOR  (0-3 critic chain, 4-5 policy on s, 6-7 policy on s')
for sel, name in ((lambda b: ((b & 7) ^ XOR) < 4, "critic chain"), (lambda b: ((b & 7) ^ XOR) in (4, 5), "policy chain s"),
                  (lambda b: ((b & 7) ^ XOR) in (6, 7), "policy chain s'")):
    bl = [b for b in blocks if sel(b)]
    ww = w[bl]
    ns = 12 if ww[:, 11].max() > 0 else 10
    med = [np.median(ww[:, i] - t0) / 100.0 for i in range(ns)]
    mx = [(ww[:, i].max() - t0) / 100.0 for i in range(ns)]
    print(f"{name} n={len(bl)}")
    for i in range(ns):
        print(f"   {names[i]:>14s}: median {med[i]:6.2f}  max {mx[i]:6.2f}")
w4 = st[4]
b4 = [b for b in range(512) if w4[b, 0] > 0]
if b4:
    t4 = min(w4[b, 0] for b in b4)
    print("k_dw_adam: start spread", (max(w4[b, 0] for b in b4) - t4) / 100.0, "end max", (w4[b4][:, 2].max() - t4) / 100.0,
          "| gap abc end -> dw start", (t4 - max(w[b, 9] for b in blocks)) / 100.0)
w3 = st[3]
b3 = [b for b in range(512) if w3[b, 0] > 0]
if b3 and b4:
    print("diagnostics block: starts", (w3[b3[0], 0] - t4) / 100.0, "after the first dW block, ends", (w3[b3[0], 1] - t4) / 100.0)
# k_dw_adam alternates between stamp slots 4 (even loop positions) and 2 (odd): the last step of the 200-step loop is odd
wl, wp = st[2], st[4]          # last step's D, the one before
bl = [b for b in range(512) if wl[b, 0] > 0]; bp = [b for b in range(512) if wp[b, 0] > 0]
if bl and bp:
    print("previous k_dw_adam: last tile block ends %.2f us before this k_abc's first block starts; its first block started %.2f us before"
          % ((t0 - wp[bp][:, 2].max()) / 100.0, (t0 - min(wp[b, 0] for b in bp)) / 100.0))
    print("this step: k_abc first start -> k_dw_adam first start %.2f us; k_dw_adam tiles end %.2f us after its first start"
          % ((min(wl[b, 0] for b in bl) - t0) / 100.0, (wl[bl][:, 2].max() - min(wl[b, 0] for b in bl)) / 100.0))
w1 = st[1]
b1 = [b for b in blocks if ((b & 7) ^ XOR) < 4 and w1[b, 0] > 0]
if b1:
    print("critic phase C (median, after C wait over): dq ready %.2f  dL/dh2 in LDS %.2f  slice GEMM done %.2f  end %.2f"
          % (tuple(np.median(w1[b1][:, i] - w[b1][:, 8]) / 100.0 for i in range(3)) + (np.median(w[b1][:, 9] - w[b1][:, 8]) / 100.0,)))
if b3:
    dd = st[3][b3[0]]
    print("diagnostics block: alpha %.2f  loads %.2f  accumulated+shuffled %.2f  after barrier %.2f  end %.2f (us after its start)"
          % tuple((dd[i] - dd[0]) / 100.0 for i in (2, 3, 4, 5, 1)))
# who is slow?  critic chain "B done" (stamp 6) and "A done" (stamp 1) by blockIdx % 8, by column part and by row-block
import collections
for st_i, nm in ((1, "A done"), (6, "B done"), (9, "end")):
    for key, f in (("b%8", lambda b: b & 7), ("part", lambda b: (2 * (b >> 3) + (b & 1)) & 3), ("rb", lambda b: (2 * (b >> 3) + (b & 1)) >> 2)):
        g = collections.defaultdict(list)
        for b in blocks:
            if ((b & 7) ^ XOR) < 4:
                g[f(b)].append((w[b, st_i] - t0) / 100.0)
        print(f"critic chain {nm:7s} by {key:5s}:", " ".join(f"{k}:{np.median(v):.2f}" for k, v in sorted(g.items())))

# the critic chain by (twin, column part): which block of a row-block's eight is the straggler, and from which stamp on
g = collections.defaultdict(list)
for b in blocks:
    xr = (b & 7) ^ XOR
    if xr < 4:
        g[((xr >> 1) & 1, (2 * (b >> 3) + (b & 1)) & 3)].append((w[b, :10] - t0) / 100.0)
print("critic chain by (twin, part): " + "  ".join(names[i] for i in (1, 3, 4, 5, 6, 7, 8, 9)))
for k, v in sorted(g.items()):
    v = np.median(np.array(v), axis=0)
    print("   ", k, " ".join(f"{v[i]:6.2f}" for i in (1, 3, 4, 5, 6, 7, 8, 9)))

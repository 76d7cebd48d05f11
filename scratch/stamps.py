import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 100_000, 17, 0)
tr.train_loop(buf, 200, batch_size=256)
lib = _lib.load()
out = np.zeros(5*512*16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
st = out.reshape(5, 512, 16).astype(np.int64)
for kid, nb in ((0,96),(3,16)):
    w = st[kid, :nb, :8]; c = st[kid, :nb, 8:]
    if w[:,0].max() == 0: continue
    t0 = w[:,0].min()
    print("kernel", kid, "blocks", nb, "start spread (us)", (w[:,0].max()-t0)/100.0)
    nz = [i for i in range(8) if w[:,i].max() > 0]
    for i in nz[1:]:
        dw = (w[:,i]-w[:,i-1])/100.0; dc = (c[:,i]-c[:,i-1])
        print(f"  seg {i-1}->{i}: wall us median {np.median(dw):.2f} max {dw.max():.2f} | cycles median {np.median(dc):.0f} | MHz {np.median(dc/np.maximum(dw,0.01)):.0f}")
    print("  total per block us: median", np.median((w[:,nz[-1]]-w[:,0])/100.0), " last end - first start:", (w[:,nz[-1]].max()-t0)/100.0)
    if kid == 0:
        for p in range(6):
            ww = w[16*p:16*p+16]
            print("   pass", p, "start", np.round((ww[:,0].min()-t0)/100.0,2), "segs", [float(np.round(np.median((ww[:,i]-ww[:,i-1])/100.0),2)) if ww[:,i].max()>0 and ww[:,i-1].max()>0 else None for i in range(1,8)], "end", np.round((ww[:,7].max()-t0)/100.0,2))

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
for kid, nb, name in ((0,256,"k_fwd_a"),(1,256,"k_fwd_b"),(4,245,"k_dw_adam")):
    w = st[kid,:nb]
    n = max(i for i in range(16) if w[:,i].max() > 0) + 1
    t0 = w[:,0].min()
    segs = [np.median(w[:,i]-w[:,i-1])/100.0 for i in range(1,n)]
    print(name, "start spread", (w[:,0].max()-t0)/100.0, "segs", [round(float(x),2) for x in segs], "block median", np.median(w[:,n-1]-w[:,0])/100.0, "first->last", (w[:,n-1].max()-t0)/100.0)
    if kid == 0:
        for lo,hi,nm in ((0,128,"pi"),(128,256,"critic")):
            ww=w[lo:hi]; print("   ", nm, [round(float(np.median(ww[:,i]-ww[:,i-1])/100.0),2) for i in range(1,n)])

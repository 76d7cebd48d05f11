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
w = st[4,:245]
order=[0,3,4,1,2]; names=["start","table","operands","mfma+reduce","adam+stores"]
t=[np.median(w[:,i]) for i in order]
print("k_dw_adam", " ".join(f"{names[k+1]}={(t[k+1]-t[k])/100.0:.2f}" for k in range(4)), "total", (t[-1]-t[0])/100.0)

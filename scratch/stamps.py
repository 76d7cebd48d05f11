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
order = [0,1,8,9,2,3,10,11,12,13,14,4,5,6,7]
names = ["start","prologue+sync","piG0","epi0","issueQ1+sync","piG1","epi1","issueQ2+sync","headG","reduce","commit+elem","tail(ticket)+sync","QG0+epi+sync","QG1","end"]
w = st[0,:96]
for p_ in range(2,6):
    ww = w[16*p_:16*p_+16]
    t = [np.median(ww[:,i]) for i in order]
    print("pass", p_, " ".join(f"{names[k+1]}={(t[k+1]-t[k])/100.0:.2f}" for k in range(len(order)-1)), " total", (t[-1]-t[0])/100.0)

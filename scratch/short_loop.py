"""Where does a short sac_train_loop call spend its time?  usage: python scratch/short_loop.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
tr.train_loop(buf, 300, batch_size=256)
for steps in (n, n, n, 100, 520, n, 1, 1, 2, 4, 5):
    tr._lib.sac_sync(tr._h)
    t0 = time.perf_counter()
    tr.train_loop(buf, steps, batch_size=256)
    t1 = time.perf_counter()
    d = tr.loop_timing_ms()
    print(f"steps {steps:4d}: wall {1e6*(t1-t0):8.1f} us  device span {1e3*d['steps']:8.1f} us  per step {1e3*d['steps']/steps:6.2f}  "
          f"sample {1e3*d['sample']:6.1f} gather {1e3*d['gather']:6.1f}  wall-device {1e6*(t1-t0)-1e3*d['steps']:6.1f}")

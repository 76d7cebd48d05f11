"""Diagnostic: in-kernel timeline of ONE k_g_gemm launch of the general step (a -DSAC_STAMPS build; SAC_GEN_STAMP_STAGE=<index
in the launch sequence>).  usage: SAC_GEN_STAMP_STAGE=1 python scratch/stamps_gemm.py <tag> 512,512 512,512 256"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from robosuite_benchmark_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsac_hip_stamps_%s.so" % sys.argv[1])
from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, SACTrainer, TanhGaussianPolicy
hp = [int(x) for x in sys.argv[2].split(",")]; hq = [int(x) for x in sys.argv[3].split(",")]; B = int(sys.argv[4])
O, A = 42, 7
rs = np.random.RandomState(0)
pol = TanhGaussianPolicy(hp, O, A, rs=rs)
qs = [FlattenMlp(hq, 1, O + A, rs=rs) for _ in range(4)]
tr = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=B, noise_seed=1)
n = 20000
buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
buf.add_block(rs.normal(0, .5, (n, O)).astype(np.float32), rs.uniform(-1, 1, (n, A)).astype(np.float32),
              rs.uniform(0, 1, (n, 1)).astype(np.float32), rs.normal(0, .5, (n, O)).astype(np.float32), np.zeros((n, 1), np.uint8))
buf.seed(3)
tr.train_loop(buf, 100, batch_size=B)
lib = _lib.load()
out = np.zeros(5 * 512 * 16, np.uint64)
lib.sac_fetch_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.sac_fetch_stamps(tr._h, out.ctypes.data_as(C.c_void_p))
w = out.reshape(5, 512, 16).astype(np.int64)[0]
blocks = [b for b in range(512) if w[b, 0] > 0]
t0 = min(w[b, 0] for b in blocks)
names = ["start", "fetch0 issued", "chunk1 top", "chunk1 in LDS", "fetch2 issued", "chunk1 multiplied", "loop end", "end"]
ww = w[blocks]
print(f"stage {os.environ.get('SAC_GEN_STAMP_STAGE')}: {len(blocks)} workgroups;",
      "  ".join(f"{nm}={np.median(ww[:, i] - t0) / 100.0:.2f}" for i, nm in enumerate(names) if (ww[:, i] > 0).all()),
      "| last end", (ww[:, 7].max() - t0) / 100.0)

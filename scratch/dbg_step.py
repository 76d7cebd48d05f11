import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_pair, rel_err, TASK_DIMS
from test_gpu_sac_step import batch_and_noise
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
O, A = TASK_DIMS["Lift"]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    oracle, hip = make_pair(O, A, B, seed=11)
    np_batch, eps = batch_and_noise(B, O, A, seed=21, term_frac=0.0)
    oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"], np_batch["terminals"],
                np_batch["next_observations"], *eps)
    hip.train(np_batch, eps=eps)
    L = oracle.last
    for name, ref in (("log_pi", L["log_pi"]), ("log_pi_next", L["log_pi2"]), ("q1", L["q1"]), ("q2", L["q2"]),
                      ("q_target", L["y"]), ("q1_new", L["q1_new"]), ("q2_new", L["q2_new"])):
        got = hip.debug_fetch(name, B); want = ref.detach().numpy().ravel()
        e = np.abs(got - want); i = int(np.argmax(e))
        print(rep, name, f"max abs err {e[i]:.3e} at row {i}  (n>1e-5: {(e>1e-5).sum()})")
    tq1 = hip.debug_fetch("tq1", B); tq2 = hip.debug_fetch("tq2", B); lp2 = hip.debug_fetch("log_pi_next", B)
    y = hip.debug_fetch("q_target", B)
    alpha = hip.debug_fetch  # placeholder
    al = float(np.exp(np.float32(-1e-3)))
    y2 = np_batch["rewards"].ravel() * 1.0 + 0.99 * (np.minimum(tq1, tq2) - al * lp2)
    e = np.abs(y - y2); i = int(np.argmax(e)); print(rep, "y vs host recompute from fetched tq/logpi:", e[i], i, (e > 1e-5).sum())
    want = L["y"].detach().numpy().ravel(); e2 = np.abs(y2 - want); print(rep, "host recompute vs oracle:", e2.max(), (e2 > 1e-5).sum())

"""Extracts the trainer/* and replay_buffer/* columns of the first rows of shipped
progress.csv files (data artefacts of the reference, /root/reference/runs/*/*/progress.csv)
into tests/golden/progress_known_answers.json.  These rows are the only known answers the
reference holds for the hot path (KA1..KA7).
Run (needs /root/reference):  python tests/golden/make_progress_fixture.py"""
import csv
import glob
import json
import os

REF = "/root/reference"
RUNS = {
    "Lift-Panda-OSC-POSE-SEED17": 7,
    "Lift-Panda-OSC-POSE-SEED59": 7,
    "Door-Panda-OSC-POSE-SEED17": 7,
    "TwoArmLift-PandaPanda-OSC-POSE-SEED17": 14,
    "Wipe-Panda-OSC-POSE-SEED17": 6,
}
out = {}
for run, act_dim in RUNS.items():
    path = glob.glob(os.path.join(REF, "runs", run, "*", "progress.csv"))[0]
    var = json.load(open(os.path.join(os.path.dirname(path), "variant.json")))
    rows = []
    with open(path) as f:
        for i, row in enumerate(csv.DictReader(f)):
            if i >= 6:
                break
            rows.append({k: float(v) for k, v in row.items()
                         if k.startswith("trainer/") or k.startswith("replay_buffer/") or k == "Epoch"})
    header = open(path).readline().strip().split(",")
    out[run] = dict(act_dim=act_dim, trainer_kwargs=var["trainer_kwargs"],
                    batch_size=var["algorithm_kwargs"]["batch_size"], rows=rows, header=header)
# global scan: Policy log std Max never above 2.0 (KA4), final buffer size (KA5)
mx = -1e9
sat = 0
for path in glob.glob(os.path.join(REF, "runs", "*", "*", "progress.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            try:
                mx = max(mx, float(row["trainer/Policy log std Max"]))
                sat = max(sat, float(row["replay_buffer/size"]))
            except (KeyError, ValueError):
                pass
out["_scan"] = dict(policy_log_std_max=mx, replay_size_max=sat)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "progress_known_answers.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, out["_scan"])

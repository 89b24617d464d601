#!/usr/bin/env python3
"""What does a k-point budget for a box-box patch cost?  MuJoCo keeps up to 8 clipped points; oracle and kernel keep the 6
deepest (with 4, the budget of the first half of round 2, 4.8 % of env-steps differed).  Two oracles, one with the study switch on, are teacher-forced from the SAME states of
the bench workload (Env03-v2, random policy, auto-reset); the per-env-step difference of their results is the effect of the
reduction alone (CPU only).    python tools/boxbox_reduction_study.py [--envs 512 --steps 200]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

ap = argparse.ArgumentParser(); ap.add_argument("--envs", type=int, default=512); ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--keep", type=int, default=6, help="points the reduced side keeps (6 = the specification)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "boxbox_reduction_study.json"))
a = ap.parse_args()
n, thr = a.envs, min(os.cpu_count() or 1, 64)
A = O.Oracle("Env03-v2", n, seed=0, auto_reset=True, noise=False, threads=thr)   # --keep deepest
B = O.Oracle("Env03-v2", n, seed=0, auto_reset=True, noise=False, threads=thr)   # all <= 8 points
A.reset(); B.reset()
rng = np.random.default_rng(1234)
errs, touched = [], 0
for t in range(a.steps):
    qpos, qvel, warm, tm = A.get_state()
    B.set_state(qpos, qvel, warm, tm); B.set_aux(A.get_aux()); B.set_xpose(*A.get_xpose())
    act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
    O.set_boxbox_max(a.keep); oa = A.step(act)
    O.set_boxbox_max(8); ob = B.step(act)
    O.set_boxbox_max(6)
    skip = oa[2] | oa[3] | ob[2] | ob[3] | (np.isnan(A.get_aux()[:, 1]) != np.isnan(B.get_aux()[:, 1]))
    d = np.abs(A.get_state()[0] - B.get_state()[0])[~skip]
    errs.append(np.stack([d[:, :9].max(axis=1), d[:, 9:].max(axis=1)], 1))
e = np.concatenate(errs)
rep = dict(kept_points=a.keep, env_steps=int(e.shape[0]), affected_env_steps=int((e.max(axis=1) > 1e-9).sum()),
           robot_qpos=dict(max=float(e[:, 0].max()), p999=float(np.quantile(e[:, 0], 0.999)), over_1e_4=int((e[:, 0] > 1e-4).sum())),
           block_qpos=dict(max=float(e[:, 1].max()), p999=float(np.quantile(e[:, 1], 0.999)), over_1e_4=int((e[:, 1] > 1e-4).sum())),
           note="difference between keeping the 4 deepest and all <= 8 clipped box-box points, oracle vs oracle, teacher-forced per env step")
json.dump(rep, open(a.out, "w"), indent=1); print(json.dumps(rep, indent=1))

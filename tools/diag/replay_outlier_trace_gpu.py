#!/usr/bin/env python3
"""Diagnostic (GPU box): one substep per launch from a dumped pre-step state, HIP path vs oracle; prints the per-dof velocity
difference and the oracle's contacts (pairs and distances) around the first substep where the difference jumps.

    python tools/diag/replay_outlier_trace_gpu.py dump.json [outlier index]
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402
from oracle import oracle as O  # noqa: E402

rep = json.load(open(sys.argv[1]))
o = rep["outliers"][int(sys.argv[2]) if len(sys.argv) > 2 else 0]
env_id, pre = rep["env"], o["pre"]
sim = BatchedSim(env_id, 1, device=0, seed=0, auto_reset=False, obs_noise=False)
orc = O.Oracle(env_id, 1, seed=0, auto_reset=False, noise=False)
qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
tm, ctrl = np.array([pre["time"]]), np.array(pre["ctrl"], dtype=np.float64)
sim.set_state(qpos, qvel, warm, tm); orc.set_state(qpos, qvel, warm, tm)
prev, shown = 0.0, 0
np.set_printoptions(precision=3, linewidth=200, suppress=False)
for k in range(250):
    fw = orc.forward(env=0, ctrl=(float(ctrl[0]), float(ctrl[1])))
    cons = [(int(c["body1"]), int(c["body2"]), round(float(c["dist"]), 6)) for c in fw["contacts"]]
    sim.physics(ctrl.astype(np.float32)[None], 1); orc.physics(ctrl[None], 1)
    vg, vo = sim.get_state()[1][0], orc.get_state()[1][0]
    ev = float(np.abs(vg - vo).max())
    if (ev > 1e-5 and ev > 5 * max(prev, 1e-8)) and shown < 6:
        shown += 1
        print(f"substep {k}: max |dqvel| {ev:.3g} (before {prev:.3g}); dqvel per dof {vg - vo}")
        print(f"   oracle contacts entering the substep: {cons}; oracle qvel {vo}")
    prev = ev
print("final |dqpos|", float(np.abs(sim.get_state()[0][0] - orc.get_state()[0][0]).max()))

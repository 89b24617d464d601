#!/usr/bin/env python3
"""Diagnostic (GPU box): replay the outliers of a tools/parity_locate.py dump on the HIP path in FUSED chunks (hints and warm start
carried inside a launch, reset between launches) against the oracle's post state.  Which chunking reproduces the difference says
whether it needs state carried across substeps; BRS_HIP_LIB selects an A/B build.

    [BRS_HIP_LIB=ab/libbrs_hip_X.so] python tools/diag/replay_outlier_fused_gpu.py dump.json
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402
from oracle import oracle as O  # noqa: E402

rep = json.load(open(sys.argv[1]))
env_id = rep["env"]
sim = BatchedSim(env_id, 1, device=0, seed=0, auto_reset=False, obs_noise=False)
orc = O.Oracle(env_id, 1, seed=0, auto_reset=False, noise=False)
for o in rep["outliers"]:
    pre = o["pre"]
    qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
    tm = np.array([pre["time"]]); ctrl = np.array(pre["ctrl"], dtype=np.float64)
    orc.set_state(qpos, qvel, warm, tm); orc.physics(ctrl[None], 250)
    qo = orc.get_state()[0][0]
    out = {}
    for chunks in ([250], [125, 125], [50] * 5, [10] * 25, [1] * 250):
        sim.set_state(qpos, qvel, warm, tm)
        for n in chunks:
            sim.physics(ctrl.astype(np.float32)[None], n)
        qg = sim.get_state()[0][0]
        d = np.abs(qg - qo)
        out[f"{len(chunks)}x{chunks[0]}"] = (float(d[:9].max()), float(d[9:].max()) if d.size > 9 else 0.0)
    print(os.environ.get("BRS_HIP_LIB", "product"), o["env"], o["step"], "recorded", o["max_dqpos"], {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in out.items()}, flush=True)

#!/usr/bin/env python3
"""Diagnostic (GPU box): replay the outliers of a tools/parity_locate.py dump THROUGH brs_step -- the folded step kernel the campaign
ran, not brs_physics (another instantiation with runtime constants, whose rounding differs: the round-3 outlier did not reproduce
there) -- from the dumped pre-step state, aux, accessor pose and action, against the oracle stepping the same.  BRS_HIP_LIB selects
an A/B build (tools/ab_build.py), so the same env-step can be tried on variants of the kernel.

    [BRS_HIP_LIB=ab/libbrs_hip_X.so] python tools/diag/replay_outlier_step_gpu.py dump.json
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402
from oracle import oracle as O  # noqa: E402

rep = json.load(open(sys.argv[1]))
env_id, ar = rep["env"], bool(rep.get("auto_reset", True))
for o in rep["outliers"]:
    pre = o["pre"]
    if "aux" not in pre:
        print("dump predates the aux / accessor-pose fields: re-run tools/parity_locate.py"); break
    # the env's Philox stream is keyed by (seed, global env index): a one-env handle at env_index_base = the env's index
    sim = BatchedSim(env_id, 1, device=0, seed=rep.get("seed", 0), env_index_base=o["env"], auto_reset=ar, obs_noise=False)
    orc = O.Oracle(env_id, 1, seed=rep.get("seed", 0), env_index_base=o["env"], auto_reset=ar, noise=False)
    qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
    tm, aux = np.array([pre["time"]]), np.array(pre["aux"], dtype=np.float64)[None]
    xq, xp = np.array(pre["xquat"], dtype=np.float64)[None], np.array(pre["xpos"], dtype=np.float64)[None]
    act = np.array(pre["action"], dtype=np.float32)[None]
    for s in (sim, orc):
        s.set_state(qpos, qvel, warm, tm); s.set_aux(aux); s.set_xpose(xq, xp)
    sim.step(torch.from_numpy(act).cuda()); orc.step(act)
    torch.cuda.synchronize()
    d = np.abs(sim.get_state()[0][0] - orc.get_state()[0][0])
    print(os.environ.get("BRS_HIP_LIB", "product"), "env", o["env"], "step", o["step"], "recorded", f"{o['max_dqpos']:.3g}",
          "replayed: robot", f"{d[:9].max():.3g}", "block", f"{d[9:].max():.3g}" if d.size > 9 else "-", flush=True)
    sim.close(); orc.close()

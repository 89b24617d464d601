#!/usr/bin/env python3
"""Second-level replay of the outliers tools/parity_locate.py dumped (GPU box): the HIP path itself, ONE substep per launch,
against the oracle from the dumped pre-step state.  parity_locate's own replay compares the kernel source on the host in
float and in double; where those two agree (the host's float rounding is not the GPU's: fast-math reciprocal / rsqrt, FMA
contraction) only the GPU can show where ITS trajectory leaves the oracle's.  For every outlier: first substep at which the
velocity difference jumps, and the oracle's contact list (body pairs) just before and after -- a jump that coincides with a
contact point appearing or disappearing is the "switches on one substep apart" mechanism of DESIGN.md 2.1.

    python tools/parity_replay_gpu.py profiles/r02_parity_config3_large.json   (rewrites the file with replay_gpu_vs_oracle added)
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def contact_pairs(orc, ctrl):
    fw = orc.forward(env=0, ctrl=(float(ctrl[0]), float(ctrl[1])))
    return sorted((int(c["body1"]), int(c["body2"])) for c in fw["contacts"])


def main():
    path = sys.argv[1]
    rep = json.load(open(path))
    from balance_robot_mujoco_rl_amd import BatchedSim
    from oracle import oracle as O
    env_id = rep["env"]
    sim = BatchedSim(env_id, 1, device=0, seed=0, auto_reset=False, obs_noise=False)
    orc = O.Oracle(env_id, 1, seed=0, auto_reset=False, noise=False)
    for o in rep["outliers"]:
        pre = o["pre"]
        qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
        tm = np.array([pre["time"]])
        ctrl = np.array(pre["ctrl"], dtype=np.float64)
        sim.set_state(qpos, qvel, warm, tm); orc.set_state(qpos, qvel, warm, tm)
        c32 = ctrl.astype(np.float32)[None]
        first_jump, prev, trace, pairs = None, 0.0, [], []
        for k in range(250):
            pairs.append(contact_pairs(orc, ctrl))
            sim.physics(c32, 1); orc.physics(c32.astype(np.float64), 1)
            qg, vg, _, _ = sim.get_state(); qo, vo, _, _ = orc.get_state()
            ev = float(np.abs(vg - vo).max())
            trace.append((float(np.abs(qg - qo).max()), ev))
            if first_jump is None and ev > 1e-3 and ev > 20 * max(prev, 1e-7):
                first_jump = k
            prev = ev
        pairs.append(contact_pairs(orc, ctrl))
        rec = dict(first_substep_dqvel_jump=first_jump, final_dqpos=trace[-1][0], final_dqvel=trace[-1][1])
        if first_jump is not None:
            lo, hi = max(0, first_jump - 2), min(250, first_jump + 2)
            rec["dqvel_before_jump"] = trace[first_jump - 1][1] if first_jump else None
            rec["dqvel_at_jump"] = trace[first_jump][1]
            rec["oracle_contact_pairs_around_jump"] = {str(k): pairs[k] for k in range(lo, hi + 1)}
            rec["oracle_contact_set_changes_within_2_substeps"] = any(pairs[k] != pairs[k + 1] for k in range(lo, hi))
        o["replay_gpu_vs_oracle"] = rec
        print(o["env"], o["step"], {k: v for k, v in rec.items() if k != "oracle_contact_pairs_around_jump"},
              rec.get("oracle_contact_pairs_around_jump"), flush=True)
    json.dump(rep, open(path, "w"), indent=1)
    sim.close(); orc.close()


if __name__ == "__main__":
    main()

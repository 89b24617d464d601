#!/usr/bin/env python3
"""Parity report (run on the GPU box): BASELINE.json config 2 and a reduced config 3, teacher-forced per env step.

    python tools/parity_report.py [--envs 4096] [--steps 1000] [--out profiles/r01_parity_report.json]

Config 2: Env01-v2, 4,096 envs, zero action, noise off, auto-reset off, 1,000 steps: at every step both sims start from
the ORACLE's state, advance one env step (250 substeps), and max |dqpos|, |dqvel|, |dobs| are recorded.
Oracle = oracle/brs_oracle.c (fp64 CPU restatement; MuJoCo itself is not installable here: physics parity vs MuJoCo is
unpinned)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim
from oracle import oracle as O


def run(env_id, n, steps, action_mode, seed=0, auto_reset=False):
    threads = min(os.cpu_count() or 1, 64)
    sim = BatchedSim(env_id, n, device=0, seed=seed, auto_reset=auto_reset, obs_noise=False)
    orc = O.Oracle(env_id, n, seed=seed, auto_reset=auto_reset, noise=False, threads=threads)
    excluded = 0
    sim.reset(); orc.reset()
    rng = np.random.default_rng(1234)
    dq, dv, dobs, over = [], [], [], 0
    up_steps = up_over = 0; up_max = 0.0   # env-steps that start upright (tilt < 60 deg; the env terminates at 50 deg pitch)
    t0 = time.time()
    for t in range(steps):
        qpos, qvel, warm, tm = orc.get_state()
        aux = orc.get_aux(); xq, xp = orc.get_xpose()
        sim.set_state(qpos, qvel, warm, tm); sim.set_aux(aux); sim.set_xpose(xq, xp)
        act = np.zeros((n, 2), np.float32) if action_mode == "zero" else rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        out_g = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(act).cuda())]
        out_o = orc.step(act)
        o_g, o_o = out_g[0], out_o[0]
        qg, vg, _, _ = sim.get_state(); qo, vo, _, _ = orc.get_state()
        e = np.abs(qg - qo).max(axis=1)
        if auto_reset:
            # an env that finished its episode on either side was re-drawn from its RNG stream, and a block removed /
            # re-thrown on one side only (|v| within rounding of the 0.1 threshold) is a different discrete outcome:
            # neither is a physics error of this step -- leave them out and count them
            skip = out_g[2].astype(bool) | out_g[3].astype(bool) | out_o[2].astype(bool) | out_o[3].astype(bool)
            skip |= np.isnan(sim.get_aux()[:, 1]) != np.isnan(orc.get_aux()[:, 1])
            excluded += int(skip.sum())
            e = np.where(skip, 0.0, e)
            vg = np.where(skip[:, None], vo, vg); o_g = np.where(skip[:, None], o_o, o_g)
        dq.append(e); dv.append(np.abs(vg - vo).max(axis=1)); dobs.append(np.abs(o_g - o_o)[:, [0, 2, 3, 4, 5]].max(axis=1))
        over += int((e > 1e-4).sum())
        upright = 1 - 2 * (qpos[:, 4] ** 2 + qpos[:, 5] ** 2) > 0.5   # body z . world z > cos 60
        up_steps += int(upright.sum()); up_over += int((e[upright] > 1e-4).sum())
        up_max = max(up_max, float(e[upright].max()) if upright.any() else 0.0)
        if t % 100 == 0:
            print(f"  {env_id} step {t}: max|dqpos| so far {np.max(dq):.3g} ({time.time() - t0:.0f} s)", flush=True)
    dq, dv, dobs = np.array(dq), np.array(dv), np.array(dobs)
    return dict(env=env_id, envs=n, steps=steps, actions=action_mode, substeps=250,
                max_dqpos=float(dq.max()), p999_dqpos=float(np.quantile(dq, 0.999)), median_dqpos=float(np.median(dq)),
                max_dqvel=float(dv.max()), p999_dqvel=float(np.quantile(dv, 0.999)), median_dqvel=float(np.median(dv)),
                max_dobs_excl_pitchdot=float(dobs.max()), env_steps_over_1e_4=over, env_steps=int(dq.size),
                auto_reset=auto_reset, excluded_env_steps=excluded, upright_env_steps=up_steps, upright_env_steps_over_1e_4=up_over, upright_max_dqpos=up_max,
                oracle="oracle/brs_oracle.c (fp64 restatement; NOT MuJoCo)", wall_s=round(time.time() - t0, 1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--only3", action="store_true", help="skip config 2")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_parity_report.json"))
    a = ap.parse_args()
    rep = {} if a.only3 else {"config2": run("Env01-v2", a.envs, a.steps, "zero")}
    rep["config3_autoreset"] = run("Env03-v2", max(256, a.envs // 4), max(50, a.steps // 4), "random", auto_reset=True)
    rep["config3_reduced"] = run("Env03-v2", max(256, a.envs // 4), max(50, a.steps // 4), "random")
    json.dump(rep, open(a.out, "w"), indent=1)
    print(json.dumps(rep, indent=1))

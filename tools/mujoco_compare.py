#!/usr/bin/env python3
"""Compare the CPU oracle (and, with --gpu, the HIP path) with MuJoCo itself -- ONLY where `import mujoco` works.

MuJoCo is not installed in the build container nor on the GPU box and cannot be installed (no network); this script
then prints "oracle unavailable: mujoco is not importable" and exits 3.  It is the tool that closes SURVEY.md App. C's
[VERIFY] list on the first machine that has MuJoCo 3.2.0 (the reference's pin, conda-environment.yaml:7).  It loads OUR
re-typed XML (tools/mujoco_assets/), never the reference's files.

    python tools/mujoco_compare.py [--env Env01-v2] [--envs 16] [--steps 200] [--gpu]
"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env01-v2"); ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200); ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args()
    try:
        import mujoco
    except Exception as e:  # noqa: BLE001
        print(f"oracle unavailable: mujoco is not importable ({type(e).__name__}: {e})")
        return 3
    from oracle import oracle as O
    fam = "env01" if a.env.startswith("Env01") else "env03"
    m = mujoco.MjModel.from_xml_path(os.path.join(ROOT, "tools", "mujoco_assets", fam + ".xml"))
    orc = O.Oracle(a.env, a.envs, seed=0, auto_reset=False, noise=False)
    mi = orc.model_info()
    rep = {"mujoco_version": mujoco.__version__,
           "model_constants": {
               "body_mass": dict(mujoco=m.body_mass.tolist(), oracle=mi["body_mass"][:m.nbody].tolist()),
               "body_invweight0": dict(mujoco=m.body_invweight0.tolist(), oracle=mi["invweight0"][:m.nbody].tolist()),
               "meaninertia": dict(mujoco=float(m.stat.meaninertia), oracle=mi["meaninertia"])}}
    orc.reset()
    sim = None
    if a.gpu:
        import torch
        from balance_robot_mujoco_rl_amd import BatchedSim
        sim = BatchedSim(a.env, a.envs, seed=0, auto_reset=False, obs_noise=False)
    rng = np.random.default_rng(0)
    datas = [mujoco.MjData(m) for _ in range(a.envs)]
    dq_o, dq_g, lag = [], [], []
    for t in range(a.steps):
        qpos, qvel, warm, tm = orc.get_state()
        act = rng.uniform(-1, 1, size=(a.envs, 2)).astype(np.float32) * (t % 3 != 0)
        ctrl = qvel[:, 6:8] + act.astype(np.float64) * 4.0
        for i, d in enumerate(datas):  # teacher-forced from the oracle's state
            d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.qacc_warmstart[:] = warm[i]; d.time = tm[i]; d.ctrl[:] = ctrl[i]
            mujoco.mj_step(m, d, nstep=250)
        if sim is not None:
            sim.set_state(qpos, qvel, warm, tm); sim.physics(ctrl.astype(np.float32), 250)
            dq_g.append(np.abs(sim.get_state()[0] - np.stack([d.qpos for d in datas])).max())
        orc.physics(ctrl, 250)
        qo = orc.get_state()[0]
        qm = np.stack([d.qpos.copy() for d in datas])
        dq_o.append(np.abs(qo - qm).max())
        # xquat after mj_step lags qpos by one substep
        lag.append(max(np.abs(d.xquat[1] - d.qpos[3:7] / np.linalg.norm(d.qpos[3:7])).max() for d in datas))
    rep["max_dqpos_oracle_vs_mujoco"] = float(np.max(dq_o)); rep["median"] = float(np.median(dq_o))
    if dq_g:
        rep["max_dqpos_hip_vs_mujoco"] = float(np.max(dq_g))
    rep["xquat_lags_qpos_max"] = float(np.max(lag))
    print(json.dumps(rep, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())

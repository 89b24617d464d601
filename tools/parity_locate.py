#!/usr/bin/env python3
"""Locate and explain the env-steps whose fp32 result leaves the 1e-4 band (VERDICT r1, next #1a).

Teacher = the kernel source in DOUBLE on the host (tests/hostsim; agrees with oracle/ to 1e-9 per env step, see
tests/test_hostsim_parity.py) or oracle/ itself (--teacher oracle); student = the same source in FLOAT on the host
(--student host) or the HIP kernel (--student gpu, needs a GPU).  Per env step both start from the teacher's state
(teacher-forced).  Every env-step with max |dqpos| > tol is dumped with its pre-step state and replayed substep by
substep in double and float (host builds) to find the first diverging substep and what happened there: contact onset
(a contact list that differs), an active-set difference, the iteration cap, or plain growth (stick-slip).

    python tools/parity_locate.py --env Env03-v2 --envs 1024 --steps 400 --out profiles/r02_parity_outliers_host.json
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim.hostsim import HostSim  # noqa: E402

GROUPS = {"torso_pos": [0, 1, 2], "torso_quat": [3, 4, 5, 6], "wheel_angles": [7, 8], "block_pos": [9, 10, 11], "block_quat": [12, 13, 14, 15]}


def group_err(dq):
    return {g: float(np.abs(dq[[i for i in idx if i < dq.size]]).max()) for g, idx in GROUPS.items() if idx[0] < dq.size}


def replay(env_id, qpos, qvel, warm, tm, aux, ctrl, tol):
    """one env, substep by substep: host double vs host float from the same pre-step state"""
    d = HostSim(env_id, 1, noise=False, double=True)
    f = HostSim(env_id, 1, noise=False, double=False)
    for s in (d, f):
        s.set_state(qpos[None], qvel[None], warm[None], np.array([tm]))
        s.set_aux(aux[None])
    first_over, first_jump, prev = None, None, 0.0
    trace = []
    for k in range(250):
        d.physics(ctrl[None], 1); f.physics(ctrl[None], 1)
        qd, vd, _, _ = d.get_state(); qf, vf, _, _ = f.get_state()
        e = float(np.abs(qd - qf).max()); ev = float(np.abs(vd - vf).max())
        trace.append((e, ev))
        if first_jump is None and ev > 1e-3 and ev > 20 * max(prev, 1e-7):
            first_jump = k
        if first_over is None and e > tol:
            first_over = k
        prev = ev
    return dict(first_substep_dqvel_jump=first_jump, first_substep_dqpos_over=first_over,
                final_dqpos=trace[-1][0], final_dqvel=trace[-1][1],
                dqvel_at_jump=None if first_jump is None else trace[first_jump][1],
                dqvel_before_jump=None if not first_jump else trace[first_jump - 1][1])


def contacts_of(env_id, qpos, qvel, tm):
    """contact list of the pre-state as the oracle sees it (body pairs and distances)"""
    from oracle import oracle as O
    o = O.Oracle(env_id, 1, noise=False)
    o.set_state(qpos[None], qvel[None], None, np.array([tm]))
    fw = o.forward(env=0, ctrl=(float(qvel[6]), float(qvel[7])))
    return [dict(b1=c["body1"], b2=c["body2"], dist=round(float(c["dist"]), 6)) for c in fw["contacts"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env03-v2")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--auto-reset", type=int, default=1)
    ap.add_argument("--teacher", choices=["hostdouble", "oracle"], default="hostdouble")
    ap.add_argument("--student", choices=["host", "gpu"], default="host")
    ap.add_argument("--actions", choices=["random", "zero", "policy"], default="random",
                    help="policy = the reference's MuJoCo-trained balance policy (tests/quant_policy.py) acting on the teacher's observations")
    ap.add_argument("--max-dump", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0, help="seed of both simulators' Philox streams and of the action generator (0 = the campaign of profiles/r03_parity_*.json)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_parity_outliers.json"))
    a = ap.parse_args()
    n, thr = a.envs, min(os.cpu_count() or 1, 64)
    ar = bool(a.auto_reset)
    if a.teacher == "oracle":
        from oracle import oracle as O
        T = O.Oracle(a.env, n, seed=a.seed, auto_reset=ar, noise=False, threads=thr)
    else:
        T = HostSim(a.env, n, seed=a.seed, auto_reset=ar, noise=False, double=True, threads=thr)
    if a.student == "gpu":
        import torch
        from balance_robot_mujoco_rl_amd import BatchedSim
        S = BatchedSim(a.env, n, device=0, seed=a.seed, auto_reset=ar, obs_noise=False)
    else:
        S = HostSim(a.env, n, seed=a.seed, auto_reset=ar, noise=False, double=False, threads=thr)
    obs_t = T.reset(); S.reset()
    pol = None
    if a.actions == "policy":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch as _torch
        from quant_policy import QuantMovePolicy
        _qp = QuantMovePolicy()
        pol = lambda o: _qp.act(_torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32)), "mean").numpy()
    rng = np.random.default_rng(1234 + a.seed)
    outl, nsteps, over, excl, worst = [], 0, 0, 0, 0.0
    upright_over, upright_n = 0, 0
    # per coordinate group x {upright, fallen}: max error and number of env-steps above tol
    gstat = {f"{g}/{u}": [0.0, 0] for g in ("robot", "block") for u in ("upright", "fallen")}
    hist = np.zeros(12, dtype=np.int64)  # log10 buckets of the per-env-step error: <1e-10 ... >=1e0
    t0 = time.time()
    for t in range(a.steps):
        qpos, qvel, warm, tm = T.get_state()
        aux = T.get_aux(); xq, xp = T.get_xpose()
        S.set_state(qpos, qvel, warm, tm); S.set_aux(aux); S.set_xpose(xq, xp)
        if pol is not None:
            act = pol(obs_t)
        else:
            act = np.zeros((n, 2), np.float32) if a.actions == "zero" else rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        if a.student == "gpu":
            og = [x.cpu().numpy().copy() for x in S.step(torch.from_numpy(act).cuda())]
        else:
            og = S.step(act)
        ot = T.step(act)
        obs_t = ot[0]
        qs = S.get_state()[0]; qt = T.get_state()[0]
        skip = np.zeros(n, bool)
        if ar:  # finished episodes were re-drawn; a block removed / re-thrown on one side only is a discrete difference
            skip = np.asarray(og[2]).astype(bool) | np.asarray(og[3]).astype(bool) | ot[2] | ot[3]
            skip |= np.isnan(S.get_aux()[:, 1]) != np.isnan(T.get_aux()[:, 1])
        e = np.where(skip, 0.0, np.abs(qs - qt).max(axis=1))
        excl += int(skip.sum()); nsteps += n
        worst = max(worst, float(e.max()))
        hist += np.bincount(np.clip(np.floor(np.log10(np.maximum(e[~skip], 1e-11))).astype(int) + 11, 0, 11), minlength=12)
        upright = 1 - 2 * (qpos[:, 4] ** 2 + qpos[:, 5] ** 2) > 0.5
        upright_n += int((upright & ~skip).sum()); upright_over += int((e[upright] > a.tol).sum())
        dqa = np.where(skip[:, None], 0.0, np.abs(qs - qt))
        for g, sl in (("robot", slice(0, 9)), ("block", slice(9, 16))):
            if dqa.shape[1] <= sl.start:
                continue
            eg = dqa[:, sl].max(axis=1)
            for u, m in (("upright", upright), ("fallen", ~upright)):
                if m.any():
                    gstat[f"{g}/{u}"][0] = max(gstat[f"{g}/{u}"][0], float(eg[m].max()))
                    gstat[f"{g}/{u}"][1] += int((eg[m] > a.tol).sum())
        bad = np.nonzero(e > a.tol)[0]
        over += bad.size
        for i in bad:
            if len(outl) >= a.max_dump:
                break
            ctrl = qvel[i, 6:8] + act[i].astype(np.float64) * 4.0
            rec = dict(env=int(i), step=int(t), max_dqpos=float(e[i]), per_group=group_err(qs[i] - qt[i]),
                       upright=bool(upright[i]), tilt_deg=float(np.degrees(np.arccos(np.clip(1 - 2 * (qpos[i, 4] ** 2 + qpos[i, 5] ** 2), -1, 1)))),
                       pre=dict(qpos=qpos[i].tolist(), qvel=qvel[i].tolist(), warm=warm[i].tolist(), time=float(tm[i]), ctrl=ctrl.tolist(),
                                aux=aux[i].tolist(), xquat=xq[i].tolist(), xpos=xp[i].tolist(), action=act[i].tolist()),  # (aux / accessor pose / action: what a replay through brs_step needs)
                       contacts_pre=contacts_of(a.env, qpos[i], qvel[i], tm[i]))
            rec["replay_host_double_vs_float"] = replay(a.env, qpos[i], qvel[i], warm[i], float(tm[i]), aux[i], ctrl, a.tol)
            outl.append(rec)
        if t % 50 == 0:
            print(f"step {t}: worst {worst:.3g}, over {over}/{nsteps} ({time.time() - t0:.0f} s)", flush=True)
    rep = dict(env=a.env, envs=n, steps=a.steps, seed=a.seed, teacher=a.teacher, student=a.student, auto_reset=ar, tol=a.tol, actions=a.actions,
               env_steps=nsteps, excluded=excl, over=over, worst=worst, upright_env_steps=upright_n, upright_over=upright_over,
               per_group={k: dict(max=v[0], over_tol=v[1]) for k, v in gstat.items()},
               log10_error_histogram={f"1e{k - 11}": int(v) for k, v in enumerate(hist)}, outliers=outl)
    json.dump(rep, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k != "outliers"}, indent=1))
    for r in outl[:12]:
        print(r["env"], r["step"], f"{r['max_dqpos']:.3g}", r["per_group"], "tilt", round(r["tilt_deg"], 1), r["replay_host_double_vs_float"], r["contacts_pre"])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak run of the HIP path (GPU box): every registered id at the benchmark batch size for thousands of env steps under a
random policy, with and without auto-reset (robots that stay down, blocks piling onto them), checking after every chunk
the invariants no physical state may break -- finite state, unit quaternions (fp64 accumulators), nothing through the floor,
no bad-state reset (the kernel's NaN / |qacc| guard, SURVEY 5: MuJoCo's mj_check*), bounded speeds.

    python tools/soak.py [--envs 65536] [--steps 2000] [--out gpurun_out/soak.json]
"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402


def soak(env_id, n, steps, auto_reset, chunk=250):
    sim = BatchedSim(env_id, n, seed=7, auto_reset=auto_reset)
    sim.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    blk = env_id.startswith("Env03")
    rep = dict(env=env_id, envs=n, steps=steps, auto_reset=auto_reset, done_flags_seen=0, violations=[])
    t0 = time.time()
    done_total = torch.zeros((), dtype=torch.int64, device="cuda")
    for k in range(steps):
        a = torch.rand((n, 2), generator=gen, device="cuda") * 2 - 1
        o, r, te, tr, to = sim.step(a)
        done_total += (te | tr).sum()
        if (k + 1) % chunk == 0 or k == steps - 1:
            qpos, qvel, warm, tm = sim.get_state()
            aux = sim.get_aux()
            v = []
            if not (np.isfinite(qpos).all() and np.isfinite(qvel).all() and np.isfinite(warm).all()): v.append("non-finite state")
            if not torch.isfinite(o).all().item() or not torch.isfinite(r).all().item(): v.append("non-finite obs/reward")
            if np.abs(np.linalg.norm(qpos[:, 3:7], axis=1) - 1).max() > 1e-9: v.append("torso quaternion not unit")
            if qpos[:, 2].min() < -0.06: v.append(f"torso below the floor: {qpos[:, 2].min():.4f}")
            if blk:
                if np.abs(np.linalg.norm(qpos[:, 12:16], axis=1) - 1).max() > 1e-9: v.append("block quaternion not unit")
                if qpos[:, 11].min() < -0.025: v.append(f"block below the floor: {qpos[:, 11].min():.4f}")
                if np.abs(qvel[:, 8:11]).max() > 50: v.append(f"block speed {np.abs(qvel[:, 8:11]).max():.1f} m/s")
            if np.abs(qvel[:, 0:3]).max() > 20: v.append(f"torso speed {np.abs(qvel[:, 0:3]).max():.1f} m/s")
            if np.abs(qvel[:, 6:8]).max() > 400: v.append(f"wheel speed {np.abs(qvel[:, 6:8]).max():.1f} rad/s")
            if aux[:, 7].max() > 0: v.append(f"bad-state resets: {int(aux[:, 7].sum())}")
            if v:
                rep["violations"].append(dict(step=k + 1, what=v))
    rep["done_flags_seen"] = int(done_total.item())
    rep["wall_s"] = round(time.time() - t0, 2)
    rep["max_abs_wheel_speed"] = float(np.abs(qvel[:, 6:8]).max())
    rep["min_torso_z"] = float(qpos[:, 2].min())
    if blk:
        rep["min_block_z"] = float(qpos[:, 11].min())
    sim.close()
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak.json"))
    a = ap.parse_args()
    reps = []
    for env_id in ("Env03-v2", "Env03-v1", "Env01-v2", "Env01-v1", "Env01-v3", "Env02-v1"):
        for ar in (True, False):
            steps = a.steps if ar else min(a.steps, 1200)
            r = soak(env_id, a.envs, steps, ar)
            print(json.dumps(r), flush=True)
            reps.append(r)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(tool="soak", runs=reps, total_violations=sum(len(r["violations"]) for r in reps)), open(a.out, "w"), indent=1)
    sys.exit(1 if any(r["violations"] for r in reps) else 0)


if __name__ == "__main__":
    main()

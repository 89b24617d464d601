#!/usr/bin/env python3
"""Linear-controller reference for the RL results: sweep PD(+wheel-speed, +yaw) gains over the batched simulator,
one gain set per group of envs, and report first-episode survival.  What a hand-tuned controller achieves on Env03-v2
bounds what "trained to stable balance" can mean in this simulator.

    python tools/pd_sweep.py [--env Env03-v2] [--per 512] [--steps 1200]
"""
import argparse, itertools, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balance_robot_mujoco_rl_amd import BatchedSim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env03-v2"); ap.add_argument("--per", type=int, default=512)
    ap.add_argument("--steps", type=int, default=1200); ap.add_argument("--out", default="")
    a = ap.parse_args()
    grid = list(itertools.product([5.0, 10.0, 20.0, 40.0], [0.3, 1.0, 2.0], [0.0, -0.02, -0.05, -0.1, -0.2], [0.0, 0.05]))
    n = len(grid) * a.per
    g = torch.tensor(grid, device="cuda").repeat_interleave(a.per, 0)
    sim = BatchedSim(a.env, n, device=0, seed=7, auto_reset=True)
    obs = sim.reset().clone()
    alive = torch.ones(n, dtype=torch.bool, device="cuda"); length = torch.zeros(n, device="cuda")
    for _ in range(a.steps):
        pitch, pdot = obs[:, 0] * 0.25, obs[:, 1]
        wl, wr = obs[:, 2] * 42.5, obs[:, 3] * 42.5           # rad/s
        u = (g[:, 0] * pitch + g[:, 1] * pdot + g[:, 2] * (wl - wr) * 0.5).clamp(-1, 1)
        yaw = g[:, 3] * (wl + wr)                               # wheel-sum = yaw rate on the ground
        act = torch.stack([-u - yaw, u - yaw], 1).clamp(-1, 1).contiguous()
        o, r, te, tr, _ = sim.step(act)
        length += alive
        alive &= ~te.bool()
        obs = o.clone()
    res = []
    for i, k in enumerate(grid):
        sl = slice(i * a.per, (i + 1) * a.per)
        res.append(dict(kp=k[0], kd=k[1], kv=k[2], kyaw=k[3], survived=float(alive[sl].float().mean()),
                        mean_len=float(length[sl].mean())))
    res.sort(key=lambda r: -r["survived"])
    for r in res[:12]:
        print(json.dumps(r))
    if a.out:
        json.dump(dict(env=a.env, per=a.per, steps=a.steps, results=res), open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()

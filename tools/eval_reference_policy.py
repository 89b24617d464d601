#!/usr/bin/env python3
"""How does the policy the reference ships (envs/RobotMovePolicy.tflite -> tests/golden/robot_move_policy.npz, dequantised
weights through the on-device policy kernels) fare on a registered id of THIS simulator?  Deterministic evaluation, episode
statistics as tools/train_ppo_torch.py's evaluate() prints them.  A reading aid for config 5: what a policy trained against
MuJoCo by the reference's author achieves on the envs the curriculum uses.

    python tools/eval_reference_policy.py --env Env01-v2 --envs 4096 --steps 1500
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402
from balance_robot_mujoco_rl_amd.policy import DevicePolicy  # noqa: E402
from quant_policy import QuantMovePolicy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env01-v2"); ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=1500)
    a = ap.parse_args()
    pol = DevicePolicy(device=0); pol.set_weights(QuantMovePolicy().float_params("mean"))
    sim = BatchedSim(a.env, a.envs, device=0, seed=123, auto_reset=True)
    obs = sim.reset().clone()
    n = a.envs
    ep_len = torch.zeros(n, device=sim.device); lens = []; ntr = nte = 0
    ever = torch.zeros(n, dtype=torch.bool, device=sim.device)
    for t in range(a.steps):
        _, ac, _, _ = pol.act(obs, t, deterministic=True)
        o, r, te, tr, _ = sim.step(ac)
        ep_len += 1
        done = (te | tr).bool()
        if done.any():
            lens.append(ep_len[done].clone()); ntr += int((tr.bool() & ~te.bool()).sum()); nte += int(te.bool().sum()); ep_len[done] = 0; ever |= done
        obs = o.clone()
    lens = torch.cat(lens) if lens else torch.zeros(0)
    print(json.dumps(dict(env=a.env, envs=n, steps=a.steps, policy="reference RobotMovePolicy (dequantised, mean output)", episodes=int(lens.numel()),
                          first_episode_still_running=int((~ever).sum()), fell=nte, reached_time_limit=ntr,
                          mean_ep_len=float(lens.mean()) if lens.numel() else None, median_ep_len=float(lens.median()) if lens.numel() else None)))


if __name__ == "__main__":
    main()

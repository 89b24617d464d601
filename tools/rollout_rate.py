#!/usr/bin/env python3
"""Cost of the on-device rollout side next to the env step it feeds: env-steps/s of DeviceRollout.collect()
(policy forward + Gaussian sample -> env step -> time-limit bootstrap, then GAE; nothing leaves the GPU) against the bare
brs_step loop on the same handle.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel rows.

    python tools/rollout_rate.py [--env Env03-v2] [--envs 65536] [--T 32] [--rounds 6] [--out profiles/r03_rollout_rate.json]
"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim  # noqa: E402
from balance_robot_mujoco_rl_amd.policy import DevicePolicy, DeviceRollout, NPARAM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env03-v2"); ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--T", type=int, default=32); ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--preroll", type=int, default=300); ap.add_argument("--out", default=None)
    a = ap.parse_args()
    n = a.envs
    sim = BatchedSim(a.env, n, device=0, seed=0, auto_reset=True)
    pol = DevicePolicy(device=0, seed=1)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal(NPARAM) * 0.1).astype(np.float32); w[-2:] = 0.0   # log_std = 0: actions ~ N(mean, 1), clipped to +-1
    pol.set_weights(w)
    ro = DeviceRollout(sim, pol, a.T)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    acts = [(torch.rand((n, 2), generator=gen, device="cuda") * 2 - 1).contiguous() for _ in range(16)]
    sim.reset()
    for k in range(a.preroll):                 # episodes de-phased, as in bench.py
        sim.step(acts[k % 16])
    torch.cuda.synchronize()
    # bare env steps
    t0 = time.perf_counter()
    for k in range(a.T * a.rounds):
        sim.step(acts[k % 16])
    torch.cuda.synchronize()
    bare = n * a.T * a.rounds / (time.perf_counter() - t0)
    ro._last_obs = sim.obs.clone()              # continue from the de-phased state (collect() would reset otherwise)
    ro.collect()                                 # warm-up round
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.rounds):
        ro.collect()
    torch.cuda.synchronize()
    roll = n * a.T * a.rounds / (time.perf_counter() - t0)
    out = dict(env=a.env, envs=n, rollout_steps=a.T, rounds=a.rounds, bare_env_steps_per_s=bare, rollout_env_steps_per_s=roll,
               rollout_over_bare=roll / bare,
               note="collect() = brs_policy_act -> brs_step -> brs_rollout_bootstrap per step (+ 5 small torch copies), then brs_policy_value "
                    "and brs_gae once per round; random-weight MlpPolicy, log_std 0; bare = brs_step with pre-generated U(-1,1) actions")
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

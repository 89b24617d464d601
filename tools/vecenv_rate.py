#!/usr/bin/env python3
"""Host-inclusive rate of the SB3-style boundary: BalanceVecEnv.step() with numpy actions in and numpy
obs/reward/done + infos out (H2D of the actions, D2H of the outputs, info dicts for finished episodes), next to the
device-resident rate bench.py reports.  DESIGN.md quotes the number; it is never bench.py's `value`.

    python tools/vecenv_rate.py [--env Env03-v2] [--envs 65536] [--steps 100]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balance_robot_mujoco_rl_amd import make_vec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env03-v2"); ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=10); ap.add_argument("--preroll", type=int, default=300)
    a = ap.parse_args()
    env = make_vec(a.env, a.envs, devices=[0])
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (a.envs, 2)).astype(np.float32) for _ in range(8)]
    for i in range(a.warmup):
        env.step(acts[i % 8])
    res = {}
    for mode in ("arrays", "dicts"):
        # "arrays": the consumer reads obs / rewards / dones and the eager arrays of the finished episodes
        # (infos.done_indices, infos.terminal_observations, ...); "dicts": it also touches infos[i] of every finished env,
        # which materialises the SB3-style per-env dicts
        for i in range(a.preroll if mode == "arrays" else 0):
            env.step(acts[i % 8])
        t0 = time.perf_counter(); ndone = 0
        for i in range(a.steps):
            _o, _r, d, infos = env.step(acts[i % 8])
            ndone += int(infos.done_indices.size)
            if mode == "dicts" and infos.done_indices.size:
                _ = infos[int(infos.done_indices[0])]["terminal_observation"]
        dt = time.perf_counter() - t0
        res[mode] = dict(ms_per_step=1e3 * dt / a.steps, env_steps_per_s=a.envs * a.steps / dt, episodes_finished=ndone)
    print(json.dumps(dict(tool="vecenv_rate", env=a.env, envs=a.envs, steps=a.steps, preroll=a.preroll, **{f"{k}_{kk}": vv for k, v in res.items() for kk, vv in v.items()},
                          note="numpy in / numpy out through BalanceVecEnv: pinned H2D of the actions, kernel, one packed pinned D2H, "
                               "host copies; 'dicts' additionally builds the per-env info dicts of the finished episodes")))
    env.close()


if __name__ == "__main__":
    main()

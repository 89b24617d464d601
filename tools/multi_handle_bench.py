#!/usr/bin/env python3
"""Experiment harness: H independent handles of N envs each, every handle on its own HIP stream of
ONE process, stepped round-robin without synchronising in between.  Compare the aggregate rate with one handle of H*N
envs (bench.py --envs) and with H processes on the same GPU (bench.py under torch.distributed.run with
BRS_BENCH_ONE_DEVICE=1 BRS_BENCH_BACKEND=gloo).  Not part of the measurement contract.

    python tools/multi_handle_bench.py [--handles 2] [--envs 65536] [--steps 60] [--env Env03-v2]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balance_robot_mujoco_rl_amd import BatchedSim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--handles", type=int, default=2); ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=60); ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--env", default="Env03-v2")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(a.handles)]
    sims, acts = [], []
    for h in range(a.handles):
        with torch.cuda.stream(streams[h]):
            s = BatchedSim(a.env, a.envs, device=0, seed=0, env_index_base=h * a.envs, auto_reset=True)
            s.reset()
            g = torch.Generator(device=dev); g.manual_seed(1234 + h)
            acts.append([(torch.rand((a.envs, 2), generator=g, device=dev) * 2 - 1).contiguous() for _ in range(8)])
            sims.append(s)
    torch.cuda.synchronize(dev)

    def run(k0, k1):
        for k in range(k0, k1):
            for h in range(a.handles):
                with torch.cuda.stream(streams[h]):   # BatchedSim launches on torch's CURRENT stream
                    sims[h].step(acts[h][k % 8])

    run(0, a.warmup)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(a.warmup, a.warmup + a.steps)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(tool="multi_handle_bench", env=a.env, handles=a.handles, envs_per_handle=a.envs, steps=a.steps,
                          ms_per_round=1e3 * dt / a.steps, env_steps_per_s=a.handles * a.envs * a.steps / dt)))
    for s in sims:
        s.close()


if __name__ == "__main__":
    main()

"""worker of tests/test_vec_env.py::test_two_process_gloo_shards_match_single_process (gloo, CPU)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd.vec_env import shard_ranges  # noqa: E402
from tests.fake_backend import OracleSim  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, steps = 10, 6
    start, cnt = shard_ranges(n, world)[rank]
    sim = OracleSim("Env03-v2", cnt, seed=21, env_index_base=start, max_episode_steps=4)
    sim.reset()
    rng = np.random.default_rng(5)
    rows = []
    for _ in range(steps):
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)  # every rank draws the global batch, uses its slice
        rows.append(sim.step(act[start:start + cnt])[0])           # no collective on the step path
    mine = torch.from_numpy(np.stack(rows))                        # [steps, cnt, 6]
    parts = [torch.zeros((steps, c, 6)) for _, c in shard_ranges(n, world)]
    dist.all_gather(parts, mine)                                   # the only exchange: concatenation of rollout tensors
    if rank == 0:
        np.save(sys.argv[1], torch.cat(parts, dim=1).numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

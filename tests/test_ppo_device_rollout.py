"""BASELINE config 5 in small: PPO on the HIP path end to end -- brs_policy_act / brs_step /
brs_rollout_bootstrap / brs_gae collect the rollouts on the GPU, torch only takes the gradient step -- trains the first
stage of the reference's curriculum (README.md:54-60, src/sb_rl.py:519-556: Env01-v2, the noisy-observation id) for one
minute and the deterministic policy then keeps the robots on their wheels.  The full two-stage recipe and its result on
Env03-v2: tools/config5_recipe.sh -> profiles/r03_ppo_config5.json."""
import os
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_stage1_of_the_curriculum_trains_to_balance_with_device_rollouts():
    import torch
    import train_ppo_torch as T
    from balance_robot_mujoco_rl_amd import BatchedSim
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = T.ActorCritic(-0.5).to(dev)
    with torch.no_grad():   # obs[1] of Env01-v2 carries +-10 rad/s of injected noise (tools/config5_recipe.sh)
        sc = torch.tensor([1, 0.02, 1, 1, 1, 1], device=dev)
        model.pi[0].weight.mul_(sc); model.v[0].weight.mul_(sc)
    torch.manual_seed(1000)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    n = 16384
    sim = BatchedSim("Env01-v2", n, device=0, seed=0, auto_reset=True)
    log = []
    t0 = time.time()
    T.train(sim, model, opt, iters=80, n_steps=64, epochs=4, minibatch=8192, gamma=0.999, lam=0.95, clip=0.2, log=log, tag="Env01-v2",
            reward_clip=1.0, device_rollout=True, seed=1000)
    sim.close()
    wall = time.time() - t0
    before = T.evaluate("Env01-v2", T.ActorCritic(-0.5).to(dev), 2048, 300)     # an untrained policy: nobody lasts
    after = T.evaluate("Env01-v2", model, 2048, 600)                             # 3 s of simulated time
    print(f"stage 1 on the HIP path: {wall:.0f} s, {log[-1]['env_steps']} env-steps; untrained: {before['first_episode_still_running']} of 2048 "
          f"still up after 300 steps; trained: {after['first_episode_still_running']} of 2048 still up after 600 steps, {after['fell']} falls")
    assert before["first_episode_still_running"] < 0.05 * 2048
    # ~15 % of Env01-v2's resets start beyond the 50-degree limit (envs/env01_v2.py:52-71: pitch drawn in +-1 rad) and end on
    # their first step whatever the policy does; of the rest the trained policy must keep most on their wheels
    assert after["first_episode_still_running"] > 0.2 * 2048, after   # measured 802 (80 iterations, 78 s); 595 after 55 iterations
    assert wall < 150, wall

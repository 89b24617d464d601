"""Host logic of BalanceVecEnv (SB3 VecEnv contract) with oracle-backed stand-in simulators, and the
multi-shard path: contiguous env-index ranges, per-env Philox streams keyed by the GLOBAL index, no collective on the
step path.  CPU only."""
import os
import subprocess
import sys

import numpy as np

from balance_robot_mujoco_rl_amd.registry import ENV_SPECS, spec
from balance_robot_mujoco_rl_amd.vec_env import BalanceVecEnv, shard_ranges
from tests.fake_backend import OracleSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_matches_reference(golden):
    reg = golden["registry"]
    for k, s in ENV_SPECS.items():
        assert reg[k]["max_episode_steps"] == s.max_episode_steps and reg[k]["reward_threshold"] == s.reward_threshold
    assert spec("Env01-v2").obs_noise and not spec("Env03-v2").obs_noise  # Env03_v2 does not inherit Env01_v2's noise
    assert sorted(ENV_SPECS) == ["Env01-v1", "Env01-v2", "Env01-v3", "Env02-v1", "Env03-v1", "Env03-v2"]


def test_shard_ranges():
    assert shard_ranges(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert shard_ranges(524288, 8) == [(i * 65536, 65536) for i in range(8)]
    assert shard_ranges(2, 4) == [(0, 1), (1, 1)]


def _make(n, shards, seed=4, max_episode_steps=6):
    sims = [OracleSim("Env03-v2", cnt, seed=seed, env_index_base=start, max_episode_steps=max_episode_steps)
            for start, cnt in shard_ranges(n, shards)]
    return BalanceVecEnv("Env03-v2", n, _sims=sims)


def test_vecenv_contract_and_autoreset_infos():
    n = 6
    env = _make(n, 1)
    assert env.num_envs == n and env.observation_space.shape == (6,) and env.action_space.shape == (2,)
    assert env.get_attr("render_mode") == [None] * n and env.env_is_wrapped(object) == [False] * n
    obs = env.reset()
    assert obs.shape == (n, 6) and obs.dtype == np.float32 and (obs[:, 1] == 0).all()
    rng = np.random.default_rng(0)
    ret = np.zeros(n)
    seen_done = False
    for t in range(14):
        env.step_async(rng.uniform(-1, 1, size=(n, 2)))
        obs, rew, dones, infos = env.step_wait()
        assert obs.shape == (n, 6) and rew.shape == (n,) and dones.dtype == bool and len(infos) == n
        ret += rew
        for i in range(n):
            if dones[i]:
                seen_done = True
                info = infos[i]
                assert set(info) == {"terminal_observation", "TimeLimit.truncated", "episode"}
                assert info["terminal_observation"].shape == (6,)
                assert abs(info["episode"]["r"] - ret[i]) < 1e-4 and 1 <= info["episode"]["l"] <= 6
                assert obs[i, 1] == 0.0, "row of a done env is the first observation of its new episode"
                if info["episode"]["l"] == 6 and info["TimeLimit.truncated"]:
                    pass
                ret[i] = 0
            else:
                assert infos[i] == {}
    assert seen_done
    env.close()


def test_sharding_is_invisible():
    """1 shard vs 3 shards: identical observations, rewards, dones -- an env's stream depends on its global index only"""
    n = 7
    a, b = _make(n, 1), _make(n, 3)
    np.testing.assert_array_equal(a.reset(), b.reset())
    rng = np.random.default_rng(3)
    for _ in range(10):
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act)
        np.testing.assert_array_equal(oa, ob); np.testing.assert_array_equal(ra, rb); np.testing.assert_array_equal(da, db)
    a.close(); b.close()


def test_two_process_gloo_shards_match_single_process(tmp_path):
    """the N > 1 launch path (one process per shard, torch.distributed rendezvous on 127.0.0.1, no collective on the
    step path, one gather of rollout tensors at the end) against one process holding all envs"""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    out = tmp_path / "gathered.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", script, str(out)], env=env, timeout=300)
    gathered = np.load(out)
    from oracle import oracle as O
    n, steps = 10, 6
    o = O.Oracle("Env03-v2", n, seed=21, auto_reset=True, max_episode_steps=4)
    o.reset()
    rng = np.random.default_rng(5)
    ref = []
    for _ in range(steps):
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        ref.append(o.step(act)[0])
    np.testing.assert_array_equal(gathered, np.stack(ref))


def test_gymnasium_style_vector_adapter():
    from balance_robot_mujoco_rl_amd.vec_env import BalanceVectorEnv
    n = 6
    sims = [OracleSim("Env03-v2", cnt, seed=1, env_index_base=start, max_episode_steps=5) for start, cnt in shard_ranges(n, 2)]
    env = BalanceVectorEnv("Env03-v2", n, _sims=sims)
    obs, info = env.reset(seed=0)
    assert obs.shape == (n, 6) and info == {}
    for t in range(5):
        obs, rew, term, trunc, infos = env.step(np.zeros((n, 2)))
    assert trunc.all() and "final_observation" in infos and infos["_final_observation"].all()
    assert infos["final_observation"][0].shape == (6,) and (obs[:, 1] == 0).all()
    env.close()

"""Stand-ins with BatchedSim's surface for CPU tests of the host logic (tests only): the oracle plays the simulator."""
import numpy as np

from oracle import oracle as O


class OracleSim:
    def __init__(self, env_id, n, seed=0, env_index_base=0, max_episode_steps=0, obs_noise=None):
        self.o = O.Oracle(env_id, n, seed=seed, env_index_base=env_index_base, auto_reset=True,
                          max_episode_steps=max_episode_steps, noise=obs_noise)
        self.n = n

    def reset(self, mask=None):
        return self.o.reset(mask)

    def step(self, actions):
        obs, rew, te, tr, tob = self.o.step(np.asarray(actions, np.float32))
        return obs, rew, te.astype(np.uint8), tr.astype(np.uint8), tob

    def close(self):
        self.o.close()

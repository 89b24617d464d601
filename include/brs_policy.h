/*
 * brs_policy.h -- C ABI of the on-device rollout side of the path: the caller of the env
 * step in the reference is Stable-Baselines3's PPO with "MlpPolicy" (src/sb_rl.py:63-71) driven by model.learn
 * (src/sb_rl.py:552-556).  Per env step SB3 runs, in Python/torch on the host [3P stable_baselines3]:
 *
 *   brs_policy_act        ActorCriticPolicy.forward(obs): mlp_extractor (separate 6-64-64 tanh towers for pi and vf),
 *                         action_net, value_net, DiagGaussianDistribution.sample() / log_prob(), then the clip of the
 *                         action to the Box before env.step (OnPolicyAlgorithm.collect_rollouts)
 *   brs_rollout_bootstrap collect_rollouts' time-limit handling: rewards[i] += gamma * V(terminal_observation[i]) for
 *                         envs whose episode was truncated, not terminated
 *   brs_gae               RolloutBuffer.compute_returns_and_advantage (GAE(lambda) over the [T][N] buffer)
 *
 * These entry points do the same arithmetic on the GPU, reading the simulator's outputs in place (device pointers),
 * so that a rollout of 65,536 envs needs no per-env Python and no PCIe traffic.  All buffers are DEVICE pointers owned
 * by the caller; every call only enqueues work on `stream`.  Same library (libbrs_hip.so), same status codes as brs.h.
 *
 * Parameter vector (host floats, brs_policy_set_weights), torch.nn.Linear layout weight[out][in]:
 *   pi: W1[64][6] b1[64] W2[64][64] b2[64] W3[2][64] b3[2]   (mlp_extractor.policy_net.0/.2, action_net)
 *   vf: W1[64][6] b1[64] W2[64][64] b2[64] W3[1][64] b3[1]   (mlp_extractor.value_net.0/.2, value_net)
 *   log_std[2]
 * Noise: z = Box-Muller of Philox4x32-10(counter = (step, 0x504f4c49 "POLI", gid_lo, gid_hi), key = seed), one block per
 * env and step, gid = env_index_base + i: independent of how envs are sharded over GPUs, disjoint from the simulator's
 * streams (whose second counter word is 0).
 */
#ifndef BRS_POLICY_H
#define BRS_POLICY_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRS_POLICY_OBS 6
#define BRS_POLICY_HID 64
#define BRS_POLICY_ACT 2
#define BRS_POLICY_NPI (64 * 6 + 64 + 64 * 64 + 64 + 2 * 64 + 2)
#define BRS_POLICY_NVF (64 * 6 + 64 + 64 * 64 + 64 + 1 * 64 + 1)
#define BRS_POLICY_NPARAM (BRS_POLICY_NPI + BRS_POLICY_NVF + 2)

typedef struct brs_policy brs_policy;

int brs_policy_create(int32_t device, brs_policy** out);
int brs_policy_destroy(brs_policy*);
const char* brs_policy_last_error(const brs_policy*);
/* copy BRS_POLICY_NPARAM host floats to the device (synchronous; once per optimiser phase, not per step) */
int brs_policy_set_weights(brs_policy*, const float* params_host);
/* or point at a device-resident parameter vector the learner updates in place (no copy; must stay alive) */
int brs_policy_use_device_weights(brs_policy*, const float* params_dev);

/* one policy step for n envs: action[n][2] (unclipped sample, what the rollout buffer stores), action_clipped[n][2]
 * (clipped to [-1, 1], what brs_step consumes), logp[n], value[n]; noise[n][2] (the standard normals used) may be NULL.
 * deterministic != 0: action = mean (SB3 predict(deterministic=True)); logp is then that of the mean. */
int brs_policy_act(brs_policy*, int32_t n, const float* obs_dev, uint64_t seed, int64_t env_index_base, uint32_t step,
                   int32_t deterministic, float* action_dev, float* action_clipped_dev, float* logp_dev, float* value_dev,
                   float* noise_dev, void* stream);
/* value head only (e.g. last_values of a rollout) */
int brs_policy_value(brs_policy*, int32_t n, const float* obs_dev, float* value_dev, void* stream);
/* reward[i] += gamma * V(terminal_obs[i]) where truncated[i] && !terminated[i] */
int brs_rollout_bootstrap(brs_policy*, int32_t n, const float* terminal_obs_dev, const uint8_t* terminated_dev,
                          const uint8_t* truncated_dev, float gamma, float* reward_dev, void* stream);
/* GAE(lambda) over a [T][N] rollout: episode_start[t][i] != 0 marks the first step of an episode (SB3's
 * episode_starts); last_value[N] / last_done[N] close the recursion after the final step.  adv and ret are [T][N]. */
int brs_gae(int32_t device, int32_t T, int32_t N, const float* reward_dev, const float* value_dev,
            const uint8_t* episode_start_dev, const float* last_value_dev, const uint8_t* last_done_dev, float gamma,
            float lam, float* adv_dev, float* ret_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif

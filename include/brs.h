/*
 * brs.h -- C ABI of the MI355X-native batched balance-robot simulator (libbrs_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (lachlanhurst/balance-robot-mujoco-rl, paths under src/balance_robot/) has no FFI of its own: its
 * hot path is the Gymnasium Env API of the registered ids, implemented as Python calling the MuJoCo C
 * library.  Each entry point below names the reference interface it replaces:
 *
 *   brs_create      gym.make(id)                        __init__.py:5-52, sb_rl.py:500; model load in
 *                                                       envs/RobotBaseEnv.py:56-65 (MujocoEnv.__init__)
 *   brs_reset       Env.reset() -> reset_model()        envs/env01_v2.py:52-71, envs/env03_v1.py:60-83
 *   brs_step        Env.step(a) for N envs at once      envs/env01_v2.py:28-50, envs/env03_v1.py:26-58
 *                   (reward, ctrl law, mj_step x250,    (+ gymnasium TimeLimit, __init__.py:15,50, and the
 *                   block state machine, termination,   SB3 VecEnv auto-reset contract)
 *                   observation, time limit, auto-reset)
 *   brs_get_state / MujocoEnv.data.qpos/qvel/time,      gymnasium MujocoEnv.set_state (used at
 *   brs_set_state   set_state(qpos, qvel)               envs/env01_v2.py:70)
 *   brs_destroy     Env.close()
 *
 * Conventions
 *   - every function returns 0 on success, a negative brs_status otherwise; brs_last_error() gives text.
 *   - I/O buffers of brs_reset / brs_step are DEVICE pointers owned by the caller (e.g. torch tensors on
 *     the handle's device); brs_get_* / brs_set_* take HOST pointers and synchronise (not on the hot path).
 *   - brs_step / brs_reset only enqueue work on `stream` (a hipStream_t, NULL = default stream); no hidden sync,
 *     no allocation.
 *   - a handle is not thread-safe; distinct handles (one per GPU) are independent.
 *   - there is NO CPU fallback: without a usable HIP device brs_create fails with BRS_ERR_HIP.
 */
#ifndef BRS_H
#define BRS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct brs_handle brs_handle;

typedef enum { BRS_ENV01_V1 = 0, BRS_ENV01_V2 = 1, BRS_ENV03_V1 = 2, BRS_ENV03_V2 = 3,
               BRS_ENV01_V3 = 4, /* envs/env01_v3.py: target-speed schedule, pitch offset, shaped reward */
               BRS_ENV02_V1 = 5  /* envs/env02_v1.py: wheel/floor friction U(0.5,1) drawn per episode */ } brs_variant;

typedef enum {
  BRS_OK = 0,
  BRS_ERR_ARG = -1,     /* bad argument */
  BRS_ERR_HIP = -2,     /* HIP runtime error (no device, launch failure, ...) */
  BRS_ERR_STATE = -3    /* handle not initialised / destroyed */
} brs_status;

enum {
  BRS_FLAG_AUTO_RESET = 1u, /* SB3 VecEnv semantics: a done env is reset inside brs_step, obs = first obs of the new episode */
  BRS_FLAG_NOISE_ON = 2u,   /* force observation noise on  (default: on for Env01-v2 only, as in the reference) */
  BRS_FLAG_NOISE_OFF = 4u,  /* force observation noise off */
  BRS_FLAG_NO_LANE_GROUPING = 8u /* keep env i on lane i (default: brs_step regroups the envs of a handle by collision cost
                                    class after every step -- a scheduling matter only, results are bit-identical) */
};

typedef struct {
  int32_t variant;            /* brs_variant */
  int32_t num_envs;           /* N > 0 */
  int32_t device;             /* HIP device ordinal */
  uint32_t flags;             /* BRS_FLAG_* */
  uint64_t seed;              /* Philox key; env i draws from stream (seed, env_index_base + i): results do not
                                 depend on how envs are sharded over GPUs */
  int64_t env_index_base;     /* global index of env 0 of this handle */
  int32_t max_episode_steps;  /* 0 = reference default (6000; 1200 for Env03-v2) */
  int32_t substeps;           /* 0 = 250 (frame_skip, envs/RobotBaseEnv.py:59) */
  double timestep;            /* 0 = 2e-5 (envs/env01_v1.xml:3) */
  int32_t block_threads;      /* 0 = default (64): threads per workgroup, multiple of 64 */
  int32_t reserved;
} brs_config;

int brs_create(const brs_config* cfg, brs_handle** out);
int brs_destroy(brs_handle* h);
const char* brs_last_error(const brs_handle* h); /* h may be NULL: error of the last failed brs_create */

/* sizes of the model behind a variant: nq = 9 / 16, nv = 8 / 14, obs = 6, act = 2 */
int brs_sizes(int32_t variant, int32_t* nq, int32_t* nv, int32_t* nobs, int32_t* nact);

/* reset envs whose mask byte is non-zero (mask NULL = all); obs_dev [N][6] f32 gets the reset observation of the
 * envs that were reset (other rows untouched). */
int brs_reset(brs_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* one env step of all N envs.
 *   actions_dev      [N][2] f32  (not clipped by the env, like the reference; ctrl = wheel speed + 4*a)
 *   obs_dev          [N][6] f32  observation after the step (after auto-reset: first observation of the new episode)
 *   reward_dev       [N]    f32  reward of the step (computed on the pre-step state, like the reference)
 *   terminated_dev   [N]    u8   |pitch| > 50 deg
 *   truncated_dev    [N]    u8   elapsed_steps >= max_episode_steps
 *   terminal_obs_dev [N][6] f32  observation before any auto-reset (SB3 "terminal_observation"); may be NULL */
int brs_step(brs_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* terminated_dev,
             uint8_t* truncated_dev, float* terminal_obs_dev, void* stream);

/* advance physics only (no env logic): nsub substeps with ctrl_dev [N][2] f32 held.  For parity tests. */
int brs_physics(brs_handle* h, const float* ctrl_dev, int32_t nsub, void* stream);

/* HOST pointers, row-major: qpos [N][nq] f64 (free joint: pos3 + quat wxyz, wheels, [block pos3 + quat]),
 * qvel [N][nv] f64 (MuJoCo convention: world linear, body-frame angular), warm [N][nv] f64 (qacc_warmstart),
 * time [N] f64.  Any pointer may be NULL.  set_state also refreshes the accessor pose (mj_forward). */
int brs_get_state(brs_handle* h, double* qpos, double* qvel, double* warm, double* time);
int brs_set_state(brs_handle* h, const double* qpos, const double* qvel, const double* warm, const double* time);
/* aux [N][14] f64: last_pitch, block_timer (NaN = None), elapsed_steps, rng_ctr, attack_side_front,
 * accessor pitch (read-only), episode return, bad-state count (read-only), 0, 0, wheel/floor friction (Env02),
 * delay_target_speed, pitch_offset, target_wheel_speed (Env01-v3) */
int brs_get_aux(brs_handle* h, double* aux);
int brs_set_aux(brs_handle* h, const double* aux);
/* accessor pose = data.body("robot_body").xquat [N][4] / .xpos [N][3] as the reference's get_pitch()/get_yaw() read it */
int brs_get_xpose(brs_handle* h, double* xquat, double* xpos);
int brs_set_xpose(brs_handle* h, const double* xquat, const double* xpos);

/* algorithmic HBM bytes one brs_step moves per env (state in + state out + action + outputs), for rooflines */
int64_t brs_step_bytes_per_env(const brs_handle* h);
/* name of the step kernel (for matching rocprof rows) */
const char* brs_step_kernel_name(const brs_handle* h);
/* identity of this BUILD: hash of the kernel sources and compile flags it was made from (set by the build, "unstamped" if
 * compiled by hand).  bench.py reports profile-derived counters only when they carry the id of the library that ran. */
const char* brs_build_id(void);

#ifdef __cplusplus
}
#endif
#endif

// brs_policy.hip -- on-device rollout side of the path (include/brs_policy.h): SB3 MlpPolicy
// actor/critic forward + diagonal-Gaussian sample, time-limit bootstrap, GAE(lambda), as HIP kernels for gfx950.
//
// Mapping: one wavefront LANE per env (like the step kernel).  The 6-64-64 towers are ~9.2 k FMAs per env and the
// weights are the same for every lane: all weight indices are compile-time constants off a kernel-argument pointer, so
// the compiler fetches them with scalar loads (s_load_dwordx8/x16 through the scalar cache) and feeds them to
// v_fmac_f32 as SGPR operands -- no LDS staging, no per-lane weight traffic.  Hidden activations stay in VGPRs (2 x 64),
// fully unrolled.  65,536 envs = 1,024 waves x ~18 k VALU instructions = tens of microseconds per policy step, < 1 % of
// the env step it feeds; MFMA would shave microseconds off a path that is not the bottleneck, fp32 VALU keeps the
// result within rounding of the fp32 torch reference the parity test compares with.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/brs.h"
#include "../../include/brs_policy.h"
#include "brs_core.hpp"  // philox4x32_10 (same generator as the simulator)

namespace {

constexpr int OBS = BRS_POLICY_OBS, HID = BRS_POLICY_HID, ACT = BRS_POLICY_ACT;
constexpr int OFF_PI = 0, OFF_VF = BRS_POLICY_NPI, OFF_LOGSTD = BRS_POLICY_NPI + BRS_POLICY_NVF;

// one 6-64-64-NOUT tanh tower; w: W1[64][6] b1[64] W2[64][64] b2[64] W3[NOUT][64] b3[NOUT] (uniform pointer)
template <int NOUT> __device__ __forceinline__ void tower(const float* __restrict__ w, const float* x, float* out) {
  const float* W1 = w;
  const float* b1 = W1 + HID * OBS;
  const float* W2 = b1 + HID;
  const float* b2 = W2 + HID * HID;
  const float* W3 = b2 + HID;
  const float* b3 = W3 + NOUT * HID;
  float h1[HID], h2[HID];
#pragma unroll
  for (int j = 0; j < HID; j++) {
    float a = b1[j];
#pragma unroll
    for (int i = 0; i < OBS; i++) a = fmaf(W1[j * OBS + i], x[i], a);
    h1[j] = tanhf(a);
  }
#pragma unroll
  for (int j = 0; j < HID; j++) {
    float a = b2[j];
#pragma unroll
    for (int i = 0; i < HID; i++) a = fmaf(W2[j * HID + i], h1[i], a);
    h2[j] = tanhf(a);
  }
#pragma unroll
  for (int k = 0; k < NOUT; k++) {
    float a = b3[k];
#pragma unroll
    for (int i = 0; i < HID; i++) a = fmaf(W3[k * HID + i], h2[i], a);
    out[k] = a;
  }
}

__global__ void __launch_bounds__(64) policy_act_kernel(const float* __restrict__ w, const int n, const float* __restrict__ obs,
                                                        const uint64_t seed, const int64_t gid_base, const uint32_t step,
                                                        const int deterministic, float* __restrict__ action,
                                                        float* __restrict__ action_clipped, float* __restrict__ logp,
                                                        float* __restrict__ value, float* __restrict__ noise) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[OBS];
#pragma unroll
  for (int k = 0; k < OBS; k++) x[k] = obs[(size_t)OBS * i + k];
  float mean[ACT], v[1];
  tower<ACT>(w + OFF_PI, x, mean);
  tower<1>(w + OFF_VF, x, v);
  float z[ACT] = {0.0f, 0.0f};
  if (!deterministic) {
    const int64_t gid = gid_base + (int64_t)i;
    uint32_t o[4];
    brs::philox4x32_10(step, 0x504f4c49u, (uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32),
                       (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), o);
    // Box-Muller on two 24-bit uniforms in (0, 1)
    const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1)), th = 6.283185307179586f * u2;
    z[0] = r * cosf(th); z[1] = r * sinf(th);
  }
  float lp = 0.0f;
#pragma unroll
  for (int k = 0; k < ACT; k++) {
    const float ls = w[OFF_LOGSTD + k];
    const float a = fmaf(expf(ls), z[k], mean[k]);
    action[(size_t)ACT * i + k] = a;
    action_clipped[(size_t)ACT * i + k] = fminf(1.0f, fmaxf(-1.0f, a));
    lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;  // log N(a; mean, exp(ls)) with (a - mean) / sigma = z
    if (noise) noise[(size_t)ACT * i + k] = z[k];
  }
  logp[i] = lp;
  value[i] = v[0];
}

__global__ void __launch_bounds__(64) policy_value_kernel(const float* __restrict__ w, const int n, const float* __restrict__ obs,
                                                          float* __restrict__ value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[OBS], v[1];
#pragma unroll
  for (int k = 0; k < OBS; k++) x[k] = obs[(size_t)OBS * i + k];
  tower<1>(w + OFF_VF, x, v);
  value[i] = v[0];
}

__global__ void __launch_bounds__(64) bootstrap_kernel(const float* __restrict__ w, const int n, const float* __restrict__ tobs,
                                                       const uint8_t* __restrict__ term, const uint8_t* __restrict__ trunc,
                                                       const float gamma, float* __restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!(trunc[i] && !term[i])) return;  // a wave without a truncated episode leaves at once (the common case)
  float x[OBS], v[1];
#pragma unroll
  for (int k = 0; k < OBS; k++) x[k] = tobs[(size_t)OBS * i + k];
  tower<1>(w + OFF_VF, x, v);
  reward[i] = fmaf(gamma, v[0], reward[i]);
}

// GAE(lambda): one lane per env walks its column of the [T][N] buffers backwards; every access is coalesced over envs
__global__ void __launch_bounds__(256) gae_kernel(const int T, const int N, const float* __restrict__ reward,
                                                  const float* __restrict__ value, const uint8_t* __restrict__ episode_start,
                                                  const float* __restrict__ last_value, const uint8_t* __restrict__ last_done,
                                                  const float gamma, const float lam, float* __restrict__ adv, float* __restrict__ ret) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float next_value = last_value[i], next_nonterminal = last_done[i] ? 0.0f : 1.0f, gae = 0.0f;
  for (int t = T - 1; t >= 0; t--) {
    const size_t k = (size_t)t * N + i;
    const float v = value[k];
    const float delta = reward[k] + gamma * next_value * next_nonterminal - v;
    gae = delta + gamma * lam * next_nonterminal * gae;
    adv[k] = gae;
    ret[k] = gae + v;
    next_value = v;
    next_nonterminal = episode_start[k] ? 0.0f : 1.0f;
  }
}

}  // namespace

struct brs_policy {
  int device = 0;
  float* w_own = nullptr;  // BRS_POLICY_NPARAM floats owned by the handle
  const float* w = nullptr;
  std::string err;
};

namespace {
thread_local std::string g_policy_create_error;
int pfail(brs_policy* p, int code, const std::string& m) {
  if (p) p->err = m; else g_policy_create_error = m;
  return code;
}
struct PGuard {
  int prev = -1;
  bool ok = true;
  explicit PGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~PGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define BRS_P_TRY(p, expr)                                                                                 \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return pfail(p, BRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
}  // namespace

extern "C" {

int brs_policy_create(int32_t device, brs_policy** out) {
  if (!out) return pfail(nullptr, BRS_ERR_ARG, "brs_policy_create: null argument");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return pfail(nullptr, BRS_ERR_HIP, std::string("brs_policy_create: no HIP device (") + hipGetErrorString(e) + "); there is no CPU fallback");
  if (device < 0 || device >= ndev) return pfail(nullptr, BRS_ERR_ARG, "brs_policy_create: device ordinal out of range");
  brs_policy* p = new brs_policy();
  p->device = device;
  PGuard g(device);
  if (!g.ok || hipMalloc(&p->w_own, BRS_POLICY_NPARAM * sizeof(float)) != hipSuccess ||
      hipMemset(p->w_own, 0, BRS_POLICY_NPARAM * sizeof(float)) != hipSuccess) {
    delete p;
    return pfail(nullptr, BRS_ERR_HIP, "brs_policy_create: device allocation failed");
  }
  p->w = p->w_own;
  *out = p;
  return BRS_OK;
}

int brs_policy_destroy(brs_policy* p) {
  if (!p) return BRS_ERR_STATE;
  {
    PGuard g(p->device);
    if (p->w_own) (void)hipFree(p->w_own);
  }
  delete p;
  return BRS_OK;
}

const char* brs_policy_last_error(const brs_policy* p) { return p ? p->err.c_str() : g_policy_create_error.c_str(); }

int brs_policy_set_weights(brs_policy* p, const float* params_host) {
  if (!p) return BRS_ERR_STATE;
  if (!params_host) return pfail(p, BRS_ERR_ARG, "brs_policy_set_weights: null pointer");
  PGuard g(p->device);
  BRS_P_TRY(p, hipMemcpy(p->w_own, params_host, BRS_POLICY_NPARAM * sizeof(float), hipMemcpyHostToDevice));
  p->w = p->w_own;
  return BRS_OK;
}

int brs_policy_use_device_weights(brs_policy* p, const float* params_dev) {
  if (!p) return BRS_ERR_STATE;
  if (!params_dev) return pfail(p, BRS_ERR_ARG, "brs_policy_use_device_weights: null pointer");
  p->w = params_dev;
  return BRS_OK;
}

int brs_policy_act(brs_policy* p, int32_t n, const float* obs_dev, uint64_t seed, int64_t env_index_base, uint32_t step,
                   int32_t deterministic, float* action_dev, float* action_clipped_dev, float* logp_dev, float* value_dev,
                   float* noise_dev, void* stream) {
  if (!p) return BRS_ERR_STATE;
  if (n <= 0 || !obs_dev || !action_dev || !action_clipped_dev || !logp_dev || !value_dev)
    return pfail(p, BRS_ERR_ARG, "brs_policy_act: bad argument");
  PGuard g(p->device);
  hipLaunchKernelGGL(policy_act_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, p->w, n, obs_dev, seed, env_index_base,
                     step, deterministic, action_dev, action_clipped_dev, logp_dev, value_dev, noise_dev);
  BRS_P_TRY(p, hipGetLastError());
  return BRS_OK;
}

int brs_policy_value(brs_policy* p, int32_t n, const float* obs_dev, float* value_dev, void* stream) {
  if (!p) return BRS_ERR_STATE;
  if (n <= 0 || !obs_dev || !value_dev) return pfail(p, BRS_ERR_ARG, "brs_policy_value: bad argument");
  PGuard g(p->device);
  hipLaunchKernelGGL(policy_value_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, p->w, n, obs_dev, value_dev);
  BRS_P_TRY(p, hipGetLastError());
  return BRS_OK;
}

int brs_rollout_bootstrap(brs_policy* p, int32_t n, const float* terminal_obs_dev, const uint8_t* terminated_dev,
                          const uint8_t* truncated_dev, float gamma, float* reward_dev, void* stream) {
  if (!p) return BRS_ERR_STATE;
  if (n <= 0 || !terminal_obs_dev || !terminated_dev || !truncated_dev || !reward_dev)
    return pfail(p, BRS_ERR_ARG, "brs_rollout_bootstrap: bad argument");
  PGuard g(p->device);
  hipLaunchKernelGGL(bootstrap_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, p->w, n, terminal_obs_dev, terminated_dev,
                     truncated_dev, gamma, reward_dev);
  BRS_P_TRY(p, hipGetLastError());
  return BRS_OK;
}

int brs_gae(int32_t device, int32_t T, int32_t N, const float* reward_dev, const float* value_dev, const uint8_t* episode_start_dev,
            const float* last_value_dev, const uint8_t* last_done_dev, float gamma, float lam, float* adv_dev, float* ret_dev,
            void* stream) {
  if (T <= 0 || N <= 0 || !reward_dev || !value_dev || !episode_start_dev || !last_value_dev || !last_done_dev || !adv_dev || !ret_dev)
    return BRS_ERR_ARG;
  PGuard g(device);
  if (!g.ok) return BRS_ERR_HIP;
  hipLaunchKernelGGL(gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, reward_dev, value_dev, episode_start_dev,
                     last_value_dev, last_done_dev, gamma, lam, adv_dev, ret_dev);
  return hipGetLastError() == hipSuccess ? BRS_OK : BRS_ERR_HIP;
}

}  // extern "C"

// brs_policy.hip -- on-device rollout side of the path (include/brs_policy.h): SB3 MlpPolicy
// actor/critic forward + diagonal-Gaussian sample, time-limit bootstrap, GAE(lambda), as HIP kernels for gfx950.
//
// The 6-64-64 towers are the one dense contraction of the repository (2 x 64 x 64 MACs per env and layer) and run on the
// MATRIX cores in fp32: v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation (the parity test compares with fp32
// torch at rtol 1e-5).  Mapping: a wave owns 64 envs = two N-tiles of 32; the 64 units of a layer are two M-tiles; the
// product is computed TRANSPOSED, D[unit][env] = sum_k W[unit][k] act[k][env], so that
//   * the A operand is a weight (lane l: W[32 mt + l%32][k(l/32)]), read from a padded LDS copy of the tower (row stride 65 /
//     7 words: the 32 lanes of a half hit 32 different banks), staged once per 256-env workgroup;
//   * the B operand is an activation of env l%32 -- and the accumulator layout of this instruction (lane l holds rows
//     8 (r/4) + 4 (l/32) + r%4 of column l%32) is exactly "16 units of MY env per M-tile": after the tanh the output
//     registers of one layer ARE the B operands of the next (the K index is walked in accumulator order, the weights are
//     fetched to match), so activations never leave their registers: no LDS round trip, no shuffle between the layers.
// The 2- and 1-unit output layers are 32 FMAs per lane plus one cross-half shuffle.  tanh = 1 - 2 / (2^(2 x log2 e) + 1) on
// v_exp_f32 / v_rcp_f32 (absolute error ~1e-7).  Round 2's kernel walked the towers lane by lane with the weights as scalar
// operands (2,097 s_load per wave, latency-bound): 115 us per policy step at 65,536 envs = 2.5 % of the env step it feeds
// (profiles/r03_rollout_kernel_stats_before_mfma.json); this one: profiles/r03_rollout_kernel_stats.json.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/brs.h"
#include "../../include/brs_policy.h"
#include "brs_core.hpp"  // philox4x32_10 (same generator as the simulator)

namespace {

constexpr int OBS = BRS_POLICY_OBS, HID = BRS_POLICY_HID, ACT = BRS_POLICY_ACT;
constexpr int OFF_PI = 0, OFF_VF = BRS_POLICY_NPI, OFF_LOGSTD = BRS_POLICY_NPI + BRS_POLICY_NVF;
static_assert(OBS == 6 && HID == 64, "the MFMA tiling below is written for the 6-64-64 MlpPolicy");

typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS image of one tower (floats): W2 [64][65], W1 [64][7], b1 [64], b2 [64], W3 [2][64], b3 [2]
constexpr int W2_LD = 65, W1_LD = 7;
constexpr int T_W2 = 0, T_W1 = T_W2 + HID * W2_LD, T_B1 = T_W1 + HID * W1_LD, T_B2 = T_B1 + HID, T_W3 = T_B2 + HID, T_B3 = T_W3 + 2 * HID,
              T_SIZE = T_B3 + 4;
constexpr int POLICY_THREADS = 256;  // 4 waves x 64 envs share one staged copy of the weights

// global parameter vector of a tower (W1[64][6] b1[64] W2[64][64] b2[64] W3[NOUT][64] b3[NOUT]) -> its LDS image; all threads
template <int NOUT> __device__ __forceinline__ void stage_tower(const float* __restrict__ w, float* __restrict__ L) {
  const float* W1 = w;
  const float* b1 = W1 + HID * OBS;
  const float* W2 = b1 + HID;
  const float* b2 = W2 + HID * HID;
  const float* W3 = b2 + HID;
  const float* b3 = W3 + NOUT * HID;
  for (int i = threadIdx.x; i < HID * HID; i += blockDim.x) L[T_W2 + (i >> 6) * W2_LD + (i & 63)] = W2[i];
  for (int i = threadIdx.x; i < HID * OBS; i += blockDim.x) L[T_W1 + (i / OBS) * W1_LD + (i % OBS)] = W1[i];
  for (int i = threadIdx.x; i < HID; i += blockDim.x) { L[T_B1 + i] = b1[i]; L[T_B2 + i] = b2[i]; }
  for (int i = threadIdx.x; i < NOUT * HID; i += blockDim.x) L[T_W3 + i] = W3[i];
  if (threadIdx.x < NOUT) L[T_B3 + threadIdx.x] = b3[threadIdx.x];
}

__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2 / (e^(2x) + 1): e = +inf -> 1, e = 0 -> -1; v_exp_f32 and v_rcp_f32 are 1-ulp instructions
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// One tower for the 64 envs of this wave.  x[nt][k]: observation k of env (32 nt + lane % 32) of the wave (both halves of
// the wave hold the same rows); out[nt][k]: output unit k for that env, complete in both halves.
template <int NOUT> __device__ __forceinline__ void tower_mfma(const float* __restrict__ L, const float (&x)[2][OBS], float (&out)[2][NOUT]) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  // unit held by accumulator register r of M-tile mt in this half of the wave: 32 mt + 8 (r / 4) + 4 h + r % 4
#define BRS_UNIT(mt, r) (32 * (mt) + 8 * ((r) >> 2) + 4 * h + ((r) & 3))
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) { const float b = L[T_B1 + BRS_UNIT(mt, r)]; acc[mt][0][r] = b; acc[mt][1][r] = b; }
  // layer 1: K = 6 = three steps of two; this half supplies feature 2 s + h
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const float a0 = L[T_W1 + c * W1_LD + 2 * s + h], a1 = L[T_W1 + (32 + c) * W1_LD + 2 * s + h];
    const float b0 = h ? x[0][2 * s + 1] : x[0][2 * s], b1 = h ? x[1][2 * s + 1] : x[1][2 * s];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  f32x16 h1[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) h1[mt][nt][r] = fast_tanh(acc[mt][nt][r]);
  // layer 2: K = 64 walked in ACCUMULATOR order: step (mtp, r) contracts the two units BRS_UNIT(mtp, r) of the two halves
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) { const float b = L[T_B2 + BRS_UNIT(mt, r)]; acc[mt][0][r] = b; acc[mt][1][r] = b; }
#pragma unroll
  for (int mtp = 0; mtp < 2; mtp++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kin = BRS_UNIT(mtp, r);
      const float a0 = L[T_W2 + c * W2_LD + kin], a1 = L[T_W2 + (32 + c) * W2_LD + kin];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, h1[mtp][0][r], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, h1[mtp][1][r], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, h1[mtp][0][r], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, h1[mtp][1][r], acc[1][1], 0, 0, 0);
    }
  // output layer on the vector ALU: this half's 32 units of each env, then the other half's partial sum
  float p[2][NOUT];
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int k = 0; k < NOUT; k++) p[nt][k] = 0.0f;
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float t0 = fast_tanh(acc[mt][0][r]), t1 = fast_tanh(acc[mt][1][r]);
#pragma unroll
      for (int k = 0; k < NOUT; k++) {
        const float w3 = L[T_W3 + k * HID + BRS_UNIT(mt, r)];
        p[0][k] = fmaf(w3, t0, p[0][k]); p[1][k] = fmaf(w3, t1, p[1][k]);
      }
    }
#undef BRS_UNIT
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int k = 0; k < NOUT; k++) out[nt][k] = p[nt][k] + __shfl_xor(p[nt][k], 32, 64) + L[T_B3 + k];
}

// observations of the two envs (32 nt + lane % 32) this lane feeds into the matrix cores; rows beyond n read as zero
__device__ __forceinline__ void load_obs_tiles(const float* __restrict__ obs, int n, int wave_base, float (&x)[2][OBS]) {
  const int c = threadIdx.x & 31;
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int e = wave_base + 32 * nt + c;
#pragma unroll
    for (int k = 0; k < OBS; k++) x[nt][k] = e < n ? obs[(size_t)OBS * e + k] : 0.0f;
  }
}

__global__ void __launch_bounds__(POLICY_THREADS) policy_act_kernel(const float* __restrict__ w, const int n, const float* __restrict__ obs,
                                                                    const uint64_t seed, const int64_t gid_base, const uint32_t step,
                                                                    const int deterministic, float* __restrict__ action,
                                                                    float* __restrict__ action_clipped, float* __restrict__ logp,
                                                                    float* __restrict__ value, float* __restrict__ noise) {
  __shared__ float Lpi[T_SIZE], Lvf[T_SIZE];
  stage_tower<ACT>(w + OFF_PI, Lpi);
  stage_tower<1>(w + OFF_VF, Lvf);
  __syncthreads();
  const int wave_base = blockIdx.x * blockDim.x + (threadIdx.x & ~63), h = (threadIdx.x >> 5) & 1;
  float x[2][OBS], m2[2][ACT], v2[2][1];
  load_obs_tiles(obs, n, wave_base, x);
  tower_mfma<ACT>(Lpi, x, m2);   // (no lane leaves before the matrix instructions: they need the whole wave)
  tower_mfma<1>(Lvf, x, v2);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // lane l finishes env l of the wave: N-tile h, column l % 32
  if (i >= n) return;
  const float mean[ACT] = {h ? m2[1][0] : m2[0][0], h ? m2[1][1] : m2[0][1]};
  const float v = h ? v2[1][0] : v2[0][0];
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
  value[i] = v;
}

__global__ void __launch_bounds__(POLICY_THREADS) policy_value_kernel(const float* __restrict__ w, const int n, const float* __restrict__ obs,
                                                                      float* __restrict__ value) {
  __shared__ float Lvf[T_SIZE];
  stage_tower<1>(w + OFF_VF, Lvf);
  __syncthreads();
  const int wave_base = blockIdx.x * blockDim.x + (threadIdx.x & ~63), h = (threadIdx.x >> 5) & 1;
  float x[2][OBS], v2[2][1];
  load_obs_tiles(obs, n, wave_base, x);
  tower_mfma<1>(Lvf, x, v2);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) value[i] = h ? v2[1][0] : v2[0][0];
}

__global__ void __launch_bounds__(POLICY_THREADS) bootstrap_kernel(const float* __restrict__ w, const int n, const float* __restrict__ tobs,
                                                                   const uint8_t* __restrict__ term, const uint8_t* __restrict__ trunc,
                                                                   const float gamma, float* __restrict__ reward) {
  __shared__ float Lvf[T_SIZE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool need = i < n && trunc[i] && !term[i];
  if (!__syncthreads_or(need)) return;  // a workgroup without a truncated episode leaves at once (the common case)
  stage_tower<1>(w + OFF_VF, Lvf);
  __syncthreads();
  const int wave_base = blockIdx.x * blockDim.x + (threadIdx.x & ~63), h = (threadIdx.x >> 5) & 1;
  float x[2][OBS], v2[2][1];
  load_obs_tiles(tobs, n, wave_base, x);
  tower_mfma<1>(Lvf, x, v2);
  if (need) reward[i] = fmaf(gamma, h ? v2[1][0] : v2[0][0], reward[i]);
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
  if (!g.ok) return pfail(p, BRS_ERR_HIP, "brs_policy_set_weights: hipSetDevice failed");
  // kernels enqueued earlier on ANY stream (PyTorch's side streams do not synchronise with the null stream) may still be
  // reading w_own: drain the device before overwriting it.  Not on the rollout path (weights change once per update).
  BRS_P_TRY(p, hipDeviceSynchronize());
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
  if (!g.ok) return pfail(p, BRS_ERR_HIP, "brs_policy_act: hipSetDevice failed");
  hipLaunchKernelGGL(policy_act_kernel, dim3((n + POLICY_THREADS - 1) / POLICY_THREADS), dim3(POLICY_THREADS), 0, (hipStream_t)stream, p->w, n, obs_dev, seed, env_index_base,
                     step, deterministic, action_dev, action_clipped_dev, logp_dev, value_dev, noise_dev);
  BRS_P_TRY(p, hipGetLastError());
  return BRS_OK;
}

int brs_policy_value(brs_policy* p, int32_t n, const float* obs_dev, float* value_dev, void* stream) {
  if (!p) return BRS_ERR_STATE;
  if (n <= 0 || !obs_dev || !value_dev) return pfail(p, BRS_ERR_ARG, "brs_policy_value: bad argument");
  PGuard g(p->device);
  if (!g.ok) return pfail(p, BRS_ERR_HIP, "brs_policy_value: hipSetDevice failed");
  hipLaunchKernelGGL(policy_value_kernel, dim3((n + POLICY_THREADS - 1) / POLICY_THREADS), dim3(POLICY_THREADS), 0, (hipStream_t)stream, p->w, n, obs_dev, value_dev);
  BRS_P_TRY(p, hipGetLastError());
  return BRS_OK;
}

int brs_rollout_bootstrap(brs_policy* p, int32_t n, const float* terminal_obs_dev, const uint8_t* terminated_dev,
                          const uint8_t* truncated_dev, float gamma, float* reward_dev, void* stream) {
  if (!p) return BRS_ERR_STATE;
  if (n <= 0 || !terminal_obs_dev || !terminated_dev || !truncated_dev || !reward_dev)
    return pfail(p, BRS_ERR_ARG, "brs_rollout_bootstrap: bad argument");
  PGuard g(p->device);
  if (!g.ok) return pfail(p, BRS_ERR_HIP, "brs_rollout_bootstrap: hipSetDevice failed");
  hipLaunchKernelGGL(bootstrap_kernel, dim3((n + POLICY_THREADS - 1) / POLICY_THREADS), dim3(POLICY_THREADS), 0, (hipStream_t)stream, p->w, n, terminal_obs_dev, terminated_dev,
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

// brs_kernels.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of include/brs.h.
//
// Kernel design:
//   * one wavefront LANE per environment instance; a 64-thread workgroup is one wave.  At the benchmark size
//     (65,536 envs) the grid is 1,024 waves = one wave per SIMD of the 256 CUs, so the register budget is the
//     full 512 VGPRs and nothing is gained by trading registers for occupancy.
//   * the whole env step (reward, control law, 250 substeps, block state machine, termination, observation,
//     time limit, auto-reset with in-kernel Philox) is ONE launch; state is read once from HBM (field-major
//     SoA: every wave-level load is one contiguous segment) and written once.
//   * the per-lane contact list lives in LDS as lane-strided columns (word w of slot s at (s*8+w)*64 + lane):
//     ds_read_b32/ds_write_b32 with consecutive lanes on consecutive banks, conflict-free by construction.
//   * the tiny dense solves (8x8 / 6x6 / 14x14 Cholesky of M + J^T D J) are fully unrolled on statically
//     indexed register arrays -- no MFMA (a 2-wheel rigid body is not a dense contraction), no scratch.
//   * no inter-lane communication, no barriers, no atomics: lanes are independent envs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/brs.h"
#if defined(BRS_TIMING)
__device__ unsigned long long brs_dbg[16];
__device__ unsigned long long brs_dbg_wave[4096];
#define BRS_TIMING_LANE_WORDS 153
#endif
#include "brs_state.hpp"

using namespace brs;

namespace {

template <bool BLK> constexpr int lane_words() { return BLK ? LDS_WORDS_ENV03 : LDS_WORDS_ENV01; }

template <bool BLK> __device__ __forceinline__ Store<float> lane_store(float* lds) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return Store<float>{lds + wave * (64 * lane_words<BLK>()) + lane, 64};
}

// VARIANT >= 0: the model constants of that registered id are folded at COMPILE time (constexpr make_params, default
// timestep / substeps): ~100 values become instruction literals instead of SGPRs.  The kernel is SGPR-bound as well as
// VGPR-bound -- with the constants in kernel arguments 5-15 % of its loop instructions were v_readlane reloads of spilled
// SGPRs, each a VALU issue slot.  Only the per-handle fields stay runtime.  VARIANT = -1: everything runtime (non-default
// timestep).
template <int VARIANT> __device__ __forceinline__ Params<float> fold_params(const Params<float>& rt) {
  if constexpr (VARIANT < 0) return rt;
  else {
    constexpr Params<float> c = make_params<float>(VARIANT, 0u, -1, 0, 0, 0.0, 0, 0);
    Params<float> p = c;
    p.seed = rt.seed; p.gid_base = rt.gid_base; p.auto_reset = rt.auto_reset; p.noise = rt.noise;
    p.max_episode_steps = rt.max_episode_steps; p.nsub = rt.nsub;
    return p;
  }
}

// Lane grouping (Env03): after every step the envs are regrouped along the lanes by the cost class of their NEXT step
// (brs_state.hpp: cost_class).  Bucket order along the lanes:
//     floor | far A | near (plain) | far B | wheel | floor + wheel
// "far" lanes (no block<->robot work at all; split by env parity into A and B) are put between the classes: a boundary wave
// then pays one expensive path, not two.  The step kernel counts its lanes per bucket as it retires (wave-aggregated atomics
// into 6 counters behind the lane map), brs_group_kernel turns the counts into bucket bases and hands out the slots with
// wave-aggregated cursors: a few microseconds over 256 workgroups.  (Round 2a: a stable counting sort in ONE workgroup,
// 95 us = 2 % of the step.)  The order inside a bucket depends on the order the atomics arrive in -- an env's arithmetic does
// not depend on its lane, so results stay bit-identical (test_determinism_and_shard_invariance).
constexpr int NBUCKET = 6;
__device__ __forceinline__ int bucket_of(int key, int e) {
  const int k = key & 3;
  if (k == 1) return 0;
  if (k == 2) return 4;
  if (k == 3) return 5;
  return (key & 4) ? ((e & 1) ? 3 : 1) : 2;
}
// counters behind the lane map and the keys: cnt[NBUCKET] (lanes per bucket, filled by the step kernel), cursor[NBUCKET], ticket
constexpr int GROUP_WORDS = 16;
template <bool BLK> __device__ __forceinline__ unsigned* group_counters(int* ii, int N) {
  return (unsigned*)(ii + ((size_t)Layout<BLK>::NI + 2) * N);
}

template <bool BLK, int VARIANT>
__device__ __forceinline__ void step_body(const Params<float>& Prt, const int N, double* __restrict__ d, float* __restrict__ f,
                                          int* __restrict__ ii, const float* __restrict__ actions, float* __restrict__ obs,
                                          float* __restrict__ reward, uint8_t* __restrict__ terminated,
                                          uint8_t* __restrict__ truncated, float* __restrict__ terminal_obs) {
  extern __shared__ float brs_lds_dyn[];
  float* lds = brs_lds_dyn;
  const int lane_slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane_slot >= N) return;  // no barriers anywhere: a partial last wave just masks lanes
  const Params<float> P = fold_params<VARIANT>(Prt);
  // lane <-> env: identity, or the cost-class grouping computed after the previous step (brs_state.hpp: cost_class).  The
  // gather makes the state accesses of a wave non-contiguous; at < 0.2 % of the HBM roofline that costs nothing.  The env
  // index is re-read wherever it is needed instead of being held across the loop (brs_state.hpp: LaneIndex).
  // The map and the cost classes live behind the int state in the SAME allocation (ii + NI * N: perm[N], then keys[N] bytes):
  // no extra kernel arguments -- the kernel is SGPR-bound too (its Params live in SGPRs), and two more pointers held across
  // the loop cost ~10 % in spill traffic (measured).  Env01 has no rare collision paths: identity, decided at compile time.
#if defined(BRS_NO_PERM)  // A/B builds only
  const LaneIndex idx{nullptr, lane_slot};
#else
  const LaneIndex idx{BLK ? ii + (size_t)Layout<BLK>::NI * N : nullptr, lane_slot};
#endif
  Store<float> st = lane_store<BLK>(lds);
#if defined(BRS_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  if ((threadIdx.x & 63) == 0) for (int k = 0; k < 16; k++) brs_tim_slots()[k] = 0;
#endif
  Stream<float> rng;
  float a0, a1;
  {
    const size_t i = idx.get();
    rng.open(P.seed, P.gid_base + (int64_t)i, 0u);
    a0 = actions[2 * i]; a1 = actions[2 * i + 1];
  }
  float o[6], to[6], rew;
  int te, tr, cls = 0;
  env_step_idx<float, BLK, float, LaneIndex>(P, st, rng, d, f, ii, (size_t)N, idx, a0, a1, o, to, rew, te, tr, BLK ? &cls : nullptr);
  const size_t i = idx.get();
  if constexpr (BLK) {
    ((uint8_t*)(ii + ((size_t)Layout<BLK>::NI + 1) * N))[i] = (uint8_t)cls;
    const int b = bucket_of(cls, (int)i);
    unsigned* cnt = group_counters<BLK>(ii, N);
#pragma unroll
    for (int k = 0; k < NBUCKET; k++) {
      const unsigned long long m = __ballot(b == k);
      if (m != 0ull && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[k], (unsigned)__popcll(m));
    }
  }
#if defined(BRS_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 12; k++) atomicAdd(&brs_dbg[k], brs_tim_slots()[k]);
    atomicMax(&brs_dbg[12], brs_tim_slots()[8]);  // slowest wave of the launch: cycles, trips (load imbalance, 1 wave per SIMD)
    atomicMax(&brs_dbg[13], brs_tim_slots()[9]);
    // per-wave record of the LAST launch: cycles, trips, HW_ID, XCC_ID (where did the slow waves run?)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w < 1024) {
      brs_dbg_wave[4 * w + 0] = brs_tim_slots()[8]; brs_dbg_wave[4 * w + 1] = brs_tim_slots()[9];
      brs_dbg_wave[4 * w + 2] = hw; brs_dbg_wave[4 * w + 3] = xcc;
    }
  }
#endif
#pragma unroll
  for (int k = 0; k < 6; k++) obs[6 * (size_t)i + k] = o[k];
  if (terminal_obs) {
#pragma unroll
    for (int k = 0; k < 6; k++) terminal_obs[6 * (size_t)i + k] = to[k];
  }
  reward[i] = rew;
  terminated[i] = (uint8_t)te;
  truncated[i] = (uint8_t)tr;
}

#define BRS_STEP_ARGS                                                                                                      \
  const Params<float> Prt, const int N, double *__restrict__ d, float *__restrict__ f, int *__restrict__ ii,              \
      const float *__restrict__ actions, float *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ terminated, \
      uint8_t *__restrict__ truncated, float *__restrict__ terminal_obs
template <bool BLK, int VARIANT> __global__ void __launch_bounds__(256) brs_step_kernel(BRS_STEP_ARGS) {
  step_body<BLK, VARIANT>(Prt, N, d, f, ii, actions, obs, reward, terminated, truncated, terminal_obs);
}
// The Env01-family body (8 dofs, 336 registers when left alone) capped at 256 registers so that TWO waves fit a SIMD
// (LDS: 14 KB per wave, no limit).  Measured, Env01-v2: 65,536 envs (1,024 waves = one per SIMD either way)
// 49.0 vs 48.0 M env-steps/s -- the 109 spilled VGPRs cost nothing visible; 131,072 / 262,144 / 524,288 envs: 70.9 / 75.5 /
// 79.1 M vs 49.8 / 50.9 / 51.5 M: the second resident wave is worth x1.42 - 1.54 as soon as a launch has more waves than the
// chip has SIMDs.  Default for 64-thread workgroups; BRS_ENV01_OCC1=1 selects the uncapped build (A/B).
template <int VARIANT> __global__ void __launch_bounds__(64, 2) brs_step_kernel_occ2(BRS_STEP_ARGS) {
  step_body<false, VARIANT>(Prt, N, d, f, ii, actions, obs, reward, terminated, truncated, terminal_obs);
}

template <bool BLK>
__global__ void __launch_bounds__(256) brs_reset_kernel(const Params<float> P, const int N, double* __restrict__ d,
                                                        float* __restrict__ f, int* __restrict__ ii,
                                                        const uint8_t* __restrict__ mask, float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (mask && !mask[i]) return;
  EnvState<float, BLK> S;
  load_state<float, BLK>(S, d, f, ii, (size_t)N, (size_t)i);
  Stream<float> rng;
  rng.open(P.seed, P.gid_base + (int64_t)i, S.rng_ctr);
  float o[6];
  Sim<float, BLK>::env_reset(P, S, rng, o);
  S.rng_ctr = rng.ctr;
  store_state<float, BLK>(S, d, f, ii, (size_t)N, (size_t)i);
#pragma unroll
  for (int k = 0; k < 6; k++) obs[6 * (size_t)i + k] = o[k];
}

template <bool BLK>
__global__ void __launch_bounds__(256) brs_physics_kernel(const Params<float> P, const int N, double* __restrict__ d,
                                                          float* __restrict__ f, int* __restrict__ ii,
                                                          const float* __restrict__ ctrl, const int nsub) {
  extern __shared__ float brs_lds_dyn[];
  float* lds = brs_lds_dyn;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  Store<float> st = lane_store<BLK>(lds);
  physics_mem<float, BLK, float>(P, st, d, f, ii, (size_t)N, (size_t)i, ctrl[2 * (size_t)i], ctrl[2 * (size_t)i + 1], nsub);
}

constexpr int GROUP_THREADS = 256, GROUP_ENVS = 1024;  // small workgroups (they have to find room between step-kernel waves
                                                       // when several handles share a GPU), 4 envs per thread
__global__ void __launch_bounds__(GROUP_THREADS) brs_group_kernel(const int N, const uint8_t* __restrict__ keys, int* __restrict__ perm,
                                                                  unsigned* __restrict__ ctr) {
  // ONE slot request per workgroup and bucket: same-address device atomics are the cost of this kernel (one request per
  // wave: 61 us for 65,536 envs; per 1,024 envs: 7 us).  Wave counts -> LDS -> exclusive offsets inside the workgroup
  constexpr int NV = GROUP_ENVS / 64;  // 64-env groups of the workgroup ("virtual waves": 4 passes x 4 waves)
  __shared__ unsigned wcnt[NV][NBUCKET], woff[NBUCKET];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int b[GROUP_ENVS / GROUP_THREADS];
  unsigned long long mine[GROUP_ENVS / GROUP_THREADS];
#pragma unroll
  for (int j = 0; j < GROUP_ENVS / GROUP_THREADS; j++) {
    const int e = blockIdx.x * GROUP_ENVS + j * GROUP_THREADS + threadIdx.x;
    b[j] = e < N ? bucket_of(keys[e], e) : -1;
    mine[j] = 0ull;
#pragma unroll
    for (int k = 0; k < NBUCKET; k++) {
      const unsigned long long m = __ballot(b[j] == k);
      if (lane == 0) wcnt[j * (GROUP_THREADS / 64) + w][k] = (unsigned)__popcll(m);
      mine[j] = b[j] == k ? m : mine[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < NBUCKET) {  // thread k: bucket base from the step kernel's counts, this workgroup's share of the cursor
    const int k = threadIdx.x;
    unsigned base = 0, tot = 0;
    for (int j = 0; j < k; j++) base += ctr[j];
    for (int v = 0; v < NV; v++) { const unsigned c = wcnt[v][k]; wcnt[v][k] = tot; tot += c; }  // exclusive over the 64-env groups
    woff[k] = base + (tot ? atomicAdd(&ctr[NBUCKET + k], tot) : 0u);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < GROUP_ENVS / GROUP_THREADS; j++) {
    const int e = blockIdx.x * GROUP_ENVS + j * GROUP_THREADS + threadIdx.x;
    if (b[j] >= 0) perm[woff[b[j]] + wcnt[j * (GROUP_THREADS / 64) + w][b[j]] + (unsigned)__popcll(mine[j] & ((1ull << lane) - 1ull))] = e;
  }
  // the last workgroup to finish clears the counters for the next step (every other one has read them by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&ctr[2 * NBUCKET], 1u) == gridDim.x - 1) {
#pragma unroll
      for (int k = 0; k <= 2 * NBUCKET; k++) ctr[k] = 0u;
    }
  }
}

thread_local std::string g_create_error;

}  // namespace

struct brs_handle {
  Params<float> P;
  int N = 0, device = 0, bt = 64;
  bool blk = false;
  double* d = nullptr;
  float* f = nullptr;
  int* ii = nullptr;
  size_t nd = 0, nf = 0, ni = 0;
  bool folded = false;       // model constants folded at compile time (default timestep): variant-specific step kernel
  bool occ2 = false;         // Env01 family: body capped at 256 registers, two waves per SIMD (default; BRS_ENV01_OCC1=1: off)
  bool grouping = false;     // Env03: regroup lanes by cost class after every step (perm / keys live behind the int state)
  int* perm() const { return ii + ni; }                            // [N] lane slot -> env
  uint8_t* keys() const { return (uint8_t*)(ii + ni + (size_t)N); }  // [N] cost class of every env for its next step
  unsigned* counters() const { return (unsigned*)(ii + ni + 2 * (size_t)N); }  // [GROUP_WORDS] bucket counts, cursors, ticket
  std::string err;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int fail(brs_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define BRS_HIP_TRY(h, expr)                                                                              \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(h, BRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

#if defined(BRS_TIMING)
size_t lds_bytes(const brs_handle* h) { return (size_t)h->bt * BRS_TIMING_LANE_WORDS * sizeof(float) + (size_t)(h->bt / 64) * 128; }
#else
size_t lds_bytes(const brs_handle* h) { return (size_t)h->bt * (h->blk ? LDS_WORDS_ENV03 : LDS_WORDS_ENV01) * sizeof(float); }
#endif
int grid_of(const brs_handle* h) { return (h->N + h->bt - 1) / h->bt; }

template <bool BLK> int upload_state(brs_handle* h, const std::vector<double>& d, const std::vector<float>& f, const std::vector<int>& ii) {
  BRS_HIP_TRY(h, hipMemcpy(h->d, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice));
  BRS_HIP_TRY(h, hipMemcpy(h->f, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
  BRS_HIP_TRY(h, hipMemcpy(h->ii, ii.data(), ii.size() * sizeof(int), hipMemcpyHostToDevice));
  return BRS_OK;
}
int download_state(brs_handle* h, std::vector<double>& d, std::vector<float>& f, std::vector<int>& ii) {
  d.resize(h->nd); f.resize(h->nf); ii.resize(h->ni);
  BRS_HIP_TRY(h, hipDeviceSynchronize());
  BRS_HIP_TRY(h, hipMemcpy(d.data(), h->d, h->nd * sizeof(double), hipMemcpyDeviceToHost));
  BRS_HIP_TRY(h, hipMemcpy(f.data(), h->f, h->nf * sizeof(float), hipMemcpyDeviceToHost));
  BRS_HIP_TRY(h, hipMemcpy(ii.data(), h->ii, h->ni * sizeof(int), hipMemcpyDeviceToHost));
  return BRS_OK;
}

}  // namespace

extern "C" {

int brs_sizes(int32_t variant, int32_t* nq, int32_t* nv, int32_t* nobs, int32_t* nact) {
  if (variant < 0 || variant > 5) return BRS_ERR_ARG;
  bool blk = variant == 2 || variant == 3;
  if (nq) *nq = blk ? 16 : 9;
  if (nv) *nv = blk ? 14 : 8;
  if (nobs) *nobs = 6;
  if (nact) *nact = 2;
  return BRS_OK;
}

int brs_create(const brs_config* cfg, brs_handle** out) {
  if (!cfg || !out) return fail(nullptr, BRS_ERR_ARG, "brs_create: null argument");
  *out = nullptr;
  if (cfg->variant < 0 || cfg->variant > 5) return fail(nullptr, BRS_ERR_ARG, "brs_create: unknown variant");
  if (cfg->num_envs <= 0) return fail(nullptr, BRS_ERR_ARG, "brs_create: num_envs must be > 0");
  if ((cfg->flags & BRS_FLAG_NOISE_ON) && (cfg->flags & BRS_FLAG_NOISE_OFF))
    return fail(nullptr, BRS_ERR_ARG, "brs_create: NOISE_ON and NOISE_OFF are exclusive");
  int bt = cfg->block_threads > 0 ? cfg->block_threads : 64;
  if (bt % 64 != 0 || bt > 256) return fail(nullptr, BRS_ERR_ARG, "brs_create: block_threads must be 64, 128, 192 or 256");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, BRS_ERR_HIP, std::string("brs_create: no HIP device (") + hipGetErrorString(e) + "); there is no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, BRS_ERR_ARG, "brs_create: device ordinal out of range");
  brs_handle* h = new brs_handle();
  h->N = cfg->num_envs; h->device = cfg->device; h->bt = bt; h->blk = cfg->variant == 2 || cfg->variant == 3;
  int noise = (cfg->flags & BRS_FLAG_NOISE_ON) ? 1 : ((cfg->flags & BRS_FLAG_NOISE_OFF) ? 0 : -1);
  h->P = make_params<float>(cfg->variant, cfg->flags & BRS_FLAG_AUTO_RESET, noise, cfg->max_episode_steps, cfg->substeps,
                            cfg->timestep, cfg->seed, cfg->env_index_base);
  DeviceGuard g(h->device);
  size_t N = (size_t)h->N;
  if (h->blk) { h->nd = Layout<true>::ND * N; h->nf = Layout<true>::NF * N; h->ni = Layout<true>::NI * N; }
  else { h->nd = Layout<false>::ND * N; h->nf = Layout<false>::NF * N; h->ni = Layout<false>::NI * N; }
  auto bail = [&](const std::string& m) { std::string mm = m; brs_destroy(h); return fail(nullptr, BRS_ERR_HIP, mm); };
  if (!g.ok) return bail("brs_create: hipSetDevice failed");
  // + 7 fp64 scratch columns addressed by lane slot (accessor pose parked during the last substep, brs_state.hpp: LaneIndex)
  if (hipMalloc(&h->d, (h->nd + 7 * N) * sizeof(double)) != hipSuccess) return bail("brs_create: hipMalloc(fp64 state) failed");
  if (hipMalloc(&h->f, h->nf * sizeof(float)) != hipSuccess) return bail("brs_create: hipMalloc(fp32 state) failed");
  if (hipMalloc(&h->ii, (h->ni + 2 * N + GROUP_WORDS) * sizeof(int)) != hipSuccess) return bail("brs_create: hipMalloc(int state) failed");
  std::vector<double> d(h->nd);
  std::vector<float> f(h->nf);
  std::vector<int> ii(h->ni);
  int rc;
  if (h->blk) { hostconv::init_state<true>(d.data(), f.data(), ii.data(), N, cfg->seed, cfg->env_index_base); rc = upload_state<true>(h, d, f, ii); }
  else { hostconv::init_state<false>(d.data(), f.data(), ii.data(), N, cfg->seed, cfg->env_index_base); rc = upload_state<false>(h, d, f, ii); }
  if (rc != BRS_OK) return bail("brs_create: initial upload failed: " + h->err);
  {  // lane map = identity until the first regrouping (always read by the Env03 step kernel)
    std::vector<int> id(2 * N, 0);
    for (size_t k = 0; k < N; k++) id[k] = (int)k;
    if (hipMemcpy(h->perm(), id.data(), 2 * N * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return bail("brs_create: lane map init failed");
    if (hipMemset(h->counters(), 0, GROUP_WORDS * sizeof(int)) != hipSuccess) return bail("brs_create: counter init failed");
    h->grouping = h->blk && !(cfg->flags & BRS_FLAG_NO_LANE_GROUPING);
    h->folded = !(cfg->timestep > 0 && cfg->timestep != 2e-5) && !std::getenv("BRS_NO_FOLD");
    h->occ2 = !h->blk && std::getenv("BRS_ENV01_OCC1") == nullptr;
  }
  // dynamic LDS above the 64 KiB default needs the attribute (Env03, 256-thread blocks: 144 KiB)
  size_t lb = lds_bytes(h);
  hipError_t ea = hipSuccess;
  auto want = [&](const void* fn) { if (ea == hipSuccess) ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); };
  if (h->blk) {
    want((const void*)brs_step_kernel<true, -1>); want((const void*)brs_step_kernel<true, ENV03_V1>);
    want((const void*)brs_step_kernel<true, ENV03_V2>); want((const void*)brs_physics_kernel<true>);
  } else {
    want((const void*)brs_step_kernel<false, -1>); want((const void*)brs_step_kernel<false, ENV01_V1>);
    want((const void*)brs_step_kernel<false, ENV01_V2>); want((const void*)brs_step_kernel<false, ENV01_V3>);
    want((const void*)brs_step_kernel<false, ENV02_V1>); want((const void*)brs_physics_kernel<false>);
    want((const void*)brs_step_kernel_occ2<-1>); want((const void*)brs_step_kernel_occ2<ENV01_V1>);
    want((const void*)brs_step_kernel_occ2<ENV01_V2>); want((const void*)brs_step_kernel_occ2<ENV01_V3>);
    want((const void*)brs_step_kernel_occ2<ENV02_V1>);
  }
  if (ea != hipSuccess) return bail(std::string("brs_create: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  *out = h;
  return BRS_OK;
}

int brs_destroy(brs_handle* h) {
  if (!h) return BRS_ERR_STATE;
  {
    DeviceGuard g(h->device);
    if (h->d) (void)hipFree(h->d);
    if (h->f) (void)hipFree(h->f);
    if (h->ii) (void)hipFree(h->ii);
  }
  delete h;
  return BRS_OK;
}

const char* brs_last_error(const brs_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int brs_reset(brs_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!h) return BRS_ERR_STATE;
  if (!obs_dev) return fail(h, BRS_ERR_ARG, "brs_reset: obs_dev is null");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (h->blk) hipLaunchKernelGGL(brs_reset_kernel<true>, dim3(grid_of(h)), dim3(h->bt), 0, s, h->P, h->N, h->d, h->f, h->ii, mask_dev, obs_dev);
  else hipLaunchKernelGGL(brs_reset_kernel<false>, dim3(grid_of(h)), dim3(h->bt), 0, s, h->P, h->N, h->d, h->f, h->ii, mask_dev, obs_dev);
  BRS_HIP_TRY(h, hipGetLastError());
  return BRS_OK;
}

int brs_step(brs_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* terminated_dev,
             uint8_t* truncated_dev, float* terminal_obs_dev, void* stream) {
  if (!h) return BRS_ERR_STATE;
  if (!actions_dev || !obs_dev || !reward_dev || !terminated_dev || !truncated_dev)
    return fail(h, BRS_ERR_ARG, "brs_step: null buffer");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  size_t lb = lds_bytes(h);
  const dim3 grid(grid_of(h)), block(h->bt);
#define BRS_LAUNCH_STEP(BLK_, VAR_)                                                                                          \
  hipLaunchKernelGGL((brs_step_kernel<BLK_, VAR_>), grid, block, lb, s, h->P, h->N, h->d, h->f, h->ii, actions_dev, obs_dev, \
                     reward_dev, terminated_dev, truncated_dev, terminal_obs_dev)
#define BRS_LAUNCH_OCC2(VAR_)                                                                                              \
  hipLaunchKernelGGL((brs_step_kernel_occ2<VAR_>), grid, block, lb, s, h->P, h->N, h->d, h->f, h->ii, actions_dev, obs_dev, \
                     reward_dev, terminated_dev, truncated_dev, terminal_obs_dev)
  if (!h->blk && h->occ2 && h->bt == 64) {
    switch (h->folded ? h->P.variant : -1) {
      case ENV01_V1: BRS_LAUNCH_OCC2(ENV01_V1); break;
      case ENV01_V2: BRS_LAUNCH_OCC2(ENV01_V2); break;
      case ENV01_V3: BRS_LAUNCH_OCC2(ENV01_V3); break;
      case ENV02_V1: BRS_LAUNCH_OCC2(ENV02_V1); break;
      default: BRS_LAUNCH_OCC2(-1);
    }
  } else
  switch (h->folded ? h->P.variant : -1) {
    case ENV01_V1: BRS_LAUNCH_STEP(false, ENV01_V1); break;
    case ENV01_V2: BRS_LAUNCH_STEP(false, ENV01_V2); break;
    case ENV01_V3: BRS_LAUNCH_STEP(false, ENV01_V3); break;
    case ENV02_V1: BRS_LAUNCH_STEP(false, ENV02_V1); break;
    case ENV03_V1: BRS_LAUNCH_STEP(true, ENV03_V1); break;
    case ENV03_V2: BRS_LAUNCH_STEP(true, ENV03_V2); break;
    default:
      if (h->blk) BRS_LAUNCH_STEP(true, -1); else BRS_LAUNCH_STEP(false, -1);
  }
#undef BRS_LAUNCH_STEP
#undef BRS_LAUNCH_OCC2
  // a step launch that failed left no bucket counts: the grouping kernel must not run on them (its output would not be a
  // permutation, and later steps would skip or double envs)
  BRS_HIP_TRY(h, hipGetLastError());
  if (h->blk) {  // lanes of the NEXT step; without grouping the counts the step kernel left are just cleared
    if (h->grouping) {
      hipLaunchKernelGGL(brs_group_kernel, dim3((h->N + GROUP_ENVS - 1) / GROUP_ENVS), dim3(GROUP_THREADS), 0, s, h->N, h->keys(), h->perm(), h->counters());
      BRS_HIP_TRY(h, hipGetLastError());
    } else
      BRS_HIP_TRY(h, hipMemsetAsync(h->counters(), 0, GROUP_WORDS * sizeof(int), s));
  }
  return BRS_OK;
}

int brs_physics(brs_handle* h, const float* ctrl_dev, int32_t nsub, void* stream) {
  if (!h) return BRS_ERR_STATE;
  if (!ctrl_dev || nsub < 0) return fail(h, BRS_ERR_ARG, "brs_physics: bad argument");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  size_t lb = lds_bytes(h);
  if (h->blk) hipLaunchKernelGGL(brs_physics_kernel<true>, dim3(grid_of(h)), dim3(h->bt), lb, s, h->P, h->N, h->d, h->f, h->ii, ctrl_dev, nsub);
  else hipLaunchKernelGGL(brs_physics_kernel<false>, dim3(grid_of(h)), dim3(h->bt), lb, s, h->P, h->N, h->d, h->f, h->ii, ctrl_dev, nsub);
  BRS_HIP_TRY(h, hipGetLastError());
  return BRS_OK;
}

#define BRS_STATE_ROUNDTRIP(h, MODIFY_T, MODIFY_F, WRITE)                                   \
  do {                                                                                     \
    if (!h) return BRS_ERR_STATE;                                                          \
    DeviceGuard g(h->device);                                                              \
    std::vector<double> d; std::vector<float> f; std::vector<int> ii;                      \
    int rc = download_state(h, d, f, ii);                                                  \
    if (rc != BRS_OK) return rc;                                                           \
    size_t N = (size_t)h->N; (void)N;                                                      \
    if (h->blk) { MODIFY_T; } else { MODIFY_F; }                                           \
    if (WRITE) { rc = h->blk ? upload_state<true>(h, d, f, ii) : upload_state<false>(h, d, f, ii); } \
    return rc;                                                                             \
  } while (0)

int brs_get_state(brs_handle* h, double* qpos, double* qvel, double* warm, double* time) {
  BRS_STATE_ROUNDTRIP(h, hostconv::get_state<true>(d.data(), f.data(), N, qpos, qvel, warm, time),
                      hostconv::get_state<false>(d.data(), f.data(), N, qpos, qvel, warm, time), false);
}
int brs_set_state(brs_handle* h, const double* qpos, const double* qvel, const double* warm, const double* time) {
  BRS_STATE_ROUNDTRIP(h, hostconv::set_state<true>(d.data(), f.data(), N, qpos, qvel, warm, time),
                      hostconv::set_state<false>(d.data(), f.data(), N, qpos, qvel, warm, time), true);
}
int brs_get_aux(brs_handle* h, double* aux) {
  if (!aux) return BRS_ERR_ARG;
  BRS_STATE_ROUNDTRIP(h, hostconv::get_aux<true>(d.data(), f.data(), ii.data(), N, aux),
                      hostconv::get_aux<false>(d.data(), f.data(), ii.data(), N, aux), false);
}
int brs_set_aux(brs_handle* h, const double* aux) {
  if (!aux) return BRS_ERR_ARG;
  BRS_STATE_ROUNDTRIP(h, hostconv::set_aux<true>(d.data(), f.data(), ii.data(), N, aux),
                      hostconv::set_aux<false>(d.data(), f.data(), ii.data(), N, aux), true);
}
int brs_get_xpose(brs_handle* h, double* xquat, double* xpos) {
  BRS_STATE_ROUNDTRIP(h, hostconv::get_xpose<true>(d.data(), N, xquat, xpos), hostconv::get_xpose<false>(d.data(), N, xquat, xpos), false);
}
int brs_set_xpose(brs_handle* h, const double* xquat, const double* xpos) {
  BRS_STATE_ROUNDTRIP(h, hostconv::set_xpose<true>(d.data(), N, xquat, xpos), hostconv::set_xpose<false>(d.data(), N, xquat, xpos), true);
}

int64_t brs_step_bytes_per_env(const brs_handle* h) {
  if (!h) return 0;
  size_t st = h->blk ? Layout<true>::bytes_per_env : Layout<false>::bytes_per_env;
  // state read + state written + action (8) + obs (24) + terminal obs (24) + reward (4) + two flags (2)
  return (int64_t)(2 * st + 8 + 24 + 24 + 4 + 2);
}
#if defined(BRS_TIMING)
// diagnostic builds only: read and clear the per-phase cycle sums
int brs_debug_waves(unsigned long long* out4096) {
  return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(brs_dbg_wave), 4096 * sizeof(unsigned long long)) == hipSuccess ? BRS_OK : BRS_ERR_HIP;
}
int brs_debug_counters(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(brs_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return BRS_ERR_HIP;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(brs_dbg), z, sizeof z) != hipSuccess) return BRS_ERR_HIP;
  return BRS_OK;
}
#endif
#ifndef BRS_BUILD_ID
#define BRS_BUILD_ID "unstamped"
#endif
const char* brs_build_id(void) { return BRS_BUILD_ID; }
const char* brs_step_kernel_name(const brs_handle* h) {
  if (!h) return "";
  // the instantiation brs_step launches, spelled as rocprofv3 prints it
  static thread_local char name[64];
  const int var = h->folded ? h->P.variant : -1;
  if (!h->blk && h->occ2 && h->bt == 64) std::snprintf(name, sizeof name, "brs_step_kernel_occ2<%d>", var);
  else std::snprintf(name, sizeof name, "brs_step_kernel<%s, %d>", h->blk ? "true" : "false", var);
  return name;
}

}  // extern "C"

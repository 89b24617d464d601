// brs_state.hpp -- persistent per-env state in HBM: struct-of-arrays, field-major ([field][N]) so that
// lane i of a wavefront reads element i of every field: each wave-level load is one contiguous
// 512 B (fp64) / 256 B (fp32) segment.  Positions/quaternions/time are fp64 accumulators (h = 2e-5 makes
// per-substep increments ~1e-5 of the value); velocities, warm start and env scalars are fp32.
//
// Also: conversion between this layout and MuJoCo-style row-major (qpos[N][nq], qvel[N][nv],
// qacc_warmstart[N][nv], time[N]) used by brs_get_state / brs_set_state (host side, fp64).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "brs_core.hpp"

namespace brs {

template <bool BLK> struct Layout {
  static constexpr int NQ = BLK ? 16 : 9, NV = BLK ? 14 : 8;
  // fp64 fields
  static constexpr int D_P = 0, D_Q = 3, D_TH = 7, D_TIME = 9, D_XQ = 10, D_XP = 14, D_BP = 17, D_BQ = 20, D_TIMER = 24;
  static constexpr int ND_POSE = BLK ? 25 : 17;
  // fp64 velocity accumulators behind the fp32 qvel fields (BRS_VEL64): v 3, w 3, ww 2, (bv 3, bw 3)
  static constexpr int D_V = ND_POSE, D_W = D_V + 3, D_WW = D_V + 6, D_BV = D_V + 8, D_BW = D_V + 11;
  static constexpr int ND = ND_POSE + (BRS_VEL64 ? NV : 0);
  // fp32 fields
  static constexpr int F_V = 0, F_W = 3, F_WW = 6, F_A = 8, F_LASTPITCH = 8 + NV, F_EPRET = 9 + NV, F_BV = 10 + NV,
                       F_BW = 13 + NV;
  static constexpr int F_MUW = (BLK ? 16 : 10) + NV, F_DTS = F_MUW + 1, F_POFF = F_MUW + 2, F_TWS = F_MUW + 3;
  static constexpr int NF = F_MUW + 4;
  // int32 fields
  static constexpr int I_ELAPSED = 0, I_RNG = 1, I_SIDE = 2, I_BAD = 3, NI = 4;
  static constexpr size_t bytes_per_env = (size_t)ND * 8 + (size_t)NF * 4 + (size_t)NI * 4;
};

// physics half of the state (what the 250-substep loop needs) ...
template <typename R, bool BLK, typename FT>
BRS_HD void load_state_phys(EnvState<R, BLK>& S, const double* d, const FT* f, const int* ii, size_t N, size_t i) {
  using L = Layout<BLK>;
#pragma unroll
  for (int k = 0; k < 3; k++) { S.p[k] = d[(L::D_P + k) * N + i]; S.xp[k] = d[(L::D_XP + k) * N + i]; }
#pragma unroll
  for (int k = 0; k < 4; k++) { S.q[k] = d[(L::D_Q + k) * N + i]; S.xq[k] = d[(L::D_XQ + k) * N + i]; }
  S.th[0] = d[(L::D_TH + 0) * N + i]; S.th[1] = d[(L::D_TH + 1) * N + i];
  S.time = d[L::D_TIME * N + i];
#pragma unroll
  for (int k = 0; k < 3; k++) { S.v[k] = (R)f[(L::F_V + k) * N + i]; S.w[k] = (R)f[(L::F_W + k) * N + i]; }
  S.ww[0] = (R)f[(L::F_WW + 0) * N + i]; S.ww[1] = (R)f[(L::F_WW + 1) * N + i];
#pragma unroll
  for (int k = 0; k < L::NV; k++) S.a[k] = (R)f[(L::F_A + k) * N + i];
  if constexpr (BLK) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      S.bp[k] = d[(L::D_BP + k) * N + i];
      S.bv[k] = (R)f[(L::F_BV + k) * N + i];
      S.bw[k] = (R)f[(L::F_BW + k) * N + i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) S.bq[k] = d[(L::D_BQ + k) * N + i];
  }
#if BRS_VEL64
#pragma unroll
  for (int k = 0; k < 3; k++) { S.vd[k] = d[(L::D_V + k) * N + i]; S.wd[k] = d[(L::D_W + k) * N + i]; S.v[k] = (R)S.vd[k]; S.w[k] = (R)S.wd[k]; }
  S.wwd[0] = d[(L::D_WW + 0) * N + i]; S.wwd[1] = d[(L::D_WW + 1) * N + i]; S.ww[0] = (R)S.wwd[0]; S.ww[1] = (R)S.wwd[1];
  if constexpr (BLK) {
#pragma unroll
    for (int k = 0; k < 3; k++) { S.bvd[k] = d[(L::D_BV + k) * N + i]; S.bwd[k] = d[(L::D_BW + k) * N + i]; S.bv[k] = (R)S.bvd[k]; S.bw[k] = (R)S.bwd[k]; }
  }
#endif
  S.rng_ctr = (uint32_t)ii[L::I_RNG * N + i];
  S.side_front = ii[L::I_SIDE * N + i];
  S.muw = (R)f[L::F_MUW * N + i];
  S.pnfr = 0; S.pnfb = 0; S.pnc = 0; S.psels = 0; S.pmR = 0; S.pmB = 0; S.pmC = 0;
}
// ... and the env-level scalars, only needed before and after the loop
template <typename R, bool BLK, typename FT>
BRS_HD void load_state_env(EnvState<R, BLK>& S, const double* d, const FT* f, const int* ii, size_t N, size_t i) {
  using L = Layout<BLK>;
  S.last_pitch = (R)f[L::F_LASTPITCH * N + i];
  S.ep_return = (R)f[L::F_EPRET * N + i];
  if constexpr (BLK) S.block_timer = d[L::D_TIMER * N + i]; else S.block_timer = -1.0;
  S.elapsed = ii[L::I_ELAPSED * N + i];
  S.bad = ii[L::I_BAD * N + i];
  S.dts = (R)f[L::F_DTS * N + i]; S.poff = (R)f[L::F_POFF * N + i]; S.tws = (R)f[L::F_TWS * N + i];
}
template <typename R, bool BLK, typename FT>
BRS_HD void load_state(EnvState<R, BLK>& S, const double* d, const FT* f, const int* ii, size_t N, size_t i) {
  load_state_phys<R, BLK, FT>(S, d, f, ii, N, i);
  load_state_env<R, BLK, FT>(S, d, f, ii, N, i);
}

template <typename R, bool BLK, typename FT>
BRS_HD void store_state(const EnvState<R, BLK>& S, double* d, FT* f, int* ii, size_t N, size_t i) {
  using L = Layout<BLK>;
#pragma unroll
  for (int k = 0; k < 3; k++) { d[(L::D_P + k) * N + i] = S.p[k]; d[(L::D_XP + k) * N + i] = S.xp[k]; }
#pragma unroll
  for (int k = 0; k < 4; k++) { d[(L::D_Q + k) * N + i] = S.q[k]; d[(L::D_XQ + k) * N + i] = S.xq[k]; }
  d[(L::D_TH + 0) * N + i] = S.th[0]; d[(L::D_TH + 1) * N + i] = S.th[1];
  d[L::D_TIME * N + i] = S.time;
#pragma unroll
  for (int k = 0; k < 3; k++) { f[(L::F_V + k) * N + i] = (FT)S.v[k]; f[(L::F_W + k) * N + i] = (FT)S.w[k]; }
  f[(L::F_WW + 0) * N + i] = (FT)S.ww[0]; f[(L::F_WW + 1) * N + i] = (FT)S.ww[1];
#pragma unroll
  for (int k = 0; k < L::NV; k++) f[(L::F_A + k) * N + i] = (FT)S.a[k];
  f[L::F_LASTPITCH * N + i] = (FT)S.last_pitch;
  f[L::F_EPRET * N + i] = (FT)S.ep_return;
  f[L::F_MUW * N + i] = (FT)S.muw; f[L::F_DTS * N + i] = (FT)S.dts; f[L::F_POFF * N + i] = (FT)S.poff; f[L::F_TWS * N + i] = (FT)S.tws;
  if constexpr (BLK) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      d[(L::D_BP + k) * N + i] = S.bp[k];
      f[(L::F_BV + k) * N + i] = (FT)S.bv[k];
      f[(L::F_BW + k) * N + i] = (FT)S.bw[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) d[(L::D_BQ + k) * N + i] = S.bq[k];
    d[L::D_TIMER * N + i] = S.block_timer;
  }
#if BRS_VEL64
#pragma unroll
  for (int k = 0; k < 3; k++) { d[(L::D_V + k) * N + i] = S.vd[k]; d[(L::D_W + k) * N + i] = S.wd[k]; }
  d[(L::D_WW + 0) * N + i] = S.wwd[0]; d[(L::D_WW + 1) * N + i] = S.wwd[1];
  if constexpr (BLK) {
#pragma unroll
    for (int k = 0; k < 3; k++) { d[(L::D_BV + k) * N + i] = S.bvd[k]; d[(L::D_BW + k) * N + i] = S.bwd[k]; }
  }
#endif
  ii[L::I_ELAPSED * N + i] = S.elapsed;
  ii[L::I_RNG * N + i] = (int)S.rng_ctr;
  ii[L::I_SIDE * N + i] = S.side_front;
  ii[L::I_BAD * N + i] = S.bad;
}

// Cost class of an env for its NEXT step (Env03): which of the rare, expensive collision paths it is likely to walk.
//   bit 0: the block can reach the floor within one env step          (plane<->box path + block<->floor rows in the solver)
//   bit 1: the block can reach a wheel within one env step            (box<->cylinder closest-feature path)
//   bit 2: the block cannot even reach the torso box                  (no block<->robot work at all: a cheap lane)
// A wave pays for a path whenever ONE of its 64 lanes walks it, and with one wave per SIMD a launch lasts as long as its
// slowest wave: brs_step therefore groups envs of one class into the same waves (a permutation env <-> lane, recomputed
// after every step from these keys).  Purely a scheduling hint: an env's arithmetic does not depend on its lane, a wrong
// guess only costs time.  Conservative reach tests: speed x step time + 3 mm.
template <typename R, bool BLK> BRS_HD int cost_class(const Params<R>& P, const EnvState<R, BLK>& S) {
  if constexpr (!BLK) return 0;
  else {
    const R T = (R)P.nsub * P.h, slack = (R)0.003;
    const R vb = sqrt_(S.bv[0] * S.bv[0] + S.bv[1] * S.bv[1] + S.bv[2] * S.bv[2]);
    const R vr = sqrt_(S.v[0] * S.v[0] + S.v[1] * S.v[1] + S.v[2] * S.v[2]) + (R)0.12 * sqrt_(S.w[0] * S.w[0] + S.w[1] * S.w[1] + S.w[2] * S.w[2]);
    const R reach = (vb + vr) * T + (R)0.5 * P.g * T * T + slack;
    int key = 0;
    const R low = (R)(S.bp[2] - P.floor_z_d) - P.block_brad - P.cc[CC_BLOCK_FLOOR].margin;
#if defined(BRS_CLASS_V1)
    key |= low < reach ? 1 : 0;
#else
    // only the block's DOWNWARD speed brings it to the floor (its orientation is covered by the bounding radius)
    key |= low < max_(-S.bv[2], (R)0) * T + (R)0.5 * P.g * T * T + slack ? 1 : 0;
#endif
    R qf[4] = {(R)S.q[0], (R)S.q[1], (R)S.q[2], (R)S.q[3]}, RT[9];
    quat2mat_(qf, RT);
    const R d[3] = {(R)(S.bp[0] - S.p[0]), (R)(S.bp[1] - S.p[1]), (R)(S.bp[2] - S.p[2])};
    R dl[3];
    mulT_(RT, d, dl);  // block centre in the torso frame
    const R rr = P.wheel_brad + P.block_brad + P.cc[CC_BLOCK_ROBOT].margin + reach;
    const R dz = dl[2] - P.wheel_pz, dxl = dl[0] + P.wheel_px, dxr = dl[0] - P.wheel_px;
    const R base = dl[1] * dl[1] + dz * dz;
    key |= (base + dxl * dxl < rr * rr || base + dxr * dxr < rr * rr) ? 2 : 0;
    // bit 2: the block cannot reach the torso box either (no block<->robot path at all this step): such lanes are cheap and
    // serve as SEPARATORS between the classes along the lane axis (brs_kernels.hip: brs_group_kernel)
    const R dzt = dl[2] - P.torso_cz, rt = P.torso_brad + P.block_brad + P.cc[CC_BLOCK_ROBOT].margin + reach;
    key |= (dl[0] * dl[0] + dl[1] * dl[1] + dzt * dzt > rt * rt) ? 4 : 0;
    return key;
  }
}

// One full env step working from / to the SoA state in memory (the HIP step kernel's body; the host test build runs
// the same function).  Register diet for the 250-substep loop: the accessor pose of the LAST forward pass is written
// straight to its HBM slot when the last substep starts and read back afterwards, and the env-level scalars are only
// loaded after the loop -- neither is live while the solver needs every VGPR.
// Where the env index of a lane comes from.  With lane grouping (brs_kernels.hip) it is a LOADED value (perm[lane slot]);
// kept in a register across the 250-substep loop it would cost two VGPRs the solver does not have (measured: +10 % VALU
// instructions from the extra register shuffling).  Every use therefore asks again: a volatile load, one L2 hit.
// The one indexed access INSIDE the loop -- parking the accessor pose when the last substep starts -- goes to a scratch
// column addressed by the LANE SLOT (affine in the thread id: SGPR base + one VGPR offset, as cheap as before the grouping);
// 7 extra fp64 fields behind the state (Layout::ND + j).  FixedIndex (host build, tests) parks it in the env's own fields.
struct FixedIndex {
  static constexpr bool SCRATCH = false;
  size_t i;
  BRS_HD size_t get() const { return i; }
  BRS_HD size_t slot_index() const { return i; }
};
struct LaneIndex {
  static constexpr bool SCRATCH = true;
  const volatile int* perm;  // nullptr = identity
  int slot;
  BRS_HD size_t get() const { return perm ? (size_t)perm[slot] : (size_t)slot; }
  BRS_HD size_t slot_index() const { return (size_t)slot; }
};

template <typename R, bool BLK, typename FT, typename IDX>
BRS_HD void env_step_idx(const Params<R>& P, Store<R>& st, Stream<R>& rng, double* d, FT* f, int* ii, size_t N, const IDX& idx,
                         float a0, float a1, float* obs, float* terminal_obs, float& reward, int& terminated, int& truncated,
                         int* next_cost_class = nullptr) {
  using L = Layout<BLK>;
  using SimT = Sim<R, BLK>;
  EnvState<R, BLK> S;
  CtrlT<R> ctrlL, ctrlR;
  R rew;
  {
    const size_t i = idx.get();
    load_state_phys<R, BLK, FT>(S, d, f, ii, N, i);
    load_state_env<R, BLK, FT>(S, d, f, ii, N, i);  // re-loaded after the loop: not live across it
    rng.ctr = S.rng_ctr;
    rew = SimT::env_pre(P, S, rng, a0, a1, ctrlL, ctrlR);
    if (P.v3) f[L::F_TWS * N + i] = (FT)S.tws;  // the schedule may have moved the target
  }
  {  // flattened substep x Newton loop: one trip = [start a substep] + [one Newton iteration] + [finish the substep]
    typename SimT::SubCtx C;
    int k = 0;
    bool fresh = true;
    while (k < P.nsub) {
      BRS_TIC(8);
      if (fresh) {
        if (k == P.nsub - 1) {
          const size_t sl = idx.slot_index();
          constexpr int FQ = IDX::SCRATCH ? L::ND : L::D_XQ, FP = IDX::SCRATCH ? L::ND + 4 : L::D_XP;
#pragma unroll
          for (int j = 0; j < 4; j++) d[(FQ + j) * N + sl] = S.q[j];
#pragma unroll
          for (int j = 0; j < 3; j++) d[(FP + j) * N + sl] = S.p[j];
        }
        SimT::sub_begin(P, st, S, ctrlL, ctrlR, C);
        fresh = false;
      }
      if (!C.conv) SimT::sub_iter(P, st, S, C);
      if (C.conv) {
        SimT::sub_end(P, S, C);
        k++;
        fresh = true;
      }
      BRS_TOC(8);
#if defined(BRS_TIMING) && defined(__HIP_DEVICE_COMPILE__)
      if (BRS_LEADER()) brs_tim_slots()[9] += 1ull;
#endif
    }
  }
  S.derive_vel32();  // (BRS_LAZY_VEL32: the fp32 velocities env_post, the cost class and the stored state read)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");  // re-read the pose from memory: do not keep it in registers across the loop
#endif
  const size_t i = idx.get();
  rng.gid = P.gid_base + (int64_t)i;  // (same value as before the loop: recomputed so that it is not live across it)
  if (P.nsub > 0) {
    const size_t sl = idx.slot_index();
    constexpr int FQ = IDX::SCRATCH ? L::ND : L::D_XQ, FP = IDX::SCRATCH ? L::ND + 4 : L::D_XP;
#pragma unroll
    for (int j = 0; j < 4; j++) S.xq[j] = d[(FQ + j) * N + sl];
#pragma unroll
    for (int j = 0; j < 3; j++) S.xp[j] = d[(FP + j) * N + sl];
  }
  load_state_env<R, BLK, FT>(S, d, f, ii, N, i);
  SimT::env_post(P, S, rng, rew, obs, terminal_obs, reward, terminated, truncated);
  S.rng_ctr = rng.ctr;
  if (next_cost_class) *next_cost_class = cost_class<R, BLK>(P, S);
  store_state<R, BLK, FT>(S, d, f, ii, N, i);
}
template <typename R, bool BLK, typename FT>
BRS_HD void env_step_mem(const Params<R>& P, Store<R>& st, Stream<R>& rng, double* d, FT* f, int* ii, size_t N, size_t i,
                         float a0, float a1, float* obs, float* terminal_obs, float& reward, int& terminated, int& truncated,
                         int* next_cost_class = nullptr) {
  env_step_idx<R, BLK, FT, FixedIndex>(P, st, rng, d, f, ii, N, FixedIndex{i}, a0, a1, obs, terminal_obs, reward, terminated, truncated,
                                       next_cost_class);
}

// physics only (parity tests): nsub substeps with ctrl held, same flattened loop
template <typename R, bool BLK, typename FT>
BRS_HD void physics_mem(const Params<R>& P, Store<R>& st, double* d, FT* f, int* ii, size_t N, size_t i, CtrlT<R> ctrlL, CtrlT<R> ctrlR, int nsub) {
  using SimT = Sim<R, BLK>;
  EnvState<R, BLK> S;
  load_state<R, BLK, FT>(S, d, f, ii, N, i);
  typename SimT::SubCtx C;
  int k = 0;
  bool fresh = true;
  while (k < nsub) {
    if (fresh) {
#pragma unroll
      for (int j = 0; j < 4; j++) S.xq[j] = S.q[j];
#pragma unroll
      for (int j = 0; j < 3; j++) S.xp[j] = S.p[j];
      SimT::sub_begin(P, st, S, ctrlL, ctrlR, C);
      fresh = false;
    }
    if (!C.conv) SimT::sub_iter(P, st, S, C);
    if (C.conv) {
      SimT::sub_end(P, S, C);
      k++;
      fresh = true;
    }
  }
  S.derive_vel32();
  store_state<R, BLK, FT>(S, d, f, ii, N, i);
}

// ---------------------------------------------------------------------------------- host conversions (fp64)
namespace hostconv {

// NaN test on the bit pattern (this header may be compiled with -ffast-math, which folds x != x)
inline bool isnan_bits(double x) { union { double f; uint64_t u; } c; c.f = x; return (c.u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }
inline void quat_norm(double* q) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int k = 0; k < 4; k++) q[k] /= n;
}
inline void quat_mat(const double* q, double* M) { quat2mat_<double>(q, M); }

// fresh envs: qpos0, zero velocity, accessor pose = identity, RNG block 0 reserved for the per-env "side" draw
template <bool BLK, typename FT>
inline void init_state(double* d, FT* f, int* ii, size_t N, uint64_t seed, int64_t gid_base) {
  using L = Layout<BLK>;
  for (size_t k = 0; k < (size_t)L::ND * N; k++) d[k] = 0;
  for (size_t k = 0; k < (size_t)L::NF * N; k++) f[k] = 0;
  for (size_t k = 0; k < (size_t)L::NI * N; k++) ii[k] = 0;
  for (size_t i = 0; i < N; i++) {
    d[(L::D_Q + 0) * N + i] = 1; d[(L::D_XQ + 0) * N + i] = 1;
    f[L::F_MUW * N + i] = (FT)1;
    if (BLK) { d[(L::D_BQ + 0) * N + i] = 1; d[L::D_TIMER * N + i] = -1.0; }
    uint32_t o[4];
    int64_t gid = gid_base + (int64_t)i;
    philox4x32_10(0u, 0u, (uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32),
                  (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), o);
    double u0 = (double)(o[0] >> 8) * (1.0 / 16777216.0);
    ii[L::I_SIDE * N + i] = u0 > 0.5 ? 1 : 0;  // Env03_v2.__init__ (envs/env03_v2.py:22)
    ii[L::I_RNG * N + i] = 1;
  }
}

// MuJoCo-style rows -> SoA.  Any of qpos/qvel/warm/time may be null (left untouched).  Setting qpos also
// refreshes the accessor pose (what set_state's mj_forward does) and normalises the quaternions.
template <bool BLK, typename FT>
inline void set_state(double* d, FT* f, size_t N, const double* qpos, const double* qvel, const double* warm,
                      const double* time) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    if (qpos) {
      const double* q = qpos + i * L::NQ;
      double qt[4] = {q[3], q[4], q[5], q[6]};
      quat_norm(qt);
      for (int k = 0; k < 3; k++) { d[(L::D_P + k) * N + i] = q[k]; d[(L::D_XP + k) * N + i] = q[k]; }
      for (int k = 0; k < 4; k++) { d[(L::D_Q + k) * N + i] = qt[k]; d[(L::D_XQ + k) * N + i] = qt[k]; }
      d[(L::D_TH + 0) * N + i] = q[7]; d[(L::D_TH + 1) * N + i] = q[8];
      if (BLK) {
        double qb[4] = {q[12], q[13], q[14], q[15]};
        quat_norm(qb);
        for (int k = 0; k < 3; k++) d[(L::D_BP + k) * N + i] = q[9 + k];
        for (int k = 0; k < 4; k++) d[(L::D_BQ + k) * N + i] = qb[k];
      }
    }
    if (qvel) {
      const double* v = qvel + i * L::NV;
      for (int k = 0; k < 3; k++) { f[(L::F_V + k) * N + i] = (FT)v[k]; f[(L::F_W + k) * N + i] = (FT)v[3 + k]; }
      f[(L::F_WW + 0) * N + i] = (FT)v[6]; f[(L::F_WW + 1) * N + i] = (FT)v[7];
      if (BLK)
        for (int k = 0; k < 3; k++) { f[(L::F_BV + k) * N + i] = (FT)v[8 + k]; f[(L::F_BW + k) * N + i] = (FT)v[11 + k]; }
#if BRS_VEL64
      for (int k = 0; k < L::NV; k++) d[(L::D_V + k) * N + i] = v[k];  // same order as qvel: v 3, w 3, ww 2, bv 3, bw 3
#endif
    }
    if (warm) {  // world-frame linear accelerations -> body-frame linear coordinates of the solver variable
      const double* a = warm + i * L::NV;
      double qt[4], M[9];
      for (int k = 0; k < 4; k++) qt[k] = d[(L::D_Q + k) * N + i];
      quat_mat(qt, M);
      for (int k = 0; k < 3; k++) f[(L::F_A + k) * N + i] = (FT)(M[k] * a[0] + M[3 + k] * a[1] + M[6 + k] * a[2]);
      for (int k = 3; k < 8; k++) f[(L::F_A + k) * N + i] = (FT)a[k];
      if (BLK) {
        for (int k = 0; k < 4; k++) qt[k] = d[(L::D_BQ + k) * N + i];
        quat_mat(qt, M);
        for (int k = 0; k < 3; k++) f[(L::F_A + 8 + k) * N + i] = (FT)(M[k] * a[8] + M[3 + k] * a[9] + M[6 + k] * a[10]);
        for (int k = 11; k < 14; k++) f[(L::F_A + k) * N + i] = (FT)a[k];
      }
    }
    if (time) d[L::D_TIME * N + i] = time[i];
  }
}

template <bool BLK, typename FT>
inline void get_state(const double* d, const FT* f, size_t N, double* qpos, double* qvel, double* warm, double* time) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    if (qpos) {
      double* q = qpos + i * L::NQ;
      for (int k = 0; k < 3; k++) q[k] = d[(L::D_P + k) * N + i];
      for (int k = 0; k < 4; k++) q[3 + k] = d[(L::D_Q + k) * N + i];
      q[7] = d[(L::D_TH + 0) * N + i]; q[8] = d[(L::D_TH + 1) * N + i];
      if (BLK) {
        for (int k = 0; k < 3; k++) q[9 + k] = d[(L::D_BP + k) * N + i];
        for (int k = 0; k < 4; k++) q[12 + k] = d[(L::D_BQ + k) * N + i];
      }
    }
    if (qvel) {
      double* v = qvel + i * L::NV;
      for (int k = 0; k < 3; k++) { v[k] = f[(L::F_V + k) * N + i]; v[3 + k] = f[(L::F_W + k) * N + i]; }
      v[6] = f[(L::F_WW + 0) * N + i]; v[7] = f[(L::F_WW + 1) * N + i];
      if (BLK)
        for (int k = 0; k < 3; k++) { v[8 + k] = f[(L::F_BV + k) * N + i]; v[11 + k] = f[(L::F_BW + k) * N + i]; }
#if BRS_VEL64
      for (int k = 0; k < L::NV; k++) v[k] = d[(L::D_V + k) * N + i];
#endif
    }
    if (warm) {
      double* a = warm + i * L::NV;
      double qt[4], M[9], ab[3];
      for (int k = 0; k < 4; k++) qt[k] = d[(L::D_XQ + k) * N + i];  // frame of the last forward pass
      quat_mat(qt, M);
      for (int k = 0; k < 3; k++) ab[k] = f[(L::F_A + k) * N + i];
      for (int k = 0; k < 3; k++) a[k] = M[3 * k] * ab[0] + M[3 * k + 1] * ab[1] + M[3 * k + 2] * ab[2];
      for (int k = 3; k < 8; k++) a[k] = f[(L::F_A + k) * N + i];
      if (BLK) {
        for (int k = 0; k < 4; k++) qt[k] = d[(L::D_BQ + k) * N + i];
        quat_mat(qt, M);
        for (int k = 0; k < 3; k++) ab[k] = f[(L::F_A + 8 + k) * N + i];
        for (int k = 0; k < 3; k++) a[8 + k] = M[3 * k] * ab[0] + M[3 * k + 1] * ab[1] + M[3 * k + 2] * ab[2];
        for (int k = 11; k < 14; k++) a[k] = f[(L::F_A + k) * N + i];
      }
    }
    if (time) time[i] = d[L::D_TIME * N + i];
  }
}

// aux rows [N][14]: last_pitch, block_timer (NaN = None), elapsed, rng_ctr, side_front, accessor pitch (read-only),
//                   ep_return, bad count, (2 unused), wheel/floor friction, delay_target_speed, pitch_offset, target_wheel_speed
template <bool BLK, typename FT> inline void get_aux(const double* d, const FT* f, const int* ii, size_t N, double* aux) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    double* a = aux + 14 * i;
    a[0] = f[L::F_LASTPITCH * N + i];
    double t = BLK ? d[L::D_TIMER * N + i] : -1.0;
    { union { double f; uint64_t u; } qn; qn.u = 0x7ff8000000000000ull; a[1] = t < 0 ? qn.f : t; }
    a[2] = ii[L::I_ELAPSED * N + i]; a[3] = (uint32_t)ii[L::I_RNG * N + i]; a[4] = ii[L::I_SIDE * N + i];
    double xq[4];
    for (int k = 0; k < 4; k++) xq[k] = d[(L::D_XQ + k) * N + i];
    double p, y;
    Sim<double, BLK>::pitch_yaw(xq, p, y);
    a[5] = p; a[6] = f[L::F_EPRET * N + i]; a[7] = ii[L::I_BAD * N + i]; a[8] = a[9] = 0;
    a[10] = f[L::F_MUW * N + i]; a[11] = f[L::F_DTS * N + i]; a[12] = f[L::F_POFF * N + i]; a[13] = f[L::F_TWS * N + i];
  }
}
template <bool BLK, typename FT> inline void set_aux(double* d, FT* f, int* ii, size_t N, const double* aux) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    const double* a = aux + 14 * i;
    f[L::F_LASTPITCH * N + i] = (FT)a[0];
    if (BLK) d[L::D_TIMER * N + i] = isnan_bits(a[1]) ? -1.0 : a[1];
    ii[L::I_ELAPSED * N + i] = (int)a[2]; ii[L::I_RNG * N + i] = (int)(uint32_t)a[3]; ii[L::I_SIDE * N + i] = a[4] != 0;
    f[L::F_EPRET * N + i] = (FT)a[6];
    f[L::F_MUW * N + i] = (FT)a[10]; f[L::F_DTS * N + i] = (FT)a[11]; f[L::F_POFF * N + i] = (FT)a[12]; f[L::F_TWS * N + i] = (FT)a[13];
  }
}
template <bool BLK> inline void get_xpose(const double* d, size_t N, double* xq, double* xp) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    if (xq) for (int k = 0; k < 4; k++) xq[4 * i + k] = d[(L::D_XQ + k) * N + i];
    if (xp) for (int k = 0; k < 3; k++) xp[3 * i + k] = d[(L::D_XP + k) * N + i];
  }
}
template <bool BLK> inline void set_xpose(double* d, size_t N, const double* xq, const double* xp) {
  using L = Layout<BLK>;
  for (size_t i = 0; i < N; i++) {
    if (xq) for (int k = 0; k < 4; k++) d[(L::D_XQ + k) * N + i] = xq[4 * i + k];
    if (xp) for (int k = 0; k < 3; k++) d[(L::D_XP + k) * N + i] = xp[3 * i + k];
  }
}

}  // namespace hostconv
}  // namespace brs

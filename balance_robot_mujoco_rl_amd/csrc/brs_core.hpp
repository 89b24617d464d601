// brs_core.hpp -- one environment instance of the balance-robot simulator, written for ONE GPU LANE.
//
// The HIP kernel (brs_kernels.hip) instantiates this with R = float, one wavefront lane per env, state in
// VGPRs for the whole env step (250 substeps fused), the per-lane contact list in LDS (lane-strided
// columns, conflict-free ds_read_b32).  The same source compiles on the host (tests/hostsim) with
// R = float or double so the algorithm can be checked against oracle/ without a GPU; that host build is
// test infrastructure and is never loaded by the product.
//
// What one substep is (the reference's hot call: mujoco.mj_step, envs/env01_v2.py:37):
//   kinematics -> smooth forces (gravity/gyroscopic bias, wheel damping, clamped velocity servos)
//   -> plane-cylinder / plane-box collision -> pyramidal-cone soft constraints (4 rows per contact)
//   -> Newton solve of the convex acceleration problem -> implicitfast -> semi-implicit advance.
// Formulation here (NOT MuJoCo's): the robot is a gyrostat; in body-frame linear coordinates its 8x8 mass matrix is
// CONSTANT and sparse (closed-form inverse); all floor contacts share the world-aligned frame; the block (Env03) is an
// isotropic free body.  One Newton solver over all dofs (8 or 14): every lane walks its own contact lists (robot<->floor,
// block<->floor, block<->robot); H = M + G^T W G per contact on packed pairs (v_pk_fma_f32), packed Cholesky, exact
// termination (a full step that reproduces its active set), previous-substep active sets as first guess.  The caller
// flattens the substep loop and the Newton loop into one per-lane state machine (sub_begin / sub_iter / sub_end).
#pragma once
#include "brs_model.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BRS_HD __host__ __device__ __forceinline__
#else
#define BRS_HD inline
#endif

#if defined(BRS_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
#define BRS_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; BRS_MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BRS_MARK(name) do { } while (0)
#endif
// diagnostic build only (-DBRS_TIMING): per-phase wave cycles.  Stamps are fenced with sched_barrier so the compiler
// cannot move work across them; lane 0 of the wave adds the interval to an LDS slot, flushed to brs_dbg at kernel end.
#if defined(BRS_TIMING) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long brs_dbg[16];
__device__ __forceinline__ unsigned long long brs_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
extern __shared__ float brs_lds_dyn[];
__device__ __forceinline__ unsigned long long* brs_tim_slots() {  // 16 x u64 per wave, placed after the contact columns
  return (unsigned long long*)(brs_lds_dyn + blockDim.x * (unsigned)BRS_TIMING_LANE_WORDS) + (threadIdx.x >> 6) * 16;
}
#define BRS_TIC(id) unsigned long long _brs_t##id = brs_stamp()
#define BRS_PIN(x) asm volatile("" : "+v"(x))
// the first ACTIVE lane books the interval (lane 0 may have left the loop before its wave has)
#define BRS_LEADER() (__lane_id() == (unsigned)__ffsll((long long)__ballot(1)) - 1u)
#define BRS_TOC(id) do { unsigned long long _d = brs_stamp() - _brs_t##id; if (BRS_LEADER()) brs_tim_slots()[id] += _d; } while (0)
#else
#define BRS_TIC(id) do { } while (0)
#define BRS_TOC(id) do { } while (0)
#define BRS_PIN(x) do { } while (0)
#endif
#ifndef BRS_MASK_HINT
#define BRS_MASK_HINT 1
#endif
#ifndef BRS_FLIP_TOL
#define BRS_FLIP_TOL 1e-6
#endif
#ifndef BRS_UNDAMPED_ITERS
#define BRS_UNDAMPED_ITERS 3
#endif
// velocities carried as fp64 ACCUMULATORS (like qpos and time): the state enters a step with the caller's fp64 qvel, h * acc is
// added in fp64 and the fp32 copy the force path reads is re-derived from it every substep
#ifndef BRS_VEL64
#define BRS_VEL64 1
#endif
// the block<->torso patch in PATCH-FRAME algebra (Sim::Patch below): per point ~60 instructions in a 6-dof twist space shared by
// all points of the patch, one 6x6 congruence per pass into H -- instead of a rank-3 update of the 12x12 dof block per point
#ifndef BRS_PATCH_FRAME
#define BRS_PATCH_FRAME 1
#endif
// clip candidates of the box-box patch parked at fixed LDS words and walked by bit scan (1) or rank-scattered (0)
#ifndef BRS_FIXED_SCATTER
#define BRS_FIXED_SCATTER 1
#endif
// fp32 velocity mirrors re-derived from the fp64 accumulators at the start of each substep (1) or carried across the solver (0)
#ifndef BRS_LAZY_VEL32
#define BRS_LAZY_VEL32 BRS_VEL64
#endif
// (Tried and not kept, round 3: the robot<->floor contacts in the same frame algebra as the patch -- they all share the world-aligned
// frame, so they can accumulate one 8x8 matrix in (frame-coordinate twist, two wheel rates) and enter H by one congruence: 840
// instead of ~1,400 instructions per trip on paper, but scalar and serially dependent where contact_into is packed and
// independent: -1.5 % Env03-v2, -1.8 % Env03-v1, +0.3 % Env01-v2 on one box.  Also not kept: the patch's ten 3x3 congruences as
// ~320 v_pk_fma_f32 on row pairs in H's own layout instead of ~540 scalar FMAs: +0.5 % / -1.3 % (v2 / v1), 50 more registers.
// profiles/r03_ab_experiments.json.)

namespace brs {

#if defined(BRS_STATS) && !defined(__HIP_DEVICE_COMPILE__)
struct Stats { long substeps, solves[3], iters[3], passA[3], backtracks[3]; int last_iters[3]; long hist[17], trips;
               long cp_calls, cp_reach, cp_torso, cp_wheel, cp_nc, cp_tight_torso, cp_tight_wheel, nfr_hist[9], nfb_hist[5], nc_hist[8];
               long flip_hist[2][9], trips_alt; int first_single; };  // rows that differ from the assembled set after the 1st / a later iteration's verify pass
inline Stats& stats() { static thread_local Stats s{}; return s; }
#define BRS_STAT(expr) do { expr; } while (0)
#else
#define BRS_STAT(expr) do { } while (0)
#endif

// ------------------------------------------------------------------------------------ math wrappers
BRS_HD float sqrt_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);  // v_sqrt_f32, 1 ulp
#else
  return sqrtf(x);
#endif
}
BRS_HD double sqrt_(double x) { return sqrt(x); }
BRS_HD float rsqrt_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);  // v_rsq_f32, 1 ulp
#else
  return 1.0f / sqrtf(x);
#endif
}
BRS_HD double rsqrt_(double x) { return 1.0 / sqrt(x); }
BRS_HD float rcp_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32, 1 ulp (an IEEE division costs ~10 instructions)
#else
  return 1.0f / x;
#endif
}
BRS_HD double rcp_(double x) { return 1.0 / x; }
// sqrt of a double in [0, ~1] without the (slow) fp64 sqrt on the device: v_rsq_f32 seed + two Newton steps in fp64
BRS_HD double sqrt64_(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (!(s > 1e-30)) return 0.0;
  double y = (double)__builtin_amdgcn_rsqf((float)s);
  y = y * (1.5 - 0.5 * s * y * y);
  y = y * (1.5 - 0.5 * s * y * y);
  return s * y;
#else
  return sqrt(s);
#endif
}
// 1/sqrt of a double in [1e-6, 1] the same way
BRS_HD double rsqrt64_(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = (double)__builtin_amdgcn_rsqf((float)s);
  y = y * (1.5 - 0.5 * s * y * y);
  y = y * (1.5 - 0.5 * s * y * y);
  return y;
#else
  return 1.0 / sqrt(s);
#endif
}
BRS_HD float abs_(float x) { return fabsf(x); }
BRS_HD double abs_(double x) { return fabs(x); }
BRS_HD float atan2_(float y, float x) { return atan2f(y, x); }
BRS_HD double atan2_(double y, double x) { return atan2(y, x); }
BRS_HD void sincos_(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
BRS_HD void sincos_(double x, double* s, double* c) { *s = sin(x); *c = cos(x); }
BRS_HD bool isbad_(float x) { union { float f; uint32_t u; } c; c.f = x; return (c.u & 0x7fffffffu) >= 0x7f800000u; }  // NaN or Inf
BRS_HD bool isbad_(double x) { union { double f; uint64_t u; } c; c.f = x; return (c.u & 0x7fffffffffffffffull) >= 0x7ff0000000000000ull; }
template <typename R> BRS_HD R max_(R a, R b) { return a > b ? a : b; }
template <typename R> BRS_HD R min_(R a, R b) { return a < b ? a : b; }

// three-way selects written as chains of two-way selects on VALUES: a nested ?: whose arms are loads is emitted as
// control flow (exec-mask branches per use), this form as two v_cndmask
template <typename R> BRS_HD R pick3(int k, R a, R b, R c) { R r = c; r = k == 1 ? b : r; r = k == 0 ? a : r; return r; }
template <typename R> BRS_HD R pick3(int k, const R* v) { const R a = v[0], b = v[1], c = v[2]; return pick3<R>(k, a, b, c); }
// value for body selector sel (0 torso, 1 left wheel, 2 right wheel): 0 / l / r
template <typename R> BRS_HD R by_wheel(int sel, R l, R r) { R x = (R)0; x = sel == 1 ? l : x; x = sel == 2 ? r : x; return x; }
template <typename R> BRS_HD void cross_(const R* a, const R* b, R* o) {
  R x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
template <typename R> BRS_HD R dot_(const R* a, const R* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// rotation matrix (row-major, body->world) of a unit quaternion (w,x,y,z)
template <typename R> BRS_HD void quat2mat_(const R* q, R* M) {
  R w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = 1 - 2 * (y * y + z * z); M[1] = 2 * (x * y - w * z);     M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z);     M[4] = 1 - 2 * (x * x + z * z); M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y);     M[7] = 2 * (y * z + w * x);     M[8] = 1 - 2 * (x * x + y * y);
}
template <typename R> BRS_HD void mulT_(const R* M, const R* v, R* o) {  // o = M^T v
  R x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2], y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2],
    z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
template <typename R> BRS_HD void mul_(const R* M, const R* v, R* o) {  // o = M v
  R x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2],
    z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

// fp64 quaternion advance q <- normalise(q) * exp(h * w / 2)  (MuJoCo mju_quatIntegrate), series in
// u = (h|w|/2)^2 (h|w| <= ~1e-2, so u^4 terms are < 1e-20); q stays unit to rounding, the renormalisation
// is a 2-term series as well -- no fp64 sqrt/div on the GPU.
BRS_HD void quat_advance(double* q, double wx, double wy, double wz, double h) {
  double hh = 0.5 * h, u = hh * hh * (wx * wx + wy * wy + wz * wz);
  double c = 1.0 + u * (-0.5 + u * (1.0 / 24.0 - u * (1.0 / 720.0)));
  double s = hh * (1.0 + u * (-1.0 / 6.0 + u * (1.0 / 120.0 - u * (1.0 / 5040.0))));
  double rx = s * wx, ry = s * wy, rz = s * wz;
  double e = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] - 1.0;
  double k = 1.0 + e * (-0.5 + 0.375 * e);
  double w0 = q[0] * k, x0 = q[1] * k, y0 = q[2] * k, z0 = q[3] * k;
  q[0] = w0 * c - x0 * rx - y0 * ry - z0 * rz;
  q[1] = w0 * rx + x0 * c + y0 * rz - z0 * ry;
  q[2] = w0 * ry - x0 * rz + y0 * c + z0 * rx;
  q[3] = w0 * rz + x0 * ry - y0 * rx + z0 * c;
}

// ------------------------------------------------------------------------------------ Philox4x32-10
// counter = (n, 0, env_gid_lo, env_gid_hi), key = seed; identical in oracle/brs_oracle.c
BRS_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// per-event uniform stream of one env: blocks are drawn lazily, leftovers are discarded at event end
template <typename R> struct Stream {
  uint64_t seed;
  int64_t gid;
  uint32_t ctr;  // next unused Philox block of this env (persistent per env)
  uint32_t b0, b1, b2, b3;
  int pos;
  const double* script;  // host test hook: scripted uniforms (always null on the GPU)
  int script_n, script_pos;
  BRS_HD void open(uint64_t s, int64_t g, uint32_t c) { seed = s; gid = g; ctr = c; pos = 4; b0 = b1 = b2 = b3 = 0; script = nullptr; script_n = script_pos = 0; }
  BRS_HD R next() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (script && script_pos < script_n) return (R)script[script_pos++];
#endif
    if (pos == 4) {
      uint32_t o[4];
      philox4x32_10(ctr, 0u, (uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32),
                    (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), o);
      b0 = o[0]; b1 = o[1]; b2 = o[2]; b3 = o[3];
      ctr++; pos = 0;
    }
    uint32_t x = pos == 0 ? b0 : (pos == 1 ? b1 : (pos == 2 ? b2 : b3));
    pos++;
    return (R)(x >> 8) * (R)(1.0 / 16777216.0);
  }
};

// ------------------------------------------------------------------------------------ per-lane contact store
// floor-contact slots hold 7 words, coupled (block<->robot) slots 10 plus 2 shared contact-frame slots of 3 words (frame 0:
// the torso<->block patch, 1: the wheel contact -- the block can touch one wheel at most); word k of the lane's column lives
// at base[k*stride] (GPU: stride 64 = one LDS row per word, consecutive lanes on consecutive banks).  7 coupled slots =
// PATCH_MAX = 6 patch points + the wheel point: nothing the generator emits is dropped.  160 words/lane = 40 KiB per wave:
// exactly 4 waves per 160-KiB CU (measured: no loss against 153 words).
#ifndef BRS_PATCH_MAX
#define BRS_PATCH_MAX 6  // (A/B builds may set 4: round-2a behaviour)
#endif
enum { PATCH_MAX = BRS_PATCH_MAX };
#if defined(BRS_TIMING)  // diagnostic build: one robot<->floor slot less makes room for the per-wave timing slots (the bench workload
enum { BRS_NRS = 7 };    // never has more than 4 robot<->floor contacts)
#else
enum { BRS_NRS = 8 };
#endif
enum { SLOT_ROBOT = 0, N_ROBOT_SLOTS = BRS_NRS, SLOT_BLOCK = BRS_NRS, N_BLOCK_SLOTS = 4, N_COUPLED_SLOTS = PATCH_MAX + 1, SLOT_WORDS = 7,
       COUPLED_WORDS = 10, COUPLED_BASE = (N_ROBOT_SLOTS + N_BLOCK_SLOTS) * SLOT_WORDS,
       FRAME_BASE = COUPLED_BASE + N_COUPLED_SLOTS * COUPLED_WORDS, N_FRAME_SLOTS = 2,
       LDS_WORDS_ENV01 = N_ROBOT_SLOTS * SLOT_WORDS,
       LDS_WORDS_ENV03 = FRAME_BASE + 3 * N_FRAME_SLOTS };
static_assert(LDS_WORDS_ENV03 <= 160, "4 waves per CU need <= 160 words per lane");
template <typename R> struct Store {
  R* base;
  int stride;
  BRS_HD R get(int s, int w) const { return base[(s * SLOT_WORDS + w) * stride]; }
  BRS_HD void set(int s, int w, R v) { base[(s * SLOT_WORDS + w) * stride] = v; }
  BRS_HD R getc(int c, int w) const { return base[(COUPLED_BASE + c * COUPLED_WORDS + w) * stride]; }
  BRS_HD void setc(int c, int w, R v) { base[(COUPLED_BASE + c * COUPLED_WORDS + w) * stride] = v; }
  BRS_HD R getf(int f, int w) const { return base[(FRAME_BASE + 3 * f + w) * stride]; }
  BRS_HD void setf(int f, int w, R v) { base[(FRAME_BASE + 3 * f + w) * stride] = v; }
};
// Per-lane bookkeeping of the contact lists lives in registers, not in LDS: 4-bit active-row masks packed 8 slots
// to a word (R: robot<->floor slots 0..7; B: block<->floor 0..3; C: block<->robot 0..6), body selectors 2 bits per slot.
struct Masks { uint32_t hR, hB, hC, nR, nB, nC; };  // h: the masks H was built with; n: masks at the latest evaluated point
BRS_HD int get4(uint32_t m, int slot) { return (int)((m >> (4 * slot)) & 15u); }
BRS_HD uint32_t put4(int v, int slot) { return (uint32_t)v << (4 * slot); }
// sels word: bits [0,16) robot<->floor body (0 torso, 1 L wheel, 2 R wheel), [16,18) the wheel of the block<->wheel contact (0 none, 1 L,
// 2 R).  Block<->robot slots 0 .. PATCH_MAX-1 hold the torso patch (contact frame 0), slot PATCH_MAX the wheel contact (frame 1)
BRS_HD int sel_robot(uint32_t sels, int c) { return (int)((sels >> (2 * c)) & 3u); }
BRS_HD int sel_wheel_contact(uint32_t sels) { return (int)((sels >> 16) & 3u); }  // 0: no block<->wheel contact, 1 L, 2 R

// ------------------------------------------------------------------------------------ env state (registers)
template <typename R, bool BLK> struct EnvState {
  static constexpr int NV = BLK ? 14 : 8;
  double p[3], q[4], th[2];  // torso position, unit quaternion (w,x,y,z), wheel angles  (fp64 accumulators)
  R v[3], w[3], ww[2];       // world-frame linear velocity, BODY-frame angular velocity, wheel rates
  double bp[3], bq[4];       // block pose
  R bv[3], bw[3];            // block: world linear, body angular
#if BRS_VEL64
  double vd[3], wd[3], wwd[2], bvd[3], bwd[3];  // the fp64 accumulators behind v, w, ww, bv, bw (those are their roundings)
#endif
  // the fp32 velocities the force path reads ARE the roundings of the fp64 accumulators: re-derived when a substep starts (and once
  // after the loop) instead of being carried across the solver -- 14 registers less at the kernel's register peak, same numbers
  BRS_HD void derive_vel32() {
#if BRS_VEL64
    for (int i = 0; i < 3; i++) { v[i] = (R)vd[i]; w[i] = (R)wd[i]; }
    ww[0] = (R)wwd[0]; ww[1] = (R)wwd[1];
    if constexpr (BLK) { for (int i = 0; i < 3; i++) { bv[i] = (R)bvd[i]; bw[i] = (R)bwd[i]; } }
#endif
  }
  BRS_HD void sync_vel64() {  // after code that wrote the fp32 velocities directly (reset, block throw)
#if BRS_VEL64
    for (int i = 0; i < 3; i++) { vd[i] = (double)v[i]; wd[i] = (double)w[i]; }
    wwd[0] = (double)ww[0]; wwd[1] = (double)ww[1];
    if constexpr (BLK) { for (int i = 0; i < 3; i++) { bvd[i] = (double)bv[i]; bwd[i] = (double)bw[i]; } }
#endif
  }
  R a[NV];                   // solver variable / warm start, BODY-frame linear coordinates (robot and block)
  double time;
  // accessor pose: what data.body("robot_body").xquat/.xpos would read (kinematics of the last forward pass)
  double xq[4], xp[3];
  // env-level
  R last_pitch;
  double block_timer;  // < 0 : None
  int elapsed;
  uint32_t rng_ctr;
  int side_front;
  R ep_return;
  int bad;
  R muw;            // Env02: wheel/floor friction of this episode (envs/env02_v1.py:57-65)
  R dts, poff, tws; // Env01-v3: delay_target_speed, pitch_offset, target_wheel_speed (envs/env01_v3.py:16-53)
  int pnfr, pnfb, pnc;  // previous substep: contact-list lengths, body selectors and final active-row masks -- the first
  uint32_t psels, pmR, pmB, pmC;  // guess of this substep's active set (not persisted across launches)
};

// type of the two servo targets (data.ctrl) held across the substeps of a step: the reference forms ctrl = qvel + 4 a in fp64
#if BRS_VEL64
template <typename R> using CtrlT = double;
#else
template <typename R> using CtrlT = R;
#endif

template <typename R> BRS_HD R impedance_(const ContactClass<R>& c, R dist) {
  if (c.inv_width == (R)0) return c.d0;
  R x = (c.margin - dist) * c.inv_width;  // >= 0 for an active contact
  x = min_(x, (R)1);
  const R om = 1 - x, lo = 2 * x * x, hi = 1 - 2 * om * om;  // both arms as values: a select, not a two-sided branch
  const R y = x <= (R)0.5 ? lo : hi;
  return c.d0 + y * (c.d1 - c.d0);
}

// closed-form solve of (M_b + diag(0..0,dL,dR)) x = f for the robot (body coordinates)
template <typename R> BRS_HD void msolve_(const Params<R>& P, const R* f, R dL, R dR, R* x) {
  x[2] = f[2] * P.inv_m;
  x[5] = f[5] * P.inv_Izz;
  x[0] = (P.Iyy * f[0] - P.mcz * f[4]) * P.inv_det_xy;
  x[4] = (P.m * f[4] - P.mcz * f[0]) * P.inv_det_xy;
  R iL = rcp_(P.Ia + dL), iR = rcp_(P.Ia + dR);
  R red = P.Ixx - P.Ia * P.Ia * (iL + iR) - P.mcz * P.cz;
  x[3] = (f[3] + P.Ia * (f[6] * iL - f[7] * iR) + P.cz * f[1]) * rcp_(red);
  x[1] = (f[1] + P.mcz * x[3]) * P.inv_m;
  x[6] = (f[6] + P.Ia * x[3]) * iL;
  x[7] = (f[7] - P.Ia * x[3]) * iR;
}
template <typename R> BRS_HD void msolve0_(const Params<R>& P, const R* f, R* x) {  // dL = dR = 0
  x[2] = f[2] * P.inv_m;
  x[5] = f[5] * P.inv_Izz;
  x[0] = (P.Iyy * f[0] - P.mcz * f[4]) * P.inv_det_xy;
  x[4] = (P.m * f[4] - P.mcz * f[0]) * P.inv_det_xy;
  x[3] = (f[3] + (f[6] - f[7]) + P.cz * f[1]) * rcp_(P.Ixx_red0);
  x[1] = (f[1] + P.mcz * x[3]) * P.inv_m;
  x[6] = f[6] * P.inv_Ia + x[3];
  x[7] = f[7] * P.inv_Ia - x[3];
}
// y = M_b x (robot)
template <typename R> BRS_HD void mmul_(const Params<R>& P, const R* x, R* y) {
  y[0] = P.m * x[0] + P.mcz * x[4];
  y[1] = P.m * x[1] - P.mcz * x[3];
  y[2] = P.m * x[2];
  y[3] = -P.mcz * x[1] + P.Ixx * x[3] + P.Ia * (x[7] - x[6]);
  y[4] = P.mcz * x[0] + P.Iyy * x[4];
  y[5] = P.Izz * x[5];
  y[6] = P.Ia * (x[6] - x[3]);
  y[7] = P.Ia * (x[7] + x[3]);
}

// packed lower-triangular index
BRS_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// in-place Cholesky of the [N0,N1) diagonal block of a packed symmetric matrix, then solve H x = b on it.
// inv-diagonals are kept in dinv.  All indices are compile-time after unrolling (registers, no scratch).
template <typename R, int N0, int N1> BRS_HD void chol_solve_block(R* H, const R* b, R* x) {
  R dinv[N1 - N0];
#pragma unroll
  for (int j = N0; j < N1; j++) {
    R s = H[tri(j, j)];
#pragma unroll
    for (int k = N0; k < j; k++) s -= H[tri(j, k)] * H[tri(j, k)];
    s = max_(s, (R)1e-30);
    R inv = rsqrt_(s);
    dinv[j - N0] = inv;
#pragma unroll
    for (int i = j + 1; i < N1; i++) {
      R t = H[tri(i, j)];
#pragma unroll
      for (int k = N0; k < j; k++) t -= H[tri(i, k)] * H[tri(j, k)];
      H[tri(i, j)] = t * inv;
    }
  }
  R y[N1 - N0];
#pragma unroll
  for (int i = N0; i < N1; i++) {
    R s = b[i];
#pragma unroll
    for (int k = N0; k < i; k++) s -= H[tri(i, k)] * y[k - N0];
    y[i - N0] = s * dinv[i - N0];
  }
#pragma unroll
  for (int i = N1 - 1; i >= N0; i--) {
    R s = y[i - N0];
#pragma unroll
    for (int k = i + 1; k < N1; k++) s -= H[tri(k, i)] * x[k];
    x[i] = s * dinv[i - N0];
  }
}

// ------------------------------------------------------------------------------------ packed pairs
// A lone wave per SIMD issues one VALU instruction every ~4 cycles whether it is v_fma_f32 or v_pk_fma_f32
// (tools/microbench/issue_rate.hip: packed = 1.6x the FMA rate), so the dense per-lane algebra is written on
// pairs: on the GPU V2<float> is a 2-wide ext vector (-> v_pk_fma_f32 / v_pk_mul_f32), on the host a plain struct.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float brs_f2 __attribute__((ext_vector_type(2)));
template <typename R> struct V2T { struct type { R x, y; }; };
template <> struct V2T<float> { typedef brs_f2 type; };
#else
template <typename R> struct V2T { struct type { R x, y; }; };
#endif
template <typename R> using V2 = typename V2T<R>::type;
template <typename R> BRS_HD V2<R> v2_make(R a, R b) { V2<R> v; v.x = a; v.y = b; return v; }
template <typename R> BRS_HD V2<R> v2_splat(R a) { V2<R> v; v.x = a; v.y = a; return v; }
#if defined(__HIP_DEVICE_COMPILE__)
BRS_HD brs_f2 v2_fma(brs_f2 a, brs_f2 b, brs_f2 c) { return __builtin_elementwise_fma(a, b, c); }
BRS_HD brs_f2 v2_mul(brs_f2 a, brs_f2 b) { return a * b; }
#endif
template <typename V> BRS_HD V v2_fma(V a, V b, V c) { V r; r.x = a.x * b.x + c.x; r.y = a.y * b.y + c.y; return r; }
template <typename V> BRS_HD V v2_mul(V a, V b) { V r; r.x = a.x * b.x; r.y = a.y * b.y; return r; }

// packed lower triangle: row a holds the pairs (cols 2k, 2k+1), k = 0..a/2; for even a the last .y is padding
BRS_HD constexpr int hp(int a, int k) { return (a / 2) * (a / 2 + 1) + ((a & 1) ? (a / 2 + 1) : 0) + k; }
BRS_HD constexpr int hp_count(int n) { return hp(n - 1, (n - 1) / 2) + 1; }
#define BRS_H2(a, b) (((b) & 1) ? H[hp((a), (b) / 2)].y : H[hp((a), (b) / 2)].x)
template <typename V, typename R> BRS_HD void h2_set(V* H, int a, int b, R v) {
  if (b & 1) H[hp(a, b / 2)].y = v; else H[hp(a, b / 2)].x = v;
}

// in-place Cholesky H = L L^T on the packed-pair layout and solve H x = b.  Row dot products run on pairs.
template <typename R, int N> BRS_HD void chol_solve_packed(V2<R>* H, const R* b, R* x) {
  R dinv[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    // s = H[j][j] - sum_{k<j} L[j][k]^2
    V2<R> acc = v2_splat<R>((R)0);
#pragma unroll
    for (int kp = 0; kp < j / 2; kp++) acc = v2_fma(H[hp(j, kp)], H[hp(j, kp)], acc);
    R s = BRS_H2(j, j) - (acc.x + acc.y);
    if (j & 1) s -= H[hp(j, j / 2)].x * H[hp(j, j / 2)].x;
    s = max_(s, (R)1e-30);
    R inv = rsqrt_(s);
    dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      V2<R> a2 = v2_splat<R>((R)0);
#pragma unroll
      for (int kp = 0; kp < j / 2; kp++) a2 = v2_fma(H[hp(i, kp)], H[hp(j, kp)], a2);
      R t = BRS_H2(i, j) - (a2.x + a2.y);
      if (j & 1) t -= H[hp(i, j / 2)].x * H[hp(j, j / 2)].x;
      if (j & 1) H[hp(i, j / 2)].y = t * inv; else H[hp(i, j / 2)].x = t * inv;
    }
  }
  // forward substitution on pairs of y
  V2<R> y2[(N + 1) / 2];
#pragma unroll
  for (int i = 0; i < N; i++) {
    V2<R> a2 = v2_splat<R>((R)0);
#pragma unroll
    for (int kp = 0; kp < i / 2; kp++) a2 = v2_fma(H[hp(i, kp)], y2[kp], a2);
    R sv = b[i] - (a2.x + a2.y);
    if (i & 1) sv -= H[hp(i, i / 2)].x * y2[i / 2].x;
    if (i & 1) y2[i / 2].y = sv * dinv[i]; else y2[i / 2].x = sv * dinv[i];
  }
  // backward substitution (column access: scalar)
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    R sv = (i & 1) ? y2[i / 2].y : y2[i / 2].x;
#pragma unroll
    for (int k = i + 1; k < N; k++) sv -= BRS_H2(k, i) * x[k];
    x[i] = sv * dinv[i];
  }
}

// ------------------------------------------------------------------------------------ one substep
template <typename R, bool BLK> struct Sim {
  static constexpr int NV = BLK ? 14 : 8;
  static constexpr int NH = NV * (NV + 1) / 2;
  using ES = EnvState<R, BLK>;

  // per-substep frame data shared by the passes
  struct Frame {
    R RT[9];           // torso body->world; its rows are the world axes in torso coordinates: the floor contact frame
    R RB[9];           // is n = +z (row 2), t1 = +y (row 1), t2 = -x (minus row 0).  Same for the block.
    BRS_HD const R* nT() const { return RT + 6; }
    BRS_HD const R* t1T() const { return RT + 3; }
    BRS_HD const R* xT() const { return RT; }
    BRS_HD const R* nB() const { return RB + 6; }
    BRS_HD const R* t1B() const { return RB + 3; }
    BRS_HD const R* xB() const { return RB; }
    R dTB[3];          // x_T - x_B (world), for coupled contacts
    R a0[NV];          // unconstrained acceleration (body coords)
    int nfr, nfb, nc;  // robot-floor, block-floor, coupled contact counts of this lane
    int pnfr, pnfb, pnc;  // the same of the previous substep
    uint32_t sels, psels, pmR, pmB, pmC;
    R muW, cDW;        // wheel<->floor friction and pyramid regulariser factor of this env
  };

  // wheel hinge column for a contact at r (torso frame) on wheel sel (1 L: axis -x at (-px,0,pz); 2 R: +x at (+px,0,pz))
  static BRS_HD void wheel_col(const Params<R>& P, int sel, const R* r, R* wc) {
    R s = by_wheel<R>(sel, (R)-1, (R)1);
    R dy = r[1], dz = r[2] - P.wheel_pz;
    wc[0] = 0; wc[1] = -s * dz; wc[2] = s * dy;  // (s e_x) x (d)
  }

  static BRS_HD void add_robot_floor(const Params<R>& P, Store<R>& st, Frame& F, const R* u, const R* w, const R* ww,
                                     int sel, int cls, const R* pt, R dist) {
    if (F.nfr >= N_ROBOT_SLOTS) return;
    const ContactClass<R>& c = P.cc[cls];
    const R cmu = cls == CC_WHEEL_FLOOR ? F.muW : c.mu, ccD = cls == CC_WHEEL_FLOOR ? F.cDW : c.cD;
    R r[3] = {pt[0] - F.nT()[0] * dist * (R)0.5, pt[1] - F.nT()[1] * dist * (R)0.5, pt[2] - F.nT()[2] * dist * (R)0.5};
    // point velocity in the torso frame
    R wc[3], wr[3];
    wheel_col(P, sel, r, wc);
    cross_(w, r, wr);
    const R wwL = ww[0], wwR = ww[1];
    R wsel = by_wheel<R>(sel, wwL, wwR);
    R pv[3] = {u[0] + wr[0] + wsel * wc[0], u[1] + wr[1] + wsel * wc[1], u[2] + wr[2] + wsel * wc[2]};
    R vn = dot_(F.nT(), pv), vt1 = dot_(F.t1T(), pv), vt2 = -dot_(F.xT(), pv);
    R imp = impedance_(c, dist);
    int s = SLOT_ROBOT + F.nfr;
    st.set(s, 0, r[0]); st.set(s, 1, r[1]); st.set(s, 2, r[2]);
    st.set(s, 3, -c.B * vn - c.K * imp * (dist - c.margin));
    st.set(s, 4, -c.B * cmu * vt1);
    st.set(s, 5, -c.B * cmu * vt2);
    st.set(s, 6, imp * rcp_((1 - imp) * ccD));
    F.sels |= (uint32_t)sel << (2 * F.nfr);
    F.nfr++;
  }

  // plane <-> cylinder (the two wheels), restating MuJoCo's primitive in the torso frame
  // The DISTANCES of the rim points -- the numbers that decide whether a contact point exists in this substep -- come from
  // the fp64 pose (Floor64, built once per substep): a wheel resting on the floor has dist ~ -1e-4 m as the difference of two
  // numbers of 3e-2 m, and in fp32 its rounding (~1e-8 m) decides on which 20-us substep a rim point switches on or off when
  // the robot rocks; that one-substep disagreement with the fp64 oracle was the source of most robot-coordinate outliers
  // above 1e-4.  ~40 fp64 operations per substep, no fp64 sqrt or division.
  struct Floor64 { double nx, ny, nz, len, zT; };
  static BRS_HD Floor64 floor64(const Params<R>& P, const ES& S) {
    const double w = S.q[0], x = S.q[1], y = S.q[2], z = S.q[3];
    Floor64 G;
    G.nx = 2 * (x * z - w * y); G.ny = 2 * (y * z + w * x); G.nz = 1 - 2 * (x * x + y * y);
    G.len = sqrt64_(G.ny * G.ny + G.nz * G.nz);
    G.zT = S.p[2] - P.floor_z_d;
    return G;
  }
  static BRS_HD void collide_wheel(const Params<R>& P, Store<R>& st, Frame& F, const R* u, const R* w, const R* ww,
                                   R zT, int sel, bool triangles, const Floor64& G) {
    const ContactClass<R>& c = P.cc[CC_WHEEL_FLOOR];
    R px = sel == 1 ? -P.wheel_px : P.wheel_px, pz = P.wheel_pz;
    R nx = F.nT()[0], ny = F.nT()[1], nz = F.nT()[2];
    R len = sqrt_(ny * ny + nz * nz);
    R vy, vz;
    if (len >= (R)1e-15) { R k = P.wheel_r * rcp_(len); vy = -ny * k; vz = -nz * k; }
    else { vy = 0; vz = -P.wheel_r; }
    R sg = nx > 0 ? (R)-1 : (R)1;  // cylinder axis (body x) flipped to point towards the plane
    R axh = sg * P.wheel_hl;
#if defined(BRS_FLOOR_DIST32)  // A/B: round-2a behaviour
    R prjaxis = -abs_(nx) * P.wheel_hl, prjvec = vy * ny + vz * nz;
    R dist0 = zT + nx * px + nz * pz;
    const R d1 = dist0 + prjaxis + prjvec, d2 = dist0 - prjaxis + prjvec, dT = dist0 + prjaxis - (R)0.5 * prjvec;
    const bool in1 = d1 < c.margin, in2 = d2 < c.margin, inT = dT < c.margin;
#else
    const double prjaxis = -abs_(G.nx) * P.wheel_hl_d, prjvec = -P.wheel_r_d * G.len;  // vy ny + vz nz = -r len
    const double dist0 = G.zT + G.nx * (sel == 1 ? -P.wheel_px_d : P.wheel_px_d) + G.nz * P.wheel_pz_d;
    const double d1d = dist0 + prjaxis + prjvec, d2d = dist0 - prjaxis + prjvec, dTd = dist0 + prjaxis - 0.5 * prjvec;
    const double mg = P.margin_d[CC_WHEEL_FLOOR];
    const bool in1 = d1d < mg, in2 = d2d < mg, inT = dTd < mg;
    const R d1 = (R)d1d, d2 = (R)d2d, dT = (R)dTd;
#endif
    if (!in1) return;
    if (!triangles) {
      R p1[3] = {px + axh, vy, pz + vz};
      add_robot_floor(P, st, F, u, w, ww, sel, CC_WHEEL_FLOOR, p1, d1);
      if (in2) {
        R p2[3] = {px - axh, vy, pz + vz};
        add_robot_floor(P, st, F, u, w, ww, sel, CC_WHEEL_FLOOR, p2, d2);
      }
    } else {
      if (inT) {
        const R k = (R)0.86602540378443864676;  // sqrt(3)/2 ; vec1 = sg*k*(0, vz, -vy)
        R pa[3] = {px + axh, sg * k * vz - (R)0.5 * vy, pz - sg * k * vy - (R)0.5 * vz};
        add_robot_floor(P, st, F, u, w, ww, sel, CC_WHEEL_FLOOR, pa, dT);
        R pb[3] = {px + axh, -sg * k * vz - (R)0.5 * vy, pz + sg * k * vy - (R)0.5 * vz};
        add_robot_floor(P, st, F, u, w, ww, sel, CC_WHEEL_FLOOR, pb, dT);
      }
    }
  }

  // plane <-> torso box: corners below the centre with dist < margin, at most 4 (MuJoCo's plane-box primitive)
  // which of the 8 box corners (+-sx, +-sy, +-sz) touch the plane: corners below the box centre with dist < margin, the
  // first 4 in index order (MuJoCo's plane-box primitive).  Returns the count and the indices packed 3 bits each, so that
  // the (expensive) contact records are built by a loop over the PRESENT corners only: lanes of a wave hold boxes in
  // different orientations, an unrolled 8-corner loop would run the record code for every corner some lane uses.
  template <typename T> static BRS_HD int box_corners(T nx, T ny, T nz, T sx, T sy, T sz, T dc, T margin, int& list) {
    int cnt = 0;
    list = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      T ld = ((i & 1) ? nx : -nx) * sx + ((i & 2) ? ny : -ny) * sy + ((i & 4) ? nz : -nz) * sz;
      bool hit = dc + ld < margin && ld <= 0 && cnt < 4;
      list |= hit ? (i << (3 * cnt)) : 0;
      cnt += hit ? 1 : 0;
    }
    return cnt;
  }
  // plane <-> torso box
  static BRS_HD void collide_torso(const Params<R>& P, Store<R>& st, Frame& F, const R* u, const R* w, const R* ww, R zT, const Floor64& G) {
    const ContactClass<R>& c = P.cc[CC_TORSO_FLOOR];
    R nx = F.nT()[0], ny = F.nT()[1], nz = F.nT()[2];
    R dc = zT + nz * P.torso_cz;
    // cheap reject: lowest corner
    R low = dc - (abs_(nx) * P.torso_sx + abs_(ny) * P.torso_sy + abs_(nz) * P.torso_sz);
    if (!(low < c.margin + (R)1e-6)) return;  // (fp32 reject with slack; the decisions below are made in fp64)
    const double dcd = G.zT + G.nz * P.torso_cz_d;
    int list, cnt = box_corners<double>(G.nx, G.ny, G.nz, P.torso_s_d[0], P.torso_s_d[1], P.torso_s_d[2], dcd, P.margin_d[CC_TORSO_FLOOR], list);
    for (int k = 0; k < cnt; k++) {
      int i = (list >> (3 * k)) & 7;
      R lx = (i & 1) ? P.torso_sx : -P.torso_sx, ly = (i & 2) ? P.torso_sy : -P.torso_sy, lz = (i & 4) ? P.torso_sz : -P.torso_sz;
      R d = (R)(dcd + G.nx * ((i & 1) ? P.torso_s_d[0] : -P.torso_s_d[0]) + G.ny * ((i & 2) ? P.torso_s_d[1] : -P.torso_s_d[1]) +
                G.nz * ((i & 4) ? P.torso_s_d[2] : -P.torso_s_d[2]));
      R pt[3] = {lx, ly, P.torso_cz + lz};
      add_robot_floor(P, st, F, u, w, ww, 0, CC_TORSO_FLOOR, pt, d);
    }
  }
  static BRS_HD void collide_block_floor(const Params<R>& P, Store<R>& st, Frame& F, const ES& S, const R* uB, const R* wB, R zB) {
    const ContactClass<R>& c = P.cc[CC_BLOCK_FLOOR];
    R nx = F.nB()[0], ny = F.nB()[1], nz = F.nB()[2], s = P.block_s;
    R low = zB - (abs_(nx) + abs_(ny) + abs_(nz)) * s;
    if (!(low < c.margin + (R)1e-6)) return;  // (fp32 reject with slack; the decisions below are made in fp64, see Floor64)
    const double qw = S.bq[0], qx = S.bq[1], qy = S.bq[2], qz = S.bq[3];
    const double nxd = 2 * (qx * qz - qw * qy), nyd = 2 * (qy * qz + qw * qx), nzd = 1 - 2 * (qx * qx + qy * qy);
    const double zBd = S.bp[2] - P.floor_z_d, sd = P.block_s_d;
    int list, cnt = box_corners<double>(nxd, nyd, nzd, sd, sd, sd, zBd, P.margin_d[CC_BLOCK_FLOOR], list);
    for (int k = 0; k < cnt; k++) {
      int i = (list >> (3 * k)) & 7;
      R lx = (i & 1) ? s : -s, ly = (i & 2) ? s : -s, lz = (i & 4) ? s : -s;
      R d = (R)(zBd + nxd * ((i & 1) ? sd : -sd) + nyd * ((i & 2) ? sd : -sd) + nzd * ((i & 4) ? sd : -sd));
      R r[3] = {lx - nx * d * (R)0.5, ly - ny * d * (R)0.5, lz - nz * d * (R)0.5};
      R wr[3];
      cross_(wB, r, wr);
      R pv[3] = {uB[0] + wr[0], uB[1] + wr[1], uB[2] + wr[2]};
      R vn = dot_(F.nB(), pv), vt1 = dot_(F.t1B(), pv), vt2 = -dot_(F.xB(), pv);
      R imp = impedance_(c, d);
      int sl = SLOT_BLOCK + F.nfb;
      st.set(sl, 0, r[0]); st.set(sl, 1, r[1]); st.set(sl, 2, r[2]);
      st.set(sl, 3, -c.B * vn - c.K * imp * (d - c.margin));
      st.set(sl, 4, -c.B * c.mu * vt1);
      st.set(sl, 5, -c.B * c.mu * vt2);
      st.set(sl, 6, imp * rcp_((1 - imp) * c.cD));
      F.nfb++;
    }
  }

  // ---- coupled (block <-> robot) contacts: this project's OWN analytic generator (MuJoCo: mjc_BoxBox / libccd)
  // signed distance of a point (box frame) to a box: Euclidean outside, max-axis inside; outward normal
  static BRS_HD R point_box(const R* p, R sx, R sy, R sz, R* nrm) {
    const R q0 = abs_(p[0]) - sx, q1 = abs_(p[1]) - sy, q2 = abs_(p[2]) - sz;
    const R s0 = p[0] >= 0 ? (R)1 : (R)-1, s1 = p[1] >= 0 ? (R)1 : (R)-1, s2 = p[2] >= 0 ? (R)1 : (R)-1;
    const bool outside = (q0 > 0) | (q1 > 0) | (q2 > 0);
    const R m0 = max_(q0, (R)0), m1 = max_(q1, (R)0), m2 = max_(q2, (R)0);
    const R dd = sqrt_(m0 * m0 + m1 * m1 + m2 * m2), idd = rcp_(max_(dd, (R)1e-30));
    int ax = 0;
    R best = q0;
    if (q1 > best) { best = q1; ax = 1; }
    if (q2 > best) { best = q2; ax = 2; }
    nrm[0] = outside ? s0 * m0 * idd : (ax == 0 ? s0 : (R)0);
    nrm[1] = outside ? s1 * m1 * idd : (ax == 1 ? s1 : (R)0);
    nrm[2] = outside ? s2 * m2 * idd : (ax == 2 ? s2 : (R)0);
    return outside ? dd : best;
  }
  // MuJoCo's mju_makeFrame on a unit normal fw[0..2]: fills tangents fw[3..8]
  static BRS_HD void make_frame(R* fw) {
    fw[3] = 0; fw[4] = 0; fw[5] = 0;
    if (fw[1] < (R)0.5 && fw[1] > (R)-0.5) fw[4] = 1; else fw[5] = 1;
    R dp = dot_(fw, fw + 3);
    fw[3] -= dp * fw[0]; fw[4] -= dp * fw[1]; fw[5] -= dp * fw[2];
    R il = rsqrt_(dot_(fw + 3, fw + 3));
    fw[3] *= il; fw[4] *= il; fw[5] *= il;
    cross_(fw, fw + 3, fw + 6);
  }
  // coupled record (10 words): rT(3) torso frame, rB(3) block frame, An, Bt1, Bt2, D; the world normal robot->block of its
  // patch goes to contact-frame slot `sel` (written by the patch's first point, share = false).
  // fw = world contact frame (normal + MuJoCo's mju_makeFrame tangents), built once per patch by the caller.
  static BRS_HD void add_coupled(const Params<R>& P, Store<R>& st, Frame& F, const ES& S, const R* rT, const R* fw, R dist,
                                 int sel, bool share) {
    if (sel == 0 && F.nc >= PATCH_MAX) return;
    const ContactClass<R>& c = P.cc[CC_BLOCK_ROBOT];
    R pw[3], rB[3], wc[3], t[3];
    mul_(F.RT, rT, pw);
    pw[0] += F.dTB[0]; pw[1] += F.dTB[1]; pw[2] += F.dTB[2];
    mulT_(F.RB, pw, rB);
    wheel_col(P, sel, rT, wc);
    // relative point velocity (block minus robot), world frame
    cross_(S.w, rT, t);
    const R wwL = S.ww[0], wwR = S.ww[1];
    R wsel = by_wheel<R>(sel, wwL, wwR);
    R pT[3] = {t[0] + wsel * wc[0], t[1] + wsel * wc[1], t[2] + wsel * wc[2]}, pTw[3], pBw[3];
    mul_(F.RT, pT, pTw);
    cross_(S.bw, rB, t);
    mul_(F.RB, t, pBw);
    R rel[3] = {S.bv[0] + pBw[0] - S.v[0] - pTw[0], S.bv[1] + pBw[1] - S.v[1] - pTw[1], S.bv[2] + pBw[2] - S.v[2] - pTw[2]};
    R vn = dot_(fw, rel), vt1 = dot_(fw + 3, rel), vt2 = dot_(fw + 6, rel);
    R imp = impedance_(c, dist);
    R cD = sel == 0 ? c.cD : P.cD_block_wheel;
    const int k = sel == 0 ? F.nc : PATCH_MAX;  // the wheel contact has its own slot: the patch loops never meet it
    st.setc(k, 0, rT[0]); st.setc(k, 1, rT[1]); st.setc(k, 2, rT[2]);
    st.setc(k, 3, rB[0]); st.setc(k, 4, rB[1]); st.setc(k, 5, rB[2]);
    st.setc(k, 6, -c.B * vn - c.K * imp * (dist - c.margin));
    st.setc(k, 7, -c.B * c.mu * vt1);
    st.setc(k, 8, -c.B * c.mu * vt2);
    st.setc(k, 9, imp * rcp_((1 - imp) * cD));
    if (!share) { const int fs = sel ? 1 : 0; st.setf(fs, 0, fw[0]); st.setf(fs, 1, fw[1]); st.setf(fs, 2, fw[2]); }
    F.sels |= (uint32_t)sel << 16;
    F.nc += sel == 0 ? 1 : 0;
  }
  static BRS_HD void world_frame(const Frame& F, const R* nTf, R* fw) {  // unit normal in the torso frame -> world contact frame
    mul_(F.RT, nTf, fw);
    make_frame(fw);
  }
  // ---- patch-frame algebra for the block<->torso patch (BRS_PATCH_FRAME).  All points of a patch share ONE contact frame
  // (n, t1, t2).  In FRAME coordinates, with c' = the block centre seen from the torso origin and rho = r' - c' the position of a
  // point seen from the BLOCK CENTRE,
  //     acceleration of the block's material point:    aB' + wB' x rho                (aB' = Ph x[8:11], wB' = Ph x[11:14])
  //     ... of the torso's:                             aT' + wT' x (c' + rho)         (aT' = Fm x[0:3], wT' = Fm x[3:6])
  // so every point's three frame rows are  K(rho) tau,  K(r) = [I | -[r]x],  tau = (Vp, Wp) = (aB' - aT' + c' x wT', wB' - wT'): the
  // RELATIVE twist at the block centre, a linear map of the 12 dofs that does not depend on the point.  The patch's contribution
  // to H is therefore  T^T (sum_c K_c^T W_c K_c) T  with a 6x6 matrix Z accumulated per point (~45 FMAs) and ONE congruence per
  // pass; the rows at x, and the wrench of the patch, are 6-vectors mapped once.
  // WHY THE BLOCK CENTRE: the twist's reference point decides which body's angular acceleration meets the long lever |c'| ~ 0.15 m.
  // Taken at the torso origin (first version of this algebra) it was the block's -- a 20-g cube that reaches 1e5 rad/s^2 in an
  // impact -- and the block's point acceleration came out as the difference of two terms ten times its size: the one campaign
  // env-step above 1e-4 of round 3 (Env03-v1, block quaternion 2.4e-4; 5.6e-5 with the per-point rank updates) was that
  // cancellation.  At the block centre the long lever multiplies the torso's angular acceleration (<= 1e3 rad/s^2).
  // Per point the passes read 7 LDS words (rho, An, Bt1, Bt2, D).  Record of patch slot k (words): 0-2 rho, 3 An, 4 Bt1, 5 Bt2,
  // 6 D; words 7-9 of slots 0-2 hold the frame axes in the TORSO frame (Fm rows), of slots 3-5 in the BLOCK frame (Ph rows);
  // contact-frame slot 0 holds c'.
  struct Patch { R Fm[9], Ph[9], c[3], cT[3]; };  // cT: the block centre in the torso frame (patch_begin -> add_patch_point only)
  static_assert(!BRS_PATCH_FRAME || PATCH_MAX == 6, "the patch frame is parked in words 7-9 of patch slots 0-5");
  static BRS_HD void patch_load(const Store<R>& st, Patch& Q) {
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int j = 0; j < 3; j++) { Q.Fm[3 * k + j] = st.getc(k, 7 + j); Q.Ph[3 * k + j] = st.getc(3 + k, 7 + j); }
    Q.c[0] = st.getf(0, 0); Q.c[1] = st.getf(0, 1); Q.c[2] = st.getf(0, 2);
  }
  // first point of a patch: frame from the unit normal nTf (torso frame), patch constants to LDS, and the CURRENT relative velocity
  // twist (V0, W0) in frame coordinates for the reference accelerations of its points
  static BRS_HD void patch_begin(Store<R>& st, const Frame& F, const ES& S, const R* nTf, Patch& Q, R* V0, R* W0) {
    R fw[9];
    world_frame(F, nTf, fw);
#pragma unroll
    for (int k = 0; k < 3; k++) { mulT_(F.RT, fw + 3 * k, Q.Fm + 3 * k); mulT_(F.RB, fw + 3 * k, Q.Ph + 3 * k); }
    R dW[3] = {-F.dTB[0], -F.dTB[1], -F.dTB[2]};  // x_B - x_T (world) -> torso frame -> frame coordinates
    mulT_(F.RT, dW, Q.cT);
    mul_(Q.Fm, Q.cT, Q.c);
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int j = 0; j < 3; j++) { st.setc(k, 7 + j, Q.Fm[3 * k + j]); st.setc(3 + k, 7 + j, Q.Ph[3 * k + j]); }
    st.setf(0, 0, Q.c[0]); st.setf(0, 1, Q.c[1]); st.setf(0, 2, Q.c[2]);
    // velocities: S.v / S.bv are WORLD linear, S.w / S.bw BODY angular; frame coordinates of a world vector u: fw . u
    R vT[3], vB[3], wT[3], wB[3], t[3];
    mul_(fw, S.v, vT); mul_(fw, S.bv, vB); mul_(Q.Fm, S.w, wT); mul_(Q.Ph, S.bw, wB);
    cross_(Q.c, wT, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { V0[k] = vB[k] - vT[k] + t[k]; W0[k] = wB[k] - wT[k]; }
  }
  static BRS_HD void add_patch_point(const Params<R>& P, Store<R>& st, Frame& F, const Patch& Q, const R* V0, const R* W0, const R* rT, R dist) {
    if (F.nc >= PATCH_MAX) return;
    const ContactClass<R>& c = P.cc[CC_BLOCK_ROBOT];
    R r[3], t[3];
    const R dT[3] = {rT[0] - Q.cT[0], rT[1] - Q.cT[1], rT[2] - Q.cT[2]};
    mul_(Q.Fm, dT, r);  // rho: the point seen from the block centre, frame coordinates
    cross_(W0, r, t);
    const R vn = V0[0] + t[0], vt1 = V0[1] + t[1], vt2 = V0[2] + t[2];
    const R imp = impedance_(c, dist);
    const int k = F.nc;
    st.setc(k, 0, r[0]); st.setc(k, 1, r[1]); st.setc(k, 2, r[2]);
    st.setc(k, 3, -c.B * vn - c.K * imp * (dist - c.margin));
    st.setc(k, 4, -c.B * c.mu * vt1);
    st.setc(k, 5, -c.B * c.mu * vt2);
    st.setc(k, 6, imp * rcp_((1 - imp) * c.cD));
    F.nc++;
  }
  // The 15-axis separation test once more, entirely from the fp64 poses (same formulas and tie rules as the fp32 code in
  // collide_coupled and as oracle/brs_oracle.c: bo_box_box_points).  Called only when the fp32 test sits within rounding of one
  // of its DISCRETE outcomes -- two face axes (or two edge axes) with the same separation, or the edge axis at its hysteresis
  // threshold against the face axis: which axis wins decides between different contact sets (one edge-edge point or a clipped
  // face patch; one reference face or its neighbour), and a kernel that takes that decision on fp32 roundings leaves the fp64
  // oracle by the whole difference of the two sets.  Round 3's second campaign seed found this: 3 env-steps of 3 M with a block
  // quaternion at 2.0-2.7e-4, each a jump at one substep with the contact set of the other branch.  ~330 fp64 operations, rare.
  struct Sat64 { double bestF, bestE; int axF, axE; bool sep; };
  static BRS_HD Sat64 sat15_f64(const Params<R>& P, const ES& S) {
    double qT[4] = {S.q[0], S.q[1], S.q[2], S.q[3]}, qB[4] = {S.bq[0], S.bq[1], S.bq[2], S.bq[3]}, T64[9], B64[9], RTB[9], Q[9], cg[3];
    quat2mat_(qT, T64); quat2mat_(qB, B64);
    const double d64[3] = {S.bp[0] - S.p[0], S.bp[1] - S.p[1], S.bp[2] - S.p[2]};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      cg[i] = T64[i] * d64[0] + T64[3 + i] * d64[1] + T64[6 + i] * d64[2] - (i == 2 ? P.torso_cz_d : 0.0);
#pragma unroll
      for (int j = 0; j < 3; j++) { RTB[3 * i + j] = T64[i] * B64[j] + T64[3 + i] * B64[3 + j] + T64[6 + i] * B64[6 + j]; Q[3 * i + j] = abs_(RTB[3 * i + j]); }
    }
    const double sT[3] = {P.torso_s_d[0], P.torso_s_d[1], P.torso_s_d[2]}, s = P.block_s_d, mg = P.margin_d[CC_BLOCK_ROBOT];
    Sat64 o;
    o.bestF = -1e300; o.bestE = -1e300; o.axF = 0; o.axE = -1; o.sep = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double sp = abs_(cg[k]) - sT[k] - s * (Q[3 * k] + Q[3 * k + 1] + Q[3 * k + 2]);
      o.sep = o.sep | (sp > mg);
      const bool better = sp > o.bestF;
      o.bestF = better ? sp : o.bestF; o.axF = better ? k : o.axF;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double dB = cg[0] * RTB[j] + cg[1] * RTB[3 + j] + cg[2] * RTB[6 + j];
      const double sp = abs_(dB) - s - (sT[0] * Q[j] + sT[1] * Q[3 + j] + sT[2] * Q[6 + j]);
      o.sep = o.sep | (sp > mg);
      const bool better = sp > o.bestF;
      o.bestF = better ? sp : o.bestF; o.axF = better ? 3 + j : o.axF;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const double len2 = 1.0 - RTB[3 * i + j] * RTB[3 * i + j];
        const double cl = cg[i2] * RTB[3 * i1 + j] - cg[i1] * RTB[3 * i2 + j];
        const double rT = sT[i1] * Q[3 * i2 + j] + sT[i2] * Q[3 * i1 + j], rB = s * (Q[3 * i + j1] + Q[3 * i + j2]);
        const bool ok = len2 >= 1e-6;
        const double sp = (abs_(cl) - rT - rB) * rsqrt64_(ok ? len2 : 1.0);
        o.sep = o.sep | (ok & (sp > mg));
        const bool better = ok & (sp > o.bestE);
        o.bestE = better ? sp : o.bestE; o.axE = better ? 3 * i + j : o.axE;
      }
    }
    return o;
  }
  // The wheel<->block candidates' distances once more from the fp64 poses: the deepest of (a) the 8 block vertices and (b) the
  // points of the 12 block edges nearest the cylinder axis, each against the capped cylinder, (c) two cylinder surface points
  // against the box -- same candidates and guards as collide_coupled (ii) and oracle/brs_oracle.c: bo_box_cyl_point.  Only the
  // DISTANCE is returned: it decides whether the wheel<->block contact exists in this substep, and that decision was the one
  // contact-existence test still taken in fp32 (seed 5 of round 3's campaign: a block arriving at a wheel had its contact one
  // substep early, 2.1e-4 on its quaternion).  Called only when the fp32 distance is within 2 um of the margin.
  static BRS_HD double wheel_block_dist_f64(const Params<R>& P, const ES& S, int wsel) {
    double qT[4] = {S.q[0], S.q[1], S.q[2], S.q[3]}, qB[4] = {S.bq[0], S.bq[1], S.bq[2], S.bq[3]}, T64[9], B64[9], RTB[9], d[3];
    quat2mat_(qT, T64); quat2mat_(qB, B64);
    const double dw[3] = {S.bp[0] - S.p[0], S.bp[1] - S.p[1], S.bp[2] - S.p[2]};
    const double wp[3] = {wsel == 1 ? -P.wheel_px_d : P.wheel_px_d, 0.0, P.wheel_pz_d};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      d[i] = T64[i] * dw[0] + T64[3 + i] * dw[1] + T64[6 + i] * dw[2] - wp[i];
#pragma unroll
      for (int j = 0; j < 3; j++) RTB[3 * i + j] = T64[i] * B64[j] + T64[3 + i] * B64[3 + j] + T64[6 + i] * B64[6 + j];
    }
    const double s = P.block_s_d, r = P.wheel_r_d, hl = P.wheel_hl_d;
    double best = 1e300;
    auto sd_cyl = [&](const double* p) -> double {
      const double rho2 = p[1] * p[1] + p[2] * p[2];
      const double rho = sqrt64_(rho2), drad = rho - r, dax = abs_(p[0]) - hl;
      const bool rim = (drad > 0) & (dax > 0);
      const double dist = rim ? sqrt64_(drad * drad + dax * dax) : max_(drad, dax);
      return rho2 < 1e-18 ? 1e300 : dist;
    };
    const double sb[3][3] = {{s * RTB[0], s * RTB[3], s * RTB[6]}, {s * RTB[1], s * RTB[4], s * RTB[7]}, {s * RTB[2], s * RTB[5], s * RTB[8]}};
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double p[3];
#pragma unroll
      for (int q = 0; q < 3; q++) p[q] = d[q] + ((i & 1) ? sb[0][q] : -sb[0][q]) + ((i & 2) ? sb[1][q] : -sb[1][q]) + ((i & 4) ? sb[2][q] : -sb[2][q]);
      best = min_(best, sd_cyl(p));
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double dir[3] = {RTB[j], RTB[3 + j], RTB[6 + j]};
      const double dd = dir[1] * dir[1] + dir[2] * dir[2];
      const bool dok = !(dd < 1e-8);
      const double idd = 1.0 / (dok ? dd : 1.0);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        double o[3];
#pragma unroll
        for (int q = 0; q < 3; q++) o[q] = d[q] + ((e & 1) ? sb[j1][q] : -sb[j1][q]) + ((e & 2) ? sb[j2][q] : -sb[j2][q]);
        const double tau = -(o[1] * dir[1] + o[2] * dir[2]) * idd;
        const double p[3] = {o[0] + tau * dir[0], o[1] + tau * dir[1], o[2] + tau * dir[2]};
        const double dist = sd_cyl(p);
        best = (dok & (abs_(tau) < s)) ? min_(best, dist) : best;
      }
    }
    const double rho2 = d[1] * d[1] + d[2] * d[2];
    if (rho2 > 1e-18) {
      const double rho = sqrt64_(rho2), ir = 1.0 / rho;
#pragma unroll
      for (int cand = 0; cand < 2; cand++) {
        double xc = min_(max_(d[0], -hl), hl), rq = r;
        if (cand == 1) { xc = d[0] >= 0 ? hl : -hl; rq = min_(rho, r); }
        const double rel[3] = {xc - d[0], rq * d[1] * ir - d[1], rq * d[2] * ir - d[2]};
        double p[3];
        mulT_(RTB, rel, p);
        const double q0 = abs_(p[0]) - s, q1 = abs_(p[1]) - s, q2 = abs_(p[2]) - s;
        const bool outside = (q0 > 0) | (q1 > 0) | (q2 > 0);
        const double m0 = max_(q0, 0.0), m1 = max_(q1, 0.0), m2 = max_(q2, 0.0);
        const double dist = outside ? sqrt64_(m0 * m0 + m1 * m1 + m2 * m2) : max_(q0, max_(q1, q2));
        best = min_(best, dist);
      }
    }
    return best;
  }
  // everything in the TORSO frame: block centre cB, block axes as columns of RTB = RT^T RB
  static BRS_HD void collide_coupled(const Params<R>& P, Store<R>& st, Frame& F, const ES& S) {
    const ContactClass<R>& c = P.cc[CC_BLOCK_ROBOT];
    R dW[3] = {-F.dTB[0], -F.dTB[1], -F.dTB[2]};  // x_B - x_T (world)
    R cB[3];
    mulT_(F.RT, dW, cB);  // block centre in the torso frame
    R d2 = dot_(cB, cB);
    R reach = P.torso_brad + P.torso_cz + P.block_brad + c.margin;  // generous: torso geom centre is cz up
    BRS_STAT(stats().cp_calls++);
    if (d2 > reach * reach) return;
    BRS_STAT(stats().cp_reach++);
    R RTB[9];  // block->torso
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) RTB[3 * i + j] = F.RT[i] * F.RB[j] + F.RT[3 + i] * F.RB[3 + j] + F.RT[6 + i] * F.RB[6 + j];
    R s = P.block_s;
    BRS_TIC(10);
    // (i) torso box <-> block box: the standard clipped-polygon box-box (15-axis SAT; face case: the incident face clipped
    // against the reference rectangle, <= 8 points, the 6 deepest kept; edge case: one point between the closest points of
    // the two edges).  Same specification as oracle/brs_oracle.c bo_box_box_points (axis choice, candidate order,
    // reduction); MuJoCo's mjc_BoxBox point sets are UNPINNED.  No runtime-indexed register arrays: the <= 16 clip
    // candidates are parked in the lane's LDS column (robot<->floor slot region, not yet written in this substep) and
    // only the kept ones are read back by the insertion loop.
    R cg[3] = {cB[0], cB[1], cB[2] - P.torso_cz};  // block centre relative to the torso geom centre
    R dd2 = dot_(cg, cg), rr0 = P.torso_brad + P.block_brad + c.margin;
    if (dd2 <= rr0 * rr0) {
      BRS_MARK("cc_sat");
      const R sT[3] = {P.torso_sx, P.torso_sy, P.torso_sz};
      R Q[9];
#pragma unroll
      for (int i = 0; i < 9; i++) Q[i] = abs_(RTB[i]);
      R bestF = (R)-1e30, bestE = (R)-1e30, bestF2 = (R)-1e30, bestE2 = (R)-1e30;  // (best and runner-up separations)
      double bestE64 = 0;
      int axF = 0, axE = -1;
      bool sep = false;
#if defined(BRS_PATCH_DIST32)
      const R sat_margin = c.margin;
#else
      // within a micrometre of the margin the fp32 axis test does not get to say "separated": the candidates' distances,
      // taken from the fp64 poses below, decide (face axes: the deepest vertex distance IS the axis separation)
      const R sat_margin = c.margin + (sizeof(R) == 4 ? (R)1e-6 : (R)0);
#endif
#pragma unroll
      for (int k = 0; k < 3; k++) {
        R sp = abs_(cg[k]) - sT[k] - s * (Q[3 * k] + Q[3 * k + 1] + Q[3 * k + 2]);
        sep = sep | (sp > sat_margin);
        const bool better = sp > bestF;
        bestF2 = better ? bestF : max_(bestF2, sp);
        bestF = better ? sp : bestF; axF = better ? k : axF;
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        R dB = cg[0] * RTB[j] + cg[1] * RTB[3 + j] + cg[2] * RTB[6 + j];
        R sp = abs_(dB) - s - (sT[0] * Q[j] + sT[1] * Q[3 + j] + sT[2] * Q[6 + j]);
        sep = sep | (sp > sat_margin);
        const bool better = sp > bestF;
        bestF2 = better ? bestF : max_(bestF2, sp);
        bestF = better ? sp : bestF; axF = better ? 3 + j : axF;
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
          // |e_i x b_j|^2 as the sum of squares of the cross product's two components, NOT as 1 - r_ij^2: for nearly parallel edges
          // (a block lying flat on the torso: len2 ~ 1e-5) the fp32 difference carries 1e-7 of absolute error, i.e. 1 % of len2 -- the
          // separation of such an axis came out wrong by 1-2 mm and beat the face axis (round 3, seed-1 campaign: a 4-point patch
          // replaced by one edge-edge point for a substep)
          R len2 = RTB[3 * i1 + j] * RTB[3 * i1 + j] + RTB[3 * i2 + j] * RTB[3 * i2 + j];
          R cl = cg[i2] * RTB[3 * i1 + j] - cg[i1] * RTB[3 * i2 + j];
          R rT = sT[i1] * Q[3 * i2 + j] + sT[i2] * Q[3 * i1 + j], rB = s * (Q[3 * i + j1] + Q[3 * i + j2]);
          R sp = (abs_(cl) - rT - rB) * rsqrt_(max_(len2, (R)1e-12));
          const bool ok = len2 >= (R)1e-6;
          sep = sep | (ok & (sp > sat_margin));
          const bool better = ok & (sp > bestE);
          bestE2 = better ? bestE : (ok ? max_(bestE2, sp) : bestE2);
          bestE = better ? sp : bestE; axE = better ? 3 * i + j : axE;
        }
      }
      bool use_edge = (axE >= 0) & (bestE > bestF + (R)0.05 * abs_(bestF) + (R)1e-5);
#if !defined(BRS_PATCH_DIST32)
      if (sizeof(R) == 4 && !sep) {
        // a DISCRETE outcome within rounding of a tie (fp32 separations carry ~1e-8 m; band 2e-6 m): two face axes level, the edge
        // axis at its threshold against the face axis, or -- where the edge axis may win -- two edge axes level.  Then all
        // fifteen separations are taken from the fp64 poses and the choices from those (sat15_f64)
        const R band = (R)2e-6, thr = bestF + (R)0.05 * abs_(bestF) + (R)1e-5;
        const bool tie = (bestF - bestF2 < band) | ((axE >= 0) & ((abs_(bestE - thr) < band) | ((bestE > thr - band) & (bestE - bestE2 < band))));
        if (tie) {
          const Sat64 d = sat15_f64(P, S);
          sep = d.sep; axF = d.axF; axE = d.axE; bestE = (R)d.bestE;
          use_edge = (d.axE >= 0) & (d.bestE > d.bestF + 0.05 * abs_(d.bestF) + 1e-5);
        }
      }
#endif
      if (!sep) {
        BRS_MARK("cc_branch");
        if (use_edge) {
          const int i = axE / 3, j = axE - 3 * i;
          const int i1 = i == 2 ? 0 : i + 1, i2 = i == 0 ? 2 : i - 1;
#if defined(BRS_PATCH_DIST32)
          const bool edge_in = bestE < c.margin;
#else
          // the separation along the chosen edge axis L = e_i x b_j once more from the fp64 poses: it is the distance of the
          // patch's only point, and `< margin` decides whether that point exists in this substep (like the face case below;
          // the axis itself is the fp32 choice).  ~90 fp64 operations, only for lanes in the edge case.
          {
            double qT[4] = {S.q[0], S.q[1], S.q[2], S.q[3]}, qB[4] = {S.bq[0], S.bq[1], S.bq[2], S.bq[3]}, T64[9], B64[9];
            quat2mat_(qT, T64); quat2mat_(qB, B64);
            const double d64[3] = {S.bp[0] - S.p[0], S.bp[1] - S.p[1], S.bp[2] - S.p[2]};
            const int j1 = j == 2 ? 0 : j + 1, j2 = j == 0 ? 2 : j - 1;
            auto colT = [&](int a, double* o) { o[0] = pick3(a, T64[0], T64[1], T64[2]); o[1] = pick3(a, T64[3], T64[4], T64[5]); o[2] = pick3(a, T64[6], T64[7], T64[8]); };
            auto colB = [&](int a, double* o) { o[0] = pick3(a, B64[0], B64[1], B64[2]); o[1] = pick3(a, B64[3], B64[4], B64[5]); o[2] = pick3(a, B64[6], B64[7], B64[8]); };
            double ti[3], ti1[3], ti2[3], bjw[3], bj1w[3], bj2w[3];
            colT(i, ti); colT(i1, ti1); colT(i2, ti2); colB(j, bjw); colB(j1, bj1w); colB(j2, bj2w);
            const double Rij = dot_(ti, bjw), Ri1j = dot_(ti1, bjw), Ri2j = dot_(ti2, bjw), Rij1 = dot_(ti, bj1w), Rij2 = dot_(ti, bj2w);
            const double cgi1 = dot_(ti1, d64) - (i1 == 2 ? P.torso_cz_d : 0.0), cgi2 = dot_(ti2, d64) - (i2 == 2 ? P.torso_cz_d : 0.0);
            const double cl = cgi2 * Ri1j - cgi1 * Ri2j;
            const double rT64 = pick3(i1, P.torso_s_d) * abs_(Ri2j) + pick3(i2, P.torso_s_d) * abs_(Ri1j), rB64 = P.block_s_d * (abs_(Rij1) + abs_(Rij2));
            const double len2 = 1.0 - Rij * Rij;  // >= 1e-6: the axis passed the fp32 parallel test
            bestE64 = (abs_(cl) - rT64 - rB64) * rsqrt64_(len2);
          }
          const bool edge_in = bestE64 < P.margin_d[CC_BLOCK_ROBOT];
          bestE = (R)bestE64;
#endif
          if (edge_in) {
            R bj[3] = {pick3(j, RTB[0], RTB[1], RTB[2]), pick3(j, RTB[3], RTB[4], RTB[5]), pick3(j, RTB[6], RTB[7], RTB[8])};
            const R bji = pick3(i, bj), bji1 = pick3(i1, bj), bji2 = pick3(i2, bj);
            const R lenE2 = bji1 * bji1 + bji2 * bji2;  // = 1 - bji^2 without the cancellation
            R il = rsqrt_(lenE2);
            // L = e_i x b_j : L[i1] = -b_j[i2], L[i2] = b_j[i1]
            R Lv1 = -bji2 * il, Lv2 = bji1 * il;
            R L[3];
#pragma unroll
            for (int m = 0; m < 3; m++) L[m] = m == i1 ? Lv1 : (m == i2 ? Lv2 : (R)0);
            R sgn = dot_(L, cg) < 0 ? (R)-1 : (R)1;
#pragma unroll
            for (int m = 0; m < 3; m++) L[m] *= sgn;
            R pA[3], pB[3] = {cg[0], cg[1], cg[2]};
#pragma unroll
            for (int m = 0; m < 3; m++) pA[m] = m == i ? (R)0 : (L[m] >= 0 ? sT[m] : -sT[m]);
#pragma unroll
            for (int m = 0; m < 3; m++) {
              R bm[3] = {RTB[m], RTB[3 + m], RTB[6 + m]};
              R sg = m == j ? (R)0 : (dot_(L, bm) >= 0 ? -s : s);
#pragma unroll
              for (int q = 0; q < 3; q++) pB[q] += sg * bm[q];
            }
            R w[3] = {pA[0] - pB[0], pA[1] - pB[1], pA[2] - pB[2]};
            R dA = pick3(i, w), dBv = dot_(w, bj), iden = rcp_(lenE2);
            R al = (bji * dBv - dA) * iden, be = (dBv - bji * dA) * iden;
            const R sTi = pick3(i, sT);
            al = max_(-sTi, min_(sTi, al)); be = max_(-s, min_(s, be));
            R pos[3];
#pragma unroll
            for (int m = 0; m < 3; m++) pos[m] = (R)0.5 * (pA[m] + (m == i ? al : (R)0) + pB[m] + be * bj[m]);
            pos[2] += P.torso_cz;
#if BRS_PATCH_FRAME
            Patch Q;
            R V0[3], W0[3];
            patch_begin(st, F, S, L, Q, V0, W0);
            add_patch_point(P, st, F, Q, V0, W0, pos, bestE);
#else
            R fw[9];
            world_frame(F, L, fw);
            add_coupled(P, st, F, S, pos, fw, bestE, 0, false);
#endif
          }
        } else {
          // face case.  Reference frame coordinates (u, v, g): u, v span the reference rectangle |u| <= ra, |v| <= rb, g is
          // the signed distance to the reference face.  Incident face = centre Cc +- H1 +- H2 in those coordinates; a
          // candidate (u, v, g) maps back to the torso geom frame as  pos = u A1 + v A2 + (half + g/2) A3 + A0.
          BRS_MARK("cc_face_setup");
          R Cc[3], H1[3], H2[3], A0[3], A1[3], A2[3], A3[3], nrm[3], ra, rb, half;
#if !defined(BRS_PATCH_DIST32)
          // The NORMAL components of the incident face (centre and half edges along the reference normal) once more, from
          // the fp64 poses: every candidate's signed distance g is affine in these three numbers, and g < margin decides
          // whether a patch point exists in this substep -- in fp32 its rounding (~1e-8 m) put points of the patch one
          // substep apart from the fp64 oracle (the block-quaternion outliers of DESIGN.md 2.1).  ~120 fp64 operations.
          double Cc2, H12, H22, fsg = 1.0, fsj = 1.0;
          int fsel = 0;
          {
            double qT[4] = {S.q[0], S.q[1], S.q[2], S.q[3]}, qB[4] = {S.bq[0], S.bq[1], S.bq[2], S.bq[3]}, T64[9], B64[9];
            quat2mat_(qT, T64); quat2mat_(qB, B64);
            const double d64[3] = {S.bp[0] - S.p[0], S.bp[1] - S.p[1], S.bp[2] - S.p[2]};
            const double cz = P.torso_cz_d, sd = P.block_s_d;
            const double sTd[3] = {P.torso_s_d[0], P.torso_s_d[1], P.torso_s_d[2]};
            if (axF < 3) {
              const int k = axF;
              const double ck[3] = {pick3(k, T64[0], T64[1], T64[2]), pick3(k, T64[3], T64[4], T64[5]), pick3(k, T64[6], T64[7], T64[8])};  // column k of RT
              const double rk[3] = {ck[0] * B64[0] + ck[1] * B64[3] + ck[2] * B64[6], ck[0] * B64[1] + ck[1] * B64[4] + ck[2] * B64[7],
                                    ck[0] * B64[2] + ck[1] * B64[5] + ck[2] * B64[8]};  // row k of RTB
              const double cgk = ck[0] * d64[0] + ck[1] * d64[1] + ck[2] * d64[2] - (k == 2 ? cz : 0.0);
              // the DISCRETE choices of the face case -- which side of the reference face, which face of the other box is
              // incident and its sign -- are taken HERE, on the fp64 values, and handed to the fp32 code below (fsel, fsg, fsj):
              // a block turned 45 degrees about the normal has two faces equally anti-parallel to it, and the two choices are
              // different patches
              const double sg = cgk >= 0 ? 1.0 : -1.0;
              int js = 0;
              if (abs_(rk[1]) > abs_(rk[0])) js = 1;
              if (abs_(rk[2]) > abs_(pick3(js, rk))) js = 2;
              {
                const double sj = -sg * (pick3(js, rk) >= 0 ? 1.0 : -1.0);
                const int a1 = js == 2 ? 0 : js + 1, a2 = js == 0 ? 2 : js - 1;
                Cc2 = sg * (cgk + sj * sd * pick3(js, rk)) - pick3(k, sTd);
                H12 = sg * sd * pick3(a1, rk); H22 = sg * sd * pick3(a2, rk);
                fsel = js; fsg = sg; fsj = sj;
              }
            } else {
              const int j = axF - 3;
              const double cj[3] = {pick3(j, B64[0], B64[1], B64[2]), pick3(j, B64[3], B64[4], B64[5]), pick3(j, B64[6], B64[7], B64[8])};  // block axis j, world
              const double bj64[3] = {T64[0] * cj[0] + T64[3] * cj[1] + T64[6] * cj[2], T64[1] * cj[0] + T64[4] * cj[1] + T64[7] * cj[2],
                                      T64[2] * cj[0] + T64[5] * cj[1] + T64[8] * cj[2]};  // the same axis in the torso frame
              const double dotbc = cj[0] * d64[0] + cj[1] * d64[1] + cj[2] * d64[2] - bj64[2] * cz;  // bj . cg
              const double sgB = dotbc >= 0 ? 1.0 : -1.0;  // (discrete choices on the fp64 values, as above)
              int ks = 0;
              if (abs_(bj64[1]) > abs_(bj64[0])) ks = 1;
              if (abs_(bj64[2]) > abs_(pick3(ks, bj64))) ks = 2;
              const double sk = sgB * (pick3(ks, bj64) >= 0 ? 1.0 : -1.0);
              fsel = ks; fsg = sgB; fsj = sk;
              const int a1 = ks == 2 ? 0 : ks + 1, a2 = ks == 0 ? 2 : ks - 1;
              Cc2 = -sgB * (sk * pick3(ks, sTd) * pick3(ks, bj64) - dotbc) - sd;
              H12 = -sgB * pick3(a1, sTd) * pick3(a1, bj64); H22 = -sgB * pick3(a2, sTd) * pick3(a2, bj64);
            }
          }
#endif
          if (axF < 3) {
            const int k = axF, j1 = k == 2 ? 0 : k + 1, j2 = k == 0 ? 2 : k - 1;
#if defined(BRS_PATCH_DIST32)
            const R sg = pick3(k, cg) >= 0 ? (R)1 : (R)-1;
            R rk[3] = {pick3(k, RTB[0], RTB[3], RTB[6]), pick3(k, RTB[1], RTB[4], RTB[7]), pick3(k, RTB[2], RTB[5], RTB[8])};  // row k
            int js = 0;
            if (abs_(rk[1]) > abs_(rk[0])) js = 1;
            if (abs_(rk[2]) > abs_(pick3(js, rk))) js = 2;
            const R sj = -sg * (pick3(js, rk) >= 0 ? (R)1 : (R)-1);
#else
            const R sg = (R)fsg, sj = (R)fsj;  // side, incident face and its sign: decided above on the fp64 values
            const int js = fsel;
#endif
            const int a1 = js == 2 ? 0 : js + 1, a2 = js == 0 ? 2 : js - 1;
            // block axes (torso frame) js, a1, a2 = columns of RTB
            R bs[3] = {pick3(js, RTB[0], RTB[1], RTB[2]), pick3(js, RTB[3], RTB[4], RTB[5]), pick3(js, RTB[6], RTB[7], RTB[8])};
            R b1[3] = {pick3(a1, RTB[0], RTB[1], RTB[2]), pick3(a1, RTB[3], RTB[4], RTB[5]), pick3(a1, RTB[6], RTB[7], RTB[8])};
            R b2[3] = {pick3(a2, RTB[0], RTB[1], RTB[2]), pick3(a2, RTB[3], RTB[4], RTB[5]), pick3(a2, RTB[6], RTB[7], RTB[8])};
            R pc[3] = {cg[0] + sj * s * bs[0], cg[1] + sj * s * bs[1], cg[2] + sj * s * bs[2]};
            const R sTk = pick3(k, sT);
            Cc[0] = pick3(j1, pc); Cc[1] = pick3(j2, pc); Cc[2] = sg * pick3(k, pc) - sTk;
            H1[0] = s * pick3(j1, b1); H1[1] = s * pick3(j2, b1); H1[2] = sg * s * pick3(k, b1);
            H2[0] = s * pick3(j1, b2); H2[1] = s * pick3(j2, b2); H2[2] = sg * s * pick3(k, b2);
            ra = pick3(j1, sT); rb = pick3(j2, sT); half = sTk;
#pragma unroll
            for (int m = 0; m < 3; m++) {
              A0[m] = 0; A1[m] = m == j1 ? (R)1 : (R)0; A2[m] = m == j2 ? (R)1 : (R)0; A3[m] = m == k ? sg : (R)0;
              nrm[m] = m == k ? sg : (R)0;
            }
          } else {
            const int j = axF - 3, i1 = j == 2 ? 0 : j + 1, i2 = j == 0 ? 2 : j - 1;
            R bj[3] = {pick3(j, RTB[0], RTB[1], RTB[2]), pick3(j, RTB[3], RTB[4], RTB[5]), pick3(j, RTB[6], RTB[7], RTB[8])};
            R bi1[3] = {pick3(i1, RTB[0], RTB[1], RTB[2]), pick3(i1, RTB[3], RTB[4], RTB[5]), pick3(i1, RTB[6], RTB[7], RTB[8])};
            R bi2[3] = {pick3(i2, RTB[0], RTB[1], RTB[2]), pick3(i2, RTB[3], RTB[4], RTB[5]), pick3(i2, RTB[6], RTB[7], RTB[8])};
#if defined(BRS_PATCH_DIST32)
            const R sgB = dot_(cg, bj) >= 0 ? (R)1 : (R)-1;
            int ks = 0;
            if (abs_(bj[1]) > abs_(bj[0])) ks = 1;
            if (abs_(bj[2]) > abs_(pick3(ks, bj))) ks = 2;
            const R sk = sgB * (pick3(ks, bj) >= 0 ? (R)1 : (R)-1);
#else
            const R sgB = (R)fsg, sk = (R)fsj;
            const int ks = fsel;
#endif
            const int a1 = ks == 2 ? 0 : ks + 1, a2 = ks == 0 ? 2 : ks - 1;
            const R sTs = pick3(ks, sT), sT1 = pick3(a1, sT), sT2 = pick3(a2, sT);
            // incident torso face: centre sk sT[ks] e_ks, half edges sT[a1] e_a1, sT[a2] e_a2; into the block frame: R^T (x - cg)
            R rel[3];
#pragma unroll
            for (int m = 0; m < 3; m++) rel[m] = (m == ks ? sk * sTs : (R)0) - cg[m];
            Cc[0] = dot_(bi1, rel); Cc[1] = dot_(bi2, rel); Cc[2] = -sgB * dot_(bj, rel) - s;
            H1[0] = sT1 * pick3(a1, bi1); H1[1] = sT1 * pick3(a1, bi2); H1[2] = -sgB * sT1 * pick3(a1, bj);
            H2[0] = sT2 * pick3(a2, bi1); H2[1] = sT2 * pick3(a2, bi2); H2[2] = -sgB * sT2 * pick3(a2, bj);
            ra = s; rb = s; half = s;
#pragma unroll
            for (int m = 0; m < 3; m++) { A0[m] = cg[m]; A1[m] = bi1[m]; A2[m] = bi2[m]; A3[m] = -sgB * bj[m]; nrm[m] = sgB * bj[m]; }
          }
          BRS_MARK("cc_clip");
          // incident quad, in order around the face
          R V[4][3];
#pragma unroll
          for (int q = 0; q < 3; q++) {
            V[0][q] = Cc[q] - H1[q] - H2[q]; V[1][q] = Cc[q] + H1[q] - H2[q];
            V[2][q] = Cc[q] + H1[q] + H2[q]; V[3][q] = Cc[q] - H1[q] + H2[q];
          }
          // the <= 16 clip candidates (u, v, g) stay in registers (static indices); `vmask` = the valid ones
          R cu_[16], cv_[16], gq[16];
          uint32_t vmask = 0;
          bool ins[4];
#if defined(BRS_PATCH_DIST32)
          typedef R GT;
          const GT Vg[4] = {V[0][2], V[1][2], V[2][2], V[3][2]}, H12x2 = 2 * H1[2], H22x2 = 2 * H2[2];
#else
          typedef double GT;
          const GT Vg[4] = {Cc2 - H12 - H22, Cc2 + H12 - H22, Cc2 + H12 + H22, Cc2 - H12 + H22}, H12x2 = 2 * H12, H22x2 = 2 * H22;
#endif
#if defined(BRS_PATCH_DIST32)
          const GT gmargin = (GT)c.margin;
#else
          const GT gmargin = P.margin_d[CC_BLOCK_ROBOT];
#endif
#pragma unroll
          for (int v = 0; v < 4; v++) {
            ins[v] = (abs_(V[v][0]) <= ra) & (abs_(V[v][1]) <= rb);
            vmask |= (ins[v] & (Vg[v] < gmargin)) ? (1u << v) : 0u;
            cu_[v] = V[v][0]; cv_[v] = V[v][1]; gq[v] = (R)Vg[v];
          }
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int e1 = (e + 1) & 3;
            const R du = V[e1][0] - V[e][0], dv = V[e1][1] - V[e][1];
            const GT dg = Vg[e1] - Vg[e];
            // Liang-Barsky against |u| <= ra, |v| <= rb
            R t0 = 0, t1 = 1;
            bool ok = true;
            const R pp[4] = {-du, du, -dv, dv}, qq[4] = {V[e][0] + ra, ra - V[e][0], V[e][1] + rb, rb - V[e][1]};
            const bool zu = du == (R)0, zv = dv == (R)0;
            const R idu = rcp_(zu ? (R)1 : du), idv = rcp_(zv ? (R)1 : dv);  // one reciprocal per direction, sign applied below
            const R ip[4] = {-idu, idu, -idv, idv};
#pragma unroll
            for (int b4 = 0; b4 < 4; b4++) {
              const bool zero = b4 < 2 ? zu : zv;
              const R r = qq[b4] * ip[b4];
              ok = ok & !(zero & (qq[b4] < 0));
              t0 = (!zero & (pp[b4] < 0) & (r > t0)) ? r : t0;
              t1 = (!zero & (pp[b4] > 0) & (r < t1)) ? r : t1;
            }
            ok = ok & (t0 < t1);
            const GT g0 = Vg[e] + (GT)t0 * dg, g1 = Vg[e] + (GT)t1 * dg;
            vmask |= (ok & !ins[e] & (g0 < gmargin)) ? (1u << (4 + 2 * e)) : 0u;
            vmask |= (ok & !ins[e1] & (g1 < gmargin)) ? (1u << (5 + 2 * e)) : 0u;
            cu_[4 + 2 * e] = V[e][0] + t0 * du; cv_[4 + 2 * e] = V[e][1] + t0 * dv; gq[4 + 2 * e] = (R)g0;
            cu_[5 + 2 * e] = V[e][0] + t1 * du; cv_[5 + 2 * e] = V[e][1] + t1 * dv; gq[5 + 2 * e] = (R)g1;
          }
          {  // rectangle corners inside the incident parallelogram: corner = V0 + al (V1 - V0) + be (V3 - V0)
            const R e1u = 2 * H1[0], e1v = 2 * H1[1], e2u = 2 * H2[0], e2v = 2 * H2[1];
            const R det = e1u * e2v - e1v * e2u;
            const bool dok = abs_(det) > (R)1e-30;
            const R idet = rcp_(dok ? det : (R)1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const R cu = (q & 1) ? ra : -ra, cv = (q & 2) ? rb : -rb, ru = cu - V[0][0], rv = cv - V[0][1];
              const R al = (ru * e2v - rv * e2u) * idet, be = (e1u * rv - e1v * ru) * idet;
              const GT g = Vg[0] + (GT)al * H12x2 + (GT)be * H22x2;
              const bool in = dok & (al > 0) & (al < 1) & (be > 0) & (be < 1);
              vmask |= (in & (g < gmargin)) ? (1u << (12 + q)) : 0u;
              cu_[12 + q] = cu; cv_[12 + q] = cv; gq[12 + q] = (R)g;
            }
          }
          BRS_MARK("cc_reduce");
          // keep the PATCH_MAX = 6 deepest (ties: lower candidate index).  tau = 6th smallest valid g from a sorting network
          // (sorted groups of 4, bitonic merges to two sorted octets, low half of their merge) -- branch-free inside: a
          // per-lane selection loop would cost every lane its worst case
          uint32_t keep = vmask;
#if defined(BRS_ALWAYS_REDUCE)  // A/B only
          {
#else
          if (__builtin_popcount(vmask) > PATCH_MAX) {  // rare with 6 slots: a wave usually skips the network
#endif
            R k_[16];
#pragma unroll
            for (int q = 0; q < 16; q++) k_[q] = ((vmask >> q) & 1u) ? gq[q] : (R)1e30;
#define BRS_CSWAP(a, b) { const R lo_ = min_(k_[a], k_[b]), hi_ = max_(k_[a], k_[b]); k_[a] = lo_; k_[b] = hi_; }
#pragma unroll
            for (int g4 = 0; g4 < 16; g4 += 4) {
              BRS_CSWAP(g4 + 0, g4 + 1); BRS_CSWAP(g4 + 2, g4 + 3); BRS_CSWAP(g4 + 0, g4 + 2); BRS_CSWAP(g4 + 1, g4 + 3); BRS_CSWAP(g4 + 1, g4 + 2);
            }
#pragma unroll
            for (int o8 = 0; o8 < 16; o8 += 8) {  // two sorted quartets -> one sorted octet
              BRS_CSWAP(o8 + 0, o8 + 7); BRS_CSWAP(o8 + 1, o8 + 6); BRS_CSWAP(o8 + 2, o8 + 5); BRS_CSWAP(o8 + 3, o8 + 4);
#pragma unroll
              for (int h4 = 0; h4 < 8; h4 += 4) {
                BRS_CSWAP(o8 + h4 + 0, o8 + h4 + 2); BRS_CSWAP(o8 + h4 + 1, o8 + h4 + 3); BRS_CSWAP(o8 + h4 + 0, o8 + h4 + 1); BRS_CSWAP(o8 + h4 + 2, o8 + h4 + 3);
              }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) k_[q] = min_(k_[q], k_[15 - q]);  // the 8 smallest of all (bitonic)
            BRS_CSWAP(0, 4); BRS_CSWAP(1, 5); BRS_CSWAP(2, 6); BRS_CSWAP(3, 7);
#pragma unroll
            for (int h4 = 0; h4 < 8; h4 += 4) {
              BRS_CSWAP(h4 + 0, h4 + 2); BRS_CSWAP(h4 + 1, h4 + 3); BRS_CSWAP(h4 + 0, h4 + 1); BRS_CSWAP(h4 + 2, h4 + 3);
            }
#undef BRS_CSWAP
            const R tau = k_[PATCH_MAX - 1];
            uint32_t lt = 0, eq = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
              const bool vq = ((vmask >> q) & 1u) != 0;
              lt |= (vq & (gq[q] < tau)) ? (1u << q) : 0u;
              eq |= (vq & (gq[q] == tau)) ? (1u << q) : 0u;
            }
            int need = PATCH_MAX - (int)__builtin_popcount(lt);
            uint32_t kp = lt;
#pragma unroll
            for (int t = 0; t < PATCH_MAX; t++) {
              const bool go = (t < need) & (eq != 0u);
              const uint32_t lowbit = eq & (0u - eq);
              kp |= go ? lowbit : 0u;
              eq &= go ? ~lowbit : ~0u;
            }
            keep = __builtin_popcount(vmask) > PATCH_MAX ? kp : keep;
          }
          BRS_MARK("cc_scatter");
          const int nkeep = (int)__builtin_popcount(keep);
          R* scr = st.base + (SLOT_ROBOT * SLOT_WORDS) * st.stride;
#if BRS_FIXED_SCATTER
          // all 16 candidates parked at FIXED words of the (not yet written) robot<->floor slot region -- 48 stores at immediate
          // offsets, no address arithmetic; the insertion loop walks `keep` by bit scan (ascending candidate index = rank order)
          static_assert(16 * 3 <= N_ROBOT_SLOTS * SLOT_WORDS, "the candidate scratch must fit the robot<->floor slot region");
#pragma unroll
          for (int q = 0; q < 16; q++) {
            scr[(3 * q) * st.stride] = cu_[q]; scr[(3 * q + 1) * st.stride] = cv_[q]; scr[(3 * q + 2) * st.stride] = gq[q];
          }
#else
          // compaction through the lane's LDS column: kept candidate with rank r -> scratch words 3r .. 3r+2 of the (not yet
          // written) robot<->floor slot region, everything else -> a dump slot; the insertion loop then reads by rank
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const bool kq = ((keep >> q) & 1u) != 0;
            const int rank = (int)__builtin_popcount(keep & ((1u << q) - 1u));
            R* dst = scr + (3 * (kq ? rank : PATCH_MAX)) * st.stride;
            dst[0] = cu_[q]; dst[st.stride] = cv_[q]; dst[2 * st.stride] = gq[q];
          }
#endif
          BRS_MARK("cc_insert");
          if (nkeep > 0) {
#if BRS_PATCH_FRAME
            Patch Q;
            R V0[3], W0[3];
            patch_begin(st, F, S, nrm, Q, V0, W0);  // one contact frame for the whole patch
#else
            R fw[9];
            world_frame(F, nrm, fw);  // one contact frame for the whole patch
#endif
#if BRS_FIXED_SCATTER
            uint32_t walk = keep;
#endif
            for (int r = 0; r < nkeep; r++) {
#if BRS_FIXED_SCATTER
              const int qi = __builtin_ctz(walk);
              walk &= walk - 1u;
              const R u = scr[(3 * qi) * st.stride], v = scr[(3 * qi + 1) * st.stride], g = scr[(3 * qi + 2) * st.stride];
#else
              const R u = scr[(3 * r) * st.stride], v = scr[(3 * r + 1) * st.stride], g = scr[(3 * r + 2) * st.stride];
#endif
              const R wv = half + (R)0.5 * g;
              R pos[3] = {A0[0] + u * A1[0] + v * A2[0] + wv * A3[0], A0[1] + u * A1[1] + v * A2[1] + wv * A3[1],
                          A0[2] + u * A1[2] + v * A2[2] + wv * A3[2] + P.torso_cz};
#if BRS_PATCH_FRAME
              add_patch_point(P, st, F, Q, V0, W0, pos, g);
#else
              add_coupled(P, st, F, S, pos, fw, g, 0, r > 0);
#endif
            }
          }
        }
      }
    }
    BRS_MARK("cc_wheels");
    BRS_TOC(10);
    BRS_TIC(11);
    // (ii) wheel cylinder <-> block box: ONE point per wheel, the deepest of the closest-feature candidates (a) block
    // vertices vs cylinder, (b) block edges vs barrel, (c) two cylinder surface points vs box -- same order and rules as
    // oracle/brs_oracle.c bo_box_cyl_point (MuJoCo: general convex collider, one point; UNPINNED)
    // ONE pass, on the wheel on the block's side of the robot: the wheels' inner faces are 2 (wheel_px - wheel_hl) = 12.2 cm
    // apart, the block's diameter plus twice the margin is 7.3 cm -- it can never be within the margin of both, so testing
    // the far wheel (as the oracle does) cannot produce a contact.  Halves the path for every wave that walks it.
    {
      const int wsel = cB[0] < 0 ? 1 : 2;
      R wp[3] = {wsel == 1 ? -P.wheel_px : P.wheel_px, (R)0, P.wheel_pz};
      R d[3] = {cB[0] - wp[0], cB[1] - wp[1], cB[2] - wp[2]};
      R rr = P.wheel_brad + P.block_brad + c.margin;
#ifdef BRS_NO_WHEELS
      if (true) { BRS_TOC(11); return; }  // ablation only
#endif
      if (dot_(d, d) > rr * rr) { BRS_TOC(11); return; }
#if defined(BRS_PATCH_DIST32)
      const R wband = (R)0;
#else
      // candidates within 2 um BEYOND the margin are still tracked: whether the contact exists is then decided below on the
      // distance taken from the fp64 poses (wheel_block_dist_f64), like every other contact-existence test
      const R wband = sizeof(R) == 4 ? (R)2e-6 : (R)0;
#endif
      R best = c.margin + wband, bpos[3] = {0, 0, 0}, bn[3] = {0, 0, 1}, wq[3] = {0, 0, 0};
      bool found = false;
      // (a) + (b): block points against the cylinder -- only the winning POINT is tracked in the loops (4 selects per
      // candidate); its normal and contact position are rebuilt once afterwards
      const R sb[3][3] = {{s * RTB[0], s * RTB[3], s * RTB[6]}, {s * RTB[1], s * RTB[4], s * RTB[7]}, {s * RTB[2], s * RTB[5], s * RTB[8]}};
      auto sd_cyl = [&](const R* p, R& rho2) {
        rho2 = p[1] * p[1] + p[2] * p[2];
        const R rho = sqrt_(rho2), drad = rho - P.wheel_r, dax = abs_(p[0]) - P.wheel_hl;
        const bool rim = (drad > 0) & (dax > 0);
        const R drim = sqrt_(drad * drad + dax * dax);
        return rim ? drim : max_(drad, dax);
      };
#pragma unroll
      for (int i = 0; i < 8; i++) {
        R p[3];
#pragma unroll
        for (int q = 0; q < 3; q++) p[q] = d[q] + ((i & 1) ? sb[0][q] : -sb[0][q]) + ((i & 2) ? sb[1][q] : -sb[1][q]) + ((i & 4) ? sb[2][q] : -sb[2][q]);
        R rho2;
        const R dist = sd_cyl(p, rho2);
        const bool take = !(rho2 < (R)1e-18) & (dist < best);
        best = take ? dist : best;
        found = found | take;
#pragma unroll
        for (int q = 0; q < 3; q++) wq[q] = take ? p[q] : wq[q];
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const R dir[3] = {RTB[j], RTB[3 + j], RTB[6 + j]};
        const R dd = dir[1] * dir[1] + dir[2] * dir[2];
        const bool dok = !(dd < (R)1e-8);
        const R idd = rcp_(dok ? dd : (R)1);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          R o[3];
#pragma unroll
          for (int q = 0; q < 3; q++) o[q] = d[q] + ((e & 1) ? sb[j1][q] : -sb[j1][q]) + ((e & 2) ? sb[j2][q] : -sb[j2][q]);
          const R tau = -(o[1] * dir[1] + o[2] * dir[2]) * idd;
          const R p[3] = {o[0] + tau * dir[0], o[1] + tau * dir[1], o[2] + tau * dir[2]};
          R rho2;
          const R dist = sd_cyl(p, rho2);
          const bool take = dok & (abs_(tau) < s) & !(rho2 < (R)1e-18) & (dist < best);
          best = take ? dist : best;
          found = found | take;
#pragma unroll
          for (int q = 0; q < 3; q++) wq[q] = take ? p[q] : wq[q];
        }
      }
      {  // normal of the cylinder at the winning point (the point-cylinder rule of the oracle's point_cyl)
        const R rho = sqrt_(wq[1] * wq[1] + wq[2] * wq[2]), drad = rho - P.wheel_r, dax = abs_(wq[0]) - P.wheel_hl;
        const bool rim = (drad > 0) & (dax > 0), radial = drad >= dax;
        const R ir = rcp_(max_(rho, (R)1e-9)), idr = rcp_(max_(sqrt_(drad * drad + dax * dax), (R)1e-20));
        const R sx = wq[0] >= 0 ? (R)1 : (R)-1;
        const R kx = rim ? dax * idr : (radial ? (R)0 : (R)1), kr = rim ? drad * idr : (radial ? (R)1 : (R)0);
        bn[0] = sx * kx; bn[1] = kr * wq[1] * ir; bn[2] = kr * wq[2] * ir;
#pragma unroll
        for (int q = 0; q < 3; q++) bpos[q] = wq[q] + wp[q] - bn[q] * best * (R)0.5;
      }
      R xid = d[0], rho = sqrt_(d[1] * d[1] + d[2] * d[2]);
      if (rho > (R)1e-9) {
        R ir = rcp_(rho);
#pragma unroll
        for (int cand = 0; cand < 2; cand++) {
          R xc = min_(max_(xid, -P.wheel_hl), P.wheel_hl), rq = P.wheel_r;
          if (cand == 1) { xc = xid >= 0 ? P.wheel_hl : -P.wheel_hl; rq = min_(rho, P.wheel_r); }
          R q[3] = {xc, rq * d[1] * ir, rq * d[2] * ir};  // relative to the wheel centre
          R rel[3] = {q[0] - d[0], q[1] - d[1], q[2] - d[2]}, p[3], nb[3];
          mulT_(RTB, rel, p);
          R dist = point_box(p, s, s, s, nb);
          const bool take = dist < best;  // (selects: see the vertex loop)
          best = take ? dist : best;
          found = found | take;
          R nw[3];
          mul_(RTB, nb, nw);
#pragma unroll
          for (int j = 0; j < 3; j++) {
            bn[j] = take ? -nw[j] : bn[j];
            bpos[j] = take ? q[j] + wp[j] - nw[j] * dist * (R)0.5 : bpos[j];
          }
        }
      }
#if !defined(BRS_PATCH_DIST32)
      if (sizeof(R) == 4 && found && best > c.margin - wband) {  // rare: a wave skips this block
        const double d64 = wheel_block_dist_f64(P, S, wsel);
        found = d64 < P.margin_d[CC_BLOCK_ROBOT];
        best = (R)d64;
      }
#endif
      if (found) {
        R fw[9];
        world_frame(F, bn, fw);
        add_coupled(P, st, F, S, bpos, fw, best, wsel, false);
      }
    }
    BRS_TOC(11);
  }

  // per-pass data of one coupled contact: frame axes (n,t1,t2) rotated into both body frames
  struct Coupled {
    R rT[3], rB[3], dT[3][3], dB[3][3], wc[3];
    R An, Bt1, Bt2, D, mu;
    int sel;
  };
  // C persists across the points of the patch: the frame rows rotated into both body frames are computed at its first point.
  // WHEEL = false: point c of the torso patch (no wheel dof involved); WHEEL = true: the wheel contact (slot PATCH_MAX).
  template <bool WHEEL> static BRS_HD void coupled_load(const Params<R>& P, const Store<R>& st, const Frame& F, int c, Coupled& C) {
    const int slot = WHEEL ? (int)PATCH_MAX : c;
#pragma unroll
    for (int j = 0; j < 3; j++) { C.rT[j] = st.getc(slot, j); C.rB[j] = st.getc(slot, 3 + j); }
    C.An = st.getc(slot, 6); C.Bt1 = st.getc(slot, 7); C.Bt2 = st.getc(slot, 8); C.D = st.getc(slot, 9);
    C.sel = WHEEL ? sel_wheel_contact(F.sels) : 0;
    C.mu = P.cc[CC_BLOCK_ROBOT].mu;
    if (WHEEL || c == 0) {
      const int fs = WHEEL ? 1 : 0;
      R fw[9] = {st.getf(fs, 0), st.getf(fs, 1), st.getf(fs, 2), 0, 0, 0, 0, 0, 0};
      make_frame(fw);
#pragma unroll
      for (int k = 0; k < 3; k++) { mulT_(F.RT, fw + 3 * k, C.dT[k]); mulT_(F.RB, fw + 3 * k, C.dB[k]); }
    }
    if constexpr (WHEEL) wheel_col(P, C.sel, C.rT, C.wc);
    else { C.wc[0] = 0; C.wc[1] = 0; C.wc[2] = 0; }
  }

  // ---- Newton solver of the convex acceleration problem  min 1/2 (x-a0)^T M (x-a0) + sum 1/2 D min(0, J x - aref)^2
  // over all NV dofs.  Every lane walks its own contact lists (robot<->floor, block<->floor, block<->robot), so a
  // wave pays max-over-lanes of the list lengths, not a sum over "modes".  One iteration = assemble (H and rhs of
  // the quadratic piece selected by the active rows; on the first iteration the rows are evaluated at the warm
  // start inside the same pass) -> Cholesky solve -> passA at the new point (cost, J^T f, active rows).  Exact
  // termination: a full step that reproduces its own active set is the minimiser of the piecewise-quadratic cost.
  struct Solver {
    static constexpr int NN = NV, NHH = NH;

    static BRS_HD void gauss(const Params<R>& P, const R* x, const R* a0, R* Md, R& cst) {
      R dx[NN];
#pragma unroll
      for (int i = 0; i < NN; i++) dx[i] = x[i] - a0[i];
      mmul_(P, dx, Md);
      if constexpr (BLK) {
#pragma unroll
        for (int i = 0; i < 3; i++) { Md[8 + i] = P.mB * dx[8 + i]; Md[11 + i] = P.IB * dx[11 + i]; }
      }
      cst = 0;
#pragma unroll
      for (int i = 0; i < NN; i++) cst += (R)0.5 * dx[i] * Md[i];
    }

    // rows of one contact from (cn, c1, c2): e = cn +- c1, cn +- c2; returns mask, accumulates cost; l = max(-e, 0)
    static BRS_HD int rows_(R cn, R c1, R c2, R D, R& cst, R* l, int mh, bool& sm) {
      R e1 = cn + c1, e2 = cn - c1, e3 = cn + c2, e4 = cn - c2;
      l[0] = max_(-e1, (R)0); l[1] = max_(-e2, (R)0); l[2] = max_(-e3, (R)0); l[3] = max_(-e4, (R)0);
      cst += (R)0.5 * D * (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] + l[3] * l[3]);
      int mk = (e1 < 0 ? 1 : 0) | (e2 < 0 ? 2 : 0) | (e3 < 0 ? 4 : 0) | (e4 < 0 ? 8 : 0);
      // a row that sits on its own boundary (|e| below the rounding of its terms) carries no force either way:
      // its flip does not invalidate the quadratic piece H was built for
      int df = mk ^ mh;
      if (df) {  // rare (a wave usually skips this block)
        R tol = (R)BRS_FLIP_TOL * (abs_(cn) + abs_(c1) + abs_(c2));
        if (((df & 1) && abs_(e1) > tol) || ((df & 2) && abs_(e2) > tol) || ((df & 4) && abs_(e3) > tol) || ((df & 8) && abs_(e4) > tol))
          sm = false;
      }
      return mk;
    }

    // rows of one block<->robot contact at x (and its force when FORCES): a point of the torso patch or the wheel contact
    template <bool FORCES, bool WHEEL>
    static BRS_HD void passA_coupled(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, const R* x, int c, Coupled& C, R& cst,
                                     R* l, R* fcon, bool& sm) {
      coupled_load<WHEEL>(P, st, F, c, C);
      R t[3];
      cross_(x + 3, C.rT, t);
      R paT[3] = {x[0] + t[0], x[1] + t[1], x[2] + t[2]};
      if constexpr (WHEEL) {
        const R x6 = x[6], x7 = x[7];
        const R xs = by_wheel<R>(C.sel, x6, x7);
        paT[0] += xs * C.wc[0]; paT[1] += xs * C.wc[1]; paT[2] += xs * C.wc[2];
      }
      cross_(x + 11, C.rB, t);
      R paB[3] = {x[8] + t[0], x[9] + t[1], x[10] + t[2]};
      int mk = rows_(dot_(C.dB[0], paB) - dot_(C.dT[0], paT) - C.An, C.mu * (dot_(C.dB[1], paB) - dot_(C.dT[1], paT)) - C.Bt1,
                     C.mu * (dot_(C.dB[2], paB) - dot_(C.dT[2], paT)) - C.Bt2, C.D, cst, l, get4(M.hC, c), sm);
      M.nC |= put4(mk, c);
      if constexpr (FORCES) {
        R fn = C.D * (l[0] + l[1] + l[2] + l[3]), f1 = C.D * C.mu * (l[0] - l[1]), f2 = C.D * C.mu * (l[2] - l[3]);
        R fT[3], fB[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          fT[j] = -(C.dT[0][j] * fn + C.dT[1][j] * f1 + C.dT[2][j] * f2);
          fB[j] = C.dB[0][j] * fn + C.dB[1][j] * f1 + C.dB[2][j] * f2;
        }
        cross_(C.rT, fT, t);
        fcon[0] += fT[0]; fcon[1] += fT[1]; fcon[2] += fT[2];
        fcon[3] += t[0]; fcon[4] += t[1]; fcon[5] += t[2];
        if constexpr (WHEEL) {
          R fw = dot_(C.wc, fT);
          fcon[6] += C.sel == 1 ? fw : (R)0;
          fcon[7] += C.sel == 2 ? fw : (R)0;
        }
        cross_(C.rB, fB, t);
        fcon[8] += fB[0]; fcon[9] += fB[1]; fcon[10] += fB[2];
        fcon[11] += t[0]; fcon[12] += t[1]; fcon[13] += t[2];
      }
    }

    // relative acceleration twist of the patch at x, frame coordinates, taken at the block centre (Sim::Patch)
    static BRS_HD void patch_twist(const Patch& Q, const R* x, R* Vp, R* Wp) {
      R aT[3], wT[3], aB[3], wB[3], t[3];
      mul_(Q.Fm, x, aT); mul_(Q.Fm, x + 3, wT); mul_(Q.Ph, x + 8, aB); mul_(Q.Ph, x + 11, wB);
      cross_(Q.c, wT, t);
#pragma unroll
      for (int k = 0; k < 3; k++) { Vp[k] = aB[k] - aT[k] + t[k]; Wp[k] = wB[k] - wT[k]; }
    }
    // rows of the whole block<->torso patch at x (and its wrench when FORCES), patch-frame algebra
    template <bool FORCES>
    static BRS_HD void passA_patch(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, const R* x, R& cst, R* l, R* fcon, bool& sm) {
      Patch Q;
      patch_load(st, Q);
      R Vp[3], Wp[3], Ft[3] = {0, 0, 0}, Mt[3] = {0, 0, 0};
      patch_twist(Q, x, Vp, Wp);
      const R mu = P.cc[CC_BLOCK_ROBOT].mu;
      for (int c = 0; c < F.nc; c++) {
        const R r[3] = {st.getc(c, 0), st.getc(c, 1), st.getc(c, 2)};
        const R An = st.getc(c, 3), Bt1 = st.getc(c, 4), Bt2 = st.getc(c, 5), D = st.getc(c, 6);
        R t[3];
        cross_(Wp, r, t);
        int mk = rows_(Vp[0] + t[0] - An, mu * (Vp[1] + t[1]) - Bt1, mu * (Vp[2] + t[2]) - Bt2, D, cst, l, get4(M.hC, c), sm);
        M.nC |= put4(mk, c);
        if constexpr (FORCES) {
          const R f[3] = {D * (l[0] + l[1] + l[2] + l[3]), D * mu * (l[0] - l[1]), D * mu * (l[2] - l[3])};  // frame coordinates, on the block
          cross_(r, f, t);
#pragma unroll
          for (int k = 0; k < 3; k++) { Ft[k] += f[k]; Mt[k] += t[k]; }
        }
      }
      if constexpr (FORCES) {  // wrench (at the block centre) back to the dofs: Ph^T on the block, -Fm^T on the torso (moment about ITS origin: + c' x F)
        R a[3], b[3], t[3];
        cross_(Q.c, Ft, t);
        R mt[3] = {Mt[0] + t[0], Mt[1] + t[1], Mt[2] + t[2]};
        mulT_(Q.Fm, Ft, a); mulT_(Q.Fm, mt, b);
#pragma unroll
        for (int k = 0; k < 3; k++) { fcon[k] -= a[k]; fcon[3 + k] -= b[k]; }
        mulT_(Q.Ph, Ft, a); mulT_(Q.Ph, Mt, b);
#pragma unroll
        for (int k = 0; k < 3; k++) { fcon[8 + k] += a[k]; fcon[11 + k] += b[k]; }
      }
    }

    // pass A: active-row masks at x (same = every mask equals the one H was built with) and, when FORCES, also the cost
    // and the constraint force J^T f.  The verify-only form is what the common path runs: at a point that reproduces its
    // active set the constraint force is M (x - a0) exactly, no need to accumulate it contact by contact.
    template <bool FORCES>
    static BRS_HD void passA(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, const R* x, const R* a0, R& cost,
                             R* fcon, bool& same) {
      R Md[NN], cst = 0, l[4];
      M.nR = 0; M.nB = 0; M.nC = 0;
      if constexpr (FORCES) gauss(P, x, a0, Md, cst);  // the cost only steers the damped fallback
#pragma unroll
      for (int i = 0; i < NN; i++) fcon[i] = 0;
      bool sm = true;
      // Env01 family: the record of the NEXT contact is requested from LDS before this one is evaluated -- a lone wave cannot hide
      // the ~100-cycle return behind another wave's work (+2.5 % Env01-v2, same box).  Not for Env03: the 7 live words cost it
      // 30 more SGPR spills and the gain is gone (profiles/r03_ab_experiments.json).
      constexpr bool PF = !BLK;
      R nx_[SLOT_WORDS];
      if constexpr (PF) {
#pragma unroll
        for (int w_ = 0; w_ < SLOT_WORDS; w_++) nx_[w_] = st.get(SLOT_ROBOT, w_);
      }
      for (int c = 0; c < F.nfr; c++) {
        int s = SLOT_ROBOT + c;
        R r[3], An, Bt1, Bt2, D;
        if constexpr (PF) {
          r[0] = nx_[0]; r[1] = nx_[1]; r[2] = nx_[2]; An = nx_[3]; Bt1 = nx_[4]; Bt2 = nx_[5]; D = nx_[6];
          const int sn = SLOT_ROBOT + (c + 1 < N_ROBOT_SLOTS ? c + 1 : c);  // (the slot after the last contact holds stale but finite words)
#pragma unroll
          for (int w_ = 0; w_ < SLOT_WORDS; w_++) nx_[w_] = st.get(sn, w_);
        } else {
          r[0] = st.get(s, 0); r[1] = st.get(s, 1); r[2] = st.get(s, 2);
          An = st.get(s, 3); Bt1 = st.get(s, 4); Bt2 = st.get(s, 5); D = st.get(s, 6);
        }
        int sel = sel_robot(F.sels, c);
        R mu = sel == 0 ? P.cc[CC_TORSO_FLOOR].mu : F.muW;
        R wc[3], t[3];
        wheel_col(P, sel, r, wc);
        cross_(x + 3, r, t);
        const R x6 = x[6], x7 = x[7];
        R xs = by_wheel<R>(sel, x6, x7);
        R pa[3] = {x[0] + t[0] + xs * wc[0], x[1] + t[1] + xs * wc[1], x[2] + t[2] + xs * wc[2]};
        int mk = rows_(dot_(F.nT(), pa) - An, mu * dot_(F.t1T(), pa) - Bt1, -mu * dot_(F.xT(), pa) - Bt2, D, cst, l, get4(M.hR, c), sm);
        M.nR |= put4(mk, c);
        if constexpr (FORCES) {
          R fn = D * (l[0] + l[1] + l[2] + l[3]), f1 = D * mu * (l[0] - l[1]), f2 = D * mu * (l[2] - l[3]);
          R fb[3] = {F.nT()[0] * fn + F.t1T()[0] * f1 - F.xT()[0] * f2, F.nT()[1] * fn + F.t1T()[1] * f1 - F.xT()[1] * f2,
                     F.nT()[2] * fn + F.t1T()[2] * f1 - F.xT()[2] * f2};
          cross_(r, fb, t);
          fcon[0] += fb[0]; fcon[1] += fb[1]; fcon[2] += fb[2];
          fcon[3] += t[0]; fcon[4] += t[1]; fcon[5] += t[2];
          R fw = dot_(wc, fb);
          fcon[6] += sel == 1 ? fw : (R)0;
          fcon[7] += sel == 2 ? fw : (R)0;
        }
      }
      if constexpr (BLK) {
        for (int c = 0; c < F.nfb; c++) {
          int s = SLOT_BLOCK + c;
          R r[3] = {st.get(s, 0), st.get(s, 1), st.get(s, 2)};
          R An = st.get(s, 3), Bt1 = st.get(s, 4), Bt2 = st.get(s, 5), D = st.get(s, 6);
          R mu = P.cc[CC_BLOCK_FLOOR].mu;
          R t[3];
          cross_(x + 11, r, t);
          R pa[3] = {x[8] + t[0], x[9] + t[1], x[10] + t[2]};
          int mk = rows_(dot_(F.nB(), pa) - An, mu * dot_(F.t1B(), pa) - Bt1, -mu * dot_(F.xB(), pa) - Bt2, D, cst, l, get4(M.hB, c), sm);
          M.nB |= put4(mk, c);
          if constexpr (FORCES) {
            R fn = D * (l[0] + l[1] + l[2] + l[3]), f1 = D * mu * (l[0] - l[1]), f2 = D * mu * (l[2] - l[3]);
            R fb[3] = {F.nB()[0] * fn + F.t1B()[0] * f1 - F.xB()[0] * f2, F.nB()[1] * fn + F.t1B()[1] * f1 - F.xB()[1] * f2,
                       F.nB()[2] * fn + F.t1B()[2] * f1 - F.xB()[2] * f2};
            cross_(r, fb, t);
            fcon[8] += fb[0]; fcon[9] += fb[1]; fcon[10] += fb[2];
            fcon[11] += t[0]; fcon[12] += t[1]; fcon[13] += t[2];
          }
        }
        Coupled C;
#if BRS_PATCH_FRAME
        if (F.nc > 0) passA_patch<FORCES>(P, st, F, M, x, cst, l, fcon, sm);
#else
        for (int c = 0; c < F.nc; c++) passA_coupled<FORCES, false>(P, st, F, M, x, c, C, cst, l, fcon, sm);
#endif
        if (sel_wheel_contact(F.sels)) passA_coupled<FORCES, true>(P, st, F, M, x, PATCH_MAX, C, cst, l, fcon, sm);
      }
      cost = cst;
      same = sm;
    }

    static constexpr int NP = (NN + 1) / 2, NH2 = hp_count(NN);

    // One contact into H and rhs through its 3x3 weight matrix in the contact frame:
    //   rows j_k = G_n +- mu G_t ;  sum_k act_k D j_k j_k^T = G^T W G ,  W = D [[sum a, mu(a0-a1), mu(a2-a3)], [., mu^2(a0+a1), 0], [., 0, mu^2(a2+a3)]]
    // G rows arrive as pairs over the dof range [2*P0, 2*(P0+NPA)); `eval`: decide the active rows at x (else from mnew).
    // SKIP: a pair of the range whose G entries are zero (the wheel dofs for a point of the torso patch): left out everywhere
    template <int P0, int NPA, int SKIP = -1>
    static BRS_HD int contact_into(V2<R>* H, V2<R>* rhs2, const V2<R>* gn, const V2<R>* g1, const V2<R>* g2, R mu, R D, R An,
                                   R Bt1, R Bt2, bool eval, int mnew, const V2<R>* x2) {
      int mk = mnew;
      if (eval) {
        V2<R> an = v2_splat<R>((R)0), a1 = an, a2 = an;
#pragma unroll
        for (int k = 0; k < NPA; k++) {
          if (k == SKIP) continue;
          an = v2_fma(gn[k], x2[P0 + k], an); a1 = v2_fma(g1[k], x2[P0 + k], a1); a2 = v2_fma(g2[k], x2[P0 + k], a2);
        }
        R cn = (an.x + an.y) - An, c1 = mu * (a1.x + a1.y) - Bt1, c2 = mu * (a2.x + a2.y) - Bt2;
        mk = (cn + c1 < 0 ? 1 : 0) | (cn - c1 < 0 ? 2 : 0) | (cn + c2 < 0 ? 4 : 0) | (cn - c2 < 0 ? 8 : 0);
      }
      R b0 = (mk & 1) ? (R)1 : (R)0, b1 = (mk & 2) ? (R)1 : (R)0, b2 = (mk & 4) ? (R)1 : (R)0, b3 = (mk & 8) ? (R)1 : (R)0;
      R Dm = D * mu, Dmm = Dm * mu;
      R Wnn = D * (b0 + b1 + b2 + b3), Wn1 = Dm * (b0 - b1), Wn2 = Dm * (b2 - b3), W11 = Dmm * (b0 + b1), W22 = Dmm * (b2 + b3);
      // rhs += sum_k act_k D aref_k j_k with aref = An +- Bt1, An +- Bt2
      R rn = Wnn * An + D * ((b0 - b1) * Bt1 + (b2 - b3) * Bt2);
      R r1 = Wn1 * An + Dm * (b0 + b1) * Bt1, r2 = Wn2 * An + Dm * (b2 + b3) * Bt2;
      V2<R> tn[NPA], t1[NPA], t2[NPA];
      const V2<R> sWnn = v2_splat(Wnn), sWn1 = v2_splat(Wn1), sWn2 = v2_splat(Wn2), sW11 = v2_splat(W11), sW22 = v2_splat(W22);
      const V2<R> srn = v2_splat(rn), sr1 = v2_splat(r1), sr2 = v2_splat(r2);
#pragma unroll
      for (int k = 0; k < NPA; k++) {
        if (k == SKIP) continue;
        tn[k] = v2_fma(sWnn, gn[k], v2_fma(sWn1, g1[k], v2_mul(sWn2, g2[k])));
        t1[k] = v2_fma(sWn1, gn[k], v2_mul(sW11, g1[k]));
        t2[k] = v2_fma(sWn2, gn[k], v2_mul(sW22, g2[k]));
        rhs2[P0 + k] = v2_fma(srn, gn[k], v2_fma(sr1, g1[k], v2_fma(sr2, g2[k], rhs2[P0 + k])));
      }
      // three sweeps over the touched part of H, one frame axis each: a v_pk_fma_f32 feeding the next v_pk_fma_f32 costs a
      // wait state (s_nop) on gfx950; chained per entry the compiler left 166 of them in the loop, swept per axis the
      // dependent pair is a whole sweep apart
#pragma unroll
      for (int axis = 0; axis < 3; axis++) {
        const V2<R>* gg = axis == 0 ? g2 : (axis == 1 ? g1 : gn);
        const V2<R>* tt = axis == 0 ? t2 : (axis == 1 ? t1 : tn);
#pragma unroll
        for (int a = 0; a < 2 * NPA; a++) {
          if (a / 2 == SKIP) continue;
          const V2<R> sg = v2_splat((a & 1) ? gg[a / 2].y : gg[a / 2].x);
#pragma unroll
          for (int k = 0; k <= a / 2; k++) {
            if (k == SKIP) continue;
            H[hp(2 * P0 + a, P0 + k)] = v2_fma(sg, tt[k], H[hp(2 * P0 + a, P0 + k)]);
          }
        }
      }
      return mk;
    }

    // one block<->robot contact into H / rhs.  A point of the torso patch does not see the wheel dofs: pair 3 of its G rows is
    // zero and 14 of the 56 pair updates fall away; the wheel contact (its own slot, met once, outside the patch loop) takes
    // the full form -- no per-iteration divergence between the two
    template <bool WHEEL>
    static BRS_HD void assemble_coupled(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, bool first, uint32_t srcC, int c,
                                        Coupled& C, V2<R>* H, V2<R>* rhs2, const V2<R>* x2) {
      coupled_load<WHEEL>(P, st, F, c, C);
      V2<R> g[3][7];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        R ct[3], cb[3];
        cross_(C.rT, C.dT[k], ct);
        cross_(C.rB, C.dB[k], cb);
        g[k][0] = v2_make(-C.dT[k][0], -C.dT[k][1]); g[k][1] = v2_make(-C.dT[k][2], -ct[0]); g[k][2] = v2_make(-ct[1], -ct[2]);
        if constexpr (WHEEL) {
          const R wk = dot_(C.wc, C.dT[k]);
          g[k][3] = v2_make(C.sel == 1 ? -wk : (R)0, C.sel == 2 ? -wk : (R)0);
        } else
          g[k][3] = v2_splat<R>((R)0);
        g[k][4] = v2_make(C.dB[k][0], C.dB[k][1]); g[k][5] = v2_make(C.dB[k][2], cb[0]); g[k][6] = v2_make(cb[1], cb[2]);
      }
      int mk;
      if constexpr (WHEEL) {
        const bool ev = first && !(BRS_MASK_HINT && sel_wheel_contact(F.psels) == C.sel);
        mk = contact_into<0, 7>(H, rhs2, g[0], g[1], g[2], C.mu, C.D, C.An, C.Bt1, C.Bt2, ev, get4(srcC, c), x2);
      } else {
        const bool ev = first && !(BRS_MASK_HINT && c < F.pnc);
        mk = contact_into<0, 7, 3>(H, rhs2, g[0], g[1], g[2], C.mu, C.D, C.An, C.Bt1, C.Bt2, ev, get4(srcC, c), x2);
      }
      M.hC |= put4(mk, c);
    }

    // the whole block<->torso patch into H / rhs: Z = sum_c K_c^T W_c K_c (6x6, twist space, frame coordinates) and
    // z = sum_c K_c^T rho_c per point (K at rho = the point seen from the block centre), then ONE congruence per dof block:
    // block dofs J = K diag(Ph), torso dofs J = -K S' diag(Fm) with the shift S' = [[I, -[c']x], [0, I]] from the torso origin to
    // the block centre
    static BRS_HD void assemble_patch(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, bool first, uint32_t srcC,
                                      V2<R>* H, V2<R>* rhs2, const R* x) {
      Patch Q;
      patch_load(st, Q);
      R Vp[3] = {0, 0, 0}, Wp[3] = {0, 0, 0};
      if (first) patch_twist(Q, x, Vp, Wp);  // rows of contacts without a hint are evaluated at the warm start
      const R mu = P.cc[CC_BLOCK_ROBOT].mu;
      R Za = 0, Zp = 0, Zq = 0, Zb = 0, Zc = 0;  // Zvv = [[a, p, q], [p, b, 0], [q, 0, c]] (the sum of the points' 3x3 weights)
      R Zvw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Zww[6] = {0, 0, 0, 0, 0, 0}, zv[3] = {0, 0, 0}, zw[3] = {0, 0, 0};  // Zww: 00 10 11 20 21 22
      for (int c = 0; c < F.nc; c++) {
        const R rx = st.getc(c, 0), ry = st.getc(c, 1), rz = st.getc(c, 2);
        const R An = st.getc(c, 3), Bt1 = st.getc(c, 4), Bt2 = st.getc(c, 5), D = st.getc(c, 6);
        int mk = get4(srcC, c);
        if (first && !(BRS_MASK_HINT && c < F.pnc)) {
          const R cn = Vp[0] + (Wp[1] * rz - Wp[2] * ry) - An, c1 = mu * (Vp[1] + (Wp[2] * rx - Wp[0] * rz)) - Bt1,
                  c2 = mu * (Vp[2] + (Wp[0] * ry - Wp[1] * rx)) - Bt2;
          mk = (cn + c1 < 0 ? 1 : 0) | (cn - c1 < 0 ? 2 : 0) | (cn + c2 < 0 ? 4 : 0) | (cn - c2 < 0 ? 8 : 0);
        }
        M.hC |= put4(mk, c);
        const R b0 = (mk & 1) ? (R)1 : (R)0, b1 = (mk & 2) ? (R)1 : (R)0, b2 = (mk & 4) ? (R)1 : (R)0, b3 = (mk & 8) ? (R)1 : (R)0;
        const R Dm = D * mu, Dmm = Dm * mu;
        const R a = D * (b0 + b1 + b2 + b3), p = Dm * (b0 - b1), q = Dm * (b2 - b3), b = Dmm * (b0 + b1), cc = Dmm * (b2 + b3);
        // rho = sum_k act_k D aref_k (row weights in frame coordinates), as in contact_into
        const R rn = a * An + D * ((b0 - b1) * Bt1 + (b2 - b3) * Bt2), r1 = p * An + Dm * (b0 + b1) * Bt1, r2 = q * An + Dm * (b2 + b3) * Bt2;
        Za += a; Zp += p; Zq += q; Zb += b; Zc += cc;
        // U = W [r]x, columns W (0, z, -y), W (-z, 0, x), W (y, -x, 0)
        const R u00 = p * rz - q * ry, u10 = b * rz, u20 = -cc * ry;
        const R u01 = q * rx - a * rz, u11 = -p * rz, u21 = cc * rx - q * rz;
        const R u02 = a * ry - p * rx, u12 = p * ry - b * rx, u22 = q * ry;
        // K^T W K = [[W, -U], [-U^T, -[r]x U]]
        Zvw[0] -= u00; Zvw[1] -= u01; Zvw[2] -= u02; Zvw[3] -= u10; Zvw[4] -= u11; Zvw[5] -= u12; Zvw[6] -= u20; Zvw[7] -= u21; Zvw[8] -= u22;
        // ([r]x U)[i][j] = (r x U[:, j])_i ; lower triangle only
        Zww[0] -= ry * u20 - rz * u10;
        Zww[1] -= rz * u00 - rx * u20; Zww[2] -= rz * u01 - rx * u21;
        Zww[3] -= rx * u10 - ry * u00; Zww[4] -= rx * u11 - ry * u01; Zww[5] -= rx * u12 - ry * u02;
        zv[0] += rn; zv[1] += r1; zv[2] += r2;
        zw[0] += ry * r2 - rz * r1; zw[1] += rz * rn - rx * r2; zw[2] += rx * r1 - ry * rn;
      }
      // 3x3 blocks as full matrices
      const R Zvv[9] = {Za, Zp, Zq, Zp, Zb, 0, Zq, 0, Zc};
      const R Zw[9] = {Zww[0], Zww[1], Zww[3], Zww[1], Zww[2], Zww[4], Zww[3], Zww[4], Zww[5]};
      // the torso sees the twist through S' = [[I, -[c]x], [0, I]] (tau = -S' diag(Fm) x_T + diag(Ph) x_B):
      // Y10 = [c]x Zvv + Zvw^T, Y11 = [c]x Zvw + Zww  (rows w of S'^T Z), B11 = -Y10 [c]x + Y11 = (S'^T Z S')_ww (symmetric)
      const R cx = Q.c[0], cy = Q.c[1], cz = Q.c[2];
      R Y10[9], Y11[9], B11[9];
#pragma unroll
      for (int j = 0; j < 3; j++) {  // column j of [c]x X = c x X[:, j]
        const R v0 = Zvv[j], v1 = Zvv[3 + j], v2 = Zvv[6 + j];
        Y10[j] = (cy * v2 - cz * v1) + Zvw[3 * j]; Y10[3 + j] = (cz * v0 - cx * v2) + Zvw[3 * j + 1]; Y10[6 + j] = (cx * v1 - cy * v0) + Zvw[3 * j + 2];
        const R w0 = Zvw[j], w1 = Zvw[3 + j], w2 = Zvw[6 + j];
        Y11[j] = (cy * w2 - cz * w1) + Zw[j]; Y11[3 + j] = (cz * w0 - cx * w2) + Zw[3 + j]; Y11[6 + j] = (cx * w1 - cy * w0) + Zw[6 + j];
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {  // (X [c]x)[i][:] = X[i, :] x c
        const R y0 = Y10[3 * i], y1 = Y10[3 * i + 1], y2 = Y10[3 * i + 2];
        // [c]x = [[0, -cz, cy], [cz, 0, -cx], [-cy, cx, 0]]
        B11[3 * i] = -(y1 * cz - y2 * cy) + Y11[3 * i]; B11[3 * i + 1] = -(y2 * cx - y0 * cz) + Y11[3 * i + 1]; B11[3 * i + 2] = -(y0 * cy - y1 * cx) + Y11[3 * i + 2];
      }
      // congruences A^T X B into the lower triangle of H (rows ra.., columns ca..): out[i][j] = sum_{m,n} A[m][i] X[m][n] B[n][j]
      auto cong = [&](const R* A, const R* X, const R* B, R sgn, auto put) {
        R XB[9];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
          for (int j = 0; j < 3; j++) XB[3 * m + j] = X[3 * m] * B[j] + X[3 * m + 1] * B[3 + j] + X[3 * m + 2] * B[6 + j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) put(i, j, sgn * (A[i] * XB[j] + A[3 + i] * XB[3 + j] + A[6 + i] * XB[6 + j]));
      };
      R ZvwT[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) ZvwT[3 * i + j] = Zvw[3 * j + i];
#define BRS_HADD(ra, ca) [&](int i, int j, R v) { if ((ra) + i >= (ca) + j) hadd(H, (ra) + i, (ca) + j, v); }
      R Y10T[9], Y11T[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Y10T[3 * i + j] = Y10[3 * j + i]; Y11T[3 * i + j] = Y11[3 * j + i]; }
      cong(Q.Fm, Zvv, Q.Fm, (R)1, BRS_HADD(0, 0));     // torso lin-lin
      cong(Q.Fm, Y10, Q.Fm, (R)1, BRS_HADD(3, 0));     // torso ang-lin
      cong(Q.Fm, B11, Q.Fm, (R)1, BRS_HADD(3, 3));     // torso ang-ang
      cong(Q.Ph, Zvv, Q.Ph, (R)1, BRS_HADD(8, 8));     // block lin-lin
      cong(Q.Ph, ZvwT, Q.Ph, (R)1, BRS_HADD(11, 8));   // block ang-lin  (Zwv = Zvw^T)
      cong(Q.Ph, Zw, Q.Ph, (R)1, BRS_HADD(11, 11));    // block ang-ang
      cong(Q.Ph, Zvv, Q.Fm, (R)-1, BRS_HADD(8, 0));    // block lin x torso lin    (T_T = -S' diag(Fm): minus; columns of Z S')
      cong(Q.Ph, Y10T, Q.Fm, (R)-1, BRS_HADD(8, 3));   // block lin x torso ang    ((Z S')_vw = Y10^T)
      cong(Q.Ph, ZvwT, Q.Fm, (R)-1, BRS_HADD(11, 0));  // block ang x torso lin
      cong(Q.Ph, Y11T, Q.Fm, (R)-1, BRS_HADD(11, 3));  // block ang x torso ang    ((Z S')_ww = Y11^T)
#undef BRS_HADD
      // rhs += J^T rho: torso -diag(Fm^T) S'^T z with S'^T z = (zv, zw + c x zv), block diag(Ph^T) z
      R a3[3], b3[3], t[3];
      cross_(Q.c, zv, t);
      R zt[3] = {zw[0] + t[0], zw[1] + t[1], zw[2] + t[2]};
      mulT_(Q.Fm, zv, a3); mulT_(Q.Fm, zt, b3);
#pragma unroll
      for (int k = 0; k < 3; k++) { radd(rhs2, k, -a3[k]); radd(rhs2, 3 + k, -b3[k]); }
      mulT_(Q.Ph, zv, a3); mulT_(Q.Ph, zw, b3);
#pragma unroll
      for (int k = 0; k < 3; k++) { radd(rhs2, 8 + k, a3[k]); radd(rhs2, 11 + k, b3[k]); }
    }

    static BRS_HD void hadd(V2<R>* H, int a, int b, R v) {
      if (b & 1) H[hp(a, b / 2)].y += v; else H[hp(a, b / 2)].x += v;
    }
    static BRS_HD void radd(V2<R>* rhs2, int i, R v) {
      if (i & 1) rhs2[i / 2].y += v; else rhs2[i / 2].x += v;
    }

    // assemble H = M + sum_active D j j^T and rhs = M a0 + sum_active D aref j  (both on the packed-pair layout)
    static BRS_HD void assemble(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, const R* x, const R* a0, bool first,
                                V2<R>* H, V2<R>* rhs2) {
      // active rows of this piece: the latest evaluated masks; on the first iteration the previous substep's final
      // masks where the slot held a contact of the same body then (else the rows are evaluated at x)
      const uint32_t srcR = first ? F.pmR : M.nR, srcB = first ? F.pmB : M.nB, srcC = first ? F.pmC : M.nC;
      M.hR = 0; M.hB = 0; M.hC = 0;
#pragma unroll
      for (int i = 0; i < NH2; i++) H[i] = v2_splat<R>((R)0);
      h2_set(H, 0, 0, P.m); h2_set(H, 1, 1, P.m); h2_set(H, 2, 2, P.m);
      h2_set(H, 3, 3, P.Ixx); h2_set(H, 4, 4, P.Iyy); h2_set(H, 5, 5, P.Izz); h2_set(H, 6, 6, P.Ia); h2_set(H, 7, 7, P.Ia);
      h2_set(H, 4, 0, P.mcz); h2_set(H, 3, 1, -P.mcz); h2_set(H, 6, 3, -P.Ia); h2_set(H, 7, 3, P.Ia);
      R rhs[NN];
      mmul_(P, a0, rhs);
      if constexpr (BLK) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          h2_set(H, 8 + i, 8 + i, P.mB); h2_set(H, 11 + i, 11 + i, P.IB);
          rhs[8 + i] = P.mB * a0[8 + i]; rhs[11 + i] = P.IB * a0[11 + i];
        }
      }
      V2<R> x2[NP];
#pragma unroll
      for (int k = 0; k < NP; k++) { x2[k] = v2_make(x[2 * k], x[2 * k + 1]); rhs2[k] = v2_make(rhs[2 * k], rhs[2 * k + 1]); }
      BRS_MARK("asm_robot_loop");
      for (int c = 0; c < F.nfr; c++) {
        int s = SLOT_ROBOT + c;
        R r[3] = {st.get(s, 0), st.get(s, 1), st.get(s, 2)};
        R An = st.get(s, 3), Bt1 = st.get(s, 4), Bt2 = st.get(s, 5), D = st.get(s, 6);
        int sel = sel_robot(F.sels, c);
        const bool ev = first && !(BRS_MASK_HINT && c < F.pnfr && sel_robot(F.psels, c) == sel);
        R mu = sel == 0 ? P.cc[CC_TORSO_FLOOR].mu : F.muW;
        R wc[3], rn[3], r1[3], r2[3];
        wheel_col(P, sel, r, wc);
        cross_(r, F.nT(), rn); cross_(r, F.t1T(), r1); cross_(F.xT(), r, r2);  // r x t2 = x_row x r
        R wn = dot_(wc, F.nT()), w1 = dot_(wc, F.t1T()), w2 = -dot_(wc, F.xT());
        R zL = sel == 1 ? (R)1 : (R)0, zR = sel == 2 ? (R)1 : (R)0;
        V2<R> gn[4] = {v2_make(F.nT()[0], F.nT()[1]), v2_make(F.nT()[2], rn[0]), v2_make(rn[1], rn[2]), v2_make(zL * wn, zR * wn)};
        V2<R> g1[4] = {v2_make(F.t1T()[0], F.t1T()[1]), v2_make(F.t1T()[2], r1[0]), v2_make(r1[1], r1[2]), v2_make(zL * w1, zR * w1)};
        V2<R> g2[4] = {v2_make(-F.xT()[0], -F.xT()[1]), v2_make(-F.xT()[2], r2[0]), v2_make(r2[1], r2[2]), v2_make(zL * w2, zR * w2)};
        int mk = contact_into<0, 4>(H, rhs2, gn, g1, g2, mu, D, An, Bt1, Bt2, ev, get4(srcR, c), x2);
        M.hR |= put4(mk, c);
      }
      if constexpr (BLK) {
        BRS_MARK("asm_block_loop");
        for (int c = 0; c < F.nfb; c++) {
          int s = SLOT_BLOCK + c;
          R r[3] = {st.get(s, 0), st.get(s, 1), st.get(s, 2)};
          R An = st.get(s, 3), Bt1 = st.get(s, 4), Bt2 = st.get(s, 5), D = st.get(s, 6);
          const bool ev = first && !(BRS_MASK_HINT && c < F.pnfb);
          R rn[3], r1[3], r2[3];
          cross_(r, F.nB(), rn); cross_(r, F.t1B(), r1); cross_(F.xB(), r, r2);
          V2<R> gn[3] = {v2_make(F.nB()[0], F.nB()[1]), v2_make(F.nB()[2], rn[0]), v2_make(rn[1], rn[2])};
          V2<R> g1[3] = {v2_make(F.t1B()[0], F.t1B()[1]), v2_make(F.t1B()[2], r1[0]), v2_make(r1[1], r1[2])};
          V2<R> g2[3] = {v2_make(-F.xB()[0], -F.xB()[1]), v2_make(-F.xB()[2], r2[0]), v2_make(r2[1], r2[2])};
          int mk = contact_into<4, 3>(H, rhs2, gn, g1, g2, P.cc[CC_BLOCK_FLOOR].mu, D, An, Bt1, Bt2, ev, get4(srcB, c), x2);
          M.hB |= put4(mk, c);
        }
        Coupled C;
        BRS_MARK("asm_patch_loop");
#if BRS_PATCH_FRAME
        if (F.nc > 0) assemble_patch(P, st, F, M, first, srcC, H, rhs2, x);
#else
        for (int c = 0; c < F.nc; c++) assemble_coupled<false>(P, st, F, M, first, srcC, c, C, H, rhs2, x2);
#endif
        BRS_MARK("asm_wheel_contact");
        if (sel_wheel_contact(F.sels)) assemble_coupled<true>(P, st, F, M, first, srcC, PATCH_MAX, C, H, rhs2, x2);
        BRS_MARK("asm_done");
      }
    }

    // ONE Newton iteration at x (in/out); fcon out = J^T f at the new x.  Returns true when the new point is the
    // minimiser (a full step that reproduced its own active set) or the iteration cap is reached.
    // State carried by the caller across iterations of one substep: first (true on entry), it (0), cost.
    static BRS_HD bool iterate(const Params<R>& P, Store<R>& st, const Frame& F, Masks& M, R* x, const R* a0, R* fcon, bool& first,
                               int& it, R& cost) {
      V2<R> H[NH2], rhs2[NP];
      R rhs[NN], xn[NN], ft[NN], ct;
      bool same;
      BRS_STAT(stats().iters[0]++; stats().last_iters[0]++);
      BRS_MARK("iter_assemble");
      BRS_TIC(4);
      assemble(P, st, F, M, x, a0, first, H, rhs2);
      first = false;
      for (int i_ = 0; i_ < NP; i_++) { BRS_PIN(rhs2[i_].x); BRS_PIN(rhs2[i_].y); }
      for (int i_ = 0; i_ < NH2; i_++) { BRS_PIN(H[i_].x); BRS_PIN(H[i_].y); }
      BRS_TOC(4);
      BRS_TIC(5);
      BRS_MARK("iter_chol");
#pragma unroll
      for (int k = 0; k < NP; k++) { rhs[2 * k] = rhs2[k].x; if (2 * k + 1 < NN) rhs[2 * k + 1] = rhs2[k].y; }
#pragma unroll
      for (int i = 0; i < NN; i++) xn[i] = 0;
      chol_solve_packed<R, NN>(H, rhs, xn);
      for (int i_ = 0; i_ < NN; i_++) BRS_PIN(xn[i_]);
      BRS_TOC(5);
      BRS_TIC(6);
      BRS_MARK("iter_passA");
      const bool lite = it + 1 < BRS_UNDAMPED_ITERS;  // the damped fallback needs costs and forces at every point
      if (lite) {
        passA<false>(P, st, F, M, xn, a0, ct, ft, same);
        gauss(P, xn, a0, ft, ct);  // ft = M (xn - a0): the constraint force if xn reproduces its active set
      } else
        passA<true>(P, st, F, M, xn, a0, ct, ft, same);
      BRS_TOC(6);
      BRS_STAT(if (!same) { int nf_ = __builtin_popcount(M.nR ^ M.hR) + __builtin_popcount(M.nB ^ M.hB) + __builtin_popcount(M.nC ^ M.hC);
                            stats().flip_hist[it == 0 ? 0 : 1][nf_ > 8 ? 8 : nf_]++; if (it == 0) stats().first_single = nf_ == 1; });
      BRS_MARK("iter_tail");
      bool full = true;
      // pure active-set iteration for the first sweeps (it terminates at once in ~97% of the substeps); if it has not
      // settled by then, fall back to cost-monotone damping, which cannot cycle
      for (int bt = 0; it >= BRS_UNDAMPED_ITERS && bt < 6 && ct > cost + (R)1e-5 * abs_(cost) + (R)1e-12; bt++) {
        full = false;
        BRS_STAT(stats().backtracks[0]++);
#pragma unroll
        for (int i = 0; i < NN; i++) xn[i] = x[i] + (R)0.5 * (xn[i] - x[i]);
        passA<true>(P, st, F, M, xn, a0, ct, ft, same);
      }
      cost = ct;
#pragma unroll
      for (int i = 0; i < NN; i++) { x[i] = xn[i]; fcon[i] = ft[i]; }
      it++;
      return (same && full) || it >= 16;
    }
  };

  // ---- the substep, in three pieces so that the caller can FLATTEN the substep loop and the Newton loop into one
  // per-lane state machine: sub_begin (kinematics, smooth forces, collision) -> sub_iter until converged -> sub_end
  // (implicitfast + advance).  In a wave, a lane that converged starts its next substep while its neighbours are
  // still iterating: the wave needs ~nsub * mean(iterations) trips instead of nsub * max-over-lanes(iterations).
  struct SubCtx {
    Frame F;
    R f[8], fcon[NV], cost;
    bool clL, clR, conv, first;
    int it;
    Masks M;
  };
  using CT = CtrlT<R>;
  static BRS_HD void sub_begin(const Params<R>& P, Store<R>& st, ES& S, CT ctrlL, CT ctrlR, SubCtx& C) {
    Frame& F = C.F;
    R* f = C.f;
    BRS_MARK("begin_kin");
    BRS_TIC(0);
#if BRS_LAZY_VEL32
    S.derive_vel32();
#endif
    // kinematics
    R qf[4] = {(R)S.q[0], (R)S.q[1], (R)S.q[2], (R)S.q[3]};
    quat2mat_(qf, F.RT);
    R u[3];
    mulT_(F.RT, S.v, u);
    R zT = (R)(S.p[2] - P.floor_z_d);
    // smooth forces in body coordinates
    R wx = S.w[0], wy = S.w[1], wz = S.w[2];
    R gb[3] = {-P.g * F.nT()[0], -P.g * F.nT()[1], -P.g * F.nT()[2]};
    f[0] = -P.mcz * wx * wz + P.m * gb[0];
    f[1] = -P.mcz * wy * wz + P.m * gb[1];
    f[2] = P.mcz * (wx * wx + wy * wy) + P.m * gb[2];
    R Lx = P.Ixx * wx + P.Ia * (S.ww[1] - S.ww[0]), Ly = P.Iyy * wy, Lz = P.Izz * wz;
    f[3] = -(wy * Lz - wz * Ly) - P.mcz * gb[1];
    f[4] = -(wz * Lx - wx * Lz) + P.mcz * gb[0];
    f[5] = -(wx * Ly - wy * Lx);
    // velocity servos (envs/robot-02.xml:22-25): ctrl clamp, force clamp; derivative dropped when clamped
#if BRS_VEL64
    // target minus wheel rate in fp64: the two are close (ctrl = rate + 4 a), their fp32 difference would carry ~1e-5 of the force,
    // and whether the force clamp binds -- which decides if the servo's derivative enters implicitfast -- is a discrete
    // outcome: a wheel gains 0.26 rad/s per substep clamped and 0.10 unclamped
    const double uLd = min_(max_(ctrlL, -P.ctrlrange_d), P.ctrlrange_d), uRd = min_(max_(ctrlR, -P.ctrlrange_d), P.ctrlrange_d);
    const double fLd = P.kv_d * (uLd - S.wwd[0]), fRd = P.kv_d * (uRd - S.wwd[1]);
    C.clL = fLd >= P.forcerange_d || fLd <= -P.forcerange_d;
    C.clR = fRd >= P.forcerange_d || fRd <= -P.forcerange_d;
    R fL = (R)min_(max_(fLd, -P.forcerange_d), P.forcerange_d), fR = (R)min_(max_(fRd, -P.forcerange_d), P.forcerange_d);
#else
    R uL = min_(max_(ctrlL, -P.ctrlrange), P.ctrlrange), uR = min_(max_(ctrlR, -P.ctrlrange), P.ctrlrange);
    R fL = P.kv * (uL - S.ww[0]), fR = P.kv * (uR - S.ww[1]);
    C.clL = fL >= P.forcerange || fL <= -P.forcerange;
    C.clR = fR >= P.forcerange || fR <= -P.forcerange;
    fL = min_(max_(fL, -P.forcerange), P.forcerange);
    fR = min_(max_(fR, -P.forcerange), P.forcerange);
#endif
    f[6] = fL - P.damping * S.ww[0];
    f[7] = fR - P.damping * S.ww[1];
    msolve0_(P, f, F.a0);
    for (int i_ = 0; i_ < 8; i_++) { BRS_PIN(F.a0[i_]); BRS_PIN(f[i_]); }
    BRS_PIN(u[0]); BRS_PIN(u[1]); BRS_PIN(u[2]); BRS_PIN(zT);
    BRS_TOC(0);
    F.nfr = 0; F.nfb = 0; F.nc = 0;
    F.pnfr = S.pnfr; F.pnfb = S.pnfb; F.pnc = S.pnc;
    F.sels = 0; F.psels = S.psels; F.pmR = S.pmR; F.pmB = S.pmB; F.pmC = S.pmC;
    F.muW = P.per_env_mu ? S.muw : P.cc[CC_WHEEL_FLOOR].mu;
    F.cDW = P.per_env_mu ? 2 * S.muw * S.muw * (1 + S.muw * S.muw) * P.tran_wheel : P.cc[CC_WHEEL_FLOOR].cD;
    R uB[3] = {0, 0, 0}, zB = 0;
    if constexpr (BLK) {
      R qb[4] = {(R)S.bq[0], (R)S.bq[1], (R)S.bq[2], (R)S.bq[3]};
      quat2mat_(qb, F.RB);
      mulT_(F.RB, S.bv, uB);
      zB = (R)(S.bp[2] - P.floor_z_d);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        F.dTB[i] = (R)(S.p[i] - S.bp[i]);
        F.a0[8 + i] = -P.g * F.nB()[i];
        F.a0[11 + i] = 0;
      }
      // block <-> robot FIRST: its clipping parks candidates in the (still unused) robot<->floor slot region of the lane
      BRS_TIC(3);
      BRS_MARK("begin_collide_coupled");
#ifndef BRS_NO_COUPLED
      collide_coupled(P, st, F, S);
#endif
      BRS_TOC(3);
    }
    BRS_TIC(1);
    BRS_MARK("begin_collide_robot");
    // collision: robot <-> floor.  Slot priority: wheel main points, torso corners, wheel triangle points
    const Floor64 G = floor64(P, S);
    collide_wheel(P, st, F, u, S.w, S.ww, zT, 1, false, G);
    collide_wheel(P, st, F, u, S.w, S.ww, zT, 2, false, G);
    collide_torso(P, st, F, u, S.w, S.ww, zT, G);
    collide_wheel(P, st, F, u, S.w, S.ww, zT, 1, true, G);
    collide_wheel(P, st, F, u, S.w, S.ww, zT, 2, true, G);
    BRS_TOC(1);
    if constexpr (BLK) {
      BRS_TIC(2);
      BRS_MARK("begin_collide_blockfloor");
#ifndef BRS_NO_BLOCKFLOOR
      collide_block_floor(P, st, F, S, uB, S.bw, zB);
#endif
      BRS_TOC(2);
    }
    BRS_MARK("begin_tail");
    C.first = true; C.it = 0; C.cost = 0;
    C.M.hR = 0; C.M.hB = 0; C.M.hC = 0; C.M.nR = 0; C.M.nB = 0; C.M.nC = 0;
    C.conv = F.nfr + F.nfb + F.nc == 0 && sel_wheel_contact(F.sels) == 0;
    if (C.conv) {  // no contacts: the unconstrained acceleration is the answer
#pragma unroll
      for (int i = 0; i < NV; i++) { S.a[i] = F.a0[i]; C.fcon[i] = 0; }
    }
    BRS_STAT(stats().last_iters[0] = 0; stats().first_single = 0; if (!C.conv) stats().solves[0]++);
    BRS_STAT(stats().nfr_hist[F.nfr]++; stats().nfb_hist[F.nfb]++; stats().nc_hist[F.nc]++);
  }
  static BRS_HD void sub_iter(const Params<R>& P, Store<R>& st, ES& S, SubCtx& C) {
    C.conv = Solver::iterate(P, st, C.F, C.M, S.a, C.F.a0, C.fcon, C.first, C.it, C.cost);
  }
  static BRS_HD void sub_end(const Params<R>& P, ES& S, SubCtx& C) {
    BRS_MARK("end_integrate");
    BRS_TIC(7);
    const Frame& F = C.F;
    const R* f = C.f;
    const R* fcon = C.fcon;
    const bool clL = C.clL, clR = C.clR;
    // implicitfast: (M + h*diag(damping + kv[unclamped])) acc = smooth + constraint
    R rhs[8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rhs[i] = f[i] + fcon[i];
    R dL = P.h * (P.damping + (clL ? (R)0 : P.kv)), dR = P.h * (P.damping + (clR ? (R)0 : P.kv));
    msolve_(P, rhs, dL, dR, acc);
    // advance: velocities first, then positions with the NEW velocities
    R aw[3];
    mul_(F.RT, acc, aw);
#if BRS_VEL64
#pragma unroll
    for (int i = 0; i < 3; i++) {
      S.vd[i] += P.h_d * (double)aw[i]; S.wd[i] += P.h_d * (double)acc[3 + i];
#if !BRS_LAZY_VEL32
      S.v[i] = (R)S.vd[i]; S.w[i] = (R)S.wd[i];
#endif
    }
    S.wwd[0] += P.h_d * (double)acc[6]; S.wwd[1] += P.h_d * (double)acc[7];
#if !BRS_LAZY_VEL32
    S.ww[0] = (R)S.wwd[0]; S.ww[1] = (R)S.wwd[1];
#endif
#pragma unroll
    for (int i = 0; i < 3; i++) S.p[i] += P.h_d * S.vd[i];
    quat_advance(S.q, S.wd[0], S.wd[1], S.wd[2], P.h_d);
    S.th[0] += P.h_d * S.wwd[0];
    S.th[1] += P.h_d * S.wwd[1];
#else
#pragma unroll
    for (int i = 0; i < 3; i++) { S.v[i] += P.h * aw[i]; S.w[i] += P.h * acc[3 + i]; }
    S.ww[0] += P.h * acc[6];
    S.ww[1] += P.h * acc[7];
#pragma unroll
    for (int i = 0; i < 3; i++) S.p[i] += P.h_d * (double)S.v[i];
    quat_advance(S.q, (double)S.w[0], (double)S.w[1], (double)S.w[2], P.h_d);
    S.th[0] += P.h_d * (double)S.ww[0];
    S.th[1] += P.h_d * (double)S.ww[1];
#endif
    if constexpr (BLK) {
      // block smooth force: gravity only (isotropic inertia: no gyroscopic torque)
      R ab[3], al[3] = {fcon[8] * P.inv_mB - P.g * F.nB()[0], fcon[9] * P.inv_mB - P.g * F.nB()[1], fcon[10] * P.inv_mB - P.g * F.nB()[2]};
      mul_(F.RB, al, ab);
#if BRS_VEL64
#pragma unroll
      for (int i = 0; i < 3; i++) {
        S.bvd[i] += P.h_d * (double)ab[i]; S.bwd[i] += P.h_d * (double)(fcon[11 + i] * P.inv_IB);
#if !BRS_LAZY_VEL32
        S.bv[i] = (R)S.bvd[i]; S.bw[i] = (R)S.bwd[i];
#endif
      }
#pragma unroll
      for (int i = 0; i < 3; i++) S.bp[i] += P.h_d * S.bvd[i];
      quat_advance(S.bq, S.bwd[0], S.bwd[1], S.bwd[2], P.h_d);
#else
#pragma unroll
      for (int i = 0; i < 3; i++) { S.bv[i] += P.h * ab[i]; S.bw[i] += P.h * fcon[11 + i] * P.inv_IB; }
#pragma unroll
      for (int i = 0; i < 3; i++) S.bp[i] += P.h_d * (double)S.bv[i];
      quat_advance(S.bq, (double)S.bw[0], (double)S.bw[1], (double)S.bw[2], P.h_d);
#endif
    }
    S.time += P.h_d;
    // first guess of the next substep's active set
    S.pnfr = F.nfr; S.pnfb = F.nfb; S.pnc = F.nc; S.psels = F.sels; S.pmR = C.M.nR; S.pmB = C.M.nB; S.pmC = C.M.nC;
    for (int i_ = 0; i_ < 3; i_++) { BRS_PIN(S.vd[i_]); BRS_PIN(S.wd[i_]); BRS_PIN(S.p[i_]); }
    for (int i_ = 0; i_ < 4; i_++) BRS_PIN(S.q[i_]);
    BRS_PIN(S.wwd[0]); BRS_PIN(S.wwd[1]);
    BRS_TOC(7);
    BRS_STAT(int li = stats().last_iters[0]; stats().hist[li > 16 ? 16 : li]++; stats().trips += li > 1 ? li : 1; stats().substeps++;
             stats().trips_alt += (li == 2 && stats().first_single) ? 1 : (li > 1 ? li : 1));  // if a single-row flip were repaired inside the trip
    BRS_MARK("end_done");
  }
  // un-flattened form (one lane at a time: host tests, single substeps)
  static BRS_HD void substep(const Params<R>& P, Store<R>& st, ES& S, CT ctrlL, CT ctrlR) {
    SubCtx C;
    sub_begin(P, st, S, ctrlL, ctrlR, C);
    while (!C.conv) sub_iter(P, st, S, C);
    sub_end(P, S, C);
  }

  // =================================================================================== env logic
  // (restates envs/RobotBaseEnv.py:127-246, env01_v2.py:16-71, env03_v1.py:26-114, env03_v2.py:14-59;
  //  pinned through oracle/ by tests/golden/envlogic.json)
  // scipy Rotation.from_quat([x, y, z, w]).as_euler('xyz') -> angles[0] (pitch), angles[2] (yaw): envs/RobotBaseEnv.py:127-135, :177-184.
  // scipy >= 1.10 (the reference pins 1.14.1) computes them from the quaternion (Bernardes & Viollet 2022): a = w - y, b = x + z,
  // c = w + y, d = z - x, second angle 2 atan2(hypot(c, d), hypot(a, b)) - pi/2, first/third = atan2(b, a) -/+ atan2(d, c) -- equal
  // to the two atan2 forms below except within 1e-7 rad of GIMBAL LOCK (the wheel axis vertical), where scipy sets the third
  // angle to 0 and puts the whole rotation about the vertical into the first: pitch = 2 atan2(b, a) (second angle -pi/2) or
  // -2 atan2(d, c) (+pi/2), wrapped to [-pi, pi].  The test on the ratio of the two hypotenuses is taken in fp64 (it resolves
  // 5e-8); pinned by tests/golden/envlogic.json: pitch_yaw_gimbal, generated from the reference's own methods with the real scipy.
  static BRS_HD void pitch_yaw(const double* xq, R& pitch, R& yaw) {
    if (xq[0] == 0.0) { pitch = 0; yaw = 0; return; }
    {
      const double a = xq[0] - xq[2], b = xq[1] + xq[3], c = xq[0] + xq[2], d = xq[3] - xq[1];
      const double ab2 = a * a + b * b, cd2 = c * c + d * d, k2 = 2.5e-15;  // tan(1e-7 / 2)^2
      const bool lock_lo = cd2 <= k2 * ab2, lock_hi = ab2 <= k2 * cd2;
      if (lock_lo || lock_hi) {  // measure-zero in a simulation; a wave skips this block
        R p = lock_lo ? (R)2 * atan2_((R)b, (R)a) : (R)-2 * atan2_((R)d, (R)c);
        const R PI = (R)3.14159265358979323846;
        if (p > PI) p -= 2 * PI;
        if (p < -PI) p += 2 * PI;
        pitch = p; yaw = 0;
        return;
      }
    }
    R w = (R)xq[0], x = (R)xq[1], y = (R)xq[2], z = (R)xq[3];
    R n2 = w * w + x * x + y * y + z * z, s = 2 * rcp_(n2);  // scipy normalises; atan2 is scale-free
    pitch = atan2_(s * (y * z + w * x), 1 - s * (x * x + y * y));
    yaw = atan2_(s * (x * y + w * z), 1 - s * (y * y + z * z));
  }
  static BRS_HD R get_pitch(const Params<R>& P, const ES& S, Stream<R>& rng) {
    R p, y;
    pitch_yaw(S.xq, p, y);
    if (P.noise) p += (rng.next() - (R)0.5) * (R)0.05;
    if (P.v3) p += S.poff;  // envs/env01_v3.py:23-25
    return p;
  }
  // envs/env01_v3.py:55-96
  static BRS_HD R get_reward_v3(const Params<R>& P, const ES& S, Stream<R>& rng) {
    R pitch = get_pitch(P, S, rng);
    R ws = (S.ww[0] - S.ww[1]) / (R)2, dv = S.tws - ws;
    R dv_s = abs_(min_(max_(dv, (R)-40), (R)40) / (R)40);
    R reward = (R)0.6 - abs_(pitch) * (R)0.05 - (R)0.15 * dv_s;
    R lean = (S.tws > 0 && S.tws > ws) ? (R)-1 : ((S.tws < 0 && S.tws < ws) ? (R)1 : ((S.tws > 0 && S.tws < ws) ? (R)1 : ((S.tws < 0 && S.tws > ws) ? (R)-1 : (R)0)));
    reward += lean * pitch * (R)10 * dv_s;
    reward -= (R)0.007 * abs_((R)0 - (S.ww[0] + S.ww[1]));
    return reward;
  }
  static BRS_HD R get_reward(const Params<R>& P, const ES& S, Stream<R>& rng) {
    if (P.v3) return get_reward_v3(P, S, rng);
    R dv = (R)0 - (S.ww[0] * (R)-1 + S.ww[1]) / (R)2;
    R dyd = (R)0 - S.w[2];
    R pitch = get_pitch(P, S, rng);
    return (R)1 - (R)0.025 * abs_(dyd) - abs_(pitch) + pitch * dv * (R)0.5;
  }
  static BRS_HD void get_obs(const Params<R>& P, ES& S, Stream<R>& rng, bool at_reset, float* obs) {
    R pitch = get_pitch(P, S, rng);
    R pitch2 = get_pitch(P, S, rng);
    R pitch_dot = 0;
    if (!at_reset) pitch_dot = (pitch2 - S.last_pitch) / ((R)P.nsub * P.h);
    S.last_pitch = pitch2;
    R vl = S.ww[0], vr = S.ww[1];
    R wheel_speed = (vl - vr) / (R)2, wheel_yaw = vl + vr;
    obs[0] = (float)(pitch / (R)0.25);
    obs[1] = (float)pitch_dot;
    obs[2] = (float)(vl / (R)170 * (R)4);
    obs[3] = (float)(vr / (R)170 * (R)4);
    obs[4] = (float)(((P.v3 ? S.tws : (R)0) - wheel_speed) / (R)170 * (R)4);
    obs[5] = (float)(((R)0 - wheel_yaw) / (R)45 * (R)3);
  }
  // scipy from_euler('xyz',[a,b,c]).as_quat() (x,y,z,w) written unchanged into MuJoCo's (w,x,y,z) slot
  static BRS_HD void euler_slot_quat(R a, R b, R c, double* q) {
    R sa, ca, sb, cb, sc, cc;
    sincos_(a * (R)0.5, &sa, &ca); sincos_(b * (R)0.5, &sb, &cb); sincos_(c * (R)0.5, &sc, &cc);
    R w = ca * cb * cc + sa * sb * sc, x = sa * cb * cc - ca * sb * sc, y = ca * sb * cc + sa * cb * sc, z = ca * cb * sc - sa * sb * cc;
    // renormalise in fp64 so the fp64 accumulator starts on the unit sphere
    double n = 1.0 / sqrt((double)x * x + (double)y * y + (double)z * z + (double)w * w);
    q[0] = x * n; q[1] = y * n; q[2] = z * n; q[3] = w * n;
  }
  static BRS_HD void set_block_pos_vel(const Params<R>& P, ES& S, Stream<R>& rng) {
    const R TWO_PI = (R)6.283185307179586476925;
    R ang, tx, tz;
    R xp0 = (R)S.xp[0], xp1 = (R)S.xp[1];
    if (P.throw_v2) {
      R p, yaw;
      pitch_yaw(S.xq, p, yaw);
      ang = -yaw;
      if (!S.side_front) ang += (R)3.14159265358979323846;
    } else
      ang = rng.next() * TWO_PI;
    R sa, ca;
    sincos_(ang, &sa, &ca);
    R ox = (R)0.3 * sa, oy = (R)0.3 * ca;  // block position relative to the robot
    if (P.throw_v2) { tx = (rng.next() - (R)0.5) * (R)0.02; tz = rng.next() * (R)0.025 + (R)0.13; }
    else { tx = (rng.next() - (R)0.5) * (R)0.06; tz = rng.next() * (R)0.075 + (R)0.1; }
    R vx = tx - ox, vy = -oy, vz = tz - (R)0.15;
    R k = P.block_speed * rsqrt_(vx * vx + vy * vy + vz * vz);
    R xr = rng.next() * TWO_PI, yr = rng.next() * TWO_PI, zr = rng.next() * TWO_PI;
    S.bp[0] = (double)(ox) + S.xp[0]; S.bp[1] = (double)(oy) + S.xp[1]; S.bp[2] = 0.15;
    (void)xp0; (void)xp1;
    euler_slot_quat(xr, yr, zr, S.bq);
    S.bv[0] = vx * k; S.bv[1] = vy * k; S.bv[2] = vz * k;
#if BRS_VEL64
    if constexpr (BLK) { S.bvd[0] = (double)S.bv[0]; S.bvd[1] = (double)S.bv[1]; S.bvd[2] = (double)S.bv[2]; }
#endif
  }
  static BRS_HD void env_reset(const Params<R>& P, ES& S, Stream<R>& rng, float* obs) {
    const R TWO_PI = (R)6.283185307179586476925;
    if (P.v3) {  // envs/env01_v3.py:40-53: two draws of the seeded generator BEFORE Env01.reset_model
      S.tws = 0;
      R d = (R)-10 + (R)20 * rng.next();
      S.dts = d > 0 ? d + (R)10 : d - (R)10;
      S.poff = (R)-0.0349066 + (R)(2 * 0.0349066) * rng.next();
    }
    // MujocoEnv.reset -> mj_resetData ; reset_model: qpos0 + U(-0.01,0.01)^nq, qpos[2] = 0
    R n[16];
#pragma unroll
    for (int i = 0; i < (BLK ? 16 : 9); i++) n[i] = (R)-0.01 + (R)0.02 * rng.next();
    S.p[0] = n[0]; S.p[1] = n[1]; S.p[2] = 0;
    S.th[0] = n[7]; S.th[1] = n[8];
#pragma unroll
    for (int i = 0; i < 3; i++) { S.v[i] = 0; S.w[i] = 0; }
    S.ww[0] = S.ww[1] = 0;
#pragma unroll
    for (int i = 0; i < NV; i++) S.a[i] = 0;
    S.time = 0; S.elapsed = 0; S.ep_return = 0;
    R xr = (rng.next() - (R)0.5) * TWO_PI, yr = (rng.next() - (R)0.5) * P.Sy, zr = (rng.next() - (R)0.5) * P.Sz;
    euler_slot_quat(xr, yr, zr, S.q);
    if (P.per_env_mu) S.muw = rng.next() / (R)2 + (R)0.5;  // envs/env02_v1.py:61-65, after the pose draws
#pragma unroll
    for (int i = 0; i < 4; i++) S.xq[i] = S.q[i];
#pragma unroll
    for (int i = 0; i < 3; i++) S.xp[i] = S.p[i];
    if constexpr (BLK) {
#pragma unroll
      for (int i = 0; i < 3; i++) { S.bw[i] = 0; }
      S.sync_vel64();
      set_block_pos_vel(P, S, rng);
      S.block_timer = -1.0;
    } else
      S.sync_vel64();
    get_obs(P, S, rng, true, obs);
  }

  // ---- one full env step = env_pre (reward + control law on the PRE-step state) -> nsub substeps -> env_post
  static BRS_HD R env_pre(const Params<R>& P, ES& S, Stream<R>& rng, float a0, float a1, CT& ctrlL, CT& ctrlR) {
    if (P.v3) {  // envs/env01_v3.py:28-36: schedule keyed on data.time at the start of step
      const double t = S.time;  // fp64 like the reference: the accumulated time sits within rounding of the thresholds
      if (t > 5.5) S.tws = (R)3 * S.dts;
      else if (t > 4.5) S.tws = (R)2 * S.dts;
      else if (t > 3.0) S.tws = -S.dts;
      else if (t > 1.0) S.tws = S.dts;
    }
    R rew = get_reward(P, S, rng);
#if BRS_VEL64
    ctrlL = S.wwd[0] + (double)a0 * 4.0;  // envs/env01_v2.py:31-36 ; the env does not clip the action
    ctrlR = S.wwd[1] + (double)a1 * 4.0;
#else
    ctrlL = S.ww[0] + (R)a0 * (R)4;  // envs/env01_v2.py:31-36 ; the env does not clip the action
    ctrlR = S.ww[1] + (R)a1 * (R)4;
#endif
    return rew;
  }
  static BRS_HD void env_post(const Params<R>& P, ES& S, Stream<R>& rng, R rew, float* obs, float* terminal_obs, float& reward,
                              int& terminated, int& truncated) {
    // mj_check*: NaN / runaway -> reset the simulation (counted)
    bool bad = isbad_(S.p[0]) || isbad_(S.p[2]) || isbad_(S.q[0]) || isbad_(S.v[0]) || isbad_(S.v[2]) || isbad_(S.w[0]) ||
               isbad_(S.ww[0]) || isbad_(S.ww[1]) || abs_(S.v[0]) > (R)1e10 || abs_(S.v[2]) > (R)1e10;
    if constexpr (BLK) bad = bad || isbad_(S.bp[0]) || isbad_(S.bp[2]) || isbad_(S.bq[0]) || isbad_(S.bv[0]) || isbad_(S.bw[0]);
    if (bad) {
      float tmp[6];
      S.bad++;
      env_reset(P, S, rng, tmp);
    }
    if constexpr (BLK) {  // envs/env03_v1.py:39-49
      R bv2 = S.bv[0] * S.bv[0] + S.bv[1] * S.bv[1] + S.bv[2] * S.bv[2];
      if (bv2 < (R)0.01 && S.block_timer < 0) {
        S.bp[0] = 10; S.bp[1] = 10; S.bp[2] = 0;
        S.block_timer = S.time;
      }
      if (S.block_timer >= 0 && (S.time - S.block_timer) > P.block_delay) {
        set_block_pos_vel(P, S, rng);
        S.block_timer = -1.0;
      }
    }
    terminated = abs_(get_pitch(P, S, rng)) > (R)(50.0 * 3.14159265358979323846 / 180.0) ? 1 : 0;
    get_obs(P, S, rng, false, obs);
    S.elapsed++;
    S.ep_return += rew;
    truncated = S.elapsed >= P.max_episode_steps ? 1 : 0;
    reward = (float)rew;
#pragma unroll
    for (int i = 0; i < 6; i++) terminal_obs[i] = obs[i];
    if (P.auto_reset && (terminated || truncated)) env_reset(P, S, rng, obs);
  }
  static BRS_HD void env_step(const Params<R>& P, Store<R>& st, ES& S, Stream<R>& rng, float a0, float a1, float* obs,
                              float* terminal_obs, float& reward, int& terminated, int& truncated) {
    CT ctrlL, ctrlR;
    R rew = env_pre(P, S, rng, a0, a1, ctrlL, ctrlR);
    for (int k = 0; k < P.nsub; k++) {
      if (k == P.nsub - 1) {  // accessor pose = kinematics of the LAST forward pass (lags qpos by one substep)
#pragma unroll
        for (int i = 0; i < 4; i++) S.xq[i] = S.q[i];
#pragma unroll
        for (int i = 0; i < 3; i++) S.xp[i] = S.p[i];
      }
      substep(P, st, S, ctrlL, ctrlR);
    }
    S.derive_vel32();
    env_post(P, S, rng, rew, obs, terminal_obs, reward, terminated, truncated);
  }
};

}  // namespace brs

// brs_model.hpp -- model constants of the two-wheel balance robot, re-derived from the reference's
// MuJoCo XML (numbers re-typed, nothing copied):
//   envs/robot-02.xml:4-25   torso box, two hinge wheels (cylinders), two velocity actuators
//   envs/env01_v1.xml:2-3    inertiafromgeom, timestep 2e-5, implicitfast, gravity
//   envs/env01_v1.xml:27     floor plane z = -0.02
//   envs/env01_v1.xml:30-33  explicit floor<->wheel pairs (mu .9, solref .02 .5, solimp .5 .5 .002)
//   envs/env03_v1.xml:31-37  free "block" box (half 0.02, margin .002, solref .005 .9); no <pair>s
// (paths under /root/reference/src/balance_robot/).
//
// Host-side, double precision.  The closed forms here (gyrostat mass matrix in body coordinates)
// are what the HIP kernel uses; oracle/ derives the same quantities numerically from a body tree.
#pragma once
#include <cmath>
#include <cstdint>

namespace brs {

enum Variant { ENV01_V1 = 0, ENV01_V2 = 1, ENV03_V1 = 2, ENV03_V2 = 3, ENV01_V3 = 4, ENV02_V1 = 5 };
enum ContactCls { CC_WHEEL_FLOOR = 0, CC_TORSO_FLOOR = 1, CC_BLOCK_FLOOR = 2, CC_BLOCK_ROBOT = 3, CC_COUNT = 4 };

// one geom-pair class of contact parameters, pre-digested for the kernel
template <typename R>
struct ContactClass {
  R mu;         // sliding friction of the pyramid
  R K, B;       // reference acceleration: aref = -B*vel - K*imp*(dist - margin)
  R d0, d1;     // impedance range (solimp[0..1]); power-2 / midpoint-0.5 sigmoid over `width`
  R inv_width;  // 1/solimp[2]  (0 => constant impedance d0)
  R margin;
  R cD;         // pyramidal regulariser: R_row = cD * (1-imp)/imp,  cD = 2 mu^2 (1+mu^2) (tran1+tran2)
};

template <typename R>
struct Params {
  // integration
  R h;
  double h_d;
  int nsub;
  R g;
  // robot as a gyrostat about the torso-frame origin (body coordinates: x axle, y forward, z up)
  R m, cz, mcz, Ixx, Iyy, Izz, Ia;
  R inv_m, inv_Izz, inv_det_xy;  // closed-form pieces of M^-1
  R Ixx_red0, inv_Ia;            // Ixx - 2 Ia - m cz^2  (d = 0 case)
  // geometry
  R wheel_px, wheel_pz, wheel_r, wheel_hl;
  R torso_sx, torso_sy, torso_sz, torso_cz;
  R floor_z;
  double floor_z_d;
  // the same geometry in fp64 for the code that DECIDES whether a contact point exists (brs_core.hpp: Floor64, collide_coupled):
  // (double)(float)0.034 is 1.8 nm off 0.034 -- a grazing wheel moves 20 nm per substep, so that bias alone moves the substep at
  // which its contact point switches on or off against the fp64 oracle
  double wheel_px_d, wheel_pz_d, wheel_r_d, wheel_hl_d, torso_s_d[3], torso_cz_d, block_s_d;
  double margin_d[CC_COUNT];
  double kv_d, ctrlrange_d, forcerange_d;  // the servo's clamp decisions are taken in fp64 as well (brs_core.hpp: sub_begin)
  // actuators / passive
  R kv, ctrlrange, forcerange, damping;
  // block
  R mB, IB, block_s, inv_mB, inv_IB;
  R torso_brad, block_brad, wheel_brad;  // bounding-sphere radii for the coupled-contact early-out
  ContactClass<R> cc[CC_COUNT];
  // wheel<->block uses CC_BLOCK_ROBOT with the wheel's invweight: separate cD
  R cD_block_wheel;
  R tran_wheel;  // wheel body_invweight0 (translational): per-episode friction (Env02) rebuilds cD = 2 mu^2 (1+mu^2) tran
  int per_env_mu, v3;  // Env02-v1: wheel/floor friction drawn per episode; Env01-v3: target-speed schedule, pitch offset, own reward
  // env level
  int variant, family, noise, auto_reset, max_episode_steps, throw_v2;
  R Sy, Sz, block_speed;
  double block_delay;
  uint64_t seed;
  int64_t gid_base;
};

struct ModelRaw {
  // raw numbers from the XML
  double torso_s[3] = {0.05, 0.0185, 0.0855}, torso_gz = 0.0995;
  double wheel_px = 0.074, wheel_pz = 0.034, wheel_r = 0.034, wheel_hl = 0.013;
  double block_s = 0.02, density = 1000.0, floor_z = -0.02;
  double kv = 4.0, ctrlrange = 78.54, forcerange = 0.65, damping = 0.01;
  double h = 0.00002, g = 9.81;
};

// constexpr square root (Newton): make_params is evaluated at COMPILE time for the step kernels (brs_kernels.hip), so that
// the ~100 model constants become instruction literals instead of SGPRs that spill
constexpr double csqrt(double x) {
  if (!(x > 0)) return 0;
  double r = x > 1 ? x : 1.0;
  for (int k = 0; k < 200; k++) {
    const double n = 0.5 * (r + x / r);
    if (n == r) break;
    r = n;
  }
  return r;
}

// solve (M_b + diag(0,..,0,dL,dR)) x = f for the robot's 8 dofs in body coordinates
// order: [alpha_x, alpha_y, alpha_z, wdot_x, wdot_y, wdot_z, wdot_L, wdot_R]
constexpr void msolve_d(double m, double cz, double Ixx, double Iyy, double Izz, double Ia, const double* f, double dL,
                     double dR, double* x) {
  double mcz = m * cz, det = m * Iyy - mcz * mcz;
  x[2] = f[2] / m;
  x[5] = f[5] / Izz;
  x[0] = (Iyy * f[0] - mcz * f[4]) / det;
  x[4] = (m * f[4] - mcz * f[0]) / det;
  double IL = Ia + dL, IR = Ia + dR;
  double red = Ixx - Ia * Ia / IL - Ia * Ia / IR - m * cz * cz;
  x[3] = (f[3] + Ia * f[6] / IL - Ia * f[7] / IR + cz * f[1]) / red;
  x[1] = (f[1] + mcz * x[3]) / m;
  x[6] = (f[6] + Ia * x[3]) / IL;
  x[7] = (f[7] - Ia * x[3]) / IR;
}

template <typename R>
constexpr Params<R> make_params(int variant, uint32_t flags_auto_reset, int noise_override /* -1 default, 0 off, 1 on */,
                             int max_episode_steps, int nsub, double timestep, uint64_t seed, int64_t gid_base) {
  const double PI = 3.14159265358979323846;
  ModelRaw r;
  Params<R> p{};
  double h = timestep > 0 ? timestep : r.h;
  p.h = (R)h; p.h_d = h; p.nsub = nsub > 0 ? nsub : 250; p.g = (R)r.g;
  // inertiafromgeom: uniform density 1000
  double mT = 8 * r.torso_s[0] * r.torso_s[1] * r.torso_s[2] * r.density;
  double IT[3] = {mT / 3 * (r.torso_s[1] * r.torso_s[1] + r.torso_s[2] * r.torso_s[2]),
                  mT / 3 * (r.torso_s[0] * r.torso_s[0] + r.torso_s[2] * r.torso_s[2]),
                  mT / 3 * (r.torso_s[0] * r.torso_s[0] + r.torso_s[1] * r.torso_s[1])};
  double mW = PI * r.wheel_r * r.wheel_r * 2 * r.wheel_hl * r.density;
  double Ia = 0.5 * mW * r.wheel_r * r.wheel_r, It = mW * (3 * r.wheel_r * r.wheel_r + 4 * r.wheel_hl * r.wheel_hl) / 12;
  double m = mT + 2 * mW;
  double cz = (mT * r.torso_gz + 2 * mW * r.wheel_pz) / m;
  // inertia about the torso-frame origin, wheels locked (parallel axis); products cancel by symmetry
  double Ixx = IT[0] + mT * r.torso_gz * r.torso_gz + 2 * (Ia + mW * r.wheel_pz * r.wheel_pz);
  double Iyy = IT[1] + mT * r.torso_gz * r.torso_gz + 2 * (It + mW * (r.wheel_px * r.wheel_px + r.wheel_pz * r.wheel_pz));
  double Izz = IT[2] + 2 * (It + mW * r.wheel_px * r.wheel_px);
  p.m = (R)m; p.cz = (R)cz; p.mcz = (R)(m * cz); p.Ixx = (R)Ixx; p.Iyy = (R)Iyy; p.Izz = (R)Izz; p.Ia = (R)Ia;
  p.inv_m = (R)(1 / m); p.inv_Izz = (R)(1 / Izz); p.inv_det_xy = (R)(1 / (m * Iyy - m * cz * m * cz));
  p.Ixx_red0 = (R)(Ixx - 2 * Ia - m * cz * cz); p.inv_Ia = (R)(1 / Ia);
  p.wheel_px = (R)r.wheel_px; p.wheel_pz = (R)r.wheel_pz; p.wheel_r = (R)r.wheel_r; p.wheel_hl = (R)r.wheel_hl;
  p.torso_sx = (R)r.torso_s[0]; p.torso_sy = (R)r.torso_s[1]; p.torso_sz = (R)r.torso_s[2]; p.torso_cz = (R)r.torso_gz;
  p.floor_z = (R)r.floor_z; p.floor_z_d = r.floor_z;
  p.wheel_px_d = r.wheel_px; p.wheel_pz_d = r.wheel_pz; p.wheel_r_d = r.wheel_r; p.wheel_hl_d = r.wheel_hl;
  p.torso_s_d[0] = r.torso_s[0]; p.torso_s_d[1] = r.torso_s[1]; p.torso_s_d[2] = r.torso_s[2]; p.torso_cz_d = r.torso_gz; p.block_s_d = r.block_s;
  p.kv = (R)r.kv; p.ctrlrange = (R)r.ctrlrange; p.forcerange = (R)r.forcerange; p.damping = (R)r.damping;
  p.kv_d = r.kv; p.ctrlrange_d = r.ctrlrange; p.forcerange_d = r.forcerange;
  double mB = 8 * r.block_s * r.block_s * r.block_s * r.density, IB = mB / 3 * 2 * r.block_s * r.block_s;
  p.mB = (R)mB; p.IB = (R)IB; p.block_s = (R)r.block_s; p.inv_mB = (R)(1 / mB); p.inv_IB = (R)(1 / IB);
  p.torso_brad = (R)csqrt(r.torso_s[0] * r.torso_s[0] + r.torso_s[1] * r.torso_s[1] + r.torso_s[2] * r.torso_s[2]);
  p.block_brad = (R)(r.block_s * csqrt(3.0));
  p.wheel_brad = (R)csqrt(r.wheel_r * r.wheel_r + r.wheel_hl * r.wheel_hl);

  // body_invweight0 (translational), as MuJoCo computes it at qpos0: mean diagonal of Jp M^-1 Jp^T at the body COM
  auto tran_of = [&](double rx, double ry, double rz, int wheel /*0 none, 1 L, 2 R*/) {
    // COM Jacobian rows (body coordinates, R = I at qpos0): e_k | (r x e_k)... row k = [e_k, r x e_k, 0, 0]
    // a wheel's COM lies on its hinge axis: the hinge column is zero
    (void)wheel;
    double tr = 0;
    const double rr[3] = {rx, ry, rz};
    for (int k = 0; k < 3; k++) {
      double e[3] = {0, 0, 0};
      e[k] = 1;
      double rowv[8] = {e[0], e[1], e[2], rr[1] * e[2] - rr[2] * e[1], rr[2] * e[0] - rr[0] * e[2], rr[0] * e[1] - rr[1] * e[0], 0, 0};
      double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      msolve_d(m, cz, Ixx, Iyy, Izz, Ia, rowv, 0, 0, x);
      for (int j = 0; j < 8; j++) tr += rowv[j] * x[j];
    }
    return tr / 3;
  };
  double tran_torso = tran_of(0, 0, r.torso_gz, 0);
  double tran_wheel = tran_of(-r.wheel_px, 0, r.wheel_pz, 1);
  double tran_block = 1 / mB;

  int family = (variant == ENV03_V1 || variant == ENV03_V2) ? 3 : 1;
  auto mk = [&](double mu, double tc, double dr, double d0, double d1, double width, double margin, double tran) {
    ContactClass<R> c{};
    if (tc < 2 * h) tc = 2 * h;  // refsafe
    c.mu = (R)mu;
    c.K = (R)(1.0 / (d1 * d1 * tc * tc * dr * dr));
    c.B = (R)(2.0 / (d1 * tc));
    c.d0 = (R)d0; c.d1 = (R)d1;
    c.inv_width = (R)((d0 == d1 || width <= 1e-15) ? 0.0 : 1.0 / width);
    c.margin = (R)margin;
    c.cD = (R)(2 * mu * mu * (1 + mu * mu) * tran);
    return c;
  };
  p.margin_d[CC_WHEEL_FLOOR] = 0.0; p.margin_d[CC_TORSO_FLOOR] = 0.0; p.margin_d[CC_BLOCK_FLOOR] = 0.002; p.margin_d[CC_BLOCK_ROBOT] = 0.002;
  if (family == 1 && variant != ENV02_V1) p.cc[CC_WHEEL_FLOOR] = mk(0.9, 0.02, 0.5, 0.5, 0.5, 0.002, 0.0, tran_wheel);  // explicit <pair>s
  else p.cc[CC_WHEEL_FLOOR] = mk(1.0, 0.02, 1.0, 0.9, 0.95, 0.001, 0.0, tran_wheel);
  p.cc[CC_TORSO_FLOOR] = mk(1.0, 0.02, 1.0, 0.9, 0.95, 0.001, 0.0, tran_torso);
  // block pairs: margin max(0, .002), solref mixed 50/50 -> (0.0125, 0.95), default solimp, mu max(1,1)
  p.cc[CC_BLOCK_FLOOR] = mk(1.0, 0.0125, 0.95, 0.9, 0.95, 0.001, 0.002, tran_block);
  p.cc[CC_BLOCK_ROBOT] = mk(1.0, 0.0125, 0.95, 0.9, 0.95, 0.001, 0.002, tran_block + tran_torso);
  p.cD_block_wheel = (R)(2 * 1.0 * (1 + 1.0) * (tran_block + tran_wheel));
  p.tran_wheel = (R)tran_wheel;
  p.per_env_mu = variant == ENV02_V1; p.v3 = variant == ENV01_V3;

  p.variant = variant; p.family = family;
  p.noise = (variant == ENV01_V2) ? 1 : 0;  // envs/env01_v2.py:16-20 ; Env03_v2 does NOT inherit it
  if (noise_override >= 0) p.noise = noise_override;
  p.auto_reset = flags_auto_reset ? 1 : 0;
  p.max_episode_steps = max_episode_steps > 0 ? max_episode_steps : (variant == ENV03_V2 ? 1200 : 6000);
  p.throw_v2 = variant == ENV03_V2;
  if (variant == ENV01_V2) { p.Sy = (R)0.2; p.Sz = (R)2.0; } else { p.Sy = (R)0.4; p.Sz = (R)0.4; }
  p.block_speed = (R)(variant == ENV03_V2 ? 7.5 : 5.0);
  p.block_delay = variant == ENV03_V2 ? 0.5 : 0.0;
  p.seed = seed; p.gid_base = gid_base;
  return p;
}

}  // namespace brs

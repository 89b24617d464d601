/*
 * brs_oracle.c -- CPU fp64 ORACLE (test infrastructure; see brs_oracle.h for the rules and the
 * parity status).  Plain C99, no dependencies.  Deliberately GENERAL and slow-ish: dynamics come from
 * numeric Jacobians / a body tree (not from the closed forms the HIP kernel uses), so agreement
 * between the two is a real cross-check of both derivations.
 *
 * Reference citations: paths under /root/reference/src/balance_robot/ .
 */
#include "brs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NVMAX BO_MAXNV
#define NQMAX BO_MAXNQ
#define MAXCON BO_MAXCON
#define MAXEFC (4 * MAXCON)
#define MJ_MINVAL 1e-15
#define PI 3.14159265358979323846

/* ============================================================================================
 * small linear algebra
 * ========================================================================================== */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* o, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline void mulMatVec3(double* o, const double* R, const double* v) { /* o = R v, R row-major */
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void mulMatTVec3(double* o, const double* R, const double* v) { /* o = R^T v */
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mulMat3(double* o, const double* A, const double* B) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(o, t, sizeof t);
}
static void quat2mat(double* R, const double* q) { /* q = (w,x,y,z) unit */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MJ_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void mulQuat(double* o, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                 a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                 a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(o, t, sizeof t);
}
/* MuJoCo mju_quatIntegrate: normalise quat, then right-multiply by exp(scale*vel) (body frame) */
static void quatIntegrate(double* q, const double* vel, double scale) {
  double ax[3] = {vel[0], vel[1], vel[2]};
  double n = norm3(ax), ang;
  if (n < MJ_MINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; n = 0; } else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
  ang = scale * n;
  double s = sin(0.5 * ang), qr[4] = {cos(0.5 * ang), ax[0] * s, ax[1] * s, ax[2] * s};
  normalize4(q);
  mulQuat(q, q, qr);
}

/* dense Cholesky A = L L^T (lower, row-major n x n with leading dim ld); returns 0 on success */
static int chol(const double* A, int n, int ld, double* L) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * ld + j];
      for (int k = 0; k < j; k++) s -= L[i * ld + k] * L[j * ld + k];
      if (i == j) {
        if (s < MJ_MINVAL) s = MJ_MINVAL;
        L[i * ld + i] = sqrt(s);
      } else
        L[i * ld + j] = s / L[j * ld + j];
    }
  return 0;
}
static void chol_solve(const double* L, int n, int ld, const double* b, double* x) {
  double y[NVMAX];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * ld + k] * y[k];
    y[i] = s / L[i * ld + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k * ld + i] * x[k];
    x[i] = s / L[i * ld + i];
  }
}

/* ============================================================================================
 * Philox4x32-10 (Salmon et al., SC'11) -- this project's RNG, shared spec with the HIP kernel.
 * The reference draws from numpy's global MT19937 (envs/env01_v2.py:19,59-63) and the gymnasium
 * seeded generator (:53); a batched simulator cannot share one sequential stream, so every env owns
 * the counter-based stream Philox(key=seed, ctr=(n, 0, env_gid_lo, env_gid_hi)).
 * ========================================================================================== */
void bo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static void philox_env(uint64_t seed, int64_t gid, uint32_t n, uint32_t out[4]) {
  uint32_t ctr[4] = {n, 0u, (uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32)};
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  bo_philox4x32_10(ctr, key, out);
}
double bo_uniform(uint64_t seed, int64_t gid, uint32_t n, int k) {
  uint32_t o[4];
  philox_env(seed, gid, n, o);
  return (double)(o[k & 3] >> 8) * (1.0 / 16777216.0);
}

/* ============================================================================================
 * model (re-typed from envs/robot-02.xml, envs/env01_v1.xml, envs/env03_v1.xml)
 * ========================================================================================== */
enum { B_WORLD = 0, B_TORSO = 1, B_LW = 2, B_RW = 3, B_BLOCK = 4 };

typedef struct {
  double mu, solref[2], solimp[5], margin;
} cparam;

typedef struct {
  int variant, family, nq, nv, nbody, has_block;
  double h;
  int nsub;
  double gravity[3];
  double mass[5], inertia[5][3], ipos[5][3];
  double wheel_pos[2][3], wheel_axis[2][3];
  double torso_size[3], torso_gpos[3];
  double wheel_r, wheel_hl, wheel_gmat[9];
  double block_size[3];
  double floor_z;
  double kv, ctrlrange, forcerange, damping;
  cparam cp_wheel_floor, cp_torso_floor, cp_block;
  double invweight0[5][2], meaninertia;
  /* solver options (MuJoCo defaults; envs/env01_v1.xml:3 sets only timestep/integrator/gravity) */
  double tolerance, ls_tolerance;
  int iterations, ls_iterations;
  /* env-level */
  int noise;
  double Sy, Sz;
  int max_episode_steps;
  double block_delay, block_speed;
  int throw_v2, auto_reset;
} model_t;

typedef struct {
  double qpos[NQMAX], qvel[NVMAX], warm[NVMAX], time;
  double xquat[4], xpos[3]; /* accessor pose: data.body("robot_body").xquat/.xpos (may lag qpos by one substep) */
  double last_pitch, block_timer; /* NaN = None */
  int elapsed;
  uint32_t rng_ctr;
  int side_front;
  double ep_return;
  int bad_count;
  double last_ctrl[2];
  /* later variants: wheel/floor friction of this episode (Env02, env02_v1.py:57-65), target-speed schedule and pitch
   * offset (Env01_v3, env01_v3.py:16-53) */
  double muw, dts, poff, tws;
  double* script;
  int script_n, script_pos;
  int stub;
  double stub_qpos[NQMAX], stub_qvel[NVMAX], stub_xquat[4], stub_xpos[3];
} env_t;

struct bo_handle {
  model_t m;
  int n;
  uint64_t seed;
  int64_t gid_base;
  env_t* e;
  int nthreads;
};

static void set_default_solimp(double* s) { s[0] = 0.9; s[1] = 0.95; s[2] = 0.001; s[3] = 0.5; s[4] = 2; }

/* ---------------- kinematics ---------------- */
typedef struct {
  double xpos[5][3], xmat[5][9], xipos[5][3];
  double tquat[4], bquat[4];
} kin_t;

static void kinematics(const model_t* m, const double* qpos, kin_t* k) {
  memset(k, 0, sizeof *k);
  for (int i = 0; i < 3; i++) k->xmat[0][4 * i] = 1;
  /* torso: free joint at world origin (envs/robot-02.xml:4-5) */
  memcpy(k->tquat, qpos + 3, 4 * sizeof(double));
  normalize4(k->tquat);
  quat2mat(k->xmat[B_TORSO], k->tquat);
  memcpy(k->xpos[B_TORSO], qpos, 3 * sizeof(double));
  double t[3];
  mulMatVec3(t, k->xmat[B_TORSO], m->ipos[B_TORSO]);
  for (int i = 0; i < 3; i++) k->xipos[B_TORSO][i] = k->xpos[B_TORSO][i] + t[i];
  /* wheels: hinge about wheel_axis (body frame) at the wheel body origin (envs/robot-02.xml:9-18) */
  for (int w = 0; w < 2; w++) {
    int b = B_LW + w;
    mulMatVec3(t, k->xmat[B_TORSO], m->wheel_pos[w]);
    for (int i = 0; i < 3; i++) k->xipos[b][i] = k->xpos[b][i] = k->xpos[B_TORSO][i] + t[i];
    double th = qpos[7 + w], s = sin(0.5 * th), qh[4] = {cos(0.5 * th), m->wheel_axis[w][0] * s, m->wheel_axis[w][1] * s, m->wheel_axis[w][2] * s};
    double Rh[9];
    quat2mat(Rh, qh);
    mulMat3(k->xmat[b], k->xmat[B_TORSO], Rh);
  }
  if (m->has_block) {
    memcpy(k->bquat, qpos + 12, 4 * sizeof(double));
    normalize4(k->bquat);
    quat2mat(k->xmat[B_BLOCK], k->bquat);
    memcpy(k->xpos[B_BLOCK], qpos + 9, 3 * sizeof(double));
    memcpy(k->xipos[B_BLOCK], qpos + 9, 3 * sizeof(double));
  }
}

/* translational (Jp) and rotational (Jr) Jacobian of a world point attached to body b; [3][nv] */
static void jac_point(const model_t* m, const kin_t* k, int b, const double* p, double Jp[3][NVMAX], double Jr[3][NVMAX]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < NVMAX; j++) Jp[i][j] = Jr[i][j] = 0;
  if (b == B_WORLD) return;
  int root = (b == B_BLOCK) ? B_BLOCK : B_TORSO, d0 = (b == B_BLOCK) ? 8 : 0;
  double r[3] = {p[0] - k->xpos[root][0], p[1] - k->xpos[root][1], p[2] - k->xpos[root][2]};
  for (int c = 0; c < 3; c++) {
    Jp[c][d0 + c] = 1; /* world-frame linear dofs */
    double ax[3] = {k->xmat[root][c], k->xmat[root][3 + c], k->xmat[root][6 + c]}, t[3]; /* body-frame angular dofs */
    cross3(t, ax, r);
    for (int i = 0; i < 3; i++) { Jr[i][d0 + 3 + c] = ax[i]; Jp[i][d0 + 3 + c] = t[i]; }
  }
  if (b == B_LW || b == B_RW) {
    int w = b - B_LW;
    double ax[3], t[3], rw[3] = {p[0] - k->xpos[b][0], p[1] - k->xpos[b][1], p[2] - k->xpos[b][2]};
    mulMatVec3(ax, k->xmat[B_TORSO], m->wheel_axis[w]);
    cross3(t, ax, rw);
    for (int i = 0; i < 3; i++) { Jr[i][6 + w] = ax[i]; Jp[i][6 + w] = t[i]; }
  }
}

/* mass matrix  M = sum_b  m Jp^T Jp + Jr^T (R I R^T) Jr  (at each body's COM) -- equals MuJoCo's CRB result */
static void mass_matrix(const model_t* m, const kin_t* k, double* M) {
  int nv = m->nv;
  memset(M, 0, sizeof(double) * NVMAX * NVMAX);
  for (int b = 1; b < m->nbody; b++) {
    double Jp[3][NVMAX], Jr[3][NVMAX], Iw[9], T[9], D[9] = {m->inertia[b][0], 0, 0, 0, m->inertia[b][1], 0, 0, 0, m->inertia[b][2]};
    jac_point(m, k, b, k->xipos[b], Jp, Jr);
    mulMat3(T, k->xmat[b], D);
    double Rt[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = k->xmat[b][3 * j + i];
    mulMat3(Iw, T, Rt);
    for (int i = 0; i < nv; i++)
      for (int j = 0; j < nv; j++) {
        double s = 0;
        for (int c = 0; c < 3; c++) s += m->mass[b] * Jp[c][i] * Jp[c][j];
        for (int c = 0; c < 3; c++)
          for (int d = 0; d < 3; d++) s += Jr[c][i] * Iw[3 * c + d] * Jr[d][j];
        M[i * NVMAX + j] += s;
      }
  }
}

/* bias = C(q,v) v - gravity forces  (recursive Newton-Euler with qacc = 0, projected with the Jacobians) */
static void bias_forces(const model_t* m, const kin_t* k, const double* qvel, double* bias) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) bias[i] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int root = (b == B_BLOCK) ? B_BLOCK : B_TORSO, d0 = (b == B_BLOCK) ? 8 : 0;
    double wroot[3], wb[3], alpha[3] = {0, 0, 0}, acom[3], r[3], t[3], t2[3];
    mulMatVec3(wroot, k->xmat[root], qvel + d0 + 3); /* body-frame angular velocity -> world */
    memcpy(wb, wroot, sizeof wb);
    /* root origin: linear dofs are world-frame velocities of the origin, angular dofs are body-frame:
     * with qacc = 0 both the origin acceleration and the angular acceleration of the root vanish */
    for (int i = 0; i < 3; i++) r[i] = k->xipos[b][i] - k->xpos[root][i];
    cross3(t, wroot, r);
    cross3(acom, wroot, t); /* centripetal acceleration of this body's COM about the root origin */
    if (b == B_LW || b == B_RW) {
      int w = b - B_LW;
      double ax[3];
      mulMatVec3(ax, k->xmat[B_TORSO], m->wheel_axis[w]);
      for (int i = 0; i < 3; i++) ax[i] *= qvel[6 + w];
      cross3(alpha, wroot, ax); /* d/dt (axis * qdot) with qacc = 0 */
      for (int i = 0; i < 3; i++) wb[i] += ax[i];
    }
    /* inertia in world frame */
    double Iw[9], T[9], Rt[9], D[9] = {m->inertia[b][0], 0, 0, 0, m->inertia[b][1], 0, 0, 0, m->inertia[b][2]};
    mulMat3(T, k->xmat[b], D);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = k->xmat[b][3 * j + i];
    mulMat3(Iw, T, Rt);
    double F[3], N[3];
    for (int i = 0; i < 3; i++) F[i] = m->mass[b] * (acom[i] - m->gravity[i]);
    mulMatVec3(t, Iw, alpha);
    mulMatVec3(t2, Iw, wb);
    cross3(N, wb, t2);
    for (int i = 0; i < 3; i++) N[i] += t[i];
    double Jp[3][NVMAX], Jr[3][NVMAX];
    jac_point(m, k, b, k->xipos[b], Jp, Jr);
    for (int j = 0; j < nv; j++)
      for (int c = 0; c < 3; c++) bias[j] += Jp[c][j] * F[c] + Jr[c][j] * N[c];
  }
}

/* ---------------- collision ---------------- */
/* MuJoCo mju_makeFrame: frame[0:3] = normal given; build tangents */
static void make_frame(double* f) {
  double n = norm3(f);
  for (int i = 0; i < 3; i++) f[i] /= n;
  f[3] = f[4] = f[5] = 0;
  if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  double d = dot3(f, f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] -= d * f[i];
  n = norm3(f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] /= n;
  cross3(f + 6, f, f + 3);
}
static void add_contact(bo_contact* con, int* n, double dist, const double* pos, const double* normal, int b1, int b2,
                        const cparam* cp) {
  if (*n >= MAXCON) return;
  bo_contact* c = con + *n;
  c->dist = dist;
  memcpy(c->pos, pos, sizeof c->pos);
  memcpy(c->frame, normal, 3 * sizeof(double));
  make_frame(c->frame);
  c->body1 = b1; c->body2 = b2;
  c->mu = cp->mu; c->margin = cp->margin;
  memcpy(c->solref, cp->solref, sizeof c->solref);
  memcpy(c->solimp, cp->solimp, sizeof c->solimp);
  (*n)++;
}

/* plane z = floor_z, normal +z, vs cylinder (restates MuJoCo's plane-cylinder primitive: up to 4 points:
 * lowest rim point of the near disc, the same on the far disc, two "triangle" points on the near disc) */
static void plane_cylinder(const model_t* m, const double* cpos, const double* cmat, int body, const cparam* cp,
                           bo_contact* con, int* n) {
  const double normal[3] = {0, 0, 1};
  double axis[3] = {cmat[2], cmat[5], cmat[8]};
  double prjaxis = dot3(normal, axis);
  if (prjaxis > 0) { for (int i = 0; i < 3; i++) axis[i] = -axis[i]; prjaxis = -prjaxis; }
  double dist0 = cpos[2] - m->floor_z;
  double vec[3] = {axis[0] * prjaxis - normal[0], axis[1] * prjaxis - normal[1], axis[2] * prjaxis - normal[2]};
  double len = norm3(vec);
  if (len >= MJ_MINVAL) { for (int i = 0; i < 3; i++) vec[i] *= m->wheel_r / len; }
  else { vec[0] = cmat[0] * m->wheel_r; vec[1] = cmat[3] * m->wheel_r; vec[2] = cmat[6] * m->wheel_r; }
  double prjvec = dot3(vec, normal);
  double ax[3] = {axis[0] * m->wheel_hl, axis[1] * m->wheel_hl, axis[2] * m->wheel_hl};
  prjaxis *= m->wheel_hl;
  double margin = cp->margin, pos[3], d;
  d = dist0 + prjaxis + prjvec;
  if (d > margin) return;
  for (int i = 0; i < 3; i++) pos[i] = cpos[i] + vec[i] + ax[i] - normal[i] * d * 0.5;
  add_contact(con, n, d, pos, normal, B_WORLD, body, cp);
  d = dist0 - prjaxis + prjvec;
  if (d <= margin) {
    for (int i = 0; i < 3; i++) pos[i] = cpos[i] + vec[i] - ax[i] - normal[i] * d * 0.5;
    add_contact(con, n, d, pos, normal, B_WORLD, body, cp);
  }
  double prjvec1 = -prjvec * 0.5;
  d = dist0 + prjaxis + prjvec1;
  if (d <= margin) {
    double vec1[3];
    cross3(vec1, vec, ax);
    double l1 = norm3(vec1);
    if (l1 >= MJ_MINVAL) for (int i = 0; i < 3; i++) vec1[i] *= m->wheel_r * sqrt(3.0) / 2 / l1;
    for (int i = 0; i < 3; i++) pos[i] = cpos[i] + vec1[i] + ax[i] - 0.5 * vec[i] - normal[i] * d * 0.5;
    add_contact(con, n, d, pos, normal, B_WORLD, body, cp);
    for (int i = 0; i < 3; i++) pos[i] = cpos[i] - vec1[i] + ax[i] - 0.5 * vec[i] - normal[i] * d * 0.5;
    add_contact(con, n, d, pos, normal, B_WORLD, body, cp);
  }
}

/* plane vs box (restates MuJoCo's plane-box primitive: corners below the centre with dist <= margin, max 4) */
static void plane_box(const model_t* m, const double* bpos, const double* bmat, const double* size, int body,
                      const cparam* cp, bo_contact* con, int* n) {
  const double normal[3] = {0, 0, 1};
  double dist = bpos[2] - m->floor_z;
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]}, corner[3];
    mulMatVec3(corner, bmat, vec);
    double ldist = dot3(normal, corner);
    if (dist + ldist > cp->margin || ldist > 0) continue;
    double d = dist + ldist, pos[3];
    for (int j = 0; j < 3; j++) pos[j] = corner[j] + bpos[j] - normal[j] * d * 0.5;
    add_contact(con, n, d, pos, normal, B_WORLD, body, cp);
    if (++cnt >= 4) return;
  }
}

/* ---------------------------------------------------------------------------------------------------------------
 * block <-> robot contact generation.  MuJoCo generates these with mjc_BoxBox (15-axis SAT, incident-face
 * clipping, <= 8 points) and its general convex collider (box <-> cylinder, one point); neither is reproducible verbatim
 * without the source, so this is the STANDARD clipped-polygon box-box (Gottschalk OBB SAT + Sutherland-Hodgman /
 * Liang-Barsky clipping as in ODE's dBoxBox) and a closest-feature box <-> capped-cylinder test, written from their
 * published descriptions.  PARITY WITH MUJOCO'S POINT SETS IS UNPINNED.  The HIP kernel (brs_core.hpp: collide_coupled)
 * follows the same specification: same axis choice, same candidate enumeration order, same reduction to 6 points.
 *
 * box_box_points works in the frame of box T (half sizes sT), box B is a cube of half size s centred at cg with axes =
 * columns of RTB.  Output: <= 6 contact points (T frame), their signed distances, the common normal T -> B.
 * code: 0..2 T face, 3..5 B face, 6 + 3 i + j edge(T axis i) x edge(B axis j), -1 no contact.
 * raw (optional): every valid candidate BEFORE the reduction to 4, as (x, y, z, dist) in the T frame.
 * ------------------------------------------------------------------------------------------------------------- */
#define BB_EDGE_REL 0.05
#define BB_EDGE_ABS 1e-5
#define BB_PAR_EPS 1e-6

/* incident quad V[4] = (u, v, g) clipped against the rectangle |u| <= a, |v| <= b; candidates in FIXED order:
 * 0..3 quad vertices inside, 4+2e / 5+2e entry / exit point of quad edge e, 12..15 rectangle corners inside the quad */
static void clip_quad_rect(double V[4][3], double a, double b, double margin, double cand[16][3], int valid[16]) {
  int inside[4];
  for (int i = 0; i < 16; i++) valid[i] = 0;
  for (int v = 0; v < 4; v++) {
    inside[v] = fabs(V[v][0]) <= a && fabs(V[v][1]) <= b;
    if (inside[v]) { memcpy(cand[v], V[v], 3 * sizeof(double)); valid[v] = 1; }
  }
  for (int e = 0; e < 4; e++) {
    const double* P0 = V[e]; const double* P1 = V[(e + 1) & 3];
    double du = P1[0] - P0[0], dv = P1[1] - P0[1], dg = P1[2] - P0[2];
    double t0 = 0, t1 = 1;
    int ok = 1;
    const double pp[4] = {-du, du, -dv, dv}, qq[4] = {P0[0] + a, a - P0[0], P0[1] + b, b - P0[1]};
    for (int c = 0; c < 4; c++) {
      if (pp[c] == 0) { if (qq[c] < 0) ok = 0; continue; }
      double r = qq[c] / pp[c];
      if (pp[c] < 0) { if (r > t0) t0 = r; } else { if (r < t1) t1 = r; }
    }
    if (!ok || !(t0 < t1)) continue;
    if (!inside[e]) { cand[4 + 2 * e][0] = P0[0] + t0 * du; cand[4 + 2 * e][1] = P0[1] + t0 * dv; cand[4 + 2 * e][2] = P0[2] + t0 * dg; valid[4 + 2 * e] = 1; }
    if (!inside[(e + 1) & 3]) { cand[5 + 2 * e][0] = P0[0] + t1 * du; cand[5 + 2 * e][1] = P0[1] + t1 * dv; cand[5 + 2 * e][2] = P0[2] + t1 * dg; valid[5 + 2 * e] = 1; }
  }
  /* rectangle corners inside the quad: corner = V0 + al (V1 - V0) + be (V3 - V0), 0 < al, be < 1 (the quad is a parallelogram) */
  double e1u = V[1][0] - V[0][0], e1v = V[1][1] - V[0][1], e2u = V[3][0] - V[0][0], e2v = V[3][1] - V[0][1];
  double det = e1u * e2v - e1v * e2u;
  if (fabs(det) > 1e-30) {
    double idet = 1.0 / det;
    for (int c = 0; c < 4; c++) {
      double cu = (c & 1) ? a : -a, cv = (c & 2) ? b : -b, ru = cu - V[0][0], rv = cv - V[0][1];
      double al = (ru * e2v - rv * e2u) * idet, be = (e1u * rv - e1v * ru) * idet;
      if (al > 0 && al < 1 && be > 0 && be < 1) {
        cand[12 + c][0] = cu; cand[12 + c][1] = cv;
        cand[12 + c][2] = V[0][2] + al * (V[1][2] - V[0][2]) + be * (V[3][2] - V[0][2]);
        valid[12 + c] = 1;
      }
    }
  }
  for (int i = 0; i < 16; i++) if (valid[i] && !(cand[i][2] < margin)) valid[i] = 0;
}

/* study switch (tests only, never set by the product's checkers): keep ALL clipped points (<= 8, as MuJoCo's mjc_BoxBox
 * does) instead of the 4 deepest -- used to measure what the kernel's 4-slot patch budget costs in fidelity */
static int g_boxbox_max = 6; /* the specification: the kernel's patch budget */
void bo_set_boxbox_keep_all(int on) { g_boxbox_max = on ? 8 : 6; }
void bo_set_boxbox_max(int n) { g_boxbox_max = n < 1 ? 1 : (n > 8 ? 8 : n); } /* study switch: keep the n deepest */

int bo_box_box_points(const double* sT, double s, const double* cg, const double* RTB, double margin, double* pos /*[4][3] ([8][3] with keep_all)*/,
                      double* dist /*[4] ([8])*/, double* nrm /*[3]*/, int* code, double* raw /*[16][4] or NULL*/, int* nraw) {
  *code = -1;
  if (nraw) *nraw = 0;
  /* --- separating axes: 6 faces, 9 edge pairs; keep the axis of LARGEST separation (least penetration) */
  double bestF = -1e300, bestE = -1e300;
  int axF = -1, axE = -1;
  for (int k = 0; k < 3; k++) {
    double sep = fabs(cg[k]) - sT[k] - s * (fabs(RTB[3 * k]) + fabs(RTB[3 * k + 1]) + fabs(RTB[3 * k + 2]));
    if (sep > margin) return 0;
    if (sep > bestF) { bestF = sep; axF = k; }
  }
  for (int j = 0; j < 3; j++) {
    double dB = cg[0] * RTB[j] + cg[1] * RTB[3 + j] + cg[2] * RTB[6 + j];
    double sep = fabs(dB) - s - (sT[0] * fabs(RTB[j]) + sT[1] * fabs(RTB[3 + j]) + sT[2] * fabs(RTB[6 + j]));
    if (sep > margin) return 0;
    if (sep > bestF) { bestF = sep; axF = 3 + j; }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double len2 = 1.0 - RTB[3 * i + j] * RTB[3 * i + j];
      if (len2 < BB_PAR_EPS) continue;
      /* L = e_i x b_j ; L[i1] = -b_j[i2], L[i2] = b_j[i1] */
      double cl = cg[i2] * RTB[3 * i1 + j] - cg[i1] * RTB[3 * i2 + j];
      double rT = sT[i1] * fabs(RTB[3 * i2 + j]) + sT[i2] * fabs(RTB[3 * i1 + j]);
      double rB = s * (fabs(RTB[3 * i + j1]) + fabs(RTB[3 * i + j2]));
      double sep = (fabs(cl) - rT - rB) / sqrt(len2);
      if (sep > margin) return 0;
      if (sep > bestE) { bestE = sep; axE = 3 * i + j; }
    }
  int use_edge = axE >= 0 && bestE > bestF + BB_EDGE_REL * fabs(bestF) + BB_EDGE_ABS;
  if (use_edge) {
    int i = axE / 3, j = axE % 3, i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    double bj[3] = {RTB[j], RTB[3 + j], RTB[6 + j]}, L[3] = {0, 0, 0};
    L[i1] = -bj[i2]; L[i2] = bj[i1];
    double il = 1.0 / sqrt(1.0 - bj[i] * bj[i]);
    for (int m = 0; m < 3; m++) L[m] *= il;
    if (dot3(L, cg) < 0) for (int m = 0; m < 3; m++) L[m] = -L[m];
    /* supporting edges: on T the edge along e_i extremal in +L, on B the edge along b_j extremal in -L */
    double pA[3], pB[3] = {cg[0], cg[1], cg[2]};
    for (int m = 0; m < 3; m++) pA[m] = m == i ? 0.0 : (L[m] >= 0 ? sT[m] : -sT[m]);
    for (int m = 0; m < 3; m++) {
      if (m == j) continue;
      double bm[3] = {RTB[m], RTB[3 + m], RTB[6 + m]};
      double sg = dot3(L, bm) >= 0 ? -s : s;
      for (int c = 0; c < 3; c++) pB[c] += sg * bm[c];
    }
    /* closest points of the lines pA + al e_i and pB + be b_j, clamped to the edge extents */
    double w[3] = {pA[0] - pB[0], pA[1] - pB[1], pA[2] - pB[2]};
    double bdot = bj[i], dA = w[i], dBv = dot3(w, bj), den = 1.0 - bdot * bdot;
    double al = (bdot * dBv - dA) / den, be = (dBv - bdot * dA) / den;
    al = fmax(-sT[i], fmin(sT[i], al)); be = fmax(-s, fmin(s, be));
    double qA[3] = {pA[0], pA[1], pA[2]}, qB[3];
    qA[i] += al;
    for (int c = 0; c < 3; c++) qB[c] = pB[c] + be * bj[c];
    if (!(bestE < margin)) return 0;
    for (int c = 0; c < 3; c++) { pos[c] = 0.5 * (qA[c] + qB[c]); nrm[c] = L[c]; }
    dist[0] = bestE;
    *code = 6 + axE;
    if (raw) { raw[0] = pos[0]; raw[1] = pos[1]; raw[2] = pos[2]; raw[3] = bestE; *nraw = 1; }
    return 1;
  }
  /* --- face contact: incident face of the other box clipped against the reference face */
  double V[4][3], cand[16][3], a, b;
  int valid[16];
  double P3[16][3]; /* candidate contact positions, T frame */
  if (axF < 3) {
    int k = axF, j1 = (k + 1) % 3, j2 = (k + 2) % 3;
    double sg = cg[k] >= 0 ? 1.0 : -1.0;
    nrm[0] = nrm[1] = nrm[2] = 0; nrm[k] = sg;
    int js = 0;
    for (int j = 1; j < 3; j++) if (fabs(RTB[3 * k + j]) > fabs(RTB[3 * k + js])) js = j;
    double sj = -sg * (RTB[3 * k + js] >= 0 ? 1.0 : -1.0);
    int a1 = (js + 1) % 3, a2 = (js + 2) % 3;
    static const int su[4] = {-1, 1, 1, -1}, sv[4] = {-1, -1, 1, 1}; /* around the face */
    for (int v = 0; v < 4; v++) {
      double loc[3], p[3];
      loc[js] = sj * s; loc[a1] = su[v] * s; loc[a2] = sv[v] * s;
      mulMatVec3(p, RTB, loc);
      for (int c = 0; c < 3; c++) p[c] += cg[c];
      V[v][0] = p[j1]; V[v][1] = p[j2]; V[v][2] = sg * p[k] - sT[k];
    }
    a = sT[j1]; b = sT[j2];
    clip_quad_rect(V, a, b, margin, cand, valid);
    for (int c = 0; c < 16; c++) {
      P3[c][j1] = cand[c][0]; P3[c][j2] = cand[c][1]; P3[c][k] = sg * (sT[k] + 0.5 * cand[c][2]);
    }
  } else {
    int j = axF - 3, i1 = (j + 1) % 3, i2 = (j + 2) % 3;
    double bj[3] = {RTB[j], RTB[3 + j], RTB[6 + j]};
    double sgB = dot3(cg, bj) >= 0 ? 1.0 : -1.0;
    for (int c = 0; c < 3; c++) nrm[c] = sgB * bj[c];
    int ks = 0;
    for (int k = 1; k < 3; k++) if (fabs(bj[k]) > fabs(bj[ks])) ks = k;
    double sk = sgB * (bj[ks] >= 0 ? 1.0 : -1.0);
    int a1 = (ks + 1) % 3, a2 = (ks + 2) % 3;
    static const int su[4] = {-1, 1, 1, -1}, sv[4] = {-1, -1, 1, 1};
    for (int v = 0; v < 4; v++) {
      double loc[3], rel[3], pB[3];
      loc[ks] = sk * sT[ks]; loc[a1] = su[v] * sT[a1]; loc[a2] = sv[v] * sT[a2];
      for (int c = 0; c < 3; c++) rel[c] = loc[c] - cg[c];
      mulMatTVec3(pB, RTB, rel);
      V[v][0] = pB[i1]; V[v][1] = pB[i2]; V[v][2] = -sgB * pB[j] - s;
    }
    a = s; b = s;
    clip_quad_rect(V, a, b, margin, cand, valid);
    for (int c = 0; c < 16; c++) {
      double q[3], back[3];
      q[i1] = cand[c][0]; q[i2] = cand[c][1]; q[j] = -sgB * (s + 0.5 * cand[c][2]);
      mulMatVec3(back, RTB, q);
      for (int m = 0; m < 3; m++) P3[c][m] = back[m] + cg[m];
    }
  }
  int cnt = 0;
  for (int c = 0; c < 16; c++) cnt += valid[c];
  if (raw) {
    int r = 0;
    for (int c = 0; c < 16; c++) if (valid[c]) { raw[4 * r] = P3[c][0]; raw[4 * r + 1] = P3[c][1]; raw[4 * r + 2] = P3[c][2]; raw[4 * r + 3] = cand[c][2]; r++; }
    *nraw = r;
  }
  if (cnt == 0) return 0;
  *code = axF;
  if (cnt > g_boxbox_max) { /* reduction: keep the deepest (ties: lower candidate index) */
    int keep[16] = {0};
    for (int pass = 0; pass < g_boxbox_max; pass++) {
      double bd = 1e300; int bi = -1;
      for (int c = 0; c < 16; c++) if (valid[c] && !keep[c] && cand[c][2] < bd) { bd = cand[c][2]; bi = c; }
      keep[bi] = 1;
    }
    for (int c = 0; c < 16; c++) valid[c] = valid[c] && keep[c];
  }
  int n = 0;
  for (int c = 0; c < 16; c++) if (valid[c]) { memcpy(pos + 3 * n, P3[c], 3 * sizeof(double)); dist[n] = cand[c][2]; n++; }
  return n;
}

static void box_box(const model_t* m, const double* tpos, const double* tmat, const double* bpos, const double* bmat,
                    const cparam* cp, bo_contact* con, int* n) {
  double dw[3] = {bpos[0] - tpos[0], bpos[1] - tpos[1], bpos[2] - tpos[2]}, cg[3], RTB[9];
  double rt = norm3(m->torso_size), rb = norm3(m->block_size);
  if (norm3(dw) > rt + rb + cp->margin) return;
  mulMatTVec3(cg, tmat, dw);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RTB[3 * i + j] = tmat[i] * bmat[j] + tmat[3 + i] * bmat[3 + j] + tmat[6 + i] * bmat[6 + j];
  double pos[24], dist[8], nT[3];
  int code, cnt = bo_box_box_points(m->torso_size, m->block_size[0], cg, RTB, cp->margin, pos, dist, nT, &code, NULL, NULL);
  for (int c = 0; c < cnt; c++) {
    double pw[3], nw[3];
    mulMatVec3(pw, tmat, pos + 3 * c);
    mulMatVec3(nw, tmat, nT);
    for (int i = 0; i < 3; i++) pw[i] += tpos[i];
    add_contact(con, n, dist[c], pw, nw, B_TORSO, B_BLOCK, cp);
  }
}

/* signed distance of point p (cylinder frame: axis x, centre at the origin) to a capped cylinder, outward normal */
static double point_cyl(const double* p, double r, double hl, double* nrm, int* okp) {
  double rho = sqrt(p[1] * p[1] + p[2] * p[2]), drad = rho - r, dax = fabs(p[0]) - hl, sx = p[0] >= 0 ? 1.0 : -1.0;
  *okp = 1;
  if (drad > 0 && dax > 0) { /* rim region: Euclidean distance to the rim circle */
    double dd = sqrt(drad * drad + dax * dax);
    nrm[0] = sx * dax / dd; nrm[1] = drad * p[1] / (rho * dd); nrm[2] = drad * p[2] / (rho * dd);
    return dd;
  }
  if (drad >= dax) {
    if (rho < 1e-9) { *okp = 0; return 0; }
    nrm[0] = 0; nrm[1] = p[1] / rho; nrm[2] = p[2] / rho;
    return drad;
  }
  nrm[0] = sx; nrm[1] = 0; nrm[2] = 0;
  return dax;
}
/* signed distance of point p (box frame) to a box (Euclidean outside, max-axis inside), outward normal */
static double point_box(const double* p, const double* s, double* nrm) {
  double q[3] = {fabs(p[0]) - s[0], fabs(p[1]) - s[1], fabs(p[2]) - s[2]};
  if (q[0] > 0 || q[1] > 0 || q[2] > 0) {
    double mv[3], dd;
    for (int k = 0; k < 3; k++) mv[k] = q[k] > 0 ? (p[k] >= 0 ? q[k] : -q[k]) : 0.0;
    dd = norm3(mv);
    for (int k = 0; k < 3; k++) nrm[k] = mv[k] / dd;
    return dd;
  }
  int ax = 0;
  if (q[1] > q[ax]) ax = 1;
  if (q[2] > q[ax]) ax = 2;
  nrm[0] = nrm[1] = nrm[2] = 0; nrm[ax] = p[ax] >= 0 ? 1.0 : -1.0;
  return q[ax];
}

/* wheel (capped cylinder, axis = torso x) <-> block: ONE point, the deepest of these closest-feature candidates, in this
 * order (a later candidate replaces an earlier one only if strictly deeper):
 *   (a) the 8 block vertices and (b) on each of the 12 block edges the point nearest the cylinder axis (if interior to the
 *       edge), each evaluated as a POINT against the cylinder (exact point-cylinder distance; points on the axis skipped);
 *       the normal is the cylinder's outward normal at the winning point;
 *   (c) two cylinder surface points nearest the block centre (barrel point, rim point) against the box.
 * Worked in the torso frame (as the HIP kernel does); d = block centre - wheel centre, RTB = block axes as columns. */
int bo_box_cyl_point(const double* d, const double* RTB, double s, double r, double hl, double margin, double* pos, double* nrm,
                     double* dist_out) {
  double best = margin, wp[3] = {0, 0, 0}, bp[3] = {0, 0, 0}, bn[3] = {0, 0, 1};
  int found = 0, from_c = 0;
  for (int i = 0; i < 8; i++) {
    double loc[3] = {(i & 1) ? s : -s, (i & 2) ? s : -s, (i & 4) ? s : -s}, v[3], nn[3];
    mulMatVec3(v, RTB, loc);
    for (int j = 0; j < 3; j++) v[j] += d[j];
    int ok;
    double dist = point_cyl(v, r, hl, nn, &ok);
    if (v[1] * v[1] + v[2] * v[2] < 1e-18) ok = 0;
    if (ok && dist < best) { best = dist; found = 1; memcpy(wp, v, sizeof wp); }
  }
  for (int j = 0; j < 3; j++) {
    int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    double dir[3] = {RTB[j], RTB[3 + j], RTB[6 + j]};
    double dd = dir[1] * dir[1] + dir[2] * dir[2];
    if (dd < 1e-8) continue;
    for (int e = 0; e < 4; e++) {
      double loc[3], o[3], nn[3];
      loc[j] = 0; loc[j1] = (e & 1) ? s : -s; loc[j2] = (e & 2) ? s : -s;
      mulMatVec3(o, RTB, loc);
      for (int c = 0; c < 3; c++) o[c] += d[c];
      double tau = -(o[1] * dir[1] + o[2] * dir[2]) / dd;
      if (!(fabs(tau) < s)) continue;
      double p[3] = {o[0] + tau * dir[0], o[1] + tau * dir[1], o[2] + tau * dir[2]};
      int ok;
      double dist = point_cyl(p, r, hl, nn, &ok);
      if (p[1] * p[1] + p[2] * p[2] < 1e-18) ok = 0;
      if (ok && dist < best) { best = dist; found = 1; memcpy(wp, p, sizeof wp); }
    }
  }
  if (found) {
    int ok;
    point_cyl(wp, r, hl, bn, &ok);
    for (int c = 0; c < 3; c++) bp[c] = wp[c] - bn[c] * best * 0.5;
  }
  double rho = sqrt(d[1] * d[1] + d[2] * d[2]);
  if (rho > 1e-9) {
    const double ss[3] = {s, s, s};
    for (int cand = 0; cand < 2; cand++) {
      double xc = d[0] > hl ? hl : (d[0] < -hl ? -hl : d[0]), rr = r;
      if (cand == 1) { xc = d[0] >= 0 ? hl : -hl; rr = rho < r ? rho : r; }
      double q[3] = {xc, rr * d[1] / rho, rr * d[2] / rho};
      double rel[3] = {q[0] - d[0], q[1] - d[1], q[2] - d[2]}, p[3], nb[3], nw[3];
      mulMatTVec3(p, RTB, rel);
      double dist = point_box(p, ss, nb);
      if (dist < best) {
        best = dist; found = 1; from_c = 1;
        mulMatVec3(nw, RTB, nb);
        for (int c = 0; c < 3; c++) { bn[c] = -nw[c]; bp[c] = q[c] + bn[c] * dist * 0.5; }
      }
    }
  }
  (void)from_c;
  if (!found) return 0;
  for (int c = 0; c < 3; c++) { pos[c] = bp[c]; nrm[c] = bn[c]; }
  *dist_out = best;
  return 1;
}

static void box_cyl(const model_t* m, const double* wpos, const double* tmat, int wbody, const double* bpos,
                    const double* bmat, const cparam* cp, bo_contact* con, int* n) {
  double dw[3] = {bpos[0] - wpos[0], bpos[1] - wpos[1], bpos[2] - wpos[2]}, d[3], RTB[9];
  double rw = sqrt(m->wheel_r * m->wheel_r + m->wheel_hl * m->wheel_hl), rb = norm3(m->block_size);
  if (norm3(dw) > rw + rb + cp->margin) return;
  mulMatTVec3(d, tmat, dw);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RTB[3 * i + j] = tmat[i] * bmat[j] + tmat[3 + i] * bmat[3 + j] + tmat[6 + i] * bmat[6 + j];
  double pos[3], nrm[3], dist, pw[3], nw[3];
  if (!bo_box_cyl_point(d, RTB, m->block_size[0], m->wheel_r, m->wheel_hl, cp->margin, pos, nrm, &dist)) return;
  mulMatVec3(pw, tmat, pos);
  mulMatVec3(nw, tmat, nrm);
  for (int i = 0; i < 3; i++) pw[i] += wpos[i];
  add_contact(con, n, dist, pw, nw, wbody, B_BLOCK, cp);
}

static int collide(const model_t* m, const kin_t* k, bo_contact* con, double muw) {
  int n = 0;
  cparam cpw = m->cp_wheel_floor;
  if (muw > 0) cpw.mu = muw; /* max(floor, wheel) friction, both set to the episode's value */
  double gpos[3], t[3], gmat[9];
  /* floor <-> wheels (Env01: explicit <pair>, envs/env01_v1.xml:30-33; Env03: geom defaults) */
  for (int w = 0; w < 2; w++) {
    mulMat3(gmat, k->xmat[B_LW + w], m->wheel_gmat);
    plane_cylinder(m, k->xpos[B_LW + w], gmat, B_LW + w, &cpw, con, &n);
  }
  /* floor <-> torso box (dynamic pair, geom defaults) */
  mulMatVec3(t, k->xmat[B_TORSO], m->torso_gpos);
  for (int i = 0; i < 3; i++) gpos[i] = k->xpos[B_TORSO][i] + t[i];
  plane_box(m, gpos, k->xmat[B_TORSO], m->torso_size, B_TORSO, &m->cp_torso_floor, con, &n);
  if (m->has_block) {
    plane_box(m, k->xpos[B_BLOCK], k->xmat[B_BLOCK], m->block_size, B_BLOCK, &m->cp_block, con, &n);
    box_box(m, gpos, k->xmat[B_TORSO], k->xpos[B_BLOCK], k->xmat[B_BLOCK], &m->cp_block, con, &n);
    for (int w = 0; w < 2; w++) /* cylinder axis = torso body x: worked in the torso frame */
      box_cyl(m, k->xpos[B_LW + w], k->xmat[B_TORSO], B_LW + w, k->xpos[B_BLOCK], k->xmat[B_BLOCK], &m->cp_block, con, &n);
  }
  return n;
}

/* ---------------- forward dynamics ---------------- */
typedef struct {
  kin_t k;
  double M[NVMAX * NVMAX], LM[NVMAX * NVMAX];
  double bias[NVMAX], passive[NVMAX], actuator[NVMAX], smooth[NVMAX], qacc_smooth[NVMAX];
  int ncon, nefc;
  bo_contact con[MAXCON];
  double J[MAXEFC][NVMAX], D[MAXEFC], aref[MAXEFC], force[MAXEFC];
  double qacc[NVMAX], qfrc_constraint[NVMAX], qacc_integ[NVMAX];
  int act_clamped[2];
  int solver_iter;
} fwd_t;

/* MuJoCo impedance sigmoid d(r) from solimp = (d0, dwidth, width, midpoint, power) */
static double impedance(const double* solimp, double pos, double margin) {
  if (solimp[0] == solimp[1] || solimp[2] <= MJ_MINVAL) return 0.5 * (solimp[0] + solimp[1]);
  double x = (pos - margin) / solimp[2];
  if (x < 0) x = -x;
  if (x >= 1) return solimp[1];
  if (x <= 0) return solimp[0];
  double y, mid = solimp[3], pw = solimp[4];
  if (pw == 1) y = x;
  else if (x <= mid) y = pow(x, pw) / pow(mid, pw - 1);
  else y = 1 - pow(1 - x, pw) / pow(1 - mid, pw - 1);
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static void make_constraints(const model_t* m, const double* qvel, fwd_t* f) {
  int nv = m->nv;
  f->nefc = 0;
  for (int c = 0; c < f->ncon; c++) {
    const bo_contact* cn = f->con + c;
    if (!(cn->dist < cn->margin)) continue; /* includemargin = margin - gap, gap = 0 */
    double J1p[3][NVMAX], J1r[3][NVMAX], J2p[3][NVMAX], J2r[3][NVMAX], Jc[3][NVMAX];
    jac_point(m, &f->k, cn->body1, cn->pos, J1p, J1r);
    jac_point(m, &f->k, cn->body2, cn->pos, J2p, J2r);
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < nv; j++) {
        double s = 0;
        for (int i = 0; i < 3; i++) s += cn->frame[3 * a + i] * (J2p[i][j] - J1p[i][j]);
        Jc[a][j] = s;
      }
    /* stiffness / damping from solref (timeconst, dampratio), refsafe: timeconst >= 2h */
    double tc = cn->solref[0] > 2 * m->h ? cn->solref[0] : 2 * m->h, dr = cn->solref[1], dmax = cn->solimp[1];
    double K = 1.0 / fmax(MJ_MINVAL, dmax * dmax * tc * tc * dr * dr), B = 2.0 / fmax(MJ_MINVAL, dmax * tc);
    double imp = impedance(cn->solimp, cn->dist, cn->margin);
    double tran = m->invweight0[cn->body1][0] + m->invweight0[cn->body2][0];
    double diagApprox = tran + cn->mu * cn->mu * tran;
    double R0 = fmax(MJ_MINVAL, (1 - imp) * diagApprox / imp);
    double Rpy = 2 * cn->mu * cn->mu * R0; /* pyramidal: every edge gets 2 mu^2 R */
    for (int e = 0; e < 4; e++) { /* rows n + mu t1, n - mu t1, n + mu t2, n - mu t2 */
      int r = f->nefc++;
      int t = 1 + e / 2;
      double sg = (e & 1) ? -1.0 : 1.0, vel = 0;
      for (int j = 0; j < nv; j++) {
        f->J[r][j] = Jc[0][j] + sg * cn->mu * Jc[t][j];
        vel += f->J[r][j] * qvel[j];
      }
      f->D[r] = 1.0 / Rpy;
      f->aref[r] = -B * vel - K * imp * (cn->dist - cn->margin);
    }
  }
}

/* cost of the convex problem at qacc (Gauss + constraint), optionally forces */
typedef struct {
  const model_t* m;
  fwd_t* f;
  int nv, nefc;
  double Ma[NVMAX], jar[MAXEFC], Mv[NVMAX], Jv[MAXEFC], search[NVMAX], grad[NVMAX], Mgrad[NVMAX];
  int active[MAXEFC];
  double cost;
  double quadGauss[3];
} ctx_t;

static void mulM(const fwd_t* f, int nv, const double* x, double* y) {
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += f->M[i * NVMAX + j] * x[j];
    y[i] = s;
  }
}
static int update_constraint(ctx_t* c) { /* returns 1 if the active set changed */
  fwd_t* f = c->f;
  int changed = 0;
  double cost = 0;
  for (int j = 0; j < c->nv; j++) f->qfrc_constraint[j] = 0;
  for (int i = 0; i < c->nefc; i++) {
    int act = c->jar[i] < 0;
    if (act != c->active[i]) changed = 1;
    c->active[i] = act;
    f->force[i] = act ? -f->D[i] * c->jar[i] : 0;
    if (act) {
      cost += 0.5 * f->D[i] * c->jar[i] * c->jar[i];
      for (int j = 0; j < c->nv; j++) f->qfrc_constraint[j] += f->J[i][j] * f->force[i];
    }
  }
  for (int j = 0; j < c->nv; j++) cost += 0.5 * (c->Ma[j] - f->smooth[j]) * (f->qacc[j] - f->qacc_smooth[j]);
  c->cost = cost;
  return changed;
}
static void newton_direction(ctx_t* c) {
  fwd_t* f = c->f;
  int nv = c->nv;
  double H[NVMAX * NVMAX], L[NVMAX * NVMAX];
  memcpy(H, f->M, sizeof H);
  for (int i = 0; i < c->nefc; i++)
    if (c->active[i])
      for (int a = 0; a < nv; a++)
        for (int b = 0; b < nv; b++) H[a * NVMAX + b] += f->D[i] * f->J[i][a] * f->J[i][b];
  chol(H, nv, NVMAX, L);
  for (int j = 0; j < nv; j++) c->grad[j] = c->Ma[j] - f->smooth[j] - f->qfrc_constraint[j];
  chol_solve(L, nv, NVMAX, c->grad, c->Mgrad);
  for (int j = 0; j < nv; j++) c->search[j] = -c->Mgrad[j];
}
typedef struct { double alpha, cost, d0, d1; } lspt;
static void ls_eval(const ctx_t* c, lspt* p) {
  const fwd_t* f = c->f;
  double a = p->alpha, q0 = c->quadGauss[0], q1 = c->quadGauss[1], q2 = c->quadGauss[2];
  for (int i = 0; i < c->nefc; i++) {
    double x = c->jar[i] + a * c->Jv[i];
    if (x < 0) {
      q0 += 0.5 * f->D[i] * c->jar[i] * c->jar[i];
      q1 += f->D[i] * c->jar[i] * c->Jv[i];
      q2 += 0.5 * f->D[i] * c->Jv[i] * c->Jv[i];
    }
  }
  p->cost = a * a * q2 + a * q1 + q0;
  p->d0 = 2 * a * q2 + q1;
  p->d1 = 2 * q2;
  if (p->d1 <= 0) p->d1 = MJ_MINVAL;
}
/* exact 1-D minimisation of the piecewise-quadratic cost along c->search (Newton on alpha with a
 * bracketing fallback), in the manner of MuJoCo's primal line search */
static double linesearch(ctx_t* c) {
  const model_t* m = c->m;
  fwd_t* f = c->f;
  int nv = c->nv;
  double snorm = 0;
  for (int j = 0; j < nv; j++) snorm += c->search[j] * c->search[j];
  snorm = sqrt(snorm);
  if (snorm < MJ_MINVAL) return 0;
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  double gtol = m->tolerance * m->ls_tolerance * snorm / scale;
  mulM(f, nv, c->search, c->Mv);
  for (int i = 0; i < c->nefc; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += f->J[i][j] * c->search[j];
    c->Jv[i] = s;
  }
  c->quadGauss[0] = c->quadGauss[1] = c->quadGauss[2] = 0;
  for (int j = 0; j < nv; j++) {
    c->quadGauss[0] += 0.5 * (c->Ma[j] - f->smooth[j]) * (f->qacc[j] - f->qacc_smooth[j]);
    c->quadGauss[1] += c->search[j] * (c->Ma[j] - f->smooth[j]);
    c->quadGauss[2] += 0.5 * c->search[j] * c->Mv[j];
  }
  lspt p0 = {0, 0, 0, 0}, p1, p2;
  ls_eval(c, &p0);
  p1.alpha = p0.alpha - p0.d0 / p0.d1;
  ls_eval(c, &p1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  int dir = p1.d0 < 0 ? 1 : -1, it = 0, p2set = 0;
  p2 = p1;
  while (p1.d0 * dir <= -gtol && it < m->ls_iterations) {
    p2 = p1; p2set = 1;
    p1.alpha -= p1.d0 / p1.d1;
    ls_eval(c, &p1);
    it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= m->ls_iterations || !p2set) return p1.alpha;
  /* bracketed between p2 (derivative sign = -dir) and p1 (derivative sign = +dir): bisection + Newton */
  while (it < m->ls_iterations) {
    lspt pm, pn;
    pm.alpha = 0.5 * (p1.alpha + p2.alpha);
    ls_eval(c, &pm);
    it++;
    if (fabs(pm.d0) < gtol) return pm.alpha;
    pn.alpha = pm.alpha - pm.d0 / pm.d1;
    double lo = p1.alpha < p2.alpha ? p1.alpha : p2.alpha, hi = p1.alpha < p2.alpha ? p2.alpha : p1.alpha;
    if (pn.alpha > lo && pn.alpha < hi) {
      ls_eval(c, &pn);
      it++;
      if (fabs(pn.d0) < gtol) return pn.alpha;
      if (pn.d0 * dir > 0) p1 = pn; else p2 = pn;
    }
    if (pm.d0 * dir > 0) { if ((pm.alpha - p2.alpha) * (p1.alpha - p2.alpha) > 0 && fabs(pm.alpha - p2.alpha) < fabs(p1.alpha - p2.alpha)) p1 = pm; }
    else { if (fabs(pm.alpha - p1.alpha) < fabs(p2.alpha - p1.alpha)) p2 = pm; }
    if (fabs(p1.alpha - p2.alpha) < 1e-300) break;
  }
  return p1.cost < p2.cost ? p1.alpha : p2.alpha;
}

static double total_cost(ctx_t* c, const double* qacc) { /* cost at an arbitrary qacc, no side effects */
  fwd_t* f = c->f;
  double Ma[NVMAX], cost = 0;
  mulM(f, c->nv, qacc, Ma);
  for (int i = 0; i < c->nefc; i++) {
    double s = -f->aref[i];
    for (int j = 0; j < c->nv; j++) s += f->J[i][j] * qacc[j];
    if (s < 0) cost += 0.5 * f->D[i] * s * s;
  }
  for (int j = 0; j < c->nv; j++) cost += 0.5 * (Ma[j] - f->smooth[j]) * (qacc[j] - f->qacc_smooth[j]);
  return cost;
}

static void solve_newton(const model_t* m, fwd_t* f, const double* warm) {
  ctx_t c;
  int nv = m->nv;
  c.m = m; c.f = f; c.nv = nv; c.nefc = f->nefc;
  /* warm start: the better of qacc_warmstart and qacc_smooth */
  double cw = total_cost(&c, warm), cs = total_cost(&c, f->qacc_smooth);
  memcpy(f->qacc, cw > cs ? f->qacc_smooth : warm, nv * sizeof(double));
  mulM(f, nv, f->qacc, c.Ma);
  for (int i = 0; i < c.nefc; i++) {
    double s = -f->aref[i];
    for (int j = 0; j < nv; j++) s += f->J[i][j] * f->qacc[j];
    c.jar[i] = s;
    c.active[i] = -1;
  }
  update_constraint(&c);
  newton_direction(&c);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < m->iterations) {
    double alpha = linesearch(&c);
    if (alpha == 0) break;
    for (int j = 0; j < nv; j++) { f->qacc[j] += alpha * c.search[j]; c.Ma[j] += alpha * c.Mv[j]; }
    for (int i = 0; i < c.nefc; i++) c.jar[i] += alpha * c.Jv[i];
    double oldcost = c.cost;
    update_constraint(&c);
    newton_direction(&c);
    double g = 0;
    for (int j = 0; j < nv; j++) g += c.grad[j] * c.grad[j];
    iter++;
    if (scale * (oldcost - c.cost) < m->tolerance || scale * sqrt(g) < m->tolerance) break;
  }
  f->solver_iter = iter;
}

static void forward(const model_t* m, const double* qpos, const double* qvel, const double* ctrl, const double* warm,
                    fwd_t* f, double muw) {
  int nv = m->nv;
  kinematics(m, qpos, &f->k);
  mass_matrix(m, &f->k, f->M);
  chol(f->M, nv, NVMAX, f->LM);
  f->ncon = collide(m, &f->k, f->con, muw);
  make_constraints(m, qvel, f);
  for (int j = 0; j < nv; j++) f->passive[j] = f->actuator[j] = 0;
  for (int w = 0; w < 2; w++) f->passive[6 + w] = -m->damping * qvel[6 + w]; /* envs/robot-02.xml:11,16 */
  bias_forces(m, &f->k, qvel, f->bias);
  for (int w = 0; w < 2; w++) { /* velocity servo, envs/robot-02.xml:22-25 */
    double u = ctrl[w] > m->ctrlrange ? m->ctrlrange : (ctrl[w] < -m->ctrlrange ? -m->ctrlrange : ctrl[w]);
    double force = m->kv * u - m->kv * qvel[6 + w];
    f->act_clamped[w] = 0;
    if (force >= m->forcerange) { force = m->forcerange; f->act_clamped[w] = 1; }
    if (force <= -m->forcerange) { force = -m->forcerange; f->act_clamped[w] = 1; }
    f->actuator[6 + w] = force;
  }
  for (int j = 0; j < nv; j++) f->smooth[j] = f->passive[j] - f->bias[j] + f->actuator[j];
  chol_solve(f->LM, nv, NVMAX, f->smooth, f->qacc_smooth);
  f->solver_iter = 0;
  if (f->nefc == 0) {
    memcpy(f->qacc, f->qacc_smooth, nv * sizeof(double));
    for (int j = 0; j < nv; j++) f->qfrc_constraint[j] = 0;
  } else
    solve_newton(m, f, warm);
  /* implicitfast: (M - h dF/dv) a = smooth + constraint; dF/dv = -damping - kv (kv dropped when the force is clamped) */
  double MM[NVMAX * NVMAX], L[NVMAX * NVMAX], rhs[NVMAX];
  memcpy(MM, f->M, sizeof MM);
  for (int w = 0; w < 2; w++) MM[(6 + w) * NVMAX + 6 + w] += m->h * (m->damping + (f->act_clamped[w] ? 0.0 : m->kv));
  for (int j = 0; j < nv; j++) rhs[j] = f->smooth[j] + f->qfrc_constraint[j];
  chol(MM, nv, NVMAX, L);
  chol_solve(L, nv, NVMAX, rhs, f->qacc_integ);
}

static int bad_number(double x) { return !(x == x) || x > 1e10 || x < -1e10; }

/* one mj_step: forward at (qpos,qvel) then semi-implicit advance; returns 0, or 1 if the state went bad */
static int substep(const model_t* m, env_t* e, const double* ctrl, fwd_t* f) {
  int nv = m->nv;
  forward(m, e->qpos, e->qvel, ctrl, e->warm, f, m->variant == BO_ENV02_V1 ? e->muw : 0.0);
  memcpy(e->warm, f->qacc, nv * sizeof(double));
  /* accessor pose = kinematics of THIS forward pass (lags the advanced qpos by one substep) */
  memcpy(e->xquat, f->k.tquat, sizeof e->xquat);
  memcpy(e->xpos, f->k.xpos[B_TORSO], sizeof e->xpos);
  for (int j = 0; j < nv; j++) {
    if (bad_number(f->qacc_integ[j])) return 1;
    e->qvel[j] += m->h * f->qacc_integ[j];
  }
  for (int i = 0; i < 3; i++) e->qpos[i] += m->h * e->qvel[i];
  quatIntegrate(e->qpos + 3, e->qvel + 3, m->h);
  e->qpos[7] += m->h * e->qvel[6];
  e->qpos[8] += m->h * e->qvel[7];
  if (m->has_block) {
    for (int i = 0; i < 3; i++) e->qpos[9 + i] += m->h * e->qvel[8 + i];
    quatIntegrate(e->qpos + 12, e->qvel + 11, m->h);
  }
  e->time += m->h;
  for (int j = 0; j < m->nq; j++)
    if (bad_number(e->qpos[j])) return 1;
  return 0;
}

/* ============================================================================================
 * model construction
 * ========================================================================================== */
static void model_init(model_t* m, int variant, uint32_t flags, int max_episode_steps, int substeps, double timestep) {
  memset(m, 0, sizeof *m);
  m->variant = variant;
  m->family = (variant == BO_ENV03_V1 || variant == BO_ENV03_V2) ? 3 : 1;
  m->has_block = m->family == 3;
  m->nq = m->has_block ? 16 : 9;
  m->nv = m->has_block ? 14 : 8;
  m->nbody = m->has_block ? 5 : 4;
  m->h = timestep > 0 ? timestep : 0.00002; /* envs/env01_v1.xml:3 */
  m->nsub = substeps > 0 ? substeps : 250;  /* envs/RobotBaseEnv.py:59 */
  m->gravity[2] = -9.81;
  m->tolerance = 1e-8; m->ls_tolerance = 0.01; m->iterations = 100; m->ls_iterations = 50;
  /* geoms (envs/robot-02.xml:6,12,17 ; envs/env03_v1.xml:34) */
  m->torso_size[0] = 0.05; m->torso_size[1] = 0.0185; m->torso_size[2] = 0.0855;
  m->torso_gpos[2] = 0.0995;
  m->wheel_r = 0.034; m->wheel_hl = 0.013;
  { double q[4] = {0.707107, 0, 0.707107, 0}; normalize4(q); quat2mat(m->wheel_gmat, q); }
  m->block_size[0] = m->block_size[1] = m->block_size[2] = 0.02;
  m->floor_z = -0.02; /* envs/env01_v1.xml:27 */
  m->wheel_pos[0][0] = -0.074; m->wheel_pos[0][2] = 0.034; m->wheel_axis[0][0] = -1;
  m->wheel_pos[1][0] = 0.074;  m->wheel_pos[1][2] = 0.034; m->wheel_axis[1][0] = 1;
  /* inertiafromgeom="true" (envs/env01_v1.xml:2): mass/inertia from the geoms at density 1000,
   * the <inertial> elements are overridden */
  double rho = 1000.0;
  {
    const double* s = m->torso_size;
    double ms = 8 * s[0] * s[1] * s[2] * rho;
    m->mass[B_TORSO] = ms;
    m->inertia[B_TORSO][0] = ms / 3 * (s[1] * s[1] + s[2] * s[2]);
    m->inertia[B_TORSO][1] = ms / 3 * (s[0] * s[0] + s[2] * s[2]);
    m->inertia[B_TORSO][2] = ms / 3 * (s[0] * s[0] + s[1] * s[1]);
    m->ipos[B_TORSO][2] = m->torso_gpos[2];
  }
  for (int w = 0; w < 2; w++) {
    double r = m->wheel_r, hl = m->wheel_hl, ms = PI * r * r * 2 * hl * rho;
    m->mass[B_LW + w] = ms;
    m->inertia[B_LW + w][0] = 0.5 * ms * r * r;                          /* axial: body x (geom z rotated onto x) */
    m->inertia[B_LW + w][1] = m->inertia[B_LW + w][2] = ms * (3 * r * r + 4 * hl * hl) / 12;
  }
  if (m->has_block) {
    double s = m->block_size[0], ms = 8 * s * s * s * rho;
    m->mass[B_BLOCK] = ms;
    for (int i = 0; i < 3; i++) m->inertia[B_BLOCK][i] = ms / 3 * 2 * s * s;
  }
  m->kv = 4.0; m->ctrlrange = 78.54; m->forcerange = 0.65; m->damping = 0.01;
  /* contact parameters */
  cparam def;
  def.mu = 1.0; def.solref[0] = 0.02; def.solref[1] = 1.0; set_default_solimp(def.solimp); def.margin = 0;
  m->cp_torso_floor = def;
  if (m->family == 1 && variant != BO_ENV02_V1) { /* explicit pairs, envs/env01_v1.xml:30-33 (Env02's scene has none: envs/env02_v1.xml) */
    cparam p = def;
    p.mu = 0.9; p.solref[0] = 0.02; p.solref[1] = 0.5;
    p.solimp[0] = 0.5; p.solimp[1] = 0.5; p.solimp[2] = 0.002; p.solimp[3] = 0.5; p.solimp[4] = 2;
    m->cp_wheel_floor = p;
  } else
    m->cp_wheel_floor = def;
  { /* block geom: margin 0.002, solref (0.005, 0.9) (envs/env03_v1.xml:34); pair = max margin, 50/50 solref mix */
    cparam p = def;
    p.margin = 0.002;
    p.solref[0] = 0.5 * (0.02 + 0.005); p.solref[1] = 0.5 * (1.0 + 0.9);
    m->cp_block = p;
  }
  /* env-level */
  m->noise = (variant == BO_ENV01_V2);
  if (flags & BO_FLAG_NOISE_ON) m->noise = 1;
  if (flags & BO_FLAG_NOISE_OFF) m->noise = 0;
  m->auto_reset = (flags & BO_FLAG_AUTO_RESET) != 0;
  if (variant == BO_ENV01_V2) { m->Sy = 0.2; m->Sz = 2.0; } else { m->Sy = 0.4; m->Sz = 0.4; } /* env01_v2.py:61-62 / env01_v1.py:48-49 */
  m->max_episode_steps = max_episode_steps > 0 ? max_episode_steps : (variant == BO_ENV03_V2 ? 1200 : 6000); /* __init__.py */
  m->block_delay = variant == BO_ENV03_V2 ? 0.5 : 0.0; /* env03_v2.py:23 / env03_v1.py:24 */
  m->block_speed = variant == BO_ENV03_V2 ? 7.5 : 5.0;
  m->throw_v2 = variant == BO_ENV03_V2;

  /* compile-time constants MuJoCo derives at qpos0: body_invweight0, stat.meaninertia */
  double qpos0[NQMAX] = {0};
  qpos0[3] = 1; qpos0[12] = 1;
  kin_t k;
  kinematics(m, qpos0, &k);
  double M[NVMAX * NVMAX], L[NVMAX * NVMAX];
  mass_matrix(m, &k, M);
  chol(M, m->nv, NVMAX, L);
  double tr = 0;
  for (int i = 0; i < m->nv; i++) tr += M[i * NVMAX + i];
  m->meaninertia = tr / m->nv;
  for (int b = 1; b < m->nbody; b++) {
    double Jp[3][NVMAX], Jr[3][NVMAX], x[NVMAX], tp = 0, trr = 0;
    jac_point(m, &k, b, k.xipos[b], Jp, Jr);
    for (int c = 0; c < 3; c++) {
      chol_solve(L, m->nv, NVMAX, Jp[c], x);
      for (int j = 0; j < m->nv; j++) tp += Jp[c][j] * x[j];
      chol_solve(L, m->nv, NVMAX, Jr[c], x);
      for (int j = 0; j < m->nv; j++) trr += Jr[c][j] * x[j];
    }
    m->invweight0[b][0] = tp / 3;
    m->invweight0[b][1] = trr / 3;
  }
}

/* ============================================================================================
 * env logic (pinned by tests/golden/envlogic.json)
 * ========================================================================================== */
typedef struct {
  const bo_handle* h;
  env_t* e;
  int64_t gid;
  uint32_t buf[4];
  int pos;
} stream_t;
static void stream_open(stream_t* s, const bo_handle* h, env_t* e, int idx) { s->h = h; s->e = e; s->gid = h->gid_base + idx; s->pos = 4; }
static double snext(stream_t* s) {
  env_t* e = s->e;
  if (e->script && e->script_pos < e->script_n) return e->script[e->script_pos++];
  if (s->pos == 4) { philox_env(s->h->seed, s->gid, e->rng_ctr++, s->buf); s->pos = 0; }
  return (double)(s->buf[s->pos++] >> 8) * (1.0 / 16777216.0);
}

/* RobotBaseEnv.py:127-135 / :177-184: scipy Rotation.from_quat(x,y,z,w).as_euler('xyz') -> [0] pitch, [2] yaw.
   scipy (un-vendored dependency, pinned at 1.14.1 by conda-environment.yaml:10) computes Euler angles from the quaternion by the
   published algorithm of Bernardes & Viollet (2022), restated here for the extrinsic sequence 'xyz' (i, j, k = 0, 1, 2; sign +1):
   second angle from the two hypotenuses, first/third from a half sum and a half difference, and within eps = 1e-7 of gimbal lock
   the third angle is set to zero.  Checked against scipy itself through tests/golden/envlogic.json (pitch_yaw, pitch_yaw_gimbal). */
void bo_pitch_yaw(const double xq[4], double* pitch, double* yaw) {
  if (xq[0] == 0) { *pitch = 0; *yaw = 0; return; }
  double q[4] = {xq[0], xq[1], xq[2], xq[3]};
  normalize4(q);
  const double w = q[0], x = q[1], y = q[2], z = q[3], eps = 1e-7;
  const double a = w - y, b = x + z, c = y + w, d = z - x;
  const double second = 2 * atan2(hypot(c, d), hypot(a, b));
  const double hs = atan2(b, a), hd = atan2(d, c);
  double first, third;
  if (fabs(second) <= eps) { first = 2 * hs; third = 0; }             /* second angle -pi/2 */
  else if (fabs(second - PI) <= eps) { first = -2 * hd; third = 0; }  /* second angle +pi/2 (extrinsic: minus) */
  else { first = hs - hd; third = hs + hd; }
  if (first < -PI) first += 2 * PI; else if (first > PI) first -= 2 * PI;
  if (third < -PI) third += 2 * PI; else if (third > PI) third -= 2 * PI;
  *pitch = first;
  *yaw = third;
}
/* env01_v2.py:16-20: every get_pitch() call adds a fresh U(-0.025,0.025) sample */
static double get_pitch(const model_t* m, env_t* e, stream_t* s) {
  double p, y;
  bo_pitch_yaw(e->xquat, &p, &y);
  if (m->noise) p += (snext(s) - 0.5) * 0.05;
  if (m->variant == BO_ENV01_V3) p += e->poff; /* env01_v3.py:23-25 */
  return p;
}
/* RobotBaseEnv.py:190-219 */
/* env01_v3.py:55-96 */
static double get_reward_v3(const model_t* m, env_t* e, stream_t* s) {
  double reward = 0.6, pitch = get_pitch(m, e, s);
  double ws = (e->qvel[6] + (-1 * e->qvel[7])) / 2, dv = e->tws - ws;
  reward -= fabs(pitch) * 0.05;
  double mdv = dv > 40.0 ? 40.0 : (dv < -40.0 ? -40.0 : dv), dv_s = fabs(mdv / 40.0);
  reward -= 0.15 * dv_s;
  if (e->tws > 0 && e->tws > ws) reward += (-1.0 * pitch) * 10.0 * dv_s;
  else if (e->tws < 0 && e->tws < ws) reward += (1.0 * pitch) * 10.0 * dv_s;
  else if (e->tws > 0 && e->tws < ws) reward += (1.0 * pitch) * 10.0 * dv_s;
  else if (e->tws < 0 && e->tws > ws) reward += (-1.0 * pitch) * 10.0 * dv_s;
  double dyd = 0 - (e->qvel[6] - (-1 * e->qvel[7]));
  reward -= 0.007 * fabs(dyd);
  return reward;
}
static double get_reward(const model_t* m, env_t* e, stream_t* s) {
  if (m->variant == BO_ENV01_V3) return get_reward_v3(m, e, s);
  double reward = 1.0, vel_l = e->qvel[6], vel_r = e->qvel[7];
  double dv = 0 - (vel_l * -1 + vel_r) / 2.0;
  double dyd = 0 - e->qvel[5];
  reward -= 0.025 * fabs(dyd);
  double pitch = get_pitch(m, e, s);
  reward -= fabs(pitch);
  reward += pitch * dv * 0.5;
  return reward;
}
/* RobotBaseEnv.py:221-246 with get_pitch_dot_alt :142-157 */
static void get_obs(const model_t* m, env_t* e, stream_t* s, int at_reset, float* obs) {
  double pitch = get_pitch(m, e, s);
  double pitch2 = get_pitch(m, e, s); /* get_pitch_dot_alt samples again */
  double pitch_dot = 0, dt = m->nsub * m->h;
  if (!at_reset) { /* at reset time = 0 <= last_time -> 0 ; otherwise dt = time - last_time = nsub*h > 0 */
    pitch_dot = (pitch2 - e->last_pitch) / dt;
  }
  e->last_pitch = pitch2;
  double vl = e->qvel[6], vr = e->qvel[7];
  double wheel_speed = (vl + (-1 * vr)) / 2, wheel_yaw = vl - (-1 * vr);
  obs[0] = (float)(pitch / 0.25);
  obs[1] = (float)(pitch_dot / 1);
  obs[2] = (float)(vl / 170.0 * 4);
  obs[3] = (float)(vr / 170.0 * 4);
  obs[4] = (float)(((m->variant == BO_ENV01_V3 ? e->tws : 0.0) - wheel_speed) / 170.0 * 4);
  obs[5] = (float)((0.0 - wheel_yaw) / 45.0 * 3);
}
/* scipy from_euler('xyz',[a,b,c]).as_quat() = (x,y,z,w) of Rz(c)Ry(b)Rx(a), stored into MuJoCo's (w,x,y,z)
 * slot unchanged (env01_v2.py:64-66): q_mj = (x_s, y_s, z_s, w_s) */
void bo_euler_slot_quat(double a, double b, double c, double q[4]) {
  double ca = cos(a / 2), sa = sin(a / 2), cb = cos(b / 2), sb = sin(b / 2), cc = cos(c / 2), sc = sin(c / 2);
  double w = ca * cb * cc + sa * sb * sc, x = sa * cb * cc - ca * sb * sc, y = ca * sb * cc + sa * cb * sc,
         z = ca * cb * sc - sa * sb * cc;
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
/* env03_v1.py:88-114 (v1) / env03_v2.py:25-59 (v2) */
static void set_block_pos_vel(const model_t* m, env_t* e, stream_t* s) {
  double ang, tx, tz;
  if (m->throw_v2) {
    double p, yaw;
    bo_pitch_yaw(e->xquat, &p, &yaw); /* get_yaw(): Env03_v2 has no noise override on yaw */
    ang = -yaw;
    if (!e->side_front) ang += PI;
  } else
    ang = snext(s) * 2 * PI;
  double bx = 0.3 * sin(ang) + e->xpos[0], by = 0.3 * cos(ang) + e->xpos[1], bz = 0.15;
  if (m->throw_v2) { tx = (snext(s) - 0.5) * 0.02 + e->xpos[0]; tz = snext(s) * 0.025 + 0.13; }
  else { tx = (snext(s) - 0.5) * 0.06 + e->xpos[0]; tz = snext(s) * 0.075 + 0.1; }
  double v[3] = {tx - bx, (0 + e->xpos[1]) - by, tz - bz}, n = norm3(v);
  for (int i = 0; i < 3; i++) v[i] = m->block_speed * (v[i] / n);
  double xr = snext(s) * 2 * PI, yr = snext(s) * 2 * PI, zr = snext(s) * 2 * PI;
  e->qpos[9] = bx; e->qpos[10] = by; e->qpos[11] = bz;
  bo_euler_slot_quat(xr, yr, zr, e->qpos + 12);
  e->qvel[8] = v[0]; e->qvel[9] = v[1]; e->qvel[10] = v[2];
}

static void env_reset(const bo_handle* h, int idx, float* obs) {
  const model_t* m = &h->m;
  env_t* e = h->e + idx;
  stream_t s;
  stream_open(&s, h, e, idx);
  /* MujocoEnv.reset -> mj_resetData */
  memset(e->qpos, 0, sizeof e->qpos); memset(e->qvel, 0, sizeof e->qvel); memset(e->warm, 0, sizeof e->warm);
  e->time = 0; e->elapsed = 0; e->ep_return = 0;
  /* reset_model (env01_v2.py:52-71): qpos0 + U(-0.01,0.01)^nq from the seeded generator, qpos[2] = 0 */
  if (m->variant == BO_ENV01_V3) { /* env01_v3.py:40-53: two draws of the seeded generator BEFORE Env01.reset_model */
    e->tws = 0;
    e->dts = -10.0 + 20.0 * snext(&s);
    if (e->dts > 0) e->dts += 10; else e->dts -= 10;
    e->poff = -0.0349066 + 2 * 0.0349066 * snext(&s);
  }
  double q0[NQMAX] = {0};
  q0[3] = 1; q0[12] = 1;
  for (int i = 0; i < m->nq; i++) e->qpos[i] = q0[i] + (-0.01 + 0.02 * snext(&s));
  e->qpos[2] = 0;
  double xr = (snext(&s) - 0.5) * 2 * PI, yr = (snext(&s) - 0.5) * m->Sy, zr = (snext(&s) - 0.5) * m->Sz;
  bo_euler_slot_quat(xr, yr, zr, e->qpos + 3);
  if (m->variant == BO_ENV02_V1) e->muw = snext(&s) / 2 + 0.5; /* env02_v1.py:61-65, after the pose draws */
  /* set_state -> mj_forward: accessor pose is current */
  memcpy(e->xquat, e->qpos + 3, sizeof e->xquat);
  normalize4(e->xquat);
  memcpy(e->xpos, e->qpos, sizeof e->xpos);
  if (m->has_block) { set_block_pos_vel(m, e, &s); e->block_timer = NAN; }
  get_obs(m, e, &s, 1, obs);
}

static void env_step(bo_handle* h, int idx, const float* action, float* obs, float* reward, uint8_t* terminated,
                     uint8_t* truncated, float* terminal_obs) {
  const model_t* m = &h->m;
  env_t* e = h->e + idx;
  stream_t s;
  stream_open(&s, h, e, idx);
  if (m->variant == BO_ENV01_V3) { /* env01_v3.py:28-36: schedule keyed on data.time at the start of step */
    if (e->time > 5.5) e->tws = 3.0 * e->dts;
    else if (e->time > 4.5) e->tws = 2.0 * e->dts;
    else if (e->time > 3.0) e->tws = -1.0 * e->dts;
    else if (e->time > 1.0) e->tws = e->dts;
  }
  double rew = get_reward(m, e, &s); /* on the PRE-step state (env01_v2.py:29) */
  double ctrl[2] = {e->qvel[6] + (double)action[0] * 4.0, e->qvel[7] + (double)action[1] * 4.0}; /* :31-36 */
  e->last_ctrl[0] = ctrl[0]; e->last_ctrl[1] = ctrl[1];
  if (e->stub) {
    memcpy(e->qpos, e->stub_qpos, sizeof(double) * m->nq);
    memcpy(e->qvel, e->stub_qvel, sizeof(double) * m->nv);
    memcpy(e->xquat, e->stub_xquat, sizeof e->xquat);
    memcpy(e->xpos, e->stub_xpos, sizeof e->xpos);
    for (int k = 0; k < m->nsub; k++) e->time += m->h;
    e->stub = 0;
  } else {
    fwd_t f;
    for (int k = 0; k < m->nsub; k++)
      if (substep(m, e, ctrl, &f)) { /* MuJoCo mj_check*: bad state -> reset the simulation */
        float tmp[6];
        e->bad_count++;
        env_reset(h, idx, tmp);
        break;
      }
  }
  if (m->has_block) { /* env03_v1.py:39-49 */
    double bv = sqrt(e->qvel[8] * e->qvel[8] + e->qvel[9] * e->qvel[9] + e->qvel[10] * e->qvel[10]);
    if (bv < 0.1 && isnan(e->block_timer)) {
      e->qpos[9] = 10; e->qpos[10] = 10; e->qpos[11] = 0; /* remove_block :85-86 */
      e->block_timer = e->time;
    }
    if (!isnan(e->block_timer) && (e->time - e->block_timer) > m->block_delay) {
      set_block_pos_vel(m, e, &s);
      e->block_timer = NAN;
    }
  }
  int term = fabs(get_pitch(m, e, &s)) > (50 * PI / 180); /* env01_v2.py:44 */
  get_obs(m, e, &s, 0, obs);
  e->elapsed++;
  e->ep_return += rew;
  int trunc = e->elapsed >= m->max_episode_steps; /* gymnasium TimeLimit, __init__.py:15,50 */
  *reward = (float)rew; *terminated = (uint8_t)term; *truncated = (uint8_t)trunc;
  if (terminal_obs) memcpy(terminal_obs, obs, 6 * sizeof(float));
  if (m->auto_reset && (term || trunc)) env_reset(h, idx, obs);
}

/* ============================================================================================
 * public API
 * ========================================================================================== */
bo_handle* bo_create(int variant, int n, uint64_t seed, int64_t gid_base, uint32_t flags, int max_episode_steps,
                     int substeps, double timestep) {
  if (variant < 0 || variant > 5 || n <= 0) return NULL;
  bo_handle* h = (bo_handle*)calloc(1, sizeof *h);
  model_init(&h->m, variant, flags, max_episode_steps, substeps, timestep);
  h->n = n; h->seed = seed; h->gid_base = gid_base; h->nthreads = 1;
  h->e = (env_t*)calloc((size_t)n, sizeof(env_t));
  for (int i = 0; i < n; i++) {
    env_t* e = h->e + i;
    /* Env03_v2.__init__ (env03_v2.py:22): side chosen once per env object; Philox block 0 is reserved for it */
    e->side_front = bo_uniform(seed, gid_base + i, 0, 0) > 0.5;
    e->rng_ctr = 1;
    e->qpos[3] = 1; e->qpos[12] = 1; e->xquat[0] = 1;
    e->block_timer = NAN; e->last_pitch = 0;
    e->muw = 1.0; /* geom default until the first reset */
  }
  return h;
}
void bo_destroy(bo_handle* h) {
  if (!h) return;
  for (int i = 0; i < h->n; i++) free(h->e[i].script);
  free(h->e);
  free(h);
}
int bo_nq(const bo_handle* h) { return h->m.nq; }
int bo_nv(const bo_handle* h) { return h->m.nv; }
void bo_set_threads(bo_handle* h, int n) { h->nthreads = n > 0 ? n : 1; }
void bo_model_info_get(const bo_handle* h, bo_model_info* o) {
  memset(o, 0, sizeof *o);
  for (int b = 0; b < 5; b++) {
    o->body_mass[b] = h->m.mass[b];
    for (int i = 0; i < 3; i++) { o->body_inertia[b][i] = h->m.inertia[b][i]; o->body_ipos[b][i] = h->m.ipos[b][i]; }
    o->invweight0[b][0] = h->m.invweight0[b][0]; o->invweight0[b][1] = h->m.invweight0[b][1];
  }
  o->meaninertia = h->m.meaninertia; o->nq = h->m.nq; o->nv = h->m.nv;
}

void bo_reset(bo_handle* h, const uint8_t* mask, float* obs) {
  for (int i = 0; i < h->n; i++)
    if (!mask || mask[i]) env_reset(h, i, obs + 6 * i);
}
void bo_step(bo_handle* h, const float* actions, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
             float* terminal_obs) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(h->nthreads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < h->n; i++)
    env_step(h, i, actions + 2 * i, obs + 6 * i, reward + i, terminated + i, truncated + i,
             terminal_obs ? terminal_obs + 6 * i : NULL);
}
void bo_physics(bo_handle* h, const double* ctrl, int nsub) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(h->nthreads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < h->n; i++) {
    fwd_t f;
    for (int k = 0; k < nsub; k++)
      if (substep(&h->m, h->e + i, ctrl + 2 * i, &f)) { h->e[i].bad_count++; break; }
  }
}
void bo_get_state(const bo_handle* h, double* qpos, double* qvel, double* warm, double* time) {
  int nq = h->m.nq, nv = h->m.nv;
  for (int i = 0; i < h->n; i++) {
    if (qpos) memcpy(qpos + (size_t)i * nq, h->e[i].qpos, nq * sizeof(double));
    if (qvel) memcpy(qvel + (size_t)i * nv, h->e[i].qvel, nv * sizeof(double));
    if (warm) memcpy(warm + (size_t)i * nv, h->e[i].warm, nv * sizeof(double));
    if (time) time[i] = h->e[i].time;
  }
}
void bo_set_state(bo_handle* h, const double* qpos, const double* qvel, const double* warm, const double* time) {
  int nq = h->m.nq, nv = h->m.nv;
  for (int i = 0; i < h->n; i++) {
    env_t* e = h->e + i;
    if (qpos) {
      memcpy(e->qpos, qpos + (size_t)i * nq, nq * sizeof(double));
      memcpy(e->xquat, e->qpos + 3, sizeof e->xquat);
      normalize4(e->xquat);
      memcpy(e->xpos, e->qpos, sizeof e->xpos);
    }
    if (qvel) memcpy(e->qvel, qvel + (size_t)i * nv, nv * sizeof(double));
    if (warm) memcpy(e->warm, warm + (size_t)i * nv, nv * sizeof(double));
    if (time) e->time = time[i];
  }
}
void bo_get_aux(const bo_handle* h, double* aux) {
  for (int i = 0; i < h->n; i++) {
    const env_t* e = h->e + i;
    double p, y;
    bo_pitch_yaw(e->xquat, &p, &y);
    double* a = aux + 14 * (size_t)i;
    a[8] = e->last_ctrl[0]; a[9] = e->last_ctrl[1];
    a[10] = e->muw; a[11] = e->dts; a[12] = e->poff; a[13] = e->tws;
    a[0] = e->last_pitch; a[1] = e->block_timer; a[2] = e->elapsed; a[3] = e->rng_ctr; a[4] = e->side_front;
    a[5] = p; a[6] = e->ep_return; a[7] = e->bad_count;
  }
}
void bo_set_aux(bo_handle* h, const double* aux) {
  for (int i = 0; i < h->n; i++) {
    env_t* e = h->e + i;
    const double* a = aux + 14 * (size_t)i;
    e->last_pitch = a[0]; e->block_timer = a[1]; e->elapsed = (int)a[2]; e->rng_ctr = (uint32_t)a[3];
    e->side_front = a[4] != 0; e->ep_return = a[6];
    e->muw = a[10]; e->dts = a[11]; e->poff = a[12]; e->tws = a[13];
  }
}
void bo_get_xpose(const bo_handle* h, double* xquat, double* xpos) {
  for (int i = 0; i < h->n; i++) {
    if (xquat) memcpy(xquat + 4 * (size_t)i, h->e[i].xquat, 4 * sizeof(double));
    if (xpos) memcpy(xpos + 3 * (size_t)i, h->e[i].xpos, 3 * sizeof(double));
  }
}
void bo_set_xpose(bo_handle* h, const double* xquat, const double* xpos) {
  for (int i = 0; i < h->n; i++) {
    if (xquat) memcpy(h->e[i].xquat, xquat + 4 * (size_t)i, 4 * sizeof(double));
    if (xpos) memcpy(h->e[i].xpos, xpos + 3 * (size_t)i, 3 * sizeof(double));
  }
}

void bo_forward(bo_handle* h, int idx, const double ctrl[2], bo_forward_out* o) {
  const model_t* m = &h->m;
  env_t* e = h->e + idx;
  fwd_t* f = (fwd_t*)malloc(sizeof *f);
  forward(m, e->qpos, e->qvel, ctrl, e->warm, f, m->variant == BO_ENV02_V1 ? e->muw : 0.0);
  memset(o, 0, sizeof *o);
  o->nv = m->nv; o->ncon = f->ncon; o->nefc = f->nefc; o->solver_iter = f->solver_iter;
  memcpy(o->M, f->M, sizeof o->M);
  memcpy(o->bias, f->bias, sizeof o->bias); memcpy(o->passive, f->passive, sizeof o->passive);
  memcpy(o->actuator, f->actuator, sizeof o->actuator); memcpy(o->qacc_smooth, f->qacc_smooth, sizeof o->qacc_smooth);
  memcpy(o->qacc, f->qacc, sizeof o->qacc); memcpy(o->qfrc_constraint, f->qfrc_constraint, sizeof o->qfrc_constraint);
  memcpy(o->qacc_integ, f->qacc_integ, sizeof o->qacc_integ);
  memcpy(o->xquat, f->k.tquat, sizeof o->xquat); memcpy(o->xpos, f->k.xpos[B_TORSO], sizeof o->xpos);
  memcpy(o->con, f->con, sizeof(bo_contact) * f->ncon);
  memcpy(o->efc_D, f->D, sizeof(double) * f->nefc); memcpy(o->efc_aref, f->aref, sizeof(double) * f->nefc);
  if (f->nefc) memcpy(o->efc_force, f->force, sizeof(double) * f->nefc);
  double Mv[NVMAX];
  mulM(f, m->nv, e->qvel, Mv);
  for (int j = 0; j < m->nv; j++) o->energy_kin += 0.5 * e->qvel[j] * Mv[j];
  for (int b = 1; b < m->nbody; b++) o->energy_pot += -m->mass[b] * m->gravity[2] * f->k.xipos[b][2];
  free(f);
}

void bo_script_uniforms(bo_handle* h, int idx, const double* u, int n) {
  env_t* e = h->e + idx;
  free(e->script);
  e->script = NULL; e->script_n = e->script_pos = 0;
  if (n > 0) {
    e->script = (double*)malloc(sizeof(double) * n);
    memcpy(e->script, u, sizeof(double) * n);
    e->script_n = n;
  }
}
int bo_script_remaining(const bo_handle* h, int idx) { return h->e[idx].script_n - h->e[idx].script_pos; }
void bo_stub_physics(bo_handle* h, int idx, const double* qpos, const double* qvel, const double* xquat, const double* xpos) {
  env_t* e = h->e + idx;
  memcpy(e->stub_qpos, qpos, sizeof(double) * h->m.nq);
  memcpy(e->stub_qvel, qvel, sizeof(double) * h->m.nv);
  memcpy(e->stub_xquat, xquat, sizeof e->stub_xquat);
  memcpy(e->stub_xpos, xpos, sizeof e->stub_xpos);
  e->stub = 1;
}

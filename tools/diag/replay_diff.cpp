// Diagnostic (host build of the kernel source): replay ONE env step from a dumped pre-state (tools/parity_locate.py) substep by
// substep in float and in double side by side; prints the first substeps at which the two disagree on a discrete quantity
// (contact counts, active-row masks, Newton iterations) and the velocity difference around them.
//   g++ -O2 -std=c++17 -ffp-contract=off -I../../balance_robot_mujoco_rl_amd/csrc replay_diff.cpp -o /tmp/diag/replay_diff
//   python - <<< '...dump "variant nq nv qpos.. qvel.. warm.. time ctrlL ctrlR" ...' | /tmp/diag/replay_diff
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "brs_state.hpp"
using namespace brs;

template <typename R, bool BLK> struct Run {
  using L = Layout<BLK>;
  using SimT = Sim<R, BLK>;
  Params<R> P;
  EnvState<R, BLK> S;
  R buf[LDS_WORDS_ENV03];
  typename SimT::SubCtx C;
  int iters;
  void init(int variant, const double* qpos, const double* qvel, const double* warm, double tm) {
    P = make_params<R>(variant, false, 0, 0, 0, 0.0, 0, 0);
    std::vector<double> d(L::ND); std::vector<R> f(L::NF); std::vector<int> ii(L::NI);
    hostconv::init_state<BLK>(d.data(), f.data(), ii.data(), 1, 0, 0);
    hostconv::set_state<BLK>(d.data(), f.data(), 1, qpos, qvel, warm, &tm);
    load_state<R, BLK>(S, d.data(), f.data(), ii.data(), 1, 0);
  }
  void substep(R cl, R cr) {
    Store<R> st{buf, 1};
    SimT::sub_begin(P, st, S, cl, cr, C);
    iters = 0;
    while (!C.conv) { SimT::sub_iter(P, st, S, C); iters++; }
    SimT::sub_end(P, S, C);
    S.derive_vel32();  // (BRS_LAZY_VEL32: the fp32 mirrors are otherwise refreshed when the next substep starts)
  }
};

template <bool BLK> int go(int variant, const double* qpos, const double* qvel, const double* warm, double tm, double cl, double cr, int verbose) {
  Run<float, BLK> F; Run<double, BLK> D;
  F.init(variant, qpos, qvel, warm, tm); D.init(variant, qpos, qvel, warm, tm);
  double prev = 0;
  int shown = 0;
  for (int k = 0; k < 250; k++) {
    F.substep((float)cl, (float)cr); D.substep(cl, cr);
    double ev = 0;
    for (int i = 0; i < 3; i++) { ev = fmax(ev, fabs((double)F.S.v[i] - D.S.v[i])); ev = fmax(ev, fabs((double)F.S.w[i] - D.S.w[i])); }
    for (int i = 0; i < 2; i++) ev = fmax(ev, fabs((double)F.S.ww[i] - D.S.ww[i]));
    if constexpr (BLK) for (int i = 0; i < 3; i++) { ev = fmax(ev, fabs((double)F.S.bv[i] - D.S.bv[i])); ev = fmax(ev, fabs((double)F.S.bw[i] - D.S.bw[i])); }
    const bool diff = F.C.F.nfr != D.C.F.nfr || F.C.F.nfb != D.C.F.nfb || F.C.F.nc != D.C.F.nc || F.C.F.sels != D.C.F.sels ||
                      F.C.M.nR != D.C.M.nR || F.C.M.nB != D.C.M.nB || F.C.M.nC != D.C.M.nC || F.iters != D.iters || F.C.clL != D.C.clL || F.C.clR != D.C.clR || F.C.M.hR != D.C.M.hR || F.C.M.hC != D.C.M.hC;
    const bool jump = ev > 1e-4 && ev > 20 * fmax(prev, 1e-8);
    if ((diff || jump || verbose) && shown < 400) {
      shown++;
      printf("substep %3d: dv %.3g%s | F nfr %d nfb %d nc %d sels %x masks R %x B %x C %x it %d cl %d%d hR %x hC %x | D nfr %d nfb %d nc %d sels %x masks R %x B %x C %x it %d cl %d%d hR %x hC %x\n", k, ev,
             jump ? " JUMP" : "", F.C.F.nfr, F.C.F.nfb, F.C.F.nc, F.C.F.sels, F.C.M.nR, F.C.M.nB, F.C.M.nC, F.iters, (int)F.C.clL, (int)F.C.clR, F.C.M.hR, F.C.M.hC, D.C.F.nfr, D.C.F.nfb, D.C.F.nc,
             D.C.F.sels, D.C.M.nR, D.C.M.nB, D.C.M.nC, D.iters, (int)D.C.clL, (int)D.C.clR, D.C.M.hR, D.C.M.hC);
    }
    prev = ev;
  }
  double eq = 0;
  for (int i = 0; i < 3; i++) eq = fmax(eq, fabs(F.S.p[i] - D.S.p[i]));
  for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(F.S.q[i] - D.S.q[i]));
  for (int i = 0; i < 2; i++) eq = fmax(eq, fabs(F.S.th[i] - D.S.th[i]));
  if constexpr (BLK) { for (int i = 0; i < 3; i++) eq = fmax(eq, fabs(F.S.bp[i] - D.S.bp[i])); for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(F.S.bq[i] - D.S.bq[i])); }
  printf("final |dqpos| float vs double: %.3g\n", eq);
  return 0;
}

int main(int argc, char** argv) {
  int variant, nq, nv;
  if (scanf("%d %d %d", &variant, &nq, &nv) != 3) return 1;
  std::vector<double> qpos(nq), qvel(nv), warm(nv);
  double tm, cl, cr;
  for (auto& x : qpos) if (scanf("%lf", &x) != 1) return 1;
  for (auto& x : qvel) if (scanf("%lf", &x) != 1) return 1;
  for (auto& x : warm) if (scanf("%lf", &x) != 1) return 1;
  if (scanf("%lf %lf %lf", &tm, &cl, &cr) != 3) return 1;
  const int verbose = argc > 1;
  return nq == 16 ? go<true>(variant, qpos.data(), qvel.data(), warm.data(), tm, cl, cr, verbose)
                  : go<false>(variant, qpos.data(), qvel.data(), warm.data(), tm, cl, cr, verbose);
}

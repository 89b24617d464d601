// Diagnostic (host build of the kernel source, fp32, -DBRS_STATS): contact-list lengths and block<->robot broad-phase
// statistics of Env03-v2 under the bench workload (random policy, auto-reset), past the episode-start transient.
//   g++ -O2 -fopenmp -std=c++17 -DBRS_STATS -ffp-contract=off -I../../balance_robot_mujoco_rl_amd/csrc contact_stats.cpp -o /tmp/contact_stats && /tmp/contact_stats 512 400 300
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include "brs_state.hpp"
using namespace brs;
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512, STEPS = argc > 2 ? atoi(argv[2]) : 400, SKIP = argc > 3 ? atoi(argv[3]) : 300;
  using R = float;
  using L = Layout<true>;
  Params<R> P = make_params<R>(3, true, -1, 0, 0, 0.0, 0, 0);
  std::vector<double> d(L::ND * (size_t)N); std::vector<R> f(L::NF * (size_t)N); std::vector<int> ii(L::NI * (size_t)N);
  hostconv::init_state<true>(d.data(), f.data(), ii.data(), N, 0, 0);
  std::vector<float> obs(6 * N);
  for (int i = 0; i < N; i++) {
    EnvState<R, true> S; load_state<R, true>(S, d.data(), f.data(), ii.data(), N, i);
    Stream<R> rng; rng.open(P.seed, P.gid_base + i, S.rng_ctr);
    Sim<R, true>::env_reset(P, S, rng, obs.data() + 6 * i); S.rng_ctr = rng.ctr;
    store_state<R, true>(S, d.data(), f.data(), ii.data(), N, i);
  }
  std::vector<float> act(2 * N);
  std::mt19937 gen(1); std::uniform_real_distribution<float> U(-1, 1);
  Stats tot{};
  long key_hist[8] = {0}, key_nc[8] = {0}, key_sat[8] = {0}, key_trips[8] = {0};
  long lane_steps = 0, lane_steps_reach = 0, lane_steps_nc = 0, wave_steps = 0, wave_steps_nc = 0, wave_steps_reach = 0, ndone = 0;
  for (int t = 0; t < STEPS; t++) {
    for (auto& a : act) a = U(gen);
    std::vector<Stats> per(N);
    std::vector<int> key(N);
    for (int i = 0; i < N; i++) { EnvState<R, true> S0; load_state<R, true>(S0, d.data(), f.data(), ii.data(), N, i); key[i] = cost_class<R, true>(P, S0); }
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < N; i++) {
      R buf[LDS_WORDS_ENV03];
      Stream<R> rng; rng.open(P.seed, P.gid_base + i, 0u);
      Store<R> st{buf, 1}; int te, trn; float tob[6], rew;
      stats() = Stats{};
      env_step_mem<R, true, R>(P, st, rng, d.data(), f.data(), ii.data(), N, i, act[2 * i], act[2 * i + 1], obs.data() + 6 * i, tob, rew, te, trn);
      per[i] = stats();
      if (te || trn) { 
#pragma omp atomic
        ndone++; }
    }
    if (t < SKIP) continue;
    for (int i = 0; i < N; i++) {
      const Stats& s = per[i];
      long* a = (long*)&tot; const long* b = (const long*)&s;
      for (size_t k = 0; k < sizeof(Stats) / sizeof(long); k++) a[k] += b[k];  // (last_iters ints are summed as garbage; unused)
      key_hist[key[i]]++; key_nc[key[i]] += (s.nc_hist[1] + s.nc_hist[2] + s.nc_hist[3] + s.nc_hist[4] + s.nc_hist[5] + s.nc_hist[6] + s.nc_hist[7]) > 0; key_sat[key[i]] += s.cp_torso; key_trips[key[i]] += s.trips;
      lane_steps++; lane_steps_reach += s.cp_reach > 0; lane_steps_nc += s.cp_nc > 0 || (s.nc_hist[1] + s.nc_hist[2] + s.nc_hist[3] + s.nc_hist[4]) > 0;
    }
    for (int w = 0; w + 64 <= N; w += 64) {
      bool anyr = false, anyc = false;
      for (int i = w; i < w + 64; i++) { anyr |= per[i].cp_reach > 0; anyc |= (per[i].nc_hist[1] + per[i].nc_hist[2] + per[i].nc_hist[3] + per[i].nc_hist[4]) > 0; }
      wave_steps++; wave_steps_reach += anyr; wave_steps_nc += anyc;
    }
  }
  double S = (double)tot.substeps;
  printf("lane-substeps %.0f (envs %d, steps %d after %d)\n", S, N, STEPS - SKIP, SKIP);
  printf("coupled broad phase: calls %.4f  sphere-reach pass %.4f  torso narrow %.4f (tight AABB would pass %.4f)  wheel narrow (x2) %.4f (tight %.4f)\n",
         tot.cp_calls / S, tot.cp_reach / S, tot.cp_torso / S, tot.cp_tight_torso / S, tot.cp_wheel / S, tot.cp_tight_wheel / S);
  printf("nfr:"); for (int k = 0; k < 9; k++) printf(" [%d]=%.4f", k, tot.nfr_hist[k] / S);
  printf("\nnfb:"); for (int k = 0; k < 5; k++) printf(" [%d]=%.4f", k, tot.nfb_hist[k] / S);
  printf("\nnc :"); for (int k = 0; k < 5; k++) printf(" [%d]=%.4f", k, tot.nc_hist[k] / S);
  printf("\nlane-steps with any sphere-reach substep %.4f, with any coupled contact %.4f\n", (double)lane_steps_reach / lane_steps, (double)lane_steps_nc / lane_steps);
  printf("64-lane wave-steps with any reach %.4f, any coupled contact %.4f\n", (double)wave_steps_reach / wave_steps, (double)wave_steps_nc / wave_steps);
  for (int k = 0; k < 8; k++) printf("key %d (floor %d wheel %d far %d): share %.4f  with coupled contact %.4f  torso-narrow substeps/step %.1f  trips/step %.1f\n", k, k & 1, (k >> 1) & 1, k >> 2,
      (double)key_hist[k] / lane_steps, key_hist[k] ? (double)key_nc[k] / key_hist[k] : 0.0, key_hist[k] ? (double)key_sat[k] / key_hist[k] : 0.0, key_hist[k] ? (double)key_trips[k] / key_hist[k] : 0.0);
  printf("episodes finished %ld (mean length %.1f env steps)\n", ndone, (double)N * STEPS / std::max(1L, ndone));
  return 0;
}

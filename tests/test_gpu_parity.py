"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): max per-step |dqpos| vs the oracle < 1e-4, teacher-forced (both sides start
every env step from the oracle's state).  The oracle is this repo's fp64 restatement; parity with MuJoCo itself
is UNPINNED (MuJoCo is not installable here) -- see oracle/brs_oracle.h."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_QPOS = 1e-4

# ---- parity gates (facts behind them: profiles/r03_parity_*.json, tools/parity_report.py)
# An env-step is UPRIGHT if the torso axis is within 60 degrees of vertical when the step starts (the env terminates at
# 50 degrees of pitch, so with auto-reset every step it keeps is upright).  FALLEN robots exist only with auto-reset off.
# Since round 3 (contact-existence and servo-clamp decisions from exact fp64 constants, fp64 velocity accumulators) the
# north-star bound holds as a STRICT maximum in every group: 0 of 8.7 M campaign env-steps above 1e-4 (worst 8.6e-5, a block
# quaternion under the balancing policy; robot coordinates 5.2e-5; fallen robots 7.2e-5).  The gates are that bound, with
# zero exceptions, plus per-test caps at ~5-10x what the test's own sample measured (r03 GPU log), so that a regression of
# one order of magnitude in the bulk fails even when no env-step crosses 1e-4:
#  G1  robot coordinates (torso position, quaternion, wheel angles), upright: max |dqpos| < 1e-4
#  G2  block coordinates, upright:                                            max |dqpos| < 1e-4
#  G3  all coordinates of fallen robots (lying flat, wheels rubbing):         max |dqpos| < 1e-4
# (round 2's gates allowed 5e-5 / 2e-4 of the env-steps above 1e-4 with caps at 1e-3, and one free outlier per test.)


class Gates:
    def __init__(self):
        self.n = {"up": 0, "fallen": 0}
        self.robot_up_max = self.block_up_max = self.fallen_max = 0.0
        self.skipped = 0.0

    def add(self, qpos_pre, q_gpu, q_orc, skip=None):
        d = np.abs(q_gpu - q_orc)
        if skip is not None:
            d = d[~skip]; qpos_pre = qpos_pre[~skip]
        up = 1 - 2 * (qpos_pre[:, 4] ** 2 + qpos_pre[:, 5] ** 2) > 0.5
        self.n["up"] += int(up.sum()); self.n["fallen"] += int((~up).sum())
        if up.any():
            self.robot_up_max = max(self.robot_up_max, float(d[up][:, :9].max()))
            if d.shape[1] > 9:
                self.block_up_max = max(self.block_up_max, float(d[up][:, 9:].max()))
        if (~up).any():
            self.fallen_max = max(self.fallen_max, float(d[~up].max()))

    def check(self, label, robot_cap=TOL_QPOS, block_cap=TOL_QPOS, fallen_cap=TOL_QPOS):
        """caps: what THIS test's sample may reach (<= the 1e-4 bound); printed values go to the GPU test log"""
        print(f"{label}: upright {self.n['up']} env-steps: robot max {self.robot_up_max:.3g}, block max {self.block_up_max:.3g}; "
              f"fallen {self.n['fallen']}: max {self.fallen_max:.3g}")
        assert max(robot_cap, block_cap, fallen_cap) <= TOL_QPOS
        assert self.robot_up_max < robot_cap, f"G1: robot coordinates {self.robot_up_max:.3g} on an upright env-step (cap {robot_cap:g})"
        assert self.block_up_max < block_cap, f"G2: block coordinates {self.block_up_max:.3g} on an upright env-step (cap {block_cap:g})"
        assert self.fallen_max < fallen_cap, f"G3: fallen robot {self.fallen_max:.3g} (cap {fallen_cap:g})"


def _mk(env_id, n, **kw):
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    from oracle import oracle as O
    sim = BatchedSim(env_id, n, device=0, **kw)
    okw = dict(seed=kw.get("seed", 0), auto_reset=kw.get("auto_reset", True), threads=min(16, os.cpu_count() or 1),
               env_index_base=kw.get("env_index_base", 0))
    if "obs_noise" in kw:
        okw["noise"] = kw["obs_noise"]
    for k in ("max_episode_steps", "substeps", "timestep"):
        if k in kw:
            okw[k] = kw[k]
    orc = O.Oracle(env_id, n, **okw)
    return torch, sim, orc


def test_library_and_sizes():
    from balance_robot_mujoco_rl_amd import BatchedSim
    s = BatchedSim("Env03-v2", 64, auto_reset=False)
    assert (s.nq, s.nv) == (16, 14) and s.max_episode_steps == 1200
    s.close()
    s = BatchedSim("Env01-v2", 64)
    assert (s.nq, s.nv) == (9, 8) and s.max_episode_steps == 6000
    s.close()


@pytest.mark.parametrize("env_id,n,steps", [("Env01-v2", 512, 120), ("Env03-v2", 512, 100), ("Env02-v1", 256, 80)])
def test_teacher_forced_physics_parity(env_id, n, steps):
    """random-action rollout, auto-reset off (robots fall and stay down, blocks pile onto them), noise off: per-step state
    parity of 250 substeps, gated per coordinate group (G1-G3 above)"""
    torch, sim, orc = _mk(env_id, n, seed=3, auto_reset=False, obs_noise=False)
    orc.reset()
    sim.set_aux(orc.get_aux())  # per-episode friction (Env02) lives in aux
    rng = np.random.default_rng(5)
    g = Gates()
    for t in range(steps):
        qpos, qvel, warm, tm = orc.get_state()
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        if t % 3 == 0:
            act[:] = 0
        ctrl = (qvel[:, 6:8] + act.astype(np.float64) * 4.0)
        sim.set_state(qpos, qvel, warm, tm)
        sim.physics(ctrl.astype(np.float32), 250)
        orc.physics(ctrl.astype(np.float32).astype(np.float64), 250)
        qg, vg, _, tg = sim.get_state()
        qo, vo, _, to = orc.get_state()
        g.add(qpos, qg, qo)
        assert np.array_equal(tg, to), "time accumulates identically (fp64, 250 additions of h)"
        assert np.isfinite(qg).all() and np.isfinite(vg).all()
    # measured (r03): robot 6.6e-9 / 3.4e-8 / 4.0e-9, block 5.1e-8, fallen 9.8e-6 / 6.3e-5 / 5.6e-8
    g.check(env_id, robot_cap=1e-6, block_cap=1e-6)
    assert g.n["up"] > 0.15 * n * steps, "the rollout must cover upright env-steps"
    assert g.n["fallen"] > (0.05 if env_id == "Env02-v1" else 0.15) * n * steps, "... and robots that stay down"


def _env_step_gates(env_id, n, steps, actions, seed=0, max_skip=0.2):
    """teacher-forced FULL env steps with auto-reset and shared Philox streams (the bench workload's dynamics).
    actions: "zero", "random" (U(-1,1)^2) or "policy" -- the reference's own MuJoCo-trained balance policy
    (tests/quant_policy.py, envs/RobotMovePolicy.tflite) acting on the ORACLE's observations: robots that stay up for whole
    episodes while blocks keep hitting them, the workload a trained policy produces"""
    torch, sim, orc = _mk(env_id, n, seed=seed, auto_reset=True, obs_noise=False)
    sim.reset(); obs_o = orc.reset()
    rng = np.random.default_rng(1234)
    pol = None
    if actions == "policy":
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from quant_policy import QuantMovePolicy
        qp = QuantMovePolicy()
        pol = lambda o: qp.act(torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32)), "mean").numpy()
    g = Gates()
    nskip = 0
    for t in range(steps):
        qpos, qvel, warm, tm = orc.get_state()
        sim.set_state(qpos, qvel, warm, tm); sim.set_aux(orc.get_aux()); sim.set_xpose(*orc.get_xpose())
        if pol is not None:
            act = pol(obs_o)
        else:
            act = np.zeros((n, 2), np.float32) if actions == "zero" else rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        og = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(act).cuda())]
        oo = orc.step(act)
        obs_o = oo[0]
        # a finished episode was re-drawn; a block removed / re-thrown on one side only is a discrete difference
        skip = og[2].astype(bool) | og[3].astype(bool) | oo[2] | oo[3]
        skip |= np.isnan(sim.get_aux()[:, 1]) != np.isnan(orc.get_aux()[:, 1])
        nskip += int(skip.sum())
        g.add(qpos, sim.get_state()[0], orc.get_state()[0], skip)
    sim.close(); orc.close()
    g.skipped = nskip / float(n * steps)
    print(f"{env_id} {actions}: skipped {nskip} of {n * steps} env-steps ({100 * g.skipped:.2f} %: finished episodes, one-sided block removals)")
    assert n < 64 or g.skipped < max_skip, f"skipped share {g.skipped:.3f}"
    return g


def test_config1_env01_v2_single_env():
    """BASELINE config 1: Env01-v2, ONE env (a single lane of a single wave), 200 env steps against the oracle"""
    g = _env_step_gates("Env01-v2", 1, 200, "random", seed=2)
    g.check("config 1 (Env01-v2, N = 1)", robot_cap=1e-7)   # measured 1.1e-9
    assert g.n["up"] > 100


def test_config2_env01_v2_4096_zero_action():
    """BASELINE config 2: Env01-v2, 4,096 envs, zero action -- 100 teacher-forced env steps under pytest (the full 1,000
    steps: tools/parity_report.py -> profiles/r02_parity_report.json)"""
    g = _env_step_gates("Env01-v2", 4096, 100, "zero")
    g.check("config 2 (Env01-v2, 4096 envs, zero action)", robot_cap=2e-6)   # measured 2.2e-7 (round 2: 1.6e-6)


def test_config3_env03_v2_random_policy_reduced():
    """BASELINE config 3 at a size the oracle finishes in a minute: Env03-v2, 2,048 envs x 60 steps, random policy"""
    g = _env_step_gates("Env03-v2", 2048, 60, "random")
    g.check("config 3 reduced (Env03-v2, 2048 envs, random policy)", robot_cap=6e-5, block_cap=3e-5)   # measured 1.3e-5 / 1.4e-6


def test_config3_under_the_reference_balancing_policy():
    """the workload of profiles/r02_parity_config3_policy.json (round 2: 4 robot-coordinate env-steps per million above 1e-4):
    Env03-v2, 1,024 envs x 150 steps with actions from the reference's MuJoCo-trained policy -- robots stay up while blocks
    keep hitting them.  G1 with zero exceptions."""
    g = _env_step_gates("Env03-v2", 1024, 150, "policy", max_skip=0.02)
    g.check("config 3 under the balancing policy (Env03-v2, 1024 envs)", robot_cap=2e-5, block_cap=3e-5)   # measured 1.9e-6 / 3.3e-6
    assert g.n["up"] > 0.95 * 1024 * 150, "under the policy the robots stay upright"


def _constructed(env_id, qpos, qvel, ctrl, nsub):
    """constructed states on the HIP path: set_state -> brs_physics(nsub) against the oracle; relative velocity error per env"""
    from tests import constructed_states as cs
    n = len(qpos)
    torch, sim, orc = _mk(env_id, n, seed=0, auto_reset=False, obs_noise=False)
    orc.set_state(qpos, qvel); sim.set_state(qpos, qvel)
    orc.physics(ctrl, nsub); sim.physics(ctrl.astype(np.float32), nsub)
    (qo, vo, _, _), (qg, vg, _, _) = orc.get_state(), sim.get_state()
    sim.close(); orc.close()
    assert np.isfinite(vg).all()
    return cs.rel_vel_error(vo, vg), vo


def test_constructed_block_robot_contact_states_on_the_hip_path():
    """SURVEY f2 on the device build (sqrt64_ with its v_rsq_f32 seed, device rcp/rsqrt, fast-math contraction of the fp64
    decision code): block against every torso face and both wheels, 5- and 6-point patches plus the wheel point -- the
    states of tests/test_hostsim_parity.py::test_constructed_block_robot_contact_states, 5 substeps via brs_physics"""
    from tests import constructed_states as cs
    qpos, qvel = cs.block_robot_states()
    err, vo = _constructed("Env03-v2", qpos, qvel, np.zeros((len(qpos), 2)), 5)
    assert (np.abs(vo[:, :6]).max(axis=1) > 1e-6).sum() > len(qpos) // 3, "a coupled contact acted on the robot"
    print(f"block<->robot constructed states on HIP: rel. velocity error q98 {np.quantile(err, 0.98):.3g}, max {err.max():.3g}")
    # measured 2.8e-7 / 3.1e-7 (deterministic arithmetic).  The cap on the maximum sits BELOW the 7.0e-7 the first version of the
    # patch-frame algebra reached (relative twist taken at the torso origin: the block's point acceleration as a difference of two
    # large terms) -- the form that put one campaign env-step at 2.4e-4
    assert np.quantile(err, 0.98) < 5e-7 and err.max() < 5e-7, (np.quantile(err, 0.98), err.max())


def test_constructed_edge_edge_states_on_the_hip_path():
    """one-point edge-edge patches between 0.5 mm outside and 1.5 mm inside the margin: existence decided from the fp64 poses"""
    from tests import constructed_states as cs
    qpos, qvel = cs.edge_edge_states()
    err, vo = _constructed("Env03-v2", qpos, qvel, np.zeros((len(qpos), 2)), 5)
    print(f"edge-edge constructed states on HIP: rel. velocity error q95 {np.quantile(err, 0.95):.3g}, max {err.max():.3g}")
    assert np.quantile(err, 0.95) < 2e-6 and err.max() < 5e-6, (np.quantile(err, 0.95), err.max())   # measured 1.1e-7 / 1.5e-7: a point
    # existing on one side only would show as ~1e-2


def test_constructed_floor_contact_states_on_the_hip_path():
    """robot pressed into the floor in every orientation (wheel rim / side / triangle points, torso corners, up to 8 slots)"""
    from tests import constructed_states as cs
    qpos, qvel = cs.floor_states()
    ctrl = np.random.default_rng(5).uniform(-30, 30, size=(len(qpos), 2)).astype(np.float32).astype(np.float64)
    err, vo = _constructed("Env01-v2", qpos, qvel, ctrl, 5)
    print(f"floor constructed states on HIP: rel. velocity error q98 {np.quantile(err, 0.98):.3g}, max {err.max():.3g}")
    assert np.quantile(err, 0.98) < 5e-7 and err.max() < 1e-6, (np.quantile(err, 0.98), err.max())   # measured 2.8e-8 / 3.8e-8


def test_round3_outlier_states_stay_fixed_on_the_hip_path():
    """tests/golden/round3_outlier_states.json: the env-steps round 3's campaigns found above 1e-4 (block quaternion 2.4-2.7e-4),
    replayed on the HIP path (250 fused substeps from the dumped pre-step state) against the oracle"""
    import json
    from balance_robot_mujoco_rl_amd import BatchedSim
    from oracle import oracle as O
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round3_outlier_states.json")))
    for st in fx["states"]:
        pre = st["pre"]
        qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
        tm, ctrl = np.array([pre["time"]]), np.array(pre["ctrl"], dtype=np.float64)
        sim = BatchedSim(st["env"], 1, device=0, seed=0, auto_reset=False, obs_noise=False)
        orc = O.Oracle(st["env"], 1, seed=0, auto_reset=False, noise=False)
        sim.set_state(qpos, qvel, warm, tm); orc.set_state(qpos, qvel, warm, tm)
        sim.physics(ctrl.astype(np.float32)[None], 250); orc.physics(ctrl[None], 250)
        d = float(np.abs(sim.get_state()[0][0] - orc.get_state()[0][0]).max())
        print(f"{st['env']}: {st['why']}: now {d:.3g}")
        assert d < 1e-5, (st["why"], d)   # measured 1e-9 - 2e-8
        sim.close(); orc.close()


def test_config4_per_node_total_on_one_gpu():
    """BASELINE config 4's per-node total (524,288 Env03-v2 envs) in ONE launch on one GPU: size-independent invariants,
    and bit-identity of a 65,536-env shard with a standalone handle over the same global indices (what the 8-GPU run
    relies on: streams keyed by global env index, no cross-env state)"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    n, m, base = 524288, 65536, 3 * 65536
    big = BatchedSim("Env03-v2", n, seed=7, auto_reset=True)
    shard = BatchedSim("Env03-v2", m, seed=7, auto_reset=True, env_index_base=base)
    ob, os_ = big.reset().clone(), shard.reset().clone()
    assert torch.equal(ob[base:base + m], os_)
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    ndone = 0
    for _ in range(20):
        a = (torch.rand((n, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
        o, r, te, tr, to = big.step(a)
        o2, r2, te2, tr2, to2 = shard.step(a[base:base + m].contiguous())
        assert torch.equal(o[base:base + m], o2) and torch.equal(r[base:base + m], r2)
        assert torch.equal(te[base:base + m], te2) and torch.equal(tr[base:base + m], tr2)
        ndone += int((te | tr).sum().item())
    qpos, qvel, _, tm = big.get_state()
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all() and ndone > 0
    assert np.abs(np.linalg.norm(qpos[:, 3:7], axis=1) - 1).max() < 1e-9
    assert (big.get_aux()[:, 7] == 0).all(), "no env hit the bad-state reset"
    big.close(); shard.close()


@pytest.mark.parametrize("env_id", ["Env01-v1", "Env01-v2", "Env03-v1", "Env03-v2", "Env01-v3", "Env02-v1"])
def test_env_step_parity_with_shared_rng(env_id):
    """full env step (reward, obs with noise, termination, block state machine, time limit, auto-reset) against the
    oracle, teacher-forced, with the SAME Philox streams on both sides"""
    n, steps = 256, 40
    torch, sim, orc = _mk(env_id, n, seed=11, auto_reset=True, max_episode_steps=25)
    og = sim.reset().cpu().numpy().copy()
    oo = orc.reset()
    np.testing.assert_allclose(og, oo, atol=2e-5, rtol=1e-5)
    rng = np.random.default_rng(9)
    if env_id == "Env01-v3":  # start near the target-speed schedule's thresholds (1.0, 3.0, 4.5, 5.5 s)
        orc.set_state(time=rng.choice([0.96, 2.96, 4.46, 5.46], size=n) + rng.integers(0, 4, size=n) * 0.005)
    n_done = n_term_disagree = n_timer_disagree = 0
    for t in range(steps):
        qpos, qvel, warm, tm = orc.get_state()
        aux = orc.get_aux(); xq, xp = orc.get_xpose()
        sim.set_state(qpos, qvel, warm, tm); sim.set_aux(aux); sim.set_xpose(xq, xp)
        act = rng.uniform(-1.5, 1.5, size=(n, 2)).astype(np.float32)
        o_g, r_g, te_g, tr_g, to_g = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(act).cuda())]
        o_o, r_o, te_o, tr_o, to_o = orc.step(act)
        np.testing.assert_allclose(r_g, r_o, atol=1e-4, rtol=1e-5)
        # termination can differ only where |pitch| is within rounding of the 50 degree threshold
        agree = (te_g.astype(bool) == te_o)
        n_term_disagree += int((~agree).sum())
        assert np.array_equal(tr_g.astype(bool), tr_o)
        ok = agree
        # obs[1] is a finite difference over 5 ms: fp32 pitch error / 0.005
        np.testing.assert_allclose(to_g[ok][:, [0, 2, 3, 4, 5]], to_o[ok][:, [0, 2, 3, 4, 5]], atol=5e-4, rtol=1e-4)
        np.testing.assert_allclose(to_g[ok][:, 1], to_o[ok][:, 1], atol=5e-3, rtol=1e-3)
        np.testing.assert_allclose(o_g[ok][:, [0, 2, 3, 4, 5]], o_o[ok][:, [0, 2, 3, 4, 5]], atol=5e-4, rtol=1e-4)
        ag, ao = sim.get_aux(), orc.get_aux()
        # the block is removed when |v| < 0.1: an env whose block speed is within rounding of the threshold may decide
        # differently in fp32 (then its timer, and for delay 0 its next throw and RNG counter, differ for this step)
        tsame = np.isnan(ag[:, 1]) == np.isnan(ao[:, 1])
        n_timer_disagree += int((~tsame).sum())
        ok2 = ok & tsame
        assert np.array_equal(ag[ok2][:, 2:5], ao[ok2][:, 2:5]), "elapsed steps, rng counter, attack side"
        tim_g, tim_o = ag[ok2][:, 1], ao[ok2][:, 1]
        assert np.array_equal(tim_g[~np.isnan(tim_g)], tim_o[~np.isnan(tim_o)])
        n_done += int((te_o | tr_o).sum())
    assert n_done > 0, "the test must exercise auto-reset"
    # discrete outcomes decided within rounding of a threshold (|pitch| vs 50 degrees; block speed vs 0.1 m/s) may differ in
    # fp32: at most 2 of the n * steps = 10,240 env-steps each (measured: see profiles/r02_gpu_tests.log)
    print(f"{env_id}: termination disagreements {n_term_disagree}, block-timer disagreements {n_timer_disagree} of {n * steps}")
    assert n_term_disagree <= 2 and n_timer_disagree <= 2


def test_determinism_and_shard_invariance():
    """same seed -> bitwise identical; an env's stream depends on its GLOBAL index only"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    n = 128
    a = torch.rand((n, 2), device="cuda") * 2 - 1

    def run(sims, slices):
        outs = []
        for s in sims:
            s.reset()
        for _ in range(5):
            outs.append(torch.cat([s.step(a[sl].contiguous())[0].clone() for s, sl in zip(sims, slices)]))
        return torch.stack(outs)

    whole = run([BatchedSim("Env03-v2", n, seed=5)], [slice(0, n)])
    again = run([BatchedSim("Env03-v2", n, seed=5)], [slice(0, n)])
    halves = run([BatchedSim("Env03-v2", 64, seed=5, env_index_base=0), BatchedSim("Env03-v2", 64, seed=5, env_index_base=64)],
                 [slice(0, 64), slice(64, 128)])
    other = run([BatchedSim("Env03-v2", n, seed=6)], [slice(0, n)])
    assert torch.equal(whole, again)
    assert torch.equal(whole, halves)
    assert not torch.equal(whole, other)
    # brs_step regroups envs by collision cost class after every step (a lane <-> env permutation): scheduling only
    run_long = lambda grouping: torch.stack([o.clone() for o in _roll(BatchedSim("Env03-v2", 512, seed=5, lane_grouping=grouping), 40)])
    assert torch.equal(run_long(True), run_long(False))


def _roll(sim, steps):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    sim.reset()
    for _ in range(steps):
        yield sim.step(torch.rand((sim.n, 2), generator=g, device="cuda") * 2 - 1)[0]


def test_closed_loop_statistics_under_pd_controller():
    """free-running (NOT teacher-forced) Env03-v2 under a PD balance controller, same seeds on both sides: trajectories
    decorrelate after the first block impact (chaotic contact dynamics), so the comparison is distributional -- the share
    of first episodes still upright after 200 steps (2-3 block impacts) and the mean reward must agree"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    from oracle import oracle as O
    n, steps = 1024, 200
    sim = BatchedSim("Env03-v2", n, device=0, seed=21, auto_reset=True)
    orc = O.Oracle("Env03-v2", n, seed=21, auto_reset=True, threads=min(64, os.cpu_count() or 1))

    def pd(obs):
        pitch, pdot, dv = obs[:, 0] * 0.25, obs[:, 1], (obs[:, 2] - obs[:, 3]) * 0.5 * 170.0 / 4.0
        u = np.clip(20.0 * pitch + 1.0 * pdot - 0.05 * dv, -1, 1)
        return np.stack([-u, u], 1).astype(np.float32)

    og, oo = sim.reset().cpu().numpy().copy(), orc.reset()
    np.testing.assert_allclose(og, oo, atol=2e-5, rtol=1e-5)
    alive_g, alive_o = np.ones(n, bool), np.ones(n, bool)
    rew_g = rew_o = 0.0
    for t in range(steps):
        o_g, r_g, te_g, tr_g, _ = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(pd(og)).cuda())]
        o_o, r_o, te_o, tr_o, _ = orc.step(pd(oo))
        alive_g &= ~te_g.astype(bool); alive_o &= ~te_o.astype(bool)
        rew_g += float(r_g.mean()); rew_o += float(r_o.mean())
        og, oo = o_g, o_o
    fg, fo = alive_g.mean(), alive_o.mean()
    assert 0.3 < fo < 0.999, f"the controller must survive some impacts and fail others to be informative ({fo})"
    assert abs(fg - fo) < 0.06, (fg, fo)
    assert abs(rew_g - rew_o) / steps < 0.05, (rew_g / steps, rew_o / steps)
    sim.close(); orc.close()


@pytest.mark.parametrize("env_id,n", [("Env03-v2", 1), ("Env03-v2", 100), ("Env01-v2", 65), ("Env03-v2", 257)])
def test_ragged_batch_sizes(env_id, n):
    """N that is not a multiple of the 64-lane wave (a partial last wave just masks lanes): reset + env steps vs the oracle"""
    torch, sim, orc = _mk(env_id, n, seed=4, auto_reset=True, obs_noise=False)
    np.testing.assert_allclose(sim.reset().cpu().numpy(), orc.reset(), atol=2e-5, rtol=1e-5)
    rng = np.random.default_rng(2)
    for t in range(12):
        qpos, qvel, warm, tm = orc.get_state()
        sim.set_state(qpos, qvel, warm, tm); sim.set_aux(orc.get_aux()); sim.set_xpose(*orc.get_xpose())
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        out_g = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(act).cuda())]
        out_o = orc.step(act)
        assert out_g[0].shape == (n, 6) and out_g[1].shape == (n,)
        np.testing.assert_allclose(out_g[1], out_o[1], atol=1e-4, rtol=1e-5)
        done = out_g[2].astype(bool) | out_g[3].astype(bool) | out_o[2] | out_o[3]
        qg, qo = sim.get_state()[0], orc.get_state()[0]
        assert np.abs(qg - qo)[~done].max(initial=0.0) < TOL_QPOS                 # G1, G2
    sim.close(); orc.close()


def test_masked_reset_touches_only_the_masked_envs():
    """brs_reset with a device mask (include/brs.h): masked envs are re-drawn from their own streams exactly like the
    oracle's, the others keep their state and their observation rows"""
    n = 192
    torch, sim, orc = _mk("Env03-v2", n, seed=9, auto_reset=False, obs_noise=False)
    sim.reset(); orc.reset()
    act = np.random.default_rng(3).uniform(-1, 1, size=(n, 2)).astype(np.float32)
    for _ in range(3):
        sim.step(torch.from_numpy(act).cuda()); orc.step(act)
    qpos, qvel, warm, tm = orc.get_state()
    sim.set_state(qpos, qvel, warm, tm); sim.set_aux(orc.get_aux()); sim.set_xpose(*orc.get_xpose())
    q_before = sim.get_state()[0].copy()
    mask = (np.arange(n) % 3 == 0)
    og = sim.reset(torch.from_numpy(mask.astype(np.uint8))).cpu().numpy().copy()
    oo = orc.reset(mask.astype(np.uint8))
    qg, qo = sim.get_state()[0], orc.get_state()[0]
    assert np.array_equal(qg[~mask], q_before[~mask]), "unmasked envs keep their state bit for bit"
    np.testing.assert_allclose(qg[mask], qo[mask], atol=1e-6)
    np.testing.assert_allclose(og[mask], oo[mask], atol=2e-5, rtol=1e-5)
    assert np.abs(qg[mask] - q_before[mask]).max() > 1e-3, "masked envs did change"
    sim.close(); orc.close()


def test_pitch_on_the_device_incl_gimbal_lock():
    """get_pitch on the DEVICE build (fast-math, fp64 lock test, v_rcp) against the reference's own get_pitch with the real scipy
    (tests/golden/envlogic.json: pitch_yaw, pitch_yaw_gimbal): one 20-us substep of free fall from the fixture's quaternion, so the
    accessor pose the observation reads IS that quaternion.  Within 1e-7 rad of gimbal lock scipy folds the rotation about the
    vertical into the pitch; the env then terminates or not on that number."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "envlogic.json")))
    cases = g["pitch_yaw"] + [c for c in g["pitch_yaw_gimbal"] if c["yaw"] == 0.0]
    cases = [c for c in cases if c["xquat"][0] != 0.0]
    n = len(cases)
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    sim = BatchedSim("Env01-v1", n, device=0, seed=1, auto_reset=False, obs_noise=False, substeps=1)
    sim.reset()
    qpos = np.zeros((n, 9)); qpos[:, 2] = 1.0
    qpos[:, 3:7] = np.array([c["xquat"] for c in cases])
    sim.set_state(qpos, np.zeros((n, 8)), np.zeros((n, 8)), np.zeros(n))
    obs, _, te, _, _ = sim.step(torch.zeros((n, 2), device="cuda"))
    torch.cuda.synchronize()
    pitch = obs.cpu().numpy()[:, 0] * 0.25
    want = np.array([c["pitch"] for c in cases])
    d = np.abs((pitch - want + np.pi) % (2 * np.pi) - np.pi)
    locked = np.array([c in g["pitch_yaw_gimbal"] for c in cases])
    print(f"device pitch vs scipy: max |d| {d.max():.3g} over {n} poses ({int(locked.sum())} at gimbal lock)")
    assert d.max() < 5e-6, (int(d.argmax()), cases[int(d.argmax())], float(pitch[d.argmax()]))
    lim = 50.0 * np.pi / 180.0
    clear = np.abs(np.abs(want) - lim) > 1e-4
    assert np.array_equal(te.cpu().numpy().astype(bool)[clear], (np.abs(want) > lim)[clear])
    sim.close()


def test_full_size_properties():
    """BASELINE size (65,536 x Env03-v2): size-independent invariants after a random-policy rollout"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    n = 65536
    sim = BatchedSim("Env03-v2", n, seed=1, auto_reset=True)
    sim.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    ndone = 0
    for _ in range(60):
        o, r, te, tr, to = sim.step(torch.rand((n, 2), generator=gen, device="cuda") * 2 - 1)
        ndone += int((te | tr).sum().item())
    qpos, qvel, warm, tm = sim.get_state()
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all() and np.isfinite(o.cpu().numpy()).all()
    assert np.abs(np.linalg.norm(qpos[:, 3:7], axis=1) - 1).max() < 1e-9, "torso quaternion stays unit (fp64 accumulator)"
    assert np.abs(np.linalg.norm(qpos[:, 12:16], axis=1) - 1).max() < 1e-9
    assert (qpos[:, 2] > -0.06).all(), "nothing sinks through the floor"
    aux = sim.get_aux()
    assert (aux[:, 7] == 0).all(), "no env hit the bad-state reset"
    assert ndone > 0 and (aux[:, 2] <= 1200).all()
    # rewards bounded by construction: 1 - penalties, |pitch| < ~pi
    assert r.max().item() <= 1.0 + 0.5 * np.pi * 80 and r.min().item() > -200


def test_errors_are_loud():
    from balance_robot_mujoco_rl_amd import BatchedSim, BrsError
    with pytest.raises(KeyError):
        BatchedSim("Env99-v0", 4)
    with pytest.raises(BrsError):
        BatchedSim("Env01-v2", 0)
    s = BatchedSim("Env01-v2", 8)
    with pytest.raises(ValueError):
        s.step(np.zeros((7, 2), np.float32))
    s.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_vec_env_on_the_gpu(devices):
    """BalanceVecEnv over real handles: one shard, and two shards of the SAME GPU on their own streams (overlapping
    kernels); both must give the env-index-keyed results of a single handle"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim, make_vec
    n = 256
    env = make_vec("Env03-v2", n, devices=devices, seed=13)
    ref = BatchedSim("Env03-v2", n, device=0, seed=13, auto_reset=True)
    o = env.reset()
    np.testing.assert_array_equal(o, ref.reset().cpu().numpy())
    rng = np.random.default_rng(0)
    ndone = 0
    for _ in range(40):
        a = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
        ro, rr, rte, rtr, rto = [x.cpu().numpy() for x in ref.step(torch.from_numpy(a).cuda())]
        np.testing.assert_array_equal(obs, ro); np.testing.assert_array_equal(rew, rr)
        np.testing.assert_array_equal(dones, (rte | rtr).astype(bool))
        # eager arrays of the finished episodes, then the SB3-style dicts built from them on first access
        np.testing.assert_array_equal(infos.done_indices, np.flatnonzero(dones))
        np.testing.assert_array_equal(infos.terminal_observations, rto[dones])
        for i in np.flatnonzero(dones):
            np.testing.assert_array_equal(infos[i]["terminal_observation"], rto[i]); ndone += 1
            assert infos[i]["TimeLimit.truncated"] == bool(rtr[i] and not rte[i]) and infos[i]["episode"]["l"] >= 1
        assert infos[int(np.flatnonzero(~dones)[0])] == {}
    assert ndone > 0
    env.close(); ref.close()


def test_gymnasium_vector_adapter_on_the_gpu():
    """BalanceVectorEnv (Gymnasium vector API) over a real handle: 5-tuple step, same-step auto-reset,
    infos["final_observation"] / ["_final_observation"] for the finished envs -- against a plain BatchedSim"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    from balance_robot_mujoco_rl_amd.vec_env import BalanceVectorEnv
    n = 192
    env = BalanceVectorEnv("Env03-v2", n, devices=[0], seed=5)
    ref = BatchedSim("Env03-v2", n, device=0, seed=5, auto_reset=True)
    obs, infos = env.reset(seed=5)
    assert infos == {}
    np.testing.assert_array_equal(obs, ref.reset().cpu().numpy())
    rng = np.random.default_rng(1)
    nfinal = 0
    for _ in range(40):
        a = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        obs, rew, term, trunc, infos = env.step(a)
        ro, rr, rte, rtr, rto = [x.cpu().numpy() for x in ref.step(torch.from_numpy(a).cuda())]
        np.testing.assert_array_equal(obs, ro); np.testing.assert_array_equal(rew, rr)
        np.testing.assert_array_equal(term, rte.astype(bool)); np.testing.assert_array_equal(trunc, rtr.astype(bool))
        done = term | trunc
        if done.any():
            np.testing.assert_array_equal(infos["_final_observation"], done)
            for i in np.flatnonzero(done):
                np.testing.assert_array_equal(infos["final_observation"][i], rto[i]); nfinal += 1
        else:
            assert "final_observation" not in infos
    assert nfinal > 0
    env.close(); ref.close()


@pytest.mark.parametrize("env_id", ["Env03-v2", "Env01-v2"])
def test_runtime_parameter_kernel_with_non_default_timestep(env_id):
    """brs_config.timestep / .substeps other than the reference's 2e-5 s x 250: the handle runs the kernel whose model
    constants are kernel ARGUMENTS (the per-id kernels fold the default ones at compile time).  Same gates vs the oracle
    created with the same settings (100 substeps of 5e-5 s = the same 5 ms env step)"""
    n, steps = 256, 60
    torch, sim, orc = _mk(env_id, n, seed=2, auto_reset=True, obs_noise=False, substeps=100, timestep=5e-5)
    assert "-1>" in sim.step_kernel_name(), sim.step_kernel_name()
    sim.reset(); orc.reset()
    rng = np.random.default_rng(7)
    g = Gates()
    for t in range(steps):
        qpos, qvel, warm, tm = orc.get_state()
        sim.set_state(qpos, qvel, warm, tm); sim.set_aux(orc.get_aux()); sim.set_xpose(*orc.get_xpose())
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        og = [x.cpu().numpy().copy() for x in sim.step(torch.from_numpy(act).cuda())]
        oo = orc.step(act)
        skip = og[2].astype(bool) | og[3].astype(bool) | oo[2] | oo[3]
        skip |= np.isnan(sim.get_aux()[:, 1]) != np.isnan(orc.get_aux()[:, 1])
        g.add(qpos, sim.get_state()[0], orc.get_state()[0], skip)
        assert np.array_equal(sim.get_state()[3][~skip], orc.get_state()[3][~skip]), "time = 100 fp64 additions of 5e-5"
    g.check(env_id + " (5e-5 s x 100)")
    sim.close(); orc.close()


def test_folded_and_runtime_constant_kernels_agree(monkeypatch):
    """the same default model through both kernel families: constants folded at compile time (default) and passed as kernel
    arguments (BRS_NO_FOLD=1).  Same arithmetic on the same constants up to how the compiler contracts literal operands:
    trajectories agree to rounding"""
    import torch
    from balance_robot_mujoco_rl_amd import BatchedSim
    n = 512
    a = BatchedSim("Env03-v2", n, seed=9, auto_reset=True)
    monkeypatch.setenv("BRS_NO_FOLD", "1")
    b = BatchedSim("Env03-v2", n, seed=9, auto_reset=True)
    monkeypatch.delenv("BRS_NO_FOLD")
    assert a.step_kernel_name() != b.step_kernel_name()
    np.testing.assert_array_equal(a.reset().cpu().numpy(), b.reset().cpu().numpy())
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    worst = 0.0
    for _ in range(20):
        act = torch.rand((n, 2), generator=gen, device="cuda") * 2 - 1
        qa = a.get_state(); b.set_state(*qa); b.set_aux(a.get_aux()); b.set_xpose(*a.get_xpose())
        oa = [x.cpu().numpy().copy() for x in a.step(act)]
        ob = [x.cpu().numpy().copy() for x in b.step(act)]
        keep = ~(oa[2].astype(bool) | oa[3].astype(bool) | ob[2].astype(bool) | ob[3].astype(bool))
        assert int((oa[2] != ob[2]).sum()) <= 1
        worst = max(worst, float(np.abs(a.get_state()[0][keep, :9] - b.get_state()[0][keep, :9]).max()))
    assert worst < 1e-5, worst
    a.close(); b.close()

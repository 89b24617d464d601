"""Kernel source (host build) vs the oracle, teacher-forced per env step on states from oracle rollouts.

double instantiation: the kernel's closed forms (gyrostat mass matrix in body coordinates, analytic M^-1, world-aligned
floor contact frame, own Newton solver) against the oracle's independent body-tree / numeric-Jacobian / MuJoCo-style
solver -- agreement at ~1e-9 validates both derivations.  float instantiation: the arithmetic the GPU runs."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.hostsim.hostsim import HostSim
from tests import constructed_states as cs


def _rollout_parity(env_id, n, steps, double, seed=11):
    rng = np.random.default_rng(7)
    o = O.Oracle(env_id, n, seed=seed, noise=False, threads=8)
    h = HostSim(env_id, n, seed=seed, noise=False, double=double)
    o.reset()
    h.set_aux(o.get_aux())  # per-episode scalars (Env02's friction) live in aux
    dqs, dvs = [], []
    for t in range(steps):
        qpos, qvel, warm, tm = o.get_state()
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        ctrl = qvel[:, 6:8] + act * 4.0
        h.set_state(qpos, qvel, warm, tm)
        o.physics(ctrl, 250)
        h.physics(ctrl, 250)
        a, b = o.get_state(), h.get_state()
        dqs.append(np.abs(a[0] - b[0]).max(axis=1)); dvs.append(np.abs(a[1] - b[1]).max(axis=1))
        assert np.array_equal(a[3], b[3])
    return np.array(dqs).ravel(), np.array(dvs).ravel()


@pytest.mark.parametrize("env_id,steps", [("Env01-v2", 70), ("Env03-v2", 60), ("Env02-v1", 50)])
def test_double_instantiation_matches_oracle(env_id, steps):
    dq, dv = _rollout_parity(env_id, 16, steps, True)
    assert dq.max() < 1e-7 and dv.max() < 1e-4, (dq.max(), dv.max())
    assert np.median(dq) < 1e-12


@pytest.mark.parametrize("env_id,steps", [("Env01-v2", 70), ("Env03-v2", 60)])
def test_float_instantiation_within_tolerance(env_id, steps):
    """north-star tolerance: per-step |dqpos| < 1e-4 (contact-onset substeps may land one substep apart in fp32)"""
    dq, dv = _rollout_parity(env_id, 16, steps, False)
    assert np.quantile(dq, 0.999) < 1e-4, np.quantile(dq, 0.999)
    assert dq.max() < 2e-3 and np.median(dq) < 1e-6


def test_free_run_env_steps_match_with_shared_rng():
    """full env steps incl. auto-reset with the same Philox streams: discrete outcomes identical for a while"""
    n = 8
    o = O.Oracle("Env03-v2", n, seed=3, auto_reset=True, max_episode_steps=20)
    h = HostSim("Env03-v2", n, seed=3, auto_reset=True, max_episode_steps=20, double=True)
    np.testing.assert_allclose(o.reset(), h.reset(), atol=1e-6)
    rng = np.random.default_rng(1)
    for t in range(30):
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        oo, ro, teo, tro, too = o.step(act)
        oh, rh, teh, trh, toh = h.step(act)
        np.testing.assert_allclose(oo, oh, atol=2e-4)
        np.testing.assert_allclose(ro, rh, atol=1e-5)
        assert np.array_equal(teo, teh) and np.array_equal(tro, trh)
        assert np.array_equal(o.get_aux()[:, 2:5], h.get_aux()[:, 2:5])


@pytest.mark.parametrize("double,tol", [(True, 1e-12), (False, 1e-5)])
def test_constructed_block_robot_contact_states(double, tol):
    """block placed (random pose, random approach velocity) against the torso faces and the wheels of an airborne robot:
    the kernel source and the oracle must generate the same contacts and the same impulses over 5 substeps -- the coupled
    path on far more configurations than a rollout visits (the same states on the HIP path: tests/test_gpu_parity.py)"""
    qpos, qvel = cs.block_robot_states()
    n = len(qpos)
    o = O.Oracle("Env03-v2", n, noise=False, threads=8)
    h = HostSim("Env03-v2", n, noise=False, double=double)
    o.set_state(qpos, qvel); h.set_state(qpos, qvel)
    # the kernel holds 7 block<->robot slots (6 patch points + the wheel point): whatever the generator emits fits
    ctrl = np.zeros((n, 2))
    most = 0
    for _ in range(5):
        most = max(most, int(cs.coupled_contact_count(o, n).max()))
        o.physics(ctrl, 1); h.physics(ctrl, 1)
    assert 5 <= most <= 7, most   # the states must exercise more than the 4 slots of round 1
    vo, vh = o.get_state()[1], h.get_state()[1]
    touched = np.abs(vo[:, :6]).max(axis=1) > 1e-6          # the robot was pushed: a coupled contact acted
    assert touched.sum() > n // 3
    err = cs.rel_vel_error(vo, vh)
    assert np.quantile(err, 0.98) < tol and err.max() < 50 * tol, (np.quantile(err, 0.98), err.max())


@pytest.mark.parametrize("double,tol", [(True, 1e-12), (False, 1e-5)])
def test_constructed_edge_edge_states(double, tol):
    """block pushed edge-first against a vertical torso edge, between 0.5 mm outside and 1.5 mm inside the margin: the patch is
    ONE point and whether it exists is the edge-axis separation against the margin (decided from the fp64 poses in the float
    build as well)"""
    qpos, qvel = cs.edge_edge_states()
    n = len(qpos)
    o = O.Oracle("Env03-v2", n, noise=False, threads=8)
    h = HostSim("Env03-v2", n, noise=False, double=double)
    o.set_state(qpos, qvel); h.set_state(qpos, qvel)
    codes = [O.box_box_points(cs.TS, cs.BS, qpos[i, 9:12] - np.array([0, 0, 1.0]) - cs.TC, cs.quat_to_mat(qpos[i, 12:16]), 0.002)[3] for i in range(n)]
    assert sum(c >= 6 for c in codes) > n // 2 and sum(c < 0 for c in codes) > 4, "edge-pair contacts and near misses"
    ctrl = np.zeros((n, 2))
    o.physics(ctrl, 5); h.physics(ctrl, 5)
    vo, vh = o.get_state()[1], h.get_state()[1]
    err = cs.rel_vel_error(vo, vh)
    assert err.max() < 50 * tol and np.quantile(err, 0.95) < tol, (np.quantile(err, 0.95), err.max())


@pytest.mark.parametrize("double,tol", [(True, 1e-12), (False, 5e-6)])
def test_constructed_floor_contact_states(double, tol):
    """robot in random orientations (upright, on a wheel's side, on the torso, upside down) pressed 0..3 mm into the floor
    with random velocities: wheel rim / side / triangle points and torso corners, up to the 8-slot capacity"""
    qpos, qvel = cs.floor_states()
    n = len(qpos)
    o = O.Oracle("Env01-v2", n, noise=False, threads=8)
    h = HostSim("Env01-v2", n, noise=False, double=double)
    o.set_state(qpos, qvel); h.set_state(qpos, qvel)
    ncon = np.array([o.forward(env=i)["ncon"] for i in range(n)])
    assert ncon.min() >= 1 and ncon.max() >= 6, (ncon.min(), ncon.max())
    ctrl = np.random.default_rng(5).uniform(-30, 30, size=(n, 2))
    o.physics(ctrl, 5); h.physics(ctrl, 5)
    vo, vh = o.get_state()[1], h.get_state()[1]
    err = cs.rel_vel_error(vo, vh)
    assert np.quantile(err, 0.98) < tol and err.max() < 50 * tol, (np.quantile(err, 0.98), err.max())


@pytest.mark.parametrize("env_id,n,steps", [("Env03-v2", 384, 90), ("Env01-v2", 384, 90)])
def test_float_build_stays_on_the_double_build_over_full_env_steps(env_id, n, steps):
    """kernel source in float vs in double, teacher-forced over full env steps with auto-reset (the bench workload's dynamics):
    the distances that decide whether a contact point exists are taken from the fp64 poses.  The same A/B on
    the CPU (this library built with -DBRS_FLOOR_DIST32 -DBRS_PATCH_DIST32, 2,048 envs x 150 steps, ~300 k env-steps):
    env-steps above 1e-5: Env03-v2 9 -> 2, Env01-v2 69 -> 13; above 1e-6: 27 -> 6 and 602 -> 76; maximum 9.9e-5 -> 2.1e-5 and
    5.2e-5 -> 3.3e-5."""
    rng = np.random.default_rng(3)
    D = HostSim(env_id, n, seed=4, auto_reset=True, noise=False, double=True, threads=8)
    F = HostSim(env_id, n, seed=4, auto_reset=True, noise=False, double=False, threads=8)
    D.reset(); F.reset()
    worst, over5, kept = 0.0, 0, 0
    for _ in range(steps):
        qpos, qvel, warm, tm = D.get_state()
        F.set_state(qpos, qvel, warm, tm); F.set_aux(D.get_aux()); F.set_xpose(*D.get_xpose())
        act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
        od, of = D.step(act), F.step(act)
        skip = od[2] | od[3] | of[2] | of[3]
        skip |= np.isnan(D.get_aux()[:, 1]) != np.isnan(F.get_aux()[:, 1])
        e = np.abs(D.get_state()[0] - F.get_state()[0]).max(axis=1)[~skip]
        worst = max(worst, float(e.max())); over5 += int((e > 1e-5).sum()); kept += int(e.size)
    assert kept > 0.9 * n * steps
    assert worst < 5e-5, worst
    assert over5 <= 3, (over5, kept)


def test_round3_outlier_states_stay_fixed():
    """the env-steps round 3's parity campaigns found above 1e-4 (block quaternion 2.4-2.7e-4) and their causes -- the patch twist
    taken at the torso origin, an edge-axis length lost to cancellation, a discrete axis choice on fp32 roundings -- as regression
    fixtures: the kernel source in FLOAT stays on its DOUBLE instantiation over the 250 substeps of that step (fourth state: the
    wheel<->block contact existence, found by seed 5)"""
    import json, os
    from tests.hostsim.hostsim import HostSim
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round3_outlier_states.json")))
    for st in fx["states"]:
        pre = st["pre"]
        qpos, qvel, warm = (np.array(pre[k])[None] for k in ("qpos", "qvel", "warm"))
        res = []
        for dbl in (True, False):
            h = HostSim(st["env"], 1, noise=False, double=dbl)
            h.set_state(qpos, qvel, warm, np.array([pre["time"]]))
            h.physics(np.array(pre["ctrl"])[None], 250)
            res.append(h.get_state()[0][0])
        d = np.abs(res[0] - res[1]).max()
        assert d < 2e-6, (st["why"], d)   # measured <= 3e-8; the three were 2.4e-4 - 2.7e-4 (the first one on the GPU only)

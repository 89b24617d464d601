"""Kernel source (host build) vs the oracle, teacher-forced per env step on states from oracle rollouts.

double instantiation: the kernel's closed forms (gyrostat mass matrix in body coordinates, analytic M^-1, world-aligned
floor contact frame, own Newton solver) against the oracle's independent body-tree / numeric-Jacobian / MuJoCo-style
solver -- agreement at ~1e-9 validates both derivations.  float instantiation: the arithmetic the GPU runs."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.hostsim.hostsim import HostSim


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

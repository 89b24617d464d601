"""Cal01 known-answer test: the ONE deterministic physics scenario the reference ships
(/root/reference/src/balance_robot/envs/cal01.py:15-55).

Cal01 puts the robot upside-down at z = 0.15 (euler 'xyz' (0, 0, pi) written as (x, y, z, w) into MuJoCo's (w, x, y, z)
slot, cal01.py:41-51 -> a half turn about y), holds ctrl = 20 on both wheel servos (cal01.py:19-20) and prints the wheel
rates.  The wheels are in the air; their hinge axes are (-1, 0, 0) and (+1, 0, 0) (robot-02.xml:9-18), so equal rates give
opposite reaction torques on the torso: the torso does not turn about x and each wheel obeys the scalar servo law
    I w' = clamp(kv (u - w), +-0.65) - 0.01 w ,   kv = 4, I = m r^2 / 2 of the wheel cylinder (robot-02.xml:11-24).
Known answers, independent of oracle/ and of the kernel:
  * the force-limited phase follows w(t) = 65 (1 - exp(-t 0.01 / I)) until 4 (20 - w) < 0.65, i.e. w > 19.8375 at t = 1.99 ms;
  * from then on the stiff servo (time constant I / 4.01 = 13.6 us < h) settles on w* = 80 / 4.01 = 19.950125 rad/s, which
    implicitfast must neither overshoot nor bias;
  * substep by substep, MuJoCo's implicitfast recurrence  (I + h (0.01 + kv [force not clamped])) a = f - 0.01 w.
This pins SURVEY a2.5 (passive damping), a2.6 (actuation and both clamps) and a2.8 (implicitfast) on the oracle, on the host
build of the kernel source, and (marked gpu) on the HIP path through the C ABI."""
import math

import numpy as np
import pytest

H = 2e-5
R_WHEEL, HL_WHEEL, DENSITY = 0.034, 0.013, 1000.0        # robot-02.xml:12,17 ; MuJoCo default density (inertiafromgeom)
I_AXIAL = 0.5 * (math.pi * R_WHEEL ** 2 * 2 * HL_WHEEL * DENSITY) * R_WHEEL ** 2
KV, DAMP, FMAX, CTRL = 4.0, 0.01, 0.65, 20.0             # robot-02.xml:11,16,22-25 ; cal01.py:19-20


def cal01_state(n=1):
    qpos = np.zeros((n, 9)); qpos[:, 2] = 0.15
    qpos[:, 3:7] = [0.0, 0.0, 1.0, 6.123233995736766e-17]  # scipy as_quat() of euler xyz (0, 0, pi), slot order as in cal01.py:51
    return qpos, np.zeros((n, 8))


def recurrence(nsub):
    w = 0.0
    out = []
    for _ in range(nsub):
        f = KV * (CTRL - w)
        clamped = abs(f) >= FMAX
        f = max(-FMAX, min(FMAX, f))
        a = (f - DAMP * w) / (I_AXIAL + H * (DAMP + (0.0 if clamped else KV)))
        w += H * a
        out.append(w)
    return np.array(out)


def _run(sim, rtol_rec, rtol_ss):
    qpos, qvel = cal01_state(sim.n)
    sim.set_state(qpos, qvel, np.zeros((sim.n, 8)), np.zeros(sim.n))
    ctrl = np.full((sim.n, 2), CTRL)
    rec = recurrence(250)
    done = 0
    for k in (25, 50, 75, 95, 110, 150, 250):              # inside the first env step: force-limited, then the stiff servo
        sim.physics(ctrl, k - done); done = k
        v = sim.get_state()[1]
        assert np.allclose(v[:, 6], v[:, 7], rtol=rtol_rec, atol=1e-9), "both wheels the same rate (cal01.py:26-27)"
        np.testing.assert_allclose(v[:, 6], rec[k - 1], rtol=rtol_rec)
        t = k * H
        if rec[k - 1] < 19.8:                              # continuous law of the force-limited phase (discretisation: O(h / tau))
            np.testing.assert_allclose(v[:, 6], 65.0 * (1 - math.exp(-t * DAMP / I_AXIAL)), rtol=3e-3)
        assert np.abs(v[:, 3]).max() < 1e-4, "reaction torques cancel: no pitch rate"
    assert rec[-1] == pytest.approx(KV * CTRL / (KV + DAMP), rel=1e-12)
    for step in range(2, 41):                              # cal01.py terminates at time > 1.0; 0.2 s is plenty: steady state
        sim.physics(ctrl, 250)
        v = sim.get_state()[1]
        np.testing.assert_allclose(v[:, 6:8], KV * CTRL / (KV + DAMP), rtol=rtol_ss)
    q = sim.get_state()[0]
    assert np.abs(np.abs(q[:, 5]) - 1.0).max() < 1e-3, "still standing on its head"
    # resting on the torso's top face: 4 corner contacts at the static penetration of the default solref/solimp
    assert -0.02 - 0.004 < (q[:, 2] - 0.185).max() < -0.02 + 1e-4


def test_recurrence_matches_the_closed_forms():
    rec = recurrence(250)
    k = int(np.argmax(rec > 19.8375))
    assert abs((k + 1) * H - (-I_AXIAL / DAMP) * math.log(1 - 19.8375 / 65.0)) < 3 * H
    assert I_AXIAL == pytest.approx(5.4577e-5, rel=1e-4)


def test_cal01_oracle():
    from oracle import oracle as O
    _run(O.Oracle("Env01-v1", 1, noise=False), 1e-9, 1e-9)


@pytest.mark.parametrize("double", [True, False])
def test_cal01_kernel_source_on_host(double):
    from tests.hostsim.hostsim import HostSim
    _run(HostSim("Env01-v1", 2, noise=False, double=double), 1e-9 if double else 2e-5, 1e-9 if double else 2e-6)


@pytest.mark.gpu
def test_cal01_hip_path():
    from balance_robot_mujoco_rl_amd import BatchedSim
    sim = BatchedSim("Env01-v1", 64, device=0, auto_reset=False, obs_noise=False)
    _run(sim, 5e-5, 5e-6)
    sim.close()

"""Known-answer and invariant tests of the oracle's physics (the half no reference fixture pins: MuJoCo is absent).

Model constants are checked against the values SURVEY.md App. A derives from the XML; dynamics against analytic
results: free fall, torque-free energy conservation, static equilibrium penetration, mirror symmetry, actuator law."""
import numpy as np
import pytest

from oracle import oracle as O

Q0 = np.array([[0, 0, 0, 1, 0, 0, 0, 0, 0.0]])


def test_model_constants():
    mi = O.Oracle("Env03-v2", 1).model_info()
    np.testing.assert_allclose(mi["body_mass"][1:], [0.6327, 0.094424, 0.094424, 0.064], rtol=2e-5)
    np.testing.assert_allclose(mi["body_inertia"][1], [1.614e-3, 2.069e-3, 5.99e-4], rtol=1e-3)
    np.testing.assert_allclose(mi["body_inertia"][2], [5.4577e-5, 3.2608e-5, 3.2608e-5], rtol=1e-4)
    np.testing.assert_allclose(mi["body_inertia"][4], [1.7067e-5] * 3, rtol=1e-4)
    np.testing.assert_allclose(mi["invweight0"][1:, 0], [1.27091, 3.37572, 3.37572, 15.625], rtol=1e-5)
    np.testing.assert_allclose(mi["invweight0"][1:, 1], [433.068, 6391.71, 6391.71, 58593.75], rtol=1e-5)
    assert abs(O.Oracle("Env01-v2", 1).model_info()["meaninertia"] - 0.31054) < 1e-5


def test_mass_matrix_at_qpos0():
    o = O.Oracle("Env01-v2", 1)
    o.set_state(Q0, np.zeros((1, 8)))
    M = o.forward()["M"]
    np.testing.assert_allclose(np.diag(M), [0.82155, 0.82155, 0.82155, 8.2053e-3, 9.6505e-3, 1.6988e-3, 5.4577e-5, 5.4577e-5], rtol=1e-4)
    assert abs(M[0, 4] - 0.069374) < 1e-6 and abs(M[1, 3] + 0.069374) < 1e-6
    assert abs(M[3, 6] + 5.4577e-5) < 1e-9 and abs(M[3, 7] - 5.4577e-5) < 1e-9
    np.testing.assert_allclose(M, M.T, atol=1e-18)


def test_free_fall_two_centimetres():
    """every episode starts with wheels 2 cm above the floor: z(t) = -g t^2/2 until first touch at t = sqrt(2*0.02/g)"""
    o = O.Oracle("Env01-v2", 1)
    o.set_state(Q0, np.zeros((1, 8)))
    t_touch = np.sqrt(2 * 0.02 / 9.81)
    n = int(0.8 * t_touch / 2e-5)
    o.physics(np.zeros((1, 2)), n)
    q, v, _, t = o.get_state()
    assert abs(v[0, 2] + 9.81 * t[0]) < 1e-9
    assert abs(q[0, 2] + 0.5 * 9.81 * t[0] * (t[0] + 2e-5)) < 1e-9  # semi-implicit Euler: v first, then x
    assert o.forward()["ncon"] == 0


def test_energy_conserved_in_free_flight():
    """torque-free gyrostat (no contact, servos at their own speed -> zero torque, damping still acts): total energy
    may only decrease, by no more than the wheel-damping work, and to O(h) conserved otherwise"""
    rng = np.random.default_rng(0)
    o = O.Oracle("Env01-v2", 1)
    q = Q0.copy(); q[0, 2] = 5.0
    qq = rng.normal(size=4); q[0, 3:7] = qq / np.linalg.norm(qq)
    v = np.zeros((1, 8)); v[0, :3] = rng.normal(size=3); v[0, 3:6] = rng.normal(size=3) * 3
    o.set_state(q, v)
    f0 = o.forward(ctrl=(0.0, 0.0))
    e0 = f0["energy_kin"] + f0["energy_pot"]
    for _ in range(20):
        _, vv, _, _ = o.get_state()
        o.physics(vv[:, 6:8].copy(), 50)  # ctrl = wheel speed -> servo force 0 at the start of each chunk
    f1 = o.forward()
    e1 = f1["energy_kin"] + f1["energy_pot"]
    assert abs(e1 - e0) < 2e-4 * abs(e0)


def test_angular_momentum_direction_free_flight():
    """zero gravity is not configurable; instead: spin about the symmetry-free axes must keep |L| (world) constant"""
    o = O.Oracle("Env01-v2", 1)
    q = Q0.copy(); q[0, 2] = 3.0
    v = np.zeros((1, 8)); v[0, 3:6] = [2.0, 1.0, 3.0]; v[0, 6:8] = [30.0, -20.0]
    o.set_state(q, v)

    def L_world():
        f = o.forward()
        qq, vv, _, _ = o.get_state()
        p = f["M"] @ vv[0]
        w, x, y, z = qq[0, 3:7]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return R @ p[3:6] - np.cross(p[0:3], qq[0, 0:3]) * 0 + np.cross(qq[0, 0:3], p[0:3]) * 0, p, R

    # generalized momentum of the body-frame angular dofs is the angular momentum about the torso origin in body axes;
    # about the COM in world axes it is conserved without external torque (gravity acts at the COM)
    def L_com():
        f = o.forward()
        qq, vv, _, _ = o.get_state()
        p = f["M"] @ vv[0]
        _, _, R = L_world()
        c_world = R @ np.array([0, 0, 0.08444])
        return R @ p[3:6] - np.cross(c_world, p[0:3])

    l0 = L_com()
    for _ in range(10):
        _, vv, _, _ = o.get_state()
        o.physics(vv[:, 6:8].copy(), 100)
    l1 = L_com()
    assert np.linalg.norm(l1 - l0) < 2e-3 * np.linalg.norm(l0)


def test_static_equilibrium_penetration():
    """upright robot at rest on 4 wheel contacts (Env01 pairs: d = 0.5, K = 40000, mu = .9): each of the 16 pyramid rows
    carries m g / 16; row force = D * (K d pen)  =>  pen = m g / (16 D K d)"""
    o = O.Oracle("Env01-v2", 1)
    o.set_state(Q0, np.zeros((1, 8)))
    o.physics(np.zeros((1, 2)), 250 * 60)
    f = o.forward()
    assert f["ncon"] == 4 and f["nefc"] == 16
    m = 0.82155
    D = 1.0 / (2 * 0.81 * (1 - 0.5) / 0.5 * 3.37572 * (1 + 0.81))
    pen = m * 9.81 / (16 * D * 40000 * 0.5)
    d = np.array([c["dist"] for c in f["contacts"]])
    np.testing.assert_allclose(-d, pen, rtol=2e-3)
    np.testing.assert_allclose(f["efc_force"], m * 9.81 / 16, rtol=2e-3)
    assert f["contacts"][0]["frame"].tolist() == [[0, 0, 1], [0, 1, 0], [-1, 0, 0]]
    xs = sorted(round(c["pos"][0], 6) for c in f["contacts"])
    assert xs == [-0.087, -0.061, 0.061, 0.087]


def test_left_right_mirror_symmetry():
    """mirroring the state in the x = 0 plane (swap wheels) must mirror the motion"""
    rng = np.random.default_rng(2)
    a, b = O.Oracle("Env01-v2", 1), O.Oracle("Env01-v2", 1)
    q = Q0.copy(); q[0, 2] = 0.0
    roll, pitch = 0.05, 0.1  # rotation about y (roll) and x (pitch), small
    cy, sy, cx, sx = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    quat = np.array([cx * cy, sx * cy, cx * sy, -sx * sy])
    qa = q.copy(); qa[0, 3:7] = quat
    qb = q.copy(); qb[0, 3:7] = quat * np.array([1, 1, -1, -1])  # mirror x -> -x: flips rotations about y and z
    va = np.zeros((1, 8)); va[0, :3] = [0.1, 0.2, 0]; va[0, 3:6] = [0.5, 0.2, -0.3]; va[0, 6:8] = [5.0, -7.0]
    vb = va.copy(); vb[0, 0] *= -1; vb[0, 4] *= -1; vb[0, 5] *= -1
    # hinge axes are (-1,0,0) left and (+1,0,0) right; angular velocity is a pseudo-vector (x component survives the
    # mirror): the image of the left wheel's -w_L e_x is the right wheel's (+e_x) rate -w_L  => swap AND negate
    vb[0, 6], vb[0, 7] = -va[0, 7], -va[0, 6]
    a.set_state(qa, va); b.set_state(qb, vb)
    ctrl_a = np.array([[3.0, -4.0]]); ctrl_b = np.array([[4.0, -3.0]])
    a.physics(ctrl_a, 2500); b.physics(ctrl_b, 2500)
    (qa1, va1, _, _), (qb1, vb1, _, _) = a.get_state(), b.get_state()
    assert abs(qa1[0, 0] + qb1[0, 0]) < 1e-9 and abs(qa1[0, 1] - qb1[0, 1]) < 1e-9 and abs(qa1[0, 2] - qb1[0, 2]) < 1e-9
    np.testing.assert_allclose(qa1[0, 3:7] * np.array([1, 1, -1, -1]), qb1[0, 3:7], atol=1e-9)
    assert abs(va1[0, 6] + vb1[0, 7]) < 1e-7 and abs(va1[0, 7] + vb1[0, 6]) < 1e-7


def test_velocity_servo_law_and_clamps():
    """Cal01-style spin-up in the air (cal01.py:19-32): wheels free, I_eff w' = clamp(4 (clamp(u) - w), +-0.65) - 0.01 w"""
    o = O.Oracle("Env01-v2", 1)
    q = Q0.copy(); q[0, 2] = 50.0
    o.set_state(q, np.zeros((1, 8)))
    f = o.forward(ctrl=(20.0, 100.0))
    assert f["actuator"][6] == 0.65 and f["actuator"][7] == 0.65  # 4*20 and 4*78.54 both exceed the force limit
    f = o.forward(ctrl=(0.1, -0.05))
    np.testing.assert_allclose(f["actuator"][6:8], [0.4, -0.2], rtol=1e-12)
    v = np.zeros((1, 8)); v[0, 6:8] = [10.0, -3.0]
    o.set_state(q, v)
    f = o.forward(ctrl=(10.0, -3.0))
    np.testing.assert_allclose(f["passive"][6:8], [-0.1, 0.03], rtol=1e-12)
    assert abs(f["actuator"][6]) < 1e-12
    # wheel spin-up: both wheels at ctrl 20 (force-limited 0.65 N m): reaction spins the torso, wheel relative accel
    o.set_state(q, np.zeros((1, 8)))
    o.physics(np.array([[20.0, 20.0]]), 500)
    _, v1, _, t = o.get_state()
    assert v1[0, 6] > 0 and v1[0, 7] > 0 and abs(v1[0, 6] - v1[0, 7]) < 1e-9
    # opposite-sign hinge axes: equal wheel rates about (-x, +x) cancel their torques on the torso about x
    assert abs(v1[0, 3]) < 1e-9


def test_block_rests_on_floor_and_is_detected():
    o = O.Oracle("Env03-v2", 1)
    q = np.zeros((1, 16)); q[0, 3] = 1; q[0, 12] = 1; q[0, 2] = 5.0
    q[0, 9:12] = [3.0, 3.0, 0.005]
    o.set_state(q, np.zeros((1, 14)))
    o.physics(np.zeros((1, 2)), 250 * 40)
    qq, vv, _, _ = o.get_state()
    f = o.forward()
    assert sum(1 for c in f["contacts"] if c["body2"] == 4 and c["body1"] == 0) == 4
    assert abs(vv[0, 10]) < 1e-3 and -0.002 < qq[0, 11] - 0.0 < 0.002  # centre ~ floor_z + half size = 0, inside the margin band


def test_solver_stationarity_on_random_contact_states():
    """the Newton solution is a stationary point of the convex acceleration problem: M (qacc - qacc_smooth) equals the
    constraint force J^T f it reports, and the pyramidal row forces are non-negative (states from a random rollout with
    wheel, torso, block<->floor and block<->robot contacts)"""
    o = O.Oracle("Env03-v2", 24, seed=5, noise=False, auto_reset=False, threads=8)
    o.reset()
    rng = np.random.default_rng(0)
    checked = with_coupled = 0
    for t in range(40):
        o.step(rng.uniform(-1, 1, size=(24, 2)).astype(np.float32))
        qv = o.get_state()[1]
        for e in range(0, 24, 3):
            f = o.forward(env=e, ctrl=(qv[e, 6], qv[e, 7]))
            if f["ncon"] == 0:
                continue
            lhs = f["M"] @ (f["qacc"] - f["qacc_smooth"])
            scale = max(1e-9, np.abs(f["qfrc_constraint"]).max())
            assert np.abs(lhs - f["qfrc_constraint"]).max() < 1e-6 * scale + 1e-10, (t, e)
            assert (f["efc_force"] >= -1e-12).all()
            checked += 1
            with_coupled += any(c["body2"] == 4 and c["body1"] != 0 for c in f["contacts"])
    assert checked > 100 and with_coupled > 0


def test_block_cannot_touch_both_wheels_at_once():
    """the kernel tests the block against ONE wheel, the one on its side of the robot (brs_core.hpp collide_coupled (ii)):
    the gap between the wheels' inner faces exceeds the block's diameter plus twice the contact margin
    (ref:envs/robot-02.xml:9-18 wheel positions / sizes, ref:envs/env03_v1.xml:31-37 block size / margin)"""
    wheel_px, wheel_hl, block_s, margin = 0.074, 0.013, 0.02, 0.002
    gap = 2 * (wheel_px - wheel_hl)
    assert gap > 2 * (block_s * np.sqrt(3.0) + margin) + 0.04
    # and the oracle agrees on states with the block between the wheels: never two wheel contacts
    rng = np.random.default_rng(4)
    o = O.Oracle("Env03-v2", 1)
    for _ in range(300):
        q = np.zeros((1, 16)); q[0, 3] = 1; q[0, 2] = 1.0
        bq = rng.normal(size=4); q[0, 12:16] = bq / np.linalg.norm(bq)
        q[0, 9:12] = [rng.uniform(-0.09, 0.09), rng.uniform(-0.06, 0.06), 1.0 + rng.uniform(-0.02, 0.09)]
        o.set_state(q, np.zeros((1, 14)))
        wheels = {c["body1"] for c in o.forward()["contacts"] if c["body2"] == 4 and c["body1"] in (2, 3)}
        assert len(wheels) <= 1

"""Geometric self-consistency of the project's OWN block<->robot contact generator (oracle/brs_oracle.c box_box_own /
box_cyl_own; MuJoCo's mjc_BoxBox / libccd point sets are not reproducible here, so parity with MuJoCo is UNPINNED for
these contacts -- DESIGN.md §2).  What can be pinned without MuJoCo is that every contact it emits is a geometrically
valid one: orthonormal frame, a point between the two surfaces, a distance that moves one-to-one with a shift of the
block along the normal (which also pins the normal's sign), and nothing at all when the bodies are apart."""
import numpy as np
import pytest

from oracle import oracle as O

TORSO_S = np.array([0.05, 0.0185, 0.0855]); TORSO_C = np.array([0.0, 0.0, 0.0995])   # ref:envs/robot-02.xml:4-7
BLOCK_S = 0.02; MARGIN = 0.002                                                      # ref:envs/env03_v1.xml:31-37
WHEEL_R, WHEEL_HL = 0.034, 0.013
WHEEL_P = {2: np.array([-0.074, 0.0, 0.034]), 3: np.array([0.074, 0.0, 0.034])}     # ref:envs/robot-02.xml:9-18


def _quat(rng):
    q = rng.normal(size=4); return q / np.linalg.norm(q)


def _rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _sd_box(p, half):  # signed distance of a point to an axis-aligned box at the origin
    d = np.abs(p) - half
    return np.linalg.norm(np.maximum(d, 0)) + min(d.max(), 0.0)


def _sd_cyl_x(p, r, hl):  # cylinder with axis x
    d = np.array([np.hypot(p[1], p[2]) - r, abs(p[0]) - hl])
    return np.linalg.norm(np.maximum(d, 0)) + min(d.max(), 0.0)


def _contacts(o, block_pos, block_quat):
    q = np.zeros((1, 16)); q[0, 3] = 1.0; q[0, 2] = 1.0            # robot 1 m above the floor: only block<->robot contacts
    q[0, 9:12] = block_pos + np.array([0, 0, 1.0]); q[0, 12:16] = block_quat
    o.set_state(q, np.zeros((1, 14)))
    f = o.forward()
    return [c for c in f["contacts"] if c["body2"] == 4 and c["body1"] in (1, 2, 3)]


def _place_near(rng, target):
    """a block pose whose centre is 0..1.5 block sizes outside the surface of `target` (1 torso, 2/3 wheels)"""
    if target == 1:
        face = rng.integers(3); sign = rng.choice([-1.0, 1.0])
        c = TORSO_C + (rng.uniform(-1, 1, 3) * TORSO_S)
        c[face] = TORSO_C[face] + sign * (TORSO_S[face] + BLOCK_S * rng.uniform(0.6, 1.6))
    else:
        th = rng.uniform(0, 2 * np.pi); rad = WHEEL_R + BLOCK_S * rng.uniform(0.6, 1.6)
        c = WHEEL_P[target] + np.array([rng.uniform(-1, 1) * WHEEL_HL, rad * np.cos(th), rad * np.sin(th)])
    return c, _quat(rng)


@pytest.mark.parametrize("target", [1, 2, 3])
def test_emitted_contacts_are_geometrically_valid(target):
    rng = np.random.default_rng(100 + target)
    o = O.Oracle("Env03-v2", 1)
    n_with = n_early = 0
    for _ in range(400):
        c, bq = _place_near(rng, target)
        RB = _rot(bq)
        cons = _contacts(o, c, bq)
        for con in cons:
            if con["body1"] != target:
                continue
            n_with += 1
            fr, pos, dist = con["frame"], con["pos"] - np.array([0, 0, 1.0]), con["dist"]
            np.testing.assert_allclose(fr @ fr.T, np.eye(3), atol=1e-9)          # orthonormal contact frame
            assert dist < MARGIN + 1e-12
            n = fr[0]
            # (the sign of the normal is pinned by test_distance_follows_a_shift_along_the_normal)
            # the point sits between the two surfaces: within |dist|/2 (+ clamping slack) of each of them
            sd_b = _sd_box(RB.T @ (pos - c), np.full(3, BLOCK_S))
            sd_r = _sd_box(pos - TORSO_C, TORSO_S) if target == 1 else _sd_cyl_x(pos - WHEEL_P[target], WHEEL_R, WHEEL_HL)
            slack = abs(dist) / 2 + 2.5e-3
            ok = abs(sd_b) <= slack and abs(sd_r) <= slack
            if not ok:
                # known limit of the face-axis SAT (no edge-edge axes): in an edge-edge configuration the single
                # "deepest vertex clamped into the reference rectangle" contact comes early, by up to ~1.5 cm
                assert len([k for k in cons if k["body1"] == target]) == 1 and max(abs(sd_b), abs(sd_r)) < 0.015 + abs(dist), (sd_b, sd_r, dist)
                n_early += 1
    assert n_with > 100, "the placement must actually produce contacts"
    assert n_early <= 0.03 * n_with, (n_early, n_with)


@pytest.mark.parametrize("target", [1, 2, 3])
def test_distance_follows_a_shift_along_the_normal(target):
    rng = np.random.default_rng(200 + target)
    o = O.Oracle("Env03-v2", 1)
    checked = jumps = 0
    for _ in range(300):
        c, bq = _place_near(rng, target)
        cons = [k for k in _contacts(o, c, bq) if k["body1"] == target]
        if not cons:
            continue
        k0 = min(cons, key=lambda k: k["dist"])
        eps = 2e-5
        cons2 = [k for k in _contacts(o, c - eps * k0["frame"][0], bq) if k["body1"] == target]   # push the block in by eps
        if not cons2:
            continue
        k1 = min(cons2, key=lambda k: k["dist"])
        if np.abs(k1["frame"][0] - k0["frame"][0]).max() > 1e-6 or len(cons2) != len(cons):
            continue   # the reference face / the contact set changed: not the same contact
        if abs(k0["dist"] - k1["dist"]) > 1e-3:
            # known discontinuity of the generator: a deep vertex laterally outside the reference rectangle is
            # reported (clamped) only while no other vertex is inside the rectangle; when one enters, the deep one drops out
            jumps += 1
            continue
        dd = k0["dist"] - k1["dist"]
        if target == 1:
            assert abs(dd - eps) < 0.2 * eps, (k0["dist"], k1["dist"])      # face contacts: exactly one-to-one
        else:
            assert 0.3 * eps < dd < 1.2 * eps, (k0["dist"], k1["dist"])      # rim points slide along the rim: 0 < d' <= 1
        checked += 1
    assert checked > 50 and jumps <= 0.05 * checked, (checked, jumps)


def test_no_contacts_when_apart_and_some_when_overlapping():
    rng = np.random.default_rng(5)
    o = O.Oracle("Env03-v2", 1)
    for _ in range(100):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        far = TORSO_C + d * 0.30                       # beyond every bounding sphere + margin
        assert _contacts(o, far, _quat(rng)) == []
    hits = 0
    for _ in range(100):
        inside = TORSO_C + rng.uniform(-0.6, 0.6, 3) * TORSO_S   # block centre well inside the torso box
        hits += bool(_contacts(o, inside, _quat(rng)))
    assert hits == 100


def test_floor_contacts_of_wheels_and_torso_are_surface_points():
    """plane<->cylinder / plane<->box restated from MuJoCo's primitives (oracle plane_cylinder / plane_box): every contact
    has the plane normal, its point is half a distance below a point ON the geom surface, and that surface point is
    `dist` above the floor (z = -0.02, ref:envs/env01_v1.xml:27)"""
    rng = np.random.default_rng(9)
    o = O.Oracle("Env01-v2", 1)
    FLOOR = -0.02
    seen = {1: 0, 2: 0, 3: 0}
    for _ in range(300):
        q = np.zeros((1, 9)); bq = _quat(rng); R = _rot(bq)
        q[0, 3:7] = bq
        # lowest point of the robot's geoms just touching the floor (+- 3 mm)
        pts = [TORSO_C + np.array([sx, sy, sz]) * TORSO_S for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
        low = min((R @ p)[2] for p in pts)
        for w in (2, 3):
            for th in np.linspace(0, 2 * np.pi, 64, endpoint=False):
                for ax in (-WHEEL_HL, WHEEL_HL):
                    low = min(low, (R @ (WHEEL_P[w] + np.array([ax, WHEEL_R * np.cos(th), WHEEL_R * np.sin(th)])))[2])
        q[0, 2] = FLOOR - low + rng.uniform(-0.003, 0.001)
        o.set_state(q, np.zeros((1, 8)))
        for con in o.forward()["contacts"]:
            assert con["body1"] == 0 and con["body2"] in (1, 2, 3)
            np.testing.assert_allclose(con["frame"][0], [0, 0, 1], atol=1e-12)
            surf_w = con["pos"] + con["frame"][0] * con["dist"] * 0.5          # world point on the geom surface
            assert abs((surf_w[2] - FLOOR) - con["dist"]) < 1e-9
            p_body = R.T @ (surf_w - q[0, :3])
            sd = _sd_box(p_body - TORSO_C, TORSO_S) if con["body2"] == 1 else _sd_cyl_x(p_body - WHEEL_P[con["body2"]], WHEEL_R, WHEEL_HL)
            assert abs(sd) < 1e-9, (con["body2"], sd)
            seen[con["body2"]] += 1
    assert min(seen.values()) > 20, seen

"""Block<->robot contact generation: oracle/brs_oracle.c box_box / box_cyl.

MuJoCo's own mjc_BoxBox / convex-collider point sets cannot be reproduced verbatim here (no source, not importable), so
parity with MuJoCo is UNPINNED for these contacts.  What IS pinned:
  * the generator against an independent brute-force numpy reference (tests/ref_boxbox.py: vertex-projection SAT over
    the 15 axes, 3-D Sutherland-Hodgman clipping, alternating-projection box distance): same axis, same point set;
  * geometry: orthonormal frame, every point between the two surfaces, distance one-to-one with a shift along the normal;
  * no early contact (nothing beyond the margin, edge-edge poses included) and no missed contact;
  * continuity of the contact patch while a vertex slides across the edge of the reference rectangle."""
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
    n_with = 0
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
            # between the two surfaces: each surface within |dist|/2 of the point along the normal; the slack covers
            # points whose nearest surface point is not along the contact normal (rim / oblique incident face)
            slack = abs(dist) / 2 + (1e-9 if target == 1 else 2.5e-3)
            if target == 1:
                # face / edge contacts: the point is never further than |dist|/2 from either surface
                assert abs(sd_b) <= abs(dist) / 2 + 1e-9 and abs(sd_r) <= abs(dist) / 2 + 1e-9, (sd_b, sd_r, dist)
            else:
                assert abs(sd_b) <= slack and abs(sd_r) <= slack, (sd_b, sd_r, dist)
    assert n_with > 100, "the placement must actually produce contacts"


@pytest.mark.parametrize("target", [1, 2, 3])
def test_distance_follows_a_shift_along_the_normal(target):
    rng = np.random.default_rng(200 + target)
    o = O.Oracle("Env03-v2", 1)
    checked = 0
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
        assert abs(k0["dist"] - k1["dist"]) < 1e-3, "the deepest point must not jump under a 20 um shift"
        dd = k0["dist"] - k1["dist"]
        if target == 1:
            assert abs(dd - eps) < 0.2 * eps, (k0["dist"], k1["dist"])      # face contacts: exactly one-to-one
        else:
            assert 0.3 * eps < dd < 1.2 * eps, (k0["dist"], k1["dist"])      # rim points slide along the rim: 0 < d' <= 1
        checked += 1
    assert checked > 50


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


# ------------------------------------------------------------------------------------------------ box-box vs reference
from tests import ref_boxbox as RB  # noqa: E402


def _random_pose(rng, spread=1.5):
    """block centre 0.4 .. spread block sizes outside a random face / edge / corner region of the torso box"""
    q = _quat(rng); R = _rot(q)
    c = rng.uniform(-1, 1, 3) * TORSO_S
    k = rng.integers(1, 4)  # how many coordinates are pushed outside: 1 face, 2 edge, 3 corner region
    for ax in rng.permutation(3)[:k]:
        c[ax] = rng.choice([-1.0, 1.0]) * (TORSO_S[ax] + BLOCK_S * rng.uniform(0.4, spread))
    return c, R


def _same_points(a, b, tol=1e-9):
    if len(a) != len(b):
        return False
    used = set()
    for p in a:
        hit = [i for i, q in enumerate(b) if i not in used and np.abs(p - q).max() < tol]
        if not hit:
            return False
        used.add(hit[0])
    return True


def test_box_box_matches_the_bruteforce_reference():
    rng = np.random.default_rng(42)
    kinds = {"T": 0, "B": 0, "E": 0}
    many = none = 0
    for _ in range(3000):
        c, R = _random_pose(rng)
        ref = RB.contacts(TORSO_S, BLOCK_S, c, R, MARGIN)
        pos, dist, nrm, code, raw = O.box_box_points(TORSO_S, BLOCK_S, c, R, MARGIN)
        if ref is None or len(ref["points"]) == 0:
            assert len(pos) == 0, (c, code)
            none += 1
            continue
        if ref["near_tie"]:
            continue
        exp_code = {"T": ref["index"], "B": 3 + ref["index"], "E": 6 + ref["index"]}[ref["kind"]]
        assert code == exp_code, (code, exp_code, c)
        np.testing.assert_allclose(nrm, ref["normal"], atol=1e-12)
        assert _same_points(raw[:, :3], ref["points"]), (raw, ref["points"])
        np.testing.assert_allclose(np.sort(raw[:, 3]), np.sort(ref["dists"]), atol=1e-12)
        # reduction: at most 6 (the kernel's patch budget), and they are the deepest of the full set
        assert len(pos) == min(6, len(raw))
        if len(raw) > 4:
            many += 1
        if len(raw) > 6:
            assert dist.max() <= np.sort(raw[:, 3])[5] + 1e-15
        for p in pos:
            assert any(np.abs(p - r[:3]).max() < 1e-12 for r in raw)
        kinds[ref["kind"]] += 1
    assert min(kinds.values()) > 30 and many > 10 and none > 100, (kinds, many, none)


def test_box_box_no_early_and_no_missed_contact():
    """a contact only if the boxes are closer than the margin (edge-edge and corner poses included: the face-axis-only
    SAT of round 1 reported those up to 1.5 cm early), and always one when they overlap"""
    rng = np.random.default_rng(7)
    apart = touching = 0
    for _ in range(600):
        c, R = _random_pose(rng, spread=1.9)
        d = RB.box_distance(TORSO_S, BLOCK_S, c, R)
        pos, dist, nrm, code, raw = O.box_box_points(TORSO_S, BLOCK_S, c, R, MARGIN)
        if d > MARGIN + 1e-6:
            assert len(pos) == 0, (d, code, dist)
            apart += 1
        elif d < MARGIN - 1e-4:
            assert len(pos) >= 1, (d, c)
            # the deepest reported distance is the true separation when apart (edge-edge: exactly; face: <=)
            if d > 1e-5:
                assert dist.min() <= d + 1e-6 and dist.min() > d - 5e-4, (dist, d)
            touching += 1
    assert apart > 100 and touching > 100, (apart, touching)


def test_box_box_edge_edge_pose_gives_one_point_at_the_true_distance():
    """block edge (cube turned 45 degrees about z... and tipped) across the torso's vertical edge"""
    c45 = np.sqrt(0.5)
    Rz = np.array([[c45, -c45, 0], [c45, c45, 0], [0, 0, 1.0]])
    th = np.radians(35)
    Ry = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    R = Ry @ Rz
    dirn = np.array([1.0, 1.0, 0.0]) / np.sqrt(2)
    corner = np.array([TORSO_S[0], TORSO_S[1], 0.0])
    for gap in (-0.001, 0.0005, 0.0015, 0.003, 0.01):
        # support of the cube along -dirn gives where its nearest feature sits; place it `gap` off the torso edge
        ext = BLOCK_S * np.abs(R.T @ dirn).sum()
        c = corner + dirn * (ext + gap)
        d = RB.box_distance(TORSO_S, BLOCK_S, c, R)
        pos, dist, nrm, code, raw = O.box_box_points(TORSO_S, BLOCK_S, c, R, MARGIN)
        if d >= MARGIN:
            assert len(pos) == 0, (gap, d, dist)
        else:
            assert len(pos) == 1 and code >= 6, (gap, code, dist)
            if d > 0:
                assert abs(dist[0] - d) < 1e-6, (dist, d)
            assert abs(np.linalg.norm(nrm) - 1) < 1e-12 and nrm @ c > 0


def test_box_box_patch_is_continuous_while_a_vertex_crosses_the_rectangle_edge():
    """slide a tilted block along the torso's front face past its side edge: the clipped contact patch (its deepest
    distance and its centroid) moves continuously -- the round-1 generator dropped / clamped the outside vertex"""
    th = np.radians(4.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    Rzz = _rot(np.array([np.cos(0.01), 0, 0, np.sin(0.01)]))
    a = np.radians(30.0)  # in-plane turn about the face normal: the vertices cross the rectangle edge one at a time
    Ryy = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    R = Rx @ Rzz @ Ryy
    prev = None
    counts = set()
    for x in np.linspace(TORSO_S[0] - 0.035, TORSO_S[0] + 0.005, 400):
        c = np.array([x, TORSO_S[1] + BLOCK_S * 0.85, 0.01])
        pos, dist, nrm, code, raw = O.box_box_points(TORSO_S, BLOCK_S, c, R, MARGIN)
        assert code == 1 and len(raw) >= 3, (x, code, len(raw))
        counts.add(len(raw))
        # centroid of the clipped polygon (area-weighted, from the raw points ordered by angle) and deepest distance
        P = raw[:, [0, 2]]; ctr = P.mean(axis=0)
        order = np.argsort(np.arctan2(P[:, 1] - ctr[1], P[:, 0] - ctr[0])); P = P[order]
        x0, y0 = P[:, 0], P[:, 1]; x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
        cr = x0 * y1 - x1 * y0; area = cr.sum() / 2
        cen = np.array([((x0 + x1) * cr).sum(), ((y0 + y1) * cr).sum()]) / (6 * area)
        cur = (raw[:, 3].min(), cen, abs(area))
        if prev is not None:
            assert abs(cur[0] - prev[0]) < 2e-5, (x, cur[0], prev[0])
            assert np.abs(cur[1] - prev[1]).max() < 2e-4, (x, cur[1], prev[1])
            assert abs(cur[2] - prev[2]) < 2e-5, (x, cur[2], prev[2])
        prev = cur
    assert len(counts) >= 2, "the sweep must change the polygon's vertex count"


def test_box_cyl_edge_on_barrel_and_vertex_cases():
    """block edge resting across the wheel's barrel: ONE contact at the edge's closest point to the axis, at the true
    radial distance (round 1 had only vertex and rim candidates there)"""
    c45 = np.sqrt(0.5)
    R = np.array([[c45, -c45, 0], [c45, c45, 0], [0, 0, 1.0]]) @ np.array([[1, 0, 0], [0, c45, -c45], [0, c45, c45]])
    # which block edge is lowest along -z?  brute force the true distance by sampling all 12 edges
    def true_dist(d):
        best = 1e9
        for j in range(3):
            for su in (-1, 1):
                for sv in (-1, 1):
                    o = np.zeros(3); o[(j + 1) % 3] = su * BLOCK_S; o[(j + 2) % 3] = sv * BLOCK_S
                    for t in np.linspace(-BLOCK_S, BLOCK_S, 2001):
                        loc = o.copy(); loc[j] = t
                        best = min(best, _sd_cyl_x(R @ loc + d, WHEEL_R, WHEEL_HL))
        return best
    for gap in (-0.002, 0.0005, 0.0015):
        ext = BLOCK_S * np.abs(R.T @ np.array([0, 0, -1.0])).sum()
        d = np.array([0.002, 0.0, WHEEL_R + ext + gap])
        got = O.box_cyl_point(d, R, BLOCK_S, WHEEL_R, WHEEL_HL, MARGIN)
        td = true_dist(d)
        assert got is not None and abs(got[2] - td) < 2e-6, (gap, got, td)
        pos, nrm, dist = got
        assert abs(np.linalg.norm(nrm) - 1) < 1e-12 and nrm[2] > 0.9
    assert O.box_cyl_point(np.array([0.0, 0.0, WHEEL_R + 0.05]), R, BLOCK_S, WHEEL_R, WHEEL_HL, MARGIN) is None


def test_box_cyl_candidates_against_the_true_distance():
    """independent anchor for bo_box_cyl_point: the TRUE cube<->cylinder distance by alternating projections
    (tests/ref_boxcyl.py) over random separated poses.  What holds without exception: a reported distance is never BELOW the
    truth (every candidate is a genuine point-to-solid distance), nothing is reported beyond the margin, and whenever the
    closest feature of the cube is a VERTEX the candidates find the true distance.  What does not, and is the measured size
    of the documented deviation from MuJoCo's convex collider: with the cube's closest feature an edge
    or a face -- against the rim or the cap of this disc-shaped wheel (r = 34 mm, half length 13 mm) -- the closest-feature
    candidates read too far, by up to the whole 2 mm margin (= the contact is only found once the solids interpenetrate)."""
    from tests import ref_boxcyl as rc
    rng = np.random.default_rng(41)
    stat = {1: [0, 0, 0], 2: [0, 0, 0], 3: [0, 0, 0]}   # closest cube feature (1 face, 2 edge, 3 vertex): poses, exact, missed
    for _ in range(3000):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        d = u * rng.uniform(0.03, 0.075)                       # from touching the barrel to well outside the margin
        true, a, b = rc.distance(d, R, BLOCK_S, WHEEL_R, WHEEL_HL)
        if true < 1e-5:
            continue                                            # intersecting (or numerically touching) solids: penetration depth is not a Euclidean distance
        got = O.box_cyl_point(d, R, BLOCK_S, WHEEL_R, WHEEL_HL, MARGIN)
        if got is not None:
            assert got[2] < MARGIN and got[2] > true - 2e-6, (got[2], true)   # inside the margin, never closer than the truth (2e-6: the reference iteration converges linearly)
        if true > MARGIN + 2e-6:
            assert got is None, ("contact beyond the margin", true, got)
            continue
        feature = int((np.abs(np.abs(R.T @ (b - d)) - BLOCK_S) < 1e-9).sum())
        if feature not in stat:
            continue
        st = stat[feature]
        st[0] += 1; st[1] += got is not None and got[2] - true < 3e-6; st[2] += got is None
    print("box<->cylinder, separated poses inside the margin, by the cube's closest feature (poses, at the true distance, not found): "
          f"face {stat[1]}, edge {stat[2]}, vertex {stat[3]}")
    assert stat[3][0] > 20 and stat[3][1] == stat[3][0] and stat[3][2] == 0, "vertex cases are exact"
    assert stat[2][0] > 30 and stat[1][0] > 15
    # measured: face 3 of 37 exact (5 not found), edge 14 of 76 (39 not found) -- a regression would move these, an improvement too
    assert stat[2][1] >= 10 and stat[2][2] <= 0.6 * stat[2][0] and stat[1][2] <= 0.25 * stat[1][0]

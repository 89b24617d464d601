"""Constructed contact states shared by tests/test_hostsim_parity.py (kernel source on the host) and tests/test_gpu_parity.py
(the HIP path): states a random rollout reaches only by chance -- the block against every torso face and both wheels
(5- and 6-point patches, edge-edge poses, the wheel barrel) and the robot pressed into the floor in every orientation.
Geometry re-typed from the reference's XML (envs/robot-02.xml:4-20, envs/env03_v1.xml:31-37).  Test infrastructure."""
import numpy as np

TC, TS, BS = np.array([0.0, 0.0, 0.0995]), np.array([0.05, 0.0185, 0.0855]), 0.02
WP = {1: np.array([-0.074, 0.0, 0.034]), 2: np.array([0.074, 0.0, 0.034])}


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def block_robot_states(n=96, seed=17):
    """Env03: airborne robot (no floor contacts), block placed with a random pose and approach velocity against the torso
    faces (every third env) or around a wheel's barrel -> (qpos [n,16], qvel [n,14])"""
    rng = np.random.default_rng(seed)
    qpos = np.zeros((n, 16)); qvel = np.zeros((n, 14))
    qpos[:, 3] = 1.0; qpos[:, 2] = 1.0
    for i in range(n):
        if i % 3 == 0:
            face = rng.integers(3); sign = rng.choice([-1.0, 1.0])
            c = TC + rng.uniform(-1, 1, 3) * TS
            c[face] = TC[face] + sign * (TS[face] + BS * rng.uniform(0.7, 1.3))
        else:
            th = rng.uniform(0, 2 * np.pi); rad = 0.034 + BS * rng.uniform(0.7, 1.3)
            c = WP[1 + i % 2] + np.array([rng.uniform(-1, 1) * 0.013, rad * np.cos(th), rad * np.sin(th)])
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        qpos[i, 9:12] = c + np.array([0, 0, 1.0]); qpos[i, 12:16] = q
        qvel[i, 8:11] = rng.normal(size=3) * 2.0
        qvel[i, 11:14] = rng.normal(size=3) * 5.0
        qvel[i, 6:8] = rng.normal(size=2) * 10.0
    return qpos, qvel


def edge_edge_states(n=64, seed=29):
    """Env03: block rotated 45 degrees about two of its axes and pushed edge-first against a vertical edge of the torso box,
    0.5 mm outside ... 1.5 mm inside the 2 mm margin: the edge-pair axis of the SAT wins and the patch is ONE point whose
    existence is the `separation < margin` decision -> (qpos, qvel)"""
    rng = np.random.default_rng(seed)
    qpos = np.zeros((n, 16)); qvel = np.zeros((n, 14))
    qpos[:, 3] = 1.0; qpos[:, 2] = 1.0
    for i in range(n):
        sx, sy = rng.choice([-1.0, 1.0]), rng.choice([-1.0, 1.0])
        # torso edge along z at (sx TS[0], sy TS[1]); approach direction = outward diagonal in the x-y plane
        out = np.array([sx, sy, 0.0]) / np.sqrt(2.0)
        # block: one edge horizontal and perpendicular to the approach (rotate 45 deg about the horizontal axis normal to `out`)
        t = np.array([-out[1], out[0], 0.0])                       # horizontal, perpendicular to out
        a = np.pi / 4 + rng.normal() * 0.05
        qa = np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * t])  # rotation about t: a block edge parallel to t leads
        yaw = np.arctan2(t[1], t[0]) + rng.normal() * 0.05          # align a block axis with t first
        qy = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
        w1, x1, y1, z1 = qa; w2, x2, y2, z2 = qy
        q = np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                      w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])
        gap = 0.002 + rng.uniform(-0.0015, 0.0005)                  # edge-to-edge distance along `out`
        edge = TC + np.array([sx * TS[0], sy * TS[1], rng.uniform(0.1, 0.7) * TS[2]])   # upper half: clear of the wheels
        c = edge + out * (gap + BS * np.sqrt(2.0))
        qpos[i, 9:12] = c + np.array([0, 0, 1.0]); qpos[i, 12:16] = q / np.linalg.norm(q)
        qvel[i, 8:11] = -out * rng.uniform(0.0, 1.0) + rng.normal(size=3) * 0.05
        qvel[i, 11:14] = rng.normal(size=3) * 0.5
    return qpos, qvel


def floor_states(n=128, seed=23):
    """Env01: robot in random orientations (upright, lying on the torso's broad face, on a wheel's flat side, anything) pressed
    0..3 mm (flat poses up to 15 mm) into the floor with random velocities -> (qpos [n,9], qvel [n,8])"""
    rng = np.random.default_rng(seed)
    wp = [WP[1], WP[2]]
    pts = [TC + np.array([sx, sy, sz]) * TS for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    for w in wp:
        for th in np.linspace(0, 2 * np.pi, 48, endpoint=False):
            for ax in (-0.013, 0.013):
                pts.append(w + np.array([ax, 0.034 * np.cos(th), 0.034 * np.sin(th)]))
    pts = np.array(pts)
    qpos = np.zeros((n, 9)); qvel = np.zeros((n, 8))
    for i in range(n):
        q = rng.normal(size=4)
        if i % 4 == 0:
            q = np.array([1.0, 0, 0, 0]) + 0.05 * rng.normal(size=4)
        elif i % 4 == 1:
            a = rng.choice([-1.0, 1.0]) * (np.pi / 2 + 0.02 * rng.normal())
            q = np.array([np.cos(a / 2), np.sin(a / 2), 0, 0]) + 0.004 * rng.normal(size=4)
        elif i % 4 == 2:
            a = rng.choice([-1.0, 1.0]) * (np.pi / 2 + 0.02 * rng.normal())
            q = np.array([np.cos(a / 2), 0, np.sin(a / 2), 0]) + 0.004 * rng.normal(size=4)
        q /= np.linalg.norm(q)
        low = (pts @ quat_to_mat(q).T)[:, 2].min()
        qpos[i, 3:7] = q; qpos[i, 2] = -0.02 - low - rng.uniform(0.0, 0.003 if i % 4 in (0, 3) else 0.015)
        qvel[i, :3] = rng.normal(size=3) * 0.3; qvel[i, 3:6] = rng.normal(size=3) * 2.0; qvel[i, 6:8] = rng.normal(size=2) * 15.0
    return qpos, qvel


def coupled_contact_count(orc, n):
    """block<->robot contacts per env as the oracle generates them"""
    return np.array([sum(1 for c in orc.forward(env=i)["contacts"] if c["body2"] == 4 and c["body1"] != 0) for i in range(n)])


def rel_vel_error(v_ref, v):
    return np.abs(v_ref - v).max(axis=1) / (1.0 + np.abs(v_ref).max(axis=1))

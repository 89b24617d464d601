"""Independent numpy reference for the block<->wheel contact generator (tests only): the TRUE Euclidean distance between
a cube and a capped cylinder, by alternating projections onto the two convex sets (von Neumann / POCS: converges to a
pair of closest points when the sets are disjoint, to a common point when they intersect).  Not a transcription of
oracle/brs_oracle.c bo_box_cyl_point, which only evaluates closest-FEATURE candidates (8 vertices, 12 edge points nearest
the axis, 2 cylinder surface points): this reference says how far that candidate set is from the truth.
Frame: cylinder centre at the origin, axis = x, radius r, half length hl; cube of half size s at d, axes = columns of R."""
import numpy as np


def proj_cyl(p, r, hl):
    q = p.copy()
    q[0] = min(max(q[0], -hl), hl)
    rho = np.hypot(q[1], q[2])
    if rho > r:
        q[1] *= r / rho; q[2] *= r / rho
    return q


def proj_box(p, d, R, s):
    loc = R.T @ (p - d)
    return R @ np.clip(loc, -s, s) + d


def distance(d, R, s, r, hl, iters=4000, tol=1e-13):
    """(distance, point on the cylinder, point on the box); 0 when the solids intersect"""
    a = proj_cyl(np.asarray(d, float), r, hl)
    prev = np.inf
    for _ in range(iters):
        b = proj_box(a, d, R, s)
        a = proj_cyl(b, r, hl)
        dist = np.linalg.norm(a - b)
        if abs(prev - dist) < tol:
            break
        prev = dist
    return dist, a, b

"""Independent numpy reference for the block<->torso contact generator (tests only).

Brute force on purpose, and NOT a transcription of oracle/brs_oracle.c: separations come from projecting all 16 box
vertices on each of the 15 candidate axes, the face case is a textbook Sutherland-Hodgman clip of the incident face
against the four side planes of the reference face (3-D polygon, not the 2-D Liang-Barsky enumeration the oracle and
the kernel use), the box-to-box distance is found by alternating projections.  Frame: box T (half sizes sT) at the
origin, axis aligned; box B = cube of half size s at cg with axes = columns of RTB."""
import itertools

import numpy as np

EDGE_REL, EDGE_ABS, PAR_EPS = 0.05, 1e-5, 1e-6  # the specification's constants (DESIGN.md, section on f2)


def verts(half):
    return np.array([[sx * half[0], sy * half[1], sz * half[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def box_vertices(sT, s, cg, RTB):
    return verts(sT), verts(np.full(3, s)) @ RTB.T + cg


def separation(axis, VT, VB):
    """signed separation of the two vertex clouds along a unit axis (> 0: apart, < 0: overlap depth)"""
    pT, pB = VT @ axis, VB @ axis
    return max(pB.min() - pT.max(), pT.min() - pB.max())


def axes15(RTB):
    out = [("T", k, np.eye(3)[k]) for k in range(3)] + [("B", j, RTB[:, j]) for j in range(3)]
    for i in range(3):
        for j in range(3):
            L = np.cross(np.eye(3)[i], RTB[:, j])
            n2 = L @ L
            if n2 < PAR_EPS:
                continue
            out.append(("E", 3 * i + j, L / np.sqrt(n2)))
    return out


def choose_axis(sT, s, cg, RTB, margin):
    """(kind, index, unit axis pointing T -> B, separation) or None if separated by more than margin; near_tie flags
    a choice that another axis misses by less than 1e-7 (tests skip those)"""
    VT, VB = box_vertices(sT, s, cg, RTB)
    best_f, best_e = None, None
    seps = []
    for kind, idx, L in axes15(RTB):
        sep = separation(L, VT, VB)
        seps.append((kind, sep))
        if sep > margin:
            return None
        if kind != "E":
            if best_f is None or sep > best_f[3]:
                best_f = (kind, idx, L, sep)
        elif best_e is None or sep > best_e[3]:
            best_e = (kind, idx, L, sep)
    use_edge = best_e is not None and best_e[3] > best_f[3] + EDGE_REL * abs(best_f[3]) + EDGE_ABS
    kind, idx, L, sep = best_e if use_edge else best_f
    if L @ cg < 0:
        L = -L
    fs = sorted(x[1] for x in seps if x[0] != "E")
    es = sorted(x[1] for x in seps if x[0] == "E")
    thr = best_f[3] + EDGE_REL * abs(best_f[3]) + EDGE_ABS
    near = (len(fs) > 1 and fs[-1] - fs[-2] < 1e-7) or (use_edge and len(es) > 1 and es[-1] - es[-2] < 1e-7) or \
           (best_e is not None and abs(best_e[3] - thr) < 1e-7)
    return kind, idx, L, sep, near


def clip_polygon(poly, n, d):
    """Sutherland-Hodgman: keep the part of the 3-D polygon with n.x <= d"""
    out = []
    for a, b in zip(poly, poly[1:] + poly[:1]):
        da, db = n @ a - d, n @ b - d
        if da <= 0:
            out.append(a)
        if (da < 0 < db) or (db < 0 < da):
            t = da / (da - db)
            out.append(a + t * (b - a))
    return out


def face_of(center, axes, half, normal):
    """the face of a box (4 vertices, in order) whose outward normal is most parallel to `normal`"""
    k = int(np.argmax(np.abs(axes.T @ normal)))
    sg = 1.0 if axes[:, k] @ normal >= 0 else -1.0
    a1, a2 = (k + 1) % 3, (k + 2) % 3
    c = center + sg * half[k] * axes[:, k]
    return [c + su * half[a1] * axes[:, a1] + sv * half[a2] * axes[:, a2] for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1))], k, sg


def contacts(sT, s, cg, RTB, margin):
    """-> None | dict(kind, index, normal, points (k,3) contact positions = midpoints, dists (k,)) -- ALL points, no
    reduction to 4"""
    ch = choose_axis(sT, s, cg, RTB, margin)
    if ch is None:
        return None
    kind, idx, L, sep, near = ch
    if kind == "E":
        i, j = divmod(idx, 3)
        # supporting edges, then the closest points of the two segments (brute force: dense sampling + refinement)
        pA = np.array([0.0 if m == i else (sT[m] if L[m] >= 0 else -sT[m]) for m in range(3)])
        pB = cg.copy()
        for m in range(3):
            if m != j:
                pB += (-s if L @ RTB[:, m] >= 0 else s) * RTB[:, m]
        ei, bj = np.eye(3)[i], RTB[:, j]
        A = np.array([[1.0, -(ei @ bj)], [-(ei @ bj), 1.0]])
        w = pA - pB
        al, be = np.linalg.solve(A, np.array([-(w @ ei), w @ bj]))
        al, be = np.clip(al, -sT[i], sT[i]), np.clip(be, -s, s)
        qa, qb = pA + al * ei, pB + be * bj
        if not sep < margin:
            return None
        return dict(kind=kind, index=idx, normal=L, points=np.array([(qa + qb) / 2]), dists=np.array([sep]), near_tie=near)
    half_b = np.full(3, s)
    if kind == "T":
        ref_c, ref_axes, ref_half = np.zeros(3), np.eye(3), np.asarray(sT, float)
        inc, _, _ = face_of(cg, RTB, half_b, -L)
        nref = L
    else:
        ref_c, ref_axes, ref_half = cg, RTB, half_b
        inc, _, _ = face_of(np.zeros(3), np.eye(3), np.asarray(sT, float), L)
        nref = -L  # outward normal of B's reference face points towards T
    k = idx
    a1, a2 = (k + 1) % 3, (k + 2) % 3
    poly = list(inc)
    for ax, h in ((a1, ref_half[a1]), (a2, ref_half[a2])):
        for sg in (1.0, -1.0):
            n = sg * ref_axes[:, ax]
            poly = clip_polygon(poly, n, n @ ref_c + h)
            if not poly:
                return dict(kind=kind, index=idx, normal=L, points=np.zeros((0, 3)), dists=np.zeros(0), near_tie=near)
    pts, ds = [], []
    for p in poly:
        g = nref @ (p - ref_c) - ref_half[k]  # signed distance of the incident point to the reference face plane
        if g < margin:
            pts.append(p - nref * g / 2)
            ds.append(g)
    # merge duplicates (a vertex exactly on a clip plane appears twice in Sutherland-Hodgman output)
    uniq_p, uniq_d = [], []
    for p, g in zip(pts, ds):
        if not any(np.abs(p - q).max() < 1e-12 for q in uniq_p):
            uniq_p.append(p); uniq_d.append(g)
    return dict(kind=kind, index=idx, normal=L, points=np.array(uniq_p).reshape(-1, 3), dists=np.array(uniq_d), near_tie=near)


def box_distance(sT, s, cg, RTB, iters=4000):
    """Euclidean distance between the two boxes (0 if they overlap): alternating projections (tests only; slow)"""
    x = np.zeros(3)
    y = cg.copy()
    for _ in range(iters):
        x = np.clip(y, -np.asarray(sT), np.asarray(sT))
        y = RTB @ np.clip(RTB.T @ (x - cg), -s, s) + cg
    return float(np.linalg.norm(x - y))

"""Fixed probe poses for the narrow-phase routines restated from MuJoCo (oracle/mzo_physics.c: capsule_box = mjc_CapsuleBox,
box_box = mjc_BoxBox, capsule_capsule = mjc_CapsuleCapsule, plane-box inside collide_plane = mjc_PlaneBox) with the contact sets DERIVED BY HAND
from the routines' documented construction (closest feature, second support point, face clipping).  Shared by
tests/test_narrowphase_probes.py (the oracle must produce exactly these) and tests/test_mujoco_crosscheck.py (real MuJoCo,
where available, must produce them too — a per-routine verdict on the restatement).

A probe = (name, kind, geom1 (pos, mat, size), geom2 (pos, mat, size), margin, expected rows [dist, px, py, pz, nx, ny, nz]);
the normal points from geom1 to geom2; rows are compared as sets (order is not part of the contract)."""
import numpy as np

I3 = np.eye(3)


def roty(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


BOX = ([0.0, 0.0, 0.0], I3, [1.0, 1.0, 1.0])
CAP = [0.1, 0.5]  # radius, half length
CAPX = roty(np.pi / 2)  # geom z axis -> +x
CUBE = [0.5, 0.5, 0.5]


def _seesaw(deg):
    """Capsule crossing the box edge (x = 1, z = 1) tangentially, descending towards +x at `deg` degrees, its middle 0.12 above the edge."""
    a = np.radians(deg)
    n = np.array([np.sin(a), 0.0, np.cos(a)])
    ctr = np.array([1.0, 0.0, 1.0]) + 0.12 * n
    first = [0.02, *(np.array([1.0, 0.0, 1.0]) + 0.01 * n), *(-n)]
    return ctr, roty(np.pi / 2 + a), first, a


def probes():
    out = []
    # ---- mjc_CapsuleBox
    out.append(("capsule flat over a face: both ends", "capsule_box", ([0, 0, 1.12], CAPX, CAP), BOX, 0.05,
                [[0.02, -0.5, 0, 1.01, 0, 0, -1], [0.02, 0.5, 0, 1.01, 0, 0, -1]]))
    out.append(("capsule flat, overhanging the face's edge: second point pulled in to the edge", "capsule_box", ([0.8, 0, 1.12], CAPX, CAP), BOX, 0.05,
                [[0.02, 0.3, 0, 1.01, 0, 0, -1], [0.02, 1.0, 0, 1.01, 0, 0, -1]]))
    out.append(("capsule standing on a face: the far end is beyond the margin", "capsule_box", ([0, 0, 1.62], I3, CAP), BOX, 0.05,
                [[0.02, 0, 0, 1.01, 0, 0, -1]]))
    ctr, m, first, a = _seesaw(2.0)
    end = ctr - 0.5 * np.array([np.cos(a), 0.0, -np.sin(a)])  # the end over the top face
    d2 = end[2] - 1.0 - 0.1
    out.append(("capsule across an edge at 2 degrees: second point over the flatter face", "capsule_box", (ctr, m, CAP), BOX, 0.05,
                [first, [d2, end[0], 0.0, 1.0 + 0.5 * d2, 0, 0, -1]]))
    ctr, m, first, a = _seesaw(20.0)
    out.append(("capsule across an edge at 20 degrees: the second point leaves the margin", "capsule_box", (ctr, m, CAP), BOX, 0.05, [first]))
    out.append(("capsule pointing at a corner: one contact", "capsule_box",
                (np.array([1.0, 1.0, 1.0]) + (0.5 + 0.12) / np.sqrt(3.0) * np.ones(3), _z_to(np.ones(3)), CAP), BOX, 0.05,
                [[0.02, *(np.ones(3) + 0.01 / np.sqrt(3.0)), *(-np.ones(3) / np.sqrt(3.0))]]))
    out.append(("capsule end inside the box: pushed out through the nearest face; the segment's exit point is the second contact", "capsule_box",
                ([0.0, 0.0, 1.3], I3, CAP), BOX, 0.05,
                [[-0.3, 0, 0, 0.85, 0, 0, -1], [-0.1, 0, 0, 0.95, 0, 0, -1]]))
    # ---- mjc_BoxBox
    out.append(("aligned cubes, face to face, partial overlap: the corners of the overlap rectangle", "box_box", ([0, 0, 0], I3, CUBE), ([0.99, 0.3, 0], I3, CUBE), 0.02,
                [[-0.01, 0.495, y, z, 1, 0, 0] for y in (-0.2, 0.5) for z in (-0.5, 0.5)]))
    out.append(("aligned cubes sharing only an edge (grid-aligned diagonal neighbours): an intersection without area, no contact [ASSUME-12]", "box_box",
                ([0, 0, 0], I3, CUBE), ([1, 1, 0], I3, CUBE), 0.02, []))
    out.append(("aligned cubes overlapping by a hair past the edge: the strip's four corners", "box_box", ([0, 0, 0], I3, CUBE), ([1.0, 0.99, 0], I3, CUBE), 0.02,
                [[0.0, 0.5, y, z, 1, 0, 0] for y in (0.49, 0.5) for z in (-0.5, 0.5)]))
    out.append(("aligned cubes separated by less than the margin", "box_box", ([0, 0, 0], I3, CUBE), ([0.2, -1.015, 0.1], I3, CUBE), 0.02,
                [[0.015, x, -0.5075, z, 0, -1, 0] for x in (-0.3, 0.5) for z in (-0.4, 0.5)]))
    out.append(("aligned cubes beyond the margin: nothing", "box_box", ([0, 0, 0], I3, CUBE), ([1.03, 0, 0], I3, CUBE), 0.02, []))
    cx = 0.5 + 0.5 * np.sqrt(2.0) * np.cos(np.radians(15.0)) - 0.01
    yv = 0.5 * np.sqrt(2.0) * np.sin(np.radians(-15.0)) * -1.0  # the penetrating corner of the cube turned by 30 degrees
    out.append(("cube turned 30 degrees about z, one vertical edge 0.01 deep in the face: top and bottom of that edge", "box_box",
                ([0, 0, 0], I3, CUBE), ([cx, 0, 0], rotz(np.radians(30.0)), CUBE), 0.0,
                [[-0.01, 0.495, yv, z, 1, 0, 0] for z in (-0.5, 0.5)]))
    out.append(("small box on a big one (a block on its platform): four bottom corners, normal from geom1 up", "box_box",
                ([0, 0, 0], I3, [2.0, 2.0, 1.0]), ([0.5, -0.25, 1.495], I3, CUBE), 0.01,
                [[-0.005, x, y, 0.9975, 0, 0, 1] for x in (0.0, 1.0) for y in (-0.75, 0.25)]))
    # ---- boxes of GENERAL orientation (round 6: SPIN plates, user robots' box geoms)
    Rz = rotz(np.radians(30.0))
    out.append(("sphere beside a face of a box turned 30 degrees about z: the face's normal, turned with the box", "sphere_box",
                (Rz @ np.array([1.3, 0.2, 0.4]), I3, [0.25, 0, 0]), ([0, 0, 0], Rz, [1.0, 1.0, 1.0]), 0.1,
                [[0.05, *(Rz @ np.array([1.025, 0.2, 0.4])), *(Rz @ np.array([-1.0, 0.0, 0.0]))]]))
    # edge-edge: cube A turned 45 degrees about y (its top edge runs along y, 0.5 sqrt2 above its centre), cube B turned 45 degrees
    # about x (its lowest edge runs along x, 0.5 sqrt2 below its centre) 0.01 lower than touching: the edges cross at right angles
    # above the origin; every face axis penetrates by 0.36, the axis edge x edge = z by 0.01 — one contact midway between the edges
    h = 0.5 * np.sqrt(2.0)
    out.append(("two cubes crossing edge on edge (mjc_BoxBox's edge-edge case): one contact midway, normal along edge x edge", "box_box",
                ([0, 0, 0], roty(np.pi / 4), CUBE), ([0, 0, 2 * h - 0.01], rotx(np.pi / 4), CUBE), 0.0,
                [[-0.01, 0, 0, h - 0.005, 0, 0, 1]]))
    # ---- mjc_CapsuleCapsule: nearest points of the two axis segments, then sphere-sphere; parallel axes: the ends of the overlap
    capy = np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])  # geom z axis -> +y
    out.append(("capsules crossing at right angles, 0.25 apart: one contact between the axes", "capsule_capsule",
                ([0, 0, 0], CAPX, CAP), ([0, 0, 0.25], capy, CAP), 0.1, [[0.05, 0, 0, 0.125, 0, 0, 1]]))
    out.append(("parallel capsules, shifted by 0.2 along the axis: the two ends of the stretch they share", "capsule_capsule",
                ([0, 0, 0], CAPX, CAP), ([0.2, 0, 0.25], CAPX, CAP), 0.1, [[0.05, 0.5, 0, 0.125, 0, 0, 1], [0.05, -0.3, 0, 0.125, 0, 0, 1]]))
    nn = np.array([0.3, 0.0, 0.2]) / np.hypot(0.3, 0.2)
    dd = np.hypot(0.3, 0.2) - 0.2
    out.append(("skew capsules, both nearest points clamped to segment ends", "capsule_capsule",
                ([0, 0, 0], CAPX, CAP), ([0.8, 0, 0.7], I3, CAP), 0.2, [[dd, *(np.array([0.5, 0, 0]) + nn * (0.1 + 0.5 * dd)), *nn]]))
    return out


def rotx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _z_to(v):
    """Rotation matrix whose third column is v / |v|."""
    z = np.asarray(v, float) / np.linalg.norm(v)
    x = np.cross([0.0, 1.0, 0.0], z)
    x /= np.linalg.norm(x)
    return np.stack([x, np.cross(z, x), z], axis=1)


def same_contact_set(got, want, tol=1e-9):
    got, want = np.asarray(got, float).reshape(-1, 7), np.asarray(want, float).reshape(-1, 7)
    if len(got) != len(want):
        return False
    used = set()
    for w in want:
        hit = [i for i, g in enumerate(got) if i not in used and np.all(np.abs(g - w) <= tol)]
        if not hit:
            return False
        used.add(hit[0])
    return True

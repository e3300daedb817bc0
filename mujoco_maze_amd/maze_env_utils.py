"""Maze cell vocabulary and the wall-segment geometry of the Point robot's manual collision path.

Host-side mirror of the plugin surface the reference exposes in `mujoco_maze/maze_env_utils.py` (`MazeCell` :19-81,
`Line` :84-128, `Collision` :131-142, `CollisionDetector` :145-206) so that user-defined tasks and code written
against those names keep working.  The device path consumes only the segment table built here
(`CollisionDetector.segments`); the per-move sweep runs inside the HIP kernel (`csrc/point_dyn.h: point_detect`).
`detect` exists for single moves on the host and is pinned bit for bit by golden vectors captured from the
reference (`tests/golden/detect.npz`, `tests/golden/line_kat.json`).

Points are Python complex numbers on the public surface, as in the reference; internally every product is written
out on (x, y) pairs in the order CPython's complex arithmetic evaluates it, which is what makes float64 results
identical to the last bit.
"""
from enum import Enum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

Point = complex


class MazeCell(Enum):
    """Cell codes of a maze grid (values as in the reference, `maze_env_utils.py:19-33`)."""

    ROBOT = -1  # start cell
    EMPTY = 0
    BLOCK = 1
    CHASM = 2
    OBJECT_BALL = 3
    XY_BLOCK = 14
    XZ_BLOCK = 15
    YZ_BLOCK = 16
    XYZ_BLOCK = 17
    XY_HALF_BLOCK = 18
    SPIN = 19


# Predicate table: method name -> the cells it holds for (reference maze_env_utils.py:35-81).  The methods are attached
# to the enum below, so `cell.can_move_x()` etc. read exactly like the reference's.
_C = MazeCell
_SLIDES_X = {_C.XY_BLOCK, _C.XY_HALF_BLOCK, _C.XZ_BLOCK, _C.XYZ_BLOCK, _C.SPIN}
_SLIDES_Y = {_C.XY_BLOCK, _C.XY_HALF_BLOCK, _C.YZ_BLOCK, _C.XYZ_BLOCK, _C.SPIN}
_SLIDES_Z = {_C.XZ_BLOCK, _C.YZ_BLOCK, _C.XYZ_BLOCK}
_PREDICATES = {
    "is_block": {_C.BLOCK},
    "is_chasm": {_C.CHASM},
    "is_object_ball": {_C.OBJECT_BALL},
    "is_empty": {_C.ROBOT, _C.EMPTY},
    "is_robot": {_C.ROBOT},
    "is_wall_or_chasm": {_C.BLOCK, _C.CHASM},
    "can_move_x": _SLIDES_X,
    "can_move_y": _SLIDES_Y,
    "can_move_z": _SLIDES_Z,
    "can_spin": {_C.SPIN},
    "can_move": _SLIDES_X | _SLIDES_Y | _SLIDES_Z,
    "is_half_block": {_C.XY_HALF_BLOCK},
}


def _predicate(members: frozenset):
    def holds(self) -> bool:
        return self in members

    return holds


for _name, _members in _PREDICATES.items():
    setattr(MazeCell, _name, _predicate(frozenset(_members)))


def _xy(p) -> Tuple[float, float]:
    if isinstance(p, complex):
        return p.real, p.imag
    return float(p[0]), float(p[1])


def _im_conj_mul(ax: float, ay: float, bx: float, by: float) -> float:
    """Im(conj(a) * b) in CPython's evaluation order: (ax - i ay)(bx + i by) -> ax*by + (-ay)*bx."""
    return ax * by + (-ay) * bx


class Line:
    """Directed segment p1 -> p2."""

    def __init__(self, p1: Union[Sequence[float], Point], p2: Union[Sequence[float], Point]) -> None:
        x1, y1 = _xy(p1)
        x2, y2 = _xy(p2)
        self.p1, self.p2 = complex(x1, y1), complex(x2, y2)
        self.v1 = self.p2 - self.p1
        self.conj_v1 = self.v1.conjugate()
        self.norm = abs(self.v1)

    # -- side tests ---------------------------------------------------------------------------------------
    def _side(self, q: Point) -> float:
        d = q - self.p1
        return _im_conj_mul(self.v1.real, self.v1.imag, d.real, d.imag)

    def _intersect(self, other: "Line") -> bool:
        """True when other's end points are on opposite sides of (or on) this line; touching counts."""
        return self._side(other.p1) * self._side(other.p2) <= 0.0

    def intersect(self, other: "Line") -> Optional[Point]:
        if not (self._intersect(other) and other._intersect(self)):
            return None
        return self._cross_point(other)

    def _cross_point(self, other: "Line") -> Point:
        w = other.p2 - other.p1
        u = self.p2 - other.p1
        denom = _im_conj_mul(self.v1.real, self.v1.imag, w.real, w.imag)
        numer = _im_conj_mul(self.v1.real, self.v1.imag, u.real, u.imag)
        return other.p1 + numer / denom * w  # ZeroDivisionError for collinear segments, as in the reference

    # -- metric helpers -----------------------------------------------------------------------------------
    def _projection(self, p: Point) -> Point:
        back = -self.v1
        d = p - self.p1
        scale = (d.real * back.real - (-d.imag) * back.imag) / abs(back) ** 2  # Re(conj(d) * back) / |back|^2
        return self.p1 + back * scale

    def reflection(self, p: Point) -> Point:
        foot = self._projection(p)
        return p + 2.0 * (foot - p)

    def distance(self, p: Point) -> float:
        return abs(p - self._projection(p))

    def __repr__(self) -> str:
        return f"Line(({self.p1.real}, {self.p1.imag}) -> ({self.p2.real}, {self.p2.imag}))"


class Collision:
    """A wall hit: where the move crosses the wall, and the mirror image of the move's end point."""

    def __init__(self, point: Point, reflection: Point) -> None:
        self._point, self._reflection = point, reflection

    @property
    def point(self) -> np.ndarray:
        return np.array(_xy(self._point))

    def rest(self) -> np.ndarray:
        return np.array(_xy(self._reflection - self._point))


class CollisionDetector:
    """The faces of BLOCK cells that border an empty cell, each pushed out by `radius` along both axes (so that the
    faces overhang the corners by `radius`), in row-major cell order x NEIGHBORS order."""

    EPS: float = 0.05
    NEIGHBORS: List[Tuple[int, int]] = [[0, -1], [-1, 0], [0, 1], [1, 0]]  # (dx, dy)

    def __init__(self, structure: list, size_scaling: float, torso_x: float, torso_y: float, radius: float) -> None:
        self.lines: List[Line] = []
        n_rows, n_cols = len(structure), len(structure[0])
        half = size_scaling * 0.5 + radius

        def free(i: int, j: int) -> bool:
            return 0 <= i < n_rows and 0 <= j < n_cols and structure[i][j].is_empty()

        for i, row in enumerate(structure):
            for j, cell in enumerate(row):
                if not cell.is_block():
                    continue
                x_lo, x_hi = j * size_scaling - torso_x - half, j * size_scaling - torso_x + half
                y_lo, y_hi = i * size_scaling - torso_y - half, i * size_scaling - torso_y + half
                for dx, dy in self.NEIGHBORS:
                    if free(i + dy, j + dx):
                        a = (x_hi if dx == 1 else x_lo, y_hi if dy == 1 else y_lo)
                        b = (x_lo if dx == -1 else x_hi, y_lo if dy == -1 else y_hi)
                        self.lines.append(Line(a, b))

    @property
    def segments(self) -> np.ndarray:
        """[S, 4] float64 table (x1, y1, x2, y2): what the device kernel reads."""
        return np.array([[ln.p1.real, ln.p1.imag, ln.p2.real, ln.p2.imag] for ln in self.lines], dtype=np.float64).reshape(-1, 4)

    def detect(self, old_pos: np.ndarray, new_pos: np.ndarray) -> Optional[Collision]:
        """First wall crossed by the move old_pos -> new_pos (smallest distance from old_pos; the earlier wall wins
        ties), or None; moves shorter than 1e-8 never collide."""
        move = Line(old_pos, new_pos)
        if move.norm <= 1e-8:
            return None
        winner: Optional[Collision] = None
        winner_dist = 0.0
        for wall in self.lines:
            crossing = wall.intersect(move)
            if crossing is None:
                continue
            dist = abs(crossing - move.p1)
            if winner is None or dist < winner_dist:
                winner, winner_dist = Collision(crossing, wall.reflection(move.p2)), dist
        return winner

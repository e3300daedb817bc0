"""Maze cell vocabulary and the 2-D wall-segment geometry of the manual
(Point-robot) collision path.

Host-side mirror of the reference's plugin surface
(`mujoco_maze/maze_env_utils.py:19-81` MazeCell, `:84-128` Line, `:131-142`
Collision, `:145-206` CollisionDetector).  The batched device path consumes only
the *segment table* this module builds (`CollisionDetector.segments`); the
per-move sweep itself runs in the HIP kernel (`csrc/point_kernels.hip`).  The
`detect` method here exists so that user code written against the reference API
keeps working on single moves, and is exercised against golden vectors captured
from the reference (tests/golden/).

Arithmetic is done on plain (x, y) floats in the same operation order the
reference's complex-number expressions expand to, so results agree to the last
bit with the reference in float64.
"""

from enum import Enum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

Point = complex


class MazeCell(Enum):
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

    # -- predicates (reference maze_env_utils.py:35-81) --------------------
    def is_block(self) -> bool:
        return self is MazeCell.BLOCK

    def is_chasm(self) -> bool:
        return self is MazeCell.CHASM

    def is_object_ball(self) -> bool:
        return self is MazeCell.OBJECT_BALL

    def is_empty(self) -> bool:
        return self in _EMPTY

    def is_robot(self) -> bool:
        return self is MazeCell.ROBOT

    def is_wall_or_chasm(self) -> bool:
        return self in _WALL_OR_CHASM

    def can_move_x(self) -> bool:
        return self in _MOVE_X

    def can_move_y(self) -> bool:
        return self in _MOVE_Y

    def can_move_z(self) -> bool:
        return self in _MOVE_Z

    def can_spin(self) -> bool:
        return self is MazeCell.SPIN

    def can_move(self) -> bool:
        return self in _MOVE_ANY

    def is_half_block(self) -> bool:
        return self is MazeCell.XY_HALF_BLOCK


_EMPTY = frozenset({MazeCell.ROBOT, MazeCell.EMPTY})
_WALL_OR_CHASM = frozenset({MazeCell.BLOCK, MazeCell.CHASM})
_MOVE_X = frozenset(
    {MazeCell.XY_BLOCK, MazeCell.XY_HALF_BLOCK, MazeCell.XZ_BLOCK, MazeCell.XYZ_BLOCK, MazeCell.SPIN}
)
_MOVE_Y = frozenset(
    {MazeCell.XY_BLOCK, MazeCell.XY_HALF_BLOCK, MazeCell.YZ_BLOCK, MazeCell.XYZ_BLOCK, MazeCell.SPIN}
)
_MOVE_Z = frozenset({MazeCell.XZ_BLOCK, MazeCell.YZ_BLOCK, MazeCell.XYZ_BLOCK})
_MOVE_ANY = _MOVE_X | _MOVE_Y | _MOVE_Z


def _as_point(p) -> complex:
    return p if isinstance(p, complex) else complex(float(p[0]), float(p[1]))


def _cross(ax: float, ay: float, bx: float, by: float) -> float:
    # Im(conj(a) * b) expanded exactly as CPython evaluates complex multiply:
    # (ax - i ay)(bx + i by) -> imag = ax*by + (-ay)*bx
    return ax * by + (-ay) * bx


class Line:
    """Directed segment p1 -> p2 (reference `Line`, maze_env_utils.py:84-128)."""

    def __init__(self, p1: Union[Sequence[float], Point], p2: Union[Sequence[float], Point]) -> None:
        self.p1 = _as_point(p1)
        self.p2 = _as_point(p2)
        self.v1 = self.p2 - self.p1
        self.conj_v1 = self.v1.conjugate()
        self.norm = abs(self.v1)

    def _intersect(self, other: "Line") -> bool:
        # Do other's end points lie on opposite sides (or on) of self's line?
        vx, vy = self.v1.real, self.v1.imag
        a = other.p1 - self.p1
        b = other.p2 - self.p1
        return _cross(vx, vy, a.real, a.imag) * _cross(vx, vy, b.real, b.imag) <= 0.0

    def _projection(self, p: Point) -> Point:
        back = -self.v1
        n2 = abs(back) ** 2
        d = p - self.p1
        # Re(conj(d) * back)
        scale = (d.real * back.real - (-d.imag) * back.imag) / n2
        return self.p1 + back * scale

    def reflection(self, p: Point) -> Point:
        return p + 2.0 * (self._projection(p) - p)

    def distance(self, p: Point) -> float:
        return abs(p - self._projection(p))

    def intersect(self, other: "Line") -> Optional[Point]:
        if self._intersect(other) and other._intersect(self):
            return self._cross_point(other)
        return None

    def _cross_point(self, other: "Line") -> Point:
        vx, vy = self.v1.real, self.v1.imag
        w = other.p2 - other.p1
        u = self.p2 - other.p1
        a = _cross(vx, vy, w.real, w.imag)
        b = _cross(vx, vy, u.real, u.imag)
        return other.p1 + b / a * w  # ZeroDivisionError when collinear, as the reference

    def __repr__(self) -> str:
        return f"Line(({self.p1.real}, {self.p1.imag}) -> ({self.p2.real}, {self.p2.imag}))"


class Collision:
    def __init__(self, point: Point, reflection: Point) -> None:
        self._point = point
        self._reflection = reflection

    @property
    def point(self) -> np.ndarray:
        return np.array([self._point.real, self._point.imag])

    def rest(self) -> np.ndarray:
        d = self._reflection - self._point
        return np.array([d.real, d.imag])


class CollisionDetector:
    """Wall faces of BLOCK cells that border an empty cell, pushed out by
    `radius` (reference maze_env_utils.py:145-206)."""

    EPS: float = 0.05
    NEIGHBORS: List[Tuple[int, int]] = [[0, -1], [-1, 0], [0, 1], [1, 0]]

    def __init__(self, structure: list, size_scaling: float, torso_x: float, torso_y: float, radius: float) -> None:
        self.lines: List[Line] = []
        rows, cols = len(structure), len(structure[0])
        reach = size_scaling * 0.5 + radius
        for i in range(rows):
            for j in range(cols):
                if not structure[i][j].is_block():
                    continue
                cy = i * size_scaling - torso_y
                cx = j * size_scaling - torso_x
                lo_y, hi_y = cy - reach, cy + reach
                lo_x, hi_x = cx - reach, cx + reach
                for dx, dy in self.NEIGHBORS:
                    ni, nj = i + dy, j + dx
                    if not (0 <= ni < rows and 0 <= nj < cols and structure[ni][nj].is_empty()):
                        continue
                    start = (hi_x if dx == 1 else lo_x, hi_y if dy == 1 else lo_y)
                    end = (lo_x if dx == -1 else hi_x, lo_y if dy == -1 else hi_y)
                    self.lines.append(Line(start, end))

    @property
    def segments(self) -> np.ndarray:
        """[S, 4] float64 table (x1, y1, x2, y2) — what the device kernel reads."""
        out = np.zeros((len(self.lines), 4), dtype=np.float64)
        for k, ln in enumerate(self.lines):
            out[k] = (ln.p1.real, ln.p1.imag, ln.p2.real, ln.p2.imag)
        return out

    def detect(self, old_pos: np.ndarray, new_pos: np.ndarray) -> Optional[Collision]:
        move = Line(old_pos, new_pos)
        if move.norm <= 1e-8:
            return None
        best, best_dist = None, None
        for wall in self.lines:
            hit = wall.intersect(move)
            if hit is None:
                continue
            dist = abs(hit - move.p1)
            if best is None or dist < best_dist:
                best, best_dist = Collision(hit, wall.reflection(move.p2)), dist
        return best

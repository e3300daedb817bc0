"""Render / export bridge (reference: MazeEnv.render, maze_env.py:389-420, websock_viewer.py).

The reference renders through mujoco-py's offscreen context (a 3-D camera image) and can push that image to a browser over a
websocket.  Neither mujoco-py nor a GL context exists next to a batch of device-resident envs, so the bridge here is:

* `render_top_down(cm, qpos, ...)`  — a software rasteriser (numpy only) that draws ONE env's state pulled from the device as
  an orthographic top view: floor, walls, chasms, goal sites (the task's colours and sizes), movable blocks, object balls and
  the robot (ant: torso + four legs from its joint angles; point: disc + heading arrow; swimmer / reacher: the link chain).
  It returns an `uint8 [H, W, 3]` array like the reference's `_render_image` (maze_env.py:389-393) — not the same pixels: a
  different camera, flat shading.
* `state_for_viewer(cm, qpos, qvel)` — the env's MJCF (`mjcf.world_to_mjcf`, the model this repository steps) plus its
  qpos / qvel as plain lists, for anyone who wants to replay device states in a real MuJoCo viewer.

Host-side and test-covered on CPU; it never touches the step path."""
import math
from typing import Sequence, Tuple

import numpy as np

from mujoco_maze_amd.model import CompiledModel

FLOOR = (232, 226, 214)
WALL = (110, 110, 118)
CHASM = (30, 30, 36)
BLOCK = (230, 26, 26)     # rgba 0.9 0.1 0.1 (maze_env.py:603)
BALL = (26, 26, 179)      # maze_task.BLUE
ROBOT = (204, 153, 102)   # the assets' geom colour 0.8 0.6 0.4
DARK = (60, 40, 20)


class _Canvas:
    """World-coordinate painter on an RGB image: x to the right, y UP (the maze's row index grows with y, as in MuJoCo)."""

    def __init__(self, xlim: Tuple[float, float], ylim: Tuple[float, float], shape: Tuple[int, int]):
        self.w, self.h = int(shape[0]), int(shape[1])
        sx, sy = (self.w - 1) / (xlim[1] - xlim[0]), (self.h - 1) / (ylim[1] - ylim[0])
        self.s = min(sx, sy)
        self.x0 = xlim[0] - 0.5 * ((self.w - 1) / self.s - (xlim[1] - xlim[0]))
        self.y0 = ylim[0] - 0.5 * ((self.h - 1) / self.s - (ylim[1] - ylim[0]))
        self.img = np.empty((self.h, self.w, 3), np.uint8)
        self.img[:] = FLOOR
        ys, xs = np.mgrid[0:self.h, 0:self.w]
        self.X = self.x0 + xs / self.s
        self.Y = self.y0 + (self.h - 1 - ys) / self.s

    def rect(self, cx, cy, hx, hy, colour):
        self.img[(np.abs(self.X - cx) <= hx) & (np.abs(self.Y - cy) <= hy)] = colour

    def disc(self, cx, cy, r, colour):
        self.img[(self.X - cx) ** 2 + (self.Y - cy) ** 2 <= r * r] = colour

    def ring(self, cx, cy, r, colour, width=0.06):
        d = np.sqrt((self.X - cx) ** 2 + (self.Y - cy) ** 2)
        self.img[np.abs(d - r) <= 0.5 * max(width, 1.5 / self.s)] = colour

    def segment(self, ax, ay, bx, by, r, colour):
        """capsule from a to b with radius r"""
        dx, dy = bx - ax, by - ay
        ll = dx * dx + dy * dy
        t = np.clip(((self.X - ax) * dx + (self.Y - ay) * dy) / ll, 0.0, 1.0) if ll > 0 else 0.0
        r = max(r, 1.0 / self.s)
        self.img[(self.X - ax - t * dx) ** 2 + (self.Y - ay - t * dy) ** 2 <= r * r] = colour


def _rgb(c) -> Tuple[int, int, int]:
    return int(round(255 * c.red)), int(round(255 * c.green)), int(round(255 * c.blue))


def _yaw_from_quat(q: Sequence[float]) -> float:
    w, x, y, z = q
    return math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))


def _block_xy(cm: CompiledModel, qpos: np.ndarray, body: int) -> Tuple[float, float]:
    m = cm.c
    p = [m.body_pos[body][0], m.body_pos[body][1]]
    for j in range(m.body_jntadr[body], m.body_jntadr[body] + m.body_jntnum[body]):
        if m.jnt_type[j] == 0:  # free joint (the Ant's object ball): qpos holds the absolute pose, not a displacement
            a = m.jnt_qposadr[j]
            return float(qpos[a]), float(qpos[a + 1])
        if m.jnt_type[j] != 2:  # hinges do not move the body origin
            continue
        q = float(qpos[m.jnt_qposadr[j]]) - m.qpos0[m.jnt_qposadr[j]]
        p[0] += m.jnt_axis[j][0] * q
        p[1] += m.jnt_axis[j][1] * q
    return p[0], p[1]


def render_top_down(cm: CompiledModel, qpos: Sequence[float], image_shape: Tuple[int, int] = (600, 480)) -> np.ndarray:
    """Top view of one env: `qpos` is the env's generalized position (row of VecMazeEnv.get_state()[0])."""
    m, world, task = cm.c, cm.world, cm.task
    qpos = np.asarray(qpos, np.float64)
    s = world.scale
    xlim = (-0.5 * s - m.torso_x, (world.cols - 0.5) * s - m.torso_x)
    ylim = (-0.5 * s - m.torso_y, (world.rows - 0.5) * s - m.torso_y)
    cv = _Canvas(xlim, ylim, image_shape)
    for i in range(world.rows):
        for j in range(world.cols):
            cell = world.structure[i][j]
            if cell.is_block() or cell.is_chasm():
                cv.rect(j * s - m.torso_x, i * s - m.torso_y, 0.5 * s, 0.5 * s, WALL if cell.is_block() else CHASM)
    for g in task.goals:  # sites: spheres of radius custom_size or scale * 0.1 (maze_env.py:203-216)
        size = g.custom_size if g.custom_size is not None else s * 0.1
        cv.disc(g.pos[0], g.pos[1], size, _rgb(g.rgb))
        cv.ring(g.pos[0], g.pos[1], g.threshold, _rgb(g.rgb))
    for k in range(m.nblock):
        b, gid = m.block_bodyid[k], m.block_geomid[k]
        x, y = _block_xy(cm, qpos, b)
        cv.rect(x, y, m.geom_size[gid][0], m.geom_size[gid][1], BLOCK)
    for k in range(m.nball):
        b, gid = m.ball_bodyid[k], m.ball_geomid[k]
        x, y = _block_xy(cm, qpos, b)
        cv.disc(x, y, m.geom_size[gid][0], BALL)
    x, y = float(qpos[0]), float(qpos[1])
    robot = cm.spec.name
    if robot == "ant":
        yaw = _yaw_from_quat(qpos[3:7])
        for leg, (sx_, sy_) in enumerate(((1, 1), (-1, 1), (-1, -1), (1, -1))):  # ant.xml: front-left, front-right, back, right-back
            hip, ank = float(qpos[7 + 2 * leg]), float(qpos[8 + 2 * leg])
            a0 = yaw + math.atan2(sy_, sx_)
            kx, ky = x + 0.2 * math.sqrt(2) * math.cos(a0), y + 0.2 * math.sqrt(2) * math.sin(a0)
            a1 = a0 + hip
            jx, jy = kx + 0.2 * math.sqrt(2) * math.cos(a1), ky + 0.2 * math.sqrt(2) * math.sin(a1)
            reach = 0.4 * math.sqrt(2) * abs(math.cos(ank))  # the ankle swings in a vertical plane: its top view shortens
            fx, fy = jx + reach * math.cos(a1), jy + reach * math.sin(a1)
            cv.segment(x, y, kx, ky, 0.08, ROBOT)
            cv.segment(kx, ky, jx, jy, 0.08, ROBOT)
            cv.segment(jx, jy, fx, fy, 0.08, DARK)
        cv.disc(x, y, 0.25, ROBOT)
        cv.segment(x, y, x + 0.25 * math.cos(yaw), y + 0.25 * math.sin(yaw), 0.03, DARK)
    elif robot == "point":
        th = float(qpos[2])
        cv.disc(x, y, 0.5, ROBOT)  # point.xml: sphere size 0.5 (the manual collision radius is 0.4)
        cv.segment(x, y, x + 0.6 * math.cos(th), y + 0.6 * math.sin(th), 0.06, DARK)
    else:  # swimmer (3 links) / reacher (2 links): unit-length capsules chained by the hinge angles
        nlink = m.nbody - 1 - m.nblock - m.nball  # links of the chain: user MJCF may bring 2..6 (mjcf.py)
        th = float(qpos[2])
        ax, ay = x, y
        # swimmer.xml: the first link extends from the torso origin along -x of the body frame
        for k in range(nlink):
            bx, by = ax - math.cos(th), ay - math.sin(th)
            cv.segment(ax, ay, bx, by, 0.1, ROBOT if k % 2 == 0 else DARK)
            ax, ay = bx, by
            if k + 1 < nlink:
                th += float(qpos[3 + k])
        cv.disc(x, y, 0.12, DARK)
    return cv.img


def state_for_viewer(cm: CompiledModel, qpos: Sequence[float], qvel: Sequence[float]) -> dict:
    """MJCF of the env's model + one state, as plain data (json-serialisable): load the xml in MuJoCo, assign qpos / qvel."""
    from mujoco_maze_amd import mjcf

    return dict(mjcf=mjcf.world_to_mjcf(cm), qpos=[float(v) for v in qpos], qvel=[float(v) for v in qvel])

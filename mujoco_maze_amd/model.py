"""Host-side model compiler: RobotSpec + maze task -> `mz_model` constant block.

Plays the role that `MazeEnv.__init__` (reference `mujoco_maze/maze_env.py:28-233`:
world generation from the cell grid) *plus* the MuJoCo model compiler (mass /
inertia from geoms, `qpos0`, `dof_invweight0`, `body_invweight0`) play in the
reference.  Cold path, pure numpy float64.  The resulting struct is handed to
the C-ABI (`include/mazestep.h: mz_model`) and — in tests only — to the oracle.

MuJoCo facts restated here (no MuJoCo source is available offline; SURVEY §8a
M-rows; each is an assumption recorded in DESIGN.md):
  * capsule inertia = cylinder + two hemispheres; `fromto` -> pos/quat/half-length
  * body inertial frame = mass-weighted combination of its geoms
  * dof_invweight0[i] = (M^-1)_ii at qpos0, averaged over the 3 translational /
    3 rotational dofs of a free joint
  * body_invweight0[b] = mean diagonal of J M^-1 J^T for the translational /
    rotational Jacobian of the body's COM frame at qpos0
"""
import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np

from mujoco_maze_amd import robots as R
from mujoco_maze_amd.maze_env_utils import CollisionDetector, MazeCell
from mujoco_maze_amd.maze_task import MazeTask, device_reward_descriptor

MZ_ABI_VERSION = 8
MAX_BODY, MAX_JNT, MAX_DOF, MAX_Q, MAX_GEOM, MAX_ACT = 24, 24, 24, 28, 24, 8
MAX_GRID, MAX_SEG, MAX_GOAL, MAX_OBS = 12, 96, 8, 48
VIEW_DIM = 75  # MZ_VIEW_DIM: the 5 x 5 x 3 top-down view (maze_env.py:95)

i32, f64, u8 = C.c_int32, C.c_double, C.c_uint8


class MzModel(C.Structure):
    """ctypes mirror of `struct mz_model` (include/mazestep.h) — keep in sync."""

    _fields_ = [
        ("abi_version", i32), ("robot", i32),
        ("nbody", i32), ("njnt", i32), ("nq", i32), ("nv", i32), ("ngeom", i32), ("nu", i32),
        ("nq_robot", i32), ("nv_robot", i32),
        ("frame_skip", i32), ("integrator_rk4", i32), ("collision_predefined", i32), ("manual_collision", i32),
        ("max_episode_steps", i32), ("obs_dim", i32), ("reset_qvel_kind", i32), ("engine", i32),
        ("timestep", f64), ("gravity", f64 * 3), ("density", f64), ("viscosity", f64), ("meaninertia", f64),
        ("body_parent", i32 * MAX_BODY), ("body_jntadr", i32 * MAX_BODY), ("body_jntnum", i32 * MAX_BODY),
        ("body_dofadr", i32 * MAX_BODY), ("body_dofnum", i32 * MAX_BODY),
        ("body_pos", (f64 * 3) * MAX_BODY), ("body_quat", (f64 * 4) * MAX_BODY), ("body_ipos", (f64 * 3) * MAX_BODY),
        ("body_inertia", (f64 * 6) * MAX_BODY), ("body_iquat", (f64 * 4) * MAX_BODY), ("body_pinertia", (f64 * 3) * MAX_BODY),
        ("body_mass", f64 * MAX_BODY), ("body_invweight0", (f64 * 2) * MAX_BODY),
        ("jnt_type", i32 * MAX_JNT), ("jnt_qposadr", i32 * MAX_JNT), ("jnt_dofadr", i32 * MAX_JNT),
        ("jnt_bodyid", i32 * MAX_JNT), ("jnt_limited", i32 * MAX_JNT),
        ("jnt_pos", (f64 * 3) * MAX_JNT), ("jnt_axis", (f64 * 3) * MAX_JNT), ("jnt_range", (f64 * 2) * MAX_JNT),
        ("jnt_margin", f64 * MAX_JNT), ("jnt_solref", (f64 * 2) * MAX_JNT), ("jnt_solimp", (f64 * 5) * MAX_JNT),
        ("jnt_stiffness", f64 * MAX_JNT), ("jnt_springref", f64 * MAX_JNT),
        ("dof_bodyid", i32 * MAX_DOF), ("dof_jntid", i32 * MAX_DOF),
        ("dof_armature", f64 * MAX_DOF), ("dof_damping", f64 * MAX_DOF), ("dof_invweight0", f64 * MAX_DOF),
        ("qpos0", f64 * MAX_Q),
        ("geom_type", i32 * MAX_GEOM), ("geom_bodyid", i32 * MAX_GEOM), ("geom_contype", i32 * MAX_GEOM),
        ("geom_conaffinity", i32 * MAX_GEOM), ("geom_condim", i32 * MAX_GEOM),
        ("geom_pos", (f64 * 3) * MAX_GEOM), ("geom_quat", (f64 * 4) * MAX_GEOM), ("geom_size", (f64 * 3) * MAX_GEOM),
        ("geom_friction", (f64 * 3) * MAX_GEOM), ("geom_solref", (f64 * 2) * MAX_GEOM), ("geom_solimp", (f64 * 5) * MAX_GEOM),
        ("geom_margin", f64 * MAX_GEOM), ("geom_gap", f64 * MAX_GEOM), ("geom_rbound", f64 * MAX_GEOM),
        ("act_dofid", i32 * MAX_ACT), ("act_ctrllimited", i32 * MAX_ACT), ("act_gear", f64 * MAX_ACT),
        ("act_ctrlrange", (f64 * 2) * MAX_ACT), ("act_gainprm", f64 * MAX_ACT), ("act_biasprm", (f64 * 3) * MAX_ACT),
        ("grid_rows", i32), ("grid_cols", i32), ("grid", (u8 * MAX_GRID) * MAX_GRID),
        ("maze_scale", f64), ("torso_x", f64), ("torso_y", f64),
        ("wall_half_xy", f64), ("wall_half_z", f64), ("wall_center_z", f64),
        ("wall_contype", i32), ("wall_conaffinity", i32), ("wall_condim", i32), ("step_kind", i32),
        ("wall_friction", f64 * 3), ("wall_solref", f64 * 2), ("wall_solimp", f64 * 5),
        ("wall_margin", f64), ("wall_gap", f64),
        ("nseg", i32), ("pad2", i32), ("seg", (f64 * 4) * MAX_SEG), ("restitution", f64), ("velocity_limit", f64),
        ("ngoal", i32), ("reward_kind", i32), ("reward_slot", i32), ("reward_binary", i32), ("term_slot", i32),
        ("goal_dim", i32 * MAX_GOAL), ("goal_pos", (f64 * 3) * MAX_GOAL), ("goal_threshold", f64 * MAX_GOAL),
        ("goal_reward_scale", f64 * MAX_GOAL),
        ("penalty", f64), ("task_scale", f64), ("inner_reward_scaling", f64),
        ("forward_reward_weight", f64), ("ctrl_cost_weight", f64),
        ("nblock", i32), ("observe_blocks", i32), ("block_bodyid", i32 * 4), ("block_geomid", i32 * 4),
        ("nball", i32), ("observe_balls", i32), ("ball_bodyid", i32 * 4), ("ball_geomid", i32 * 4),
        ("elevated", i32), ("top_down_view", i32), ("height_offset", f64),
    ]


ROBOT_ID = {"point": 0, "ant": 1, "swimmer": 2, "reacher": 2, "generic": 3}  # the Reacher is the 2-link member of the swimmer family; "generic": a user robot of any tree topology
RESET_KIND = {"normal": 0, "uniform01": 1, "uniform_sym": 2}
CELL_CODE = {MazeCell.EMPTY: 0, MazeCell.BLOCK: 1, MazeCell.CHASM: 2, MazeCell.ROBOT: 255}


# ---------------------------------------------------------------- small math
def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-14:
        return np.array([1.0, 0.0, 0.0, 0.0])
    s = math.sin(angle / 2.0)
    return np.concatenate([[math.cos(angle / 2.0)], axis / n * s])


def quat_z_to_vec(vec):
    """Rotation taking +z onto `vec` (MuJoCo's fromto convention)."""
    vec = np.asarray(vec, dtype=np.float64)
    vec = vec / np.linalg.norm(vec)
    ax = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(ax)
    ang = math.atan2(s, vec[2])
    if s < 1e-10:
        return np.array([1.0, 0.0, 0.0, 0.0]) if vec[2] > 0 else np.array([0.0, 1.0, 0.0, 0.0])
    return axis_angle_quat(ax / s, ang)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


# ---------------------------------------------------------------- geoms -> inertia
def resolve_geom(g: R.GeomSpec):
    """Returns (pos, quat, size3) in the body frame."""
    size = [0.0, 0.0, 0.0]
    for k, v in enumerate(g.size[:3]):
        size[k] = float(v)
    if g.fromto is not None:
        a, b = np.array(g.fromto[:3], dtype=np.float64), np.array(g.fromto[3:], dtype=np.float64)
        pos = 0.5 * (a + b)
        quat = quat_z_to_vec(a - b)
        size[1] = 0.5 * np.linalg.norm(a - b)
    else:
        pos = np.array(g.pos, dtype=np.float64)
        quat = np.array(g.quat if getattr(g, "quat", None) is not None else [1.0, 0.0, 0.0, 0.0], dtype=np.float64)
    return pos, quat, size


def geom_mass_inertia(g: R.GeomSpec, size):
    """Mass and principal inertia (geom frame, about geom centre)."""
    if g.type == R.SPHERE:
        r = size[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        unit = np.array([0.4 * r * r] * 3)
    elif g.type == R.CAPSULE:
        r, hl = size[0], size[1]
        h = 2.0 * hl
        v_cyl, v_sph = math.pi * r * r * h, 4.0 / 3.0 * math.pi * r ** 3
        vol = v_cyl + v_sph
        lat = v_cyl * (3 * r * r + h * h) / 12.0 + v_sph * (0.4 * r * r + h * (0.375 * r + 0.25 * h))
        axial = v_cyl * r * r / 2.0 + v_sph * 0.4 * r * r
        unit = np.array([lat, lat, axial]) / vol
    elif g.type == R.BOX:
        a, b, c = size
        vol = 8.0 * a * b * c
        unit = np.array([b * b + c * c, a * a + c * c, a * a + b * b]) / 3.0
    else:
        return 0.0, np.zeros(3)
    mass = g.mass if g.mass is not None else g.density * vol
    return mass, unit * mass


def mat_to_quat(Rm):
    """Unit quaternion (w, x, y, z) of a proper rotation matrix."""
    t = np.trace(Rm)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (Rm[2, 1] - Rm[1, 2]) / s, (Rm[0, 2] - Rm[2, 0]) / s, (Rm[1, 0] - Rm[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(Rm)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + Rm[i, i] - Rm[j, j] - Rm[k, k]) * 2
        q = [0.0] * 4
        q[0] = (Rm[k, j] - Rm[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (Rm[j, i] + Rm[i, j]) / s
        q[1 + k] = (Rm[k, i] + Rm[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def principal_frame(I, geom_frames):
    """(iquat, principal moments) of a body's inertia tensor `I` (about its COM, body frame), the way MuJoCo's compiler
    stores it (mjModel.body_iquat / body_inertia): a body with a single geom inherits that geom's frame and moments as they
    are (so a capsule keeps its fromto frame, axial moment last); otherwise the tensor is diagonalised, moments in
    decreasing order, right-handed frame.  `geom_frames` = [(quat, principal moments)] of the body's geoms."""
    if len(geom_frames) == 1:
        return np.asarray(geom_frames[0][0], np.float64), np.asarray(geom_frames[0][1], np.float64)
    if len(geom_frames) == 0 or not np.any(I):
        return np.array([1.0, 0.0, 0.0, 0.0]), np.diag(I).copy()
    if np.allclose(I, np.diag(np.diag(I)), rtol=0.0, atol=1e-14 * max(1.0, np.abs(I).max())):  # already diagonal: keep the body frame
        return np.array([1.0, 0.0, 0.0, 0.0]), np.diag(I).copy()
    w, V = np.linalg.eigh(I)
    order = np.argsort(-w)
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return mat_to_quat(V), w


def geom_rbound(gtype, size):
    if gtype == R.SPHERE:
        return size[0]
    if gtype == R.CAPSULE:
        return size[0] + size[1]
    if gtype == R.BOX:
        return float(np.linalg.norm(size))
    return 0.0


# ---------------------------------------------------------------- world from the maze grid
class MazeWorld:
    """Cell grid -> world description (reference maze_env.py:57-68, 116-152)."""

    def __init__(self, structure: List[List[MazeCell]], scale: float, maze_height: float = 0.5, put_spin_near_agent: bool = False):
        self.structure = structure
        self.scale = float(scale)
        self.put_spin_near_agent = bool(put_spin_near_agent)  # MazeTask.PUT_SPIN_NEAR_AGENT (maze_task.py:67, maze_env.py:51,119-120)
        self.rows, self.cols = len(structure), len(structure[0])
        self.elevated = any(c is MazeCell.CHASM for row in structure for c in row)
        # maze_env.py:61: `self.blocks` looks at the structure as the task wrote it — a plate that PUT_SPIN_NEAR_AGENT drops
        # onto the robot's cell does NOT switch the default geoms to solimp .995
        self.has_blocks = any(c.can_move() for row in structure for c in row)
        self.has_balls = any(c.is_object_ball() for row in structure for c in row)
        robots = [(j * scale, i * scale) for i in range(self.rows) for j in range(self.cols) if structure[i][j].is_robot()]
        if not robots:
            raise ValueError("No robot in maze specification.")
        self.torso_x, self.torso_y = robots[0]
        self.init_positions = [(x - self.torso_x, y - self.torso_y) for x, y in robots]
        self.half_z = maze_height / 2.0 * scale
        self.height_offset = maze_height * scale if self.elevated else 0.0

    def cell_as_built(self, i, j):
        """The cell the world generator acts on (maze_env.py:116-120): a ROBOT cell becomes a SPIN plate under PUT_SPIN_NEAR_AGENT."""
        c = self.structure[i][j]
        return MazeCell.SPIN if (self.put_spin_near_agent and c.is_robot()) else c

    def movable_cells(self):
        """[(i, j, cell)] of movable-block cells in row-major order (maze_env.py:153-166)."""
        return [(i, j, self.cell_as_built(i, j)) for i in range(self.rows) for j in range(self.cols) if self.cell_as_built(i, j).can_move()]

    def ball_cells(self):
        """[(i, j)] of object-ball cells in row-major order (maze_env.py:167-191)."""
        return [(i, j) for i in range(self.rows) for j in range(self.cols) if self.structure[i][j].is_object_ball()]

    def cell_center(self, i, j):
        return j * self.scale - self.torso_x, i * self.scale - self.torso_y

    def wall_boxes(self):
        """[(x, y, z, hx, hy, hz)] in row-major cell order, as the reference emits them."""
        out = []
        for i in range(self.rows):
            for j in range(self.cols):
                if self.structure[i][j].is_block():
                    x, y = self.cell_center(i, j)
                    out.append((x, y, self.half_z + self.height_offset, self.scale * 0.5, self.scale * 0.5, self.half_z))
        return out

    def xy_limits(self):
        free = [(i, j) for i in range(self.rows) for j in range(self.cols) if not self.structure[i][j].is_block()]
        imin, imax = min(i for i, _ in free), max(i for i, _ in free)
        jmin, jmax = min(j for _, j in free), max(j for _, j in free)
        s = self.scale
        return ((jmin - 0.5) * s - self.torso_x, (jmax + 0.5) * s - self.torso_x,
                (imin - 0.5) * s - self.torso_y, (imax + 0.5) * s - self.torso_y)


# ---------------------------------------------------------------- what the device kernels can step
def needs_general_engine(cm) -> bool:
    """Mirror of csrc/generic_dyn.h gen_model_needs_general_engine: True when no specialised kernel steps this model (a user robot,
    a SPIN plate's ball joint, more than three movable blocks, blocks of several sizes, a three-slide block anywhere but the
    one-block ant) or the caller asked for the general engine."""
    m = cm.c
    if m.robot == ROBOT_ID["generic"] or m.engine == 1 or m.nblock > 3 or m.integrator_rk4 == 0:
        return True
    if any(m.jnt_type[j] == R.BALL or m.jnt_stiffness[j] != 0.0 for j in range(m.njnt)):
        return True
    if any(m.act_gainprm[a] != 1.0 or any(m.act_biasprm[a][k] != 0.0 for k in range(3)) for a in range(m.nu)):
        return True
    if any(m.body_jntnum[m.block_bodyid[k]] != 2 for k in range(m.nblock)) and not (m.robot == ROBOT_ID["ant"] and m.nblock == 1):
        return True
    sizes = {tuple(m.geom_size[m.block_geomid[k]]) for k in range(m.nblock)}
    return len(sizes) > 1


def device_unsupported_reason(cm) -> Optional[str]:
    """None when the HIP kernels can step this compiled model, else why not.  Since round 6 the general engine
    (csrc/generic_dyn.h) steps whatever the specialised kernels do not, so every model `compile_model` accepts has a device path;
    the library itself still refuses what exceeds its buffers (mz_create returns the reason)."""
    return None


# ---------------------------------------------------------------- compile
class CompiledModel:
    """The `mz_model` struct plus host-side metadata the façade needs."""

    def __init__(self, c: MzModel, spec: R.RobotSpec, world: MazeWorld, task: MazeTask, device_rewards: bool):
        self.c, self.spec, self.world, self.task = c, spec, world, task
        self.device_rewards = device_rewards

    def __getattr__(self, name):
        return getattr(self.c, name)


def _kinematics_qpos0(nbody, parent, bpos, bquat):
    xpos = np.zeros((nbody, 3))
    xquat = np.zeros((nbody, 4))
    xquat[0] = (1, 0, 0, 0)
    for b in range(1, nbody):
        p = parent[b]
        xpos[b] = xpos[p] + quat_to_mat(xquat[p]) @ bpos[b]
        xquat[b] = quat_mul(xquat[p], bquat[b])
    return xpos, xquat


def compile_model(robot: str, task: MazeTask, scale: float, *, inner_reward_scaling: Optional[float] = None,
                  restitution_coef: float = 0.8, maze_height: float = 0.5, max_episode_steps: int = 1000,
                  forward_reward_weight: float = 1.0, ctrl_cost_weight: float = 1e-4,
                  manual_collision: Optional[bool] = None, radius: Optional[float] = None,
                  robot_xml: Optional[str] = None, frame_skip: Optional[int] = None, reset_qvel: Optional[str] = None,
                  engine: str = "auto", step: Optional[str] = None, velocity_limit: Optional[float] = None) -> CompiledModel:
    """`robot_xml`: path or text of an MJCF variant of the built-in robot (see `mjcf.py`); default: the built-in asset data.
    `engine`: "auto" — the robot family's specialised kernels where they can step the model, the general engine
    (csrc/generic_dyn.h) where they cannot (SPIN plates, user robots, more than three blocks); "general" — the general engine
    whatever the robot.  `step` (user robots): "motors" (ant.py:61-73) or "point" (point.py:44-61)."""
    if engine not in ("auto", "general"):
        raise ValueError("engine must be \"auto\" or \"general\"")
    if step not in (None, "motors", "point"):
        raise ValueError("step must be \"motors\" or \"point\"")
    if robot == "generic":
        # a user's AgentModel (agent_model.py:12-41): nothing built in — the MJCF is the robot; frame_skip and the reset
        # distribution come from the AgentModel class (kwargs below)
        if robot_xml is None:
            raise ValueError("a generic robot needs its MJCF (AgentModel.FILE or robot_xml=...)")
        from mujoco_maze_amd import mjcf

        spec = mjcf.spec_from_mjcf(robot_xml, None, frame_skip=frame_skip or 1, reset_qvel=reset_qvel or "normal")
    else:
        spec = R.robot_spec(robot)
        if robot_xml is not None:
            from mujoco_maze_amd import mjcf

            spec = mjcf.spec_from_mjcf(robot_xml, spec)
    structure = task.create_maze()
    world = MazeWorld(structure, scale, maze_height, put_spin_near_agent=bool(getattr(task, "PUT_SPIN_NEAR_AGENT", False)))
    balls = world.ball_cells()
    if balls and robot not in ("point", "ant"):
        raise ValueError(f"OBJBALL_TYPE is not registered for the {robot}")  # maze_env.py:189-191: only PointEnv and AntEnv define one
    if len(balls) > 1:
        raise NotImplementedError("more than one object ball")
    blocks = world.movable_cells()
    if balls and blocks:
        raise NotImplementedError("object balls together with movable blocks")
    spins = any(cell.can_spin() for _, _, cell in blocks)
    if len({cell.is_half_block() for _, _, cell in blocks if not cell.can_spin()}) > 1 and not (spins or robot == "generic"):
        raise NotImplementedError("mazes that mix half blocks with full-size blocks (the specialised kernels keep one block size per maze)")
    if len(blocks) > (4 if (spins or robot == "generic") else 3):
        raise NotImplementedError("more than 3 movable blocks (4 on the general-engine path)")
    if world.rows > MAX_GRID or world.cols > MAX_GRID:
        raise ValueError("maze grid larger than MZ_MAX_GRID")

    m = MzModel()
    m.abi_version = MZ_ABI_VERSION
    m.robot = ROBOT_ID[robot]
    m.timestep = spec.timestep
    m.frame_skip = spec.frame_skip
    m.integrator_rk4 = 0 if getattr(spec, "integrator", "RK4") == "Euler" else 1
    m.collision_predefined = int(spec.collision_predefined)
    m.gravity[:] = (0.0, 0.0, -9.81)
    m.density, m.viscosity = spec.density, spec.viscosity
    m.max_episode_steps = max_episode_steps
    m.reset_qvel_kind = RESET_KIND[spec.reset_qvel]
    m.engine = 1 if engine == "general" else 0
    m.step_kind = {None: 0, "motors": 1, "point": 2}[step]
    m.nq_robot, m.nv_robot = spec.nq_robot, spec.nv_robot

    # movable blocks change the default geom class (maze_env.py:108-112) and add bodies (maze_env.py:563-660)
    import copy
    import dataclasses

    spec = copy.deepcopy(spec)
    if world.has_blocks:
        stiff = (0.995, 0.995, 0.01, 0.5, 2.0)
        for gs in [spec.floor, spec.wall_geom_defaults] + [g for b in spec.bodies for g in b.geoms]:
            if not gs.explicit_solimp:
                gs.solimp = stiff
    if world.elevated:
        # maze_env.py:102-107: the torso is lifted onto the platforms — for EVERY robot, the planar ones included
        # (`torso.set("pos", f"0 0 {0.75 + height_offset:.2f}")`)
        spec.bodies[0] = dataclasses.replace(spec.bodies[0], pos=(0.0, 0.0, float(f"{0.75 + world.height_offset:.2f}")))
    block_body_index = []
    for (bi_, bj_, cell) in blocks:
        # maze_env.py:563-660 (_add_movable_block).  "Falling" blocks (z-moving: XZ / YZ / XYZ) are shrunk to 99 %, weigh
        # 1 g instead of 0.2 g and have LIMITED slides (x, y: +-scale; z: [-height_offset, 0]).  The body is placed at
        # z = h — NOT h + height_offset: the reference passes height_offset but does not add it — so in an elevated maze
        # a block spawns INSIDE the platform box of its own cell (DESIGN.md section 8 describes what follows from that).
        bx, by = world.cell_center(bi_, bj_)
        falling = cell.can_move_z()
        bh = world.half_z
        # shrink (maze_env.py:575-586): SPIN plates 0.1 (a tenth of the height as well, moved a quarter cell along x), falling
        # blocks 0.99, XY_HALF_BLOCK 0.5, else 1
        if cell.can_spin():
            bh, bx, shrink = bh * 0.1, bx + world.scale * 0.25, 0.1
        else:
            shrink = 0.99 if falling else 0.5 if cell.is_half_block() else 1.0
        half = world.scale * 0.5 * shrink
        geom = dataclasses.replace(spec.wall_geom_defaults, name=f"block_{bi_}_{bj_}", type=R.BOX, size=(half, half, bh),
                                   pos=(0.0, 0.0, 0.0), fromto=None, mass=0.001 if falling else 0.0002, contype=1, conaffinity=1)
        joints = []
        if cell.can_move_x():
            joints.append(R.JointSpec(f"movable_x_{bi_}_{bj_}", R.SLIDE, axis=(1.0, 0.0, 0.0), margin=0.01, limited=falling,
                                      range=(-world.scale, world.scale) if falling else (0.0, 0.0)))
        if cell.can_move_y():
            joints.append(R.JointSpec(f"movable_y_{bi_}_{bj_}", R.SLIDE, axis=(0.0, 1.0, 0.0), margin=0.01, limited=falling,
                                      range=(-world.scale, world.scale) if falling else (0.0, 0.0)))
        if cell.can_move_z():
            joints.append(R.JointSpec(f"movable_z_{bi_}_{bj_}", R.SLIDE, axis=(0.0, 0.0, 1.0), margin=0.01, limited=True,
                                      range=(-world.height_offset, 0.0)))
        if cell.can_spin():  # maze_env.py:649-660: a ball joint behind the two slides (the plate tilts and spins freely)
            joints.append(R.JointSpec(f"spinable_{bi_}_{bj_}", R.BALL, axis=(0.0, 0.0, 1.0)))
        body = R.BodySpec(f"movable_{bi_}_{bj_}", -1, (bx, by, bh), joints=joints, geoms=[geom])
        block_body_index.append(1 + len(spec.bodies))
        spec.bodies.append(body)

    # object balls, hinge flavour (PointEnv.OBJBALL_TYPE; maze_env.py:489-536): body at (x, y, 0) with slide-x, slide-y and a
    # z hinge (joint defaults of the asset), sphere of radius r at height r, mass 1e-4 r^3, own solimp
    ball_body_index = []
    for (bi_, bj_) in balls:
        bx, by = world.cell_center(bi_, bj_)
        r_ = float(task.OBJECT_BALL_SIZE)
        if robot == "ant":
            # free-joint flavour (AntEnv.OBJBALL_TYPE; maze_env.py:539-560): body at (x, y, 0) on a <freejoint> (which takes no
            # defaults: armature = damping = margin = 0), sphere of radius r at height r with the asset's default density
            # (ant.xml: 5.0 => 5 * 4/3 pi r^3), own solimp
            geom = dataclasses.replace(spec.wall_geom_defaults, name=f"objball_{bi_}_{bj_}_geom", type=R.SPHERE, size=(r_,), pos=(0.0, 0.0, r_),
                                       fromto=None, mass=None, contype=1, conaffinity=1,
                                       solimp=(0.9, 0.99, 0.001, 0.5, 2.0), explicit_solimp=True)
            body = R.BodySpec(f"objball_{bi_}_{bj_}", -1, (bx, by, 0.0), joints=[R.JointSpec(f"objball_{bi_}_{bj_}_root", R.FREE)], geoms=[geom])
            ball_body_index.append(1 + len(spec.bodies))
            spec.bodies.append(body)
            continue
        geom = dataclasses.replace(spec.wall_geom_defaults, name=f"objball_{bi_}_{bj_}_geom", type=R.SPHERE, size=(r_,), pos=(0.0, 0.0, r_),
                                   fromto=None, mass=0.0001 * r_ ** 3, contype=1, conaffinity=1,
                                   solimp=(0.9, 0.99, 0.001, 0.5, 2.0), explicit_solimp=True)
        body = R.BodySpec(f"objball_{bi_}_{bj_}", -1, (bx, by, 0.0),
                          joints=[R.JointSpec(f"objball_{bi_}_{bj_}_x", R.SLIDE, axis=(1.0, 0.0, 0.0)),
                                  R.JointSpec(f"objball_{bi_}_{bj_}_y", R.SLIDE, axis=(0.0, 1.0, 0.0)),
                                  R.JointSpec(f"objball_{bi_}_{bj_}_rot", R.HINGE, axis=(0.0, 0.0, 1.0))],
                          geoms=[geom])
        ball_body_index.append(1 + len(spec.bodies))
        spec.bodies.append(body)

    # ---- bodies / joints / dofs / geoms
    nbody = 1 + len(spec.bodies)
    parent = [0] * nbody
    bpos = np.zeros((nbody, 3))
    bquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    bmass = np.zeros(nbody)
    bipos = np.zeros((nbody, 3))
    binertia = np.zeros((nbody, 3, 3))
    biquat = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (nbody, 1))
    bpinertia = np.zeros((nbody, 3))
    jrows, geoms = [], []
    nq = nv = 0
    qpos0: List[float] = []
    m.body_parent[0] = 0
    # geom 0: floor on the world body
    geoms.append((0, spec.floor))
    for bi, b in enumerate(spec.bodies, start=1):
        parent[bi] = b.parent + 1
        bpos[bi] = b.pos
        bquat[bi] = getattr(b, "quat", (1.0, 0.0, 0.0, 0.0))
        m.body_jntadr[bi] = len(jrows) if b.joints else -1
        m.body_jntnum[bi] = len(b.joints)
        m.body_dofadr[bi] = nv if b.joints else -1
        ndof_b = 0
        for j in b.joints:
            jid = len(jrows)
            m.jnt_type[jid] = j.type
            m.jnt_qposadr[jid] = nq
            m.jnt_dofadr[jid] = nv
            m.jnt_bodyid[jid] = bi
            m.jnt_limited[jid] = int(j.limited)
            ax = np.array(j.axis, dtype=np.float64)
            ax = ax / np.linalg.norm(ax)
            m.jnt_axis[jid][:] = tuple(ax)
            m.jnt_pos[jid][:] = j.pos
            rng = j.range
            if j.type == R.HINGE:
                rng = (math.radians(rng[0]), math.radians(rng[1]))
            m.jnt_range[jid][:] = rng
            m.jnt_margin[jid] = j.margin
            m.jnt_solref[jid][:] = j.solref
            m.jnt_solimp[jid][:] = j.solimp
            if getattr(j, "stiffness", 0.0) != 0.0:
                if j.type not in (R.HINGE, R.SLIDE):
                    raise NotImplementedError(f"joint {j.name!r}: springs are implemented on hinge and slide joints")
                m.jnt_stiffness[jid] = j.stiffness
                m.jnt_springref[jid] = math.radians(j.springref) if j.type == R.HINGE else j.springref
            if j.type == R.FREE:
                qpos0 += [b.pos[0], b.pos[1], b.pos[2], *getattr(b, "quat", (1.0, 0.0, 0.0, 0.0))]
                dq, dv = 7, 6
            elif j.type == R.BALL:
                if j.limited:
                    raise NotImplementedError(f"joint {j.name!r}: limited ball joints are not implemented")
                qpos0 += [1.0, 0.0, 0.0, 0.0]
                dq, dv = 4, 3
            else:
                qpos0 += [0.0]
                dq, dv = 1, 1
            for d in range(dv):
                m.dof_bodyid[nv + d] = bi
                m.dof_jntid[nv + d] = jid
                m.dof_armature[nv + d] = j.armature
                m.dof_damping[nv + d] = j.damping
            nq += dq
            nv += dv
            ndof_b += dv
            jrows.append(j)
        m.body_dofnum[bi] = ndof_b
        # inertial properties from geoms
        tot, com = 0.0, np.zeros(3)
        parts = []
        for g in b.geoms:
            gpos, gquat, gsize = resolve_geom(g)
            gm, gI = geom_mass_inertia(g, gsize)
            parts.append((gm, gpos, quat_to_mat(gquat) @ np.diag(gI) @ quat_to_mat(gquat).T))
            tot += gm
            com += gm * gpos
            geoms.append((bi, g))
        com = com / tot if tot > 0 else com
        I = np.zeros((3, 3))
        for gm, gpos, gI in parts:
            d = gpos - com
            I += gI + gm * (d @ d * np.eye(3) - np.outer(d, d))
        bmass[bi], bipos[bi], binertia[bi] = tot, com, I
        biquat[bi], bpinertia[bi] = principal_frame(I, [(resolve_geom(g)[1], geom_mass_inertia(g, resolve_geom(g)[2])[1]) for g in b.geoms])
    # free-joint bodies: MuJoCo keeps body_pos but the pose comes from qpos
    m.nbody, m.njnt, m.nq, m.nv = nbody, len(jrows), nq, nv
    assert nq <= MAX_Q and nv <= MAX_DOF and nbody <= MAX_BODY and len(geoms) <= MAX_GEOM
    for b in range(nbody):
        m.body_parent[b] = parent[b]
        m.body_pos[b][:] = tuple(bpos[b])
        m.body_quat[b][:] = tuple(bquat[b])
        m.body_ipos[b][:] = tuple(bipos[b])
        I = binertia[b]
        m.body_inertia[b][:] = (I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])
        m.body_iquat[b][:] = tuple(biquat[b])
        m.body_pinertia[b][:] = tuple(bpinertia[b])
        m.body_mass[b] = bmass[b]
    for k, v in enumerate(qpos0):
        m.qpos0[k] = v
    m.ngeom = len(geoms)
    for gi, (bi, g) in enumerate(geoms):
        gpos, gquat, gsize = resolve_geom(g)
        m.geom_type[gi] = g.type
        m.geom_bodyid[gi] = bi
        m.geom_contype[gi], m.geom_conaffinity[gi], m.geom_condim[gi] = g.contype, g.conaffinity, g.condim
        m.geom_pos[gi][:] = tuple(gpos)
        m.geom_quat[gi][:] = tuple(gquat)
        m.geom_size[gi][:] = tuple(gsize)
        m.geom_friction[gi][:] = g.friction
        m.geom_solref[gi][:] = g.solref
        m.geom_solimp[gi][:] = g.solimp
        m.geom_margin[gi], m.geom_gap[gi] = g.margin, g.gap
        m.geom_rbound[gi] = geom_rbound(g.type, gsize)
    # actuators
    names = [j.name for j in jrows]
    m.nu = len(spec.actuators)
    for a, act in enumerate(spec.actuators):
        m.act_dofid[a] = m.jnt_dofadr[names.index(act.joint)]
        m.act_gear[a] = act.gear
        m.act_ctrlrange[a][:] = act.ctrlrange
        m.act_ctrllimited[a] = int(act.ctrllimited)
        kind, k = getattr(act, "kind", "motor"), getattr(act, "gain", 1.0)
        m.act_gainprm[a] = 1.0 if kind == "motor" else k
        m.act_biasprm[a][:] = {"motor": (0.0, 0.0, 0.0), "position": (0.0, -k, 0.0), "velocity": (0.0, 0.0, -k)}[kind]
        if kind != "motor" and spec.bodies and m.jnt_type[names.index(act.joint)] not in (R.HINGE, R.SLIDE):
            raise NotImplementedError(f"actuator on {act.joint!r}: servos act on hinge and slide joints")

    # ---- constants at qpos0: M, invweights
    xpos, xquat = _kinematics_qpos0(nbody, parent, bpos, bquat)
    J = np.zeros((nbody, 6, nv))  # rows 0-2 translational (at body COM), 3-5 rotational
    xipos = np.array([xpos[b] + quat_to_mat(xquat[b]) @ bipos[b] for b in range(nbody)])
    for b in range(1, nbody):
        anc = b
        while anc != 0:
            for jid in range(m.body_jntadr[anc], m.body_jntadr[anc] + m.body_jntnum[anc]) if m.body_jntnum[anc] else []:
                Rw = quat_to_mat(xquat[anc])
                d0 = m.jnt_dofadr[jid]
                jt = m.jnt_type[jid]
                anchor = xpos[anc] + Rw @ np.array(m.jnt_pos[jid][:])
                if jt == R.FREE:
                    for k in range(3):
                        J[b, k, d0 + k] = 1.0
                        axis = Rw[:, k]
                        J[b, 3:, d0 + 3 + k] = axis
                        J[b, :3, d0 + 3 + k] = np.cross(axis, xipos[b] - xpos[anc])
                elif jt == R.BALL:
                    for k in range(3):
                        axis = Rw[:, k]
                        J[b, 3:, d0 + k] = axis
                        J[b, :3, d0 + k] = np.cross(axis, xipos[b] - anchor)
                elif jt == R.SLIDE:
                    J[b, :3, d0] = Rw @ np.array(m.jnt_axis[jid][:])
                elif jt == R.HINGE:
                    axis = Rw @ np.array(m.jnt_axis[jid][:])
                    J[b, 3:, d0] = axis
                    J[b, :3, d0] = np.cross(axis, xipos[b] - anchor)
            anc = parent[anc]
    M = np.diag([m.dof_armature[d] for d in range(nv)]).astype(np.float64)
    for b in range(1, nbody):
        Rw = quat_to_mat(xquat[b])
        Iw = Rw @ binertia[b] @ Rw.T
        M += bmass[b] * J[b, :3].T @ J[b, :3] + J[b, 3:].T @ Iw @ J[b, 3:]
    Minv = np.linalg.inv(M)
    m.meaninertia = float(np.mean(np.diag(M)))
    dinv = np.diag(Minv).copy()
    for jid in range(len(jrows)):
        d0 = m.jnt_dofadr[jid]
        if m.jnt_type[jid] == R.FREE:
            dinv[d0:d0 + 3] = dinv[d0:d0 + 3].mean()
            dinv[d0 + 3:d0 + 6] = dinv[d0 + 3:d0 + 6].mean()
        elif m.jnt_type[jid] == R.BALL:
            dinv[d0:d0 + 3] = dinv[d0:d0 + 3].mean()
    for d in range(nv):
        m.dof_invweight0[d] = dinv[d]
    for b in range(1, nbody):
        A = J[b] @ Minv @ J[b].T
        m.body_invweight0[b][:] = (max(np.trace(A[:3, :3]) / 3.0, 1e-15), max(np.trace(A[3:, 3:]) / 3.0, 1e-15))
    # expose for tests
    extra = dict(M0=M, xpos0=xpos, xquat0=xquat, J0=J)

    # ---- maze world
    m.grid_rows, m.grid_cols = world.rows, world.cols
    for i in range(world.rows):
        for j in range(world.cols):
            m.grid[i][j] = CELL_CODE.get(structure[i][j], 0)  # movable cells are bodies, not grid walls
    m.maze_scale, m.torso_x, m.torso_y = world.scale, world.torso_x, world.torso_y
    m.wall_half_xy, m.wall_half_z = world.scale * 0.5, world.half_z
    m.wall_center_z = world.half_z + world.height_offset
    wd = spec.wall_geom_defaults
    m.wall_contype, m.wall_conaffinity, m.wall_condim = 1, 1, wd.condim  # maze_env.py:148-149
    m.wall_friction[:] = wd.friction
    m.wall_solref[:] = wd.solref
    m.wall_solimp[:] = wd.solimp
    m.wall_margin, m.wall_gap = wd.margin, wd.gap

    # ---- Point manual collision
    from mujoco_maze_amd.agent_model import ROBOT_CLASSES  # late import (cycle)
    rcls = ROBOT_CLASSES.get(robot)  # None: a user's generic robot (no manual collision, no velocity clip)
    manual = (rcls.MANUAL_COLLISION if rcls is not None else False) if manual_collision is None else manual_collision
    m.manual_collision = int(bool(manual))
    m.restitution = restitution_coef
    m.velocity_limit = getattr(rcls, "VELOCITY_LIMITS", 0.0) if velocity_limit is None else float(velocity_limit)
    if manual:
        rad = (rcls.RADIUS if rcls is not None else None) if radius is None else radius
        if rad is None:
            raise ValueError("Manual collision needs radius of the model")
        det = CollisionDetector(structure, scale, world.torso_x, world.torso_y, rad)
        segs = det.segments
        if len(segs) > MAX_SEG:
            raise ValueError("too many wall segments")
        m.nseg = len(segs)
        for k, s in enumerate(segs):
            m.seg[k][:] = tuple(s)

    # ---- task
    desc = device_reward_descriptor(task)
    device_rewards = desc is not None
    if len(task.goals) > MAX_GOAL:
        raise ValueError("too many goals")
    m.ngoal = len(task.goals)
    for gi, g in enumerate(task.goals):
        m.goal_dim[gi] = g.dim
        for k in range(g.dim):
            m.goal_pos[gi][k] = float(g.pos[k])
        m.goal_threshold[gi] = g.threshold
        m.goal_reward_scale[gi] = g.reward_scale
    if device_rewards:
        m.reward_kind, m.reward_slot, m.reward_binary, m.term_slot = desc[0], desc[1], int(desc[2]), desc[3]
    m.penalty = task.PENALTY if task.PENALTY is not None else 0.0
    m.task_scale = task.scale
    m.inner_reward_scaling = task.INNER_REWARD_SCALING if inner_reward_scaling is None else inner_reward_scaling
    m.forward_reward_weight, m.ctrl_cost_weight = forward_reward_weight, ctrl_cost_weight
    m.nblock = len(blocks)
    m.observe_blocks = int(bool(task.OBSERVE_BLOCKS))
    for k, bidx in enumerate(block_body_index):
        m.block_bodyid[k] = bidx
        m.block_geomid[k] = next(gi for gi, (bb, _g) in enumerate(geoms) if bb == bidx)
    m.elevated = int(world.elevated)
    m.height_offset = world.height_offset
    m.nball = len(balls)
    m.observe_balls = int(bool(task.OBSERVE_BALLS))
    for k, bidx in enumerate(ball_body_index):
        m.ball_bodyid[k] = bidx
        m.ball_geomid[k] = next(gi for gi, (bb, _g) in enumerate(geoms) if bb == bidx)
    if robot in ("swimmer", "reacher"):
        # swimmer.py:50-69 / reacher.py: _get_obs returns the WHOLE qpos / qvel and reset_model re-randomises all of it —
        # the slide joints of a movable block included (SwimmerPush: obs 18 = 7 + 7 + block xyz + t)
        m.nq_robot, m.nv_robot = m.nq, m.nv
    m.top_down_view = int(bool(task.TOP_DOWN_VIEW))
    m.obs_dim = (m.nq_robot + m.nv_robot + 1 + (3 * len(blocks) if task.OBSERVE_BLOCKS else 0)
                 + (3 * len(balls) if task.OBSERVE_BALLS else 0) + (VIEW_DIM if task.TOP_DOWN_VIEW else 0))
    cm = CompiledModel(m, spec, world, task, device_rewards)
    cm.extra = extra
    return cm

"""Robot plugin surface (reference `mujoco_maze/agent_model.py:12-41`,
`point.py:19-42`, `ant.py:38-54`, `swimmer.py:16-30`).

In the reference these classes *are* the simulators (gym `MujocoEnv` subclasses
driving MuJoCo).  Here the simulation lives in the HIP kernels; the classes keep
the reference's class-level knobs (`FILE`, `MANUAL_COLLISION`, `ORI_IND`,
`RADIUS`, `OBJBALL_TYPE`, `VELOCITY_LIMITS`) so that `MazeEnv(model_cls=...)`
and user subclasses that tweak those knobs keep working, and name the built-in
`RobotSpec` (`robots.py`) the kernels are specialised for.
"""
from typing import Optional


class AgentModel:
    """Subclass it for a robot of your own (reference README.md:127): `ROBOT = "generic"`, `FILE` = path (or text) of its MJCF —
    free / ball / slide / hinge joints (with springs: stiffness / springref), sphere / capsule / box geoms in any orientation (the
    robot's own limbs may collide with each other: MuJoCo's contype / conaffinity and parent-child rules), motors and position / velocity servos, RK4 or MuJoCo's default Euler; what the reader does
    not implement it refuses by name — plus `FRAME_SKIP` and, optionally, `RESET_QVEL` ("normal" | "uniform01" |
    "uniform_sym": the reset noise of ant.py / point.py / swimmer.py) and `STEP`: "motors" (default; the ant's step shape,
    ant.py:61-73: clamped motors, forward reward |dxy| / dt, control cost) or "point" (point.py:44-61: the action turns and moves
    the robot, velocities are clipped to `VELOCITY_LIMITS`, no inner reward; with `MANUAL_COLLISION = True` and `RADIUS` the maze's
    wall bounce of maze_env.py:451-464 follows).  Such a robot is stepped on the device by the general engine
    (csrc/generic_dyn.h) in any maze — movable blocks, object balls, platforms and SPIN plates included; observation
    qpos[:3] | observed balls / blocks | qpos[3:] | qvel | t / 1000."""

    FILE: str
    ROBOT: str  # key into robots.ROBOTS, or "generic"
    MANUAL_COLLISION: bool
    ORI_IND: Optional[int] = None
    RADIUS: Optional[float] = None
    OBJBALL_TYPE: Optional[str] = None
    FRAME_SKIP: int = 1
    STEP: str = "motors"


class PointEnv(AgentModel):
    FILE = "point.xml"
    ROBOT = "point"
    ORI_IND = 2
    MANUAL_COLLISION = True
    RADIUS = 0.4
    OBJBALL_TYPE = "hinge"
    VELOCITY_LIMITS = 10.0
    FRAME_SKIP = 1


class AntEnv(AgentModel):
    FILE = "ant.xml"
    ROBOT = "ant"
    ORI_IND = 3
    MANUAL_COLLISION = False
    OBJBALL_TYPE = "freejoint"
    FRAME_SKIP = 5


class SwimmerEnv(AgentModel):
    FILE = "swimmer.xml"
    ROBOT = "swimmer"
    MANUAL_COLLISION = False
    FRAME_SKIP = 4


class ReacherEnv(AgentModel):
    """reacher.py:15-80 / assets/reacher.xml: the swimmer's first two links with one motor (the asset even
    declares `<mujoco model="swimmer">`), same medium, same step / reward code.  Runs on the swimmer kernels
    instantiated for two links (`csrc/swimmer_dyn.h`)."""

    FILE = "reacher.xml"
    ROBOT = "reacher"
    MANUAL_COLLISION = False
    FRAME_SKIP = 4


ROBOT_CLASSES = {"point": PointEnv, "ant": AntEnv, "swimmer": SwimmerEnv, "reacher": ReacherEnv}

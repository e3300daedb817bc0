"""Robot model descriptions (what the reference keeps as MJCF assets).

The numbers are the model constants of the reference assets
(`mujoco_maze/assets/ant.xml:1-80`, `point.xml:1-33`, `swimmer.xml:1-39`),
re-expressed as plain Python data so that no XML parser or MuJoCo compiler is
needed at run time.  `model.compile_model` turns a `RobotSpec` + a maze world
into the flat constant block the C-ABI consumes.

Angles are degrees here (the assets use `angle="degree"`); compile converts.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

# geom types (MuJoCo ordering matters for pair ordering: plane < sphere < capsule < box)
PLANE, SPHERE, CAPSULE, BOX = 0, 2, 3, 6
FREE, BALL, SLIDE, HINGE = 0, 1, 2, 3


@dataclass
class GeomSpec:
    name: str
    type: int
    size: Tuple[float, ...]
    pos: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    fromto: Optional[Tuple[float, ...]] = None
    density: float = 1000.0
    mass: Optional[float] = None
    contype: int = 1
    conaffinity: int = 1
    condim: int = 3
    friction: Tuple[float, float, float] = (1.0, 0.005, 0.0001)
    solref: Tuple[float, float] = (0.02, 1.0)
    solimp: Tuple[float, float, float, float, float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    margin: float = 0.0
    gap: float = 0.0
    explicit_solimp: bool = False  # set on the geom itself (not inherited from <default>)
    quat: Optional[Tuple[float, float, float, float]] = None  # orientation in the body frame (quat / euler / axisangle / zaxis / xyaxes of the MJCF); fromto wins


@dataclass
class JointSpec:
    name: str
    type: int
    axis: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    pos: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    limited: bool = False
    range: Tuple[float, float] = (0.0, 0.0)  # degrees for hinges
    armature: float = 0.0
    damping: float = 0.0
    margin: float = 0.0
    solref: Tuple[float, float] = (0.02, 1.0)
    solimp: Tuple[float, float, float, float, float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    stiffness: float = 0.0   # hinge / slide spring: passive force -stiffness * (q - springref)
    springref: float = 0.0   # its rest position (degrees for hinges, like `range`)


@dataclass
class BodySpec:
    name: str
    parent: int  # index into RobotSpec.bodies; -1 = world
    pos: Tuple[float, float, float]
    joints: List[JointSpec] = field(default_factory=list)
    geoms: List[GeomSpec] = field(default_factory=list)
    quat: Tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)  # orientation in the parent's frame


@dataclass
class ActuatorSpec:
    joint: str
    gear: float
    ctrlrange: Tuple[float, float]
    ctrllimited: bool = True
    kind: str = "motor"  # "motor" (force = gear * ctrl) | "position" (gain kp: kp (ctrl - gear q)) | "velocity" (gain kv: kv (ctrl - gear qvel))
    gain: float = 1.0     # kp / kv of a servo


@dataclass
class RobotSpec:
    name: str
    bodies: List[BodySpec]
    actuators: List[ActuatorSpec]
    floor: GeomSpec
    wall_geom_defaults: GeomSpec  # <default><geom> of the asset: what maze boxes inherit
    timestep: float
    frame_skip: int
    nq_robot: int
    nv_robot: int
    density: float = 0.0  # medium density (swimmer)
    viscosity: float = 0.0
    collision_predefined: bool = False  # swimmer.xml:3 collision="predefined": no dynamic pairs
    reset_qvel: str = "normal"  # "normal" | "uniform01" | "uniform_sym"
    torso_z: float = 0.0
    integrator: str = "RK4"  # "RK4" (every reference asset) | "Euler" (MuJoCo's default: semi-implicit, implicit in joint damping; general engine)


def _ant() -> RobotSpec:
    def g(name, **kw):  # ant.xml:7-10 defaults
        base = dict(density=5.0, contype=1, conaffinity=0, condim=3, friction=(1.0, 0.5, 0.5),
                    solref=(0.02, 1.0), solimp=(0.8, 0.8, 0.01, 0.5, 2.0), margin=0.01)
        base.update(kw)
        return GeomSpec(name=name, **base)

    def hinge(name, axis, lo, hi):
        return JointSpec(name=name, type=HINGE, axis=axis, limited=True, range=(lo, hi), armature=1.0, damping=1.0)

    bodies = [BodySpec("torso", -1, (0.0, 0.0, 0.75),
                       joints=[JointSpec("root", FREE, margin=0.01)],
                       geoms=[g("torso_geom", type=SPHERE, size=(0.25,))])]
    # leg body, aux body, x sign, y sign, ankle axis, ankle range, geom names — ant.xml:24-66
    legs = [("front_left_leg", "aux_1", +1, +1, (-1.0, 1.0, 0.0), (30.0, 70.0), ("aux_1_geom", "left_leg_geom", "left_ankle_geom")),
            ("front_right_leg", "aux_2", -1, +1, (1.0, 1.0, 0.0), (-70.0, -30.0), ("aux_2_geom", "right_leg_geom", "right_ankle_geom")),
            ("back_leg", "aux_3", -1, -1, (-1.0, 1.0, 0.0), (-70.0, -30.0), ("aux_3_geom", "back_leg_geom", "third_ankle_geom")),
            ("right_back_leg", "aux_4", +1, -1, (1.0, 1.0, 0.0), (30.0, 70.0), ("aux_4_geom", "rightback_leg_geom", "fourth_ankle_geom"))]
    for k, (leg, aux, sx, sy, ax, rng, gn) in enumerate(legs, start=1):
        a, b = 0.2 * sx, 0.2 * sy
        i_leg = len(bodies)
        bodies.append(BodySpec(leg, 0, (0.0, 0.0, 0.0),
                               geoms=[g(gn[0], type=CAPSULE, size=(0.08,), fromto=(0, 0, 0, a, b, 0))]))
        bodies.append(BodySpec(aux, i_leg, (a, b, 0.0),
                               joints=[hinge(f"hip_{k}", (0.0, 0.0, 1.0), -30.0, 30.0)],
                               geoms=[g(gn[1], type=CAPSULE, size=(0.08,), fromto=(0, 0, 0, a, b, 0))]))
        bodies.append(BodySpec(f"ankle_body_{k}", i_leg + 1, (a, b, 0.0),
                               joints=[hinge(f"ankle_{k}", ax, rng[0], rng[1])],
                               geoms=[g(gn[2], type=CAPSULE, size=(0.08,), fromto=(0, 0, 0, 2 * a, 2 * b, 0))]))
    acts = [ActuatorSpec(j, 1.0, (-30.0, 30.0)) for j in
            ("hip_4", "ankle_4", "hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3")]  # ant.xml:71-78
    floor = g("floor", type=PLANE, size=(40.0, 40.0, 40.0), conaffinity=1)
    return RobotSpec("ant", bodies, acts, floor, g("wall", type=BOX, size=(1, 1, 1), conaffinity=1),
                     timestep=0.02, frame_skip=5, nq_robot=15, nv_robot=14, reset_qvel="normal", torso_z=0.75)


def _point() -> RobotSpec:
    def g(name, **kw):  # point.xml:4-7 defaults
        base = dict(density=100.0, contype=1, conaffinity=0, condim=3, friction=(1.0, 0.5, 0.5), margin=0.0)
        base.update(kw)
        return GeomSpec(name=name, **base)

    stiff = (0.9, 0.99, 0.001, 0.5, 2.0)  # point.xml:21-22
    torso = BodySpec("torso", -1, (0.0, 0.0, 0.0),
                     joints=[JointSpec("ballx", SLIDE, axis=(1.0, 0.0, 0.0)),
                             JointSpec("bally", SLIDE, axis=(0.0, 1.0, 0.0)),
                             JointSpec("rot", HINGE, axis=(0.0, 0.0, 1.0))],
                     geoms=[g("pointbody", type=SPHERE, size=(0.5,), pos=(0.0, 0.0, 0.5), solimp=stiff, explicit_solimp=True),
                            g("pointarrow", type=BOX, size=(0.5, 0.1, 0.1), pos=(0.6, 0.0, 0.5), solimp=stiff, explicit_solimp=True)])
    acts = [ActuatorSpec("ballx", 1.0, (-1.0, 1.0)), ActuatorSpec("rot", 1.0, (-0.25, 0.25))]
    floor = g("floor", type=PLANE, size=(40.0, 40.0, 40.0), conaffinity=1)
    return RobotSpec("point", [torso], acts, floor, g("wall", type=BOX, size=(1, 1, 1), conaffinity=1),
                     timestep=0.02, frame_skip=1, nq_robot=3, nv_robot=3, reset_qvel="uniform01")


def _swimmer() -> RobotSpec:
    def g(name, **kw):  # swimmer.xml:5 defaults (density overridden per geom to 1000)
        base = dict(density=1000.0, contype=1, conaffinity=1, condim=1)
        base.update(kw)
        return GeomSpec(name=name, **base)

    def hinge(name, limited, rng=(0.0, 0.0)):
        return JointSpec(name=name, type=HINGE, axis=(0.0, 0.0, 1.0), limited=limited, range=rng, armature=0.1)

    bodies = [
        BodySpec("torso", -1, (0.0, 0.0, 0.0),
                 joints=[JointSpec("slider1", SLIDE, axis=(1.0, 0.0, 0.0), armature=0.1),
                         JointSpec("slider2", SLIDE, axis=(0.0, 1.0, 0.0), armature=0.1),
                         hinge("rot", False)],
                 geoms=[g("frontbody", type=CAPSULE, size=(0.1,), fromto=(1.5, 0, 0, 0.5, 0, 0))]),
        BodySpec("mid", 0, (0.5, 0.0, 0.0), joints=[hinge("rot2", True, (-100.0, 100.0))],
                 geoms=[g("midbody", type=CAPSULE, size=(0.1,), fromto=(0, 0, 0, -1, 0, 0))]),
        BodySpec("back", 1, (-1.0, 0.0, 0.0), joints=[hinge("rot3", True, (-100.0, 100.0))],
                 geoms=[g("backbody", type=CAPSULE, size=(0.1,), fromto=(0, 0, 0, -1, 0, 0))]),
    ]
    acts = [ActuatorSpec("rot2", 150.0, (-1.0, 1.0)), ActuatorSpec("rot3", 150.0, (-1.0, 1.0))]
    floor = g("floor", type=PLANE, size=(40.0, 40.0, 0.1), pos=(0.0, 0.0, -0.1), condim=3)
    return RobotSpec("swimmer", bodies, acts, floor, g("wall", type=BOX, size=(1, 1, 1)),
                     timestep=0.01, frame_skip=4, nq_robot=5, nv_robot=5, density=4000.0, viscosity=0.1,
                     collision_predefined=True, reset_qvel="uniform_sym")


def _reacher() -> RobotSpec:
    """assets/reacher.xml:1-36 (`<mujoco model="swimmer">`): the swimmer's first two links, one motor on `rot2`."""
    sw = _swimmer()
    return RobotSpec("reacher", sw.bodies[:2], sw.actuators[:1], sw.floor, sw.wall_geom_defaults,
                     timestep=0.01, frame_skip=4, nq_robot=4, nv_robot=4, density=4000.0, viscosity=0.1,
                     collision_predefined=True, reset_qvel="uniform_sym")


ROBOTS = {"ant": _ant, "point": _point, "swimmer": _swimmer, "reacher": _reacher}


def robot_spec(name: str) -> RobotSpec:
    return ROBOTS[name]()

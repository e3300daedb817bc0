"""MJCF subset reader / writer for robot assets ("custom AgentModel" XMLs).

The reference keeps its robots as MJCF files next to the package and lets a user point an `AgentModel` subclass at
another file (`agent_model.py:12-17`, `maze_env.py:97-100`).  Here the built-in robots live as `robots.RobotSpec`
data; this module converts between that data and the MJCF subset those assets use, so a user's variant of a
built-in robot — other masses, sizes, gears, limits, contact parameters — loads through the same path:

    env = mm.make("AntUMaze-v0", num_envs=4096, robot_xml="my_ant.xml")

Supported elements: `<compiler angle coordinate inertiafromgeom>`, `<option timestep integrator (RK4; Euler for user robots) density viscosity
collision>`, `<default>` with `<geom>`, `<joint>`, `<motor>` and nested classes (`class=`, `childclass=`); `<worldbody>` with one plane geom (the
floor) and one robot body tree of `<body>`, `<joint type=free|ball|slide|hinge>`, `<freejoint>`,
`<geom type=sphere|capsule|box>` (orientation by fromto, quat, axisangle, euler, zaxis or xyaxes — bodies too);
`<actuator>` with `<motor>`, `<position kp>`, `<velocity kv>`.
Lights, cameras, sites, assets and materials are skipped; **anything else that would change the physics is an error, not a skip**
(joint springs, dry friction, tendons, equality constraints, contact pairs, gear vectors ...).
The device kernels are written for the topologies of the four built-in robots (`csrc/*_dyn.h` check it when the
model is created), so an XML may change parameters, not structure.
"""
import os
import xml.etree.ElementTree as ET
from typing import List, Optional, Sequence

from mujoco_maze_amd import robots as R

_GEOM_TYPE = {"plane": R.PLANE, "sphere": R.SPHERE, "capsule": R.CAPSULE, "box": R.BOX}
_GEOM_NAME = {v: k for k, v in _GEOM_TYPE.items()}
_JOINT_TYPE = {"free": R.FREE, "ball": R.BALL, "slide": R.SLIDE, "hinge": R.HINGE}
_JOINT_NAME = {v: k for k, v in _JOINT_TYPE.items()}


def _floats(text: str) -> tuple:
    return tuple(float(t) for t in text.split())


# Attributes that do not change the physics: skipped.  Every OTHER attribute an element carries must be one the reader implements —
# an MJCF that says `stiffness="8"` or `euler="0 90 0"` and is stepped without it would be a different robot, silently (round 6).
_COSMETIC = {"name", "rgba", "material", "group", "user", "priority"}


def _check_attrs(e: ET.Element, known: set, what: str) -> None:
    extra = sorted(set(e.attrib) - known - _COSMETIC)
    if extra:
        raise ValueError(f"{what}: attribute(s) {', '.join(extra)} are not implemented by this MJCF subset (known: {', '.join(sorted(known))})")


def _orientation(a, radians: bool, eulerseq: str, what: str):
    """quat | axisangle | euler | zaxis | xyaxes of an MJCF body / geom -> unit quaternion (w, x, y, z), or None when the element has none."""
    import math

    import numpy as np

    from mujoco_maze_amd.model import axis_angle_quat, quat_mul, quat_z_to_vec

    given = [k for k in ("quat", "axisangle", "euler", "zaxis", "xyaxes") if k in a]
    if not given:
        return None
    if len(given) > 1:
        raise ValueError(f"{what}: more than one orientation attribute ({', '.join(given)})")
    k, v = given[0], _floats(a[given[0]])
    ang = (lambda x: x) if radians else math.radians
    if k == "quat":
        if len(v) != 4:
            raise ValueError(f"{what}: quat needs four numbers")
        q = np.array(v, dtype=np.float64)
    elif k == "axisangle":
        if len(v) != 4:
            raise ValueError(f"{what}: axisangle needs four numbers")
        ax = np.array(v[:3], dtype=np.float64)
        q = axis_angle_quat(ax / np.linalg.norm(ax), ang(v[3]))
    elif k == "euler":
        if len(v) != 3 or len(eulerseq) != 3 or any(c not in "xyzXYZ" for c in eulerseq):
            raise ValueError(f"{what}: euler needs three numbers and a three-letter eulerseq")
        q = np.array([1.0, 0.0, 0.0, 0.0])
        for c, x in zip(eulerseq, v):  # lower case: rotations about the MOVING axes (post-multiply); upper case: about the fixed ones
            r = axis_angle_quat(np.eye(3)["xyz".index(c.lower())], ang(x))
            q = quat_mul(q, r) if c.islower() else quat_mul(r, q)
    elif k == "zaxis":
        if len(v) != 3:
            raise ValueError(f"{what}: zaxis needs three numbers")
        q = quat_z_to_vec(v)
    else:
        if len(v) != 6:
            raise ValueError(f"{what}: xyaxes needs six numbers")
        x = np.array(v[:3], dtype=np.float64); x /= np.linalg.norm(x)
        y = np.array(v[3:], dtype=np.float64); y -= x * (x @ y); y /= np.linalg.norm(y)
        R3 = np.column_stack([x, y, np.cross(x, y)])
        w = math.sqrt(max(0.0, 1.0 + R3[0, 0] + R3[1, 1] + R3[2, 2])) / 2.0
        if w > 1e-6:
            q = np.array([w, (R3[2, 1] - R3[1, 2]) / (4 * w), (R3[0, 2] - R3[2, 0]) / (4 * w), (R3[1, 0] - R3[0, 1]) / (4 * w)])
        else:  # a half turn: the axis is the eigenvector of R3 for +1
            vals, vecs = np.linalg.eigh((R3 + R3.T) / 2.0)
            ax = vecs[:, int(np.argmax(vals))]
            q = np.array([0.0, *ax])
    q = np.asarray(q, dtype=np.float64)
    return tuple(q / np.linalg.norm(q))


def _fmt(values: Sequence[float]) -> str:
    return " ".join(repr(float(v)) for v in values)


# ---------------------------------------------------------------- writer
_GEOM_FIELDS = ("density", "contype", "conaffinity", "condim", "friction", "solref", "solimp", "margin", "gap")
_JOINT_FIELDS = ("armature", "damping", "margin", "solref", "solimp")


def _geom_attrs(g: R.GeomSpec, default: Optional[R.GeomSpec]) -> dict:
    a = {"name": g.name, "type": _GEOM_NAME[g.type]}
    if g.fromto is not None:
        a["fromto"] = _fmt(g.fromto)
        a["size"] = _fmt(g.size[:1])
    else:
        a["size"] = _fmt(g.size)
        if g.quat is not None:
            a["quat"] = _fmt(g.quat)
        a["pos"] = _fmt(g.pos)
    if g.mass is not None:
        a["mass"] = repr(float(g.mass))
    for f in _GEOM_FIELDS:
        v = getattr(g, f)
        if default is not None and getattr(default, f) == v and not (f == "solimp" and g.explicit_solimp):
            continue
        a[f] = _fmt(v) if isinstance(v, tuple) else repr(v)
    return a


def spec_to_mjcf(spec: R.RobotSpec) -> str:
    """MJCF text of a RobotSpec (lossless for `spec_from_mjcf`)."""
    root = ET.Element("mujoco", model=spec.name)
    ET.SubElement(root, "compiler", angle="degree", coordinate="local", inertiafromgeom="true")
    opt = {"integrator": getattr(spec, "integrator", "RK4"), "timestep": repr(spec.timestep)}
    if spec.density or spec.viscosity:
        opt["density"], opt["viscosity"] = repr(spec.density), repr(spec.viscosity)
    if spec.collision_predefined:
        opt["collision"] = "predefined"
    ET.SubElement(root, "option", **opt)
    default = ET.SubElement(root, "default")
    dg = spec.wall_geom_defaults
    ET.SubElement(default, "geom", **{f: (_fmt(getattr(dg, f)) if isinstance(getattr(dg, f), tuple) else repr(getattr(dg, f)))
                                      for f in _GEOM_FIELDS})
    world = ET.SubElement(root, "worldbody")
    ET.SubElement(world, "geom", **_geom_attrs(spec.floor, dg))
    elems = []
    for b in spec.bodies:
        parent = world if b.parent < 0 else elems[b.parent]
        e = ET.SubElement(parent, "body", name=b.name, pos=_fmt(b.pos), **({"quat": _fmt(b.quat)} if tuple(b.quat) != (1.0, 0.0, 0.0, 0.0) else {}))
        elems.append(e)
        for g in b.geoms:
            ET.SubElement(e, "geom", **_geom_attrs(g, dg))
        for j in b.joints:
            a = {"name": j.name, "type": _JOINT_NAME[j.type]}
            if j.type == R.BALL:
                a.update(pos=_fmt(j.pos), limited="false")
            elif j.type != R.FREE:
                a.update(axis=_fmt(j.axis), pos=_fmt(j.pos), limited="true" if j.limited else "false", range=_fmt(j.range))
            for f in _JOINT_FIELDS:
                v = getattr(j, f)
                # (MJCF's names for a joint's LIMIT parameters are solreflimit / solimplimit: real MuJoCo refuses `solref` on a joint)
                a[{"solref": "solreflimit", "solimp": "solimplimit"}.get(f, f)] = _fmt(v) if isinstance(v, tuple) else repr(v)
            if j.stiffness != 0.0:
                a.update(stiffness=repr(j.stiffness), springref=repr(j.springref))
            ET.SubElement(e, "joint", **a)
    act = ET.SubElement(root, "actuator")
    for m in spec.actuators:
        kind = getattr(m, "kind", "motor")
        ET.SubElement(act, kind, joint=m.joint, gear=repr(m.gear), ctrlrange=_fmt(m.ctrlrange), ctrllimited="true" if m.ctrllimited else "false",
                      **({} if kind == "motor" else {"kp" if kind == "position" else "kv": repr(m.gain)}))
    ET.indent(root)
    return ET.tostring(root, encoding="unicode")


# ---------------------------------------------------------------- reader
def _apply_geom(base: R.GeomSpec, e: ET.Element, name: str, radians: bool = False, eulerseq: str = "xyz") -> R.GeomSpec:
    import dataclasses

    kw = {}
    a = e.attrib
    _check_attrs(e, {"type", "size", "pos", "fromto", "mass", "density", "margin", "gap", "contype", "conaffinity", "condim", "friction", "solref",
                     "solimp", "quat", "axisangle", "euler", "zaxis", "xyaxes", "class"}, f"geom {name!r}")
    q = _orientation(a, radians, eulerseq, f"geom {name!r}")
    if q is not None:
        kw["quat"] = q
    if "type" in a:
        if a["type"] not in _GEOM_TYPE:
            raise ValueError(f"geom {name!r}: type {a['type']!r} is not supported (plane, sphere, capsule, box)")
        kw["type"] = _GEOM_TYPE[a["type"]]
    if "size" in a:
        kw["size"] = _floats(a["size"])
    if "pos" in a:
        kw["pos"] = _floats(a["pos"])
    if "fromto" in a:
        kw["fromto"] = _floats(a["fromto"])
        if len(kw["fromto"]) != 6:
            raise ValueError(f"geom {name!r}: fromto needs six numbers")
    if "mass" in a:
        kw["mass"] = float(a["mass"])
    for f in ("density", "margin", "gap"):
        if f in a:
            kw[f] = float(a[f])
    for f in ("contype", "conaffinity", "condim"):
        if f in a:
            kw[f] = int(a[f])
    if "friction" in a:
        fr = _floats(a["friction"])
        kw["friction"] = tuple(fr) + tuple(base.friction[len(fr):])
    if "solref" in a:
        kw["solref"] = _floats(a["solref"])
    if "solimp" in a:
        si = _floats(a["solimp"])
        kw["solimp"] = tuple(si) + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]
        kw["explicit_solimp"] = True
    return dataclasses.replace(base, name=name, **kw)


def _apply_joint(base: R.JointSpec, e: ET.Element, name: str, radians: bool) -> R.JointSpec:
    import dataclasses
    import math

    a = e.attrib
    kw = {}
    _check_attrs(e, {"type", "axis", "pos", "limited", "range", "armature", "damping", "margin", "solref", "solimp", "solreflimit", "solimplimit",
                     "ref", "frictionloss", "stiffness", "springref", "class"}, f"joint {name!r}")
    for f in ("ref", "frictionloss"):  # parsed so that a ZERO passes; anything else is physics this subset lacks
        if f in a and float(a[f]) != 0.0:
            raise ValueError(f"joint {name!r}: {f}={a[f]!r} is not implemented (reference offsets and dry friction)")
    if "stiffness" in a:
        kw["stiffness"] = float(a["stiffness"])
    if "springref" in a:
        kw["springref"] = float(a["springref"])
    jt = "free" if e.tag == "freejoint" else a.get("type", "hinge")  # MuJoCo's default joint type is hinge
    if jt not in _JOINT_TYPE:
        raise ValueError(f"joint {name!r}: type {jt!r} is not supported (free, ball, slide, hinge)")
    kw["type"] = _JOINT_TYPE[jt]
    if "axis" in a:
        kw["axis"] = _floats(a["axis"])
    if "pos" in a:
        kw["pos"] = _floats(a["pos"])
    if "limited" in a:
        kw["limited"] = a["limited"] == "true"
    if "range" in a:
        rng = _floats(a["range"])
        if radians and kw["type"] == R.HINGE:
            rng = tuple(math.degrees(v) for v in rng)
        kw["range"] = rng
    if "springref" in kw and radians and kw["type"] == R.HINGE:
        kw["springref"] = math.degrees(kw["springref"])
    for f in ("armature", "damping", "margin"):
        if f in a:
            kw[f] = float(a[f])
    for key in ("solref", "solreflimit"):  # (MJCF's name for a joint's limit parameters is solreflimit / solimplimit; the short forms are this repo's own writer's)
        if key in a:
            kw["solref"] = _floats(a[key])
    for key in ("solimp", "solimplimit"):
        if key in a:
            si = _floats(a[key])
            kw["solimp"] = tuple(si) + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]
    return dataclasses.replace(base, name=name, **kw)


def spec_from_mjcf(source: str, like: Optional[R.RobotSpec], frame_skip: int = 1, reset_qvel: str = "normal") -> R.RobotSpec:
    """Parse an MJCF file (path) or text into a RobotSpec.  `like` is the built-in spec of the robot family: it supplies
    what MJCF does not carry (frame_skip, reset distribution, robot coordinate counts — `ant.py`, `point.py`, ...).
    `like = None`: a robot of the user's own (any tree of free / ball / slide / hinge joints with sphere / capsule / box geoms and
    motors); it is stepped by the general engine (csrc/generic_dyn.h), frame_skip / reset distribution as given."""
    text = source if "<" in source else open(source).read()  # XML text or a file path
    root = ET.fromstring(text)
    if root.tag != "mujoco":
        raise ValueError("not an MJCF document (root element must be <mujoco>)")
    for ch in root:  # sections that change the physics and are not implemented must not be skipped silently
        if ch.tag in ("tendon", "equality", "contact", "extension", "deformable"):
            raise ValueError(f"<{ch.tag}> is not implemented by this MJCF subset")
        if ch.tag not in ("compiler", "option", "default", "worldbody", "actuator", "asset", "visual", "size", "statistic", "custom", "sensor", "keyframe"):
            raise ValueError(f"unknown top-level element <{ch.tag}>")
    comp = root.find("compiler")
    radians = comp is not None and comp.get("angle", "degree") == "radian"
    eulerseq = comp.get("eulerseq", "xyz") if comp is not None else "xyz"
    if comp is not None:
        _check_attrs(comp, {"angle", "coordinate", "inertiafromgeom", "eulerseq", "meshdir", "texturedir", "autolimits", "balanceinertia", "boundmass", "boundinertia"}, "<compiler>")
        if comp.get("autolimits", "false") == "true":
            raise ValueError("<compiler autolimits> is not implemented: say limited=\"true\" where a range applies")
    if comp is not None and comp.get("coordinate", "local") != "local":
        raise ValueError("only coordinate=\"local\" is supported")
    if comp is None or comp.get("inertiafromgeom", "auto") not in ("true", "auto"):
        raise ValueError("inertiafromgeom must be true (bodies get their inertia from their geoms)")
    opt = root.find("option")
    oa = opt.attrib if opt is not None else {}
    if opt is not None:
        _check_attrs(opt, {"timestep", "integrator", "density", "viscosity", "collision", "gravity", "iterations", "tolerance", "solver", "cone", "jacobian"}, "<option>")
        if "gravity" in oa and tuple(_floats(oa["gravity"])) != (0.0, 0.0, -9.81):
            raise ValueError("<option gravity>: only MuJoCo's default (0 0 -9.81) is implemented")
        if oa.get("cone", "pyramidal") != "pyramidal" or oa.get("solver", "Newton") != "Newton":
            raise ValueError("<option>: the pyramidal cone and the Newton solver are what is implemented")
    integrator = oa.get("integrator", "Euler")  # (MuJoCo's default)
    if integrator not in ("RK4", "Euler") or (integrator == "Euler" and like is not None):
        raise ValueError("integrator: RK4 (all reference assets) or, for an AgentModel with ROBOT = \"generic\", MuJoCo's default Euler; "
                         "implicit / implicitfast are not implemented")
    dflt = root.find("default")
    geom0 = R.GeomSpec(name="", type=R.SPHERE, size=(0.0,))  # MuJoCo's built-in defaults
    joint0 = R.JointSpec(name="", type=R.HINGE)
    motor_attrs = {}
    # default classes (MJCF: the top-level <default> is class "main"; a nested <default class="x"> inherits everything of the class
    # around it; an element names its class with class="x", a body hands one down to its whole subtree with childclass="x")
    classes = {"main": (geom0, joint0, motor_attrs)}

    def read_defaults(d: ET.Element, cname: str, parent: str):
        g, j, mo = classes[parent]
        for ch in d:
            if ch.tag not in ("default", "geom", "joint", "motor", "site", "camera", "light", "material", "mesh"):
                raise ValueError(f"<default class={cname!r}>: settings for <{ch.tag}> are not implemented")
        if d.find("geom") is not None:
            g = _apply_geom(g, d.find("geom"), "", radians, eulerseq)
            g.explicit_solimp = False
        if d.find("joint") is not None:
            j = _apply_joint(j, d.find("joint"), "", radians)
        if d.find("motor") is not None:
            mo = dict(mo)
            mo.update(d.find("motor").attrib)
        classes[cname] = (g, j, mo)
        for sub in d.findall("default"):
            if not sub.get("class"):
                raise ValueError("a nested <default> needs a class name")
            read_defaults(sub, sub.get("class"), cname)

    if dflt is not None:
        read_defaults(dflt, "main", "main")
    geom0, joint0, motor_attrs = classes["main"]

    def cls_of(e: ET.Element, inherited: str) -> str:
        c = e.get("class", inherited)
        if c not in classes:
            raise ValueError(f"{e.tag} {e.get('name', '')!r}: unknown default class {c!r}")
        return c
    world = root.find("worldbody")
    if world is None:
        raise ValueError("<worldbody> missing")
    floors = [g for g in world.findall("geom") if g.get("type") == "plane"]
    if len(floors) != 1:
        raise ValueError("expected exactly one plane geom (the floor) in <worldbody>")
    floor = _apply_geom(geom0, floors[0], floors[0].get("name", "floor"), radians, eulerseq)
    bodies: List[R.BodySpec] = []

    def walk(e: ET.Element, parent: int, inherited: str = "main"):
        name = e.get("name", f"body{len(bodies)}")
        _check_attrs(e, {"pos", "quat", "axisangle", "euler", "zaxis", "xyaxes", "childclass"}, f"body {name!r}")
        if "childclass" in e.attrib:
            inherited = cls_of(ET.Element("body", {"class": e.get("childclass"), "name": name}), inherited)
        b = R.BodySpec(name, parent, _floats(e.get("pos", "0 0 0")))
        bq = _orientation(e.attrib, radians, eulerseq, f"body {name!r}")
        if bq is not None:
            b.quat = bq
        idx = len(bodies)
        bodies.append(b)
        for k, ch in enumerate(e):
            if ch.tag == "inertial" and (comp is None or comp.get("inertiafromgeom", "auto") != "true"):
                raise ValueError(f"body {name!r}: <inertial> under inertiafromgeom=\"auto\" would override the geoms' inertia; not implemented")
            if ch.tag not in ("geom", "joint", "freejoint", "body", "inertial", "site", "camera", "light"):
                raise ValueError(f"body {name!r}: child element <{ch.tag}> is not implemented")
            if ch.tag == "geom":
                b.geoms.append(_apply_geom(classes[cls_of(ch, inherited)][0], ch, ch.get("name", f"{name}_geom{k}"), radians, eulerseq))
            elif ch.tag == "joint":
                b.joints.append(_apply_joint(classes[cls_of(ch, inherited)][1], ch, ch.get("name", f"{name}_joint{k}"), radians))
            elif ch.tag == "freejoint":  # takes no defaults (MuJoCo: armature = damping = 0, never limited)
                b.joints.append(_apply_joint(R.JointSpec(name="", type=R.FREE), ch, ch.get("name", f"{name}_joint{k}"), radians))
        for ch in e.findall("body"):
            walk(ch, idx, inherited)

    roots = world.findall("body")
    if len(roots) != 1:
        raise ValueError("expected exactly one robot body tree in <worldbody>")
    walk(roots[0], -1)
    acts = []
    for mtr in (root.find("actuator") if root.find("actuator") is not None else []):
        if mtr.tag not in ("motor", "position", "velocity"):
            raise ValueError(f"actuator <{mtr.tag}> is not implemented (motor, position, velocity)")
        # (class defaults are read for motors; a servo carries its settings itself)
        a = dict(classes[cls_of(mtr, "main")][2]) if mtr.tag == "motor" else {}
        a.update(mtr.attrib)
        a.pop("class", None)
        gain_key = {"motor": None, "position": "kp", "velocity": "kv"}[mtr.tag]
        extra = sorted(set(a) - {"joint", "gear", "ctrlrange", "ctrllimited", "forcelimited", "forcerange", gain_key} - _COSMETIC)
        if extra or a.get("forcelimited", "false") == "true" or any(x != 0.0 for x in _floats(a.get("gear", "1"))[1:]):
            raise ValueError(f"{mtr.tag} on {a.get('joint')!r}: only joint, a scalar gear, ctrlrange, ctrllimited{' and ' + gain_key if gain_key else ''} are implemented "
                             f"({', '.join(extra) or 'force limits / gear vector'})")
        gear = _floats(a.get("gear", "1"))[0]
        acts.append(R.ActuatorSpec(a["joint"], gear, _floats(a.get("ctrlrange", "0 0")), a.get("ctrllimited", "false") == "true",
                                   kind=mtr.tag, gain=float(a.get(gain_key, "1")) if gain_key else 1.0))
    nq = sum({R.FREE: 7, R.BALL: 4}.get(j.type, 1) for b in bodies for j in b.joints)
    nv = sum({R.FREE: 6, R.BALL: 3}.get(j.type, 1) for b in bodies for j in b.joints)
    # The swimmer family's kernels are written for planar chains of 2..6 links (csrc/swimmer_dyn.h is generic in the link count;
    # Swimmer = 3, Reacher = 2): a user's chain may be longer or shorter than the built-in asset's.  Every other family keeps
    # its structure (parameters may change, topology not).
    if like is None:
        import dataclasses

        wall_defaults = dataclasses.replace(geom0, name="wall", type=R.BOX, size=(1.0, 1.0, 1.0), contype=1, conaffinity=1)
        return R.RobotSpec("generic", bodies, acts, floor, wall_defaults, timestep=float(oa.get("timestep", 0.002)), frame_skip=int(frame_skip),
                           nq_robot=nq, nv_robot=nv, density=float(oa.get("density", 0.0)), viscosity=float(oa.get("viscosity", 0.0)),
                           collision_predefined=oa.get("collision", "all") == "predefined", reset_qvel=reset_qvel, torso_z=bodies[0].pos[2],
                           integrator=integrator)
    if any(a.kind != "motor" for a in acts) or any(tuple(b.quat) != (1.0, 0.0, 0.0, 0.0) or any(g.quat is not None for g in b.geoms) or any(j.stiffness != 0.0 for j in b.joints) for b in bodies):
        raise ValueError(f"{like.name}: a variant of a built-in robot may change parameters, not structure — turned bodies / geoms and joint springs "
                         "and servo actuators need an AgentModel with ROBOT = \"generic\" (the general engine)")
    chain = (like.name in ("swimmer", "reacher") and nq == nv == len(bodies) + 2 and 2 <= len(bodies) <= 6 and len(acts) == len(bodies) - 1
             and all(b.parent == i - 1 for i, b in enumerate(bodies)))
    if not chain and ((nq, nv) != (like.nq_robot, like.nv_robot) or len(acts) != len(like.actuators)):
        raise ValueError(f"{like.name}: the XML has {nq} / {nv} coordinates and {len(acts)} motors; the {like.name} kernels are "
                         f"written for {like.nq_robot} / {like.nv_robot} and {len(like.actuators)} (parameters may change, structure not)")
    import dataclasses

    wall_defaults = dataclasses.replace(geom0, name="wall", type=R.BOX, size=(1.0, 1.0, 1.0), contype=1, conaffinity=1)  # maze_env.py:134-135
    return R.RobotSpec(like.name, bodies, acts, floor, wall_defaults, timestep=float(oa.get("timestep", 0.002)), frame_skip=like.frame_skip,
                       nq_robot=nq, nv_robot=nv, density=float(oa.get("density", 0.0)), viscosity=float(oa.get("viscosity", 0.0)),
                       collision_predefined=oa.get("collision", "all") == "predefined", reset_qvel=like.reset_qvel,
                       torso_z=bodies[0].pos[2])


def world_to_mjcf(cm, name: str = "") -> str:
    """MJCF of a compiled model as this repository steps it: the robot spec (with the movable bodies compile_model appended),
    the maze boxes and platforms, the goal sites — in the element order of the reference's generator (maze_env.py:97-218).
    Used by tools/export_mjcf.py, tests/test_mujoco_crosscheck.py and render.state_for_viewer."""
    root = ET.fromstring(spec_to_mjcf(cm.spec))
    root.set("model", f"{cm.spec.name}:{name}" if name else cm.spec.name)
    wb = root.find("worldbody")
    fmt = lambda v: " ".join(repr(float(x)) for x in v)  # noqa: E731
    w = cm.world
    # maze boxes: children of the world body, contype = conaffinity = 1 (maze_env.py:134-135,149-150), everything else
    # from the asset's <default><geom>
    for i in range(w.rows):
        for j in range(w.cols):
            cell = w.structure[i][j]
            x, y = w.cell_center(i, j)
            if w.elevated and not cell.is_chasm():  # platform under every cell that is not a chasm (maze_env.py:124-137)
                ET.SubElement(wb, "geom", name=f"elevated_{i}_{j}", type="box", pos=fmt((x, y, w.half_z)),
                              size=fmt((w.scale * 0.5, w.scale * 0.5, w.half_z)), contype="1", conaffinity="1")
            if cell.is_block():
                ET.SubElement(wb, "geom", name=f"block_{i}_{j}", type="box", pos=fmt((x, y, w.half_z + w.height_offset)),
                              size=fmt((w.scale * 0.5, w.scale * 0.5, w.half_z)), contype="1", conaffinity="1")
    for gi, g in enumerate(cm.task.goals):
        z = float(g.pos[2]) if g.dim >= 3 else 0.0
        size = w.scale * 0.1 if g.custom_size is None else g.custom_size
        ET.SubElement(wb, "site", name=f"goal_site{gi}", pos=fmt((g.pos[0], g.pos[1], z)), size=repr(float(size)))
    ET.indent(root)
    return ET.tostring(root, encoding="unicode")

#!/usr/bin/env python3
"""Round-2 fixtures, generated from the reference in the build container (needs /root/reference; never runs on the GPU box).
Kept apart from make_golden.py so that the round-1 fixtures (whose values depend on that script's RNG call order) stay
byte-identical.  Outputs are data only — values the reference's assets hold or its own code computes:

  robots.json      the robot assets (mujoco_maze/assets/{ant,point,swimmer,reacher}.xml) as resolved model constants:
                   compiler / option attributes, bodies in depth-first order with their joints and geoms AFTER applying
                   the asset's <default> classes the way the MuJoCo compiler does, actuators.  Pins mujoco_maze_amd/robots.py
                   (SURVEY §8a row A16).
  obs_layout.json  for every robot x a set of mazes: which state entry lands in which observation slot
                   (the reference's own `_get_obs` of AntEnv / PointEnv / SwimmerEnv / ReacherEnv + MazeEnv._get_obs run on
                   sentinel-valued qpos / qvel / body positions), and which qpos / qvel entries `reset_model` re-randomises
                   and with which distribution (the reference's reset_model run with a marker RNG).
  views.json       MazeEnv.get_top_down_view (maze_env.py:262-349) and the observation that carries it (`TOP_DOWN_VIEW` tasks,
                   maze_env.py:351-369) for custom tasks derived from the reference's UMaze / Push / Fall / 4Rooms tasks: robot
                   and block positions in, the reference's 5 x 5 x 3 view and full observation out.
  custom_worlds.json  the world the reference generates for a custom maze with XY_HALF_BLOCK cells (maze_env.py:583-584).
"""
import json
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the gym / mujoco stubs and imports the reference)

from mujoco_maze import ant, maze_env, point, reacher, swimmer  # noqa: E402

ASSETS = os.path.join(G.REF, "mujoco_maze", "assets")


# ---------------------------------------------------------------- robots.json
def _floats(s):
    return [float(v) for v in s.split()]


GEOM_KEYS = ("type", "size", "pos", "fromto", "density", "mass", "contype", "conaffinity", "condim", "friction", "solref", "solimp", "margin")
JOINT_KEYS = ("type", "axis", "pos", "limited", "range", "armature", "damping", "margin")
NUMERIC = {"size", "pos", "fromto", "axis", "range", "friction", "solref", "solimp"}
SCALAR = {"density", "mass", "armature", "damping", "margin", "gear"}
INTS = {"contype", "conaffinity", "condim"}


def _norm(key, val):
    if val is None:
        return None
    if key in NUMERIC or key == "ctrlrange":
        return _floats(val)
    if key in SCALAR:
        return float(val)
    if key in INTS:
        return int(val)
    if key in ("limited", "ctrllimited"):
        return val == "true"
    return val


def dump_robot(xml_name):
    root = ET.parse(os.path.join(ASSETS, xml_name)).getroot()
    dflt = root.find("default")
    d_geom = dict(dflt.find("geom").attrib) if dflt is not None and dflt.find("geom") is not None else {}
    d_joint = dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {}
    d_motor = dict(dflt.find("motor").attrib) if dflt is not None and dflt.find("motor") is not None else {}

    def geom(e):
        a = dict(d_geom)
        a.update(e.attrib)
        return dict(name=e.get("name"), **{k: _norm(k, a.get(k)) for k in GEOM_KEYS})

    def joint(e):
        a = dict(d_joint)
        a.update(e.attrib)
        if e.tag == "freejoint":
            a["type"] = "free"
        return dict(name=e.get("name"), **{k: _norm(k, a.get(k)) for k in JOINT_KEYS})

    bodies = []

    def walk(b, parent):
        idx = len(bodies)
        bodies.append(dict(name=b.get("name"), parent=parent, pos=_floats(b.get("pos", "0 0 0")),
                           joints=[joint(j) for j in b if j.tag in ("joint", "freejoint")],
                           geoms=[geom(g) for g in b.findall("geom")]))
        for c in b.findall("body"):
            walk(c, idx)

    wb = root.find("worldbody")
    for b in wb.findall("body"):
        walk(b, -1)
    acts = []
    for m in root.find("actuator"):
        a = dict(d_motor)
        a.update(m.attrib)
        acts.append(dict(kind=m.tag, joint=a.get("joint"), gear=_norm("gear", a.get("gear")), ctrlrange=_norm("ctrlrange", a.get("ctrlrange")),
                         ctrllimited=_norm("ctrllimited", a.get("ctrllimited"))))
    return dict(model=root.get("model"), compiler=dict(root.find("compiler").attrib), option=dict(root.find("option").attrib),
                default_geom={k: _norm(k, v) for k, v in d_geom.items() if k in GEOM_KEYS},
                default_joint={k: _norm(k, v) for k, v in d_joint.items() if k in JOINT_KEYS},
                world_geoms=[geom(g) for g in wb.findall("geom")], bodies=bodies, actuators=acts)


# ---------------------------------------------------------------- obs_layout.json
QBASE, VBASE, BBASE = 1000.0, 2000.0, 3000.0


class _MarkerRng:
    """np_random stand-in: every draw returns a recognisable constant, so that `reset_model`'s arithmetic shows which
    entries receive which kind of noise."""

    def uniform(self, low=0.0, high=1.0, size=None):
        return np.full(size, 0.25 * (high - low))  # marks "uniform(low, high)": 0.05 for (-0.1, 0.1)

    def randn(self, *shape):
        return np.full(shape, 7.0)

    def standard_normal(self, size=None):
        return np.full(size, 7.0)

    def random(self, size=None):
        return np.full(size, 3.0)


def _make_fake(real_cls):
    class Fake(real_cls):
        def __init__(self, file_path=None, **kw):  # bypass gym's MujocoEnv: only what _get_obs / reset_model touch
            self.file_path = file_path
            tree = ET.parse(file_path)
            self.tree = tree
            nq = nv = 0
            for j in tree.getroot().iter():
                if j.tag == "freejoint" or (j.tag == "joint" and j.get("type") == "free"):
                    nq, nv = nq + 7, nv + 6
                elif j.tag == "joint" and j.get("name") is not None and j.get("type") != "free":
                    if j.get("type") == "ball":
                        nq, nv = nq + 4, nv + 3
                    else:
                        nq, nv = nq + 1, nv + 1
            self.model = types.SimpleNamespace(nq=nq, nv=nv)
            d = types.SimpleNamespace(qpos=QBASE + np.arange(nq, dtype=np.float64), qvel=VBASE + np.arange(nv, dtype=np.float64))
            self.data = d
            self.sim = types.SimpleNamespace(data=d)
            self.init_qpos = np.zeros(nq)
            self.init_qvel = np.zeros(nv)
            self.np_random = _MarkerRng()
            no = len(self._get_obs())  # the reference's own robot _get_obs
            self.observation_space = types.SimpleNamespace(shape=(no,), high=np.full(no, np.inf), low=np.full(no, -np.inf))
            self.body_names = [b.get("name") for b in tree.findall(".//worldbody/body") if b.get("name") != "torso"]
            self.state_set = None

        def set_state(self, qpos, qvel):
            self.state_set = (np.array(qpos, dtype=np.float64), np.array(qvel, dtype=np.float64))

        def get_body_com(self, name):
            k = self.body_names.index(name)
            return BBASE + 10.0 * k + np.arange(3, dtype=np.float64)

    Fake.__name__ = real_cls.__name__
    return Fake


ROBOT_CLASSES = {"AntEnv": ant.AntEnv, "PointEnv": point.PointEnv, "SwimmerEnv": swimmer.SwimmerEnv, "ReacherEnv": reacher.ReacherEnv}


def _decode(v, env):
    if v >= BBASE:
        k, c = divmod(int(round(v - BBASE)), 10)
        return f"body:{env.wrapped_env.body_names[k]}:{'xyz'[c]}"
    if v >= VBASE:
        return f"v{int(round(v - VBASE))}"
    if v >= QBASE:
        return f"q{int(round(v - QBASE))}"
    return f"t*{v / env.t:g}" if env.t else "t"


def dump_layout(env_id):
    kw = next(r for r in G.REGISTERED if r["id"] == env_id)["kwargs"]
    fake = _make_fake(ROBOT_CLASSES[kw["model_cls"].__name__])
    env = maze_env.MazeEnv(**dict(kw, model_cls=fake))
    env.t = 8
    obs = env._get_obs()
    layout = [_decode(float(v), env) for v in obs]
    w = env.wrapped_env
    w.reset_model()
    qpos, qvel = w.state_set
    kinds = {0.05: "uniform(-0.1,0.1)", 0.7: "randn*0.1", 0.3: "random*0.1", 0.0: "none"}

    def kind(v):
        for k, name in kinds.items():
            if abs(v - k) < 1e-12:
                return name
        raise ValueError(v)

    return dict(robot=kw["model_cls"].__name__, nq=w.model.nq, nv=w.model.nv, obs_dim=len(layout), layout=layout,
                reset_qpos=[kind(float(v)) for v in qpos], reset_qvel=[kind(float(v)) for v in qvel],
                bodies=list(w.body_names))


LAYOUT_IDS = ["AntUMaze-v0", "AntPush-v0", "AntMultiPush-v0", "AntFall-v0", "AntSmallBilliard-v0", "PointUMaze-v0", "PointPush-v0",
              "PointPushMaze-v0", "PointBilliard-v0", "PointFall-v0", "SwimmerUMaze-v0", "SwimmerPush-v0", "SwimmerFall-v0",
              "ReacherUMaze-v0", "ReacherPush-v0"]


# ---------------------------------------------------------------- views.json / custom_worlds.json
from mujoco_maze import maze_task  # noqa: E402


class _ViewRobot(G._FakeAnt):
    """Kinematic stand-in whose torso and movable bodies sit wherever the generator puts them."""

    def get_body_com(self, name):
        if name == "torso":
            return np.array([self.xy[0], self.xy[1], 0.75])
        return self.bodies[name].copy()


def _view_task(base, name):
    return type(name, (base,), dict(TOP_DOWN_VIEW=True))


VIEW_CASES = [("ViewUMaze", maze_task.GoalRewardUMaze, 8.0), ("ViewPush", maze_task.GoalRewardPush, 8.0),
              ("ViewFall", maze_task.GoalRewardFall, 8.0), ("View4Rooms", maze_task.GoalReward4Rooms, 4.0),
              ("ViewMultiPush", maze_task.GoalRewardMultiPush, 4.0)]


def dump_views(rng):
    out = {}
    for name, base, scale in VIEW_CASES:
        env = maze_env.MazeEnv(model_cls=_ViewRobot, maze_task=_view_task(base, name), maze_size_scaling=scale)
        w = env.wrapped_env
        rows, cols = len(env._maze_structure), len(env._maze_structure[0])
        lo = np.array([-0.5 * scale - env._init_torso_x, -0.5 * scale - env._init_torso_y])
        hi = np.array([(cols - 0.5) * scale - env._init_torso_x, (rows - 0.5) * scale - env._init_torso_y])
        cases = []
        for k in range(36):
            # inside the maze mostly; every 6th sample up to two cells outside (negative / beyond-the-grid rows and columns)
            pad = 2.0 * scale if k % 6 == 5 else 0.0
            w.xy = rng.uniform(lo - pad, hi + pad)
            if k == 0:
                w.xy = np.zeros(2)  # the spawn position: everything sits on cell centres
            blocks = []
            for bname in env.movable_blocks:
                p = w.bodies[bname]
                p[:2] = rng.uniform(lo, hi) if k else p[:2]
                blocks.append([float(p[0]), float(p[1])])
            env.t = k
            view = env.get_top_down_view().copy()
            obs = env._get_obs()
            cases.append(dict(robot_xy=[float(w.xy[0]), float(w.xy[1])], block_xy=blocks, view=[float(v) for v in view.flat],
                              obs=[float(v) for v in obs]))
        out[name] = dict(base=base.__name__, scale=scale, grid=G.grid_text(env._maze_structure), obs_dim=len(cases[0]["obs"]),
                         movable=list(env.movable_blocks), cases=cases)
    return out


class _HalfBlockMaze(maze_task.GoalRewardPush):
    @staticmethod
    def create_maze():
        E, B, R, H, M = (maze_task.MazeCell.EMPTY, maze_task.MazeCell.BLOCK, maze_task.MazeCell.ROBOT, maze_task.MazeCell.XY_HALF_BLOCK,
                         maze_task.MazeCell.XY_BLOCK)
        return [[B, B, B, B, B],
                [B, E, H, E, B],
                [B, R, E, H, B],
                [B, E, E, E, B],
                [B, B, B, B, B]]


def dump_custom_worlds():
    out = {}
    for fake, tag in ((G._FakeAnt, "ant"), (G._FakePoint, "point")):
        env = maze_env.MazeEnv(model_cls=fake, maze_task=_HalfBlockMaze, maze_size_scaling=4.0)
        real_build = G.build_env
        G.build_env = lambda _id, env=env: env
        try:
            out[f"HalfBlockMaze/{tag}"] = dict(G.dump_world("custom"), grid=G.grid_text(env._maze_structure), scale=4.0)
        finally:
            G.build_env = real_build
    # the Ant's object ball hangs on a <freejoint> (AntEnv.OBJBALL_TYPE, maze_env.py:539-560): world of a registered id
    out["AntSmallBilliard-v0"] = dict(G.dump_world("AntSmallBilliard-v0"), scale=2.0)
    return out


def main():
    robots = {n: dump_robot(f"{n}.xml") for n in ("ant", "point", "swimmer", "reacher")}
    with open(os.path.join(HERE, "robots.json"), "w") as f:
        json.dump(robots, f, indent=0, sort_keys=True)
    layouts = {e: dump_layout(e) for e in LAYOUT_IDS}
    with open(os.path.join(HERE, "obs_layout.json"), "w") as f:
        json.dump(layouts, f, indent=0, sort_keys=True)
    for e, l in layouts.items():
        print(e, l["obs_dim"], l["layout"][:8], "...")
    views = dump_views(np.random.default_rng(20260929))
    with open(os.path.join(HERE, "views.json"), "w") as f:
        json.dump(views, f, indent=0, sort_keys=True)
    for n, v in views.items():
        print(n, v["obs_dim"], v["grid"], "nonzero view entries of case 0:", sum(1 for x in v["cases"][0]["view"] if x))
    worlds = dump_custom_worlds()
    with open(os.path.join(HERE, "custom_worlds.json"), "w") as f:
        json.dump(worlds, f, indent=0, sort_keys=True)
    for n, v in worlds.items():
        print(n, [(b["name"], b["geom"]["size"]) for b in v["movable"]])


if __name__ == "__main__":
    main()

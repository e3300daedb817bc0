#!/usr/bin/env python3
"""Generate golden fixtures from the *reference's own pure-Python modules*.

Run in the build container only (needs /root/reference; the GPU box never runs
this).  Nothing of the reference's source text is stored: the outputs are data —
inputs and the values the reference computes on them:

  registry.json      every registered env id -> task class, grid, scale, goals,
                     flags, resolved reward/termination implementation
  worlds.json        the MJCF world the reference generates (maze_env.py:97-218)
                     for the BASELINE configs: geoms, movable bodies, joints,
                     default solimp, goal sites, obs dim, xy limits
  segments.json      CollisionDetector wall-segment tables (maze_env_utils.py:151-184)
  detect.npz         seeded moves -> detect() hit / point / reflection, and the
                     full MazeEnv.step bounce / give-up result (maze_env.py:457-464)
  reward.npz         seeded observations -> reward / termination / first-match index
  line_kat.json      the known-answer cases of the reference's tests/test_intersect.py
                     evaluated through the reference's Line class

gym / mujoco are absent here; they are replaced by in-memory stub modules that
only provide the names the reference imports (never written to disk).
"""
import json
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Box:
        def __init__(self, low, high, *a, **k):
            self.low, self.high = np.asarray(low), np.asarray(high)
            self.shape = self.low.shape

    class MujocoEnv:
        def __init__(self, *a, **k):
            pass

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    registered = []
    spaces = mod("gym.spaces", Box=Box, Space=object)
    envs = mod("gym.envs", register=lambda **kw: registered.append(kw))
    mod("gym.envs.mujoco")
    mod("gym.envs.mujoco.mujoco_env", MujocoEnv=MujocoEnv)
    mod("gym.utils", EzPickle=EzPickle)
    mod("gym.core", ObsType=object)
    mod("gym", Env=object, spaces=spaces, envs=envs)
    mod("mujoco")
    return registered


REGISTERED = _install_stubs()
sys.path.insert(0, REF)
import mujoco_maze  # noqa: E402  (runs the reference's registration loop)
from mujoco_maze import maze_env, maze_env_utils, maze_task  # noqa: E402

CODE = {
    maze_env_utils.MazeCell.ROBOT: "R", maze_env_utils.MazeCell.EMPTY: "E", maze_env_utils.MazeCell.BLOCK: "B",
    maze_env_utils.MazeCell.CHASM: "C", maze_env_utils.MazeCell.OBJECT_BALL: "O", maze_env_utils.MazeCell.XY_BLOCK: "M",
    maze_env_utils.MazeCell.XZ_BLOCK: "x", maze_env_utils.MazeCell.YZ_BLOCK: "y", maze_env_utils.MazeCell.XYZ_BLOCK: "Z",
    maze_env_utils.MazeCell.XY_HALF_BLOCK: "H", maze_env_utils.MazeCell.SPIN: "S",
}


def grid_text(structure):
    return "/".join("".join(CODE[c] for c in row) for row in structure)


# ---------------------------------------------------------------- registry
def dump_registry():
    out = {}
    for kw in REGISTERED:
        k = kw["kwargs"]
        cls = k["maze_task"]
        task = cls(k["maze_size_scaling"])
        out[kw["id"]] = dict(
            robot=k["model_cls"].__name__,
            task=cls.__name__,
            scale=k["maze_size_scaling"],
            inner_reward_scaling=k["inner_reward_scaling"],
            max_episode_steps=kw["max_episode_steps"],
            reward_threshold=kw["reward_threshold"],
            grid=grid_text(cls.create_maze()),
            goals=[dict(pos=g.pos.tolist(), reward_scale=g.reward_scale, threshold=g.threshold,
                        custom_size=g.custom_size) for g in task.goals],
            penalty=cls.PENALTY,
            observe_blocks=cls.OBSERVE_BLOCKS,
            observe_balls=cls.OBSERVE_BALLS,
            object_ball_size=cls.OBJECT_BALL_SIZE,
            reward_impl=cls.reward.__qualname__,
            termination_impl=cls.termination.__qualname__,
        )
    return out


# ---------------------------------------------------------------- worlds
class _FakeRobot:
    """Kinematic stand-in for AgentModel: records the MJCF the reference hands
    to MuJoCo and teleports to the commanded xy on step()."""

    MANUAL_COLLISION = False
    RADIUS = None
    OBJBALL_TYPE = "freejoint"
    NQ = 15
    NV = 14

    def __init__(self, file_path, **kw):
        self.file_path = file_path
        self.tree = ET.parse(file_path)
        self.xy = np.zeros(2)
        self.observation_space = types.SimpleNamespace(
            shape=(self.NQ + self.NV,), high=np.full(self.NQ + self.NV, np.inf), low=np.full(self.NQ + self.NV, -np.inf))
        self.bodies = {b.get("name"): np.array([float(v) for v in b.get("pos").split()])
                       for b in self.tree.findall(".//worldbody/body") if b.get("name") != "torso"}

    def _get_obs(self):
        o = np.zeros(self.NQ + self.NV)
        o[:2] = self.xy
        return o

    def get_body_com(self, name):
        return self.bodies[name].copy()

    def get_xy(self):
        return self.xy.copy()

    def set_xy(self, xy):
        self.xy = np.array(xy, dtype=np.float64)

    def step(self, action):
        self.xy = np.array(action[:2], dtype=np.float64)
        return self._get_obs(), 0.0, False, {}

    def reset(self):
        self.xy = np.zeros(2)


class _FakeAnt(_FakeRobot):
    FILE = "ant.xml"


class _FakePoint(_FakeRobot):
    FILE = "point.xml"
    MANUAL_COLLISION = True
    RADIUS = 0.4
    OBJBALL_TYPE = "hinge"
    NQ = 3
    NV = 3


def build_env(env_id):
    kw = next(r for r in REGISTERED if r["id"] == env_id)["kwargs"]
    fake = _FakePoint if kw["model_cls"].__name__ == "PointEnv" else _FakeAnt
    kw = dict(kw, model_cls=fake)
    return maze_env.MazeEnv(**kw)


def dump_world(env_id):
    env = build_env(env_id)
    root = env.wrapped_env.tree.getroot()
    wb = root.find(".//worldbody")
    geoms = []
    for g in wb.findall("./geom"):
        if g.get("type") == "box":
            geoms.append(dict(name=g.get("name"), pos=[float(v) for v in g.get("pos").split()],
                              size=[float(v) for v in g.get("size").split()],
                              contype=int(g.get("contype")), conaffinity=int(g.get("conaffinity"))))
    bodies = []
    for b in wb.findall("./body"):
        if b.get("name") == "torso":
            continue
        bg = b.find("geom")
        bodies.append(dict(
            name=b.get("name"), pos=[float(v) for v in b.get("pos").split()],
            geom=dict(type=bg.get("type"), size=[float(v) for v in bg.get("size").split()],
                      mass=float(bg.get("mass")) if bg.get("mass") else None, pos=[float(v) for v in bg.get("pos").split()]),
            joints=[dict(name=j.get("name"), type=j.get("type", j.tag), axis=[float(v) for v in j.get("axis", "0 0 0").split()],
                         limited=j.get("limited"), range=j.get("range"), margin=j.get("margin"),
                         armature=j.get("armature"), damping=j.get("damping")) for j in list(b.findall("joint")) + list(b.findall("freejoint"))]))
    sites = [dict(name=s.get("name"), pos=[float(v) for v in s.get("pos").split()], size=float(s.get("size")))
             for s in wb.findall("./site")]
    torso = root.find(".//body[@name='torso']")
    return dict(
        boxes=geoms, movable=bodies, sites=sites,
        default_geom_solimp=root.find(".//default").find(".//geom").get("solimp"),
        torso_pos=[float(v) for v in torso.get("pos").split()],
        obs_dim=int(env._get_obs().shape[0]),
        xy_limits=list(env._xy_limits()),
        init_torso=[env._init_torso_x, env._init_torso_y],
        elevated=bool(env.elevated), blocks=bool(env.blocks),
        movable_names=list(env.movable_blocks), ball_names=list(env.object_balls),
    )


# ---------------------------------------------------------------- segments / detect / bounce
POINT_MAZES = ["PointUMaze-v0", "Point4Rooms-v0", "PointPush-v0", "PointCorridor-v0", "PointTRoom-v0", "PointBilliard-v0"]


def dump_segments_and_moves(rng):
    seg, arrays = {}, {}
    for env_id in POINT_MAZES:
        env = build_env(env_id)
        det = env._collision
        seg[env_id] = dict(
            robot=[[ln.p1.real, ln.p1.imag, ln.p2.real, ln.p2.imag] for ln in det.lines],
            ball=[[ln.p1.real, ln.p1.imag, ln.p2.real, ln.p2.imag] for ln in env._objball_collision.lines],
            ball_radius=env._task.OBJECT_BALL_SIZE)
        xmin, xmax, ymin, ymax = env._xy_limits()
        n = 3000
        old = np.stack([rng.uniform(xmin, xmax, n), rng.uniform(ymin, ymax, n)], 1)
        step = rng.normal(0.0, 1.0, (n, 2)) * rng.choice([0.05, 0.5, 2.0], (n, 1))
        new = old + step
        # a few exactly-zero and sub-threshold moves (maze_env_utils.py:189)
        new[:5] = old[:5]
        new[5:10] = old[5:10] + 1e-9
        # moves that end exactly on a wall line (touching counts: `<= 0`)
        first = det.lines[0]
        new[10] = [first.p1.real * 0.5 + first.p2.real * 0.5, first.p1.imag * 0.5 + first.p2.imag * 0.5]
        hit = np.zeros(n, np.uint8)
        point = np.zeros((n, 2))
        refl = np.zeros((n, 2))
        final = np.zeros((n, 2))
        gave_up = np.zeros(n, np.uint8)
        valid = np.ones(n, np.uint8)
        for k in range(n):
            try:
                col = det.detect(old[k], new[k])
                env.wrapped_env.set_xy(old[k])
                obs, _, _, info = env.step(np.array([new[k, 0], new[k, 1]]))
            except ZeroDivisionError:  # collinear move (latent reference bug)
                valid[k] = 0
                continue
            final[k] = info["position"]
            if col is not None:
                hit[k] = 1
                point[k] = col.point
                refl[k] = col.point + col.rest()
                gave_up[k] = np.array_equal(final[k], old[k])
            assert np.array_equal(obs[:2], final[k])
        tag = env_id.replace("-", "_")
        for name, arr in dict(old=old, new=new, hit=hit, point=point, refl=refl, final=final, gave_up=gave_up, valid=valid).items():
            arrays[f"{tag}__{name}"] = arr
    return seg, arrays


# ---------------------------------------------------------------- reward / termination
def dump_rewards(rng):
    arrays, meta = {}, {}
    seen = set()
    for kw in REGISTERED:
        k = kw["kwargs"]
        cls = k["maze_task"]
        key = (cls.__name__, k["maze_size_scaling"])
        if key in seen:
            continue
        seen.add(key)
        task = cls(k["maze_size_scaling"])
        n = 400
        obs = rng.normal(0.0, 3.0 * k["maze_size_scaling"], (n, 12))
        # concentrate half of the samples around the goals, both in the agent slot and the object slot
        for gi, g in enumerate(task.goals):
            sel = slice(40 * gi, 40 * gi + 40)
            jitter = rng.normal(0.0, g.threshold * 0.8, (40, g.dim))
            obs[sel, : g.dim] = g.pos + jitter
            sel2 = slice(200 + 40 * gi, 240 + 40 * gi)
            jitter = rng.normal(0.0, g.threshold * 0.8, (40, g.dim))
            obs[sel2, 3 : 3 + g.dim] = g.pos + jitter
        # exactly on the threshold circle (inclusive `<=`): 3-4-5 triangles keep it exact in binary
        for gi, g in enumerate(task.goals[:1]):
            obs[390, : g.dim] = g.pos
            obs[390, 0] = g.pos[0] + g.threshold
            obs[391, 3 : 3 + g.dim] = g.pos
            obs[391, 3] = g.pos[0] + g.threshold
        rew = np.array([task.reward(o) for o in obs], dtype=np.float64)
        term = np.array([task.termination(o) for o in obs], dtype=np.uint8)
        tag = f"{cls.__name__}__{k['maze_size_scaling']}"
        arrays[f"{tag}__obs"] = obs
        arrays[f"{tag}__reward"] = rew
        arrays[f"{tag}__term"] = term
        meta[tag] = dict(task=cls.__name__, scale=k["maze_size_scaling"])
    return arrays, meta


# ---------------------------------------------------------------- Line known answers
def dump_line_kat():
    L = maze_env_utils.Line
    dist_cases = [((0.0, 0.0), (4.0, 4.0), (1.0, 3.0)), ((-3.0, -3.0), (0.0, 1.0), (-3.0, 1.0))]
    isect_cases = [((0.0, 0.0), (1.0, 0.0), (0.0, -1.0), (1.0, 1.0)), ((1.0, 1.0), (2.0, 3.0), (-1.0, 1.5), (1.5, 1.0)),
                   ((1.5, 1.5), (2.0, 3.0), (-1.0, 1.5), (1.5, 1.0)), ((0.0, 0.0), (2.0, 0.0), (1.0, 0.0), (1.0, 3.0))]
    out = dict(distance=[], intersect=[])
    for a, b, p in dist_cases:
        out["distance"].append(dict(l1=a, l2=b, p=p, ans=L(a, b).distance(complex(*p))))
    for a, b, c, d in isect_cases:
        r = L(a, b).intersect(L(c, d))
        out["intersect"].append(dict(l1p1=a, l1p2=b, l2p1=c, l2p2=d, ans=None if r is None else [r.real, r.imag]))
    # reflection / projection spot values
    ln = L((1.0, -2.0), (4.0, 2.0))
    out["reflection"] = [dict(l1=(1.0, -2.0), l2=(4.0, 2.0), p=(x, y), ans=[ln.reflection(complex(x, y)).real, ln.reflection(complex(x, y)).imag])
                         for x, y in [(0.0, 0.0), (3.5, -1.25), (-2.0, 7.0)]]
    return out


def main():
    rng = np.random.default_rng(20260928)
    with open(os.path.join(OUT, "registry.json"), "w") as f:
        json.dump(dump_registry(), f, indent=0, sort_keys=True)
    worlds = {e: dump_world(e) for e in ["PointUMaze-v0", "AntUMaze-v0", "Ant4Rooms-v0", "AntPush-v0", "Point4Rooms-v0",
                                         "PointPush-v0", "AntFall-v0", "PointBilliard-v0", "AntBlockMaze-v0"]}
    with open(os.path.join(OUT, "worlds.json"), "w") as f:
        json.dump(worlds, f, indent=0, sort_keys=True)
    seg, moves = dump_segments_and_moves(rng)
    with open(os.path.join(OUT, "segments.json"), "w") as f:
        json.dump(seg, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "detect.npz"), **moves)
    rew, meta = dump_rewards(rng)
    np.savez_compressed(os.path.join(OUT, "reward.npz"), **rew)
    with open(os.path.join(OUT, "reward_meta.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    with open(os.path.join(OUT, "line_kat.json"), "w") as f:
        json.dump(dump_line_kat(), f, indent=0, sort_keys=True)
    print("registered ids:", len(REGISTERED))


if __name__ == "__main__":
    main()

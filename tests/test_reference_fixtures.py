"""Round-2 fixtures (tests/golden/make_golden_r02.py) and the bit-exact pieces of the kernels, on the CPU.

* robots.json pins mujoco_maze_amd/robots.py against the reference's XML assets.
* obs_layout.json pins the observation layout and the reset noise pattern of every robot family against the reference's
  own `_get_obs` / `reset_model` code.
* The kernel sources that must reproduce the reference's float64 DECISIONS (csrc/point_dyn.h: mz_hypot, point_detect,
  point_bounce; csrc/ant_dyn.h: task_eval_dev) are compiled for the host (tests/emu) and run on ALL golden moves /
  observations.  The same vectors go through the HIP build in tests/test_gpu_golden.py.
"""
import json
import math
import os

import numpy as np
import pytest

import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd import model
from mujoco_maze_amd import robots as R

G = os.path.join(os.path.dirname(__file__), "golden")
ROBOTS = json.load(open(os.path.join(G, "robots.json")))
LAYOUT = json.load(open(os.path.join(G, "obs_layout.json")))
DET = np.load(os.path.join(G, "detect.npz"))
REW = np.load(os.path.join(G, "reward.npz"))
REW_F32 = np.load(os.path.join(G, "reward_f32.npz"))  # the reference's verdicts on float64(fp32(observations)): make_golden_r04.py
REW_META = json.load(open(os.path.join(G, "reward_meta.json")))

GEOM_TYPE = {"plane": R.PLANE, "sphere": R.SPHERE, "capsule": R.CAPSULE, "box": R.BOX, None: R.SPHERE}  # MuJoCo default geom type: sphere
JOINT_TYPE = {"free": R.FREE, "ball": R.BALL, "slide": R.SLIDE, "hinge": R.HINGE, None: R.HINGE}  # MuJoCo default joint type: hinge
MJ_GEOM_DEFAULT = dict(density=1000.0, contype=1, conaffinity=1, condim=3, friction=[1.0, 0.005, 0.0001], solref=[0.02, 1.0],
                       solimp=[0.9, 0.95, 0.001], margin=0.0)


def _v(ref, key, default):
    return default if ref.get(key) is None else ref[key]


def _check_geom(g: R.GeomSpec, ref, where):
    assert g.type == GEOM_TYPE[ref["type"]], where
    assert list(g.size)[: len(ref["size"])] == ref["size"], where
    if ref["fromto"] is not None:
        assert g.fromto is not None and [float(v) for v in g.fromto] == ref["fromto"], where
    else:
        assert g.fromto is None and list(g.pos) == _v(ref, "pos", [0.0, 0.0, 0.0]), where
    assert g.density == _v(ref, "density", MJ_GEOM_DEFAULT["density"]), where
    assert g.mass == ref["mass"], where
    for k in ("contype", "conaffinity", "condim", "margin"):
        assert getattr(g, k) == _v(ref, k, MJ_GEOM_DEFAULT[k]), (where, k)
    assert list(g.friction) == _v(ref, "friction", MJ_GEOM_DEFAULT["friction"]), where
    assert list(g.solref) == _v(ref, "solref", MJ_GEOM_DEFAULT["solref"]), where
    assert list(g.solimp)[:3] == _v(ref, "solimp", MJ_GEOM_DEFAULT["solimp"])[:3], where


@pytest.mark.parametrize("name", ["ant", "point", "swimmer", "reacher"])
def test_robot_spec_matches_the_reference_asset(name):
    ref, spec = ROBOTS[name], R.robot_spec(name)
    assert ref["compiler"] == {"angle": "degree", "coordinate": "local", "inertiafromgeom": "true"}  # what robots.py / compile_model assume
    opt = ref["option"]
    assert opt["integrator"] == "RK4" and float(opt["timestep"]) == spec.timestep
    assert float(opt.get("density", 0.0)) == spec.density and float(opt.get("viscosity", 0.0)) == spec.viscosity
    assert (opt.get("collision") == "predefined") == spec.collision_predefined
    assert len(ref["bodies"]) == len(spec.bodies)
    for i, (b, rb) in enumerate(zip(spec.bodies, ref["bodies"])):
        where = f"{name} body {i} ({rb['name']})"
        assert b.parent == rb["parent"] and list(b.pos) == rb["pos"], where
        if rb["name"] is not None:
            assert b.name == rb["name"], where
        assert len(b.joints) == len(rb["joints"]) and len(b.geoms) == len(rb["geoms"]), where
        for j, rj in zip(b.joints, rb["joints"]):
            assert j.name == rj["name"] and j.type == JOINT_TYPE[rj["type"]], where
            if j.type != R.FREE:
                assert list(j.axis) == rj["axis"] and list(j.pos) == _v(rj, "pos", [0.0, 0.0, 0.0]), (where, j.name)
            assert j.limited == bool(rj["limited"]), (where, j.name)
            if j.limited:
                assert list(j.range) == rj["range"], (where, j.name)
            for k in ("armature", "damping", "margin"):
                assert getattr(j, k) == _v(rj, k, 0.0), (where, j.name, k)
        for g, rg in zip(b.geoms, rb["geoms"]):
            _check_geom(g, rg, f"{where} geom {rg['name']}")
    floor = next(g for g in ref["world_geoms"] if g["name"] == "floor")
    _check_geom(spec.floor, floor, f"{name} floor")
    # maze boxes are emitted into the asset's worldbody and inherit its <default><geom> (maze_env.py:116-152)
    dg = ref["default_geom"]
    w = spec.wall_geom_defaults
    for k in ("contype", "condim", "margin", "density"):
        assert getattr(w, k) == _v(dg, k, MJ_GEOM_DEFAULT[k]), (name, "wall", k)
    assert list(w.friction) == _v(dg, "friction", MJ_GEOM_DEFAULT["friction"]) and list(w.solimp)[:3] == _v(dg, "solimp", MJ_GEOM_DEFAULT["solimp"])[:3]
    assert len(spec.actuators) == len(ref["actuators"])
    for a, ra in zip(spec.actuators, ref["actuators"]):  # order = ctrl index (ant.xml:71-78)
        assert ra["kind"] == "motor" and a.joint == ra["joint"] and a.gear == _v(ra, "gear", 1.0) and list(a.ctrlrange) == ra["ctrlrange"]
        assert a.ctrllimited == ra["ctrllimited"]


def _compile(env_id):
    spec = mm.REGISTRY[env_id]
    scale = spec.kwargs["maze_size_scaling"]
    return model.compile_model(spec.kwargs["model_cls"].ROBOT, spec.kwargs["maze_task"](scale), scale)


@pytest.mark.parametrize("env_id", sorted(LAYOUT))
def test_observation_layout_and_reset_pattern(env_id, oracle):
    ref = LAYOUT[env_id]
    try:
        cm = _compile(env_id)
    except NotImplementedError:
        pytest.skip("maze not on the device path yet")
    m = cm.c
    assert (m.nq, m.nv, m.obs_dim) == (ref["nq"], ref["nv"], ref["obs_dim"])
    # observation: which state entry lands in which slot (sentinel state through the oracle's obs assembly)
    st = dict(qpos=np.tile(1000.0 + np.arange(m.nq), (1, 1)), qvel=np.tile(2000.0 + np.arange(m.nv), (1, 1)),
              warm=np.zeros((1, m.nv)), t=np.array([8], np.int32))
    from tests import oracle_lib  # noqa: F401  (ctypes structs)
    import ctypes as C
    obs = np.zeros(m.obs_dim)
    es = (C.c_double * (28 + 24 + 24))()  # mzo_env_state: qpos[MZ_MAX_Q] qvel[MZ_MAX_DOF] warmstart[MZ_MAX_DOF] | t, status

    class EnvState(C.Structure):
        _fields_ = [("qpos", C.c_double * 28), ("qvel", C.c_double * 24), ("warm", C.c_double * 24), ("t", C.c_int32), ("status", C.c_int32)]

    s = EnvState()
    for i in range(m.nq):
        s.qpos[i] = st["qpos"][0, i]
    for i in range(m.nv):
        s.qvel[i] = st["qvel"][0, i]
    s.t = 8
    oracle.lib.mzo_env_obs(C.byref(m), C.byref(s), obs.ctypes.data_as(C.c_void_p))
    bodies = {}
    for k in range(m.nblock):
        bodies[m.block_bodyid[k]] = None
    for k, slot in enumerate(ref["layout"]):
        if slot.startswith("q"):
            assert obs[k] == 1000.0 + int(slot[1:]), (k, slot)
        elif slot.startswith("v"):
            assert obs[k] == 2000.0 + int(slot[1:]), (k, slot)
        elif slot.startswith("t*"):
            assert obs[k] == pytest.approx(8 * float(slot[2:]), rel=1e-15), (k, slot)
        else:  # body:<name>:<axis> — the body's frame origin: spawn position + its slide coordinate on that axis
            _, bname, axis = slot.split(":")
            b = ref["bodies"].index(bname)
            ids = list(m.ball_bodyid[: m.nball]) + list(m.block_bodyid[: m.nblock])
            body = ids[b]
            c = "xyz".index(axis)
            expect = m.body_pos[body][c]
            if m.jnt_type[m.body_jntadr[body]] == 0:  # free-joint object ball (AntSmallBilliard): the pose IS qpos
                expect = 1000.0 + m.jnt_qposadr[m.body_jntadr[body]] + c
            for j in range(m.body_jntadr[body], m.body_jntadr[body] + m.body_jntnum[body]):
                if m.jnt_type[j] == 2 and m.jnt_axis[j][c] == 1.0:  # slide along this axis
                    expect += 1000.0 + m.jnt_qposadr[j] - m.qpos0[m.jnt_qposadr[j]]
            assert obs[k] == expect, (k, slot)
    # reset: which entries are re-randomised (the library's RNG streams differ from numpy's: pattern and ranges only)
    rs, _ = oracle.reset(cm, 64, 5)
    q0 = np.array(m.qpos0[: m.nq])
    free = m.jnt_type[0] == 0
    for i, kind in enumerate(ref["reset_qpos"]):
        d = rs["qpos"][:, i] - q0[i]
        if kind == "none":
            assert np.all(d == 0.0), i
        elif free and 3 <= i < 7:
            assert np.ptp(rs["qpos"][:, i]) > 0.01  # noise, then normalised with the quaternion [ASSUME-8]
        else:
            assert kind == "uniform(-0.1,0.1)" and np.all(np.abs(d) <= 0.1) and np.ptp(d) > 0.1, i
    kinds = {"normal": "randn*0.1", "uniform01": "random*0.1", "uniform_sym": "uniform(-0.1,0.1)"}
    for i, kind in enumerate(ref["reset_qvel"]):
        v = rs["qvel"][:, i]
        if kind == "none":
            assert np.all(v == 0.0), i
        else:
            assert kind == kinds[cm.spec.reset_qvel] and np.ptp(v) > 0.05, i
            if kind == "random*0.1":
                assert v.min() >= 0.0 and v.max() < 0.1
            elif kind == "uniform(-0.1,0.1)":
                assert v.min() >= -0.1 and v.max() <= 0.1 and v.min() < 0.0


def test_device_hypot_is_glibc_hypot():
    """mz_hypot (csrc/point_dyn.h) against abs(complex), i.e. libm hypot — what the reference computes (maze_env_utils.py:96,189,201)."""
    from tests import emu_lib

    rng = np.random.default_rng(0)
    n = 200000
    x = rng.uniform(-30, 30, n) * rng.choice([1.0, 1e-3, 1e-9, 1e-14], n)
    y = rng.uniform(-30, 30, n) * rng.choice([1.0, 1e-3, 1e-9, 1e-14], n)
    x[:100] = 0.0
    y[50:150] = 0.0
    x[150:200] = y[150:200]
    y[200:260] = x[200:260] * 2.0 ** -rng.integers(40, 70, 60)  # around the ax >= ay * 2^54 shortcut
    mine = emu_lib.hypot(x, y)
    ref = np.array([abs(complex(a, b)) for a, b in zip(x, y)])  # abs(complex) = libm hypot (math.hypot is a different algorithm)
    assert np.array_equal(mine, ref)


@pytest.mark.parametrize("env_id", ["PointUMaze-v0", "Point4Rooms-v0", "PointPush-v0", "PointCorridor-v0", "PointTRoom-v0", "PointBilliard-v0"])
def test_kernel_detect_source_on_all_golden_moves(env_id):
    """Every one of the 3000 golden moves of the maze (zero-length, 1e-9, ending exactly on a wall line, bounces, give-ups)
    through the kernel's point_detect / point_bounce source compiled for the host: hit, give-up, collision point and final
    position identical to what the reference's MazeEnv.step computed."""
    from tests import emu_lib

    tag = env_id.replace("-", "_")
    old, new = DET[f"{tag}__old"], DET[f"{tag}__new"]
    hit, point, final = DET[f"{tag}__hit"], DET[f"{tag}__point"], DET[f"{tag}__final"]
    gave_up, valid = DET[f"{tag}__gave_up"].astype(bool), DET[f"{tag}__valid"].astype(bool)
    cm = _compile(env_id)
    h, pt, fin = emu_lib.point_detect(cm, old, new)
    assert np.array_equal(h[~valid], np.full((~valid).sum(), -1))  # collinear: the reference raised ZeroDivisionError
    v = valid
    assert np.array_equal(h[v] > 0, hit[v].astype(bool))
    assert np.array_equal((h[v] == 2), gave_up[v] & (hit[v] > 0)) or np.array_equal(fin[v], final[v])
    assert np.array_equal(fin[v], final[v])
    assert np.array_equal(pt[v & (hit > 0)], point[v & (hit > 0)])
    assert hit[v].sum() > 200


def _task_cases():
    for tag, meta in sorted(REW_META.items()):
        yield tag, meta


@pytest.mark.parametrize("tag,meta", list(_task_cases()))
def test_kernel_task_eval_source_on_all_golden_observations(tag, meta):
    """task_eval_dev (fp64 predicate on the fp32 observation) on all 400 golden observations of the task, rounded to fp32:
    termination flag, first matching goal and reward equal what the REFERENCE returned on those rounded observations
    (tests/golden/reward_f32.npz) — and so does this repository's Python mirror of maze_task.py — including the rows that
    sit exactly on the threshold circle (390 / 391) and every -v2 sub-goal task."""
    from tests import emu_lib

    cls = getattr(T, meta["task"])
    scale = meta["scale"]
    task = cls(scale)
    robot = next((r for r in ("point", "ant", "swimmer") if getattr(cls.MAZE_SIZE_SCALING, r) == scale), None)
    try:
        cm = model.compile_model(robot, task, scale)
    except NotImplementedError:
        pytest.skip("maze not on the device path yet")
    obs32 = REW[f"{tag}__obs"].astype(np.float32)
    obs = obs32.astype(np.float64)
    exp_r, exp_t, exp_g = REW_F32[f"{tag}__reward"], REW_F32[f"{tag}__term"].astype(bool), REW_F32[f"{tag}__goal"]
    assert np.array_equal(np.array([task.reward(o) for o in obs]), exp_r)  # the host mirror (host-judged user tasks run it)
    assert np.array_equal(np.array([bool(task.termination(o)) for o in obs]), exp_t)
    rew, done, gi = emu_lib.task_eval(cm, obs32)
    assert np.array_equal(done.astype(bool), exp_t)
    assert np.array_equal(rew, exp_r.astype(np.float32))
    assert np.array_equal(gi, exp_g)
    if task.goals:
        assert exp_t.sum() >= 5  # the fixture concentrates samples around the goals (3-D goals: fewer inside the sphere)

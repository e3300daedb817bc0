"""The general engine (csrc/generic_dyn.h) and what it opens (SURVEY 8f rank 3 leftover: SPIN plates, maze_env.py:119-120,575,649-660,
maze_env_utils.py:33,74-75, maze_task.py:67; rank 4: user robots with box geoms and in mazes with movable bodies,
agent_model.py:12-41, README.md:127) — on the CPU: the worlds against the MJCF the reference generates
(tests/golden/spin_worlds.json, made by tests/golden/make_golden_r06.py from the imported reference), the oracle's ball joint
against physical invariants, and the kernel's code on one lane (tests/emu) against the float64 oracle.  The device run of the
same cases is tests/test_gpu_general_engine.py."""
import json
import os

import numpy as np
import pytest

import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd import model
from mujoco_maze_amd.maze_env_utils import MazeCell
from tests import user_robots

G = os.path.join(os.path.dirname(__file__), "golden")
SPIN = json.load(open(os.path.join(G, "spin_worlds.json")))


class SpinUMaze(T.GoalRewardUMaze):
    PUT_SPIN_NEAR_AGENT = True
    OBSERVE_BLOCKS = True


class SpinCellMaze(T.GoalRewardPush):
    OBSERVE_BLOCKS = True

    @staticmethod
    def create_maze():
        E, B, R, S, M = MazeCell.EMPTY, MazeCell.BLOCK, MazeCell.ROBOT, MazeCell.SPIN, MazeCell.XY_BLOCK
        return [[B, B, B, B, B],
                [B, E, S, E, B],
                [B, R, E, M, B],
                [B, E, E, E, B],
                [B, B, B, B, B]]


SPIN_TASKS = {"SpinUMaze": SpinUMaze, "SpinCellMaze": SpinCellMaze}


@pytest.mark.parametrize("name", sorted(SPIN))
def test_spin_world_matches_the_reference(name, oracle):
    """Body positions, plate sizes, masses, joint lists, names, the solimp switch (`self.blocks` ignores a plate dropped by
    PUT_SPIN_NEAR_AGENT, maze_env.py:61,108-112) and the spawn observation, against what the reference's generator writes."""
    ref = SPIN[name]
    tname, robot = name.split("/")
    cm = model.compile_model(robot, SPIN_TASKS[tname](ref["scale"]), ref["scale"])
    m = cm.c
    assert [f"movable_{i}_{j}" for i, j, _ in cm.world.movable_cells()] == ref["movable_names"] and m.nblock == len(ref["movable"])
    assert cm.world.has_blocks == ref["blocks"] and m.obs_dim == ref["obs_dim"]
    nrobot_q = {"ant": 15, "point": 3}[robot]
    q = nrobot_q
    for k, mv in enumerate(ref["movable"]):
        b, g = m.block_bodyid[k], m.block_geomid[k]
        assert list(m.body_pos[b]) == mv["pos"] and list(m.geom_size[g]) == mv["geom"]["size"] and list(m.geom_pos[g]) == mv["geom"]["pos"]
        assert m.body_mass[b] == pytest.approx(mv["geom"]["mass"], rel=1e-12) and m.geom_type[g] == 6
        j0 = m.body_jntadr[b]
        want = [dict(slide=2, ball=1)[j["type"]] for j in mv["joints"]]
        assert [m.jnt_type[j0 + i] for i in range(m.body_jntnum[b])] == want
        for i, j in enumerate(mv["joints"]):
            assert m.jnt_limited[j0 + i] == 0 and j["limited"] == "false" and m.jnt_qposadr[j0 + i] == q
            if j["type"] == "slide":
                assert list(m.jnt_axis[j0 + i]) == j["axis"] and m.jnt_margin[j0 + i] == float(j["margin"])
            q += 4 if j["type"] == "ball" else 1
    assert q == m.nq
    stiff = ref["default_geom_solimp"] is not None and ref["default_geom_solimp"].startswith(".995")
    assert (m.wall_solimp[0] == 0.995) == stiff
    assert [list(bx) for bx in cm.world.wall_boxes()] == [bb["pos"] + bb["size"] for bb in ref["boxes"] if bb["name"].startswith("block_")]
    # spawn observation: qpos[:3] | block body positions | the rest (the reference's fake robot reports zeros for its own part)
    st, obs = oracle.reset(cm, 1, 0)
    want = np.array(ref["obs0"])
    nb3 = 3 * m.nblock
    assert np.array_equal(obs[0, 3:3 + nb3], want[3:3 + nb3]) and len(want) == m.obs_dim
    # the ball joint's coordinates are a unit quaternion that the robot's reset does not touch
    for k, mv in enumerate(ref["movable"]):
        for i, j in enumerate(mv["joints"]):
            if j["type"] == "ball":
                qa = m.jnt_qposadr[m.body_jntadr[m.block_bodyid[k]] + i]
                assert list(st["qpos"][0, qa:qa + 4]) == [1.0, 0.0, 0.0, 0.0]


TOP = """
<mujoco model="top">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.002"/>
  <default><geom conaffinity="0" contype="0" condim="3" density="500"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 0.1" pos="0 0 0" conaffinity="1" condim="3"/>
    <body name="torso" pos="0 0 3.0">
      <geom name="brick" type="box" size="0.3 0.2 0.1" pos="{off}"/>
      <joint name="spin" type="ball" pos="0 0 0"/>
    </body>
  </worldbody>
</mujoco>"""


def _world_ang_momentum(cm, q, w):
    """L = R I_body R^T (R w) of a single body on a ball joint through its centre of mass (w in the body frame)."""
    R = model.quat_to_mat(q / np.linalg.norm(q))
    I6 = np.array(cm.c.body_inertia[1])
    I = np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])
    return R @ (I @ w)


def test_ball_joint_torque_free_top(oracle):
    """A brick on a ball joint through its centre of mass feels no torque: kinetic energy and the WORLD-frame angular momentum are
    conserved while the body-frame angular velocity tumbles (three distinct moments: Euler's equations).  Pins the ball joint's
    kinematics, its motion axes (body-frame angular velocity), the gyroscopic bias and the quaternion integration."""
    cm = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=TOP.format(off="0 0 0"), frame_skip=1, reset_qvel="normal")
    m = cm.c
    assert (m.nq, m.nv, m.njnt, m.jnt_type[0]) == (4, 3, 1, 1) and list(m.qpos0[:4]) == [1.0, 0.0, 0.0, 0.0]
    q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q)
    w = np.array([1.0, 7.0, 0.5])  # near the unstable middle axis: it tumbles
    L0 = _world_ang_momentum(cm, q, w)
    drift = {}
    for h, n in ((0.002, 1500), (0.001, 3000)):
        cm.c.timestep = h
        qq, vv, en = oracle.raw_steps(cm, q.copy(), w.copy(), None, n)
        drift[h] = np.abs(_world_ang_momentum(cm, qq, vv) - L0).max() / np.abs(L0).max()
        assert np.abs(en - en[0]).max() < 1e-9 * abs(en[0]) + 1e-12, np.abs(en - en[0]).max()
        assert np.abs(vv - w).max() > 0.5 and abs(np.linalg.norm(qq) - 1.0) < 1e-12
    # MuJoCo-style RK4 updates the quaternion with q <- q * exp(h * sum_j b_j w_j) (no commutator correction): second order on SO(3) —
    # the residual is small AND shrinks ~4x when the step is halved, as for the free joint (tests/test_oracle_physics.py); a wrong
    # axis convention or bias term would not shrink at all
    assert drift[0.002] < 1e-5 and drift[0.001] < drift[0.002] / 3.5, drift


def test_ball_joint_pendulum_energy(oracle):
    """The same brick hung off-centre: a spherical pendulum under gravity — total energy conserved, and the mass matrix of the ball
    joint's three dofs equals the inertia about the pivot in the body frame."""
    cm = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=TOP.format(off="0.25 0.1 -0.4"), frame_skip=1, reset_qvel="normal")
    q = np.array([0.8, 0.3, 0.4, -0.1]); q /= np.linalg.norm(q)
    w = np.array([0.4, -1.2, 2.0])
    out = oracle.forward(cm, q[None], w[None])
    ms, c = cm.c.body_mass[1], np.array(cm.c.body_ipos[1])
    I6 = np.array(cm.c.body_inertia[1])
    I = np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]]) + ms * (c @ c * np.eye(3) - np.outer(c, c))
    assert np.allclose(out["M"][0], I, rtol=1e-12, atol=1e-15)
    drift = {}
    for h, n in ((0.002, 1500), (0.001, 3000)):
        cm.c.timestep = h
        _, _, en = oracle.raw_steps(cm, q.copy(), w.copy(), None, n)
        drift[h] = np.abs(en - en[0]).max()
    # of a kinetic energy of a few joules; second order in the step (the quaternion update, as above)
    assert drift[0.002] < 5e-4 and drift[0.001] < drift[0.002] / 3.5, drift


def _emu_vs_oracle(oracle, cm, st, acts, checks, tol=1e-6):
    """Step `st` (float64 dict) through the oracle with the action list; at the steps in `checks` run the kernel code on one lane from
    the same fp32-rounded state and compare.  Returns the number of contact rows seen."""
    from tests import emu_lib

    ncon = 0
    for k, act in enumerate(acts):
        act = act.astype(np.float32)  # both sides see the action the API carries (the Point's action is a displacement: 1e-8 of it matters)
        if k in checks:
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            s32 = dict(qpos=s64["qpos"].astype(np.float32), qvel=s64["qvel"].astype(np.float32), warm=s64["warm"].astype(np.float32), t=s64["t"].copy())
            ncon += int(oracle.forward(cm, s64["qpos"], s64["qvel"], act.astype(np.float64), s64["warm"])["counts"][:, 0].sum())
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=4)
            re_ = emu_lib.generic_env_step(cm, s32, act.astype(np.float32))
            for key in ("qvel", "qpos"):
                err = np.abs(s32[key] - s64[key]) - (tol + 2e-7 * np.abs(s64[key]))
                assert np.all(err <= 0), (k, key, float(err.max()), np.argwhere(err > 0)[:4])
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= tol + 2e-7 * np.abs(ro["obs"])), (k, np.abs(re_["obs"] - ro["obs"]).max())
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-6 and np.array_equal(re_["done"], ro["done"]) and np.array_equal(re_["goal_idx"], ro["goal_idx"])
            assert np.all((re_["status"] & 7) == 0) and np.all(ro["status"] == 0), (k, re_["status"], ro["status"])
        oracle.step(cm, st, act.astype(np.float64), nthreads=4)
    return ncon


@pytest.mark.parametrize("name", sorted(SPIN))
def test_spin_mazes_on_the_general_engine(name, oracle):
    """SPIN mazes step on the general engine (kernel code on one lane) as the oracle steps them: the robot next to — and on — the
    plate, the plate tilting under it (plane-box with tilted corners, sphere / capsule / box against a rotated box)."""
    ref = SPIN[name]
    tname, robot = name.split("/")
    cm = model.compile_model(robot, SPIN_TASKS[tname](ref["scale"]), ref["scale"])
    assert model.needs_general_engine(cm)
    m = cm.c
    n = 16
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(1)
    b = m.block_bodyid[0]
    px = m.body_pos[b][0]
    j0 = m.body_jntadr[b]
    qx, qb = m.jnt_qposadr[j0], m.jnt_qposadr[j0 + 2]
    # move the plate under / against the robot: slide it towards the origin, tilt it a little in half of the envs
    reach = 0.9 if robot == "ant" else 0.55
    st["qpos"][:, qx] = -px + rng.uniform(-reach, reach, n) if tname == "SpinUMaze" else st["qpos"][:, qx]
    if tname == "SpinCellMaze":  # bring the robot to the plate instead
        st["qpos"][:, 0] = m.body_pos[b][0] + rng.uniform(-reach, reach, n)
        st["qpos"][:, 1] = m.body_pos[b][1] + rng.uniform(-reach, reach, n)
    tilt = rng.normal(0, 0.15, (n, 3)) * (np.arange(n) % 2)[:, None]
    quat = np.concatenate([np.ones((n, 1)), 0.5 * tilt], 1)
    st["qpos"][:, qb:qb + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    amp = np.array([30.0] * 8) if robot == "ant" else np.array([1.0, 0.25])
    acts = [rng.uniform(-1, 1, (n, m.nu)) * amp for _ in range(13)]
    ncon = _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 1, 4, 12))
    assert ncon > 40  # the plate's floor corners, the robot on the plate
    dq = np.abs(st["qpos"][:, qb + 1:qb + 4]).max(1)
    assert (dq > 1e-3).sum() >= n // 4  # plates really tilt / spin


GENERAL_IDS = ["PointUMaze-v0", "PointPush-v0", "PointFall-v0", "PointBilliard-v0", "AntUMaze-v0", "AntPush-v0", "AntFall-v0", "AntMultiFall-v0",
               "AntSmallBilliard-v0", "SwimmerUMaze-v0", "ReacherUMaze-v0", "AntPushMaze-v0", "PointPushMaze-v0"]


@pytest.mark.parametrize("env_id", GENERAL_IDS)
def test_registered_mazes_on_the_general_engine(env_id, oracle):
    """engine="general": any registered id steps through the tree-walking engine — a second implementation of the whole path
    (teleport + wall bounce of the Point, object balls on hinge and free joints, falling blocks, platforms, the fluid model)."""
    spec = mm.REGISTRY[env_id]
    scale = spec.kwargs["maze_size_scaling"]
    robot = spec.kwargs["model_cls"].ROBOT
    cm = model.compile_model(robot, spec.kwargs["maze_task"](scale), scale, engine="general")
    assert model.needs_general_engine(cm) and cm.c.engine == 1
    m = cm.c
    n = 12
    st, _ = oracle.reset(cm, n, 2)
    rng = np.random.default_rng(3)
    lo = np.array([m.act_ctrlrange[a][0] for a in range(m.nu)]); hi = np.array([m.act_ctrlrange[a][1] for a in range(m.nu)])
    acts = [rng.uniform(lo, hi, (n, m.nu)) for _ in range(9)]
    _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 2, 8))


POINT_XML = os.path.join(os.path.dirname(mm.__file__), "assets", "point.xml")


def test_user_robot_with_a_box_geom_in_a_block_maze(oracle):
    """A user robot with box geoms in a maze with a movable block (both refused until round 6): the biped of tests/user_robots.py
    in the Push maze, pushed against the block; and a box-bodied slider (the reference's point.xml geometry, sphere + box) declared
    as ROBOT = "generic" with the Point's step shape."""
    cm = model.compile_model("generic", T.DistRewardPush(4.0), 4.0, robot_xml=user_robots.BIPED_ANT, frame_skip=5, reset_qvel="normal")
    n = 12
    st, _ = oracle.reset(cm, n, 7)
    rng = np.random.default_rng(5)
    m = cm.c
    bx = m.body_pos[m.block_bodyid[0]]
    st["qpos"][:, 0] = bx[0] - 2.0 - rng.uniform(0.2, 0.6, n)  # next to the block's western face
    st["qpos"][:, 1] = bx[1] + rng.uniform(-1.0, 1.0, n)
    acts = [rng.uniform(-20, 20, (n, m.nu)) for _ in range(9)]
    assert _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 3, 8)) > 20


def test_user_robot_whose_limbs_collide_with_each_other(oracle):
    """MuJoCo's default collision rule (every geom pair but parent-child) on a user robot: the arms of tests/user_robots.py PINCER close
    onto each other and a third arm's ball comes down on them — capsule-capsule (mjc_CapsuleCapsule) and sphere-capsule contacts between
    sibling bodies of ONE robot, two-body Jacobians; kernel code (host build) against the oracle."""
    cm = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=user_robots.PINCER, frame_skip=2, reset_qvel="normal")
    m = cm.c
    assert m.nv == 9 and m.nu == 3
    n = 12
    st, _ = oracle.reset(cm, n, 4)
    rng = np.random.default_rng(2)
    st["qpos"][:, 7] = -rng.uniform(0.10, 0.22, n)   # left arm swung towards -y ...
    st["qpos"][:, 8] = rng.uniform(0.10, 0.22, n)    # ... right arm towards +y: the arms cross / press near their tips
    st["qpos"][:, 9] = rng.uniform(0.15, 0.30, n)    # the top arm lowered: its ball between / on the arms
    pairs = oracle.forward(cm, st["qpos"], st["qvel"], np.zeros((n, 3)), st["warm"])["counts"][:, 0]
    assert pairs.min() >= 1  # (every env has a limb-limb contact to begin with; the bar floats 0.2 above the floor)
    acts = [np.column_stack([-rng.uniform(2, 10, n), rng.uniform(2, 10, n), rng.uniform(0, 10, n)]) for _ in range(8)]
    assert _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 1, 3, 7)) > 30


def test_joint_spring_is_a_harmonic_oscillator(oracle):
    """MJCF joint stiffness / springref (mj_passive: -k (q - springref)) in the oracle: an arm on a vertical hinge with a torsion spring and
    no damping swings about the spring's rest angle with omega = sqrt(k / I) — I the pivot's entry of the mass matrix (inertia + armature).
    Half a period from rest at 0 it stands at twice the rest angle, a full period later it is back; energy is not the test (the spring's
    potential is not in the oracle's energy sum), the closed form is."""
    k, ref = 1.5, 20.0
    cm = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=user_robots.SPRING_ARM.format(k=k, ref=ref, b=0, integ='integrator="RK4"', act=user_robots.SPRING_ARM_MOTOR), frame_skip=1, reset_qvel="normal")
    m = cm.c
    assert m.nv == 1 and m.jnt_stiffness[0] == k and np.isclose(m.jnt_springref[0], np.radians(ref)) and model.needs_general_engine(cm)
    inertia = oracle.forward(cm, np.zeros((1, 1)), np.zeros((1, 1)))["M"][0][0, 0]
    period = 2.0 * np.pi / np.sqrt(k / inertia)
    h = m.timestep
    half = int(round(0.5 * period / h))
    q, v, _ = oracle.raw_steps(cm, np.zeros(1), np.zeros(1), None, half)
    # (the step count rounds the half period to h: |q - 2 ref| <= amplitude * (omega h / 2)^2 / 2 + RK4's h^4 error)
    assert abs(q[0] - 2.0 * np.radians(ref)) < 2e-5 and abs(v[0]) < np.radians(ref) * np.sqrt(k / inertia) * (np.sqrt(k / inertia) * h), (q, v)
    q, v, _ = oracle.raw_steps(cm, q, v, None, half)
    assert abs(q[0]) < 1e-4


def test_euler_integrator_is_mujocos_semi_implicit_rule(oracle):
    """MuJoCo's default integrator for user robots (mj_EulerSkip, >= 2.1.2: implicit in the joint damping): on the damped spring arm — one dof,
    no constraint — a step is  a = (-k (q - r) - b v) / I;  a' = I a / (I + h b);  v += h a';  q += h v.  The oracle follows that recurrence
    to round-off (the kernel code follows the oracle: the next test — a one-dof arm is no maze robot for the device)."""
    k, ref, b = 1.5, 20.0, 0.3
    cm = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=user_robots.SPRING_ARM.format(k=k, ref=ref, b=b, integ="", act=user_robots.SPRING_ARM_MOTOR), frame_skip=1,
                             reset_qvel="normal")
    m = cm.c
    assert m.integrator_rk4 == 0 and model.needs_general_engine(cm)
    inertia = oracle.forward(cm, np.zeros((1, 1)), np.zeros((1, 1)))["M"][0][0, 0]
    h, r = m.timestep, np.radians(ref)
    q, v = 0.1, -0.4
    for _ in range(400):
        a = (-k * (q - r) - b * v) / inertia
        v += h * (inertia * a / (inertia + h * b))
        q += h * v
    qo, vo, _ = oracle.raw_steps(cm, np.array([0.1]), np.array([-0.4]), None, 400)
    assert abs(qo[0] - q) < 1e-12 and abs(vo[0] - v) < 1e-12


def test_user_robot_under_the_euler_integrator(oracle):
    """The biped with no integrator attribute (MuJoCo's default: Euler) at a 5 ms step: floor contacts, limits, motors — kernel code (host
    build) against the oracle over a rollout."""
    cm = model.compile_model("generic", T.GoalRewardUMaze(4.0), 4.0, robot_xml=user_robots.EULER_BIPED, frame_skip=8, reset_qvel="normal")
    assert cm.c.integrator_rk4 == 0
    n = 12
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-20, 20, (n, cm.c.nu)) for _ in range(9)]
    assert _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 3, 8)) > 20


def test_servo_actuators_closed_forms(oracle):
    """MJCF <position kp> and <velocity kv> (mj_fwdActuation: force = gain ctrl + bias1 length + bias2 velocity, length = gear q) on the
    one-dof arm.  Position servo, no spring, no damping: kp (ctrl - q) is a spring towards the target — from rest at 0 the arm stands at
    2 ctrl after half a period 2 pi / sqrt(kp / I).  Velocity servo: I v' = kv (ctrl - v), so v(t) = ctrl (1 - exp(-kv t / I))."""
    kp, target = 2.0, 0.3
    arm = lambda act: model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, frame_skip=1, reset_qvel="normal",  # noqa: E731
                                          robot_xml=user_robots.SPRING_ARM.format(k=0, ref=0, b=0, integ='integrator="RK4"', act=act))
    cm = arm(f'<position joint="pivot" kp="{kp}" ctrllimited="true" ctrlrange="-1 1"/>')
    m = cm.c
    assert m.act_gainprm[0] == kp and tuple(m.act_biasprm[0]) == (0.0, -kp, 0.0) and model.needs_general_engine(cm)
    inertia = oracle.forward(cm, np.zeros((1, 1)), np.zeros((1, 1)))["M"][0][0, 0]
    half = int(round(np.pi / np.sqrt(kp / inertia) / m.timestep))
    q, v, _ = oracle.raw_steps(cm, np.zeros(1), np.zeros(1), np.array([target]), half)
    assert abs(q[0] - 2.0 * target) < 2e-5, q
    q, v, _ = oracle.raw_steps(cm, np.zeros(1), np.zeros(1), np.array([5.0]), half)  # the control is clamped to its range first
    assert abs(q[0] - 2.0) < 2e-4, q
    kv, vt = 0.5, 0.8
    cm = arm(f'<velocity joint="pivot" kv="{kv}" ctrllimited="true" ctrlrange="-1 1"/>')
    assert tuple(cm.c.act_biasprm[0]) == (0.0, 0.0, -kv)
    n = 500
    q, v, _ = oracle.raw_steps(cm, np.zeros(1), np.zeros(1), np.array([vt]), n)
    assert abs(v[0] - vt * (1.0 - np.exp(-kv * n * cm.c.timestep / inertia))) < 1e-9, v


def test_user_robot_with_servo_actuators(oracle):
    """The biped with position servos on its knees and a velocity servo on its tail (tests/user_robots.py SERVO_BIPED) through the kernel
    code (host build) against the oracle: the state-dependent actuator forces are re-evaluated at every RK4 stage on both sides."""
    cm = model.compile_model("generic", T.GoalRewardUMaze(4.0), 4.0, robot_xml=user_robots.SERVO_BIPED, frame_skip=5, reset_qvel="normal")
    assert model.needs_general_engine(cm) and [cm.c.act_gainprm[a] for a in range(cm.c.nu)] == [1.0, 15.0, 1.0, 15.0, 2.0]
    n = 12
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-3, 3, (n, cm.c.nu)) * np.array([6.0, 1.0, 6.0, 1.0, 1.0]) for _ in range(9)]
    assert _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 3, 8)) > 20


def test_user_robot_with_joint_springs(oracle):
    """The springy biped (tests/user_robots.py: stiffness / springref on knees and tail) through the kernel code (host build) against the
    oracle: the passive spring forces reach qfrc_smooth on both sides."""
    cm = model.compile_model("generic", T.GoalRewardUMaze(4.0), 4.0, robot_xml=user_robots.SPRINGY_BIPED, frame_skip=5, reset_qvel="normal")
    plain = model.compile_model("generic", T.GoalRewardUMaze(4.0), 4.0, robot_xml=user_robots.BIPED_ANT, frame_skip=5, reset_qvel="normal")
    n = 12
    st, _ = oracle.reset(cm, n, 3)
    a0 = oracle.forward(cm, st["qpos"], st["qvel"], np.zeros((n, cm.c.nu)), st["warm"])["qacc"]
    a1 = oracle.forward(plain, st["qpos"], st["qvel"], np.zeros((n, cm.c.nu)), st["warm"])["qacc"]
    assert np.abs(a0 - a1).max() > 0.1  # the springs matter: knees at 30..70 degrees pulled towards 50, the tail towards -10
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-20, 20, (n, cm.c.nu)) for _ in range(9)]
    assert _emu_vs_oracle(oracle, cm, st, acts, checks=(0, 3, 8)) > 20


def test_plane_box_with_tilted_corners(oracle):
    """mjc_PlaneBox for a box of general orientation, through the model path: the SPIN plate of SpinUMaze/ant (half sizes 0.4, 0.4,
    0.2, centre 0.2 above the floor) tilted by 0.1 rad about y.  By hand: a corner (sx 0.4, sy 0.4, sz 0.2) sits at height
    0.2 - sx 0.4 sin(0.1) + sz 0.2 cos(0.1); the two bottom corners on the +x side are 0.0389 deep, the two on the -x side 0.0409
    above the floor (beyond the 0.01 margin), top corners never count."""
    cm = model.compile_model("ant", SpinUMaze(8.0), 8.0)
    m = cm.c
    q = np.array(m.qpos0[: m.nq])
    q[2] = 30.0  # the ant far above
    qb = m.jnt_qposadr[m.body_jntadr[m.block_bodyid[0]] + 2]
    th = 0.1
    q[qb:qb + 4] = [np.cos(th / 2), 0.0, np.sin(th / 2), 0.0]
    c = oracle.contacts(cm, q)
    g = m.block_geomid[0]
    floor = c[(c[:, 7] == 0) & (c[:, 8] == g)]
    depth = 0.2 - 0.4 * np.sin(th) - 0.2 * np.cos(th)
    assert len(floor) == 2 and np.allclose(floor[:, 0], depth, atol=1e-12) and depth < -0.038
    cx = 2.0 + 0.4 * np.cos(th) - 0.2 * np.sin(th)  # the plate's centre is at x = 2 (a quarter cell from the robot, maze_env.py:577)
    want = sorted((cx, y, 0.5 * depth) for y in (-0.4, 0.4))  # a contact sits midway between the corner and the plane
    assert np.allclose(sorted(map(tuple, floor[:, 1:4])), want, atol=1e-12) and np.allclose(floor[:, 4:7], [0, 0, 1])


def test_general_engine_narrow_phase_equals_the_oracles_on_random_poses(oracle):
    """csrc/generic_dyn.h gen_sphere_vs_box / gen_capsule_vs_box / gen_box_vs_box / gen_capsule_vs_capsule (host build, tests/emu) against the
    oracle's routines on 8000 random poses of boxes of general orientation — same contact count, same order, 1e-12."""
    import ctypes as C
    from tests import emu_lib

    lib = emu_lib.load()
    lib.emu_probe_pair.restype = C.c_int
    lib.emu_probe_pair.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_int, C.c_void_p]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rng = np.random.default_rng(0)
    kinds = ["capsule_box", "box_box", "sphere_box", "capsule_capsule"]
    seen = [0, 0, 0, 0]
    for it in range(8000):
        kind = it % 4
        q1, q2 = rng.normal(size=4), rng.normal(size=4)
        m1 = np.ascontiguousarray(model.quat_to_mat(q1 / np.linalg.norm(q1)))
        m2 = np.eye(3) if it % 5 == 0 else np.ascontiguousarray(model.quat_to_mat(q2 / np.linalg.norm(q2)))
        p1, p2, s1, s2 = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), rng.uniform(0.1, 0.6, 3), rng.uniform(0.1, 0.8, 3)
        want = np.asarray(oracle.probe_pair(kinds[kind], p1, m1, s1, p2, m2, s2, 0.01)).reshape(-1, 7)
        out = np.zeros((16, 7))
        n = lib.emu_probe_pair(kind, vp(p1), vp(m1), vp(s1), vp(p2), vp(m2), vp(s2), 0.01, 16, vp(out))
        assert n == len(want) and (n == 0 or np.abs(out[:n] - want).max() < 1e-12), (it, kinds[kind], n, len(want))
        seen[kind] += n
    assert min(seen) > 300

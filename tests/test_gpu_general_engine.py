"""GPU tests of the general engine (csrc/generic_dyn.h, generic_kernels.hip) through the C-ABI: SPIN plates (SURVEY 8f rank 3
leftover: maze_env.py:119-120,575,649-660, maze_task.py:67 PUT_SPIN_NEAR_AGENT), user robots with box geoms and in mazes with
movable bodies (8f rank 4: agent_model.py:12-41, README.md:127), and the cross-check of the specialised kernels against this second
device implementation (engine="general").  Float64 on both sides of the oracle comparison: 1e-6 on the fp32 state that is stored.
Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from tests.test_general_engine import SPIN, SPIN_TASKS, GENERAL_IDS
from tests.test_gpu_parity import _assert_step_parity, _close, _f32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def _step_against_oracle(torch, oracle, env, st, acts, checks, atol=1e-6, max_outlier_frac=0.0):
    cm = env.model
    contacts = 0
    for k, act in enumerate(acts):
        act = act.astype(np.float32)
        if k in checks:
            s64 = _f32(st)
            contacts += int(oracle.forward(cm, s64["qpos"], s64["qvel"], act.astype(np.float64), s64["warm"])["counts"][:, 0].sum())
            env.set_state(s64["qpos"], s64["qvel"], s64["warm"], s64["t"])
            obs, rew, done, info = env.step(torch.as_tensor(act, device=env.device))
            qpos, qvel, warm, t = [x.cpu().numpy() for x in env.get_state()]
            ref = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            dev_out = (obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy())
            ok = _assert_step_parity(oracle, cm, _f32(st), act, qpos, qvel, s64, atol=atol, max_outlier_frac=max_outlier_frac, dev_out=dev_out, ref_done=ref["done"])
            good = ok
            assert np.all(_close(dev_out[0][good], ref["obs"][good], atol=atol)) and np.all(_close(dev_out[1][good], ref["reward"][good], atol=atol))
            assert np.array_equal(dev_out[2][good], ref["done"][good]) and np.array_equal(info["goal_index"].cpu().numpy()[good], ref["goal_idx"][good])
            assert np.array_equal(t, s64["t"]) and np.all((env.status().cpu().numpy() & 7) == 0)
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    return contacts


@pytest.mark.parametrize("name", sorted(SPIN))
def test_spin_plate_mazes_on_the_device(torch, oracle, name):
    """A custom task with PUT_SPIN_NEAR_AGENT = True, and a maze with an explicit SPIN cell next to an XY block, for the Ant and the
    Point: compiled, stepped on the device (general engine), equal to the float64 oracle at 1e-6 — robot beside / on the plate, the
    plate tilted (plane-box with tilted corners, sphere / capsule / box against a box of general orientation)."""
    from mujoco_maze_amd.agent_model import AntEnv, PointEnv
    from mujoco_maze_amd.maze_env import VecMazeEnv

    ref = SPIN[name]
    tname, robot = name.split("/")
    n = 256
    env = VecMazeEnv(AntEnv if robot == "ant" else PointEnv, SPIN_TASKS[tname], num_envs=n, maze_size_scaling=ref["scale"])
    cm, m = env.model, env.model.c
    assert env.obs_dim == ref["obs_dim"] and any(m.jnt_type[j] == 1 for j in range(m.njnt))
    obs0 = env.reset(seed=5).cpu().numpy()
    st, ref_obs0 = oracle.reset(cm, n, 5)
    assert np.abs(obs0 - ref_obs0).max() < 2e-6
    rng = np.random.default_rng(1)
    b = m.block_bodyid[0]
    j0 = m.body_jntadr[b]
    qx, qb = m.jnt_qposadr[j0], m.jnt_qposadr[j0 + 2]
    reach = 0.9 if robot == "ant" else 0.55
    if tname == "SpinUMaze":
        st["qpos"][:, qx] = -m.body_pos[b][0] + rng.uniform(-reach, reach, n)  # the plate under / beside the robot
    else:
        st["qpos"][:, 0] = m.body_pos[b][0] + rng.uniform(-reach, reach, n)
        st["qpos"][:, 1] = m.body_pos[b][1] + rng.uniform(-reach, reach, n)
    tilt = rng.normal(0, 0.15, (n, 3)) * (np.arange(n) % 2)[:, None]
    quat = np.concatenate([np.ones((n, 1)), 0.5 * tilt], 1)
    st["qpos"][:, qb:qb + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    amp = np.array([30.0] * 8) if robot == "ant" else np.array([1.0, 0.25])
    acts = [rng.uniform(-1, 1, (n, m.nu)) * amp for _ in range(21)]
    # the stiff, feather-light plate (0.2 g) under a robot is as ill-conditioned as these mazes get: a handful of envs sit on
    # activation thresholds of the plate's corner contacts (the discontinuity route of _assert_step_parity proves each one)
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 1, 5, 20), max_outlier_frac=0.02)
    assert contacts > 600
    assert (np.abs(st["qpos"][:, qb + 1:qb + 4]).max(1) > 1e-3).sum() > n // 4  # plates really tilt / spin
    # a rollout under auto-reset: TimeLimit, no NaN / overflow / solver cap
    env.set_auto_reset(True)
    env.reset(seed=9)
    for k in range(60):
        obs, rew, done, info = env.step(torch.as_tensor(acts[k % len(acts)].astype(np.float32), device=env.device))
    assert torch.isfinite(obs).all() and np.all((env.status().cpu().numpy() & 7) == 0)
    env.close()


@pytest.mark.parametrize("env_id", GENERAL_IDS)
def test_registered_mazes_general_engine_against_oracle_and_specialised_kernels(torch, oracle, env_id):
    """engine="general": the tree-walking engine steps a registered id; it agrees with the float64 oracle at 1e-6 and with the
    robot family's specialised kernel at the parity bar (1e-5) from the same states, flags and goal indices equal."""
    n = 192
    gen = mm.make(env_id, num_envs=n, force_vec=True, engine="general")
    spe = mm.make(env_id, num_envs=n, force_vec=True)
    cm = gen.model
    assert cm.c.engine == 1 and spe.model.c.engine == 0
    st, _ = oracle.reset(cm, n, 2)
    rng = np.random.default_rng(3)
    m = cm.c
    lo = np.array([m.act_ctrlrange[a][0] for a in range(m.nu)]); hi = np.array([m.act_ctrlrange[a][1] for a in range(m.nu)])
    acts = [rng.uniform(lo, hi, (n, m.nu)) for _ in range(16)]
    frac = 0.02 if (m.nblock or m.nball) else 0.006
    _step_against_oracle(torch, oracle, gen, st, acts, checks=(0, 3, 15), max_outlier_frac=frac)
    # the two device implementations from one state
    s64 = _f32(st)
    act = torch.as_tensor(acts[5].astype(np.float32), device=gen.device)
    outs = []
    for env in (gen, spe):
        env.set_state(s64["qpos"], s64["qvel"], s64["warm"], s64["t"])
        obs, rew, done, info = env.step(act)
        outs.append((obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info["goal_index"].cpu().numpy()))
    close = np.all(_close(outs[0][0], outs[1][0], atol=1e-5), axis=1)
    assert close.mean() >= 1.0 - 2 * frac, (close.mean(), np.abs(outs[0][0] - outs[1][0]).max())
    assert np.array_equal(outs[0][2][close], outs[1][2][close]) and np.array_equal(outs[0][3][close], outs[1][3][close])
    assert np.all(np.abs(outs[0][1][close] - outs[1][1][close]) <= 1e-5 + 1e-4 * np.abs(outs[1][1][close]))
    gen.close(); spe.close()


def test_reference_point_xml_as_a_generic_robot_matches_the_point_kernel(torch, oracle):
    """The reference's own assets/point.xml — a sphere AND a box on a slide-slide-hinge root — declared by a user as ROBOT = "generic"
    with the Point's step shape (point.py:44-61), MANUAL_COLLISION and RADIUS: stepped by the general engine it reproduces the
    specialised Point kernel (observations at 1e-6: both compute in float64; wall-bounce decisions, done flags, goal indices equal)."""
    from mujoco_maze_amd import mjcf, robots
    from mujoco_maze_amd.agent_model import AgentModel
    from mujoco_maze_amd.maze_env import VecMazeEnv

    class MyPoint(AgentModel):  # what a user of the reference writes (agent_model.py:12-41), plus the data the device needs
        FILE = mjcf.spec_to_mjcf(robots.robot_spec("point"))  # the reference asset, resolved (tests/golden/robots.json pins it)
        ROBOT = "generic"
        STEP = "point"
        MANUAL_COLLISION = True
        RADIUS = 0.4
        VELOCITY_LIMITS = 10.0
        FRAME_SKIP = 1
        RESET_QVEL = "uniform01"
        ORI_IND = 2

    n = 1024
    gen = VecMazeEnv(MyPoint, T.DistRewardUMaze, num_envs=n, maze_size_scaling=4.0)
    spe = mm.make("PointUMaze-v0", num_envs=n, force_vec=True)
    assert gen.model.c.robot == 3 and gen.model.c.step_kind == 2 and gen.model.c.nseg == spe.model.c.nseg > 0
    st, _ = oracle.reset(spe.model, n, 4)
    rng = np.random.default_rng(7)
    st["qpos"][:, 0] = rng.uniform(-1.5, 9.5, n); st["qpos"][:, 1] = rng.uniform(-1.5, 9.5, n)  # anywhere in the maze, walls included
    st["qpos"][:, 2] = rng.uniform(-3.1, 3.1, n)
    s64 = _f32(st)
    hits = 0
    for k in range(6):
        act = torch.as_tensor(rng.uniform([-1, -0.25], [1, 0.25], (n, 2)).astype(np.float32), device=gen.device)
        outs = []
        for env in (gen, spe):
            env.set_state(s64["qpos"], s64["qvel"], s64["warm"], s64["t"])
            obs, rew, done, info = env.step(act)
            outs.append((obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info["goal_index"].cpu().numpy(), [x.cpu().numpy() for x in env.get_state()]))
        close = np.all(_close(outs[0][0], outs[1][0], atol=1e-6), axis=1)
        assert close.mean() > 0.995, (k, close.mean(), np.abs(outs[0][0] - outs[1][0]).max())  # the rest: contact-threshold envs of the penetration regime
        assert np.array_equal(outs[0][2][close], outs[1][2][close]) and np.array_equal(outs[0][3][close], outs[1][3][close])
        hits += int((np.abs(outs[1][4][0][:, :2] - s64["qpos"][:, :2]).max(1) > 1e-9).sum())
        s64 = _f32(dict(qpos=outs[1][4][0].astype(np.float64), qvel=outs[1][4][1].astype(np.float64), warm=outs[1][4][2].astype(np.float64), t=outs[1][4][3]))
    assert hits > n
    gen.close(); spe.close()


def test_user_robot_with_boxes_in_a_block_maze_on_the_device(torch, oracle):
    """SURVEY 8f rank 4 widened: a user robot in a maze WITH a movable block — its capsules against the block's box,
    the block against walls and floor — on the device, equal to the oracle at 1e-6."""
    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    BipedAnt, _ = user_robots.robot_classes()
    n = 256
    env = VecMazeEnv(BipedAnt, T.DistRewardPush, num_envs=n, maze_size_scaling=4.0)
    cm, m = env.model, env.model.c
    assert m.nblock == 1 and env.obs_dim == m.nq_robot + m.nv_robot + 1 + 3
    st, _ = oracle.reset(cm, n, 7)
    rng = np.random.default_rng(5)
    bx = m.body_pos[m.block_bodyid[0]]
    st["qpos"][:, 0] = bx[0] - 2.0 - rng.uniform(0.2, 0.6, n)
    st["qpos"][:, 1] = bx[1] + rng.uniform(-1.0, 1.0, n)
    acts = [rng.uniform(-20, 20, (n, m.nu)) for _ in range(16)]
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 3, 15), max_outlier_frac=0.02)
    assert contacts > 500
    moved = np.abs(st["qpos"][:, m.nq_robot:m.nq_robot + 2]).max(1)
    assert (moved > 1e-3).sum() > n // 8  # blocks get pushed
    env.close()


def test_user_robot_whose_limbs_collide_on_the_device(torch, oracle):
    """MuJoCo's default collision rule on a user robot (tests/user_robots.py PINCER: every geom pair but parent-child): capsule-capsule
    (mjc_CapsuleCapsule) and sphere-capsule contacts between sibling bodies of one robot, on the device, equal to the oracle at 1e-6."""
    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    n = 256
    env = VecMazeEnv(user_robots.pincer_class(), T.DistRewardUMaze, num_envs=n, maze_size_scaling=4.0)
    cm, m = env.model, env.model.c
    assert env.launch_info()["engine"] == 1 and m.nv == 9
    st, _ = oracle.reset(cm, n, 4)
    rng = np.random.default_rng(2)
    st["qpos"][:, 7] = -rng.uniform(0.10, 0.22, n)
    st["qpos"][:, 8] = rng.uniform(0.10, 0.22, n)
    st["qpos"][:, 9] = rng.uniform(0.15, 0.30, n)
    acts = [np.column_stack([-rng.uniform(2, 10, n), rng.uniform(2, 10, n), rng.uniform(0, 10, n)]) for _ in range(12)]
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 2, 11), max_outlier_frac=0.02)
    assert contacts > 600
    env.close()


def test_user_robot_with_joint_springs_on_the_device(torch, oracle):
    """MJCF joint stiffness / springref on a user robot (tests/user_robots.py SPRINGY_BIPED): the passive spring forces on the device, equal to
    the oracle at 1e-6 over a rollout with floor contacts."""
    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    n = 256
    env = VecMazeEnv(user_robots.springy_class(), T.GoalRewardUMaze, num_envs=n, maze_size_scaling=4.0)
    cm = env.model
    assert env.launch_info()["engine"] == 1 and any(cm.c.jnt_stiffness[j] != 0.0 for j in range(cm.c.njnt))
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-20, 20, (n, cm.c.nu)) for _ in range(16)]
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 3, 15), max_outlier_frac=0.02)
    assert contacts > 300
    env.close()


def test_user_robot_under_the_euler_integrator_on_the_device(torch, oracle):
    """MuJoCo's default integrator (Euler, implicit in the joint damping) for a user robot whose MJCF names none (tests/user_robots.py
    EULER_BIPED): one forward evaluation per mj_step on the device, equal to the oracle at 1e-6 over a rollout with floor contacts."""
    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    n = 256
    env = VecMazeEnv(user_robots.euler_class(), T.GoalRewardUMaze, num_envs=n, maze_size_scaling=4.0)
    cm = env.model
    assert env.launch_info()["engine"] == 1 and cm.c.integrator_rk4 == 0
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-20, 20, (n, cm.c.nu)) for _ in range(16)]
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 3, 15), max_outlier_frac=0.02)
    assert contacts > 300
    env.close()


def test_user_robot_with_servo_actuators_on_the_device(torch, oracle):
    """MJCF <position kp> / <velocity kv> actuators on a user robot (tests/user_robots.py SERVO_BIPED): state-dependent actuator forces on the
    device, equal to the oracle at 1e-6 over a rollout with floor contacts."""
    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    n = 256
    env = VecMazeEnv(user_robots.servo_class(), T.GoalRewardUMaze, num_envs=n, maze_size_scaling=4.0)
    cm = env.model
    assert env.launch_info()["engine"] == 1 and cm.c.act_biasprm[1][1] == -15.0
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(1)
    acts = [rng.uniform(-3, 3, (n, cm.c.nu)) * np.array([6.0, 1.0, 6.0, 1.0, 1.0]) for _ in range(16)]
    contacts = _step_against_oracle(torch, oracle, env, st, acts, checks=(0, 3, 15), max_outlier_frac=0.02)
    assert contacts > 300
    env.close()


def test_general_engine_top_down_view_and_record(torch, oracle):
    """TOP_DOWN_VIEW tasks and the sharded run's packed record on the general engine: view entries filled from the row's own robot /
    block positions, time entry behind the view, record = obs | reward | done."""
    from mujoco_maze_amd.agent_model import PointEnv
    from mujoco_maze_amd.maze_env import VecMazeEnv

    class ViewSpin(SPIN_TASKS["SpinCellMaze"]):
        TOP_DOWN_VIEW = True

    n = 64
    env = VecMazeEnv(PointEnv, ViewSpin, num_envs=n, maze_size_scaling=4.0, auto_reset=True)
    cm = env.model
    assert cm.c.top_down_view == 1 and env.obs_dim == 7 + 6 + 75
    rec = torch.zeros((n, env.obs_dim + 2), device=env.device)
    env.bind_record(rec)
    obs0 = env.reset(seed=3).cpu().numpy()
    st, ref0 = oracle.reset(cm, n, 3)
    assert np.abs(obs0 - ref0).max() < 2e-6
    rng = np.random.default_rng(0)
    for k in range(5):
        act = rng.uniform([-1, -0.25], [1, 0.25], (n, 2)).astype(np.float32)
        obs, rew, done, info = env.step(torch.as_tensor(act, device=env.device))
        ref = oracle.step(cm, st, act.astype(np.float64), nthreads=4)
        assert np.all(_close(obs.cpu().numpy(), ref["obs"], atol=2e-6)), np.abs(obs.cpu().numpy() - ref["obs"]).max()
        assert torch.equal(rec, torch.cat([obs, rew[:, None], done.float()[:, None]], 1))
    env.close()

"""CPU-side checks of the product's boundary and of the kernel logic.

* the C-ABI library loads and exports every symbol include/mazestep.h declares (no
  compute calls without a GPU);
* the product refuses to run without a GPU (no CPU fallback);
* the kernel's lane-group source (csrc/ant_dyn.h), compiled for the host by the
  one-lane emulation in tests/emu, agrees with the float64 oracle after one full
  MazeEnv.step — this is how kernel logic and fp32 numerics are iterated on a box
  without a GPU; the GPU tests repeat it on the real device."""
import os
import re

import numpy as np
import pytest

import mujoco_maze_amd as mm
from mujoco_maze_amd import _capi, model
from mujoco_maze_amd import maze_task as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "mazestep.h")).read()
    declared = set(re.findall(r"\b(mz_[a-z_0-9]+)\s*\(", header))
    declared -= {"mz_handle", "mz_model"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mz_abi_version() == model.MZ_ABI_VERSION
    assert lib.mz_model_sizeof() == model.C.sizeof(model.MzModel)


def test_dpp_hazards_of_the_built_library():
    """The row solver's hand-written `asm` blocks (csrc/ant_newton_rows.h: v_fmac_f32_dpp / v_mul_f32_dpp row_newbcast) are
    outside the compiler's hazard recogniser.  tools/check_dpp_hazards.py disassembles the built library and checks, on every
    path into every DPP instruction of every kernel, that no operand was written in the two preceding issue slots."""
    import subprocess
    import sys

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), "--so", _capi.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"(\d+) DPP instructions checked, 0 hazard", out.stdout)
    assert m and int(m.group(1)) > 10000, out.stdout[-500:]  # the fused row solver alone holds ~1000 per instantiation


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.MazeStepError):
        mm.make("AntUMaze-v0", num_envs=4)
    # and the package never imports the oracle or the emulation
    for root, _, files in os.walk(os.path.join(ROOT, "mujoco_maze_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", "") or f == "model.py", f
                assert "emu_lib" not in src and "libmzo" not in src and "libantemu" not in src, f


def _f32(st):
    out = {k: v.copy() for k, v in st.items()}
    for k in ("qpos", "qvel", "warm"):
        out[k] = out[k].astype(np.float32).astype(np.float64)
    return out


@pytest.mark.parametrize("env_id,scale", [("UMaze", 8.0), ("4Rooms", 4.0)])
def test_kernel_logic_emulation_matches_oracle(oracle, env_id, scale):
    from tests import emu_lib

    task = T.TaskRegistry.tasks(env_id)[0](scale)
    cm = model.compile_model("ant", task, scale)
    n = 128
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(0)
    worst_q = worst_v = 0.0
    for k in range(61):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32).astype(np.float64)
        if k in (0, 1, 10, 60):
            s64 = _f32(st)
            s32 = emu_lib.f32_state(s64)
            ro = oracle.step(cm, s64, act, nthreads=8)
            re_ = emu_lib.env_step(cm, s32, act)
            assert np.all(np.abs(s32["qpos"] - s64["qpos"]) <= 1e-5 + 1e-5 * np.abs(s64["qpos"]))
            assert np.all(np.abs(s32["qvel"] - s64["qvel"]) <= 1e-5 + 1e-5 * np.abs(s64["qvel"]))
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 1e-5 + 1e-5 * np.abs(ro["obs"]))
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-6
            assert np.array_equal(re_["done"], ro["done"]) and np.array_equal(re_["goal_idx"], ro["goal_idx"])
            assert np.all((re_["status"] & 3) == 0)
            worst_q = max(worst_q, np.abs(s32["qpos"] - s64["qpos"]).max())
            worst_v = max(worst_v, np.abs(s32["qvel"] - s64["qvel"]).max())
        oracle.step(cm, st, act, nthreads=8)
    assert worst_q < 2e-6 and worst_v < 1e-5


def test_kernel_logic_wall_contacts(oracle):
    """Sphere-box / capsule-box contacts against maze walls: identical contact sets and accelerations."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0)
    n = 64
    st, _ = oracle.reset(cm, n, 9)
    rng = np.random.default_rng(1)
    for _ in range(30):
        oracle.step(cm, st, rng.uniform(-30, 30, (n, 8)), nthreads=8)
    st["qpos"][:, 0] = 18.9 + rng.uniform(0.0, 0.5, n)  # east wall face of the start row is at x = 20: leg tips touch / dig in <= 0.3
    st["qpos"][:, 1] = rng.uniform(-3.5, 3.5, n)        # some near the corner walls at y = +-4
    st["qvel"][:, 0] = 2.0
    st = _f32(st)
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    f = emu_lib.forward(cm, st["qpos"], st["qvel"], act, st["warm"])
    g = oracle.forward(cm, st["qpos"], st["qvel"], act.astype(np.float64), st["warm"])
    assert np.array_equal(f["counts"][:, 0], g["counts"][:, 0])
    floor_only = oracle.forward(cm, st["qpos"] - np.array([10.0] + [0] * 14), st["qvel"], act.astype(np.float64), st["warm"])
    assert (g["counts"][:, 0] > floor_only["counts"][:, 0]).sum() > n // 4  # wall contacts really present
    err = np.abs(f["qacc"] - g["qacc"])
    assert np.all(err <= 1e-3 + 5e-6 * np.abs(g["qacc"])), err.max()
    s32 = emu_lib.f32_state(st)
    ro = oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    re_ = emu_lib.env_step(cm, s32, act)
    assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 2e-5 + 1e-5 * np.abs(ro["obs"]))


def test_kernel_logic_movable_block(oracle):
    """AntPush (BASELINE config 5): one movable XY block — block-floor / block-wall / robot-block contacts, 8-dof
    hub in the arrow solver, block xyz in the observation.  Contact *counts* may differ at the activation
    boundary (the resting block sits exactly at dist = margin, where the row force is zero), so parity is
    asserted on the results."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.DistRewardPush(8.0), 8.0)
    n = 96
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(0)
    st["qpos"][:32, 1] = 3.0 + rng.uniform(0.2, 0.6, 32)  # a third of the ants walk into the block (its face is at y = 4)
    st["qvel"][:32, 1] = 1.5
    moved = 0.0
    for k in range(41):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (2, 10, 40):
            s64 = _f32(st)
            s32 = emu_lib.f32_state(s64)
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.env_step(cm, s32, act)
            assert re_["obs"].shape == (n, 33)
            assert np.all(np.abs(s32["qvel"] - s64["qvel"]) <= 2e-5 + 1e-5 * np.abs(s64["qvel"])), np.abs(s32["qvel"] - s64["qvel"]).max()
            assert np.all(np.abs(s32["qpos"] - s64["qpos"]) <= 1e-5 + 1e-5 * np.abs(s64["qpos"]))
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 2e-5 + 1e-5 * np.abs(ro["obs"]))
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-6
            assert np.array_equal(re_["done"], ro["done"]) and np.all((re_["status"] & 7) == 0)
            moved = max(moved, np.abs(s64["qpos"][:, 15:]).max())
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    assert moved > 0.1  # the block really gets pushed


def test_kernel_logic_object_ball_on_a_free_joint(oracle):
    """AntSmallBilliard (AntEnv.OBJBALL_TYPE = "freejoint", maze_env.py:539-560): the ball's six dofs join the hub (its own
    mass block, centripetal + gravity bias), contacts floor -> ball, ball -> wall, torso -> ball, ball -> leg capsule, ball xyz
    (the body origin, which rolls around the centre) in the observation, goal test on the ball."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.GoalRewardSmallBilliard(2.0), 2.0)
    assert (cm.c.nq, cm.c.nv, cm.c.obs_dim, cm.c.nball) == (22, 20, 33, 1)
    n = 96
    st, obs0 = oracle.reset(cm, n, 5)
    assert np.array_equal(obs0[:, 3:6], np.tile([0.0, -2.0, 0.0], (n, 1)))  # the ball's cell is one row before the robot's
    rng = np.random.default_rng(0)
    st["qpos"][:48, 1] = -0.9 + rng.uniform(-0.1, 0.1, 48)  # half of the ants start next to the ball, walking into it
    st["qvel"][:48, 1] = -2.0
    rolled = 0.0
    for k in range(61):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (0, 3, 12, 30, 60):
            s64 = _f32(st)
            s32 = emu_lib.f32_state(s64)
            # one forward evaluation: mass matrix (ball block included), bias, contact count, qacc
            fo = oracle.forward(cm, s64["qpos"], s64["qvel"], act.astype(np.float64), s64["warm"])
            fe = emu_lib.forward(cm, s32["qpos"], s32["qvel"], act, s32["warm"])
            assert np.array_equal(fo["counts"][:, 0], fe["counts"][:, 0])
            assert np.abs(fe["M"] - fo["M"]).max() < 2e-6 and np.abs(fe["bias"] - fo["bias"]).max() < 2e-5
            assert np.all(np.abs(fe["qacc"] - fo["qacc"]) <= 5e-4 + 2e-4 * np.abs(fo["qacc"]))
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.env_step(cm, s32, act)
            assert re_["obs"].shape == (n, 33)
            ok = np.all(np.abs(s32["qvel"] - s64["qvel"]) <= 2e-5 + 1e-5 * np.abs(s64["qvel"]), axis=1) & \
                np.all(np.abs(s32["qpos"] - s64["qpos"]) <= 1e-5 + 1e-5 * np.abs(s64["qpos"]), axis=1)
            assert ok.mean() >= 0.98, (k, np.abs(s32["qvel"] - s64["qvel"]).max(1))
            assert np.all(np.abs(re_["obs"][ok] - ro["obs"][ok]) <= 2e-5 + 1e-5 * np.abs(ro["obs"][ok]))
            assert np.abs(re_["reward"][ok] - ro["reward"][ok]).max() < 1e-6
            assert np.array_equal(re_["done"][ok], ro["done"][ok]) and np.all((re_["status"] & 7) == 0)
            rolled = max(rolled, np.abs(s64["qpos"][:, 18] - 1.0).max())
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    assert rolled > 0.05  # the ball really gets kicked and rolls (quaternion leaves the identity)
    assert np.abs(st["qpos"][:, 15:17] - [0.0, -2.0]).max() > 0.5


@pytest.mark.parametrize("name,nblock,obs_dim", [("MultiPush", 2, 36), ("MultiPushSmall", 3, 39), ("PushMaze", 3, 39)])
def test_kernel_logic_multi_block_mazes(oracle, name, nblock, obs_dim):
    """Mazes with two / three movable XY blocks (maze_task.py:259-330; scale 2 = the reference's default for
    them is 4, halved here so that the ant starts within reach of walls and blocks): block-block contacts, hub of
    10 / 12 dofs, up to 72 contact slots.  Blocks are kicked so that they hit walls and each other."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.TaskRegistry.tasks(name)[0](2.0), 2.0)
    m = cm.c
    assert m.nblock == nblock and m.obs_dim == obs_dim
    n = 48
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(0)
    st["qvel"][:, 14:] = rng.uniform(-3, 3, (n, 2 * nblock))
    worst = []
    for k in range(21):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (0, 5, 20):
            s64 = _f32(st)
            s32 = emu_lib.f32_state(s64)
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.env_step(cm, s32, act)
            assert np.all((re_["status"] & 7) == 0) and np.all((ro["status"] & 7) == 0)
            per_env = (np.abs(s32["qvel"] - s64["qvel"]) / (1.0 + np.abs(s64["qvel"]))).max(1)
            worst.append(per_env)
            ok = per_env <= 1e-4
            assert np.all(np.abs(re_["obs"][ok] - ro["obs"][ok]) <= 1e-4 + 1e-5 * np.abs(ro["obs"][ok]))
            assert np.array_equal(re_["done"], ro["done"])
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    worst = np.concatenate(worst)
    assert np.median(worst) < 3e-6 and (worst <= 2e-5).mean() >= 0.97, (np.median(worst), (worst <= 2e-5).mean(), worst.max())


def test_wall_broad_phase_small_cells(oracle):
    """Cells narrower than twice the ant's reach (scale 2): walls on BOTH sides of the torso's cell are within
    reach at spawn, so the torso-level broad phase must look at all eight neighbours."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.TaskRegistry.tasks("UMaze")[0](2.0), 2.0)
    n = 64
    st, _ = oracle.reset(cm, n, 9)
    rng = np.random.default_rng(1)
    st["qpos"][:, 7:15:2] += rng.uniform(-0.5, 0.5, (n, 4))  # swing the hips so that the feet sweep the walls
    st["qpos"][:, 2] = 0.6
    s64 = _f32(st)
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    fo = oracle.forward(cm, s64["qpos"], s64["qvel"], act.astype(np.float64), s64["warm"])
    fe = emu_lib.forward(cm, s64["qpos"], s64["qvel"], act, s64["warm"])
    assert fo["counts"][:, 0].max() >= 2  # some feet do touch walls in this pose
    assert np.array_equal(fo["counts"][:, 0], fe["counts"][:, 0])
    assert np.all(np.abs(fo["qacc"] - fe["qacc"]) <= 2e-3 + 1e-5 * np.abs(fo["qacc"]))


def test_contact_overflow_is_flagged_not_fatal(oracle):
    """More contacts than slots (ants pressed into the floor of a maze with 2 m cells, legs across the walls): the kernel
    logic keeps the first 16, raises MZ_STATUS_CONTACT_OVERFLOW for the env and still returns finite numbers."""
    from tests import emu_lib

    cm = model.compile_model("ant", T.DistRewardUMaze(2.0), 2.0)
    n = 64
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(0)
    st["qpos"][:, 2] = rng.uniform(0.02, 0.2, n)
    st["qpos"][:, 7:15] += rng.uniform(-1, 1, (n, 8))
    st["qpos"][:, :2] += rng.uniform(-0.6, 0.6, (n, 2))
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    fe = emu_lib.forward(cm, st["qpos"], st["qvel"], act, st["warm"])
    assert fe["counts"][:, 0].max() == 16
    re_ = emu_lib.env_step(cm, emu_lib.f32_state(st), act)
    assert ((re_["status"] & 2) != 0).sum() >= 5 and np.all((re_["status"] & 1) == 0) and np.isfinite(re_["obs"]).all()


def test_point_step_logic_with_mujoco_wall_contacts(oracle):
    """Point (BASELINE config 2): teleport + RK4 + MuJoCo sphere-box / arrow box-box wall contacts + manual bounce,
    over the whole maze (about half of the random states have active MuJoCo contacts)."""
    from tests import emu_lib

    cm = model.compile_model("point", T.DistRewardUMaze(4.0), 4.0)
    n = 4096
    rng = np.random.default_rng(9)
    xmin, xmax, ymin, ymax = cm.world.xy_limits()
    st = dict(qpos=np.stack([rng.uniform(xmin, xmax, n), rng.uniform(ymin, ymax, n), rng.uniform(-3.2, 3.2, n)], 1),
              qvel=np.stack([rng.uniform(0, 0.1, n), rng.uniform(0, 0.1, n), rng.uniform(-12, 12, n)], 1),
              warm=np.zeros((n, 3)), t=rng.integers(0, 990, n).astype(np.int32))
    st = _f32(st)
    act = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.25, 0.25, n)], 1).astype(np.float32)
    act[: n // 4, 0] *= 4.0  # unclipped actions (point.py:44-55): long teleports -> bounces / give-ups
    s32 = dict(qpos=st["qpos"].astype(np.float32), qvel=st["qvel"].astype(np.float32), t=st["t"].copy())
    g = oracle.forward(cm, st["qpos"], st["qvel"])
    assert (g["counts"][:, 1] > 0).sum() > n // 4  # MuJoCo contact rows really active in the fixture
    start = st["qpos"].copy()
    ro = oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    re_ = emu_lib.point_env_step(cm, s32, act)
    ok = (ro["status"] & ~8) == 0  # 8 = collinear move (the reference raises ZeroDivisionError there)
    assert ok.mean() > 0.999 and np.all((ro["status"] & 256) == 0)
    assert np.all(np.abs(re_["obs"][ok] - ro["obs"][ok]) <= 1e-6 + 2e-7 * np.abs(ro["obs"][ok]))
    assert np.array_equal(re_["done"][ok], ro["done"][ok]) and np.array_equal(re_["goal_idx"][ok], ro["goal_idx"][ok])
    assert np.array_equal(re_["reward"][ok], ro["reward"][ok].astype(np.float32))
    gave_up = np.all(ro["obs"][:, :2] == start[:, :2].astype(np.float32).astype(np.float64), axis=1)
    assert gave_up.sum() > 20  # the give-up branch of the manual bounce is exercised


def test_point_deep_overlap_aligned_and_rotated(oracle):
    """csrc/point_bare.h, the deep overlap of the arrow with wall cells (least-penetration axis vertical): the per-lane rectangle
    clipping (pb_cand) for rotated headings and its serial duplicate filter for a heading of exactly 0 (edges parallel to the cell's:
    candidates coincide) — against the oracle's literal mjc_BoxBox, incl. arrows that straddle two or more cells."""
    from tests import emu_lib

    cm = model.compile_model("point", T.DistRewardUMaze(4.0), 4.0)
    boxes = np.array(cm.world.wall_boxes())
    rng = np.random.default_rng(5)
    xmin, xmax, ymin, ymax = cm.world.xy_limits()
    rows = []
    for theta0 in (0.0, None):
        got = np.zeros((0, 3))
        while len(got) < 256:
            th = np.zeros(4096) if theta0 is not None else rng.uniform(-3.1, 3.1, 4096)
            c = np.stack([rng.uniform(xmin - 1, xmax + 1, 4096), rng.uniform(ymin - 1, ymax + 1, 4096), th], 1)
            arrow = c[:, :2] + 0.6 * np.stack([np.cos(th), np.sin(th)], 1)
            near = ((np.abs(arrow[:, None, 0] - boxes[None, :, 0]) < boxes[None, :, 3] + 0.05) & (np.abs(arrow[:, None, 1] - boxes[None, :, 1]) < boxes[None, :, 4] + 0.05)).any(1)
            got = np.concatenate([got, c[near]])
        rows.append(got[:256])
    qpos = np.concatenate(rows)
    n = len(qpos)
    st = _f32(dict(qpos=qpos, qvel=np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n), rng.uniform(-3, 3, n)], 1), warm=np.zeros((n, 3)),
                   t=np.zeros(n, np.int32)))
    assert np.all(st["qpos"][:256, 2] == 0.0)
    g = oracle.forward(cm, st["qpos"], st["qvel"])
    assert (g["counts"][:, 1] >= 4).mean() > 0.5
    act = np.zeros((n, 2), np.float32)
    s32 = dict(qpos=st["qpos"].astype(np.float32), qvel=st["qvel"].astype(np.float32), t=st["t"].copy())
    ro = oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    re_ = emu_lib.point_env_step(cm, s32, act)
    assert np.all(ro["status"] == 0) and np.all(re_["status"] == 0)
    assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 1e-6 + 2e-7 * np.abs(ro["obs"])), np.abs(re_["obs"] - ro["obs"]).max()


def test_point_sincos_kernel():
    """pt_sincos (point_bare.h: Cody-Waite reduction + fdlibm kernels) against libm over the range the heading can take, the
    quadrant boundaries included; beyond 1e5 it IS libm."""
    from tests import emu_lib

    rng = np.random.default_rng(0)
    k = np.arange(-40, 41) * (np.pi / 2)
    x = np.concatenate([rng.uniform(-4, 4, 20000), rng.uniform(-1e4, 1e4, 20000), k, np.nextafter(k, 1e9), np.nextafter(k, -1e9), k + 1e-9, [0.0, 1e-300, 3e5, -7e8]])
    s, c = emu_lib.pt_sincos(x)
    assert np.max(np.abs(s - np.sin(x))) < 2.5e-16 and np.max(np.abs(c - np.cos(x))) < 2.5e-16


@pytest.mark.parametrize("name,nblock", [("Push", 1), ("MultiPush", 2), ("PushMaze", 3), ("BlockMaze", 1)])
def test_point_with_movable_blocks(oracle, name, nblock):
    """Point + movable XY blocks (PointPush & co., maze_task.py:179-330): the lane-group planar code (csrc/planar_dyn.h:
    sphere / arrow vs block, block vs wall, block vs block contacts; dense (3 + 2 NB)-dof Newton) against the oracle's
    general rigid-body path.  The robot teleports by its action every step, so it regularly lands inside a block."""
    from tests import emu_lib

    task_cls = T.TaskRegistry.tasks(name)[0]
    scale = task_cls.MAZE_SIZE_SCALING.point
    cm = model.compile_model("point", task_cls(scale), scale)
    m = cm.c
    assert m.nblock == nblock and m.nv == 3 + 2 * nblock and m.obs_dim == 7 + 3 * nblock
    n = 192
    st, _ = oracle.reset(cm, n, 1)
    assert np.all(st["qpos"][:, 3:] == 0.0) and np.all(st["qvel"][:, 3:] == 0.0)  # point.py:78-80: blocks reset to their spawn state
    rng = np.random.default_rng(0)
    moved, errs = 0.0, []
    for k in range(101):
        act = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.25, 0.25, n)], 1).astype(np.float32)
        if k % 20 == 0:
            s64 = _f32(st)
            s32 = dict(qpos=s64["qpos"].astype(np.float32), qvel=s64["qvel"].astype(np.float32), t=s64["t"].copy())
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.point_env_step(cm, s32, act)
            errs.append(np.abs(re_["obs"] - ro["obs"]).max(1))
            assert np.array_equal(re_["done"], ro["done"]) and np.array_equal(re_["goal_idx"], ro["goal_idx"])
            assert np.all((re_["status"] & ~8) == 0) and np.all((ro["status"] & ~8) == 0)
            moved = max(moved, np.abs(s64["qpos"][:, 3:]).max())
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    errs = np.concatenate(errs)
    # stiff contacts on a 0.2 g block: an env whose contact switches within round-off lands on the other branch
    assert np.median(errs) < 3e-7 and (errs <= 2e-6).mean() >= 0.99, (np.median(errs), (errs <= 2e-6).mean(), errs.max())
    assert moved > 0.5  # blocks really get pushed around


@pytest.mark.parametrize("name,idx", [("Billiard", 0), ("Billiard", 3), ("SmallBilliard", 0)])
def test_point_billiard_object_ball(oracle, name, idx):
    """PointBilliard / PointSmallBilliard (maze_task.py:627-762): one object ball (slide-x, slide-y, z hinge; 0.1 g) — ball vs wall,
    robot sphere vs ball, ball vs arrow contacts on the planar code; object-slot reward / termination on obs[3:6]."""
    from tests import emu_lib

    task_cls = T.TaskRegistry.tasks(name)[idx]
    scale = task_cls.MAZE_SIZE_SCALING.point
    cm = model.compile_model("point", task_cls(scale), scale)
    m = cm.c
    assert m.nball == 1 and m.nv == 6 and m.obs_dim == 10
    n = 192
    st, _ = oracle.reset(cm, n, 1)
    rng = np.random.default_rng(0)
    moved, errs, terminated = 0.0, [], 0
    for k in range(121):
        act = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.25, 0.25, n)], 1).astype(np.float32)
        if k % 20 == 0:
            s64 = _f32(st)
            s32 = dict(qpos=s64["qpos"].astype(np.float32), qvel=s64["qvel"].astype(np.float32), t=s64["t"].copy())
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.point_env_step(cm, s32, act)
            errs.append(np.abs(re_["obs"] - ro["obs"]).max(1))
            assert np.array_equal(re_["done"], ro["done"]) and np.array_equal(re_["goal_idx"], ro["goal_idx"])
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-6
            assert np.all((re_["status"] & ~8) == 0) and np.all((ro["status"] & ~8) == 0)
            moved = max(moved, np.abs(s64["qpos"][:, 3:5]).max())
            terminated += int((ro["done"] & 1).sum())
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    errs = np.concatenate(errs)
    assert np.median(errs) < 3e-7 and (errs <= 2e-6).mean() >= 0.99, (np.median(errs), (errs <= 2e-6).mean(), errs.max())
    assert moved > 0.5  # the ball gets knocked around


@pytest.mark.parametrize("robot", ["swimmer", "reacher"])
def test_swimmer_world_with_a_movable_block(oracle, robot):
    """SwimmerPush / ReacherPush.  The reference's swimmer `_get_obs` returns the WHOLE qpos / qvel (swimmer.py:50-54), so the
    block's two slide coordinates and velocities are part of the observation (18 numbers for SwimmerPush; layout pinned by
    tests/golden/obs_layout.json) and `reset_model` puts U(-0.1, 0.1) noise on them too (swimmer.py:56-69).
    `collision="predefined"` (swimmer.xml:3) leaves the block without any contact pair: at rest it just sits there; once it
    has a velocity — i.e. after every reference-style reset — the medium's drag on a 0.2 g box (viscous rate ~1.6e4 / s
    against h = 0.01) makes the explicit integration blow up, in MuJoCo (mujoco-py raises) as in both of our paths, which
    flag the env."""
    from tests import emu_lib

    cm = model.compile_model(robot, T.DistRewardPush(4.0), 4.0)
    m = cm.c
    nr = m.nv - 2
    assert m.nblock == 1 and m.nv_robot == m.nv and m.nq_robot == m.nq and m.obs_dim == m.nq + m.nv + 1 + 3
    n = 64
    st, obs0 = oracle.reset(cm, n, 3)
    assert np.ptp(st["qpos"][:, nr]) > 0.1 and np.ptp(st["qvel"][:, nr + 1]) > 0.1  # reset noise reaches the block
    assert np.array_equal(obs0[:, 6:6 + m.nq - 3], st["qpos"][:, 3:]) and np.array_equal(obs0[:, 3 + m.nq:3 + m.nq + m.nv], st["qvel"])
    blown = _f32(st)
    ro = oracle.step(cm, blown, np.zeros((n, m.nu)), nthreads=4)
    s32 = emu_lib.f32_state(_f32(st))
    re_ = emu_lib.swimmer_env_step(cm, s32, np.zeros((n, m.nu), np.float32))
    assert np.all(ro["status"] & 1) and np.all(re_["status"] & 1)  # unstable in both, every env
    # a block at rest: inert, and the two paths agree on the whole 2 nv + 4 observation
    st["qvel"][:, nr:] = 0.0
    rng = np.random.default_rng(0)
    for k in range(21):
        act = rng.uniform(-1.5, 1.5, (n, m.nu)).astype(np.float32)
        if k in (0, 20):
            s64 = _f32(st)
            s32 = emu_lib.f32_state(s64)
            q_block = s64["qpos"][:, nr:].copy()
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=4)
            re_ = emu_lib.swimmer_env_step(cm, s32, act)
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 1e-6 + 2e-7 * np.abs(ro["obs"]))
            assert np.allclose(re_["obs"][:, 3:5], np.array([0.0, 4.0]) + q_block, atol=1e-6) and np.all(re_["obs"][:, 5] == 1.0)  # block xyz slot
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-7 and np.all(re_["status"] == 0) and np.all(ro["status"] == 0)
            assert np.array_equal(s32["qpos"][:, nr:], q_block.astype(np.float32)) and np.all(s32["qvel"][:, nr:] == 0)
        oracle.step(cm, st, act.astype(np.float64), nthreads=4)


@pytest.mark.parametrize("robot", ["swimmer", "reacher"])
def test_swimmer_family_on_the_fall_maze(oracle, robot):
    """SwimmerFall / ReacherFall: the block has limited y / z slides and gravity pulls on the z slide (maze_env.py:607-648), the
    torso is lifted with the platforms (a planar chain does not notice), there are no contacts.  The observation carries the
    block's two slide coordinates and velocities (layout pinned by tests/golden/obs_layout.json: SwimmerFall 18 numbers).
    Dynamics: the medium's drag on the 1 g block is ~3000 / s against h = 0.01, so ANY motion of the block diverges under the
    explicit integrator — and gravity makes it move even from rest: the reference's env is unusable (mujoco-py raises), the
    oracle and the kernel source both flag every env after one step."""
    from tests import emu_lib

    cm = model.compile_model(robot, T.DistRewardFall(4.0), 4.0)
    m = cm.c
    nr = m.nv - 2
    assert m.elevated == 1 and m.nblock == 1 and m.nv_robot == m.nv and m.obs_dim == m.nq + m.nv + 4 and list(m.body_pos[1]) == [0.0, 0.0, 2.75]
    n = 32
    st, obs0 = oracle.reset(cm, n, 3)
    b = m.block_bodyid[0]
    assert np.allclose(obs0[:, 3], m.body_pos[b][0]) and np.allclose(obs0[:, 4], m.body_pos[b][1] + st["qpos"][:, nr])
    assert np.allclose(obs0[:, 5], m.body_pos[b][2] + st["qpos"][:, nr + 1])  # block xyz = spawn position + (y, z) slides
    for rest in (False, True):
        s64 = _f32(st)
        if rest:
            s64["qvel"][:, nr:] = 0.0
        s32 = emu_lib.f32_state(s64)
        ro = oracle.step(cm, s64, np.zeros((n, m.nu)), nthreads=4)
        re_ = emu_lib.swimmer_env_step(cm, s32, np.zeros((n, m.nu), np.float32))
        assert np.all(ro["status"] & 1) and np.all(re_["status"] & 1), rest


@pytest.mark.parametrize("robot,nq", [("swimmer", 5), ("reacher", 4)])
def test_swimmer_step_logic(oracle, robot, nq):
    """Swimmer (north_star) and Reacher (its 2-link variant, reacher.py / reacher.xml): the kernel's
    planar closed-form dynamics (link chain, inertia-box fluid forces, motor gear 150 with ctrl clamp, +-100 deg
    limits, RK4 x 4 frames) against the oracle's general rigid-body path — two independent formulations of the
    same model."""
    from tests import emu_lib

    cm = model.compile_model(robot, T.DistRewardUMaze(4.0), 4.0)
    nu = nq - 3
    assert cm.c.obs_dim == 2 * nq + 1 and cm.c.nq == nq and cm.c.nu == nu and cm.c.frame_skip == 4  # tests/test_envs.py:77-78
    n = 512
    st, obs0 = oracle.reset(cm, n, 3)
    assert np.all(np.abs(obs0[:, :2 * nq]) <= 0.1 + 1e-12)  # swimmer.py:55-66 / reacher.py:58-72: U(-.1,.1) on qpos and qvel
    rng = np.random.default_rng(0)
    st["qpos"][:, 3:] = rng.uniform(-1.9, 1.9, (n, nu))  # beyond the +-100 deg limits for some envs
    st["qvel"] = rng.normal(size=(n, nq)) * 2
    hit_limit = 0
    for k in range(21):
        act = rng.uniform(-1.5, 1.5, (n, nu)).astype(np.float32)  # outside the ctrl range too
        if k in (0, 5, 20):
            s64 = _f32(st)
            s32 = dict(qpos=s64["qpos"].astype(np.float32), qvel=s64["qvel"].astype(np.float32), t=s64["t"].copy())
            hit_limit += int((np.abs(s64["qpos"][:, 3:]) > np.radians(100)).any(1).sum())
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.swimmer_env_step(cm, s32, act)
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 1e-6 + 2e-7 * np.abs(ro["obs"]))
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-7 and np.abs(re_["info"] - ro["info"]).max() < 1e-6
            assert np.array_equal(re_["done"], ro["done"]) and np.all(re_["status"] == 0)
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    assert hit_limit > 50 * nu


def test_experiment_switches_cannot_reach_the_product_library(tmp_path):
    """the kernel sources carry MZ_EXP_* timing-experiment switches, several of them wrong physics by design.
    A translation unit that sees one without -DMZ_EXPERIMENTS does not compile (csrc/ant_model.h, which every unit includes), and
    the Makefile's product target refuses flags that carry either."""
    import subprocess

    csrc = os.path.join(ROOT, "mujoco_maze_amd", "csrc")
    src = tmp_path / "probe.cpp"
    src.write_text('#include <cstring>\n#include <cmath>\n#include <cstdint>\n#include "ant_model.h"\nint main() { return 0; }\n')
    base = ["g++", "-std=c++17", "-fsyntax-only", "-I", csrc, str(src)]
    assert subprocess.run(base, capture_output=True).returncode == 0
    for sw in ("MZ_EXP_NOWALL", "MZ_EXP_NONEWTON", "MZ_EXP_STAMPS"):
        bad = subprocess.run(base + [f"-D{sw}"], capture_output=True, text=True)
        assert bad.returncode != 0 and "MZ_EXPERIMENTS" in bad.stderr, sw
        assert subprocess.run(base + [f"-D{sw}", "-DMZ_EXPERIMENTS"], capture_output=True).returncode == 0, sw
    for var in ("EXTRA=-DMZ_EXP_NOWALL", "CXXFLAGS=-DMZ_EXPERIMENTS", "FAST=-freciprocal-math -DMZ_EXP_NOSEARCH"):
        out = subprocess.run(["make", "-C", csrc, "-n", "libmazestep.so", var], capture_output=True, text=True)
        assert out.returncode != 0 and "experiment switches" in out.stderr, var
    assert subprocess.run(["make", "-C", csrc, "-n", "libmazestep.so"], capture_output=True).returncode == 0

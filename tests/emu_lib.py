"""ctypes access to the CPU emulation of the kernel's lane-group code (tests/emu).
TEST INFRASTRUCTURE ONLY — see tests/emu/ant_emu.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from mujoco_maze_amd.model import MzModel

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
LIB = os.path.join(HERE, "libantemu.so")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

_lib = None


def load():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = C.CDLL(LIB)
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def env_step(cm, st, actions, max_iter=0, tol=0.0, rtol=-1.0):
    """st: dict of float32 qpos [n,15], qvel [n,14], warm [n,14], int32 t [n] (updated in place)."""
    lib = load()
    n = st["qpos"].shape[0]
    a = np.ascontiguousarray(actions, np.float32)
    out = dict(obs=np.zeros((n, cm.c.obs_dim), np.float32), reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
               goal_idx=np.zeros(n, np.int32), info=np.zeros((n, 4), np.float32), status=np.zeros(n, np.int32),
               iters=np.zeros(n, np.int32))
    rc = lib.emu_ant_env_step(C.byref(cm.c), n, _vp(st["qpos"]), _vp(st["qvel"]), _vp(st["warm"]), _vp(st["t"]), _vp(a),
                              _vp(out["obs"]), _vp(out["reward"]), _vp(out["done"]), _vp(out["goal_idx"]), _vp(out["info"]),
                              _vp(out["status"]), _vp(out["iters"]), C.c_int(max_iter), C.c_float(tol), C.c_float(rtol))
    assert rc == 0, rc
    return out


def forward(cm, qpos, qvel, actions=None, warm=None, max_iter=0, tol=0.0, rtol=-1.0):
    lib = load()
    qpos = np.ascontiguousarray(np.atleast_2d(qpos), np.float32)
    qvel = np.ascontiguousarray(np.atleast_2d(qvel), np.float32)
    n = qpos.shape[0]
    warm = None if warm is None else np.ascontiguousarray(np.atleast_2d(warm), np.float32)
    actions = None if actions is None else np.ascontiguousarray(np.atleast_2d(actions), np.float32)
    nv = cm.c.nv
    out = dict(qacc=np.zeros((n, nv), np.float32), counts=np.zeros((n, 2), np.int32), M=np.zeros((n, nv, nv), np.float32),
               bias=np.zeros((n, nv), np.float32), qas=np.zeros((n, nv), np.float32))
    rc = lib.emu_ant_forward(C.byref(cm.c), n, _vp(qpos), _vp(qvel), _vp(warm), _vp(actions), _vp(out["qacc"]), _vp(out["counts"]),
                             _vp(out["M"]), _vp(out["bias"]), _vp(out["qas"]), C.c_int(max_iter), C.c_float(tol), C.c_float(rtol))
    assert rc == 0, rc
    return out


def f32_state(st):
    return dict(qpos=st["qpos"].astype(np.float32), qvel=st["qvel"].astype(np.float32), warm=st["warm"].astype(np.float32),
                t=st["t"].copy())


def point_env_step(cm, st, actions):
    """st: dict of float32 qpos [n,nv], qvel [n,nv], int32 t [n] (updated in place); nv = 3 + 2 * movable blocks."""
    lib = load()
    n = st["qpos"].shape[0]
    a = np.ascontiguousarray(actions, np.float32)
    out = dict(obs=np.zeros((n, cm.c.obs_dim), np.float32), reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
               goal_idx=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
    rc = lib.emu_point_env_step(C.byref(cm.c), n, _vp(st["qpos"]), _vp(st["qvel"]), _vp(st["t"]), _vp(a), _vp(out["obs"]),
                                _vp(out["reward"]), _vp(out["done"]), _vp(out["goal_idx"]), _vp(out["status"]))
    assert rc == 0, rc
    return out


def pt_sincos(x):
    lib = load()
    x = np.ascontiguousarray(x, np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    lib.emu_pt_sincos(len(x), _vp(x), _vp(s), _vp(c))
    return s, c


def swimmer_env_step(cm, st, actions, lanes=False):
    """st: dict of float32 qpos [n,nv], qvel [n,nv], int32 t [n] (updated in place); nv = 5 Swimmer, 4 Reacher.
    lanes=True: the lane-group form of the step (the branch of swimmer_dyn.h the kernel runs), G host threads in lock step."""
    lib = load()
    n = st["qpos"].shape[0]
    a = np.ascontiguousarray(actions, np.float32)
    out = dict(obs=np.zeros((n, cm.c.obs_dim), np.float32), reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
               goal_idx=np.zeros(n, np.int32), info=np.zeros((n, 4), np.float32), status=np.zeros(n, np.int32))
    rc = (lib.emu_swimmer_env_step_lanes if lanes else lib.emu_swimmer_env_step)(C.byref(cm.c), n, _vp(st["qpos"]), _vp(st["qvel"]), _vp(st["t"]), _vp(a), _vp(out["obs"]),
                                  _vp(out["reward"]), _vp(out["done"]), _vp(out["goal_idx"]), _vp(out["info"]), _vp(out["status"]))
    assert rc == 0, rc
    return out


def hypot(x, y):
    """mz_hypot of csrc/point_dyn.h (the device's restatement of glibc hypot), element-wise on float64 arrays."""
    lib = load()
    lib.emu_hypot.restype = C.c_double
    lib.emu_hypot.argtypes = [C.c_double, C.c_double]
    return np.array([lib.emu_hypot(float(a), float(b)) for a, b in zip(np.ravel(x), np.ravel(y))])


def point_detect(cm, old_xy, new_xy):
    """point_bounce of csrc/point_dyn.h on float64 moves [n, 2]: (hit, point, final_xy)."""
    lib = load()
    o, w = np.ascontiguousarray(old_xy, np.float64), np.ascontiguousarray(new_xy, np.float64)
    n = o.shape[0]
    hit, pt, fin = np.zeros(n, np.int32), np.zeros((n, 2)), np.zeros((n, 2))
    rc = lib.emu_point_detect(C.byref(cm.c), n, _vp(o), _vp(w), _vp(hit), _vp(pt), _vp(fin))
    assert rc == 0, rc
    return hit, pt, fin


def top_down_view(cm, robot_xy, block_xy=()):
    """mzv_entry of csrc/mz_view.h (the code of view_fill_kernel), float64: view [75] for a torso at robot_xy, blocks at block_xy."""
    lib = load()
    b = np.zeros(8)
    flat = np.asarray(block_xy, np.float64).reshape(-1)
    b[:len(flat)] = flat
    view = np.zeros(75)
    lib.emu_top_down_view.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.emu_top_down_view.restype = None
    lib.emu_top_down_view(C.byref(cm.c), float(robot_xy[0]), float(robot_xy[1]), _vp(b), _vp(view))
    return view


def view_fill_rows(cm, rows, view_off):
    """mzv_fill_row on fp32 observation rows [n, obs_dim] whose view slots hold the parked block positions (in place)."""
    lib = load()
    r = np.ascontiguousarray(rows, np.float32)
    lib.emu_view_fill_rows.restype = None
    lib.emu_view_fill_rows(C.byref(cm.c), r.shape[0], r.shape[1], int(view_off), _vp(r))
    return r


def task_eval(cm, obs):
    """task_eval_dev of csrc/ant_dyn.h on fp32 observation rows [n, obs_dim]: (reward, done, goal_idx)."""
    lib = load()
    o = np.ascontiguousarray(obs, np.float32)
    n, d = o.shape
    rew, done, gi = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.int32)
    rc = lib.emu_task_eval(C.byref(cm.c), n, d, _vp(o), _vp(rew), _vp(done), _vp(gi))
    assert rc == 0, rc
    return rew, done, gi


def generic_env_step(cm, st, actions):
    """A user robot of any tree topology on the generic kernel code (csrc/generic_dyn.h, one lane): st = dict of float32
    qpos [n,nq], qvel [n,nv], warm [n,nv], int32 t [n] (updated in place)."""
    lib = load()
    n = st["qpos"].shape[0]
    a = np.ascontiguousarray(actions, np.float32)
    out = dict(obs=np.zeros((n, cm.c.obs_dim), np.float32), reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
               goal_idx=np.zeros(n, np.int32), info=np.zeros((n, 4), np.float32), status=np.zeros(n, np.int32))
    err = C.create_string_buffer(256)
    rc = lib.emu_generic_env_step(C.byref(cm.c), n, _vp(st["qpos"]), _vp(st["qvel"]), _vp(st["warm"]), _vp(st["t"]), _vp(a), _vp(out["obs"]),
                                  _vp(out["reward"]), _vp(out["done"]), _vp(out["goal_idx"]), _vp(out["info"]), _vp(out["status"]), err, 256)
    assert rc == 0, (rc, err.value.decode())
    return out


def generic_raw_steps(cm, qpos, qvel, nsteps, ctrl=None):
    """`nsteps` mj_step of the general engine's kernel code (one lane, float64 state in and out): (qpos, qvel) after them."""
    lib = load()
    qp, qv = np.array(qpos, np.float64), np.array(qvel, np.float64)
    c = None if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
    status = C.c_int32(0)
    err = C.create_string_buffer(256)
    rc = lib.emu_generic_raw_steps(C.byref(cm.c), _vp(qp), _vp(qv), None, None if c is None else _vp(c), int(nsteps), C.byref(status), err, 256)
    assert rc == 0 and (status.value & 7) == 0, (rc, status.value, err.value.decode())
    return qp, qv

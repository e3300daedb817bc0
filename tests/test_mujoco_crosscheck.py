"""Cross-check of the physics restatement against REAL MuJoCo — runs only where `import mujoco` works.

The build image has no MuJoCo, so the physics half of the oracle (oracle/mzo_physics.c) and with it the
kernels are PARITY UNPINNED against the reference's arithmetic.  This module is the one-command
path to pinning them: on a machine with `pip install mujoco`,

    python -m pytest tests/test_mujoco_crosscheck.py -q

loads the MJCF `tools/export_mjcf.py` writes for an env id (the model this repository believes it steps), and compares,
stage by stage so that a difference can be bisected to one of the numbered assumptions of DESIGN.md section 5:

  compile     body masses / inertias, dof_invweight0, body_invweight0, joint ranges        [ASSUME-10]
  kinematics  mass matrix (mj_fullM), bias forces                                         (M2, M3, M6)
  collision   contact count, distances, positions, normals per geom pair                   [ASSUME-5, 6, 7, 12, 13]
  rows        nefc, efc_D (= 1 / R), efc_aref                                              [ASSUME-2, 3, 4]
  solve       qacc of mj_forward (Newton, tolerance 1e-10 on both sides)                   [ASSUME-11]
  step        qpos / qvel after frame_skip x mj_step (RK4)                                 [ASSUME-1, 8] (M1)

Reference call sites being replaced: mujoco_maze/ant.py:57-63, point.py:56-59, swimmer.py:37-39 (`do_simulation` /
`mj_step`).  Needs no GPU: the comparison is oracle vs MuJoCo; the kernels are compared with the oracle elsewhere.
"""
import os
import sys

import numpy as np
import pytest

mujoco = pytest.importorskip("mujoco", reason="MuJoCo is not installed (it is not in the build image): physics parity stays unpinned")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from export_mjcf import compile_for, export_mjcf  # noqa: E402

ENVS = ["AntUMaze-v0", "Ant4Rooms-v0", "AntPush-v0", "PointUMaze-v0", "PointPush-v0", "SwimmerUMaze-v0", "ReacherUMaze-v0"]


def _load(env_id):
    cm = compile_for(env_id)
    model = mujoco.MjModel.from_xml_string(export_mjcf(env_id, cm=cm))
    model.opt.tolerance = 1e-10
    return cm, model, mujoco.MjData(model)


def _states(oracle, cm, n, steps, seed):
    """Seeded states along an oracle rollout (reset distribution, random actions): airborne, landing, walking."""
    rng = np.random.default_rng(seed)
    st, _ = oracle.reset(cm, n, seed)
    lo = np.array([cm.c.act_ctrlrange[a][0] for a in range(cm.c.nu)])
    hi = np.array([cm.c.act_ctrlrange[a][1] for a in range(cm.c.nu)])
    out = []
    for k in range(steps + 1):
        act = rng.uniform(lo, hi, (n, cm.c.nu))
        if k in (0, 1, steps // 4, steps):
            out.append(({kk: v.copy() for kk, v in st.items()}, act.copy()))
        oracle.step(cm, st, act, nthreads=4)
    return out


@pytest.mark.parametrize("env_id", ENVS)
def test_compiled_constants(env_id):
    cm, model, data = _load(env_id)
    m = cm.c
    assert (model.nq, model.nv, model.nu, model.nbody) == (m.nq, m.nv, m.nu, m.nbody)
    np.testing.assert_allclose(model.body_mass, np.array(m.body_mass[: m.nbody]), rtol=1e-12, err_msg="body masses (inertiafromgeom)")
    ours = np.array([[m.body_inertia[b][k] for k in range(3)] for b in range(m.nbody)])
    np.testing.assert_allclose(np.sort(model.body_inertia, axis=1), np.sort(ours, axis=1), rtol=1e-9, atol=1e-14, err_msg="principal inertias")
    np.testing.assert_allclose(model.dof_invweight0, np.array(m.dof_invweight0[: m.nv]), rtol=1e-9, err_msg="[ASSUME-10] dof_invweight0")
    np.testing.assert_allclose(model.body_invweight0, np.array([list(m.body_invweight0[b]) for b in range(m.nbody)]), rtol=1e-9, atol=1e-14,
                               err_msg="[ASSUME-10] body_invweight0")
    np.testing.assert_allclose(model.stat.meaninertia, m.meaninertia, rtol=1e-9)
    lim = [j for j in range(m.njnt) if m.jnt_limited[j]]
    np.testing.assert_allclose(model.jnt_range[lim], np.array([list(m.jnt_range[j]) for j in lim]), rtol=1e-12)


@pytest.mark.parametrize("env_id", ENVS)
def test_forward_dynamics_stage_by_stage(env_id, oracle):
    cm, model, data = _load(env_id)
    m = cm.c
    for st, act in _states(oracle, cm, 16, 60, 7):
        for e in range(16):
            data.qpos[:] = st["qpos"][e]
            data.qvel[:] = st["qvel"][e]
            data.qacc_warmstart[:] = st["warm"][e]
            data.ctrl[:] = 0.0 if m.robot == 0 else act[e]  # point.py never writes ctrl
            mujoco.mj_forward(model, data)
            ref = oracle.forward(cm, st["qpos"][e], st["qvel"][e], None if m.robot == 0 else act[e], st["warm"][e])
            M = np.zeros((model.nv, model.nv))
            mujoco.mj_fullM(model, M, data.qM)
            np.testing.assert_allclose(ref["M"][0], M, rtol=1e-9, atol=1e-12, err_msg="mass matrix (CRBA)")
            np.testing.assert_allclose(ref["bias"][0], data.qfrc_bias, rtol=1e-8, atol=1e-10, err_msg="bias forces (RNE)")
            assert ref["counts"][0, 0] == data.ncon, f"contact count: oracle {ref['counts'][0, 0]} vs MuJoCo {data.ncon} [ASSUME-5/6/7/12/13]"
            assert ref["counts"][0, 1] == data.nefc, f"constraint rows: oracle {ref['counts'][0, 1]} vs MuJoCo {data.nefc} [ASSUME-7]"
            cons = oracle.contacts(cm, st["qpos"][e])
            mine = sorted((round(float(c[0]), 9), tuple(np.round(c[1:4], 7))) for c in cons)
            theirs = sorted((round(float(data.contact[c].dist), 9), tuple(np.round(data.contact[c].pos, 7))) for c in range(data.ncon))
            assert mine == theirs, "contact distances / positions [ASSUME-5 capsule-plane, 6 capsule-box, 12 box-box, 13 rotated box]"
            np.testing.assert_allclose(ref["qacc"][0], data.qacc, rtol=1e-6, atol=1e-7, err_msg="qacc of mj_forward [ASSUME-2/3/4 rows, 11 solver]")


def _probe_geom_xml(name, kind_is_capsule, g, sphere=False):
    pos, mat, size = g
    q = np.zeros(4)
    mujoco.mju_mat2Quat(q, np.asarray(mat, float).ravel())
    if sphere:
        return f'<geom name="{name}" type="sphere" size="{size[0]}" pos="{pos[0]} {pos[1]} {pos[2]}"/>'
    if kind_is_capsule:
        return f'<geom name="{name}" type="capsule" size="{size[0]} {size[1]}" pos="{pos[0]} {pos[1]} {pos[2]}" quat="{q[0]} {q[1]} {q[2]} {q[3]}"/>'
    return f'<geom name="{name}" type="box" size="{size[0]} {size[1]} {size[2]}" pos="{pos[0]} {pos[1]} {pos[2]}" quat="{q[0]} {q[1]} {q[2]} {q[3]}"/>'


def test_narrowphase_probes_against_mujoco():
    """Per-routine verdict on the restated narrow phase: the hand-derived contact sets of tests/narrowphase_probes.py (which the
    oracle reproduces exactly, tests/test_narrowphase_probes.py) against mjc_CapsuleBox / mjc_BoxBox / mjc_CapsuleCapsule themselves.  Each probe is
    a two-geom model: geom1 on a free body, geom2 in the world (both static for mj_forward's collision stage)."""
    from tests import narrowphase_probes as NP

    verdict = {}
    for name, kind, g1, g2, margin, want in NP.probes():
        xml = f"""<mujoco><option gravity="0 0 0"/><default><geom margin="{margin}" contype="1" conaffinity="1"/></default><worldbody>
          <body name="a">{_probe_geom_xml("g1", kind in ("capsule_box", "capsule_capsule"), g1, sphere=kind == "sphere_box")}<freejoint/></body>
          {_probe_geom_xml("g2", kind == "capsule_capsule", g2)}</worldbody></mujoco>"""
        model = mujoco.MjModel.from_xml_string(xml)
        data = mujoco.MjData(model)
        mujoco.mj_forward(model, data)
        got = []
        for c in range(data.ncon):
            con = data.contact[c]
            n = np.array(con.frame[:3])
            if con.geom1 != mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_GEOM, "g1"):
                n = -n
            got.append([con.dist, *con.pos, *n])
        verdict[name] = NP.same_contact_set(got, want, tol=1e-7)
    bad = [k for k, v in verdict.items() if not v]
    assert not bad, "restatement differs from MuJoCo on: " + "; ".join(bad)


@pytest.mark.parametrize("env_id", ENVS)
def test_env_step_matches_mj_step(env_id, oracle):
    """What the hot path replaces: `do_simulation(action, frame_skip)` (ant.py:62, swimmer.py:38) / the Point's teleport +
    mj_step (point.py:45-59).  One MazeEnv.step of the oracle against frame_skip x mj_step from the same state."""
    cm, model, data = _load(env_id)
    m = cm.c
    for st, act in _states(oracle, cm, 8, 40, 3):
        for e in range(8):
            one = {k: v[e:e + 1].copy() for k, v in st.items()}
            qpos = st["qpos"][e].copy()
            qvel = st["qvel"][e].copy()
            if m.robot == 0:  # point.py:45-56
                qpos[2] += act[e, 1]
                if qpos[2] < -np.pi:
                    qpos[2] += 2 * np.pi
                elif np.pi < qpos[2]:
                    qpos[2] -= 2 * np.pi
                qpos[0] += np.cos(qpos[2]) * act[e, 0]
                qpos[1] += np.sin(qpos[2]) * act[e, 0]
                qvel = np.clip(qvel, -m.velocity_limit, m.velocity_limit)
            mujoco.mj_resetData(model, data)
            data.qpos[:] = qpos
            data.qvel[:] = qvel
            data.qacc_warmstart[:] = st["warm"][e]
            data.ctrl[:] = 0.0 if m.robot == 0 else act[e]
            for _ in range(m.frame_skip):
                mujoco.mj_step(model, data)
            old_xy = st["qpos"][e, :2].copy()
            oracle.step(cm, one, act[e:e + 1])
            exp_q = data.qpos.copy()
            if m.manual_collision:  # the maze's own wall rule runs after the physics (maze_env.py:451-464); golden-pinned elsewhere
                _, fin = oracle.bounce(cm, old_xy, exp_q[:2])
                exp_q[:2] = fin
            np.testing.assert_allclose(one["qpos"][0], exp_q, rtol=1e-7, atol=1e-8, err_msg="qpos after one env.step [ASSUME-1 RK4, 8 quaternion]")
            np.testing.assert_allclose(one["qvel"][0], data.qvel, rtol=1e-6, atol=1e-7, err_msg="qvel after one env.step")

"""MJCF subset reader / writer (mujoco_maze_amd/mjcf.py): user variants of the built-in robot assets."""
import numpy as np
import pytest

from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd import mjcf, model
from mujoco_maze_amd import robots as R


@pytest.mark.parametrize("robot,task,scale", [("ant", T.DistRewardUMaze, 8.0), ("ant", T.DistRewardPush, 8.0), ("point", T.DistRewardUMaze, 4.0),
                                              ("point", T.GoalRewardBilliard, 3.0), ("swimmer", T.DistRewardUMaze, 4.0),
                                              ("reacher", T.DistRewardUMaze, 4.0)])
def test_round_trip_is_lossless(robot, task, scale):
    """spec -> MJCF -> spec -> mz_model must be bit-identical to the built-in path (every model constant survives)."""
    xml = mjcf.spec_to_mjcf(R.robot_spec(robot))
    a = model.compile_model(robot, task(scale), scale)
    b = model.compile_model(robot, task(scale), scale, robot_xml=xml)
    assert bytes(a.c) == bytes(b.c)


# A terse, hand-written variant in the style of the reference assets (defaults carry most attributes, `freejoint`,
# motor defaults): a heavier ant with stronger, wider-ranged hips, longer ankles and a softer floor contact.
HEAVY_ANT = """
<mujoco model="heavy_ant">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.02"/>
  <default>
    <joint armature="1.5" damping="2" limited="true"/>
    <geom conaffinity="0" condim="3" density="8.0" friction="1.2 0.5 0.5" margin="0.01" solimp="0.85 0.85 0.01" solref="0.03 1"/>
    <motor ctrllimited="true" ctrlrange="-40 40"/>
  </default>
  <worldbody>
    <light name="ignored" pos="0 0 3"/>
    <geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
    <body name="torso" pos="0 0 0.75">
      <geom name="torso_geom" type="sphere" size="0.3"/>
      <freejoint name="root"/>
      {legs}
    </body>
  </worldbody>
  <actuator>
    {motors}
  </actuator>
</mujoco>
"""


def _heavy_ant_xml():
    legs, motors = [], []
    for k, (sx, sy, ax, lo, hi) in enumerate([(1, 1, "-1 1 0", 30, 70), (-1, 1, "1 1 0", -70, -30), (-1, -1, "-1 1 0", -70, -30),
                                              (1, -1, "1 1 0", 30, 70)], start=1):
        a, b = 0.2 * sx, 0.2 * sy
        legs.append(f"""
      <body name="leg_{k}" pos="0 0 0">
        <geom name="leg_geom_{k}" type="capsule" size="0.08" fromto="0 0 0 {a} {b} 0"/>
        <body name="aux_{k}" pos="{a} {b} 0">
          <joint name="hip_{k}" type="hinge" axis="0 0 1" pos="0 0 0" range="-35 35" margin="0"/>
          <geom name="aux_geom_{k}" type="capsule" size="0.08" fromto="0 0 0 {a} {b} 0"/>
          <body name="ankle_body_{k}" pos="{a} {b} 0">
            <joint name="ankle_{k}" type="hinge" axis="{ax}" pos="0 0 0" range="{lo} {hi}" margin="0"/>
            <geom name="ankle_geom_{k}" type="capsule" size="0.08" fromto="0 0 0 {2 * a} {2 * b} 0"/>
          </body>
        </body>
      </body>""")
    for j in ("hip_4", "ankle_4", "hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3"):
        motors.append(f'<motor joint="{j}" gear="1.5"/>')
    return HEAVY_ANT.format(legs="".join(legs), motors="\n    ".join(motors))


def test_user_variant_of_the_ant(oracle):
    """A parameter variant of the ant loads through `robot_xml`, changes exactly the constants it names, and the kernel
    logic (host emulation) follows the oracle on it."""
    from tests import emu_lib

    xml = _heavy_ant_xml()
    base = model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0).c
    cm = model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0, robot_xml=xml)
    m = cm.c
    assert (m.nq, m.nv, m.nu, m.nbody, m.ngeom) == (base.nq, base.nv, base.nu, base.nbody, base.ngeom)
    assert m.geom_size[1][0] == 0.3 and base.geom_size[1][0] == 0.25  # torso radius
    assert m.body_mass[1] == pytest.approx(8.0 * 4.0 / 3.0 * np.pi * 0.3 ** 3) and m.body_mass[1] > 2 * base.body_mass[1]
    assert m.act_gear[0] == 1.5 and list(m.act_ctrlrange[0]) == [-40.0, 40.0]
    assert m.dof_armature[6] == 1.5 and m.dof_damping[6] == 2.0 and m.dof_armature[0] == 0.0  # hinge defaults; free joint untouched
    assert np.allclose(list(m.jnt_range[1]), np.radians([-35.0, 35.0])) and m.jnt_limited[1]
    assert list(m.geom_solimp[1][:3]) == [0.85, 0.85, 0.01] and list(m.geom_solref[1]) == [0.03, 1.0]
    assert m.wall_contype == 1 and m.wall_conaffinity == 1 and m.geom_conaffinity[1] == 0 and m.geom_conaffinity[0] == 1
    n = 96
    st, _ = oracle.reset(cm, n, 2)
    rng = np.random.default_rng(0)
    for k in range(31):
        act = rng.uniform(-40, 40, (n, 8)).astype(np.float32)
        if k in (0, 10, 30):
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            s32 = emu_lib.f32_state(s64)
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            re_ = emu_lib.env_step(cm, s32, act)
            err = (np.abs(s32["qvel"] - s64["qvel"]) / (1.0 + np.abs(s64["qvel"]))).max(1)
            assert np.median(err) < 3e-6 and (err <= 2e-5).mean() >= 0.97, (np.median(err), err.max())
            assert np.all((re_["status"] & 7) == 0) and np.array_equal(re_["done"], ro["done"])
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)


def test_structural_changes_are_refused():
    xml = mjcf.spec_to_mjcf(R.robot_spec("ant"))
    with pytest.raises(ValueError, match="structure"):
        model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0, robot_xml=xml.replace('<motor joint="hip_4"', '<motor joint="hip_4" gear="2.0"/><motor joint="hip_4"'))
    with pytest.raises(ValueError, match="RK4"):
        model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0, robot_xml=xml.replace('integrator="RK4"', 'integrator="Euler"'))
    with pytest.raises(ValueError, match="not supported"):
        model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0, robot_xml=xml.replace('type="sphere"', 'type="ellipsoid"'))


# ------------------------------------------------------------------------------------------------ planar chains of other lengths
def chain_swimmer_xml(nlink, seg=0.8):
    """An n-link swimmer in the style of the reference's swimmer.xml (a user-supplied robot of another
    topology behind the same plugin surface): two slides + a hinge on the head, one limited, motorised hinge per further link."""
    body_open, body_close, motors = "", "", ""
    for k in range(1, nlink):
        pos = "0.5 0 0" if k == 1 else f"{-seg} 0 0"
        body_open += f"""
        <body name="link{k}" pos="{pos}">
          <geom name="g{k}" type="capsule" fromto="0 0 0 {-seg} 0 0" size="{0.1 - 0.01 * k}"/>
          <joint name="rot{k + 1}" type="hinge" axis="0 0 1" pos="0 0 0" limited="true" range="-80 80"/>"""
        body_close += "\n        </body>"
        motors += f'\n    <motor joint="rot{k + 1}" gear="{150 - 10 * k}" ctrllimited="true" ctrlrange="-1 1"/>'
    return f"""
<mujoco model="chain{nlink}">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.01" density="4000" viscosity="0.1" collision="predefined"/>
  <default>
    <geom conaffinity="1" condim="1" contype="1" density="1000"/>
    <joint armature="0.1"/>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 0.1" pos="0 0 -0.1" condim="3"/>
    <body name="torso" pos="0 0 0">
      <geom name="head" type="capsule" fromto="1.5 0 0 0.5 0 0" size="0.1"/>
      <joint name="slider1" type="slide" axis="1 0 0" pos="0 0 0"/>
      <joint name="slider2" type="slide" axis="0 1 0" pos="0 0 0"/>
      <joint name="rot" type="hinge" axis="0 0 1" pos="0 0 0"/>{body_open}{body_close}
    </body>
  </worldbody>
  <actuator>{motors}
  </actuator>
</mujoco>
"""


@pytest.mark.parametrize("nlink", [4, 5, 6])
def test_user_chain_of_another_length(nlink, oracle):
    """The swimmer family takes planar chains of 2..6 links from user MJCF: model constants from the XML, observation = whole
    qpos / qvel (swimmer.py:50-54), and the kernel's step function (csrc/swimmer_dyn.h through tests/emu) against the float64
    oracle's generic tree dynamics."""
    from tests import emu_lib

    cm = model.compile_model("swimmer", T.DistRewardUMaze(4.0), 4.0, robot_xml=chain_swimmer_xml(nlink))
    m = cm.c
    assert (m.nq, m.nv, m.nu, m.obs_dim) == (nlink + 2, nlink + 2, nlink - 1, 2 * (nlink + 2) + 1)
    assert m.act_gear[0] == 140.0 and m.act_gear[nlink - 2] == 150.0 - 10 * (nlink - 1) and m.nbody == nlink + 1
    n = 64
    st, obs0 = oracle.reset(cm, n, 3)
    assert np.array_equal(obs0[:, : nlink + 2], st["qpos"]) and np.array_equal(obs0[:, nlink + 2: -1], st["qvel"])
    rng = np.random.default_rng(1)
    moved = 0.0
    for k in range(61):
        act = rng.uniform(-1, 1, (n, nlink - 1)).astype(np.float32)
        if k in (0, 5, 30, 60):
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype == np.float64 else v.copy()) for kk, v in st.items()}
            s32 = emu_lib.f32_state(s64)
            ro = oracle.step(cm, s64, act.astype(np.float64))
            re_ = emu_lib.swimmer_env_step(cm, s32, act)
            assert np.abs(s32["qpos"] - s64["qpos"]).max() < 2e-6 and np.abs(s32["qvel"] - s64["qvel"]).max() < 2e-5
            assert np.abs(re_["obs"] - ro["obs"]).max() < 2e-5 and np.abs(re_["reward"] - ro["reward"]).max() < 1e-5
            assert np.array_equal(re_["done"], ro["done"]) and np.all(re_["status"] == 0)
        oracle.step(cm, st, act.astype(np.float64))
        moved = max(moved, np.abs(st["qpos"][:, 3:]).max())
    assert moved > 0.5  # the joints really bend (limits at +-80 degrees come into play)
    with pytest.raises(ValueError):
        model.compile_model("swimmer", T.DistRewardUMaze(4.0), 4.0, robot_xml=chain_swimmer_xml(7))


@pytest.mark.gpu
def test_user_chain_on_the_device(oracle):
    """A five-link chain from user MJCF on the HIP path (swimmer_step_kernel<5, 0>): single-step parity from rollout states."""
    import torch

    import mujoco_maze_amd as mm

    n = 512
    env = mm.make("SwimmerUMaze-v0", num_envs=n, robot_xml=chain_swimmer_xml(5))
    cm = env.model
    assert (env.nq, env.nu, env.obs_dim) == (7, 4, 15)
    st, _ = oracle.reset(cm, n, 7)
    rng = np.random.default_rng(2)
    for k in range(41):
        act = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        if k in (0, 10, 40):
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype == np.float64 else v.copy()) for kk, v in st.items()}
            env.set_state(s64["qpos"], s64["qvel"], None, s64["t"])
            obs, rew, done, info = env.step(torch.as_tensor(act, device=env.device))
            ref = oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
            qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
            assert np.abs(qpos - s64["qpos"]).max() < 2e-6 and np.abs(qvel - s64["qvel"]).max() < 2e-5
            assert np.abs(obs.cpu().numpy() - ref["obs"]).max() < 2e-5 and np.array_equal(done.cpu().numpy(), ref["done"])
            assert np.abs(rew.cpu().numpy() - ref["reward"]).max() < 1e-5
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    assert np.all(env.status().cpu().numpy() == 0)
    env.close()


# ------------------------------------------------------------------------------------------------ the chain kernel's lane-group branch on CPU
@pytest.mark.parametrize("robot,nlink", [("swimmer", 3), ("reacher", 2), ("swimmer", 4), ("swimmer", 5)])
def test_chain_lane_group_form_matches_one_lane_and_oracle(oracle, robot, nlink):
    """swimmer_step_kernel runs the `C::nlanes > 1` branches of csrc/swimmer_dyn.h (lane b = link b: per-lane link constants in ONE
    instruction stream, group sums, the slides' constant corner of M, closed-form limit rows for 2 / 3 links, the structured
    Cholesky start + hinge-column refactorisation of the Newton iteration for longer chains).  tests/emu runs that branch on
    G host threads in lock step (group sums in the order of the device's DPP butterfly): every lane must end with the same state,
    and the step must agree with the one-lane form and with the float64 oracle — with joints at their limits."""
    from tests import emu_lib

    xml = None if nlink <= 3 else chain_swimmer_xml(nlink)
    cm = model.compile_model(robot, T.DistRewardUMaze(4.0), 4.0, robot_xml=xml) if xml else model.compile_model(robot, T.DistRewardUMaze(4.0), 4.0)
    m = cm.c
    assert m.nv == nlink + 2
    n = 24
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(2)
    lim = np.array([m.jnt_range[3 + k][1] for k in range(nlink - 1)])
    at_limit = 0
    hold = np.sign(rng.uniform(-1, 1, (n, m.nu)))  # sustained torques drive the inner hinges into their limits
    for k in range(46):
        act = (hold * rng.uniform(0.5, 1.5, (n, m.nu))).astype(np.float32)
        if k % 15 == 0:
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype == np.float64 else v.copy()) for kk, v in st.items()}
            s_one, s_grp = emu_lib.f32_state(s64), emu_lib.f32_state(s64)
            at_limit += int((np.abs(s64["qpos"][:, 3:]) > lim).any(1).sum())
            ro = oracle.step(cm, s64, act.astype(np.float64))
            r1 = emu_lib.swimmer_env_step(cm, s_one, act)
            rg = emu_lib.swimmer_env_step(cm, s_grp, act, lanes=True)  # rc != 0 (asserted inside) if the lanes of a group disagree
            assert np.all(rg["status"] == 0) and np.array_equal(rg["done"], r1["done"]) and np.array_equal(rg["done"], ro["done"])
            assert np.abs(rg["obs"] - r1["obs"]).max() <= 2e-6 and np.abs(rg["reward"] - r1["reward"]).max() <= 1e-6
            assert np.abs(rg["obs"] - ro["obs"]).max() < 2e-5 and np.abs(rg["reward"] - ro["reward"]).max() < 1e-5
            assert np.abs(s_grp["qpos"] - s_one["qpos"]).max() <= 1e-6 and np.abs(s_grp["qvel"] - s_one["qvel"]).max() <= 2e-5
        oracle.step(cm, st, act.astype(np.float64))
        if k % 15 == 14: hold = np.sign(rng.uniform(-1, 1, (n, m.nu)))
    assert at_limit > 0  # limit rows were active at a checkpoint

_ORIENT = """
<mujoco model="orient">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"{seq}/>
  <option integrator="RK4" timestep="0.01"/>
  <default><geom conaffinity="0" condim="3" density="100"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
    <body name="torso" pos="0 0 0.5">
      <freejoint name="root"/>
      <geom name="g" type="capsule" {geom}/>
      <body name="arm" pos="0.3 0 0" {body}>
        <joint name="j" type="hinge" axis="0 0 1" limited="true" range="-30 30"/>
        <geom name="a" type="box" size="0.2 0.05 0.02" pos="0.2 0 0"/>
      </body>
    </body>
  </worldbody>
  <actuator><motor joint="j" gear="1" ctrllimited="true" ctrlrange="-1 1"/></actuator>
</mujoco>
"""


def _compiled(geom, body="", seq=""):
    from mujoco_maze_amd import maze_task as T
    from mujoco_maze_amd import model

    return model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=_ORIENT.format(geom=geom, body=body, seq=seq), frame_skip=1, reset_qvel="normal").c


def test_orientation_attributes_of_geoms_and_bodies():
    """quat / axisangle / euler / zaxis / xyaxes (MJCF's alternatives to fromto) on geoms and bodies: a capsule along (1, 1, 0) said five
    ways compiles to the same geom frame, inertia and mass as its fromto form; a body turned by 90 degrees about z carries its child
    geoms and joint axes with it.  Angles follow <compiler angle>, euler its eulerseq (lower case: moving axes)."""
    import numpy as np

    from mujoco_maze_amd import model

    ref = _compiled('size="0.05" fromto="-0.2 -0.2 0 0.2 0.2 0"')
    hl = 0.2 * np.sqrt(2.0)
    forms = [f'size="0.05 {hl}" pos="0 0 0" zaxis="1 1 0"',
             f'size="0.05 {hl}" axisangle="-1 1 0 90"',                       # z -> (1, 1, 0) / sqrt 2: a quarter turn about (-1, 1, 0)
             f'size="0.05 {hl}" euler="0 90 45" ',                           # with eulerseq="XYZ" (fixed axes), see below
             f'size="0.05 {hl}" xyaxes="0 0 -1 -1 1 0"',                     # x = -z, y = (-1, 1, 0): z = x x y = (1, 1, 0)
             'size="0.05 %.17g" quat="%.17g %.17g %.17g 0"' % (hl, np.cos(np.pi / 4), -np.sin(np.pi / 4) / np.sqrt(2), np.sin(np.pi / 4) / np.sqrt(2))]
    for k, form in enumerate(forms):
        seq = ' eulerseq="XYZ"' if "euler" in form else ""                     # fixed axes: about X by 0, then Y by 90 (z -> x), then Z by 45 (x -> (1, 1, 0))
        m = _compiled(form, seq=seq)
        zax = model.quat_to_mat(np.array(m.geom_quat[1]))[:, 2]
        assert np.allclose(np.abs(zax @ np.array([1, 1, 0]) / np.sqrt(2)), 1.0, atol=1e-12), (k, form, zax)
        assert np.allclose(m.geom_size[1][:2], ref.geom_size[1][:2]) and np.allclose(m.geom_pos[1], ref.geom_pos[1])
        assert np.isclose(m.body_mass[1], ref.body_mass[1], rtol=1e-12) and np.allclose(m.body_inertia[1], ref.body_inertia[1], atol=1e-12), (k, form)
    # a body turned by a quarter about z: its box geom's long axis and its hinge axis in the parent frame
    m = _compiled('size="0.05" fromto="-0.2 0 0 0.2 0 0"', body='euler="0 0 90"')
    R = model.quat_to_mat(np.array(m.body_quat[2]))
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-12) and np.allclose(R @ [0, 0, 1], [0, 0, 1], atol=1e-12)
    m2 = _compiled('size="0.05" fromto="-0.2 0 0 0.2 0 0"', body='quat="%.17g 0 0 %.17g"' % (np.cos(np.pi / 4), np.sin(np.pi / 4)))
    assert np.allclose(m.body_quat[2], m2.body_quat[2], atol=1e-12)
    # the turned arm's centre of mass sits at +y of the torso now: the composite inertia of the tree knows it
    # and the moving-axes default: xyz with (0, 90, 45) turns about y first, then about the NEW z — the geom z axis ends on +x
    m3 = _compiled('size="0.05 0.2" euler="0 90 45"')
    assert np.allclose(model.quat_to_mat(np.array(m3.geom_quat[1]))[:, 2], [1, 0, 0], atol=1e-12)


def test_physics_this_subset_lacks_is_an_error_not_a_skip():
    """An MJCF attribute or section that changes the physics and is not implemented stops the load (round 6: it was skipped — a robot with
    `frictionloss="0.1"` stepped without its dry friction; joint springs, first on that list, are implemented now).  Cosmetic attributes (rgba, material, group) and sections (asset, visual) pass."""
    ok = 'size="0.05" fromto="-0.2 0 0 0.2 0 0" rgba="1 0 0 1" material="m" group="1"'
    _compiled(ok)
    base = _ORIENT.format(geom=ok, body="", seq="")
    from mujoco_maze_amd import maze_task as T
    from mujoco_maze_amd import model

    def load(xml):
        return model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=xml, frame_skip=1, reset_qvel="normal")

    load(base.replace("</mujoco>", "<asset><texture name=\"t\" type=\"2d\" builtin=\"flat\" width=\"8\" height=\"8\"/></asset></mujoco>"))
    for bad, word in ((base.replace('range="-30 30"', 'range="-30 30" ref="10"'), "ref"),
                      (base.replace('range="-30 30"', 'range="-30 30" frictionloss="0.1"'), "frictionloss"),
                      (base.replace('range="-30 30"', 'range="-30 30" springdamper="1 1"'), "springdamper"),
                      (base.replace('gear="1"', 'gear="1 0 0 0 0 1"'), "gear"),
                      (base.replace('gear="1"', 'gear="1" forcelimited="true" forcerange="-1 1"'), "force"),
                      (base.replace("</mujoco>", "<tendon><fixed name=\"t\"><joint joint=\"j\" coef=\"1\"/></fixed></tendon></mujoco>"), "tendon"),
                      (base.replace("</mujoco>", "<equality><weld body1=\"torso\" body2=\"arm\"/></equality></mujoco>"), "equality"),
                      (base.replace('<option integrator="RK4"', '<option gravity="0 0 -1.6" integrator="RK4"'), "gravity"),
                      (base.replace('<option integrator="RK4"', '<option cone="elliptic" integrator="RK4"'), "cone"),
                      (base.replace('pos="0.2 0 0"/>', 'pos="0.2 0 0" fluidshape="ellipsoid"/>'), "fluidshape"),
                      (base.replace('<body name="arm" pos="0.3 0 0" >', '<body name="arm" pos="0.3 0 0" gravcomp="1">'), "gravcomp")):
        with pytest.raises(ValueError) as ei:
            load(bad)
        assert word in str(ei.value), (word, str(ei.value))

def test_default_classes_and_childclass():
    """MJCF default classes: a nested <default class=...> inherits the class around it; `class=` on an element picks one, `childclass=` on a
    body hands one down its subtree.  The same robot written flat (every attribute on the element) compiles to the same model."""
    import numpy as np

    from mujoco_maze_amd import maze_task as T
    from mujoco_maze_amd import model

    head = '<mujoco model="cls"><compiler angle="degree" coordinate="local" inertiafromgeom="true"/><option integrator="RK4" timestep="0.01"/>'
    nested = head + """
      <default>
        <geom conaffinity="0" condim="3" density="100" friction="1 0.5 0.5"/>
        <joint armature="0.1" damping="0.2" limited="true"/>
        <motor ctrllimited="true" ctrlrange="-1 1"/>
        <default class="heavy"><geom density="400"/>
          <default class="stiff"><joint damping="2" stiffness="3"/></default>
        </default>
      </default>
      <worldbody><geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
        <body name="torso" pos="0 0 0.5"><freejoint name="root"/>
          <geom name="g" type="capsule" size="0.05" fromto="-0.2 0 0 0.2 0 0"/>
          <body name="arm" pos="0.2 0 0" childclass="stiff">
            <joint name="j1" type="hinge" axis="0 0 1" range="-30 30"/>
            <geom name="a" type="capsule" size="0.04" fromto="0 0 0 0.3 0 0"/>
            <body name="hand" pos="0.3 0 0">
              <joint name="j2" type="hinge" axis="0 1 0" range="-20 20" class="main"/>
              <geom name="h" type="sphere" size="0.06" class="heavy"/>
            </body>
          </body>
        </body>
      </worldbody>
      <actuator><motor joint="j1" gear="2"/><motor joint="j2" gear="1"/></actuator></mujoco>"""
    flat = head + """
      <default><geom conaffinity="0" condim="3" friction="1 0.5 0.5"/><motor ctrllimited="true" ctrlrange="-1 1"/></default>
      <worldbody><geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
        <body name="torso" pos="0 0 0.5"><freejoint name="root"/>
          <geom name="g" type="capsule" size="0.05" fromto="-0.2 0 0 0.2 0 0" density="100"/>
          <body name="arm" pos="0.2 0 0">
            <joint name="j1" type="hinge" axis="0 0 1" range="-30 30" limited="true" armature="0.1" damping="2" stiffness="3"/>
            <geom name="a" type="capsule" size="0.04" fromto="0 0 0 0.3 0 0" density="400"/>
            <body name="hand" pos="0.3 0 0">
              <joint name="j2" type="hinge" axis="0 1 0" range="-20 20" limited="true" armature="0.1" damping="0.2"/>
              <geom name="h" type="sphere" size="0.06" density="400"/>
            </body>
          </body>
        </body>
      </worldbody>
      <actuator><motor joint="j1" gear="2"/><motor joint="j2" gear="1"/></actuator></mujoco>"""
    a = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=nested, frame_skip=1, reset_qvel="normal").c
    b = model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=flat, frame_skip=1, reset_qvel="normal").c
    for f in ("body_mass", "body_inertia", "dof_damping", "dof_armature", "jnt_stiffness", "jnt_range", "jnt_limited", "geom_friction", "act_gear", "act_ctrlrange"):
        assert np.allclose(np.array(getattr(a, f)), np.array(getattr(b, f))), f
    assert a.jnt_stiffness[1] == 3.0 and a.jnt_stiffness[2] == 0.0 and a.dof_damping[6] == 2.0 and a.dof_damping[7] == 0.2
    with pytest.raises(ValueError):
        model.compile_model("generic", T.DistRewardUMaze(4.0), 4.0, robot_xml=nested.replace('class="heavy"/>', 'class="nope"/>'), frame_skip=1, reset_qvel="normal")

"""A user's own robot behind the AgentModel surface (agent_model.py:12-41, README.md:127): MJCF of any tree
topology -> mz_model -> the generic tree kernel (csrc/generic_dyn.h).  Here the kernel code on one lane (tests/emu) against the
float64 oracle; the device run of the same robots is in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd import model
from tests import user_robots


def _compile(xml, task, scale, frame_skip, reset):
    return model.compile_model("generic", task(scale), scale, robot_xml=xml, frame_skip=frame_skip, reset_qvel=reset)


def test_generic_models_compile():
    cm = _compile(user_robots.BIPED_ANT, T.DistRewardUMaze, 4.0, 5, "normal")
    m = cm.c
    assert (m.robot, m.nbody, m.nq, m.nv, m.nu) == (3, 7, 12, 11, 5) and m.obs_dim == 12 + 11 + 1 and m.frame_skip == 5
    assert m.ngeom == 8 and m.body_parent[6] == 1 and m.act_gear[4] == 0.5  # floor + 7 robot geoms; the tail hangs off the torso
    cm = _compile(user_robots.BRANCHING_SWIMMER, T.DistRewardUMaze, 4.0, 4, "uniform_sym")
    m = cm.c
    assert (m.nbody, m.nq, m.nv, m.nu) == (5, 6, 6, 3) and m.body_parent[3] == 2 and m.body_parent[4] == 2 and m.collision_predefined == 1
    with pytest.raises(NotImplementedError):  # movable blocks: built-in robots only
        _compile(user_robots.BIPED_ANT, T.DistRewardPush, 4.0, 5, "normal")


@pytest.mark.parametrize("name,xml,fs,reset,amp", [("biped", user_robots.BIPED_ANT, 5, "normal", 20.0), ("yswimmer", user_robots.BRANCHING_SWIMMER, 4, "uniform_sym", 1.0)],
                         ids=["biped_ant", "y_swimmer"])
def test_generic_kernel_logic_matches_the_oracle(oracle, name, xml, fs, reset, amp):
    from tests import emu_lib

    cm = _compile(xml, T.DistRewardUMaze, 4.0, fs, reset)
    n = 48
    st, _ = oracle.reset(cm, n, 3)
    rng = np.random.default_rng(0)
    if name == "biped":  # a third of them next to the wall east of the start cell (its face is at x = 2): capsule-box contacts
        st["qpos"][: n // 3, 0] = 1.2 + rng.uniform(0.0, 0.4, n // 3)
    ncon_seen = 0
    for k in range(31):
        act = rng.uniform(-amp, amp, (n, cm.c.nu)).astype(np.float32)
        if k in (0, 3, 10, 30):
            s64 = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            s32 = dict(qpos=s64["qpos"].astype(np.float32), qvel=s64["qvel"].astype(np.float32), warm=s64["warm"].astype(np.float32), t=s64["t"].copy())
            ncon_seen += int(oracle.forward(cm, s64["qpos"], s64["qvel"], act.astype(np.float64), s64["warm"])["counts"][:, 0].sum())
            ro = oracle.step(cm, s64, act.astype(np.float64), nthreads=4)
            re_ = emu_lib.generic_env_step(cm, s32, act)
            # both sides compute in float64 from the same fp32 state: what is left is the fp32 rounding of the stored result
            assert np.all(np.abs(s32["qvel"] - s64["qvel"]) <= 1e-6 + 2e-7 * np.abs(s64["qvel"])), (k, np.abs(s32["qvel"] - s64["qvel"]).max())
            assert np.all(np.abs(s32["qpos"] - s64["qpos"]) <= 1e-6 + 2e-7 * np.abs(s64["qpos"]))
            assert np.all(np.abs(re_["obs"] - ro["obs"]) <= 1e-6 + 2e-7 * np.abs(ro["obs"]))
            assert np.abs(re_["reward"] - ro["reward"]).max() < 1e-6 and np.array_equal(re_["done"], ro["done"])
            assert np.all((re_["status"] & 7) == 0)
        oracle.step(cm, st, act.astype(np.float64), nthreads=4)
    if name == "biped":
        assert ncon_seen > 100  # floor and wall contacts really occur

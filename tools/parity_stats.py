#!/usr/bin/env python3
"""GPU box helper: single-step parity of the HIP path against the float64 oracle over rollout checkpoints, per config.
Prints a markdown table: median / 99 % / 99.9 % / max of the per-env max error over qpos and qvel normalised by 1 + |x| (the test
suite's 1e-5 + 1e-5 |x| bar), and — round 6, VERDICT r05 #2 — the ABSOLUTE errors of qpos and of qvel separately (99.9 % quantile
over all envs, maximum over the envs that are not branch flips) next to the largest |qpos| of the sample, flag mismatches.
`+far` configs start the ants 16 m from the origin in x and y (the far corner of the U maze: |x|, |y| up to 20 m).  For the Fall
family a second table splits the outliers by checkpoint (block still nested in its platform at k = 0 .. 2, free afterwards)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib

oracle = oracle_lib.load()
CONFIGS = [("AntUMaze-v0", 2048, (0, 1, 10, 50, 100, 200)), ("AntUMaze-v0+far", 2048, (0, 1, 10, 50, 100, 200)), ("Ant4Rooms-v0", 2048, (0, 10, 100)), ("AntPush-v0", 1024, (0, 10, 50, 100)),
           ("AntPushMaze-v0", 512, (0, 10, 50)), ("PointUMaze-v0", 2048, (0, 10, 50, 100)), ("PointPush-v0", 2048, (0, 10, 50, 100)),
           ("PointBilliard-v0", 2048, (0, 10, 50, 100)), ("SwimmerUMaze-v0", 2048, (0, 10, 100)), ("ReacherUMaze-v0", 2048, (0, 10, 100)),
           ("AntFall-v0", 1024, (0, 5, 20, 60)), ("AntMultiFall-v0", 1024, (5, 20, 60)), ("PointFall-v0", 2048, (0, 5, 20, 60)),
           ("AntSmallBilliard-v0", 1024, (0, 5, 20, 60)), ("PointPushMaze-v0", 1024, (0, 10, 50, 100)), ("AntMultiPush-v0@2", 512, (0, 5, 20)),
           ("AntPushMaze-v0@2", 512, (0, 5, 20)),
           # the seam experiment: the elevated mazes' floor is an array of platform boxes with a seam every
           # maze_size_scaling metres; the same mazes at the other one's scale
           ("AntMultiFall-v0@8", 1024, (5, 20, 60)), ("AntFall-v0@2", 1024, (0, 5, 20, 60))]  # "@2": at maze scale 2, as tests/test_gpu_parity.py steps the multi-block mazes
print("| config | envs x checkpoints | median | 99 % | 99.9 % | max | abs dqpos 99.9 % | abs dqpos max (no flips) | abs dqvel 99.9 % | abs dqvel max (no flips) | max abs qpos | envs > 1e-5 | of which the float64 oracle is itself discontinuous there | done / goal-index mismatches |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
BY_CHECK = []


def oracle_sensitive(cm, start, act, ref_qvel, e, rng, atol=1e-5):
    """The test suite's excuse for an outlier (tests/test_gpu_parity.py _assert_step_parity): re-running the ORACLE from the same
    state perturbed at the scale of the device's round-off moves its own answer by more than the tolerance."""
    for scale in (2e-7, 1e-6, 5e-6):
        for _ in range(8):
            p = {k: v[e:e + 1].copy() for k, v in start.items()}
            p["qpos"] = p["qpos"] + rng.uniform(-scale, scale, p["qpos"].shape) * np.maximum(1.0, np.abs(p["qpos"]))
            p["qvel"] = p["qvel"] + rng.uniform(-scale, scale, p["qvel"].shape) * np.maximum(1.0, np.abs(p["qvel"]))
            oracle.step(cm, p, act[e:e + 1].astype(np.float64))
            if np.abs(p["qvel"] - ref_qvel[e]).max() > atol:
                return True
    return False
# `parity_stats.py long`: the large-sample version (4 x the envs, a checkpoint every 10 steps of a 300-step rollout; the Push / Fall
# mazes every 10 steps of 100) — minutes of oracle time on the GPU box's host cores, written to profiles/<round>/parity_long.md
LONG = len(sys.argv) > 1 and sys.argv[1] == "long"
# `parity_stats.py general`: the general engine (csrc/generic_dyn.h) — registered ids forced onto it ("!general"), the SPIN-plate mazes of
# tests/test_general_engine.py ("spin:<task>/<robot>") and the user robots of tests/user_robots.py ("user:<robot>/<task class>")
if len(sys.argv) > 1 and sys.argv[1] == "general":
    CONFIGS = [("AntUMaze-v0!general", 512, (0, 1, 10, 50, 100)), ("AntPush-v0!general", 256, (0, 10, 50)), ("AntFall-v0!general", 256, (5, 20, 60)),
               ("PointUMaze-v0!general", 1024, (0, 1, 10, 50, 100)), ("PointPush-v0!general", 512, (0, 10, 50, 100)), ("SwimmerUMaze-v0!general", 1024, (0, 10, 100)),
               ("spin:SpinUMaze/ant", 256, (0, 1, 10, 30)), ("spin:SpinCellMaze/ant", 256, (0, 1, 10, 30)), ("spin:SpinUMaze/point", 512, (0, 1, 10, 30)),
               ("user:BipedAnt/DistRewardPush", 256, (0, 10, 50)), ("user:BipedAnt/GoalRewardUMaze", 512, (0, 10, 50, 100)), ("user:Pincer/DistRewardUMaze", 512, (0, 10, 50))]


def make_env(env_id, n):
    if env_id.startswith("spin:") or env_id.startswith("user:"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from mujoco_maze_amd import maze_task as T
        from mujoco_maze_amd.maze_env import VecMazeEnv
        from tests import user_robots
        what, task = env_id[5:].split("/")
        if env_id.startswith("spin:"):
            from tests.test_general_engine import SPIN_TASKS
            return VecMazeEnv(mm.AntEnv if task == "ant" else mm.PointEnv, SPIN_TASKS[what], num_envs=n, maze_size_scaling=4.0)
        cls = user_robots.pincer_class() if what == "Pincer" else user_robots.robot_classes()[0]
        return VecMazeEnv(cls, getattr(T, task), num_envs=n, maze_size_scaling=4.0)
    kw = {"maze_size_scaling": float(env_id.split("@")[1].split("!")[0])} if "@" in env_id else {}
    if env_id.endswith("!general"):
        kw["engine"] = "general"
    return mm.make(env_id.split("@")[0].split("+")[0].split("!")[0], num_envs=n, force_vec=True, **kw)
if LONG:
    CONFIGS = [(e, 4 * n, tuple(range(0, (301 if max(c) >= 100 else 101), 10))) for e, n, c in CONFIGS]
# experiments: MZ_PS_CONFIGS="AntUMaze-v0,AntPush-v0" keeps those configs, MZ_PS_OPTS="ls_iterations=0,max_iterations=60" sets handle options
if os.environ.get("MZ_PS_CONFIGS"):
    CONFIGS = [c for c in CONFIGS if c[0] in os.environ["MZ_PS_CONFIGS"].split(",")]
OPTS = [kv.split("=") for kv in os.environ.get("MZ_PS_OPTS", "").split(",") if kv]
for env_id, n, checks in CONFIGS:
    far = env_id.endswith("+far")
    env = make_env(env_id, n)
    carries_warm = env_id.startswith("Ant") or env.launch_info()["engine"] == 1
    for k_, v_ in OPTS: env.set_option(k_, float(v_))
    cm = env.model
    prng = np.random.default_rng(5)
    nsens, nout = 0, 0
    rng = np.random.default_rng(11)
    st, _ = oracle.reset(cm, n, 11)
    if far:
        st["qpos"][:, 0] += 16.0; st["qpos"][:, 1] += 16.0
    lo, hi = env.action_space.low, env.action_space.high
    errs, flags = [], 0
    aq, av, smooth, qmax, per_check, side_stats = [], [], [], 0.0, [], []
    for k in range(max(checks) + 1):
        act = rng.uniform(lo, hi, (n, env.nu)).astype(np.float32)
        if k in checks:
            s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            env.set_state(s["qpos"], s["qvel"], s["warm"] if carries_warm else None, s["t"])
            obs, rew, done, info = env.step(torch.as_tensor(act, device=env.device))
            qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
            start = {kk: v.copy() for kk, v in s.items()}
            ref = oracle.step(cm, s, act.astype(np.float64), nthreads=16)
            bad = np.where(~(np.all(np.abs(qvel - s["qvel"]) <= 1e-5 + 1e-5 * np.abs(s["qvel"]), axis=1) & np.all(np.abs(qpos - s["qpos"]) <= 1e-5 + 1e-5 * np.abs(s["qpos"]), axis=1)))[0]
            nout += len(bad)
            nsens += sum(oracle_sensitive(cm, start, act, s["qvel"], int(b), prng) for b in bad[:200]) + max(0, len(bad) - 200)
            e = np.maximum((np.abs(qvel - s["qvel"]) / (1 + np.abs(s["qvel"]))).max(1), (np.abs(qpos - s["qpos"]) / (1 + np.abs(s["qpos"]))).max(1))
            errs.append(e)
            aq.append(np.abs(qpos - s["qpos"]).max(1)); av.append(np.abs(qvel - s["qvel"]).max(1))
            sm = np.ones(n, bool); sm[bad] = False
            smooth.append(sm); qmax = max(qmax, float(np.abs(s["qpos"]).max()))
            per_check.append((k, len(bad)))
            if "Fall" in env_id and k > 0:
                # what distinguishes the outliers: contacts of robot geoms with the SIDE faces / edges of platform and wall boxes (an ant
                # hanging over a chasm's rim or leaning on a wall), against a random sample of the other envs at the same checkpoint
                def side(e):
                    c = oracle.contacts(cm, start["qpos"][e])
                    c = c[(c[:, 9] > 0) & ((c[:, 7] < 0) | (c[:, 8] < 0))] if len(c) else c
                    return int((np.abs(c[:, 6]) < 0.9).sum()) if len(c) else 0
                inl = prng.choice(np.setdiff1d(np.arange(n), bad), size=min(100, n - len(bad)), replace=False)
                side_stats.append((sum(side(int(e)) for e in bad[:100]), min(len(bad), 100), sum(side(int(e)) for e in inl), len(inl)))
            flags += int((done.cpu().numpy() != ref["done"]).sum()) + int((info["goal_index"].cpu().numpy() != ref["goal_idx"]).sum())
        oracle.step(cm, st, act.astype(np.float64), nthreads=16)
    e = np.concatenate(errs)
    aq, av, smooth = np.concatenate(aq), np.concatenate(av), np.concatenate(smooth)
    print(f"| {env_id} | {n} x {len(checks)} | {np.median(e):.1e} | {np.quantile(e, 0.99):.1e} | {np.quantile(e, 0.999):.1e} | {e.max():.1e} | "
          f"{np.quantile(aq, 0.999):.1e} | {aq[smooth].max():.1e} | {np.quantile(av, 0.999):.1e} | {av[smooth].max():.1e} | {qmax:.1f} | "
          f"{nout} ({100.0 * nout / len(e):.2f} %) | {nsens} | {flags} |", flush=True)
    if "Fall" in env_id:
        so, no, si, ni = (sum(x[i] for x in side_stats) for i in range(4))
        BY_CHECK.append((env_id, n, per_check, (so / max(no, 1), no, si / max(ni, 1), ni)))
    env.close()

if BY_CHECK:
    print()
    print("Fall family, outliers (envs > 1e-5) by checkpoint — the falling block spawns nested in its platform and is expelled within "
          "three env-steps ([ASSUME-14]):")
    print()
    print("| config | outliers at checkpoint k (after k rollout steps) |")
    print("|---|---|")
    for env_id, n, pc, _ in BY_CHECK:
        print(f"| {env_id} ({n} envs) | " + ", ".join(f"k={k}: {b} ({100.0 * b / n:.2f} %)" for k, b in pc) + " |")
    print()
    print("active contacts of robot geoms with SIDE faces / edges of maze boxes (platform rims, walls: |n_z| < 0.9), per env — outliers against a random sample of the other envs:")
    print()
    print("| config | outliers (envs) | others (envs) |")
    print("|---|---|---|")
    for env_id, n, pc, (mo, no, mi, ni) in BY_CHECK:
        print(f"| {env_id} | {mo:.2f} ({no}) | {mi:.2f} ({ni}) |")

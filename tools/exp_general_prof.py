"""Phase cycles of the general engine (instrumented library of tools/exp_general_prof.sh): mean over the sampled envs of ONE env-step.
    MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=<repo>/mujoco_maze_amd/csrc/exp_GENPROF.so python tools/exp_general_prof.py [envs]"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import sys, os
sys.path.insert(0, %(root)r)
import torch
import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd.maze_env import VecMazeEnv
from tests import user_robots
from tests.test_general_engine import SPIN_TASKS
which, n = sys.argv[1], int(sys.argv[2])
if which == "spin_ant": env = VecMazeEnv(mm.AntEnv, SPIN_TASKS["SpinUMaze"], num_envs=n, maze_size_scaling=4.0)
elif which == "spin_point": env = VecMazeEnv(mm.PointEnv, SPIN_TASKS["SpinUMaze"], num_envs=n, maze_size_scaling=4.0)
elif which == "biped_push": env = VecMazeEnv(user_robots.robot_classes()[0], T.DistRewardPush, num_envs=n, maze_size_scaling=4.0)
else: env = mm.make(which, num_envs=n, force_vec=True, engine="general")
lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
g = torch.Generator(device=env.device).manual_seed(0)
env.set_auto_reset(True); env.reset(seed=0)
for k in range(12):
    env.step(lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g))
torch.cuda.synchronize(); print("=== MARK", flush=True)
env.step(lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g))
torch.cuda.synchronize(); env.close()
"""


def main():
    n = sys.argv[1] if len(sys.argv) > 1 else "512"
    lib = os.path.join(ROOT, "mujoco_maze_amd", "csrc", "exp_GENPROF.so")
    envv = dict(os.environ, MZ_DEBUG="1", MZ_LIBMAZESTEP_EXPERIMENT=lib)
    for which in (os.environ.get("GENPROF_WHICH", "AntUMaze-v0,spin_ant,biped_push,AntPush-v0,PointUMaze-v0,spin_point,SwimmerUMaze-v0").split(",")):
        out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), which, n], capture_output=True, text=True, env=envv, timeout=900)
        text = out.stdout.split("=== MARK")[-1]
        rows = [ln for ln in text.splitlines() if ln.startswith("GENPROF")]
        if not rows:
            print(which, "no samples", out.stderr[-500:])
            continue
        keys = re.findall(r"([A-Za-z0-9_+/:\-]+(?: [a-z]+)?) (\d+)", rows[0].replace("|", ""))
        names = [k for k, _ in keys][1:]
        acc = [0.0] * len(names)
        for ln in rows:
            vals = [int(v) for _, v in re.findall(r"([A-Za-z0-9_+/:\-]+(?: [a-z]+)?) (\d+)", ln.replace("|", ""))][1:]
            for i, v in enumerate(vals):
                acc[i] += v / len(rows)
        print(f"{which}: {len(rows)} sampled envs, mean shader cycles of one env-step per phase")
        tot = acc[0]
        for nm, v in zip(names, acc):
            print("   %-24s %12.0f  %5.1f %%" % (nm, v, 100.0 * v / tot if tot else 0.0))


if __name__ == "__main__":
    main()

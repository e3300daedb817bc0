import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import mujoco_maze_amd as mm
for env_id in ("PointUMaze-v0", "AntUMaze-v0"):
    env = mm.make(env_id)
    env.reset()
    a = env.action_space.sample() if hasattr(env.action_space, "sample") else np.zeros(env.action_space.low.shape, np.float32)
    for _ in range(200): env.step(a)
    t0 = time.perf_counter()
    for _ in range(2000): o, r, d, i = env.step(a)
    dt = time.perf_counter() - t0
    print("%s single env (numpy in / out, one sync per step): %.0f steps/s, %.1f us per step; obs %s reward %.4f info %s" % (env_id, 2000 / dt, dt / 2000 * 1e6, o.shape, r, sorted(i)))
    env.close()

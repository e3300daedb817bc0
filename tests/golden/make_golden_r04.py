#!/usr/bin/env python3
"""Round-4 fixture: the reference's OWN verdicts on the fp32 observations the device sees.

reward.npz holds float64 observations and the reference's reward / termination on them.  The kernels observe fp32, so the
GPU test used to round the observations and recompute the expected values with this repository's Python mirror of
maze_task.py.  This script closes that loop: it rounds the very observations of reward.npz to fp32, hands float64(fp32) to
the REFERENCE's `task.reward()` / `task.termination()` / `MazeGoal.neighbor()` (maze_task.py:43-47,77-81,110-111,403-407,
592-604,646-658) and stores what they return:

  reward_f32.npz   <tag>__reward  float64 [400]   task.reward(obs)
                   <tag>__term    uint8   [400]   task.termination(obs)
                   <tag>__goal    int32   [400]   index of the first goal (list order) whose neighbor() holds on the slot the
                                                  task's own termination() looks at (obs[:dim], or obs[3:6] for the object-slot
                                                  tasks, whose termination is overridden), -1 if none

Run in the build container only (imports /root/reference through make_golden.py's in-memory gym stubs).  Data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference, runs its registration loop)


def main():
    src = np.load(os.path.join(HERE, "reward.npz"))
    out, seen = {}, set()
    for kw in G.REGISTERED:
        k = kw["kwargs"]
        cls = k["maze_task"]
        key = (cls.__name__, k["maze_size_scaling"])
        if key in seen:
            continue
        seen.add(key)
        tag = f"{cls.__name__}__{k['maze_size_scaling']}"
        task = cls(k["maze_size_scaling"])
        obs = src[f"{tag}__obs"].astype(np.float32).astype(np.float64)  # what a float32 device buffer holds, as the reference would read it
        object_slot = cls.termination is not G.maze_task.MazeTask.termination  # BlockCarry / Billiard families judge obs[3:6]
        goal = np.full(len(obs), -1, np.int32)
        for r, o in enumerate(obs):
            sl = o[3:6] if object_slot else o
            for gi, g in enumerate(task.goals):
                if g.neighbor(sl):
                    goal[r] = gi
                    break
        out[f"{tag}__reward"] = np.array([task.reward(o) for o in obs], dtype=np.float64)
        out[f"{tag}__term"] = np.array([task.termination(o) for o in obs], dtype=np.uint8)
        out[f"{tag}__goal"] = goal
        assert np.array_equal(out[f"{tag}__term"].astype(bool), goal >= 0), tag  # termination == "some goal is neighbour"
    np.savez_compressed(os.path.join(HERE, "reward_f32.npz"), **out)
    print("tags:", len(seen))


if __name__ == "__main__":
    main()

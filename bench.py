#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched MazeEnv.step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one MazeEnv.step for every env of the batch (Ant: 5 MuJoCo frames x RK4
= 20 forward-dynamics evaluations per env).  Workload: AntUMaze-v0, 4096 envs per GPU
(BASELINE configs[2], the configuration the metric is quoted on), synthetic inputs:
state from the reference's reset distribution, i.i.d. uniform actions in the action
box, auto-reset on termination / 1000-step truncation.  For N > 1 each rank owns its
own 4096 envs (weak scaling) and the only collective is the RCCL all-gather of the
packed [N_local, obs_dim + 2] record (obs | reward | done) that north_star names.

The measured regime does not depend on --warmup: SETTLE_STEPS untimed batch-steps always
run first (the ants land and stay on the ground: SURVEY §8d's "100 warm-up" protocol),
then the caller's --warmup, then the timed --steps.

`--gpus N` with N > 1 launches the N ranks itself when it is not already running under torchrun (WORLD_SIZE unset): it re-executes
this file under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU,
backend nccl (= RCCL); it refuses (exit code 2) when fewer than N GPUs are visible.  Under the driver's own torchrun the
ranks are already there and WORLD_SIZE decides.

Other BASELINE configs with the same script: --env Ant4Rooms-v0 (configs[3], with --gpus 8),
--env AntPush-v0 --envs 2048 (configs[4]), --env PointUMaze-v0 (configs[1]).

Prints ONE JSON line (rank 0) with the fields the driver expects plus
  roofline     : algorithmic HBM bytes per launch / average kernel duration (HIP events on
                 the launch stream, recorded inside the timed region)
  cpu_baseline : the CPU oracle (a restatement — MuJoCo itself is not available) timed on
                 this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ENV_ID = "AntUMaze-v0"
SETTLE_STEPS = 100             # untimed, before --warmup: the timed window is the settled regime whatever the caller's --warmup
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # AMD's MI355X spec sheet: FP64 vector = half the FP32 vector rate (the guide lists FP32 only)
N_SIMD = 256 * 4                 # 256 CUs x 4 SIMDs
PMC_PASS_TIMEOUT_S = 240         # each of the two live rocprofv3 --pmc child passes (about 20 s each on a fresh box)
SUSTAINED_STEPS = 1000           # the untimed-by-the-metric leg behind the timed window: the settled kernel over a long rollout
# tools/fetch_calib.hip under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on 1 GiB (past the 256 MiB Infinity Cache), profiles/r05/fetch_calibration.txt
FETCH_CALIBRATION = "FETCH_SIZE = 0.50 x bytes for 4 / 8 / 16-byte-per-lane and 192-byte-record data streams, 1.0 x per XCD for instruction fetch; WRITE_SIZE = 1.00 x (profiles/r05/fetch_calibration.txt)"


def algo_bytes_per_env_step(m):
    """SURVEY §8d counting: fp32 persistent state read + written once per env-step, action in, obs / reward / done out:
    reads (nq + nv + nv + nu + 1) words; writes (nq + nv + nv + 1 + obs_dim + 1) words + 1 byte.  The Point / Swimmer
    kernels keep no warm start (their RK4 stages re-solve from the previous stage): SURVEY counts the same record for them."""
    rd = (m.nq + 2 * m.nv + m.nu + 1) * 4
    wr = (m.nq + 2 * m.nv + 1 + m.obs_dim + 1) * 4 + 1
    return rd + wr


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def pmc_counters(env_id, n_envs):
    """Per-launch PMC averages of the step kernel of THIS workload from the newest committed summary
    `profiles/r*/pmc_<env id>_<envs>.csv` (collected by tools/profile.sh in separate `rocprofv3 --pmc ...` passes of the bench
    command with the same --env / --envs; counters cannot be collected from inside the process), or {} when the workload has
    no committed counters.  (`pmc_ant_step_kernel.csv` is the round-1/2 name of the default workload's file.)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{env_id}_{n_envs}.csv")))
    if not files and n_envs == ENVS_PER_GPU and env_id == ENV_ID:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_ant_step_kernel.csv")))
    if not files:
        return {}
    vals = {"_file": os.path.relpath(files[-1], ROOT)}
    for line in open(files[-1]).read().splitlines()[1:]:
        k, v, _ = line.split(",")
        vals[k] = float(v)
    return vals


def live_pmc_traffic(args):
    """HBM traffic of the step kernel measured WITH this run: when `rocprofv3` is on PATH, the same
    workload is stepped twice more in child processes under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, counters only — no tracing domain — as MI355X_MICROARCH.md prescribes: the two do not fit one pass), 100 settle +
    64 profiled launches each; returns ({"FETCH_SIZE": KB, "WRITE_SIZE": KB} per-launch averages over the launches after the
    settle steps, note) or (None, why).  The children run with --no-live-pmc / --no-cpu-baseline and print nothing we use."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals, settle, steps = {}, 100, 64
    tmp = tempfile.mkdtemp(prefix="mz_bench_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--env", args.env,
                   "--envs", str(args.envs), "--settle", str(settle), "--warmup", "0", "--steps", str(steps), "--no-cpu-baseline", "--no-live-pmc", "--sustained", "0"]
            if args.lanes:
                cmd += ["--lanes", str(args.lanes)]
            for kv in args.opt:
                cmd += ["--opt", kv]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=PMC_PASS_TIMEOUT_S)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-200:]}"
            per_dispatch = {}  # one value per launch: rows of one Dispatch_Id (per XCD / dimension, should rocprofv3 emit several) are summed
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "_step_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        d = int(row["Dispatch_Id"]) if row.get("Dispatch_Id") not in (None, "") else len(per_dispatch)
                        per_dispatch[d] = per_dispatch.get(d, 0.0) + float(row["Counter_Value"])
            timed = [per_dispatch[d] for d in sorted(per_dispatch)][settle:]
            if not timed:
                return None, f"rocprofv3 --pmc {counter}: no step-kernel rows in the counter file"
            vals[counter] = sum(timed) / len(timed)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return vals, f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this command ({settle} settle + {steps} profiled launches each), per-launch average; rocprofv3 reports KB, x 1024 here; calibration of the counters for this kernel's 4-byte-per-lane accesses: " + FETCH_CALIBRATION


def pmc_traffic(vals, corrected=True):
    """HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (rocprofv3 reports KB).  Calibrated on this hardware with streams of known
    size (tools/fetch_calib.hip, profiles/r05/fetch_calibration.txt): FETCH_SIZE reports HALF the bytes of a data stream at every
    access width the step kernels use — 4, 8 and 16 bytes per lane, and the Ant's 192-byte record pattern — i.e. MI355X_MICROARCH.md's
    gfx950 correction (x 2) applies to them too; WRITE_SIZE is exact; INSTRUCTION fetch is counted in full (128 KiB of straight-line
    code: 8.07 x its size per launch — once per XCD).  `corrected`: 2 x FETCH + WRITE, the guide's correction — an upper bound, because
    the code's share of FETCH (0.4-0.6 MB per launch of these kernels) is doubled with the data; raw: FETCH + WRITE, the lower bound."""
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return ((2.0 if corrected else 1.0) * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def valu_roofline(vals, kernel_ms, env_steps_per_s, env_id, f64=False):
    """The informative roofline for this latency / VALU-bound path: how busy the vector pipes are, how
    many of their lanes do work, and the measured floating-point work against the FP32 vector peak.
    SQ_* cycle counters tick once per 4 shader cycles and are summed over the SIMDs; SQ_THREAD_CYCLES_VALU counts active
    lanes in the same unit; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  flops/env-step: tools/count_flops.py (gcov
    execution counts of the float64 oracle x floating-point operators per line)."""
    out = {"bound": "valu", "source": vals.get("_file")}
    if "SQ_ACTIVE_INST_VALU" in vals and "SQ_INSTS_VALU" in vals:
        launch_cycles = vals["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in vals else (kernel_ms or 0.0) * 1e-3 * 2.4e9
        out.update({"valu_insts_per_launch": vals["SQ_INSTS_VALU"],
                    "valu_busy_frac": 4.0 * vals["SQ_ACTIVE_INST_VALU"] / (N_SIMD * launch_cycles) if launch_cycles else None,
                    "wait_frac_of_wave_life": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in vals and vals.get("SQ_WAVE_CYCLES") else None,
                    "note": "valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x launch cycles): share of the launch in which a SIMD executes a vector "
                            "instruction (tail included); active_lane_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes enabled per vector instruction"})
        if "SQ_THREAD_CYCLES_VALU" in vals and vals["SQ_ACTIVE_INST_VALU"]:
            out["active_lane_frac"] = vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"])
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"flops_{env_id}.json")))
    if files:
        fl = json.load(open(files[-1]))
        out["flops_per_env_step"] = fl["flops_per_env_step"]
        out["flops_source"] = os.path.relpath(files[-1], ROOT) + " (float64 oracle formulation, gcov-counted)"
        out["achieved"] = fl["flops_per_env_step"] * env_steps_per_s / 1e12
        out["peak"] = FP64_VECTOR_PEAK_TFLOPS if f64 else FP32_VECTOR_PEAK_TFLOPS
        out["peak_dtype"] = "f64 vector" if f64 else "f32 vector"
        out["unit"] = "TFLOP/s"
        out["frac"] = out["achieved"] / out["peak"]
    return out if len(out) > 2 else None


def cpu_baseline(model, env_id, n, lo, hi, seconds_target=12.0):
    """The CPU oracle's sources built -O3 with AVX2 / FMA code generation (oracle/libmzo_fast.so, kind 'port': a
    restatement — MuJoCo itself is not available) on the host cores, OpenMP over envs."""
    import numpy as np

    from tests import oracle_lib

    oracle = oracle_lib.load_fast()
    cores = usable_cores()
    nu = len(lo)
    st, _ = oracle.reset(model, n, 20260928)
    rng = np.random.default_rng(0)
    oracle.step(model, st, rng.uniform(lo, hi, (n, nu)), nthreads=cores)  # warm-up (page-in, first contacts)
    # settle first (the GPU leg is timed on settled ants too), untimed
    for _ in range(20):
        oracle.step(model, st, rng.uniform(lo, hi, (n, nu)), nthreads=cores)
    t0 = time.perf_counter()
    steps = 0
    while True:
        oracle.step(model, st, rng.uniform(lo, hi, (n, nu)), nthreads=cores)
        steps += 1
        if time.perf_counter() - t0 > seconds_target or steps >= 200:
            break
    dt = time.perf_counter() - t0
    # one core, a slice of the same batch (SURVEY 8d asks for both)
    n1 = min(512, n)
    st1 = {k: v[:n1].copy() for k, v in st.items()}
    t1 = time.perf_counter()
    steps1 = 0
    while time.perf_counter() - t1 < 4.0:
        oracle.step(model, st1, rng.uniform(lo, hi, (n1, nu)), nthreads=1)
        steps1 += 1
    dt1 = time.perf_counter() - t1
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "value_1core": n1 * steps1 / dt1,
            "sample": f"{env_id}, {n} envs x {steps} batch-steps ({dt:.1f} s) on {cores} cores + {n1} envs x {steps1} batch-steps on 1 core, "
                      "after 20 untimed settling steps; float64 CPU oracle (restatement, not mujoco-py) built -O3 -march=x86-64-v3 "
                      "(oracle/libmzo_fast.so; the strict-fp libmzo.so is the parity checker, not timed), OpenMP over envs"}


def _capi_path():
    from mujoco_maze_amd import _capi
    return _capi.LIB_PATH


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`--gpus N` outside torchrun: start the N ranks (one process per GPU) and pass their output through."""
    import subprocess
    rehearsal = os.environ.get("MZ_BENCH_SINGLE_GPU") == "1"
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        need = 1 if rehearsal else args.gpus
        if have < need:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {need} visible GPU(s), found {have}; refusing to report a "
                             f"{have}-GPU number as a {args.gpus}-GPU one (MZ_BENCH_SINGLE_GPU=1 rehearses the N > 1 path on one GPU over gloo)\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Launcher / process-group plumbing only (CPU test of the N > 1 launch path; no GPU work, value = null): the ranks
    rendezvous over gloo, time an empty K-step loop between the same barriers and rank 0 prints the line."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dt = 0.0
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
        t0 = time.perf_counter()
        dist.barrier()
        tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": f"env-steps/sec (whole node), {args.env}, {args.envs} envs/GPU", "value": None, "unit": "env-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "barrier_s": dt,
                          "config": {"workload": "launcher dry run: no environment was stepped"}}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)  # 100 warm-up + >= 1000 timed batch-steps
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--env", type=str, default=ENV_ID, help="registered env id (default: the metric's AntUMaze-v0)")
    ap.add_argument("--settle", type=int, default=SETTLE_STEPS, help="untimed steps before --warmup (regime independent of --warmup)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per env (8/16/32/64); 0 = library default")
    ap.add_argument("--wpb", type=int, default=0, help="wavefronts per workgroup (1/2/4); 0 = library default")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL obs all-gather for N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained", type=int, default=SUSTAINED_STEPS, help="steps of the extra, separately reported leg behind the timed window (0 = off)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc child passes for roofline.traffic (fall back to profiles/)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group plumbing only, no GPU work (CPU test of --gpus N)")
    ap.add_argument("--opt", action="append", default=[], help="extra library option key=value (tuning experiments)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is what runs and what n_gpus reports\n")
    if args.dry_run:
        sys.exit(dry_run(args))

    import torch

    import mujoco_maze_amd as mm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MZ_BENCH_SINGLE_GPU=1: rehearsal of the N > 1 code path on a one-GPU box (all ranks on cuda:0, gloo instead of
        # RCCL, which refuses two ranks per device).  Never set by the driver; the line it prints says so.
        rehearsal = os.environ.get("MZ_BENCH_SINGLE_GPU") == "1"
        if rehearsal:
            local = 0
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    n = args.envs
    env_id = args.env
    env = mm.make(env_id, num_envs=n, auto_reset=True, device=dev, force_vec=True)
    if args.lanes:
        env.set_option("lanes_per_env", args.lanes)
    if args.wpb:
        env.set_option("waves_per_block", args.wpb)
    for kv in args.opt:
        k, v = kv.split("=")
        env.set_option(k, float(v))
    from mujoco_maze_amd import sharding

    lo, _ = sharding.shard_range(rank, world, n)
    env.set_option("env_index_offset", float(lo))  # reset noise keyed by the global env slot
    env.reset(seed=20260928)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    a_lo = torch.as_tensor(env.action_space.low, device=dev)
    a_hi = torch.as_tensor(env.action_space.high, device=dev)
    pool = [(a_lo + (a_hi - a_lo) * torch.rand((n, env.nu), device=dev, generator=g)) for _ in range(32)]  # uniform in the action box
    gatherer = sharding.RecordGatherer(n, env.obs_dim, dev) if (world > 1 and not args.no_gather) else None

    def one_step(i, gather=True):
        # kernel -> collective: the step kernel writes the packed record (obs | reward | done) into the send buffer the
        # previous all-gather is NOT reading; that previous gather overlaps this step's kernel
        g = gatherer if gather else None
        if g is not None:
            env.bind_record(g.flip())
        env.step(pool[i % len(pool)])
        if g is not None:
            g.wait()       # previous step's gather (its receive buffer is reused)
            g.start()      # async RCCL all-gather of this step's record

    def timed(steps, gather=True):
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(i, gather)
        if gatherer is not None and gather:
            gatherer.wait()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    for i in range(args.settle):
        one_step(i)
    for i in range(args.warmup):
        one_step(args.settle + i)
    # HIP event pairs around the step kernel on its own stream, over the timed region — around every `stride`-th launch: a pair costs the
    # queue 4.6 us per launch (7.6 with default event flags; profiles/r06/event_overhead.txt), 10 % of a Point or Swimmer step, and the
    # instrument must not be what `value` measures.  kernel_ms = mean over the sampled launches (>= 5 of them even for --steps 20)
    def time_every(k_steps):
        stride = 8 if k_steps >= 64 else (4 if k_steps >= 20 else 1)
        env.set_option("time_kernels", (k_steps + stride - 1) // stride)
        env.set_option("time_kernels_stride", stride)
        return stride

    ev_stride = time_every(args.steps)
    dt = timed(args.steps)
    kernel_ms = env.kernel_ms()
    status = env.status()
    bad = int(((status & 3) != 0).sum().item())
    kernel_ms_ranks, no_gather = [kernel_ms], None
    # the sustained number: the driver's short window sits a few hundred launches into the rollout; the same
    # loop once more over SUSTAINED_STEPS launches, NOT part of `value` / `ms_per_step`
    sustained = None
    if world == 1 and args.sustained > 0:
        time_every(args.sustained)
        dt_s = timed(args.sustained)
        sustained = {"value": n * args.sustained / dt_s, "kernel_ms": env.kernel_ms(), "ms_per_step": dt_s / args.sustained * 1e3, "steps": args.sustained,
                     "note": "same loop, run after the timed window; not part of value"}
    if world > 1:
        km = torch.zeros(world, dtype=torch.float64, device=dev)
        km[rank] = kernel_ms
        dist.all_reduce(km)
        kernel_ms_ranks = [float(x) for x in km.tolist()]
        if gatherer is not None:  # the same K steps once more without the collective
            env.bind_record(None)
            dt_ng = timed(args.steps, gather=False)
            no_gather = {"value": n * world * args.steps / dt_ng, "ms_per_step": dt_ng / args.steps * 1e3}

    if rank == 0:
        total_env_steps = n * world * args.steps
        value = total_env_steps / dt
        per_env = algo_bytes_per_env_step(env.model.c)
        algo_bytes = per_env * n
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        pmc = pmc_counters(env_id, n)
        live, live_note = (None, "disabled (--no-live-pmc)") if (args.no_live_pmc or world > 1) else live_pmc_traffic(args)
        traffic = pmc_traffic(live) if live else pmc_traffic(pmc)
        traffic_source = live_note if live else (f"fallback ({live_note}): " + pmc.get("_file", "none committed for this workload") +
                                                 ": FETCH_SIZE + WRITE_SIZE (KB) of the committed rocprofv3 --pmc passes of this command, bytes per launch")
        robot = env.model.c.robot
        kernel = {1: "ant_step_kernel", 0: "planar_step_kernel", 2: "swimmer_step_kernel"}[robot]
        out = {
            "metric": f"env-steps/sec (whole node), {env_id}, {n} envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if robot == 1 else "f64", "data": "synthetic",
            "config": {"workload": f"{env_id}, {n} envs/GPU, frame_skip {env.model.c.frame_skip} x RK4, random actions uniform in the action box, auto-reset; "
                                   f"{args.settle} untimed settle steps + {args.warmup} warm-up steps before the timed window",
                       "envs_per_gpu": n, "obs_allgather": bool(world > 1 and not args.no_gather),
                       "library": os.path.relpath(_capi_path(), ROOT),
                       "lanes_per_env": args.lanes or "library default", "waves_per_block": args.wpb or 1, "bad_envs": bad,
                       "launch": env.launch_info(),  # the instantiation this batch size / device selected (mz_get_info)
                       **({"rehearsal": "all ranks on one GPU over gloo (MZ_BENCH_SINGLE_GPU=1): not a scaling measurement"}
                          if os.environ.get("MZ_BENCH_SINGLE_GPU") == "1" else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": kernel, "kernel_ms": kernel_ms, "kernel_ms_per_rank": kernel_ms_ranks,
                         "kernel_ms_source": "HIP event pairs on the kernel's stream around every %d-th launch of the timed region (mz_last_kernel_ms)" % ev_stride,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "algorithmic_bytes_per_env_step": per_env,
                         "traffic_source": traffic_source,
                         "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                         "traffic_lower": pmc_traffic(live or pmc, corrected=False) if traffic else None,
                         "traffic_definition": "v2 (since round 5): 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction; the lines of rounds 1-4 carried "
                                               "v1 = FETCH_SIZE + WRITE_SIZE, which is this line's traffic_lower — compare like with like",
                         "traffic_note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts data reads at half their bytes — calibrated for this "
                                         "kernel's access widths, tools/fetch_calib.hip — but instruction fetch in full: an upper bound); traffic_lower = FETCH_SIZE + WRITE_SIZE",
                         "traffic_fetch_bytes": ((live or pmc).get("FETCH_SIZE", 0.0) * 1024.0) if traffic else None,
                         "traffic_write_bytes": ((live or pmc).get("WRITE_SIZE", 0.0) * 1024.0) if traffic else None,
                         "note": "latency/VALU-bound path (SURVEY 8d): ~0.5 KB of HBM traffic per 20 forward-dynamics evaluations; HBM fraction reported because north_star asks for it"},
        }
        rv = valu_roofline(pmc, kernel_ms, n * args.steps / (kernel_ms * 1e-3 * args.steps) if kernel_ms > 0 else 0.0, env_id, f64=robot != 1)
        if rv is not None:
            out["roofline_valu"] = rv
        if sustained is not None:
            out["sustained"] = sustained
        if no_gather is not None:
            out["without_allgather"] = no_gather  # same ranks, same steps, collective off (with / without)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(env.model, env_id, n, env.action_space.low.astype("float64"), env.action_space.high.astype("float64"))
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

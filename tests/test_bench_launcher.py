"""`bench.py --gpus N` launches its own ranks (the flag used to be parsed and ignored).

CPU checks of the launch path: with no (or too few) GPUs visible the launcher refuses instead of reporting a smaller run under
the requested N; `--dry-run` drives the very same launcher (re-exec under torch.distributed.run, one process per rank,
rendezvous on 127.0.0.1, barrier-bracketed timing, rank 0 prints the single JSON line) over gloo without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MZ_BENCH_SINGLE_GPU"):
        e.pop(k, None)
    return e


def test_gpus_flag_refuses_when_too_few_devices_are_visible():
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = max(have, 1) + 1
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(want), "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=300, env=_env())
    assert out.returncode == 2, out.stdout + out.stderr
    assert "refusing" in out.stderr and f"--gpus {want}" in out.stderr
    assert out.stdout.strip() == ""  # no JSON line that could be mistaken for a measurement


def test_gpus_flag_launches_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                         timeout=600, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 3


def test_world_size_from_torchrun_wins_over_the_flag():
    """Under the driver's own torchrun the ranks already exist: no second launch, n_gpus = WORLD_SIZE."""
    e = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stdout + out.stderr
    assert json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1

#!/usr/bin/env python3
"""Measured floating-point operation count of one env-step ("refine with the oracle's op counter").

Method: the CPU oracle (oracle/mzo_physics.c + mzo_env.c, the float64 restatement of the same algorithm the kernels run)
is built with gcov instrumentation in a scratch directory and driven through a settled rollout of the bench workload;
every source line's EXECUTION COUNT (exact, from gcov) is multiplied by the number of floating-point operators on that
line (+ - * / on non-index expressions, sqrt / fabs / sin / cos / fmax / fmin calls as one operation each — a lexical
count, good to ~ +-20 %).  The result is written to profiles/<tag>/flops.json and quoted by bench.py next to the VALU
roofline.  Runs on the CPU (build container or GPU box); test infrastructure, never on the product path.

    python tools/count_flops.py [env id] [envs] [tag]
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FUNCS = r"\b(sqrt|fabs|sin|cos|fmax|fmin|hypot|pow|exp|log|floor|atan2)\s*\("


def line_flops(code: str) -> int:
    code = re.sub(r"/\*.*?\*/", "", code)
    code = re.sub(r"//.*", "", code)
    if re.match(r"\s*(#|for\s*\(\s*int\b[^;]*;[^;]*;[^)]*\)\s*$)", code):
        return 0
    code = re.sub(r"\[[^\[\]]*\]", "[]", code)        # index expressions are integer arithmetic
    code = re.sub(r"\[[^\[\]]*\]", "[]", code)
    code = re.sub(r"\bfor\s*\([^)]*\)", "", code)    # loop headers
    code = re.sub(r"\bint\b[^;]*;", "", code)         # integer declarations
    n = len(re.findall(FUNCS, code))
    code = code.replace("->", "").replace("++", "").replace("--", "").replace("+=", "+").replace("-=", "-").replace("*=", "*").replace("/=", "/")
    code = re.sub(r"(^|[=(,?:<>&|!]|return)\s*-", r"\1", code)  # unary minus
    code = re.sub(r"\b(const\s+)?(double|float|mz_model|mzo_data|mzo_contact|pairparam|void|int|char)\s*\*+", "", code)  # pointer declarators
    code = re.sub(r"\(\s*\*", "(", code)                # dereference
    n += len(re.findall(r"[+\-*/]", code))
    return n


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "AntUMaze-v0"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    work = tempfile.mkdtemp(prefix="mzo_gcov_")
    src = [os.path.join(ROOT, "oracle", f) for f in ("mzo_physics.c", "mzo_env.c")]
    lib = os.path.join(work, "libmzo_cov.so")
    subprocess.check_call(["gcc", "-O0", "--coverage", "-fPIC", "-std=gnu99", "-ffp-contract=off", f"-I{os.path.join(ROOT, 'include')}",
                           f"-I{os.path.join(ROOT, 'oracle')}", "-shared", "-o", lib] + src + ["-lm"], cwd=work)
    driver = f"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})
from export_mjcf import compile_for
from tests import oracle_lib
cm = compile_for({env_id!r})
ora = oracle_lib.Oracle(C.CDLL({os.path.join(ROOT, 'oracle', 'libmzo.so')!r}))   # settle with the regular build (uncounted)
st, _ = ora.reset(cm, {n}, 20260928)
rng = np.random.default_rng(0)
lo = np.array([cm.c.act_ctrlrange[a][0] for a in range(cm.c.nu)]); hi = np.array([cm.c.act_ctrlrange[a][1] for a in range(cm.c.nu)])
for _ in range(120): ora.step(cm, st, rng.uniform(lo, hi, ({n}, cm.c.nu)), nthreads=8)
cov = oracle_lib.Oracle(C.CDLL({lib!r}))
STEPS = 5
for _ in range(STEPS): cov.step(cm, st, rng.uniform(lo, hi, ({n}, cm.c.nu)), nthreads=1)
print("ENV_STEPS", {n} * STEPS)
"""
    out = subprocess.check_output([sys.executable, "-c", driver], cwd=work, text=True)
    env_steps = int(re.search(r"ENV_STEPS (\d+)", out).group(1))
    for gcda in glob.glob(os.path.join(work, "*.gcda")):  # libmzo_cov.so-<source>.gcda, written when the driver process exits
        subprocess.check_call(["gcov", "-o", work, os.path.basename(gcda)], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    total, per_file, per_func = 0, {}, {}
    for g in glob.glob(os.path.join(work, "*.gcov")):
        name = os.path.basename(g)[:-5]
        func = "?"
        for line in open(g, errors="replace"):
            m = re.match(r"\s*([0-9#=\-]+)\*?:\s*(\d+):(.*)", line)
            if not m:
                continue
            cnt, code = m.group(1), m.group(3)
            fm = re.match(r"(?:static\s+)?(?:inline\s+)?(?:void|int|double)\s+(\w+)\s*\(", code)
            if fm:
                func = fm.group(1)
            if not cnt.isdigit():
                continue
            f = int(cnt) * line_flops(code)
            total += f
            per_file[name] = per_file.get(name, 0) + f
            per_func[func] = per_func.get(func, 0) + f
    res = {"env_id": env_id, "env_steps_counted": env_steps, "flops_per_env_step": total / env_steps,
           "by_file": {k: v / env_steps for k, v in per_file.items()},
           "by_function_top": {k: v / env_steps for k, v in sorted(per_func.items(), key=lambda kv: -kv[1])[:14]},
           "method": "gcov line execution counts of the float64 CPU oracle x lexical count of floating-point operators per line (+-20 %); "
                     "settled rollout (120 steps), random actions in the action box"}
    os.makedirs(os.path.join(ROOT, "profiles", tag), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", tag, f"flops_{env_id}.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()

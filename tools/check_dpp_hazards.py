#!/usr/bin/env python3
"""Static check of the DPP read-after-write hazard on built gfx950 code.

Rule (CDNA ISA guide, "manually inserted wait states"; LLVM GCNHazardRecognizer DppVgprWaitStates = 2):
a VGPR written by a VALU instruction must not be read by a DPP instruction — any operand, including the tied
destination of v_fmac / the `old` value of v_mov_b32_dpp — in the next two issue slots (`s_nop N` fills N + 1).
The compiler keeps the rule for the instructions it selects; it cannot see inside the hand-written `asm` blocks of
csrc/ant_newton_rows.h (v_fmac_f32_dpp / v_mul_f32_dpp with `row_newbcast`), so those are checked here, on the
instructions that were actually emitted:

    tools/check_dpp_hazards.py file.s                 compiler output (hipcc -S; labels mark branch targets)
    tools/check_dpp_hazards.py --so libmazestep.so    the built library (llvm-objdump; branch targets from offsets)

Every path into a DPP instruction is examined: the fall-through one and, when the instruction sits within two slots
after a branch target, the path through each branch that jumps there.  Exit status 1 and one line per violation.
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
DPP_RE = re.compile(r"\b(row_newbcast|quad_perm|row_mirror|row_half_mirror|row_shr|row_shl|row_ror|row_bcast|wave_shr|wave_shl|wave_ror|wave_rol)\b")
VREG_RE = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG_RE.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def valu_writes(mn, ops):
    """VGPRs a VALU instruction writes (first operand, when it is a vector register)."""
    if not mn.startswith("v_") or mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    first = ops.split(",")[0] if ops else ""
    return vregs(first)


def slots_of(inst):
    """issue slots an instruction occupies, as a list of per-slot VGPR write sets"""
    _, mn, ops = inst
    if mn == "s_nop":
        return [set()] * (int(ops.strip() or "0", 0) + 1)
    return [valu_writes(mn, ops)]


def history_before(insts, i, depth=2):
    """write sets of the `depth` issue slots that precede instruction i on the straight-line path"""
    h = []
    j = i - 1
    while j >= 0 and len(h) < depth:
        h = slots_of(insts[j]) + h
        j -= 1
    return h[-depth:]


def check_stream(name, insts, jumps):
    """insts: list of (key, mnemonic, operand text); jumps: {index of a branch target: [indices of the branches that jump to it]}.
    Every path into a DPP instruction is examined: the fall-through one and, for an instruction within two slots after a branch
    target, the one through each jumping branch (the branch itself fills one slot)."""
    bad, ndpp = [], 0
    for i, (key, mn, ops) in enumerate(insts):
        if not (DPP_RE.search(ops) or mn.endswith("_dpp")):
            continue
        ndpp += 1
        reads = vregs(DPP_RE.split(ops)[0])
        paths = [history_before(insts, i)]
        # branch targets at i, i-1 (one slot in between) — collect the jumping paths
        acc = []
        for back in range(0, 2):
            t = i - back
            if t < 0:
                break
            if t in jumps:
                for b in jumps[t]:
                    pre = (history_before(insts, b + 1) + acc)[-2:]  # ..., the branch's own slot, then the slots between target and i
                    paths.append(pre)
            if back < 1 and t - 1 >= 0:
                acc = slots_of(insts[t - 1])[-2:] + acc if back == 0 else acc
            if len(acc) >= 2:
                break
        for h in paths:
            hit = set().union(*h) & reads if h else set()
            if hit:
                bad.append(f"{name}: {key}: {mn} {ops.strip()}  <- reads v{sorted(hit)} written less than two issue slots earlier")
                break
    return bad, ndpp


def parse_compiler_s(path):
    funcs, cur = [], None
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].rstrip()
        if re.match(r"^[A-Za-z_][\w$.]*:\s*$", t) and not t.startswith((".L", "__hip")):
            cur = {"name": t[:-1].strip(), "insts": [], "labels": {}, "branches": []}
            funcs.append(cur)
            continue
        if cur is None:
            continue
        lm = re.match(r"^(\.LBB\d+_\d+):", t)
        if lm:
            cur["labels"][lm.group(1)] = len(cur["insts"])
            continue
        t = t.strip()
        m = re.match(r"^((?:v|s|ds|global|buffer|flat|scratch)_\w+)\s*(.*)$", t)
        if m:
            mn = m.group(1) if m.group(1).endswith("_dpp") else re.sub(r"_e(32|64)$", "", m.group(1))
            if mn.startswith(("s_cbranch", "s_branch")):
                cur["branches"].append((len(cur["insts"]), m.group(2).strip()))
            cur["insts"].append((f"line {ln}", mn, m.group(2)))
    for f in funcs:
        f["jumps"] = {}
        for bi, lab in f["branches"]:
            if lab in f["labels"]:
                f["jumps"].setdefault(f["labels"][lab], []).append(bi)
    return funcs


def parse_objdump(text):
    funcs, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:\s*$", line)
        if m:
            cur = {"name": m.group(1), "insts": [], "addr": {}, "branches": []}
            funcs.append(cur)
            continue
        if cur is None or "//" not in line:
            continue
        body, comment = line.split("//", 1)
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        bm = re.match(r"^\s*((?:v|s|ds|global|buffer|flat|scratch)_\w+)\s*(.*)$", body.rstrip())
        if not am or not bm:
            continue
        addr = int(am.group(1), 16)
        mn, ops = bm.group(1), bm.group(2)
        mn = mn if mn.endswith("_dpp") else re.sub(r"_e(32|64)$", "", mn)
        if mn.startswith(("s_cbranch", "s_branch")):
            off = int(ops.strip().split()[0], 0)
            if off >= 0x8000:
                off -= 0x10000
            cur["branches"].append((len(cur["insts"]), addr + 4 + 4 * off))
        cur["addr"][addr] = len(cur["insts"])
        cur["insts"].append((hex(addr), mn, ops))
    for f in funcs:
        f["jumps"] = {}
        for bi, ta in f["branches"]:
            if ta in f["addr"]:
                f["jumps"].setdefault(f["addr"][ta], []).append(bi)
    return funcs


def disassemble_so(path):
    tmp = tempfile.mkdtemp(prefix="mzdpp_")
    local = os.path.join(tmp, os.path.basename(path))
    with open(path, "rb") as src, open(local, "wb") as dst:
        dst.write(src.read())
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
    assert cos, "no device code object found in " + path
    text = ""
    for co in cos:
        text += subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return text


def main(argv):
    if len(argv) >= 2 and argv[0] == "--so":
        funcs = parse_objdump(disassemble_so(argv[1]))
    else:
        funcs = parse_compiler_s(argv[0])
    total_bad, total_dpp = [], 0
    for f in funcs:
        bad, ndpp = check_stream(f["name"][:60], f["insts"], f["jumps"])
        total_bad += bad
        total_dpp += ndpp
    for b in total_bad[:50]:
        print(b)
    print(f"{len(funcs)} functions, {total_dpp} DPP instructions checked, {len(total_bad)} hazard violation(s)")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

#!/usr/bin/env python3
"""Developer aid (no GPU): loop structure of a kernel's ISA (tools/isa_point.sh, tools/isa_one.sh write /tmp/isa/*.s) — every
backward branch with the static instruction mix of its body, so that the cost of a Newton iteration or a line-search step can be
read off before spending a GPU call:  tools/isa_loops.py /tmp/isa/point.s"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
ins, labels = [], {}
for l in lines:
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not s or s[0] in ';.' or s.endswith(':'):
        continue
    ins.append(s)
print("total instructions", len(ins))
loops = []
for i, s in enumerate(ins):
    m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', s)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i, t))
P = {"valu": r'v_', "f64": r'v_\w+_f64', "ds": r'ds_', "salu": r's_', "wait": r's_waitcnt', "rcp/rsq/sqrt64": r'v_(rcp|rsq|sqrt)_f64',
     "readlane": r'v_readlane', "branch": r's_cbranch|s_branch', "saveexec": r's_\w+_saveexec', "nop": r's_nop', "vmem": r'global_|buffer_|scratch_'}
for a, b, t in sorted(loops):
    body = ins[a:b + 1]
    mix = {k: sum(1 for x in body if re.match(p, x)) for k, p in P.items()}
    mix["dpp"] = sum(1 for x in body if 'dpp' in x or 'row_' in x or 'quad_perm' in x)
    print(f"loop {t}: ins {a}-{b} len {b - a + 1}  " + "  ".join(f"{k} {v}" for k, v in mix.items() if v))

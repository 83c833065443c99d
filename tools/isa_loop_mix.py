#!/usr/bin/env python3
"""Instruction histogram of every loop (label .. backward branch to it) of one kernel in a hipcc -save-temps .s file, and
the issue order of the biggest one (MFMA runs collapsed).  Usage: python tools/isa_loop_mix.py file.s <mangled-substring> [--order]"""
import collections
import re
import sys

t = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*" + re.escape(sys.argv[2]) + r"\S*):.*?\n(.*?)\n\s*s_endpgm", t, re.S | re.M)
print(m.group(1))
body = m.group(2).split("\n")
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\S+:", l)}
loops = []
for i, l in enumerate(body):
    mm = re.search(r"s_cbranch_\S+\s+(\.LBB\S+)", l) or re.search(r"s_branch\s+(\.LBB\S+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))
for a, b in loops:
    c = collections.Counter()
    for l in body[a:b + 1]:
        w = l.strip().split()
        if w and not w[0].startswith((".", ";")):
            c[w[0]] += 1
    if c.get("v_mfma_f32_16x16x32_f16", 0) + c.get("v_mfma_f32_16x16x32_bf16", 0) == 0:
        continue
    print(f"loop lines {a}..{b}:", ", ".join(f"{v} {k}" for k, v in c.most_common(14)))
    if "--order" in sys.argv:
        run = 0
        for l in body[a:b + 1]:
            w = l.strip()
            if not w or w.startswith((";", ".")):
                continue
            if w.startswith("v_mfma"):
                run += 1
                continue
            if run:
                print(f"    ... {run} mfma")
                run = 0
            print("   ", w[:90])
        if run:
            print(f"    ... {run} mfma")

#!/usr/bin/env python3
"""Audit of loads issued from inline asm (global_load_dwordx4 into VGPRs the compiler does not track): for every such load
print the first later instruction that touches one of its destination registers and the hand-written `s_waitcnt vmcnt`
statements between the two.  A destination touched with no wait in between (a copy, a spill, an early use) would read
stale data.  Usage: python tools/isa_async_reg_check.py file.s <mangled-name-substring>"""
import re
import sys

t = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*" + re.escape(sys.argv[2]) + r"\S*):.*?\n(.*?)\n\s*s_endpgm", t, re.S | re.M)
print(m.group(1))
body = m.group(2).split("\n")


def regs(tok):
    mm = re.match(r"v\[(\d+):(\d+)\]", tok)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r"v(\d+)$", tok)
    return {int(mm.group(1))} if mm else set()


lines = []
in_asm = False
for i, l in enumerate(body):
    s = l.strip()
    if s.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if s.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not s or s.startswith((";", ".")):
        continue
    lines.append((i, s, in_asm))
ok = True
for k, (i, s, a) in enumerate(lines):
    if not (a and s.startswith("global_load_dwordx4")):
        continue
    toks = re.findall(r"v\[\d+:\d+\]|v\d+", s)
    dst = regs(toks[0])
    waits = []
    for (i2, s2, a2) in lines[k + 1:]:
        if a2 and s2.startswith("s_waitcnt vmcnt"):
            waits.append(s2.split()[1])
            continue
        if a2 and s2.startswith("global_load_dwordx4"):
            t2 = re.findall(r"v\[\d+:\d+\]|v\d+", s2)
            if regs(t2[0]) & dst:
                print(f"  line {i}: {toks[0]} overwritten by another asm load at {i2} before any use"); ok = False
                break
            continue                                   # (its address operand may be anything)
        used = set()
        for tk in re.findall(r"v\[\d+:\d+\]|v\d+", s2):
            used |= regs(tk)
        if used & dst:
            flag = "ok" if waits else "NO WAIT"
            if not waits:
                ok = False
            print(f"  line {i}: {toks[0]:12s} first touched at {i2}: {s2[:60]:60s} waits between: {waits} {flag}")
            break
print("all asm-loaded registers are first touched behind a hand-written wait" if ok else "PROBLEM")

#!/usr/bin/env python3
"""Order of the vector-memory instructions, waits and barriers of one kernel in a hipcc -save-temps .s file (to audit
hand-counted s_waitcnt vmcnt(N) values).  Usage: python tools/isa_vm_trace.py file.s <mangled-name-substring>"""
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\S*" + re.escape(key) + r"\S*):.*?\n(.*?)\n\s*s_endpgm", text, re.S | re.M)
print(m.group(1))
body = m.group(2).split("\n")
pat = re.compile(r"scratch_|s_waitcnt|global_load|global_store|buffer_|s_barrier|s_cbranch|^\.LBB|v_mfma")
run = 0
for i, l in enumerate(body):
    if pat.search(l):
        t = l.strip()
        if t.startswith("v_mfma"):
            run += 1
            continue
        if run:
            print(f"        ... {run} v_mfma")
            run = 0
        print(f"{i:6d}  {t[:100]}")

#!/usr/bin/env python3
"""Audit of loads issued from inline asm into VGPRs the compiler does not track (global_load_dword / _dwordx2 / _dwordx4
inside ;;#ASMSTART .. ;;#ASMEND): for every such load, walk every path the program can take from it -- straight on, and
around every backward branch met on the way -- and report the first instruction that touches one of its destination
registers together with the hand-written `s_waitcnt vmcnt` statements passed on that path.  A destination touched with no
hand-written wait in between (a copy, a spill, an early use) would read stale data; the compiler's own waits do not count,
it does not know about these loads.  LDS-DMA copies (global_load_lds_*) have no VGPR destination and are skipped.
kind="lds": the same audit for LDS reads issued from inline asm (ds_read_b*) and released by hand-counted `s_waitcnt lgkmcnt`
(conv3x3_c64_kernel's fragment ring).

Usage: python tools/isa_async_reg_check.py file.s <mangled-name-substring>      (exit code 1 on a finding)
       from tools.isa_async_reg_check import audit; ok, report, n_loads = audit(text, key)
tests/test_isa_audit.py runs it on the emitted ISA of the GEMM kernels that use such loads."""
import re
import sys

_REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)\b")
_LOAD = re.compile(r"global_load_dword(x2|x3|x4)?\s")
_LDS_LOAD = re.compile(r"ds_read_b(32|64|96|128)\s")      # kind="lds": asm-issued LDS reads, released by hand-counted lgkmcnt waits


def _regs(text):
    out = set()
    for a, b, c in _REG.findall(text):
        if c:
            out.add(int(c))
        else:
            out |= set(range(int(a), int(b) + 1))
    return out


def _dst(line):
    m = _REG.search(line)
    return _regs(m.group(0)) if m else set()


def audit(text, key, kind="vm"):
    load_re, counter = (_LOAD, "vmcnt") if kind == "vm" else (_LDS_LOAD, "lgkmcnt")
    m = re.search(r"^(_Z\S*" + re.escape(key) + r"\S*):.*?\n(.*?)\n\s*s_endpgm", text, re.S | re.M)
    if not m:
        raise KeyError(f"no kernel matching {key!r}")
    name, body = m.group(1), m.group(2).split("\n")
    lines, labels, in_asm = [], {}, False
    for raw in body:
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        lm = re.match(r"^(\.LBB\w+):", s)
        if lm:
            labels[lm.group(1)] = len(lines)
            continue
        if not s or s.startswith((";", ".")):
            continue
        lines.append((s.split(";")[0].strip(), in_asm))
    report, ok, n_loads = [name], True, 0
    for k, (s, a) in enumerate(lines):
        if not (a and load_re.match(s)):
            continue
        n_loads += 1
        dst = _dst(s)
        # paths: (next index, stop index or None, hand-written waits seen so far); a backward branch adds the path label..k
        todo, seen_labels, verdicts = [(k + 1, None, ())], set(), []
        while todo:
            i, stop, waits = todo.pop()
            end = stop if stop is not None else len(lines)
            while i < end:
                s2, a2 = lines[i]
                if a2 and s2.startswith("s_waitcnt") and counter in s2:
                    # a wait that names the registers as operands is itself the fence; it touches nothing
                    waits = waits + (re.search(counter + r"\(\d+\)", s2).group(0),)
                    i += 1
                    continue
                if a2 and load_re.match(s2):
                    if _dst(s2) & dst:
                        verdicts.append((bool(waits), f"overwritten by another asm load: {s2[:50]}", waits))
                        break
                    i += 1
                    continue                               # (its address operand may be anything)
                br = re.match(r"s_c?branch\S*\s+(\.LBB\w+)", s2)
                if br and br.group(1) in labels and labels[br.group(1)] <= k and br.group(1) not in seen_labels:
                    seen_labels.add(br.group(1))
                    todo.append((labels[br.group(1)], k, waits))
                if _regs(s2) & dst:
                    verdicts.append((bool(waits), s2[:60], waits))
                    break
                i += 1
        for good, what, waits in verdicts:
            if not good:
                ok = False
            report.append(f"  {s[:44]:44s} -> {what:60s} waits on the path: {list(waits)} {'ok' if good else 'NO WAIT'}")
    report.append("all asm-loaded registers are first touched behind a hand-written wait" if ok else "PROBLEM")
    return ok, "\n".join(report), n_loads


if __name__ == "__main__":
    good, rep, n = audit(open(sys.argv[1]).read(), sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "vm")
    print(rep)
    print(f"{n} asm-issued register loads audited")
    sys.exit(0 if good else 1)

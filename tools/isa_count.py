#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S listing (whole body, and the hottest loop = the
longest basic-block chain between a backward branch and its target).

    python tools/isa_count.py listing.s kernel_substring
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and sub in l:
            start = i
            break
    assert start is not None, "kernel not found"
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end") or "s_endpgm" in l and False:
            break
        body.append(l)
    labels, insts = {}, []
    for l in body:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        insts.append(s.split(";")[0].strip())
    tot = Counter(classify(i.split()[0]) for i in insts)
    print("whole kernel:", dict(tot), "total", len(insts))
    # loops: backward branches
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    for a, b in sorted(loops, key=lambda ab: ab[0] - ab[1])[:3]:
        c = Counter(classify(i.split()[0]) for i in insts[a:b + 1])
        print(f"loop [{a}:{b}] {b - a + 1} insts:", dict(c))


if __name__ == "__main__":
    main()

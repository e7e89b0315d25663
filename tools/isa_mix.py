#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only).

For every kernel: total and innermost-hot-loop (the longest backward-branch span) instruction counts by class.
A cheap stand-in for a profiler pass when deciding what a kernel's loop spends its issue slots on.
usage: tools/isa_mix.py file.s [kernel-name-substring]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    kernels, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if ln.strip().startswith(".end_amdhsa_kernel") or ln.strip().startswith(".section"):
            cur = None
        if cur is not None:
            cur.append(ln)
    for name, body in kernels.items():
        if want not in name or not any("s_endpgm" in l for l in body):
            continue
        labels, insts = {}, []
        for ln in body:
            s = ln.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            insts.append(s)
        loops = []
        for idx, s in enumerate(insts):
            m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", s)
            if m:
                tgt = labels.get(m.group(1) or m.group(2))
                if tgt is not None and tgt <= idx:
                    loops.append((idx - tgt, tgt, idx))
        tot = Counter(classify(s.split()[0]) for s in insts)
        print(f"== {name}\n   total {len(insts)}: {dict(tot)}")
        for span, a, b in sorted(loops, reverse=True)[:3]:
            c = Counter(classify(s.split()[0]) for s in insts[a:b + 1])
            print(f"   loop [{a}:{b}] {span} insts: {dict(c)}")


if __name__ == "__main__":
    main()

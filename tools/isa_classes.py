#!/usr/bin/env python
"""Count the instructions of an ISA listing (hipcc -S --cuda-device-only) by issue class, for a range of lines.

    python tools/isa_classes.py file.s FIRST LAST [FIRST LAST ...]

Classes as priced by tools/micro/valu_issue.hip (profiles/r05_valu_issue.txt): plain fp32 / integer VALU 1, packed (v_pk_*),
DPP, v_readlane / v_readfirstlane, compares writing an SGPR pair, v_cndmask with an SGPR mask 1.5, transcendentals
(v_rcp / v_exp / v_log / v_sqrt / v_rsq) and v_permlane32_swap 2.9; SALU, LDS, global memory and waits are listed beside them."""
import re
import sys

COST = {"valu_plain": 1.0, "valu_packed": 1.5, "valu_dpp": 1.5, "valu_lane": 1.5, "valu_cmp": 1.5, "valu_cndmask": 1.5, "valu_trans": 2.9}


def classify(op, line):
    if op.startswith("v_"):
        if "dpp" in op or " row_" in line or "quad_perm" in line:
            return "valu_dpp"
        if op.startswith("v_pk_"):
            return "valu_packed"
        if op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_permlane")):
            return "valu_trans"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith("v_cmp"):
            return "valu_cmp"
        if op.startswith("v_cndmask"):
            return "valu_cndmask"
        if op.startswith("v_mfma"):
            return "mfma"
        return "valu_plain"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def count(lines, a, b):
    out = {}
    for ln in lines[a - 1:b]:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or re.match(r"^[.\w$]+:", t):
            continue
        op = t.split()[0]
        c = classify(op, t)
        out[c] = out.get(c, 0) + 1
    return out


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().splitlines()
    r = [int(v) for v in sys.argv[2:]]
    for a, b in zip(r[::2], r[1::2]):
        c = count(lines, a, b)
        valu = sum(v for k, v in c.items() if k.startswith("valu_"))
        slots = sum(v * COST[k] for k, v in c.items() if k in COST)
        print(f"lines {a}-{b}: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())) + f" | VALU {valu}, plain-equivalent issue slots {slots:.1f}")

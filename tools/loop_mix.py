#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only csrc/stft1024.hip -o /tmp/k.s
    python tools/loop_mix.py /tmp/k.s <kernel-name-substring>

For every backward branch (label .. branch) prints the number of instructions by class; VALU x 4 cycles is the
issue time of one wave64 iteration on its SIMD.  Nested loops are reported separately (outer includes inner).
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem_st"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    labels = {}
    insts = []     # (line no, opcode, text)
    for i in range(start, end + 1):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not l or l.startswith((";", ".", "//")):
            continue
        op = l.split()[0]
        insts.append((i, op, l))
    print("%s: %d instructions" % (lines[start].split(":")[0], len(insts)))
    for j, (i, op, l) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = l.split()[1]
            if tgt in labels and labels[tgt] <= j:
                body = insts[labels[tgt]:j + 1]
                c = Counter(classify(o) for _, o, _ in body)
                valu = c["valu"] + c["valu_pk"] + c["valu_trans"]
                print("loop %-10s %5d instr | valu %4d (pk %d, trans %d) lds %3d ld %2d st %2d salu %3d wait %2d | "
                      "valu issue >= %d cycles" % (tgt, len(body), valu, c["valu_pk"], c["valu_trans"], c["lds"],
                                                   c["vmem_ld"], c["vmem_st"], c["salu"], c["waitcnt"], 4 * valu))
                if len(sys.argv) > 3 and sys.argv[3] == "-v":
                    ops = Counter(o for _, o, _ in body)
                    print("   ", ", ".join("%s:%d" % kv for kv in ops.most_common(40)))


if __name__ == "__main__":
    main()

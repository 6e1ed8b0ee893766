#!/usr/bin/env python3
"""Flags, in gfx950 assembly, a transcendental result read by the very next vector instruction.

The hardware needs one wait state between v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos and a non-transcendental vector
instruction reading their result.  The compiler's hazard recognizer pads that itself -- except inside an inline-asm statement,
which it does not look into (round 3: a v_mul right after v_exp_f32 inside one asm block produced garbage in the masked tile).
So the check is on the ISA: for every transcendental instruction, the next instruction of the same block must not be a vector
instruction naming its destination, unless an s_nop (or any other instruction) stands between them.

    python tools/isa_hazard_check.py file.s [...]      -> lists the offending pairs, exit status 1 if there are any
"""
import re
import sys

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path):
    bad = []
    kernel = "?"
    prev = None                     # (line number, text, destination registers) of a transcendental just seen
    for n, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            if not line.startswith("."):
                kernel = line[:-1]
            prev = None             # a label: the next instruction can be reached from elsewhere, the pad is the compiler's business
            continue
        if line.startswith("."):
            continue
        op, _, rest = line.partition(" ")
        if prev is not None:
            if op.startswith("v_") and not TRANS.match(op):
                operands = rest.split(",")
                # a destination that is overwritten is a hazard of its own kind as well; treat every naming as a read
                if prev[2] & regs(",".join(operands[1:]) if len(operands) > 1 else rest):
                    bad.append((kernel, prev[0], prev[1], n, line))
            prev = None
        if TRANS.match(op):
            prev = (n, line, regs(rest.split(",")[0]))
    return bad


def main(argv):
    total = 0
    for path in argv:
        for kernel, n0, a, n1, b in check(path):
            print(f"{path}:{n0}: in {kernel}: `{a}` is read by the next instruction `{b}` (line {n1}) without a wait state")
            total += 1
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

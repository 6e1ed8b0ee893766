#!/usr/bin/env python3
"""Where do a kernel's spilled VGPRs get stored / reloaded -- inside or outside its loops?

    python tools/isa_spill_report.py <file.s> [kernel-name-regex]

Reads the gfx950 assembly hipcc leaves with -save-temps (``*-hip-amdgcn-amd-amdhsa-gfx950.s``).  For every kernel whose
name matches: VGPRs / scratch bytes from the metadata comments, every basic block that belongs to a loop (LLVM annotates
them ``in Loop: Header=BBx_y Depth=d`` and the header ``Loop Header``) with its MFMA / LDS / global / scratch instruction
counts, and the list of scratch accesses with the block they sit in.  VERDICT r02 item 6 asked for exactly this: a
scratch reload in front of an MFMA waits on vmcnt, so spills inside the channel loop would cost; outside they do not.
"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m and "@" in line:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name:
            body.append(line.rstrip("\n"))
    if name:
        yield name, body


def report(name, body):
    out = []
    blocks = []          # [label, loop header or None, depth, instructions]
    cur = ["entry", None, 0, []]

    def annotate(text):
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", text)
        if m and not cur[3]:
            cur[1], cur[2] = m.group(1), int(m.group(2))
        m = re.search(r"Loop Header: Depth=(\d+)", text)
        if m and not cur[3]:
            cur[1], cur[2] = cur[0].lstrip(".").lstrip("L"), int(m.group(1))

    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", ln) or re.match(r"^; (%bb\.\d+):(.*)$", ln)
        if m:
            blocks.append(cur)
            cur = [m.group(1), None, 0, []]
            annotate(m.group(2))
            continue
        st = ln.strip()
        if st.startswith(";"):
            annotate(st)
            continue
        if st and not st.startswith("."):
            cur[3].append(st)
    blocks.append(cur)
    meta = {}
    for ln in body:
        for key in ("NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs"):
            m = re.match(rf"^; {key}: (\d+)", ln)
            if m:
                meta[key] = int(m.group(1))
    out.append(f"{name}\n  {meta}")
    loops = {}
    for label, hdr, depth, ins in blocks:
        if hdr is None:
            continue
        d = loops.setdefault(hdr, dict(blocks=0, ins=0, mfma=0, ds=0, glob=0, scratch=0, valu=0))
        d["blocks"] += 1
        d["ins"] += len(ins)
        d["mfma"] += sum(i.startswith("v_mfma") for i in ins)
        d["ds"] += sum(i.startswith("ds_") for i in ins)
        d["glob"] += sum(i.startswith(("global_", "flat_", "buffer_")) for i in ins)
        d["scratch"] += sum(i.startswith("scratch_") for i in ins)
        d["valu"] += sum(i.startswith("v_") and not i.startswith("v_mfma") for i in ins)
    for hdr, d in loops.items():
        out.append(f"  loop {hdr}: {d['blocks']} blocks, {d['ins']} instructions: {d['mfma']} mfma, {d['valu']} other vector, "
                   f"{d['ds']} lds, {d['glob']} global, {d['scratch']} scratch")
    n_in = n_out = 0
    for label, hdr, depth, ins in blocks:
        for i in ins:
            if i.startswith("scratch_"):
                where = f"inside loop {hdr}" if hdr else "outside every loop"
                n_in += hdr is not None
                n_out += hdr is None
                out.append(f"    {i.split(';')[0].strip():<48} in block {label:<10} {where}")
    out.append(f"  scratch accesses: {n_in} inside loops, {n_out} outside")
    return "\n".join(out)


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for name, body in kernels(path):
        if pat and not pat.search(name):
            continue
        print(report(name, body))
        print()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Speed-of-light table of one kernel from the per-dispatch counter files tools/sol_counters.sh leaves under gpurun_out/
(VERDICT r03 item 1b):  python tools/sol_table.py <tag> "<kernel name substring>" [waves per workgroup]

Counter units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4), SQ_VALU_MFMA_BUSY_CYCLES
and SQ_VALU_MFMA_COEXEC_CYCLES count cycles per SIMD (= 32 x the bf16 32x32x16 MFMAs), SQ_LDS_* count cycles per CU,
SQ_INSTS_* count wave-level instructions.  Everything is put against the lifetime of a wave (the kernels here are persistent: a
wave lives as long as the launch), i.e. as a fraction of the launch on each SIMD / CU."""
import glob
import os
import sys


def load(tag, kern):
    vals = {}
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    for path in sorted(glob.glob(os.path.join(root, f"sol_{tag}_*_dispatches.csv"))):
        rows = []
        for ln in open(path):                         # dispatch id, kernel name (may hold commas), grid size, NAME=value ...
            f = ln.rstrip("\n").split(",")
            first = next((i for i, c in enumerate(f) if "=" in c), len(f))
            if first < 3:
                continue
            name = ",".join(f[1:first - 1])
            if kern in name:
                rows.append((int(f[first - 1] or 0), f[first:]))
        if not rows:
            continue
        grid, cs = rows[-1]                           # the last dispatch: warm
        for c in cs:
            k, v = c.split("=")
            vals[k] = float(v)
        vals["_grid"] = grid
    return vals


def main():
    tag, kern = sys.argv[1], sys.argv[2]
    wpw = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    v = load(tag, kern)
    if not v:
        raise SystemExit(f"no dispatches of '{kern}' under gpurun_out/sol_{tag}_*")
    threads = v["_grid"]
    waves = threads / 64
    wgs = waves / wpw
    simds = wgs * 4 if wpw >= 4 else waves            # one workgroup per CU
    cus = wgs
    life = v["SQ_WAVE_CYCLES"] * 4 / waves            # cycles a wave lives = the launch, in shader cycles
    pct = lambda x: f"{100.0 * x / life:5.1f} %"
    out = []
    out.append(f"kernel {kern}   grid {int(threads)} threads = {int(wgs)} workgroups x {wpw} waves; a wave lives {life / 1e3:.1f} k cycles")
    mf = v.get("SQ_INSTS_MFMA", 0.0)
    rows = [
        ("matrix pipe busy (per SIMD)", v["SQ_VALU_MFMA_BUSY_CYCLES"] / simds),
        ("vector ALU busy incl. MFMA issue (per SIMD)", v["SQ_ACTIVE_INST_VALU"] * 4 / simds),
        ("  of it: MFMA issue, 8 cycles each", mf * 8 / simds),
        ("matrix and vector executing together (per SIMD)", v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / simds),
        ("LDS busy (per CU)", v.get("SQ_LDS_IDX_ACTIVE", 0.0) / cus),
        ("  of it: bank-conflict cycles", v.get("SQ_LDS_BANK_CONFLICT", 0.0) / cus),
        ("  LDS command FIFO full (per CU)", v.get("SQ_LDS_CMD_FIFO_FULL", 0.0) / cus),
        ("vector-memory instruction issue (per SIMD)", v.get("SQ_ACTIVE_INST_VMEM", 0.0) * 4 / simds),
        ("a wave: issuing", v.get("SQ_ACTIVE_INST_ANY", 0.0) * 4 / waves),
        ("a wave: stalled at issue (pipe taken by another wave, dependency)", v.get("SQ_WAIT_INST_ANY", 0.0) * 4 / waves),
        ("  of it: waiting to issue an LDS instruction", v.get("SQ_WAIT_INST_LDS", 0.0) * 4 / waves),
        ("a wave: parked at s_waitcnt / s_barrier", v.get("SQ_WAIT_ANY", 0.0) * 4 / waves),
    ]
    for name, cyc in rows:
        out.append(f"  {name:68s} {cyc / 1e3:8.1f} k cycles  {pct(cyc)}")
    iv = v.get("SQ_INSTS_VALU", 0.0)
    out.append(f"  instructions per wave: vector {iv / waves:.0f} (incl. {mf / waves:.0f} MFMA, {v.get('SQ_INSTS_VALU_TRANS_F32', 0) / waves:.0f} transcendental), "
               f"LDS {v.get('SQ_INSTS_LDS', 0) / waves:.0f}, scalar {v.get('SQ_INSTS_SALU', 0) / waves:.0f}, "
               f"vector-memory {(v.get('SQ_INSTS_VMEM_RD', 0) + v.get('SQ_INSTS_VMEM_WR', 0)) / waves:.0f};  "
               f"{v['SQ_ACTIVE_INST_VALU'] * 4 / max(iv, 1):.2f} busy cycles per vector instruction, "
               f"{(iv - mf) / max(mf, 1):.1f} vector instructions per MFMA")
    if "GRBM_GUI_ACTIVE" in v:
        out.append(f"  GRBM_GUI_ACTIVE / 8 = {v['GRBM_GUI_ACTIVE'] / 8 / 1e3:.1f} k cycles; FETCH_SIZE x 2 = {v.get('FETCH_SIZE', 0) * 2 / 1024:.1f} MB")
    print("\n".join(out))


if __name__ == "__main__":
    main()

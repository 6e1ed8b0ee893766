#!/usr/bin/env python3
"""Assembles profiles/<round>/pmc_summary.json from the per-kernel averages that tools/prof_pmc.sh wrote for three
separate rocprofv3 counter passes over `bench.py --steps 2 --warmup 1` (counters never share a run with traces):

    bash tools/prof_pmc.sh sq    SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \\
                                 SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
    bash tools/prof_pmc.sh fetch FETCH_SIZE GRBM_GUI_ACTIVE
    bash tools/prof_pmc.sh write WRITE_SIZE
    python tools/make_pmc_summary.py gpurun_out profiles/r01/pmc_summary.json

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so reads = 2 * FETCH_SIZE * 1024, writes = WRITE_SIZE * 1024.
`_conv3x3_all` is what bench.py prints as roofline.traffic: HBM bytes of every kernel a 3x3 convolution launches
(Winograd, direct 3x3, small-Cout, K-split reduction) per sisic_conv2d call (52 per step).
"""
import json
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) < 3:
            continue
        name = parts[0].replace("void ", "")
        n = int(parts[1].split()[1])
        vals = {k: float(v) for k, v in (kv.split("=") for kv in parts[2].split())}
        out[name] = (n, vals)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sq, fe, wr = (parse(f"{src}/{t}_counters.csv") for t in ("sq", "fetch", "write"))
    summary = {}
    conv_bytes = 0.0
    conv_calls = 0
    main_bytes = 0.0
    main_calls = 0
    bf3_bytes = 0.0
    bf3_calls = 0
    for name, (n, v) in sorted(sq.items()):
        if not name.startswith("sisic::"):
            continue
        e = {"dispatches": n}
        busy = v.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy:
            e["mfma_busy_frac_est"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy)   # four SIMDs per CU
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            e["wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["active_inst_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if v.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_per_lds_active"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
        rd = 2.0 * fe.get(name, (0, {}))[1].get("FETCH_SIZE", 0.0) * 1024
        wb = wr.get(name, (0, {}))[1].get("WRITE_SIZE", 0.0) * 1024
        e["hbm_read_MB_per_launch_corrected"] = rd / 1e6
        e["hbm_write_MB_per_launch"] = wb / 1e6
        summary[name] = e
        # the dominant kernel family of bench.py's roofline (profile slot "conv3x3_winograd_main"): the stride-1, non-upsampled
        # F(2x2,3x3) launches in their three geometries
        is_main = (re.search(r"conv_winograd_(wide|col)_kernel<\d+, \d+, \d, false>", name) or      # (not the image-pair form of the 8x8 level)
                   re.search(r"conv_winograd_kernel<1, 8, 8, \d, 16, false>", name))
        if is_main:
            main_bytes += (rd + wb) * n
            main_calls += n
        if "conv_winograd_bf3_kernel" in name:         # the bf16x3 form: profile slot "conv3x3_winograd_bf16x3", the dominant kernel
            bf3_bytes += (rd + wb) * n
            bf3_calls += n
        is_conv3 = ("conv_winograd_kernel" in name or "conv_winograd_wide_kernel" in name or "conv_winograd_col_kernel" in name or "conv_winograd_bf3_kernel" in name or re.search(r"conv_mfma_kernel<3,", name) or
                    "conv3x3_smallcout" in name or "splitk_reduce" in name)
        if is_conv3:
            conv_bytes += (rd + wb) * n
            if "splitk_reduce" not in name:       # the reduction belongs to its convolution's call
                conv_calls += n
    summary["_conv3x3_all"] = {"hbm_MB_per_launch": conv_bytes / max(conv_calls, 1) / 1e6, "launches": conv_calls,
                               "note": "HBM bytes of all kernels launched by 3x3 convolutions / sisic_conv2d calls"}
    summary["_winograd_main"] = {"hbm_MB_per_launch": main_bytes / max(main_calls, 1) / 1e6, "launches": main_calls,
                                 "note": "HBM bytes per launch of the stride-1 F(2x2,3x3) kernels (conv_winograd_col_kernel<128,16>, <64,8>; "
                                         "earlier rounds: conv_winograd_wide_kernel, conv_winograd_kernel<1,8,8,*,16,false>) = bench.py roofline.traffic"}
    summary["_winograd_bf16x3"] = {"hbm_MB_per_launch": bf3_bytes / max(bf3_calls, 1) / 1e6, "launches": bf3_calls,
                                   "note": "HBM bytes per launch of conv_winograd_bf3_kernel<PRO> = bench.py roofline.traffic when that "
                                           "kernel is the dominant one"}
    with open(dst, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps(summary["_conv3x3_all"]))


if __name__ == "__main__":
    main()

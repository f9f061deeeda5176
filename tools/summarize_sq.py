#!/usr/bin/env python
"""Summarise the SQ + GRBM PMC passes of tools/gpu_r03_prof.sh into profiles/<tag>/valu_busy.json: per kernel the VALU
instructions issued per SIMD and the share of the kernel's cycles they occupy AT THE MEASURED ISSUE RATES of
tools/valu_ubench.hip / tools/bank_ubench.hip (profiles/r03/valu_ubench.txt, bank_ubench.txt): with >= 2 waves per
SIMD a wave64 add / xor / shift / mov / cndmask issues every 1.03 ns, a min / max / cmp / 3-operand form every 1.8 ns
— 2.5 and 4.3 cycles of the ~2.4 GHz clock that GRBM_GUI_ACTIVE / kernel time shows (the "2.1 / 3.7" quoted elsewhere
are the same times at a nominal 2.0 GHz); the VGPR banks of the sources make no difference.

    cycles       = GRBM_GUI_ACTIVE / 8          (the counter is summed over the 8 XCDs)
    insts / SIMD = SQ_INSTS_VALU / 1024         (256 CUs x 4 SIMDs)
    busy_lo / hi = insts / SIMD x 2.5 (4.3) / cycles

busy_hi >= 1 with busy_lo around 0.6 means the kernel cannot run much faster without issuing fewer VALU instructions,
whatever its memory traffic.  usage: tools/summarize_sq.py gpurun_out/prof_<tag> profiles/<tag>
"""
import collections
import csv
import glob
import json
import os
import sys

NSIMD, NXCD, FAST, SLOW = 1024, 8, 2.5, 4.3


def per_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main(src, dst):
    sq = per_kernel(os.path.join(src, "sq", "**", "*counter_collection.csv"))
    gr = per_kernel(os.path.join(src, "grbm", "**", "*counter_collection.csv"))
    out = {}
    for k, d in sorted(sq.items()):
        if k.startswith("__amd") or k not in gr:
            continue
        mean = lambda n: d[n][1] / max(d[n][0], 1) if n in d else 0.0  # noqa: E731
        cycles = gr[k]["GRBM_GUI_ACTIVE"][1] / gr[k]["GRBM_GUI_ACTIVE"][0] / NXCD
        valu = mean("SQ_INSTS_VALU")
        per_simd = valu / NSIMD
        out[k] = {
            "launches": d["SQ_INSTS_VALU"][0], "waves": mean("SQ_WAVES"), "valu_insts": valu, "valu_insts_per_simd": per_simd,
            "salu_insts": mean("SQ_INSTS_SALU"), "lds_insts": mean("SQ_INSTS_LDS"), "vmem_rd_insts": mean("SQ_INSTS_VMEM_RD"),
            "cycles": cycles, "busy_lo": per_simd * FAST / cycles, "busy_hi": per_simd * SLOW / cycles,
            "wait_inst_share_of_wave_cycles": mean("SQ_WAIT_INST_ANY") / max(mean("SQ_WAVE_CYCLES"), 1.0),
        }
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, "valu_busy.json"), "w"), indent=1, sort_keys=True)
    print("kernel                                   launches   VALU/SIMD     cycles  busy@2.5  busy@4.3")
    for k, v in out.items():
        print(f"{k[:40]:40s} {v['launches']:8d} {v['valu_insts_per_simd']:11.0f} {v['cycles']:10.0f} {v['busy_lo']:9.2f} {v['busy_hi']:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

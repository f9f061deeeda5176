#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in separate runs by tools/profile_bench.sh)
into HBM bytes per launch per kernel -> profiles/<tag>/pmc_hbm_traffic.json.

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read, so it is doubled for the streaming
kernels (calibration on this code: k_threshold_count reads x (1.514 GB) + the fp64 table (3.028 GB) = 4.542 GB and
FETCH_SIZE*1024*2 = 4.542 GB; k_run_max_fused reads 1.514 GB, FETCH_SIZE*1024*2 = 1.514 GB; k_fill_synthetic writes
1.514 GB and WRITE_SIZE*1024 = 1.514 GB, i.e. writes need no correction).  The multi-year gather of k_pdoy_quad (256 bytes
per row and wave from rows 365 rows apart) is calibrated by tools/gather_ubench.hip, the same pattern without
arithmetic: it reads 45.41 GB and FETCH_SIZE*1024*2 = 45.41 - 45.50 GB for every variant (tools/experiments/r04/gpu_r04_p7.sh,
profiles/r04/pdoy_anatomy.txt #9).  The strided-gather select kernels and k_pdoy_top16 are NOT calibrated: their derived
byte figures are written as null (the raw counter means stay).

usage: tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<tag> [out.json [kernel-prefix,kernel-prefix,...]]
(the optional prefix list keeps only kernels that run at ONE grid size in that profile run: a mean over launches of
different sizes says nothing)
"""
import collections
import csv
import glob
import json
import os
import sys

UNCALIBRATED = ("k_select_grp", "k_select_lean", "k_select_tm", "__amd_rocclr", "k_pdoy_top16", "k_plane_linear")  # strided gathers: FETCH_SIZE x2 not calibrated
# (round 5: k_select_regsort / k_qdm_regsort ARE calibrated — tools/regsort_ubench.hip reads 1.518 GB by construction in their
#  load pattern and FETCH_SIZE x 1024 x 2 = 1.518 GB, profiles/r05/fetch_calibration_regsort.txt)


def main(src, dst, name="pmc_hbm_traffic.json", only=None, largest=False):
    """largest: a run that launches a kernel at several grid sizes (round 5: the 30-year configurations AND the c5 slab AND the
    365-step grid in one bench run) — keep, per kernel, only the launches of the largest problem (the dispatches of the FETCH
    pass whose value is within 10 % of the kernel's largest one; the same dispatch positions in the WRITE pass)."""
    out = {}
    keep = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
        agg = collections.defaultdict(lambda: [0, 0.0])
        per = collections.defaultdict(list)
        for f in files:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname:
                    continue
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if only and not k.startswith(only):
                    continue
                per[k].append(float(r["Counter_Value"]))
        for k, vals in per.items():
            if largest:
                if cname == "FETCH_SIZE":
                    keep[k] = [i for i, v in enumerate(vals) if v >= 0.9 * max(vals)]
                vals = [vals[i] for i in keep.get(k, range(len(vals))) if i < len(vals)]
            agg[k][0] = len(vals)
            agg[k][1] = sum(vals)
        for k, (n, v) in agg.items():
            out.setdefault(k, {})[cname + "_KiB_mean"] = v / n
            out[k]["launches"] = n
    for k, d in out.items():
        w = d.get("WRITE_SIZE_KiB_mean", 0.0) * 1024
        d["hbm_write_bytes_per_launch"] = w
        if k.startswith(UNCALIBRATED):
            # no figure that is known to be wrong: the raw counter stays, the derived bytes do not
            d["fetch_x2_correction"] = None
            d["hbm_read_bytes_per_launch"] = None
            d["hbm_bytes_per_launch"] = None
            d["reason"] = ("FETCH_SIZE is not calibrated for this kernel's strided 256-byte gathers (neither x1 nor x2 "
                           "reproduces its known input bytes)")
            continue
        f = d.get("FETCH_SIZE_KiB_mean", 0.0) * 1024 * 2
        d["fetch_x2_correction"] = True
        d["hbm_read_bytes_per_launch"] = f
        d["hbm_bytes_per_launch"] = f + w
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, name), "w"), indent=1, sort_keys=True)
    for k, d in sorted(out.items()):
        rd = "   n/a " if d["hbm_read_bytes_per_launch"] is None else f"{d['hbm_read_bytes_per_launch'] / 1e9:7.3f}"
        print(f"{k:40s} read {rd} GB  write {d['hbm_write_bytes_per_launch'] / 1e9:7.3f} GB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4] or ["pmc_hbm_traffic.json"]),
         only=tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 and sys.argv[4] else None, largest=len(sys.argv) > 5 and sys.argv[5] == "largest")

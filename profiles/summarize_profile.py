#!/usr/bin/env python3
"""Condense the rocprofv3 output directories collect_profile.sh wrote into the files kept under profiles/:
kernel_stats_1M.csv (the --stats table) and pmc_hbm_1M.json (HBM bytes per launch / per base and kernel).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KB, and FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads, so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.
"""
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "fpl::"):
        name = name.replace(pre, "")
    base = name.split("<")[0]
    if base == "k_stats" and "true" in name:
        return "k_stats_extra"
    return base


def main(out):
    stats = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[-1], os.path.join(out, "kernel_stats_1M.csv"))
    bench = {}
    try:
        bench = json.loads(open(os.path.join(out, "bench_1M.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    n_bases = bench.get("config", {}).get("bases_per_gpu")
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
        acc = {}
        for f in files:
            # one row per (dispatch, counter); a counter split over several rows of a dispatch sums up
            disp = {}
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != ctr:
                    continue
                k = (row["Dispatch_Id"], short(row["Kernel_Name"]))
                disp[k] = disp.get(k, 0.0) + float(row["Counter_Value"])
            for (_, kn), v in disp.items():
                acc.setdefault(kn, []).append(v)
        for kn, vals in acc.items():
            per.setdefault(kn, {})[ctr + "_KB"] = sum(vals) / len(vals)
            per[kn]["launches_" + ctr] = len(vals)
    kernels = {}
    for kn, d in per.items():
        if not kn.startswith("k_"):
            continue
        b = 2 * d.get("FETCH_SIZE_KB", 0.0) * 1024 + d.get("WRITE_SIZE_KB", 0.0) * 1024
        d["hbm_bytes_per_launch"] = b
        if n_bases:
            d["hbm_bytes_per_base"] = b / n_bases
        kernels[kn] = d
    rec = {
        "workload": "bench.py defaults (%s)" % bench.get("config", {}).get("workload", "?")[:60],
        "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB), averaged over launches. gfx950: "
                "FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM), "
                "so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
        "bases_per_launch": n_bases,
        "kernels": kernels,
    }
    json.dump(rec, open(os.path.join(out, "pmc_hbm_1M.json"), "w"), indent=1)
    print(json.dumps({k: round(v.get("hbm_bytes_per_base", 0), 3) for k, v in kernels.items()}))


if __name__ == "__main__":
    main(sys.argv[1])

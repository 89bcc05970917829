#!/usr/bin/env python3
"""Condense the rocprofv3 output directories collect_profile.sh wrote into the files kept under profiles/<tag>/:
kernel_stats_<workload>.csv (the --stats table) and pmc_<workload>.json (per kernel: HBM bytes and SQ instruction counts per
launch / per base).  `--merge <dir>` folds every pmc_<workload>.json of a directory into profiles/kernel_counters.json, the
file bench.py reads its `roofline.traffic`, `roofline.path` and `roofline.issue` figures from.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KB, and FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads, so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.
"""
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SQ = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES")
LDS = ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS")


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "fpl::"):
        name = name.replace(pre, "")
    base = name.split("<")[0]
    if base == "k_stats" and "true" in name:
        return "k_stats_extra"
    return base


def per_kernel(pattern, counters):
    """{kernel: {counter: mean over launches}} from the counter_collection.csv files under `pattern`"""
    acc = {}
    for f in glob.glob(pattern, recursive=True):
        disp = {}
        for row in csv.DictReader(open(f)):
            c = row.get("Counter_Name")
            if c not in counters:
                continue
            k = (row["Dispatch_Id"], short(row["Kernel_Name"]), c)
            disp[k] = disp.get(k, 0.0) + float(row["Counter_Value"])  # a counter split over several rows of a dispatch sums up
        for (_, kn, c), v in disp.items():
            acc.setdefault(kn, {}).setdefault(c, []).append(v)
    return {kn: {c: sum(v) / len(v) for c, v in d.items()} for kn, d in acc.items()}


def timed_stats(trace_csv, n_timed, dst):
    """per kernel over its LAST n_timed dispatches (the timed steps of the bench command; the warm-up launches come first):
    calls, average / min / max duration -- the rows of rocprofv3's --stats table without the warm-up launches"""
    per = {}
    for row in csv.DictReader(open(trace_csv)):
        try:
            per.setdefault(row["Kernel_Name"], []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        except (KeyError, ValueError):
            continue
    rows = []
    for name, ts in per.items():
        ts.sort()
        launches_per_step = max(1, len(ts) // (n_timed + 4))  # (a kernel launched several times per step keeps all of a step's launches)
        d = [e - s for s, e in ts[-n_timed * launches_per_step:]]
        rows.append((sum(d), name, len(d), sum(d) / len(d), min(d), max(d)))
    rows.sort(reverse=True)
    with open(dst, "w") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "TimedCalls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for tot, name, n, avg, lo, hi in rows:
            w.writerow([name, n, tot, round(avg, 1), lo, hi])


def main(out, wl, n_timed=0):
    work = os.path.join(out, "work_" + wl)
    stats = sorted(glob.glob(os.path.join(work, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[-1], os.path.join(out, "kernel_stats_%s.csv" % wl))
    trace = sorted(glob.glob(os.path.join(work, "stats", "**", "*kernel_trace.csv"), recursive=True))
    if trace and n_timed > 0:
        try:
            timed_stats(trace[-1], n_timed, os.path.join(out, "kernel_stats_timed_%s.csv" % wl))
        except (OSError, ValueError, ZeroDivisionError) as e:
            print("timed stats failed:", e)
    bench = {}
    try:
        bench = json.loads(open(os.path.join(out, "bench_%s.json" % wl)).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    n_bases = bench.get("config", {}).get("bases_per_gpu")
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for kn, d in per_kernel(os.path.join(work, "pmc_" + ctr, "**", "*counter_collection.csv"), (ctr,)).items():
            per.setdefault(kn, {})[ctr + "_KB"] = d[ctr]
    for kn, d in per_kernel(os.path.join(work, "pmc_SQ", "**", "*counter_collection.csv"), SQ).items():
        per.setdefault(kn, {}).update(d)
    for kn, d in per_kernel(os.path.join(work, "pmc_LDS", "**", "*counter_collection.csv"), LDS).items():
        per.setdefault(kn, {}).update(d)
    kernels = {}
    for kn, d in per.items():
        if not kn.startswith("k_"):
            continue
        b = 2 * d.get("FETCH_SIZE_KB", 0.0) * 1024 + d.get("WRITE_SIZE_KB", 0.0) * 1024
        d["hbm_bytes_per_launch"] = b
        if n_bases:
            d["hbm_bytes_per_base"] = b / n_bases
            if "SQ_INSTS_VALU" in d:
                d["valu_insts_per_base"] = d["SQ_INSTS_VALU"] / n_bases
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        kernels[kn] = d
    rec = {
        "workload": wl,
        "bench_line": bench.get("config", {}).get("workload", "?"),
        "note": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_*; SQ_LDS_*), averaged over launches. "
                "lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles). gfx950: "
                "FETCH_SIZE / WRITE_SIZE are in KB and FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads "
                "(MI355X_MICROARCH.md, HBM), so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; SQ_INSTS_* are wave-instructions",
        "bases_per_launch": n_bases,
        "kernels": kernels,
    }
    json.dump(rec, open(os.path.join(out, "pmc_%s.json" % wl), "w"), indent=1)
    print(json.dumps({k: [round(v.get("hbm_bytes_per_base", 0), 3), round(v.get("valu_insts_per_base", 0), 4)] for k, v in kernels.items()}))


def merge(d):
    path = os.path.join(HERE, "kernel_counters.json")
    try:
        cur = json.load(open(path))
    except (OSError, ValueError):
        cur = {"workloads": {}}
    for f in sorted(glob.glob(os.path.join(d, "pmc_*.json"))):
        rec = json.load(open(f))
        wl = rec.get("workload")
        if not wl or wl not in os.path.basename(f):
            continue
        ks = rec["kernels"]
        cur["workloads"][wl] = {
            "source": os.path.relpath(f, os.path.dirname(HERE)),
            "bases_per_launch": rec.get("bases_per_launch"),
            "hbm_bytes_per_base": {k: v["hbm_bytes_per_base"] for k, v in ks.items() if "hbm_bytes_per_base" in v},
            "valu_insts_per_base": {k: v["valu_insts_per_base"] for k, v in ks.items() if "valu_insts_per_base" in v},
        }
    json.dump(cur, open(path, "w"), indent=1)
    print("merged into", path, sorted(cur["workloads"]))


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "c3_full_pipeline", int(sys.argv[3]) if len(sys.argv) > 3 else 0)

#!/usr/bin/env python3
"""Kernel statistics of a rocprofv3 run from its rocpd sqlite database (the default output format of this
ROCm): per kernel name -- calls, total, average, min, max duration, share; written as CSV.
    python scripts/rocpd_stats.py gpurun_out/<dir> profiles/<name>.csv
"""
import csv
import glob
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    dbs = glob.glob(src + "/**/*.db", recursive=True) if not src.endswith(".db") else [src]
    rows = {}
    for db in dbs:
        con = sqlite3.connect(db)
        for name, dur, scratch, vgpr, agpr, lds in con.execute("select name, duration, scratch_size, vgpr_count, accum_vgpr_count, lds_size from kernels"):
            r = rows.setdefault(name, {"calls": 0, "total": 0, "min": 1 << 62, "max": 0, "scratch": scratch, "vgpr": vgpr, "agpr": agpr, "lds": lds})
            r["calls"] += 1; r["total"] += dur; r["min"] = min(r["min"], dur); r["max"] = max(r["max"], dur)
    tot = sum(r["total"] for r in rows.values()) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "ScratchBytesPerLane", "VGPR", "AGPR", "LDSBytes"])
        for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["total"]):
            w.writerow([name, r["calls"], r["total"], "%.1f" % (r["total"] / r["calls"]), "%.2f" % (100.0 * r["total"] / tot), r["min"], r["max"],
                        r["scratch"], r["vgpr"], r["agpr"], r["lds"]])
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["total"])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
        print("%-100s %5d %9.2f ms %8.3f avg %5.1f%% scr %s vgpr %s+%s" % (name[:100], r["calls"], r["total"] / 1e6, r["total"] / r["calls"] / 1e6, 100.0 * r["total"] / tot, r["scratch"], r["vgpr"], r["agpr"]))


if __name__ == "__main__":
    main()

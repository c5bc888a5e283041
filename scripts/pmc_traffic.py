#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` into
profiles/r01_pmc_traffic.json for the dominant kernel (jk_incore_kernel, 12-wave dimer variant).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request on wide
coalesced streams -> doubled; both counters are in KiB.

    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write
"""
import collections
import csv
import glob
import json
import os
import sys


def per_launch(directory, counter, kernel_substr):
    f = (glob.glob(os.path.join(directory, "*counter_collection.csv")) + glob.glob(os.path.join(directory, "*", "*counter_collection.csv")))[0]
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return tot, len(disp)


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    kern = "jk_incore_kernel<1, true, 12, 19, true, true>"      # the 12-wave tuned dimer variant
    fetch_kib, nf = per_launch(fetch_dir, "FETCH_SIZE", kern)
    write_kib, nw = per_launch(write_dir, "WRITE_SIZE", kern)
    out = {
        "kernel": kern, "launches": nf,
        "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
        "correction": "FETCH_SIZE x2 on gfx950 (64 B tallied per 128-B request); KiB -> bytes x1024",
        "hbm_bytes_total": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
    }
    out["hbm_bytes_per_launch"] = out["hbm_bytes_total"] / max(nf, 1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r01_pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

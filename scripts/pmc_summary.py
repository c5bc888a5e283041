#!/usr/bin/env python3
"""Reduces separate rocprofv3 --pmc passes (CSV output) of bench.py into profiles/<round>_pmc_summary.json and a
per-kernel CSV per pass (MQC_PROFILE_ROUND=r03 by default).  One pass per counter group, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a
pass; never together with trace domains).

    python scripts/pmc_summary.py <tag>=<dir> [...]       tags: fetch write sq mfma_b3lyp ...

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-B request on wide coalesced streams
-> doubled; FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("MQC_PROFILE_ROUND", "r03")


def read_pass(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    table = collections.defaultdict(lambda: collections.defaultdict(float))     # kernel -> counter -> sum
    launches = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            table[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    return table, {k: len(v) for k, v in launches.items()}


def read_durations(directory):
    """kernel name -> summed duration in seconds, from the kernel trace of the same pass"""
    out = collections.defaultdict(float)
    for f in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    return out


N_SIMD = 1024            # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9         # nominal; the chip runs lower under load (MI355X_MICROARCH.md, DVFS): fractions below are lower bounds


def short(name):
    return name.replace("void mqc::", "").replace("mqc::", "").split("(")[0]


def main():
    out = {"passes": {}}
    for arg in sys.argv[1:]:
        tag, directory = arg.split("=", 1)
        table, launches = read_pass(directory)
        counters = sorted({c for v in table.values() for c in v})
        rows = sorted(table.items(), key=lambda kv: -sum(kv[1].values()))
        with open(os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (ROUND, tag)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Launches"] + counters)
            for k, v in rows:
                w.writerow([k, launches[k]] + ["%.6g" % v.get(c, 0.0) for c in counters])
        dur = read_durations(directory)
        kern = {}
        for k, v in rows[:40]:
            e = dict(v, launches=launches[k], seconds_in_this_pass=dur.get(k, 0.0))
            if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) > 0 and dur.get(k, 0.0) > 0:
                # MFMA-busy SIMD-cycles over the SIMD-cycles the kernel had; MOPS_F64 counts 512-flop units
                e["mfma_busy_fraction"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * N_SIMD * CLOCK_HZ)
                e["mfma_tflops"] = 512.0 * v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / dur[k] / 1e12
            fl = 64.0 * (2.0 * v.get("SQ_INSTS_VALU_FMA_F64", 0.0) + v.get("SQ_INSTS_VALU_ADD_F64", 0.0) + v.get("SQ_INSTS_VALU_MUL_F64", 0.0))
            if fl > 0 and dur.get(k, 0.0) > 0:
                e["valu_fp64_tflops"] = fl / dur[k] / 1e12
            kern[short(k)] = e
        out["passes"][tag] = {"counters": counters, "kernels": kern}
    p = out["passes"]
    # J/K stream: HBM bytes per launch of the tuned dimer kernel
    if "fetch" in p and "write" in p:
        def pick(tag, counter, sub):
            for k, v in p[tag]["kernels"].items():
                if sub in k and ("12, 19" in k or sub == "jk_tri_kernel"):
                    return v.get(counter, 0.0), v["launches"], k
            return 0.0, 0, None
        # the triangular-tensor kernel where the batch runs from it, else the tuned square kernel
        fk, nf, jk_kernel = pick("fetch", "FETCH_SIZE", "jk_tri_kernel")
        wk, nw, _ = pick("write", "WRITE_SIZE", "jk_tri_kernel")
        if not nf:
            fk, nf, jk_kernel = pick("fetch", "FETCH_SIZE", "jk_incore_kernel")
            wk, nw, _ = pick("write", "WRITE_SIZE", "jk_incore_kernel")
        if nf:
            out["jk_kernel"] = jk_kernel
            out["jk_hbm_bytes_per_launch"] = (2.0 * fk * 1024.0) / nf + (wk * 1024.0) / max(nw, 1)
            out["jk_correction"] = "FETCH_SIZE x2 on gfx950 (64 B tallied per 128-B request); KiB -> bytes x1024"
        # integral stage: all eri_* / schwarz_* kernels
        tot_f = sum(v.get("FETCH_SIZE", 0.0) for k, v in p["fetch"]["kernels"].items() if k.startswith(("eri_", "schwarz_")) or "fillBuffer" in k)
        tot_w = sum(v.get("WRITE_SIZE", 0.0) for k, v in p["write"]["kernels"].items() if k.startswith(("eri_", "schwarz_")) or "fillBuffer" in k)
        out["eri_stage_hbm_bytes_total"] = 2.0 * tot_f * 1024.0 + tot_w * 1024.0
    # density-fitted J/K: HBM bytes per launch of df_jk_mfma_kernel from its own FETCH_SIZE / WRITE_SIZE passes of bench.py --df
    if "fetch_df" in p and "write_df" in p:
        def pick_df(tag, counter):
            best = (0.0, 0, 0.0)
            for k, v in p[tag]["kernels"].items():
                if k.startswith("df_jk_mfma_kernel") and v.get(counter, 0.0) > best[0]:
                    best = (v.get(counter, 0.0), v["launches"], v.get("seconds_in_this_pass", 0.0))
            return best
        fk, nf, sec = pick_df("fetch_df", "FETCH_SIZE")
        wk, nw, _ = pick_df("write_df", "WRITE_SIZE")
        if nf:
            out["df_jk_hbm_bytes_per_launch"] = (2.0 * fk * 1024.0) / nf + (wk * 1024.0) / max(nw, 1)
            out["df_jk_launches_in_pass"] = nf
            if sec > 0:
                out["df_jk_hbm_gbs_in_pass"] = (2.0 * fk * 1024.0 + wk * 1024.0 * nf / max(nw, 1)) / sec / 1e9
    if "sq" in p:
        fl = 0.0
        for k, v in p["sq"]["kernels"].items():
            if k.startswith(("eri_", "schwarz_")):
                fl += 64.0 * (2.0 * v.get("SQ_INSTS_VALU_FMA_F64", 0.0) + v.get("SQ_INSTS_VALU_ADD_F64", 0.0) + v.get("SQ_INSTS_VALU_MUL_F64", 0.0))
        out["eri_fp64_flop_total"] = fl
        out["eri_fp64_evaluations_in_pass"] = 2          # bench.py --steps 1 --warmup 1: the cold evaluation + one timed step
        out["eri_fp64_note"] = "64 lanes x (2 FMA + ADD + MUL) wave-instructions of the eri_* / schwarz_* kernels; upper bound (inactive lanes counted)"
    with open(os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % ROUND), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "passes"}, indent=1))


if __name__ == "__main__":
    main()

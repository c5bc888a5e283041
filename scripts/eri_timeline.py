#!/usr/bin/env python3
"""Timeline of one evaluation's integral stage from a rocprofv3 kernel trace (scripts/eri_timeline.sh):
per hardware queue, the kernels in start order with start / end relative to the evaluation's first dispatch.

    python scripts/eri_timeline.py gpurun_out/tl1/kernel_trace.csv [evaluation index, default last]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void mqc::", "").replace("(anonymous namespace)::", "").replace("mqc::", "")
    return n[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"]), int(r["Grid_Size_X"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])) for r in rows]
    ev.sort()
    # evaluations are separated by idle gaps; cut at the big jk launches' first occurrence after a gap > 2 ms
    cuts = [0]
    last_end = ev[0][1]
    for i, e in enumerate(ev):
        if e[0] - last_end > 1_500_000:
            cuts.append(i)
        last_end = max(last_end, e[1])
    cuts.append(len(ev))
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(cuts) - 2
    seg = ev[cuts[which]:cuts[which + 1]]
    t0 = seg[0][0]
    print("evaluation %d of %d: %d dispatches, %.2f ms" % (which, len(cuts) - 1, len(seg), (max(e[1] for e in seg) - t0) / 1e6))
    for e in seg:
        if e[1] - e[0] < 150_000:
            continue
        print("q%-3d %8.2f -> %8.2f  (%6.2f ms)  grid %9d  regs %3d  %s" % (e[3], (e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[4], e[5], e[2]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Private-segment (scratch) bytes per lane and registers of every kernel in the built libmqc_hip.so, from the code
objects' metadata (llvm-objdump --offloading + llvm-readelf --notes; no GPU needed).

    python scripts/scratch_report.py [--all]        default: kernels with a private segment only

ROCm reserves scratch per hardware queue for a full device of waves the first time a kernel with a private segment
runs on that queue: bytes per lane x 64 lanes x (CUs x 32 waves).  The engine subtracts that reservation for its
hardware queues from the HBM budget of its pools (engine.cpp: scratch_reservation_bytes, SCRATCH_BOUND_PER_LANE) and
tests/test_host_logic.py asserts that no kernel the dispatchers launch exceeds the bound."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=None):
    """-> list of (name, private_segment_bytes_per_lane, vgpr, sgpr, lds_bytes)"""
    lib = lib or os.path.join(ROOT, "metalquicha_amd", "libmqc_hip.so")
    tmp = tempfile.mkdtemp(prefix="mqc_scratch_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copyfile(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        out = []
        for f in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950"))):
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count")[1:]:
                g = lambda k: int(re.search(k + r":\s+(\d+)", blk).group(1))      # noqa: E731
                out.append((re.search(r"\.name:\s+(\S+)", blk).group(1), g(r"\.private_segment_fixed_size"), g(r"\.vgpr_count"),
                            g(r"\.sgpr_count"), g(r"\.group_segment_fixed_size")))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pretty(name):
    """_ZN3mqc17eri_digest_kernelILi2ELi2ELi1ELi1EEEv... -> ("eri_digest_kernel", [2, 2, 1, 1]) (this library's kernels only
    carry integer / bool template arguments)"""
    m = re.match(r"_ZN3mqc(?:\d+_GLOBAL__N_1)?(\d+)", name)
    if not m:
        return name, []
    n = int(m.group(1)); start = m.end()
    base = name[start:start + n]
    rest = name[start + n:]
    args = []
    if rest.startswith("I"):
        for a in re.finditer(r"L([ib])(\d+)E", rest.split("EEv")[0] + "E"):
            args.append(int(a.group(2)))
    return base, args


def demangle(names):
    out = []
    for n in names:
        b, a = pretty(n)
        out.append("%s<%s>" % (b, ", ".join(map(str, a))) if a else b)
    return out


if __name__ == "__main__":
    ks = kernels()
    show = ks if "--all" in sys.argv else [k for k in ks if k[1] > 0]
    show.sort(key=lambda k: -k[1])
    names = demangle([k[0] for k in show])
    print("# %d kernels in the library, %d with a private segment" % (len(ks), sum(1 for k in ks if k[1] > 0)))
    print("# scratch B/lane  vgpr  sgpr  LDS B  kernel")
    for (n, p, v, s, l), d in zip(show, names):
        print("%8d %5d %5d %7d  %s" % (p, v, s, l, d.split("(")[0]))

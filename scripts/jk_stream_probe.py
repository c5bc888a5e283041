#!/usr/bin/env python3
"""Measurement aid: J/K kernel rate with the exchange part switched off (MQC_HIP_JK_SKIP_EXCHANGE=1 gives wrong K,
so fragment errors are ignored here); tells how much of the kernel time is the K contraction and how much the stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalquicha_amd import mbe, methods
system = mbe.water_cluster(4)
terms = [t for t in mbe.generate_mbe_term_list(system, 2) if len(t) == 2]
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", max_iter=6, allow_crap_scf=True)
mbe.run_mbe(system, st, level=2, terms=terms)
methods.get_stats()
mbe.run_mbe(system, st, level=2, terms=terms)
s = methods.get_stats()
print("jk GB/s %.0f  kernel s %.4f launches %d" % (s.fock_bytes / s.fock_kernel_seconds / 1e9, s.fock_kernel_seconds, s.fock_launches))

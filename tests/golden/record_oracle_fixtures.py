"""Records the slow oracle results of tests/test_gpu_workloads.py into tests/golden/oracle_fixtures.json.

    python tests/golden/record_oracle_fixtures.py [case ...]        # cases: fmo_df_rks gmbe tpss df_grad df_f f_grad (default: all)

CPU only: the oracle is numpy + oracle/liboracle_ints.so; nothing here touches the engine.  The keys are those the
tests compute (label + digest of geometry and settings), so a changed input can never pick up a stale record."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402

from tests import helpers, workload_cases as wc      # noqa: E402


def main(argv):
    want = set(argv) or {"fmo_df_rks", "gmbe", "tpss", "df_grad", "df_f", "f_grad"}
    path = helpers._FIXTURE_PATH
    cur = json.load(open(path)) if os.path.isfile(path) else {}
    if "fmo_df_rks" in want:
        system = wc.fmo_df_rks_system()
        key = helpers._fixture_key("fmo2_df_b3lyp_water8", helpers.fragment_bohr(system.element_numbers, system.coordinates.T),
                                   wc.FMO_DF_RKS_KEY)
        t0 = time.time()
        cur[key] = wc.fmo_df_rks_oracle()
        print(key, cur[key]["energy"], "%.0f s" % (time.time() - t0), flush=True)
    if "gmbe" in want:
        system = wc.gmbe_system()
        z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
        sets = [tuple(int(a) for a in m) for m in system.monomers]
        sets += [sets[i] + sets[j] for i in range(3) for j in range(i + 1, 3)]
        for atoms in sets:
            f = helpers.fragment_bohr(z[list(atoms)], xyz[list(atoms)])
            key = helpers._fixture_key("gmbe2_b3lyp_def2tzvp", f, wc.GMBE_KEY)
            t0 = time.time()
            cur[key] = wc.gmbe_fragment_oracle(f)
            print(key, cur[key]["energy"], "%.0f s" % (time.time() - t0), flush=True)
    if "tpss" in want:
        for df in (False, True):
            for f in wc.tpss_fragments():
                key = helpers._fixture_key("tpss_water_batch", f, wc.TPSS_KEY % ("df:" + wc.AUX if df else "exact"))
                t0 = time.time()
                cur[key] = wc.tpss_oracle(f, df)
                print(key, cur[key]["energy"], "%.0f s" % (time.time() - t0), flush=True)
    if "df_f" in want:
        for fn in ("", "b3lyp"):
            for f in wc.df_f_fragments():
                key = helpers._fixture_key("df_f_orbitals", f, wc.DF_F_KEY % (fn or "rhf"))
                t0 = time.time()
                cur[key] = wc.df_f_oracle(f, fn)
                print(key, cur[key]["energy"], "%.0f s" % (time.time() - t0), flush=True)
    if "f_grad" in want:
        for df in (False, True):
            key = helpers._fixture_key("f_shell_gradient_co", helpers.fragment_bohr(wc.F_GRAD_Z, wc.F_GRAD_XYZ),
                                       wc.F_GRAD_KEY % ("df:" + wc.AUX if df else "exact"))
            t0 = time.time()
            cur[key] = wc.f_gradient_oracle(df)
            print(key, np.abs(np.array(cur[key]["gradient"])).max(), "%.0f s" % (time.time() - t0), flush=True)
    if "df_grad" in want:
        key = helpers._fixture_key("df_rhf_gradient_water", helpers.fragment_bohr([8, 1, 1], wc.DF_GRAD_XYZ), wc.DF_GRAD_KEY)
        t0 = time.time()
        cur[key] = wc.df_gradient_oracle()
        print(key, np.abs(np.array(cur[key]["gradient"])).max(), "%.0f s" % (time.time() - t0), flush=True)
    with open(path, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])

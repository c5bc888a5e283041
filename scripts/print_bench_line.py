"""Print the figures of one bench.py JSON line that the round's records quote."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("it/s %.0f  ms/step %.2f  %s %.3f ms/launch frac %.3f  E %.10f" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], d["mbe2_energy_hartree"]))
if d.get("b3lyp"):
    print("b3lyp %.3f s" % d["b3lyp"]["mbe2_wall_s"])
sec = d.get("secondary") or {}
if "single_call" in sec:
    print("single call %.2f ms/fragment" % sec["single_call"]["ms_per_fragment"])
if "fmo2_point_charge_field" in sec:
    print("fmo2", sec["fmo2_point_charge_field"].get("wall_s"), sec["fmo2_point_charge_field"].get("energy_hartree"))
if "benzene_b3lyp_df_single_fragment" in sec:
    print("benzene %.4f s" % sec["benzene_b3lyp_df_single_fragment"]["seconds_per_scf"])

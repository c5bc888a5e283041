#!/usr/bin/env python3
"""Harvests known-answer fixtures (DATA only: geometry, settings, expected energy) from the
reference's validation manifest for the cases this repository can reproduce with the basis
sets it ships (STO-3G, cc-pVDZ; H, C, N, O).  Run in the build container, where
/root/reference exists; the output tests/golden/manifest_subset.json is committed so that the
tests never read the reference at run time.

    python tests/golden/harvest_manifest.py
"""
import json
import os

REF = "/root/reference/validation"
HERE = os.path.dirname(os.path.abspath(__file__))
ALLOWED_ELEMENTS = {"H", "C", "N", "O"}
ALLOWED_BASES = {"sto-3g", "cc-pvdz"}


def read_xyz(path):
    lines = open(path).read().splitlines()
    n = int(lines[0].split()[0])
    sym, xyz = [], []
    for ln in lines[2:2 + n]:
        p = ln.split()
        sym.append(p[0]); xyz.append([float(p[1]), float(p[2]), float(p[3])])
    return sym, xyz


def main():
    manifest = json.load(open(os.path.join(REF, "validation_tests_cpu.json")))
    out = {"source": "validation/validation_tests_cpu.json (tolerance %g)" % manifest["tolerance"], "cases": []}
    for t in manifest["tests"]:
        deck_path = os.path.join(REF, t["input"])
        if not os.path.isfile(deck_path):
            continue
        deck = json.load(open(deck_path))
        model = deck.get("model", {})
        method = model.get("method", "").lower()
        basis = model.get("basis", "").lower()
        if method not in ("hf", "dft") or basis not in ALLOWED_BASES:
            continue
        kw = deck.get("keywords", {})
        scf = kw.get("scf", {})
        # an aux_basis named without density_fitting is inert on the CPU path (exact ERIs)
        if scf.get("unrestricted") or scf.get("density_fitting"):
            continue
        if deck.get("driver", "Energy") not in ("Energy", "Gradient"):
            continue
        mol = deck["molecules"][0]
        if mol.get("molecular_multiplicity", 1) != 1:
            continue
        sym, xyz = read_xyz(os.path.normpath(os.path.join(os.path.dirname(deck_path), mol["xyz"])))
        if not set(sym) <= ALLOWED_ELEMENTS:
            continue
        frag = kw.get("fragmentation")
        case = {
            "name": t["name"], "expected_energy": t["expected_energy"], "deck": t["input"],
            "symbols": sym, "xyz_angstrom": xyz, "charge": mol.get("molecular_charge", 0),
            "method": method, "basis": basis, "functional": model.get("functional", ""),
            "tolerance": scf.get("tolerance", 1e-8), "maxiter": scf.get("maxiter", 100),
            "grid_level": kw.get("dft", {}).get("grid_level", 3),
        }
        if frag:
            if frag.get("method", "").upper() != "MBE" or frag.get("embedding", "none") != "none":
                continue
            case["mbe_level"] = frag["level"]
            case["fragments"] = mol["fragments"]
        if "pcm" in kw or "solvent" in json.dumps(kw).lower():
            continue
        out["cases"].append(case)
    with open(os.path.join(HERE, "manifest_subset.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in out["cases"]:
        print("%-45s %s %s %s %.12f" % (c["name"], c["method"], c["basis"], c["functional"], c["expected_energy"]))


if __name__ == "__main__":
    main()

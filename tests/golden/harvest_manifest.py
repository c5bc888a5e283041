#!/usr/bin/env python3
"""Harvests known-answer fixtures (DATA only: geometry, settings, expected energy / gradient) from the
reference's validation manifest for the cases this repository can reproduce with the basis sets it ships
(metalquicha_amd/basis_data: STO-3G, cc-pVDZ, 6-31G, 6-31G*, 6-31G**, def2-SVP, def2-TZVP; H, C, N, O).
Run in the build container, where /root/reference exists; the output tests/golden/manifest_subset.json is
committed so that the tests never read the reference at run time.

    python tests/golden/harvest_manifest.py
"""
import json
import os

REF = "/root/reference/validation"
HERE = os.path.dirname(os.path.abspath(__file__))
BASIS_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "metalquicha_amd", "basis_data")
ALLOWED_ELEMENTS = {"H", "C", "N", "O"}


def have_basis(name):
    return os.path.isfile(os.path.join(BASIS_DIR, name.strip().lower().replace("*", "_st_") + ".json"))


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
        if method not in ("hf", "dft") or not have_basis(basis):
            continue
        kw = deck.get("keywords", {})
        scf = kw.get("scf", {})
        df = bool(scf.get("density_fitting"))
        aux = model.get("aux_basis", "").lower()
        # an aux_basis named without density_fitting is inert on the CPU path (exact ERIs)
        if df and not have_basis(aux):
            continue
        if deck.get("driver", "Energy") not in ("Energy", "Gradient"):
            continue
        if len(deck["molecules"]) != 1:
            continue
        mol = deck["molecules"][0]
        sym, xyz = read_xyz(os.path.normpath(os.path.join(os.path.dirname(deck_path), mol["xyz"])))
        if not set(sym) <= ALLOWED_ELEMENTS:
            continue
        extra = {k: v for k, v in kw.items() if k not in ("scf", "dft", "fragmentation")}
        if extra or "pcm" in json.dumps(deck).lower() or "properties" in deck or "guess_steps" in json.dumps(scf):
            continue            # solvation, post-HF, analyses: outside the SCF hot path
        if any(k in t["name"] for k in ("MP2", "DH ", "Fukui", "quasi", "Quasi", "No-sharing", "SAPT", "EFP", "CC", "CAS")):
            continue
        frag = kw.get("fragmentation")
        mult = mol.get("molecular_multiplicity", 1)
        case = {
            "name": t["name"], "expected_energy": t["expected_energy"], "deck": t["input"],
            "symbols": sym, "xyz_angstrom": xyz, "charge": mol.get("molecular_charge", 0), "multiplicity": mult,
            "method": method, "basis": basis, "functional": model.get("functional", ""),
            "density_fitting": df, "aux_basis": aux if df else "",
            "unrestricted": bool(scf.get("unrestricted")) or mult != 1,
            "guess": scf.get("guess", "auto"),
            "tolerance": scf.get("tolerance", 1e-8), "maxiter": scf.get("maxiter", 100),
            "grid_level": kw.get("dft", {}).get("grid_level", 3),
            "driver": deck.get("driver", "Energy"),
        }
        if "expected_gradient" in t:
            case["expected_gradient"] = t["expected_gradient"]
            case["gradient_tolerance"] = t.get("gradient_tolerance", 1e-8)
        if frag:
            fm = frag.get("method", "").upper()
            if fm != "MBE" or frag.get("embedding", "none") != "none" or frag.get("allow_overlapping_fragments"):
                continue
            case["mbe_level"] = frag["level"]
            case["expansion"] = frag.get("expansion", "mbe")       # "fmo": FMO-2 with embedding potentials
            case["fragments"] = mol["fragments"]
        out["cases"].append(case)
    with open(os.path.join(HERE, "manifest_subset.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in out["cases"]:
        print("%-70s %s %s %s %.12f" % (c["name"], c["method"], c["basis"], c["functional"], c["expected_energy"]))
    print(len(out["cases"]), "cases")


if __name__ == "__main__":
    main()

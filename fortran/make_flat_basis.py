"""Writes the flat text basis files the stand-in reader of fortran/stubs/reference_interface_stubs.f90 reads
(test infrastructure of the Fortran bridge check; inside metalquicha's tree the bridge uses the reference's own
Basis-Set-Exchange JSON reader and none of this).

    python fortran/make_flat_basis.py <out_dir> [basis ...]

One shell per contraction row, SP shells split, RAW coefficients -- metalquicha_amd.basis.read_element delivers exactly
what build_molecular_basis_json does (src/basis/mqc_json_basis_reader.f90:213-309)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metalquicha_amd import basis      # noqa: E402

DEFAULT = ["sto-3g", "sto-3g-check_rhf", "cc-pvdz", "6-31g", "def2-svp", "mqc-even-tempered-jkfit"]


def write_flat(name: str, out_dir: str):
    path = basis.find_basis_file(name)
    elements = []
    for z in range(1, 11):
        try:
            eb = basis.read_element(path, z)
        except basis.BasisError:
            continue
        if eb.cartesian or not eb.shells:
            continue
        elements.append((basis.SYMBOLS[z], eb))
    with open(os.path.join(out_dir, name + ".flat"), "w") as f:
        f.write("%d\n" % len(elements))
        for sym, eb in elements:
            f.write("%s %d\n" % (sym, len(eb.shells)))
            for s in eb.shells:
                f.write("%d %d\n" % (s.l, s.nprim))
                for e, c in zip(s.exps, s.coefs):
                    f.write("%.17e %.17e\n" % (e, c))


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for name in (sys.argv[2:] or DEFAULT):
        write_flat(name, out)

#!/usr/bin/env python3
"""Writes the Basis-Set-Exchange JSON files this repository ships.

The reference's bundle `basis_sets/basis_sets-json-0.12.tar.bz2` is absent from the
container (/root/reference/.MISSING_LARGE_BLOBS), so the published tables are typed in
here (BSE v0.12 values: STO-3G, Hehre/Stewart/Pople 1969; cc-pVDZ, Dunning 1989) and
emitted in the format `src/basis/mqc_json_basis_reader.f90` reads: elements keyed by Z,
`electron_shells[*]` with `function_type`, `angular_momentum`, string `exponents`, and
`coefficients` as a list of rows (several rows = general contraction; [0,1] = SP shell).
The data are pinned by the reference's goldens (tests/test_oracle_golden.py): a wrong
digit in H or O moves the H2O energies by far more than 1e-9 Eh.

`def2-universal-jkfit` / `cc-pvdz-rifit` / `cc-pvtz-jkfit` are NOT reproducible from
memory; `mqc-even-tempered-jkfit` is this repo's own documented auxiliary set (see
DESIGN.md) and is used for DF parity between the HIP path and the oracle only.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def fmt(x):
    return "%.10E" % x


def shell(am, exps, rows, ftype="gto"):
    return {
        "function_type": ftype,
        "region": "",
        "angular_momentum": list(am),
        "exponents": [fmt(e) for e in exps],
        "coefficients": [[fmt(c) for c in row] for row in rows],
    }


def write(name, description, elements):
    doc = {
        "molssi_bse_schema": {"schema_type": "complete", "schema_version": "0.1"},
        "name": name,
        "description": description,
        "function_types": ["gto", "gto_spherical"],
        "elements": {str(z): {"electron_shells": sh} for z, sh in elements.items()},
    }
    fn = os.path.join(HERE, name.lower().replace("*", "_st_") + ".json")
    with open(fn, "w") as f:
        json.dump(doc, f, indent=1)
    return fn


# ---------------------------------------------------------------- STO-3G (BSE, 10 digits)
STO_1S = [0.1543289673, 0.5353281423, 0.4446345422]
STO_2S = [-0.09996722919, 0.3995128261, 0.7001154689]
STO_2P = [0.1559162750, 0.6076837186, 0.3919573931]


def sto3g_row2(e1s, e2sp):
    return [shell([0], e1s, [STO_1S]), shell([0, 1], e2sp, [STO_2S, STO_2P])]


sto3g = {
    1: [shell([0], [3.425250914, 0.6239137298, 0.1688554040], [STO_1S])],
    6: sto3g_row2([71.61683735, 13.04509632, 3.530512160], [2.941249355, 0.6834830964, 0.2222899159]),
    7: sto3g_row2([99.10616896, 18.05231239, 4.885660238], [3.780455879, 0.8784966449, 0.2857143744]),
    8: sto3g_row2([130.7093214, 23.80886605, 6.443608313], [5.033151319, 1.169596125, 0.3803889600]),
}

# The values written inline in the reference's validation/check_rhf.f90:149-177 (8 digits),
# which is what its -74.9658162796 / -1.1167143251 goldens were computed with.
sto3g_check_rhf = {
    1: [shell([0], [3.42525091, 0.62391373, 0.16885540], [[0.15432897, 0.53532814, 0.44463454]])],
    8: [
        shell([0], [130.7093200, 23.8088610, 6.4436083], [[0.15432897, 0.53532814, 0.44463454]]),
        shell([0, 1], [5.0331513, 1.1695961, 0.3803890],
              [[-0.09996723, 0.39951283, 0.70011547], [0.15591627, 0.60768372, 0.39195739]]),
    ],
}

# ---------------------------------------------------------------- cc-pVDZ (BSE general contractions)
S = "gto_spherical"
ccpvdz = {
    1: [
        shell([0], [13.01, 1.962, 0.4446, 0.122],
              [[0.019685, 0.137977, 0.478148, 0.50124], [0.0, 0.0, 0.0, 1.0]], S),
        shell([1], [0.727], [[1.0]], S),
    ],
    6: [
        shell([0], [6665.0, 1000.0, 228.0, 64.71, 21.06, 7.495, 2.797, 0.5215, 0.1596],
              [[0.000692, 0.005329, 0.027077, 0.101718, 0.27474, 0.448564, 0.285074, 0.015204, -0.003191],
               [-0.000146, -0.001154, -0.005725, -0.023312, -0.063955, -0.149981, -0.127262, 0.544529, 0.580496],
               [0, 0, 0, 0, 0, 0, 0, 0, 1.0]], S),
        shell([1], [9.439, 2.002, 0.5456, 0.1517],
              [[0.038109, 0.20948, 0.508557, 0.468842], [0, 0, 0, 1.0]], S),
        shell([2], [0.55], [[1.0]], S),
    ],
    7: [
        shell([0], [9046.0, 1357.0, 309.3, 87.73, 28.56, 10.21, 3.838, 0.7466, 0.2248],
              [[0.0007, 0.005389, 0.027406, 0.103207, 0.278723, 0.44854, 0.278238, 0.01544, -0.002864],
               [-0.000153, -0.001208, -0.005992, -0.024544, -0.067459, -0.158078, -0.121831, 0.549003, 0.578815],
               [0, 0, 0, 0, 0, 0, 0, 0, 1.0]], S),
        shell([1], [13.55, 2.917, 0.7973, 0.2185],
              [[0.039919, 0.217169, 0.510319, 0.462214], [0, 0, 0, 1.0]], S),
        shell([2], [0.817], [[1.0]], S),
    ],
    8: [
        shell([0], [11720.0, 1759.0, 400.8, 113.7, 37.03, 13.27, 5.025, 1.013, 0.3023],
              [[0.00071, 0.00547, 0.027837, 0.1048, 0.283062, 0.448719, 0.270952, 0.015458, -0.002585],
               [-0.00016, -0.001263, -0.006267, -0.025716, -0.070924, -0.165411, -0.116955, 0.557368, 0.572759],
               [0, 0, 0, 0, 0, 0, 0, 0, 1.0]], S),
        shell([1], [17.7, 3.854, 1.046, 0.2753],
              [[0.043018, 0.228913, 0.508728, 0.460531], [0, 0, 0, 1.0]], S),
        shell([2], [1.185], [[1.0]], S),
    ],
}


def even_tempered_aux(z):
    """This repo's own auxiliary set (NOT a published one): even-tempered, uncontracted.

    Exponents beta^k * alpha0 per angular momentum, spanning products of the cc-pVDZ
    orbital exponents.  Used only for DF parity (HIP vs oracle), never against a golden.
    """
    if z == 1:
        spec = {0: (0.18, 2.6, 5), 1: (0.3, 2.6, 3), 2: (0.6, 2.6, 2)}
    else:
        scale = {6: 0.75, 7: 0.9, 8: 1.0}[z]
        spec = {0: (0.22 * scale, 2.4, 11), 1: (0.3 * scale, 2.5, 7), 2: (0.35 * scale, 2.6, 5),
                3: (0.6 * scale, 2.8, 3)}
    shells = []
    for l, (a0, beta, n) in sorted(spec.items()):
        for k in reversed(range(n)):
            shells.append(shell([l], [a0 * beta ** k], [[1.0]], S))
    return shells


if __name__ == "__main__":
    print(write("sto-3g", "STO-3G minimal basis (BSE values)", sto3g))
    print(write("sto-3g-check_rhf",
                "STO-3G as written inline in the reference's validation/check_rhf.f90 (8 digits)",
                sto3g_check_rhf))
    print(write("cc-pvdz", "cc-pVDZ (BSE general-contraction form)", ccpvdz))
    print(write("mqc-even-tempered-jkfit",
                "repo-own even-tempered auxiliary basis for DF parity tests (not a published set)",
                {z: even_tempered_aux(z) for z in (1, 6, 7, 8)}))

"""Inputs and oracle recipes of tests/test_gpu_workloads.py, importable without a GPU so that
tests/golden/record_oracle_fixtures.py can run the oracle side on the CPU and write tests/golden/oracle_fixtures.json."""
import numpy as np

from metalquicha_amd import mbe
from metalquicha_amd.methods import ScfSettings

AUX = "mqc-even-tempered-jkfit"

# ---- configs[4] at reduced size: FMO-2, density-fitted B3LYP, (H2O)8 / cc-pVDZ ------------------------------------------
FMO_DF_RKS_KEY = "cc-pvdz|b3lyp|df:%s|grid3|fmo2|ptc|mulliken|1e-9|1e-7|gwh|outer 1e-7" % AUX


def fmo_df_rks_system():
    return mbe.water_cluster(2, seed=11)


def fmo_df_rks_settings():
    return ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set=AUX,
                       energy_tol=1e-9, density_tol=1e-7, guess="gwh")


def fmo_df_rks_oracle():
    from oracle import fmo_oracle, xc_oracle
    from tests.helpers import oracle_make_mol
    system = fmo_df_rks_system()
    make, make_aux = oracle_make_mol(system, "cc-pvdz"), oracle_make_mol(system, AUX)
    cache = {}

    def extra(atoms, mol):
        # the grid, the fitted tensor's ingredients and the functional object of a fragment do not change between passes
        key = tuple(atoms)
        if key not in cache:
            cache[key] = (make_aux(atoms), xc_oracle.XCOracle(mol, "b3lyp", 3))
        aux, xc = cache[key]
        return {"aux": aux, "xc": xc}
    frags = [list(map(int, m)) for m in system.monomers]
    r = fmo_oracle.run_fmo2(make, np.asarray(system.element_numbers), np.ascontiguousarray(system.coordinates.T), frags,
                            expansion="fmo", scf_extra=extra)
    return {"energy": float(r.energy), "iterations": int(r.outer_iterations), "converged": bool(r.converged),
            "monomer_energy": [float(v) for v in r.monomer_energy], "response_sum": float(r.response_sum),
            "charges": [float(v) for v in r.charges]}


# ---- configs[3] at reduced size: GMBE(2), B3LYP / def2-TZVP, (H2O)3 -----------------------------------------------------
GMBE_KEY = "def2-tzvp|b3lyp|grid3|1e-9|1e-7|gwh"


def gmbe_system():
    """The first three waters of a 2 x 2 x 2 cluster at 2.9 Angstrom spacing, fragments = molecules."""
    full = mbe.water_cluster(2, spacing=2.9, seed=7)
    keep = [int(a) for m in full.monomers[:3] for a in m]
    return mbe.FragmentedSystem(np.asarray(full.element_numbers)[keep], np.ascontiguousarray(full.coordinates[:, keep]),
                                [np.arange(3 * k, 3 * k + 3) for k in range(3)])


def gmbe_settings():
    return ScfSettings(basis_set="def2-tzvp", functional="b3lyp", energy_tol=1e-9, density_tol=1e-7, guess="gwh")


def gmbe_fragment_oracle(frag):
    from oracle import scf_oracle as so, xc_oracle
    from tests.helpers import oracle_mol, scf_record
    mol = oracle_mol("def2-tzvp", frag)
    return scf_record(so.run_rhf(mol, int(frag.nelec), 100, 1e-9, 1e-7, xc=xc_oracle.XCOracle(mol, "b3lyp", 3)))


# ---- meta-GGA: TPSS on a water and a water dimer, cc-pVDZ, exact and density-fitted Coulomb -------------------------------
TPSS_KEY = "cc-pvdz|tpss|%s|grid3|1e-9|1e-7|gwh"


def tpss_fragments():
    from tests.helpers import fragment_bohr, water_at
    rng = np.random.default_rng(21)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.3, 0.4, -0.2])]
    return [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws))]


def tpss_settings(df):
    return ScfSettings(basis_set="cc-pvdz", functional="tpss", density_fitting=df, aux_basis_set=AUX if df else "",
                       energy_tol=1e-9, density_tol=1e-7, guess="gwh")


def tpss_oracle(frag, df):
    from oracle import scf_oracle, xc_oracle
    from tests.helpers import oracle_mol, scf_record
    mol = oracle_mol("cc-pvdz", frag)
    aux = oracle_mol(AUX, frag) if df else None
    o = scf_oracle.run_rhf(mol, int(frag.nelec), 100, 1e-9, 1e-7, aux=aux, xc=xc_oracle.XCOracle(mol, "tpss", 3))
    return scf_record(o)


# ---- density-fitted gradient: four-point differences of the oracle's DF-RHF energy on a bent, asymmetric water -------------
DF_GRAD_KEY = "cc-pvdz|df:%s|rhf|1e-12|1e-10|four-point h=2e-3" % AUX
DF_GRAD_XYZ = np.array([[0.03, -0.02, -0.13], [0.10, 1.45, 1.05], [-0.05, -1.38, 1.12]])


def df_gradient_oracle():
    """-> {"gradient": [3][natoms]} by f' = [8 (f(h) - f(-h)) - (f(2h) - f(-2h))] / 12h of the oracle's density-fitted RHF energy
    (check_gradient's procedure, validation/check_gradient.f90, with a higher-order stencil)."""
    from oracle import scf_oracle
    from tests.helpers import fragment_bohr, oracle_mol
    h = 2e-3

    def energy(x):
        f = fragment_bohr([8, 1, 1], x)
        o = scf_oracle.run_rhf(oracle_mol("cc-pvdz", f), 10, 200, 1e-12, 1e-10, aux=oracle_mol(AUX, f))
        assert o.converged
        return o.energy
    g = np.zeros((3, 3))
    for a in range(3):
        for c in range(3):
            e = {}
            for k in (-2, -1, 1, 2):
                x = DF_GRAD_XYZ.copy(); x[a, c] += k * h
                e[k] = energy(x)
            g[c, a] = (8.0 * (e[1] - e[-1]) - (e[2] - e[-2])) / (12.0 * h)
    return {"gradient": g.tolist()}


# ---- density fitting with f shells in the orbital basis (def2-TZVP): the density-fitted form of configs[3]'s method --------
DF_F_KEY = "def2-tzvp|%s|df:" + AUX + "|grid3|1e-9|1e-7|gwh"


def df_f_fragments():
    from tests.helpers import fragment_bohr, water_at
    rng = np.random.default_rng(33)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.4, -0.5, 0.3])]
    return [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws))]


def df_f_settings(functional):
    return ScfSettings(basis_set="def2-tzvp", functional=functional, density_fitting=True, aux_basis_set=AUX,
                       energy_tol=1e-9, density_tol=1e-7, guess="gwh")


def df_f_oracle(frag, functional):
    from oracle import scf_oracle, xc_oracle
    from tests.helpers import oracle_mol, scf_record
    mol = oracle_mol("def2-tzvp", frag)
    xc = xc_oracle.XCOracle(mol, functional, 3) if functional else None
    return scf_record(scf_oracle.run_rhf(mol, int(frag.nelec), 100, 1e-9, 1e-7, aux=oracle_mol(AUX, frag), xc=xc))


# ---- f-shell gradients: carbon monoxide in def2-TZVP (an f shell on both atoms: every f class, the chunked ones included) ----
F_GRAD_KEY = "def2-tzvp|%s|1e-12|1e-10|four-point h=2e-3"
F_GRAD_Z = [6, 8]
F_GRAD_XYZ = np.array([[0.02, -0.03, 0.01], [0.11, 0.07, 2.17]])


def f_gradient_oracle(df):
    """Four-point differences of the oracle's RHF energy (exact or density-fitted) -> {"gradient": [3][2]}."""
    from oracle import scf_oracle
    from tests.helpers import fragment_bohr, oracle_mol
    h = 2e-3

    def energy(x):
        f = fragment_bohr(F_GRAD_Z, x)
        o = scf_oracle.run_rhf(oracle_mol("def2-tzvp", f), 14, 200, 1e-12, 1e-10, aux=oracle_mol(AUX, f) if df else None)
        assert o.converged
        return o.energy
    g = np.zeros((3, 2))
    for a in range(2):
        for c in range(3):
            e = {}
            for k in (-2, -1, 1, 2):
                x = F_GRAD_XYZ.copy(); x[a, c] += k * h
                e[k] = energy(x)
            g[c, a] = (8.0 * (e[1] - e[-1]) - (e[2] - e[-2])) / (12.0 * h)
    return {"gradient": g.tolist()}

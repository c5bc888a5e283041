"""Shared helpers for the parity tests: build the same fragment for the engine and the oracle."""
import numpy as np

from metalquicha_amd.basis import build_flat_basis
from metalquicha_amd.methods import PhysicalFragment
from oracle import scf_oracle as so

W1_ANGSTROM = [[0.0, 0.00000000009155, 0.10077199490609],
               [0.0, 0.77250895271063, -0.46780199741728],
               [0.0, -0.77250895280218, -0.46780199748881]]   # validation/inputs/sample_inputs/w1.xyz


def fragment_bohr(Z, xyz_bohr, **kw) -> PhysicalFragment:
    return PhysicalFragment(np.array(Z), np.asarray(xyz_bohr, dtype=float).reshape(-1, 3).T.copy(), **kw)


def oracle_mol(basis: str, frag: PhysicalFragment):
    fb = build_flat_basis(basis, frag.element_numbers, allow_cartesian=True)
    return so.make_mol(frag.element_numbers, frag.coordinates.T, fb.nshell_per_atom, fb.shell_l, fb.shell_nprim,
                       fb.exps, fb.coefs, ghost=frag.ghost, cart=not fb.spherical)


def random_rotation(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)],
                     [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)],
                     [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]])


def water_at(rng, centre_bohr):
    from metalquicha_amd.basis import ANGSTROM_TO_BOHR
    w = np.array(W1_ANGSTROM) * ANGSTROM_TO_BOHR
    R = random_rotation(rng)
    return (w - w.mean(axis=0)) @ R.T + np.asarray(centre_bohr)


def synthetic_density(n):
    """The reference's direct-vs-in-core test density sin(0.7 i + 1.3 j) + transpose
    (test/test_mqc_libcint_direct.f90:59-102), made symmetric."""
    i, j = np.meshgrid(np.arange(1, n + 1), np.arange(1, n + 1), indexing="ij")
    d = np.sin(0.7 * i + 1.3 * j)
    return 0.5 * (d + d.T)

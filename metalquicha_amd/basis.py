"""Basis-set input side of the drop-in boundary.

Mirrors what the reference keeps on the Fortran side of the bridge:
  * the Basis Set Exchange JSON reader, `src/basis/mqc_json_basis_reader.f90:154-309`
    (elements keyed by Z; SP shells split into an s and a p shell on shared exponents;
    one shell per coefficient row of a general contraction, in file order),
  * `load_basis` in the cuEST driver, `backends/cuest/backend/mqc_cuest_driver.f90:299-344`
    (spherical only: a Cartesian basis is refused), and
  * the flattening into plain arrays that the C ABI takes:
    per atom a shell count, per shell (l, nprim), and flat RAW exponents/coefficients.
    Normalisation happens below the boundary (csrc/basis_norm.cpp), once, identically
    to `backends/libcint/mqc_libcint_integrals.F90:519-555`.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

BASIS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "basis_data")

SYMBOLS = ["X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P",
           "S", "Cl", "Ar", "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn",
           "Ga", "Ge", "As", "Se", "Br", "Kr"]
SYMBOL_TO_Z = {s.lower(): z for z, s in enumerate(SYMBOLS)}

ANGSTROM_TO_BOHR = 1.0 / 0.529177210903   # CODATA 2018, the reference default (src/core/mqc_physical_constants.F90:33-37)


class BasisError(ValueError):
    """Raised for what the reference reports as ERROR_IO / ERROR_PARSE / ERROR_VALIDATION."""


@dataclass
class Shell:
    l: int
    exps: np.ndarray      # (nprim,)
    coefs: np.ndarray     # (nprim,)  raw BSE coefficients of unnormalised primitives

    @property
    def nprim(self) -> int:
        return len(self.exps)


@dataclass
class ElementBasis:
    shells: List[Shell]
    cartesian: bool = False


def basis_file_name(name: str) -> str:
    """File name = lower-cased BSE name with '*' -> '_st_' (basis_sets/PROVENANCE.md)."""
    return name.strip().lower().replace("*", "_st_") + ".json"


def find_basis_file(name: str) -> str:
    """Search $MQC_BASIS_PATH, ./basis_sets, then the packaged basis_data directory."""
    fn = basis_file_name(name)
    candidates = []
    env = os.environ.get("MQC_BASIS_PATH")
    if env:
        candidates += [os.path.join(p, fn) for p in env.split(os.pathsep) if p]
    candidates += [os.path.join("basis_sets", fn), os.path.join(BASIS_DIR, fn)]
    for c in candidates:
        if os.path.isfile(c):
            return c
    raise BasisError("basis set file not found: %s (looked in %s)" % (fn, ", ".join(candidates)))


_CACHE: Dict[str, dict] = {}


def _load_json(path: str) -> dict:
    # parsed once per process, like the reference's cached_basis_t (mqc_json_basis_reader.f90:43-70)
    if path not in _CACHE:
        with open(path) as f:
            _CACHE[path] = json.load(f)
    return _CACHE[path]


def read_element(path: str, z: int) -> ElementBasis:
    doc = _load_json(path)
    try:
        entry = doc["elements"][str(z)]
    except KeyError:
        raise BasisError("element Z=%d is not in basis file %s" % (z, path))
    shells: List[Shell] = []
    cartesian = False
    for sh in entry["electron_shells"]:
        ams = [int(a) for a in sh["angular_momentum"]]
        exps = np.array([float(e) for e in sh["exponents"]], dtype=np.float64)
        rows = [np.array([float(c) for c in row], dtype=np.float64) for row in sh["coefficients"]]
        ftype = sh.get("function_type", "gto")
        if len(ams) > 1:
            # SP (L) shell: row k belongs to angular momentum ams[k]
            if len(ams) != len(rows):
                raise BasisError("combined shell with %d momenta but %d coefficient rows" % (len(ams), len(rows)))
            for l, row in zip(ams, rows):
                shells.append(Shell(l, exps.copy(), row))
        else:
            for row in rows:   # general contraction: one shell per row, file order
                shells.append(Shell(ams[0], exps.copy(), row))
        if ftype == "gto_cartesian" and max(ams) > 1:
            cartesian = True   # one Cartesian shell above p makes the molecule Cartesian
    for s in shells:
        if len(s.exps) != len(s.coefs):
            raise BasisError("shell has %d exponents but %d coefficients" % (len(s.exps), len(s.coefs)))
    return ElementBasis(shells, cartesian)


@dataclass
class FlatBasis:
    """Plain arrays exactly as they cross the C ABI (include/mqc_hip.h: mqc_hip_basis_t)."""
    spherical: bool
    nshell_per_atom: np.ndarray          # int64 (n_atoms,)
    shell_l: np.ndarray                  # int32 (nshell,)
    shell_nprim: np.ndarray              # int32 (nshell,)
    exps: np.ndarray                     # float64 (sum nprim,)
    coefs: np.ndarray                    # float64 (sum nprim,) RAW
    name: str = ""
    _keep: list = field(default_factory=list, repr=False)

    @property
    def nshell(self) -> int:
        return int(self.shell_l.shape[0])

    @property
    def nao(self) -> int:
        return int(np.sum(2 * self.shell_l + 1))

    def shell_atoms(self) -> np.ndarray:
        return np.repeat(np.arange(len(self.nshell_per_atom)), self.nshell_per_atom).astype(np.int32)


def build_flat_basis(name: str, atomic_numbers: Sequence[int], allow_cartesian: bool = False) -> FlatBasis:
    """`load_basis` + flattening for one fragment (ghost atoms keep their functions).  `allow_cartesian` is for
    the test oracle only (the CPU reference routes Cartesian sets, the GPU boundary refuses them)."""
    path = find_basis_file(name)
    per_atom, ls, nps, ex, co = [], [], [], [], []
    any_cart = False
    for z in atomic_numbers:
        eb = read_element(path, int(z))
        any_cart = any_cart or eb.cartesian
        if eb.cartesian and not allow_cartesian:
            # mqc_cuest_driver.f90:331-341 -- the GPU path refuses Cartesian sets
            raise BasisError("basis %s is Cartesian (gto_cartesian above p); the HIP backend "
                             "supports spherical sets only" % name)
        per_atom.append(len(eb.shells))
        for s in eb.shells:
            ls.append(s.l); nps.append(s.nprim); ex.append(s.exps); co.append(s.coefs)
    return FlatBasis(
        spherical=not any_cart,
        nshell_per_atom=np.array(per_atom, dtype=np.int64),
        shell_l=np.array(ls, dtype=np.int32),
        shell_nprim=np.array(nps, dtype=np.int32),
        exps=np.ascontiguousarray(np.concatenate(ex)) if ex else np.zeros(0),
        coefs=np.ascontiguousarray(np.concatenate(co)) if co else np.zeros(0),
        name=name,
    )

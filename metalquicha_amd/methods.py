"""Method layer: the host-side mirror of the reference's L3 boundary for the SCF hot path.

  PhysicalFragment    <- physical_fragment_t   (src/fragmentation/common/mqc_physical_fragment.f90:45-94)
  ScfSettings         <- cuest_scf_settings_t  (src/methods/mqc_cuest_iface.f90:35-142)
  CalculationResult   <- calculation_result_t  (src/core/mqc_result_types.f90:92-196), the fields
                         run_cuest_scf fills (backends/cuest/backend/mqc_cuest_driver.f90:211-275)
  run_hip_scf         <- run_cuest_scf         (backends/cuest/backend/mqc_cuest_bridge.f90:32-39)
  hip_backend_available <- cuest_backend_available (:20-30)
  HFMethod.calc_energy  <- hf_calc_energy / hf_run (src/methods/mqc_method_hf.F90:113-217)
  run_hip_scf_batch   <- the batch-submit entry the worker loop would use

Same names, argument meaning and error behaviour; the numerics all happen in libmqc_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .basis import ANGSTROM_TO_BOHR, FlatBasis, build_flat_basis, BasisError, SYMBOL_TO_Z

SCF_NOT_RUN, SCF_CONVERGED, SCF_NOT_CONVERGED = capi.SCF_NOT_RUN, capi.SCF_CONVERGED, capi.SCF_NOT_CONVERGED
BACKEND_AUTO, BACKEND_HIP, BACKEND_LIBCINT = 0, 1, 2


def parse_backend_name(name: str) -> int:
    """auto|hip|gpu|cuest -> this engine; libcint|cpu -> the CPU backend (not part of this repo).
    Mirrors parse_backend_name, src/methods/mqc_cuest_iface.f90:146-180."""
    n = (name or "auto").strip().lower()
    if n in ("auto",):
        return BACKEND_AUTO
    if n in ("hip", "gpu", "cuest", "mi355x"):
        return BACKEND_HIP
    if n in ("libcint", "cpu"):
        return BACKEND_LIBCINT
    raise ValueError("unknown backend '%s'" % name)


@dataclass
class PhysicalFragment:
    element_numbers: np.ndarray           # (n_atoms,) int
    coordinates: np.ndarray               # (3, n_atoms) Bohr, column per atom like the reference
    charge: int = 0
    multiplicity: int = 1
    ghost: Optional[np.ndarray] = None    # (n_atoms,) bool
    nelec: Optional[int] = None

    def __post_init__(self):
        self.element_numbers = np.asarray(self.element_numbers, dtype=np.int32)
        self.coordinates = np.asarray(self.coordinates, dtype=np.float64)
        if self.coordinates.shape != (3, len(self.element_numbers)):
            raise ValueError("coordinates must be (3, n_atoms) in Bohr")
        if self.ghost is not None:
            self.ghost = np.asarray(self.ghost, dtype=bool)
        if self.nelec is None:
            real = self.element_numbers if self.ghost is None else self.element_numbers[~self.ghost]
            self.nelec = int(np.sum(real)) - int(self.charge)     # compute_nelec, :73-83

    @property
    def n_atoms(self) -> int:
        return int(len(self.element_numbers))

    @classmethod
    def from_angstrom(cls, symbols: Sequence[str], xyz_angstrom, **kw) -> "PhysicalFragment":
        z = [SYMBOL_TO_Z[s.lower()] for s in symbols]
        xyz = np.asarray(xyz_angstrom, dtype=np.float64).reshape(-1, 3) * ANGSTROM_TO_BOHR
        return cls(np.array(z), xyz.T.copy(), **kw)


@dataclass
class ScfSettings:
    basis_set: str = "sto-3g"
    aux_basis_set: str = "def2-universal-jkfit"
    density_fitting: bool = False
    functional: str = ""
    spherical: bool = True
    verbose: bool = False
    guess: str = "auto"
    device_rank: int = 0
    unrestricted: bool = False
    allow_crap_scf: bool = False
    max_iter: int = 100
    energy_tol: float = 1.0e-8
    density_tol: float = 1.0e-6
    use_diis: bool = True
    diis_size: int = 8
    grid_level: int = 3
    radial_points: int = 0
    angular_points: int = 0
    backend: int = BACKEND_AUTO
    schwarz_tol: float = 0.0
    eri_mode: str = "auto"          # auto | incore | direct


@dataclass
class EnergyT:
    scf: float = 0.0

    def total(self) -> float:
        return self.scf


@dataclass
class CalculationResult:
    energy: EnergyT = field(default_factory=EnergyT)
    has_energy: bool = False
    scf_status: int = SCF_NOT_RUN
    scf_iterations: int = 0
    homo: float = 0.0
    lumo: float = 0.0
    has_orbitals: bool = False
    has_error: bool = False
    error_code: int = 0
    error_message: str = ""
    distance: float = 0.0
    e_nuclear: float = 0.0
    e_electronic: float = 0.0
    orbital_energies: Optional[np.ndarray] = None
    dipole: Optional[np.ndarray] = None     # (3,) electron-Bohr, origin = centre of nuclear charge
    has_dipole: bool = False
    gradient: Optional[np.ndarray] = None   # (3, n_atoms) Hartree/Bohr, like result%gradient
    has_gradient: bool = False
    s_squared: float = 0.0
    orbital_energies_beta: Optional[np.ndarray] = None
    n_alpha: int = 0
    n_beta: int = 0
    hessian: Optional[np.ndarray] = None               # (3 n_atoms, 3 n_atoms) Hartree/Bohr^2, atom-major like result%hessian
    has_hessian: bool = False
    dipole_derivatives: Optional[np.ndarray] = None    # (3, 3 n_atoms) d mu / d R
    has_dipole_derivatives: bool = False


DEFAULT_DISPLACEMENT = 0.005         # Bohr, src/core/mqc_calculation_defaults.f90:13

_GUESS = {"auto": capi.GUESS_AUTO, "gwh": capi.GUESS_GWH, "core": capi.GUESS_CORE, "sad": capi.GUESS_SAD, "sac": capi.GUESS_SAC}


def hip_backend_available() -> bool:
    try:
        return bool(capi.load_library().mqc_hip_backend_available())
    except capi.HipBackendError:
        return False


def _options(settings: ScfSettings, want_gradient: bool) -> capi.ScfOptions:
    o = capi.default_options()
    o.functional = settings.functional.encode()[:31]
    o.density_fitting = int(settings.density_fitting)
    o.grid_level = settings.grid_level
    o.radial_points = settings.radial_points
    o.angular_points = settings.angular_points
    o.max_iter = settings.max_iter
    o.energy_tol = settings.energy_tol
    o.density_tol = settings.density_tol
    o.use_diis = int(settings.use_diis)
    o.diis_size = settings.diis_size
    g = settings.guess.strip().lower()
    if g not in _GUESS:
        # the cuEST path refuses guesses it does not implement rather than running another one
        raise capi.HipBackendError(capi.ERR_UNSUPPORTED, "initial guess '%s' is not available on the HIP backend "
                                                         "(core, gwh, auto, sad, sac)" % settings.guess)
    o.guess = _GUESS[g]
    o.unrestricted = int(settings.unrestricted)
    o.want_gradient = int(want_gradient)
    o.allow_crap_scf = int(settings.allow_crap_scf)
    o.verbose = int(settings.verbose)
    o.schwarz_tol = settings.schwarz_tol
    o.eri_mode = {"auto": capi.ERI_AUTO, "incore": capi.ERI_INCORE, "direct": capi.ERI_DIRECT}[settings.eri_mode.lower()]
    return o


class _Marshalled:
    """Keeps the numpy arrays alive for as long as the C structs that point into them."""

    def __init__(self, fragment: PhysicalFragment, fb: FlatBasis, aux: Optional[FlatBasis] = None):
        self.z = np.ascontiguousarray(fragment.element_numbers, dtype=np.int32)
        self.xyz = np.ascontiguousarray(fragment.coordinates.T, dtype=np.float64)   # atom-major
        self.ghost = None if fragment.ghost is None else np.ascontiguousarray(fragment.ghost, dtype=np.uint8)
        self.fb = fb
        self.mol = capi.Molecule(fragment.n_atoms, self.z.ctypes.data_as(capi.c_int32_p), capi.dptr(self.xyz),
                                 None if self.ghost is None else self.ghost.ctypes.data_as(capi.c_uint8_p),
                                 int(fragment.charge), int(fragment.multiplicity), int(fragment.nelec))
        self.bas = capi.Basis(1 if fb.spherical else 0, fragment.n_atoms,
                              fb.nshell_per_atom.ctypes.data_as(capi.c_int64_p), fb.nshell,
                              fb.shell_l.ctypes.data_as(capi.c_int32_p), fb.shell_nprim.ctypes.data_as(capi.c_int32_p),
                              capi.dptr(fb.exps), capi.dptr(fb.coefs))
        self.aux = aux
        self.aux_bas = None
        if aux is not None:
            self.aux_bas = capi.Basis(1 if aux.spherical else 0, fragment.n_atoms,
                                      aux.nshell_per_atom.ctypes.data_as(capi.c_int64_p), aux.nshell,
                                      aux.shell_l.ctypes.data_as(capi.c_int32_p), aux.shell_nprim.ctypes.data_as(capi.c_int32_p),
                                      capi.dptr(aux.exps), capi.dptr(aux.coefs))


_BASIS_CACHE = {}


def _flat_basis_z(name: str, element_numbers) -> FlatBasis:
    # flattened shells cached per (basis, element sequence): SURVEY.md 8f item 4
    key = (name.lower(), tuple(int(z) for z in element_numbers))
    if key not in _BASIS_CACHE:
        _BASIS_CACHE[key] = build_flat_basis(name, element_numbers)
    return _BASIS_CACHE[key]


def _flat_basis(name: str, fragment: PhysicalFragment) -> FlatBasis:
    return _flat_basis_z(name, fragment.element_numbers)


def _fill(result: CalculationResult, r: capi.ScfResult, eps: Optional[np.ndarray]) -> CalculationResult:
    result.scf_status = int(r.scf_status)
    result.scf_iterations = int(r.iterations)
    if r.has_error:
        # driver: result%error%set(...), has_error = .true., has_energy = .false. (:385-393)
        result.has_error = True
        result.error_code = capi.ERR_GENERIC
        result.error_message = r.message.decode(errors="replace")
        result.has_energy = False
        return result
    result.energy.scf = float(r.e_total)
    result.e_nuclear = float(r.e_nuclear)
    result.e_electronic = float(r.e_electronic)
    result.has_energy = True
    result.homo = float(r.homo)
    result.lumo = float(r.lumo)
    result.has_orbitals = bool(r.has_orbitals)
    if eps is not None:
        result.orbital_energies = eps[: int(r.n_mo)].copy()
    if r.has_dipole:
        result.dipole = np.array([r.dipole[0], r.dipole[1], r.dipole[2]])
        result.has_dipole = True
    result.s_squared = float(r.s_squared)
    result.n_alpha = int(r.n_alpha); result.n_beta = int(r.n_beta)
    return result


def run_hip_scf(settings: ScfSettings, fragment: PhysicalFragment, result: Optional[CalculationResult] = None,
                want_gradient: bool = False) -> CalculationResult:
    """One fragment through the engine: the drop-in for run_cuest_scf."""
    result = result if result is not None else CalculationResult()
    try:
        lib = capi.load_library()
        ctx = capi.get_context(settings.device_rank)
        opts = _options(settings, want_gradient)
        fb = _flat_basis(settings.basis_set, fragment)
        aux = _flat_basis(settings.aux_basis_set, fragment) if settings.density_fitting else None
        m = _Marshalled(fragment, fb, aux)
        eps = np.zeros(fb.nao)
        epsb = np.zeros(fb.nao)
        r = capi.ScfResult()
        r.orbital_energies = capi.dptr(eps)
        r.orbital_energies_beta = capi.dptr(epsb)
        grad = np.zeros((fragment.n_atoms, 3)) if want_gradient else None
        if grad is not None:
            r.gradient = capi.dptr(grad)
        rc = lib.mqc_hip_scf_run(ctx, C.byref(m.mol), C.byref(m.bas), C.byref(m.aux_bas) if aux is not None else None,
                                 C.byref(opts), C.byref(r))
        if rc != capi.MQC_HIP_OK and not r.has_error:
            capi.check(rc)
        _fill(result, r, eps)
        if grad is not None and r.has_gradient and not r.has_error:
            result.gradient = grad.T.copy()              # (3, n_atoms) like result%gradient
            result.has_gradient = True
        if result.has_energy and (int(r.n_alpha) != int(r.n_beta) or settings.unrestricted):
            result.orbital_energies_beta = epsb[: int(r.n_mo)].copy()
        return result
    except (capi.HipBackendError, BasisError) as e:
        result.has_error = True
        result.has_energy = False
        result.error_code = getattr(e, "code", capi.ERR_VALIDATION)
        result.error_message = str(getattr(e, "message", e))
        return result


_MOL_DTYPE = np.dtype({"names": ["n_atoms", "atomic_numbers", "xyz", "ghost", "charge", "multiplicity", "nelec",
                                 "n_point_charges", "point_charge_xyz", "point_charges", "h_extra"],
                       "formats": ["<i4", "<u8", "<u8", "<u8", "<i4", "<i4", "<i4", "<i4", "<u8", "<u8", "<u8"],
                       "offsets": [0, 8, 16, 24, 32, 36, 40, 44, 48, 56, 64], "itemsize": C.sizeof(capi.Molecule)})
_BAS_DTYPE = np.dtype({"names": ["spherical", "n_atoms", "nshell_per_atom", "n_shells", "shell_l", "shell_nprim",
                                 "exponents", "coefficients"],
                       "formats": ["<i4", "<i4", "<u8", "<i4", "<u8", "<u8", "<u8", "<u8"],
                       "offsets": [0, 4, 8, 16, 24, 32, 40, 48], "itemsize": C.sizeof(capi.Basis)})


def _basis_record(fb: FlatBasis, n_atoms: int):
    return (1 if fb.spherical else 0, n_atoms, fb.nshell_per_atom.ctypes.data, fb.nshell, fb.shell_l.ctypes.data,
            fb.shell_nprim.ctypes.data, fb.exps.ctypes.data, fb.coefs.ctypes.data)


def _struct_dtype(struct) -> np.dtype:
    """numpy view of a ctypes struct (pointers as u8, char arrays as bytes) with the C offsets."""
    names, formats, offsets = [], [], []
    for name, ctype in struct._fields_:
        size = C.sizeof(ctype)
        if ctype is C.c_double:
            fmt = "<f8"
        elif ctype is C.c_int32:
            fmt = "<i4"
        elif ctype is C.c_int64:
            fmt = "<i8"
        elif issubclass(ctype, C.Array) and ctype._type_ is C.c_char:
            fmt = "S%d" % size
        elif issubclass(ctype, C.Array) and ctype._type_ is C.c_double:
            fmt = ("<f8", (ctype._length_,))
        else:
            fmt = "<u8"          # pointers
        names.append(name); formats.append(fmt); offsets.append(getattr(struct, name).offset)
    return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": C.sizeof(struct)})


_RES_DTYPE = _struct_dtype(capi.ScfResult)


@dataclass
class FragmentGroup:
    """Fragments that share their element sequence: the unit the engine batches over."""
    element_numbers: np.ndarray            # (n_atoms,)
    xyz: np.ndarray                        # (m, n_atoms, 3) Bohr
    charge: np.ndarray                     # (m,)
    multiplicity: Optional[np.ndarray] = None
    ghost: Optional[np.ndarray] = None     # (n_atoms,) bool, same for the whole group
    nelec: Optional[np.ndarray] = None     # (m,), default sum(Z of real atoms) - charge
    point_charge_xyz: Optional[np.ndarray] = None   # (m, n_pc, 3) Bohr: the embedding field of an FMO / EE-MBE fragment
    point_charges: Optional[np.ndarray] = None      # (m, n_pc)
    h_extra: Optional[np.ndarray] = None            # (m, n_ao, n_ao): a further one-electron operator added to H (run_libcint_rhf's h_extra)


def run_hip_scf_groups(settings: ScfSettings, groups: Sequence[FragmentGroup], want_gradient: bool = False,
                       gradients_out: Optional[list] = None, extras: Sequence[str] = (),
                       extras_out: Optional[list] = None) -> List[np.ndarray]:
    """All fragments of all groups in ONE mqc_hip_scf_run_batch call; returns, per group, a structured array
    viewing the engine's result records (fields of capi.ScfResult: e_total, iterations, has_error, message ...).

    The C structs are filled as numpy structured arrays with the header's exact layout
    (tests/test_host_logic.py checks the sizes), so marshalling a few thousand fragments is a
    handful of vector operations rather than a Python loop over ctypes objects.

    `extras` names per-fragment arrays to bring back besides the records -- "density" (m, n, n), "embedding_matrix"
    (m, n, n), "mulliken_charges" (m, n_atoms) -- appended to `extras_out` as one dict per group."""
    sizes = [int(g.xyz.shape[0]) for g in groups]
    n = int(sum(sizes))
    if n == 0:
        return [np.zeros(0, dtype=_RES_DTYPE) for _ in groups]
    lib = capi.load_library()
    ctx = capi.get_context(settings.device_rank)
    opts = _options(settings, want_gradient)
    df = settings.density_fitting
    mols = np.zeros(n, dtype=_MOL_DTYPE)
    bass = np.zeros(n, dtype=_BAS_DTYPE)
    auxs = np.zeros(n, dtype=_BAS_DTYPE) if df else None
    keep = []                                   # arrays the structs point into
    lo = 0
    for g, m in zip(groups, sizes):
        sl = slice(lo, lo + m)
        lo += m
        if m == 0:
            continue
        z = np.ascontiguousarray(g.element_numbers, dtype=np.int32)
        na = int(len(z))
        ghost = None if g.ghost is None else np.ascontiguousarray(g.ghost, dtype=np.uint8)
        xyz = np.ascontiguousarray(g.xyz, dtype=np.float64)
        if xyz.shape != (m, na, 3):
            raise ValueError("FragmentGroup.xyz must be (m, n_atoms, 3) in Bohr")
        charge = np.broadcast_to(np.asarray(g.charge, dtype=np.int32), (m,))
        if g.nelec is None:
            zsum = int(np.sum(z if ghost is None else z[ghost == 0]))
            nelec = zsum - charge                                        # compute_nelec, :73-83
        else:
            nelec = np.asarray(g.nelec, dtype=np.int32)
        fb = _flat_basis_z(settings.basis_set, z)
        keep += [z, ghost, xyz, fb]
        mols["n_atoms"][sl] = na
        mols["atomic_numbers"][sl] = z.ctypes.data
        mols["xyz"][sl] = xyz.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(na * 3 * 8)
        mols["ghost"][sl] = 0 if ghost is None else ghost.ctypes.data
        mols["charge"][sl] = charge
        mols["multiplicity"][sl] = 1 if g.multiplicity is None else np.asarray(g.multiplicity, dtype=np.int32)
        mols["nelec"][sl] = nelec
        if g.point_charges is not None:
            pq = np.ascontiguousarray(g.point_charges, dtype=np.float64)
            px = np.ascontiguousarray(g.point_charge_xyz, dtype=np.float64)
            npc = int(pq.shape[1])
            if pq.shape != (m, npc) or px.shape != (m, npc, 3):
                raise ValueError("FragmentGroup.point_charges must be (m, n_pc) with point_charge_xyz (m, n_pc, 3)")
            keep += [pq, px]
            mols["n_point_charges"][sl] = npc
            mols["point_charges"][sl] = pq.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(npc * 8)
            mols["point_charge_xyz"][sl] = px.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(npc * 3 * 8)
        if g.h_extra is not None:
            hx = np.ascontiguousarray(g.h_extra, dtype=np.float64)
            if hx.shape != (m, fb.nao, fb.nao):
                raise ValueError("FragmentGroup.h_extra must be (m, n_ao, n_ao)")
            keep.append(hx)
            mols["h_extra"][sl] = hx.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(fb.nao * fb.nao * 8)
        bass[sl] = _basis_record(fb, na)
        if df:
            ab = _flat_basis_z(settings.aux_basis_set, z)
            keep.append(ab)
            auxs[sl] = _basis_record(ab, na)
    res = (capi.ScfResult * n)()
    grads = []
    if want_gradient:
        # one (m, n_atoms, 3) array per group; every record points at its own slice
        rec0 = np.frombuffer(res, dtype=_RES_DTYPE)
        lo = 0
        for g, m in zip(groups, sizes):
            na = int(len(g.element_numbers))
            ga = np.zeros((m, na, 3))
            grads.append(ga)
            if m:
                rec0["gradient"][lo:lo + m] = ga.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(na * 3 * 8)
            lo += m
        if gradients_out is not None:
            gradients_out.extend(grads)
    if extras:
        rec0 = np.frombuffer(res, dtype=_RES_DTYPE)
        lo = 0
        for g, m in zip(groups, sizes):
            z = np.ascontiguousarray(g.element_numbers, dtype=np.int32)
            nao = _flat_basis_z(settings.basis_set, z).nao
            got = {}
            for name in extras:
                if name not in ("density", "embedding_matrix", "mulliken_charges"):
                    raise ValueError("unknown extra output " + name)
                width = len(z) if name == "mulliken_charges" else nao * nao
                arr = np.zeros((m, width))
                if m:
                    rec0[name][lo:lo + m] = arr.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(width * 8)
                got[name] = arr if name == "mulliken_charges" else arr.reshape(m, nao, nao)
            lo += m
            if extras_out is not None:
                extras_out.append(got)
            keep.append(got)
    rc = lib.mqc_hip_scf_run_batch(ctx, n, mols.ctypes.data_as(C.POINTER(capi.Molecule)),
                                   bass.ctypes.data_as(C.POINTER(capi.Basis)),
                                   auxs.ctypes.data_as(C.POINTER(capi.Basis)) if df else None, C.byref(opts), res)
    rec = np.frombuffer(res, dtype=_RES_DTYPE)
    if rc != capi.MQC_HIP_OK and not np.any(rec["has_error"]):
        capi.check(rc)
    del keep
    out, lo = [], 0
    for m in sizes:
        out.append(rec[lo:lo + m])
        lo += m
    return out


def run_hip_scf_batch(settings: ScfSettings, fragments: Sequence[PhysicalFragment]) -> List[CalculationResult]:
    """Many independent fragments in one call (mqc_hip_scf_run_batch), results as CalculationResult objects."""
    n = len(fragments)
    results = [CalculationResult() for _ in range(n)]
    if n == 0:
        return results
    by_key = {}
    for i, f in enumerate(fragments):
        key = (f.element_numbers.tobytes(), None if f.ghost is None else f.ghost.tobytes())
        by_key.setdefault(key, []).append(i)
    groups, index = [], []
    for idx in by_key.values():
        f0 = fragments[idx[0]]
        groups.append(FragmentGroup(f0.element_numbers, np.stack([fragments[i].coordinates.T for i in idx]),
                                    np.array([fragments[i].charge for i in idx], dtype=np.int32),
                                    np.array([fragments[i].multiplicity for i in idx], dtype=np.int32),
                                    f0.ghost, np.array([fragments[i].nelec for i in idx], dtype=np.int32)))
        index.append(idx)
    for idx, rec in zip(index, run_hip_scf_groups(settings, groups)):
        for i, r in zip(idx, rec):
            _fill_record(results[i], r)
    return results


def _fill_record(result: CalculationResult, r) -> CalculationResult:
    """_fill for one row of the structured result view."""
    result.scf_status = int(r["scf_status"])
    result.scf_iterations = int(r["iterations"])
    if r["has_error"]:
        result.has_error = True
        result.error_code = capi.ERR_GENERIC
        result.error_message = bytes(r["message"]).split(b"\0", 1)[0].decode(errors="replace")
        result.has_energy = False
        return result
    result.energy.scf = float(r["e_total"])
    result.e_nuclear = float(r["e_nuclear"])
    result.e_electronic = float(r["e_electronic"])
    result.has_energy = True
    result.homo = float(r["homo"])
    result.lumo = float(r["lumo"])
    result.has_orbitals = bool(r["has_orbitals"])
    if r["has_dipole"]:
        result.dipole = np.array(r["dipole"], dtype=float)
        result.has_dipole = True
    result.s_squared = float(r["s_squared"])
    return result


class HFMethod:
    """qc_method_t for Hartree-Fock (src/methods/mqc_method_base.f90:13-60, mqc_method_hf.F90)."""

    def __init__(self, settings: Optional[ScfSettings] = None, **kw):
        self.settings = settings or ScfSettings(**kw)

    def calc_energy(self, fragment: PhysicalFragment, result: Optional[CalculationResult] = None) -> CalculationResult:
        backend = self.settings.backend
        if backend == BACKEND_LIBCINT:
            r = result or CalculationResult()
            r.has_error = True
            r.error_message = "the libcint CPU backend is not part of this build; use backend hip|auto"
            return r
        return run_hip_scf(self.settings, fragment, result)

    def calc_gradient(self, fragment, result=None):
        """hf_calc_gradient (src/methods/mqc_method_hf.F90): the SCF with want_gradient, result%gradient (3, n_atoms)."""
        return run_hip_scf(self.settings, fragment, result, want_gradient=True)

    def calc_hessian(self, fragment, result=None):
        """hf_calc_hessian -> finite_difference_hessian (src/methods/mqc_method_hf.F90:228-257,
        src/methods/mqc_semi_numerical_hessian.f90): central differences of analytic gradients at +-0.005 Bohr,
        H[i, j] = (g_j(x_i + h) - g_j(x_i - h)) / 2h, symmetrised (finite_diff_hessian_from_gradients,
        src/utils/mqc_finite_differences.f90); the dipoles of the same displacements give d mu / d R; the undisplaced
        point supplies energy, gradient and dipole.  The reference walks the 6 N + 1 geometries one SCF at a time; here
        they are ONE batch of the engine (one topology group)."""
        h = DEFAULT_DISPLACEMENT
        na = len(fragment.element_numbers)
        base = np.asarray(fragment.coordinates, dtype=float)              # (3, n_atoms)
        coords = [base.T.copy()]
        for a in range(na):
            for c in range(3):
                for sgn in (+1.0, -1.0):
                    x = base.T.copy(); x[a, c] += sgn * h
                    coords.append(x)
        m = len(coords)
        group = FragmentGroup(fragment.element_numbers, np.stack(coords), np.full(m, fragment.charge, dtype=np.int32),
                              np.full(m, fragment.multiplicity, dtype=np.int32), fragment.ghost, np.full(m, fragment.nelec, dtype=np.int32))
        grads: list = []
        rec = run_hip_scf_groups(self.settings, [group], want_gradient=True, gradients_out=grads)[0]
        r = result or CalculationResult()
        bad = [k for k in range(m) if rec["has_error"][k] or not rec["has_gradient"][k]]
        if bad:
            k = bad[0]
            r.has_error = True
            r.error_message = ("Hessian: the gradient at displaced geometry %d failed: " % k) + \
                bytes(rec["message"][k]).split(b"\0", 1)[0].decode(errors="replace")
            return r
        _fill_record(r, rec[0])
        g = grads[0]                                                        # (m, n_atoms, 3)
        r.gradient = g[0].T.copy(); r.has_gradient = True
        hess = np.zeros((3 * na, 3 * na))
        dmu = np.zeros((3, 3 * na))
        have_dip = all(bool(rec["has_dipole"][k]) for k in range(m))
        for i in range(3 * na):
            fwd, bwd = 1 + 2 * i, 2 + 2 * i
            hess[i, :] = (g[fwd] - g[bwd]).reshape(-1) / (2.0 * h)
            if have_dip:
                dmu[:, i] = (np.array(rec["dipole"][fwd]) - np.array(rec["dipole"][bwd])) / (2.0 * h)
        r.hessian = 0.5 * (hess + hess.T); r.has_hessian = True
        if have_dip:
            r.dipole_derivatives = dmu; r.has_dipole_derivatives = True
        return r


def get_stats() -> capi.Stats:
    st = capi.Stats()
    capi.check(capi.load_library().mqc_hip_get_stats(capi.get_context(), C.byref(st)))
    return st

"""ctypes binding of include/mqc_hip.h -- the reference-side stub a maintainer would write
if metalquicha's Python front end (python/mqc, itself a ctypes wrapper) called the engine
directly.  Field order and types mirror the header exactly; tests/test_capi_symbols.py
checks every declared symbol is exported.

The library is in-tree (metalquicha_amd/libmqc_hip.so, built by __graft_entry__.build()).
There is NO Python or CPU implementation behind these calls: if the library is missing or
no HIP device is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# MQC_HIP_LIBRARY: another build of the SAME library (instrumented A/B builds under profiling); never a different backend
LIB_PATH = os.environ.get("MQC_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmqc_hip.so")

MQC_HIP_OK = 0
ERR_VALIDATION, ERR_GENERIC, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_DEVICE = 1, 2, 3, 4, 5
SCF_NOT_RUN, SCF_CONVERGED, SCF_NOT_CONVERGED = 0, 1, 2
GUESS_AUTO, GUESS_CORE, GUESS_GWH, GUESS_SAD, GUESS_SAC = 0, 1, 2, 3, 4
ERI_AUTO, ERI_INCORE, ERI_DIRECT = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class Molecule(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("atomic_numbers", c_int32_p), ("xyz", c_double_p),
                ("ghost", c_uint8_p), ("charge", C.c_int32), ("multiplicity", C.c_int32),
                ("nelec", C.c_int32),
                # ABI 3: external point charges (embedding field of the FMO / EE-MBE callers)
                ("n_point_charges", C.c_int32), ("point_charge_xyz", c_double_p), ("point_charges", c_double_p),
                ("h_extra", c_double_p)]


class Basis(C.Structure):
    _fields_ = [("spherical", C.c_int32), ("n_atoms", C.c_int32), ("nshell_per_atom", c_int64_p),
                ("n_shells", C.c_int32), ("shell_l", c_int32_p), ("shell_nprim", c_int32_p),
                ("exponents", c_double_p), ("coefficients", c_double_p)]


class ScfOptions(C.Structure):
    _fields_ = [("functional", C.c_char * 32), ("density_fitting", C.c_int32), ("grid_level", C.c_int32),
                ("radial_points", C.c_int32), ("angular_points", C.c_int32), ("max_iter", C.c_int32),
                ("energy_tol", C.c_double), ("density_tol", C.c_double), ("use_diis", C.c_int32),
                ("diis_size", C.c_int32), ("guess", C.c_int32), ("unrestricted", C.c_int32),
                ("want_gradient", C.c_int32), ("allow_crap_scf", C.c_int32), ("verbose", C.c_int32),
                ("eri_mode", C.c_int32), ("schwarz_tol", C.c_double)]


class ScfResult(C.Structure):
    _fields_ = [("e_total", C.c_double), ("e_electronic", C.c_double), ("e_nuclear", C.c_double),
                ("e_xc", C.c_double), ("scf_status", C.c_int32), ("iterations", C.c_int32),
                ("n_ao", C.c_int32), ("n_mo", C.c_int32), ("n_occ", C.c_int32), ("homo", C.c_double),
                ("lumo", C.c_double), ("has_orbitals", C.c_int32), ("orbital_energies", c_double_p),
                ("density", c_double_p), ("has_error", C.c_int32), ("message", C.c_char * 256),
                ("dipole", C.c_double * 3), ("has_dipole", C.c_int32), ("gradient", c_double_p), ("has_gradient", C.c_int32),
                ("orbital_energies_beta", c_double_p), ("n_alpha", C.c_int32), ("n_beta", C.c_int32), ("s_squared", C.c_double),
                ("e_embedding", C.c_double), ("embedding_matrix", c_double_p), ("mulliken_charges", c_double_p)]


class Stats(C.Structure):
    _fields_ = [("t_setup", C.c_double), ("t_int1e", C.c_double), ("t_eri", C.c_double),
                ("t_fock", C.c_double), ("t_scf_step", C.c_double), ("t_total", C.c_double),
                ("fock_launches", C.c_int64), ("eri_quartets", C.c_int64),
                ("scf_iterations_total", C.c_int64), ("fock_kernel_seconds", C.c_double),
                ("fock_bytes", C.c_double), ("eri_kernel_seconds", C.c_double),
                ("xc_kernel_seconds", C.c_double), ("xc_points", C.c_double),
                ("fock_big_launches", C.c_int64), ("fock_big_seconds", C.c_double), ("fock_big_bytes", C.c_double),
                ("xc_flops", C.c_double), ("scf_step_seconds", C.c_double), ("eri_survivors", C.c_int64),
                ("df_flops", C.c_double), ("df_bytes", C.c_double)]


DECLARED_SYMBOLS = [
    "mqc_hip_backend_available", "mqc_hip_context_get", "mqc_hip_finalize", "mqc_hip_last_error",
    "mqc_hip_abi_version", "mqc_hip_default_options", "mqc_hip_scf_run", "mqc_hip_scf_run_batch",
    "mqc_hip_int1e", "mqc_hip_eri_packed", "mqc_hip_jk_incore", "mqc_hip_coulomb_batch", "mqc_hip_syev",
    "mqc_hip_diis_coefficients", "mqc_hip_get_stats", "mqc_hip_device_name",
]

_lib = None



class HipBackendError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mqc_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


def load_library():
    """dlopen the in-tree library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipBackendError(ERR_GENERIC, "libmqc_hip.so is not built (run __graft_entry__.build()); "
                                           "there is no fallback implementation")
    lib = C.CDLL(LIB_PATH)
    lib.mqc_hip_last_error.restype = C.c_char_p
    lib.mqc_hip_context_get.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    lib.mqc_hip_default_options.argtypes = [C.POINTER(ScfOptions)]
    lib.mqc_hip_default_options.restype = None
    lib.mqc_hip_scf_run.argtypes = [C.c_void_p, C.POINTER(Molecule), C.POINTER(Basis), C.POINTER(Basis),
                                    C.POINTER(ScfOptions), C.POINTER(ScfResult)]
    lib.mqc_hip_scf_run_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Molecule), C.POINTER(Basis),
                                          C.POINTER(Basis), C.POINTER(ScfOptions), C.POINTER(ScfResult)]
    lib.mqc_hip_int1e.argtypes = [C.c_void_p, C.POINTER(Molecule), C.POINTER(Basis), c_double_p, c_double_p, c_double_p]
    lib.mqc_hip_eri_packed.argtypes = [C.c_void_p, C.POINTER(Molecule), C.POINTER(Basis), C.c_double, c_double_p]
    lib.mqc_hip_jk_incore.argtypes = [C.c_void_p, C.POINTER(Molecule), C.POINTER(Basis), c_double_p, c_double_p, c_double_p]
    lib.mqc_hip_coulomb_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Molecule), C.POINTER(Basis), C.c_int32, c_double_p, c_double_p]
    lib.mqc_hip_syev.argtypes = [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_double_p]
    lib.mqc_hip_diis_coefficients.argtypes = [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_int32_p]
    lib.mqc_hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.mqc_hip_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    _lib = lib
    return lib


def check(rc):
    if rc != MQC_HIP_OK:
        raise HipBackendError(rc, load_library().mqc_hip_last_error().decode(errors="replace"))


_ctx = None


def get_context(local_rank: int = 0):
    """Process-wide context (get_cuest_context): device = local_rank mod device_count."""
    global _ctx
    if _ctx is None:
        lib = load_library()
        h = C.c_void_p()
        check(lib.mqc_hip_context_get(C.c_int32(local_rank), C.byref(h)))
        _ctx = h
    return _ctx


def finalize():
    global _ctx
    if _lib is not None:
        _lib.mqc_hip_finalize()
    _ctx = None


def dptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def default_options() -> ScfOptions:
    o = ScfOptions()
    load_library().mqc_hip_default_options(C.byref(o))
    return o

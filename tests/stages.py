"""Stage-level entry points of the C ABI (mqc_hip_int1e, _eri_packed, _jk_incore, _syev,
_diis_coefficients) as numpy-in / numpy-out functions.  They run the same kernels the SCF
driver launches and exist so that each row of the hot-path table can be parity-tested alone.
Test infrastructure: a ctypes helper for tests/, not part of the product package."""
from __future__ import annotations

import ctypes as C

import numpy as np

from metalquicha_amd import capi
from metalquicha_amd.methods import PhysicalFragment, _Marshalled, _flat_basis


def _marshal(basis_set: str, fragment: PhysicalFragment):
    return _Marshalled(fragment, _flat_basis(basis_set, fragment))


def int1e(basis_set: str, fragment: PhysicalFragment):
    m = _marshal(basis_set, fragment)
    n = m.fb.nao
    S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    capi.check(capi.load_library().mqc_hip_int1e(capi.get_context(), C.byref(m.mol), C.byref(m.bas),
                                                 capi.dptr(S), capi.dptr(T), capi.dptr(V)))
    return S, T, V


def eri_packed(basis_set: str, fragment: PhysicalFragment, schwarz_tol: float = 0.0) -> np.ndarray:
    m = _marshal(basis_set, fragment)
    n = m.fb.nao
    npair = n * (n + 1) // 2
    M = np.zeros((npair, npair))
    capi.check(capi.load_library().mqc_hip_eri_packed(capi.get_context(), C.byref(m.mol), C.byref(m.bas),
                                                      C.c_double(schwarz_tol), capi.dptr(M)))
    return M


def jk_incore(basis_set: str, fragment: PhysicalFragment, D: np.ndarray):
    m = _marshal(basis_set, fragment)
    n = m.fb.nao
    D = np.ascontiguousarray(D, dtype=np.float64)
    J, K = np.zeros((n, n)), np.zeros((n, n))
    capi.check(capi.load_library().mqc_hip_jk_incore(capi.get_context(), C.byref(m.mol), C.byref(m.bas),
                                                     capi.dptr(D), capi.dptr(J), capi.dptr(K)))
    return J, K


def syev(A: np.ndarray):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    w, V = np.zeros(n), np.zeros((n, n))
    capi.check(capi.load_library().mqc_hip_syev(capi.get_context(), n, capi.dptr(A), capi.dptr(w), capi.dptr(V)))
    return w, V


def diis_coefficients(overlap: np.ndarray):
    B = np.ascontiguousarray(overlap, dtype=np.float64)
    n = B.shape[0]
    c = np.zeros(n)
    ok = C.c_int32(0)
    capi.check(capi.load_library().mqc_hip_diis_coefficients(capi.get_context(), n, capi.dptr(B), capi.dptr(c), C.byref(ok)))
    return c, bool(ok.value)


def pack_eri(eri4: np.ndarray) -> np.ndarray:
    """Full (n,n,n,n) tensor -> the engine's pair matrix layout, for comparisons."""
    n = eri4.shape[0]
    idx = [(i, j) for i in range(n) for j in range(i + 1)]
    ii = np.array([p[0] for p in idx]); jj = np.array([p[1] for p in idx])
    return eri4[ii[:, None], jj[:, None], ii[None, :], jj[None, :]]


def coulomb_batch(basis_set: str, fragments, D: np.ndarray, n_source_atoms: int = 0) -> np.ndarray:
    """mqc_hip_coulomb_batch for fragments of one element sequence: D (m, n, n) -> J (m, n, n)."""
    ms = [_marshal(basis_set, f) for f in fragments]
    mols = (capi.Molecule * len(ms))(*[m.mol for m in ms])
    D = np.ascontiguousarray(D, dtype=np.float64)
    J = np.zeros_like(D)
    capi.check(capi.load_library().mqc_hip_coulomb_batch(capi.get_context(), len(ms), mols, C.byref(ms[0].bas),
                                                         n_source_atoms, capi.dptr(D), capi.dptr(J)))
    return J

"""CPU oracle for the per-fragment SCF hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the shipped path (metalquicha_amd + libmqc_hip.so) never does.

It is a numpy restatement of the reference's libcint CPU path, function by function
(all citations relative to /root/reference):

  normalise_basis      backends/libcint/mqc_libcint_integrals.F90:519-555
  nuclear_repulsion    backends/libcint/mqc_libcint_integrals.F90:481-488 (ghosts carry Z=0)
  build_orthogonalizer src/scf/mqc_scf_common.f90:43-83      (canonical, eig > 1e-7)
  guess_fock_gwh       backends/libcint/mqc_libcint_rhf.f90:1354-1380 (K = 1.75)
  Diis                 src/methods/mqc_diis.f90:91-273       (ring, age-ordered B, max|B|
                                                               scaling, pivoted elimination)
  commutator           backends/libcint/mqc_libcint_rhf.f90:1326-1352
  diagonalize          backends/libcint/mqc_libcint_rhf.f90:1464-1489
  build_fock_incore    backends/libcint/mqc_libcint_rhf.f90:1491-1574
  build_fock_df        backends/libcint/mqc_libcint_rhf.f90:1576-1646
  df_tensor            backends/libcint/mqc_libcint_integrals.F90:913-1038 (eigen cut 1e-10)
  run_rhf              backends/libcint/mqc_libcint_rhf.f90:321-680 (dE and rms dD test,
                       final full rebuild of F and E from the converged density)

The integrals come from oracle_ints.c (McMurchie-Davidson restatement of libcint's
conventions).  Parity pin: tests/test_oracle_golden.py checks this module against the
reference's known-answer energies (validation/check_rhf.f90:79-143, check_df.f90:55-61,
validation_tests_cpu.json) -- see DESIGN.md "Oracle".
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int_p = ctypes.POINTER(ctypes.c_int)
c_dbl_p = ctypes.POINTER(ctypes.c_double)


def build_oracle_lib(force: bool = False) -> str:
    so = os.path.join(HERE, "liboracle_ints.so")
    src = os.path.join(HERE, "oracle_ints.c")
    if force or not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle_lib())
    return _LIB


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _dp(a):
    return a.ctypes.data_as(c_dbl_p)


# --------------------------------------------------------------------------- basis
def gto_norm(l: int, a: float) -> float:
    """libcint's CINTgto_norm: 1/sqrt(int_0^inf r^(2l+2) exp(-2 a r^2) dr)."""
    n = 2 * l + 2
    gint = math.gamma((n + 1) / 2.0) / (2.0 * (2.0 * a) ** ((n + 1) / 2.0))
    return 1.0 / math.sqrt(gint)


def normalise_basis(shell_l, shell_nprim, exps, coefs):
    """Raw BSE coefficients -> radial coefficients libcint consumes (primitive norm, then the
    per-contraction rescale so that <chi|chi> = 1)."""
    out = np.array(coefs, dtype=np.float64, copy=True)
    off = 0
    for l, n in zip(shell_l, shell_nprim):
        l = int(l); n = int(n)
        a = np.asarray(exps[off:off + n], dtype=np.float64)
        c = np.array([coefs[off + i] * gto_norm(l, a[i]) for i in range(n)])
        norm2 = 0.0
        for i in range(n):
            for j in range(n):
                norm2 += (c[i] * c[j] / (gto_norm(l, a[i]) * gto_norm(l, a[j]))
                          * (2.0 * math.sqrt(a[i] * a[j]) / (a[i] + a[j])) ** (l + 1.5))
        if norm2 > 0.0:
            c = c / math.sqrt(norm2)
        out[off:off + n] = c
        off += n
    return out


@dataclass
class OracleMol:
    z: np.ndarray            # effective charges (ghost -> 0), float64
    xyz: np.ndarray          # (natom,3) Bohr
    sh_l: np.ndarray
    sh_nprim: np.ndarray
    sh_poff: np.ndarray
    sh_aoff: np.ndarray
    sh_xyz: np.ndarray       # (nshell,3)
    exps: np.ndarray
    coefs: np.ndarray        # normalised
    nao: int
    cart: bool = False       # Cartesian components above p (a basis with gto_cartesian d shells, e.g. 6-31G*)

    @property
    def nshell(self):
        return len(self.sh_l)

    def _basis_args(self, with_aoff=True, with_nao=True):
        lib().orc_set_cartesian(ctypes.c_int(1 if self.cart else 0))      # every integral call goes through here
        args = [ctypes.c_int(self.nshell), _ip(self.sh_l), _ip(self.sh_nprim), _ip(self.sh_poff)]
        if with_aoff:
            args.append(_ip(self.sh_aoff))
        args += [_dp(self.sh_xyz), _dp(self.exps), _dp(self.coefs)]
        if with_nao:
            args.append(ctypes.c_int(self.nao))
        return args


def nfun(l, cart=False):
    l = np.asarray(l)
    return np.where(cart & (l >= 2), (l + 1) * (l + 2) // 2, 2 * l + 1)


def make_mol(atomic_numbers, xyz_bohr, nshell_per_atom, shell_l, shell_nprim, exps, coefs_raw,
             ghost=None, cart=False) -> OracleMol:
    z = np.array(atomic_numbers, dtype=np.float64)
    if ghost is not None:
        z = np.where(np.asarray(ghost, dtype=bool), 0.0, z)
    xyz = np.ascontiguousarray(np.asarray(xyz_bohr, dtype=np.float64).reshape(-1, 3))
    sh_l = np.ascontiguousarray(shell_l, dtype=np.int32)
    sh_np = np.ascontiguousarray(shell_nprim, dtype=np.int32)
    poff = np.zeros(len(sh_l), dtype=np.int32)
    poff[1:] = np.cumsum(sh_np)[:-1]
    aoff = np.zeros(len(sh_l), dtype=np.int32)
    aoff[1:] = np.cumsum(nfun(sh_l, cart))[:-1]
    atoms = np.repeat(np.arange(len(nshell_per_atom)), np.asarray(nshell_per_atom, dtype=np.int64))
    sh_xyz = np.ascontiguousarray(xyz[atoms])
    ex = np.ascontiguousarray(exps, dtype=np.float64)
    co = np.ascontiguousarray(normalise_basis(sh_l, sh_np, ex, np.asarray(coefs_raw, dtype=np.float64)))
    return OracleMol(z, xyz, sh_l, sh_np, poff, aoff, sh_xyz, ex, co, int(np.sum(nfun(sh_l, cart))), bool(cart))


def nuclear_repulsion(mol: OracleMol) -> float:
    e = 0.0
    for i in range(len(mol.z)):
        for j in range(i):
            if mol.z[i] != 0.0 and mol.z[j] != 0.0:
                e += mol.z[i] * mol.z[j] / np.linalg.norm(mol.xyz[i] - mol.xyz[j])
    return float(e)


# --------------------------------------------------------------------------- integrals
def int1e(mol: OracleMol):
    n = mol.nao
    S = np.zeros((n, n)); T = np.zeros((n, n)); V = np.zeros((n, n))
    lib().orc_int1e(*mol._basis_args(), ctypes.c_int(len(mol.z)), _dp(mol.z), _dp(mol.xyz),
                    _dp(S), _dp(T), _dp(V))
    return S, T, V


def point_charge_potential(mol: OracleMol, points, charges) -> np.ndarray:
    """u = -sum_g q_g <mu| 1/|r - R_g| |nu>: the point-charge part of the FMO embedding operator
    (embedding_operator, backends/libcint/mqc_libcint_fmo.f90:1143-1151).  The nuclear-attraction routine with the
    charges in the place of the nuclei is exactly this operator."""
    n = mol.nao
    S = np.zeros((n, n)); T = np.zeros((n, n)); U = np.zeros((n, n))
    q = np.ascontiguousarray(charges, dtype=np.float64)
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    lib().orc_int1e(*mol._basis_args(), ctypes.c_int(len(q)), _dp(q), _dp(pts), _dp(S), _dp(T), _dp(U))
    return U


def mulliken_charges(mol: OracleMol, D, S) -> np.ndarray:
    """q_A = Z_A - sum_{mu on A} (D S)_mu,mu (mulliken_charges via fragment_charges, mqc_libcint_fmo.f90:2001-2021)."""
    pop = np.einsum("ij,ji->i", D, S)
    q = np.array(mol.z, dtype=np.float64)
    nf = nfun(mol.sh_l, mol.cart)
    for sh in range(mol.nshell):
        atom = int(np.argmin(np.sum((mol.xyz - mol.sh_xyz[sh]) ** 2, axis=1)))
        q[atom] -= float(np.sum(pop[mol.sh_aoff[sh]: mol.sh_aoff[sh] + nf[sh]]))
    return q


def eri4(mol: OracleMol) -> np.ndarray:
    n = mol.nao
    eri = np.zeros((n, n, n, n))
    lib().orc_eri4(*mol._basis_args(), _dp(eri))
    return eri


def schwarz(mol: OracleMol) -> np.ndarray:
    q = np.zeros((mol.nshell, mol.nshell))
    lib().orc_schwarz(*mol._basis_args(with_aoff=False, with_nao=False), _dp(q))
    return q


def eri3c(mol: OracleMol, aux: OracleMol) -> np.ndarray:
    out = np.zeros((mol.nao, mol.nao, aux.nao))
    lib().orc_eri3c(*mol._basis_args(), *aux._basis_args(), _dp(out))
    return out


def eri2c(aux: OracleMol) -> np.ndarray:
    out = np.zeros((aux.nao, aux.nao))
    lib().orc_eri2c(*aux._basis_args(), _dp(out))
    return out


def eval_ao(mol: OracleMol, pts: np.ndarray, deriv: bool = False):
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    npts = pts.shape[0]
    ao = np.zeros((npts, mol.nao))
    grad = np.zeros((3, npts, mol.nao)) if deriv else None
    lib().orc_eval_ao(*mol._basis_args(), ctypes.c_long(npts), _dp(pts), _dp(ao),
                      _dp(grad) if deriv else None)
    return (ao, grad) if deriv else ao


# --------------------------------------------------------------------------- SCF pieces
OVERLAP_EIG_TOL = 1.0e-7     # src/scf/mqc_scf_common.f90:33
GWH_K = 1.75                 # src/scf/mqc_scf_common.f90:39
PIVOT_FLOOR = 1.0e-14        # src/methods/mqc_diis.f90:30
METRIC_EIG_TOL = 1.0e-10     # backends/libcint/mqc_libcint_integrals.F90:1002


def build_orthogonalizer(S):
    w, U = np.linalg.eigh(S)
    keep = w > OVERLAP_EIG_TOL
    return U[:, keep] / np.sqrt(w[keep])[None, :]


def guess_fock_gwh(S, H):
    d = np.diag(H)
    F = 0.5 * GWH_K * S * (d[:, None] + d[None, :])
    F[np.diag_indices_from(F)] = d
    return F


def diagonalize(F, X):
    Fp = X.T @ F @ X
    eps, Cp = np.linalg.eigh(Fp)
    return X @ Cp, eps


def density_closed_shell(C, nocc):
    Co = C[:, :nocc]
    return 2.0 * Co @ Co.T


def commutator(F, D, S, X):
    return X.T @ (F @ D @ S - S @ D @ F) @ X


class Diis:
    """Bit-for-bit the reference's ring-buffer DIIS."""

    def __init__(self, max_vectors, n_fock, n_error):
        self.max_vectors = max_vectors
        self.fock = np.zeros((max(max_vectors, 1), n_fock))
        self.err = np.zeros((max(max_vectors, 1), n_error))
        self.overlap = np.zeros((max(max_vectors, 1), max(max_vectors, 1)))
        self.n_stored = 0
        self.newest = 0      # 1-based slot, 0 = empty

    def slot_of_age(self, age):   # age = 1 .. n_stored (oldest first); returns 1-based slot
        return (self.newest - self.n_stored + age - 1) % self.max_vectors + 1

    def push(self, fock_flat, err_flat):
        if self.max_vectors <= 0:
            return
        self.newest = self.newest % self.max_vectors + 1
        if self.n_stored < self.max_vectors:
            self.n_stored += 1
        s = self.newest - 1
        self.fock[s] = fock_flat
        self.err[s] = err_flat
        for age in range(1, self.n_stored + 1):
            o = self.slot_of_age(age) - 1
            v = float(np.sum(self.err[s] * self.err[o]))
            self.overlap[s, o] = v
            self.overlap[o, s] = v

    def coefficients(self):
        n = self.n_stored
        if self.max_vectors <= 0 or n < 2:
            return None
        B = -np.ones((n + 1, n + 1))
        B[n, n] = 0.0
        for j in range(n):
            for i in range(n):
                B[i, j] = self.overlap[self.slot_of_age(i + 1) - 1, self.slot_of_age(j + 1) - 1]
        scale = np.max(np.abs(B[:n, :n]))
        if scale > 0.0:
            B[:n, :n] /= scale
        return solve_diis(B)

    def extrapolate(self, fock_flat):
        c = self.coefficients()
        if c is None:
            return fock_flat, False
        out = np.zeros_like(fock_flat)
        for i in range(self.n_stored):
            out += c[i] * self.fock[self.slot_of_age(i + 1) - 1]
        return out, True


def solve_diis(B):
    """Gaussian elimination with partial pivoting, rhs = (0,...,0,-1) (mqc_diis.f90:232-273)."""
    n = B.shape[0]
    A = np.zeros((n, n + 1))
    A[:, :n] = B
    A[n - 1, n] = -1.0
    for i in range(n):
        piv = i
        for j in range(i + 1, n):
            if abs(A[j, i]) > abs(A[piv, i]):
                piv = j
        if piv != i:
            A[[i, piv], :] = A[[piv, i], :]
        p = A[i, i]
        if abs(p) < PIVOT_FLOOR:
            return None
        for j in range(i + 1, n):
            f = A[j, i] / p
            A[j, i:] -= f * A[i, i:]
    c = np.zeros(n)
    for i in range(n - 1, -1, -1):
        c[i] = (A[i, n] - np.sum(A[i, i + 1:n] * c[i + 1:n])) / A[i, i]
    return c


# --------------------------------------------------------------------------- Fock builders
def build_jk_incore(eri, D):
    J = np.einsum("ijkl,kl->ij", eri, D, optimize=True)
    K = np.einsum("ikjl,kl->ij", eri, D, optimize=True)
    return J, K


def df_tensor(mol: OracleMol, aux: OracleMol):
    """B = (mu nu|Q) J^{-1/2}, shape (n, n, A);  J^{-1/2} = U s^{-1/2} U^T over eig > 1e-10."""
    j3 = eri3c(mol, aux)
    j2 = eri2c(aux)
    w, U = np.linalg.eigh(j2)
    keep = w > METRIC_EIG_TOL
    jm12 = (U[:, keep] / np.sqrt(w[keep])[None, :]) @ U[:, keep].T
    return j3 @ jm12


def build_jk_df(B, D, Cocc=None):
    """DF-J from the density, DF-K through the occupied orbitals (K = 2 sum_P W_P W_P^T)."""
    cP = np.einsum("ijP,ij->P", B, D, optimize=True)
    J = np.einsum("ijP,P->ij", B, cP, optimize=True)
    if Cocc is not None:
        W = np.einsum("ijP,jo->ioP", B, Cocc, optimize=True)
        K = 2.0 * np.einsum("ioP,joP->ij", W, W, optimize=True)
    else:
        K = np.einsum("ikP,kl,jlP->ij", B, D, B, optimize=True)
    return J, K


@dataclass
class ScfResult:
    energy: float
    electronic: float
    nuclear: float
    converged: bool
    iterations: int
    eps: np.ndarray
    C: np.ndarray
    D: np.ndarray
    F: np.ndarray
    exc: float = 0.0
    energies: Optional[list] = None


def run_rhf(mol: OracleMol, nelec: int, max_iter=100, e_tol=1e-8, d_tol=1e-6, diis_vectors=8,
            guess="gwh", aux: Optional[OracleMol] = None, xc=None, k_scale=1.0,
            eri=None, B=None, h_extra=None) -> ScfResult:
    """Closed-shell SCF with the reference CPU path's semantics.

    h_extra: optional one-electron operator added to H before anything else reads it
        (mqc_libcint_rhf.f90:479-484), the FMO embedding field.

    xc: optional object with `.exx` (exact-exchange fraction) and
        `.potential(D) -> (exc, vxc)` (see oracle/xc_oracle.py).
    """
    if nelec % 2:
        raise ValueError("RHF needs an even electron count")
    nocc = nelec // 2
    S, T, V = int1e(mol)
    H = T + V
    if h_extra is not None:
        H = H + h_extra
    n = mol.nao
    if aux is not None and B is None:
        B = df_tensor(mol, aux)
    if aux is None and eri is None:
        eri = eri4(mol)
    X = build_orthogonalizer(S)
    m = X.shape[1]
    if nocc > m:
        raise ValueError("more occupied orbitals than the basis supports")
    exx = k_scale if xc is None else xc.exx

    def assemble(D, C):
        if B is not None:
            J, K = build_jk_df(B, D, C[:, :nocc] if C is not None else None)
        else:
            J, K = build_jk_incore(eri, D)
        F = H + J - 0.5 * exx * K
        e = 0.5 * float(np.sum(D * (H + F)))      # energy from the Fock BEFORE V_xc is added
        exc = 0.0
        if xc is not None:
            exc, vxc = xc.potential(D)
            F = F + vxc
            e += exc
        return F, e, exc

    if guess == "core":
        F = H.copy()
    elif guess == "gwh":
        F = guess_fock_gwh(S, H)
    else:
        raise ValueError("oracle supports core and gwh guesses")
    C, eps = diagonalize(F, X)
    D = density_closed_shell(C, nocc)

    diis = Diis(diis_vectors, n * n, m * m)
    e_old = 0.0
    converged = False
    iters = 0
    hist = []
    for it in range(1, max_iter + 1):
        D_old = D.copy()
        F, e_elec, _ = assemble(D, C)
        err = commutator(F, D, S, X)
        # Fortran reshape is column-major; inner products are layout independent as long as
        # both operands use the same flattening.
        diis.push(F.reshape(-1), err.reshape(-1))
        ff, ok = diis.extrapolate(F.reshape(-1))
        if ok:
            F = ff.reshape(n, n)
        C, eps = diagonalize(F, X)
        D = density_closed_shell(C, nocc)
        de = abs(e_elec - e_old)
        drms = math.sqrt(float(np.sum((D - D_old) ** 2)) / (n * n))
        hist.append(e_elec)
        e_old = e_elec
        iters = it
        if it > 1 and de < e_tol and drms < d_tol:
            converged = True
            break
    F, e_final, exc = assemble(D, C)
    enuc = nuclear_repulsion(mol)
    return ScfResult(e_final + enuc, e_final, enuc, converged, iters, eps, C, D, F, exc, hist)


# --------------------------------------------------------------------------- unrestricted SCF
UHF_DIIS_START = 4           # backends/libcint/mqc_libcint_rhf.f90:51 (DEFAULT_UHF_DIIS_START)


def spin_contamination(Ca, Cb, S, n_alpha, n_beta):
    """<S^2> of the unrestricted determinant: S_z (S_z + 1) + n_beta - sum_ij |<a_i|b_j>|^2."""
    sz = 0.5 * (n_alpha - n_beta)
    ov = Ca[:, :n_alpha].T @ S @ Cb[:, :n_beta]
    return sz * (sz + 1.0) + n_beta - float(np.sum(ov ** 2))


@dataclass
class UhfResult:
    energy: float
    electronic: float
    nuclear: float
    converged: bool
    iterations: int
    eps_a: np.ndarray
    eps_b: np.ndarray
    Ca: np.ndarray
    Cb: np.ndarray
    Da: np.ndarray
    Db: np.ndarray
    s_squared: float
    n_alpha: int
    n_beta: int


def run_uhf(mol: OracleMol, nelec: int, multiplicity: int, max_iter=100, e_tol=1e-8, d_tol=1e-6, diis_vectors=8,
            guess="gwh", diis_start=UHF_DIIS_START, eri=None, xc=None, aux: Optional[OracleMol] = None) -> UhfResult:
    """Unrestricted Hartree-Fock with the reference CPU path's semantics (run_libcint_uhf,
    backends/libcint/mqc_libcint_rhf.f90:682-974): F_s = H + J[D_a + D_b] - K[D_s], one DIIS over both spins
    (Fock matrices and commutators laid end to end) from iteration `diis_start`, dE and the rms over BOTH density
    changes, final full rebuild; symmetric core / GWH guess -- the occupations separate the spins."""
    if multiplicity < 1 or (nelec + multiplicity - 1) % 2:
        raise ValueError("UHF: electron count and multiplicity cannot be paired")
    na = (nelec + multiplicity - 1) // 2
    nb = nelec - na
    if nb < 0 or na < 1:
        raise ValueError("UHF: bad occupation")
    S, T, V = int1e(mol)
    H = T + V
    n = mol.nao
    B = df_tensor(mol, aux) if aux is not None else None       # density-fitted J / K, as the cuEST path's run_uks_scf
    if eri is None and B is None:
        eri = eri4(mol)
    X = build_orthogonalizer(S)
    m = X.shape[1]
    if na > m:
        raise ValueError("UHF: more alpha electrons than the basis supports")

    exx = 1.0 if xc is None else xc.exx

    def assemble(Da, Db):
        # unrestricted Kohn-Sham (run_libcint_uhf with an XC context): exchange scaled by the functional's fraction,
        # the energy from the Fock matrices BEFORE the spin potentials are added, plus E_xc
        if B is not None:
            J, Ka = build_jk_df(B, Da + Db)[0], build_jk_df(B, Da)[1]
            Kb = build_jk_df(B, Db)[1]
        else:
            J = np.einsum("ijkl,kl->ij", eri, Da + Db, optimize=True)
            Ka = np.einsum("ikjl,kl->ij", eri, Da, optimize=True) if exx != 0.0 else 0.0
            Kb = np.einsum("ikjl,kl->ij", eri, Db, optimize=True) if exx != 0.0 else 0.0
        Fa = H + J
        Fb = H + J
        if exx != 0.0:
            Fa = Fa - exx * Ka
            Fb = Fb - exx * Kb
        e = 0.5 * float(np.sum(Da * (H + Fa)) + np.sum(Db * (H + Fb)))
        if xc is not None:
            exc, Va, Vb = xc.potential_uks(Da, Db)
            Fa = Fa + Va; Fb = Fb + Vb
            e += exc
        return Fa, Fb, e

    F0 = H.copy() if guess == "core" else guess_fock_gwh(S, H)
    Ca, ea = diagonalize(F0, X)
    Cb, eb = Ca.copy(), ea.copy()
    Da = Ca[:, :na] @ Ca[:, :na].T
    Db = Cb[:, :nb] @ Cb[:, :nb].T
    diis = Diis(diis_vectors, 2 * n * n, 2 * m * m)
    e_old = 0.0
    converged = False
    iters = 0
    for it in range(1, max_iter + 1):
        Da_old, Db_old = Da.copy(), Db.copy()
        Fa, Fb, e_elec = assemble(Da, Db)
        erra = commutator(Fa, Da, S, X)
        errb = commutator(Fb, Db, S, X)
        ff = np.concatenate([Fa.reshape(-1), Fb.reshape(-1)])
        diis.push(ff, np.concatenate([erra.reshape(-1), errb.reshape(-1)]))
        if it >= diis_start:
            ex, ok = diis.extrapolate(ff)
            if ok:
                Fa = ex[: n * n].reshape(n, n)
                Fb = ex[n * n:].reshape(n, n)
        Ca, ea = diagonalize(Fa, X)
        Cb, eb = diagonalize(Fb, X)
        Da = Ca[:, :na] @ Ca[:, :na].T
        Db = Cb[:, :nb] @ Cb[:, :nb].T
        de = abs(e_elec - e_old)
        drms = math.sqrt((float(np.sum((Da - Da_old) ** 2)) + float(np.sum((Db - Db_old) ** 2))) / (2 * n * n))
        e_old = e_elec
        iters = it
        if it > 1 and de < e_tol and drms < d_tol:
            converged = True
            break
    _, _, e_final = assemble(Da, Db)
    enuc = nuclear_repulsion(mol)
    return UhfResult(e_final + enuc, e_final, enuc, converged, iters, ea, eb, Ca, Cb, Da, Db,
                     spin_contamination(Ca, Cb, S, na, nb), na, nb)

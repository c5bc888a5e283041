"""CPU oracle for the exchange-correlation quadrature.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy, what the reference's CPU path does per SCF iteration
(backends/libcint/mqc_libcint_xc.F90:796-927 xc_add_potential, :1379-1455 accumulate_xc_matrix;
backends/libcint/mqc_libcint_ao.f90:345-448 eval_rho; src/methods/mqc_xc_spec.f90:135-242):

    rho = rowdot(chi D, chi), grad rho = 2 rowdot(chi D, grad chi), sigma = |grad rho|^2
    E_xc = sum_g w rho eps_xc,   N_e = sum_g w rho
    V   += (w v_rho chi)^T chi + [(w 2 v_sigma grad rho . grad chi)^T chi + transpose]

The functional arithmetic lives in libxc 7.1.2 (third party, absent from /root/reference;
CMakeLists.txt:387-392).  Its published closed forms are restated here for the unpolarised case
(lda_x, lda_c_vwn, lda_c_vwn_rpa, gga_x_b88, gga_c_lyp, gga_x_pbe, gga_c_pbe with lda_c_pw_mod)
with derivatives by forward-mode dual numbers, and pinned by the reference's KS goldens
(validation/validation_tests_cpu.json: SVWN, PBE, B3LYP, PBE0 on H2O/cc-pVDZ, PBE on CH4).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import grid_oracle, scf_oracle

DENS_THRESHOLD = 1.0e-20     # points below this density contribute nothing (libxc's own thresholds
                             # are 1e-12..1e-24 per functional; the difference is far below 1e-10 Eh)


class Dual:
    """value + d/d rho + d/d sigma, elementwise on arrays."""
    __slots__ = ("v", "r", "s")

    def __init__(self, v, r=None, s=None):
        self.v = v
        self.r = np.zeros_like(v) if r is None else r
        self.s = np.zeros_like(v) if s is None else s

    @staticmethod
    def lift(x, like):
        return x if isinstance(x, Dual) else Dual(np.full_like(like.v, float(x)))

    def __add__(self, o):
        o = Dual.lift(o, self); return Dual(self.v + o.v, self.r + o.r, self.s + o.s)
    __radd__ = __add__

    def __neg__(self):
        return Dual(-self.v, -self.r, -self.s)

    def __sub__(self, o):
        o = Dual.lift(o, self); return Dual(self.v - o.v, self.r - o.r, self.s - o.s)

    def __rsub__(self, o):
        return Dual.lift(o, self) - self

    def __mul__(self, o):
        o = Dual.lift(o, self)
        return Dual(self.v * o.v, self.r * o.v + self.v * o.r, self.s * o.v + self.v * o.s)
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual.lift(o, self)
        inv = 1.0 / o.v
        q = self.v * inv
        return Dual(q, (self.r - q * o.r) * inv, (self.s - q * o.s) * inv)

    def __rtruediv__(self, o):
        return Dual.lift(o, self) / self

    def _chain(self, f, df):
        return Dual(f, df * self.r, df * self.s)

    def __pow__(self, p):
        f = self.v ** p
        return self._chain(f, p * self.v ** (p - 1.0))


def dexp(x): f = np.exp(x.v); return x._chain(f, f)
def dlog(x): return x._chain(np.log(x.v), 1.0 / x.v)
def dsqrt(x): f = np.sqrt(x.v); return x._chain(f, 0.5 / f)
def datan(x): return x._chain(np.arctan(x.v), 1.0 / (1.0 + x.v * x.v))
def dasinh(x): return x._chain(np.arcsinh(x.v), 1.0 / np.sqrt(1.0 + x.v * x.v))


# ------------------------------------------------------------------ functionals: energy per volume
def lda_x(rho, sigma):
    return -0.75 * (3.0 / math.pi) ** (1.0 / 3.0) * rho ** (4.0 / 3.0)


def _vwn(rho, A, x0, b, c):
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    x = dsqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = math.sqrt(4.0 * c - b * b)
    at = datan(Q / (2.0 * x + b))
    ec = A * (dlog(x * x / X) + (2.0 * b / Q) * at
              - (b * x0 / X0) * (dlog((x - x0) * (x - x0) / X) + (2.0 * (b + 2.0 * x0) / Q) * at))
    return rho * ec


def lda_c_vwn(rho, sigma):        # VWN5
    return _vwn(rho, 0.0310907, -0.10498, 3.72744, 12.9352)


def lda_c_vwn_rpa(rho, sigma):    # the "VWN3"/RPA fit libxc's hyb_gga_xc_b3lyp uses
    return _vwn(rho, 0.0310907, -0.409286, 13.0720, 42.7198)


def gga_x_b88(rho, sigma):
    beta = 0.0042
    cx = 1.5 * (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0)
    rs_ = 0.5 * rho                       # one spin channel
    r43 = rs_ ** (4.0 / 3.0)
    x = dsqrt(0.25 * sigma) / r43
    e = -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * dasinh(x))
    return 2.0 * e


def gga_c_lyp(rho, sigma):
    a, b, c, d = 0.04918, 0.132, 0.2533, 0.349
    cf = 0.3 * (3.0 * math.pi ** 2) ** (2.0 / 3.0)
    rm13 = rho ** (-1.0 / 3.0)
    den = 1.0 + d * rm13
    delta = c * rm13 + d * rm13 / den
    return -a * rho / den - a * b * dexp(-c * rm13) / den * (cf * rho - rho ** (-5.0 / 3.0) * sigma * ((3.0 + 7.0 * delta) / 72.0))


PBE_BETA = 0.06672455060314922
PBE_GAMMA = (1.0 - math.log(2.0)) / math.pi ** 2
PBE_MU = 0.2195149727645171
PBE_KAPPA = 0.804


def gga_x_pbe(rho, sigma):
    kf = (3.0 * math.pi ** 2) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)
    s2 = sigma / (4.0 * kf * kf * rho * rho)
    fx = 1.0 + PBE_KAPPA - PBE_KAPPA / (1.0 + PBE_MU * s2 / PBE_KAPPA)
    return lda_x(rho, sigma) * fx


def _pw_mod(rs):
    A, a1, b1, b2, b3, b4 = 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294   # libxc lda_c_pw_mod
    srs = dsqrt(rs)
    q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs)
    return -2.0 * A * (1.0 + a1 * rs) * dlog(1.0 + 1.0 / q)


def gga_c_pbe(rho, sigma):
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    ec = _pw_mod(rs)
    kf = (3.0 * math.pi ** 2) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)
    ks2 = 4.0 * kf / math.pi
    t2 = sigma / (4.0 * ks2 * rho * rho)
    A = (PBE_BETA / PBE_GAMMA) / (dexp(-ec / PBE_GAMMA) - 1.0)
    at2 = A * t2
    H = PBE_GAMMA * dlog(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2))
    return rho * (ec + H)


# name -> ([(weight, functional)], exact-exchange fraction, needs gradient)   (mqc_xc_spec.f90:135-242)
FUNCTIONALS = {
    "svwn": ([(1.0, lda_x), (1.0, lda_c_vwn)], 0.0, False),
    "lda": ([(1.0, lda_x), (1.0, lda_c_vwn)], 0.0, False),
    "lsda": ([(1.0, lda_x), (1.0, lda_c_vwn)], 0.0, False),
    "pbe": ([(1.0, gga_x_pbe), (1.0, gga_c_pbe)], 0.0, True),
    "blyp": ([(1.0, gga_x_b88), (1.0, gga_c_lyp)], 0.0, True),
    "b3lyp": ([(0.08, lda_x), (0.72, gga_x_b88), (0.19, lda_c_vwn_rpa), (0.81, gga_c_lyp)], 0.20, True),
    "pbe0": ([(0.75, gga_x_pbe), (1.0, gga_c_pbe)], 0.25, True),
}


def eval_functional(name: str, rho: np.ndarray, sigma: np.ndarray):
    """-> (f = rho*eps per volume, v_rho, v_sigma), zero where rho < threshold."""
    comps, _, _ = FUNCTIONALS[name.lower()]
    ok = rho > DENS_THRESHOLD
    r = np.where(ok, rho, 1.0)
    s = np.where(ok, np.maximum(sigma, 1.0e-40), 1.0e-40)   # sqrt'(0) guard; 1e-40 is numerically zero here
    R = Dual(r, np.ones_like(r), np.zeros_like(r))
    S = Dual(s, np.zeros_like(r), np.ones_like(r))
    f = vr = vs = 0.0
    for wgt, fn in comps:
        d = fn(R, S)
        d = Dual.lift(d, R)
        f = f + wgt * d.v; vr = vr + wgt * d.r; vs = vs + wgt * d.s
    z = np.zeros_like(rho)
    return np.where(ok, f, z), np.where(ok, vr, z), np.where(ok, vs, z)


@dataclass
class XCOracle:
    """`xc` object for scf_oracle.run_rhf: .exx and .potential(D) -> (E_xc, V_xc)."""
    mol: scf_oracle.OracleMol
    name: str
    level: int = 3
    block: int = 4096          # AO_POINT_BLOCK, mqc_libcint_ao.f90:47

    def __post_init__(self):
        self.comps, self.exx, self.gga = FUNCTIONALS[self.name.lower()]
        numbers = [int(round(z)) for z in self.mol.z]
        self.pts, self.w, self.owner = grid_oracle.build_grid(numbers, self.mol.xyz, self.level)
        self.n_electrons = 0.0

    def potential(self, D: np.ndarray) -> Tuple[float, np.ndarray]:
        n = self.mol.nao
        V = np.zeros((n, n))
        exc = 0.0
        nel = 0.0
        for b0 in range(0, len(self.w), self.block):
            p = self.pts[b0:b0 + self.block]; w = self.w[b0:b0 + self.block]
            if self.gga:
                ao, g = scf_oracle.eval_ao(self.mol, p, deriv=True)
            else:
                ao, g = scf_oracle.eval_ao(self.mol, p), None
            X = ao @ D
            rho = np.einsum("pi,pi->p", X, ao)
            if self.gga:
                grho = 2.0 * np.einsum("pi,dpi->dp", X, g)
                sigma = np.einsum("dp,dp->p", grho, grho)
            else:
                sigma = np.zeros_like(rho)
            f, vr, vs = eval_functional(self.name, rho, sigma)
            exc += float(np.dot(w, f)); nel += float(np.dot(w, rho))
            V += (ao * (w * vr)[:, None]).T @ ao
            if self.gga:
                gc = np.einsum("p,dp,dpi->pi", 2.0 * w * vs, grho, g)
                A = gc.T @ ao
                V += A + A.T
        self.n_electrons = nel
        return exc, V

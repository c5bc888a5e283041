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
(validation/validation_tests_cpu.json: SVWN, PBE, B3LYP, PBE0 on H2O/cc-pVDZ, PBE on CH4), plus the meta-GGA
mgga_x_tpss + mgga_c_tpss pair (restricted only) pinned by the TPSS row of the same manifest.
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


# ------------------------------------------------------------------ spin-polarised forms (unrestricted Kohn-Sham)
class DualN:
    """value + derivatives with respect to k independent variables (here rho_a, rho_b, sigma_aa, sigma_ab,
    sigma_bb), elementwise on arrays."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v = v
        self.d = d

    @staticmethod
    def var(v, i, k):
        return DualN(v, [np.ones_like(v) if j == i else np.zeros_like(v) for j in range(k)])

    def _lift(self, x):
        return x if isinstance(x, DualN) else DualN(np.full_like(self.v, float(x)), [np.zeros_like(self.v) for _ in self.d])

    def __add__(self, o):
        o = self._lift(o); return DualN(self.v + o.v, [a + b for a, b in zip(self.d, o.d)])
    __radd__ = __add__

    def __neg__(self):
        return DualN(-self.v, [-a for a in self.d])

    def __sub__(self, o):
        o = self._lift(o); return DualN(self.v - o.v, [a - b for a, b in zip(self.d, o.d)])

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o); return DualN(self.v * o.v, [a * o.v + self.v * b for a, b in zip(self.d, o.d)])
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._lift(o)
        inv = 1.0 / o.v
        q = self.v * inv
        return DualN(q, [(a - q * b) * inv for a, b in zip(self.d, o.d)])

    def __rtruediv__(self, o):
        return self._lift(o) / self

    def chain(self, f, df):
        return DualN(f, [df * a for a in self.d])

    def __pow__(self, p):
        return self.chain(self.v ** p, p * self.v ** (p - 1.0))


def nexp(x): f = np.exp(x.v); return x.chain(f, f)
def nlog(x): return x.chain(np.log(x.v), 1.0 / x.v)
def nsqrt(x): f = np.sqrt(x.v); return x.chain(f, 0.5 / f)
def natan(x): return x.chain(np.arctan(x.v), 1.0 / (1.0 + x.v * x.v))
def nasinh(x): return x.chain(np.arcsinh(x.v), 1.0 / np.sqrt(1.0 + x.v * x.v))


def _zeta_f(z):
    """f(zeta) = [(1+z)^(4/3) + (1-z)^(4/3) - 2] / (2^(4/3) - 2)"""
    return ((1.0 + z) ** (4.0 / 3.0) + (1.0 - z) ** (4.0 / 3.0) - 2.0) / (2.0 ** (4.0 / 3.0) - 2.0)


def _vwn_aux(rs, A, x0, b, c):
    x = nsqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = math.sqrt(4.0 * c - b * b)
    at = natan(Q / (2.0 * x + b))
    return A * (nlog(x * x / X) + (2.0 * b / Q) * at
                - (b * x0 / X0) * (nlog((x - x0) * (x - x0) / X) + (2.0 * (b + 2.0 * x0) / Q) * at))


def lda_x_pol(ra, rb, saa, sab, sbb):
    return -0.75 * (3.0 / math.pi) ** (1.0 / 3.0) * 2.0 ** (1.0 / 3.0) * (ra ** (4.0 / 3.0) + rb ** (4.0 / 3.0))


def lda_c_vwn_pol(ra, rb, saa, sab, sbb):
    """VWN5 with the spin stiffness: e_P + alpha_c f/f''(0) (1 - z^4) + (e_F - e_P) f z^4."""
    rho = ra + rb
    z = (ra - rb) / rho
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    eP = _vwn_aux(rs, 0.0310907, -0.10498, 3.72744, 12.9352)
    eF = _vwn_aux(rs, 0.01554535, -0.32500, 7.06042, 18.0578)
    aC = _vwn_aux(rs, -1.0 / (6.0 * math.pi ** 2), -0.0047584, 1.13107, 13.0045)
    fpp = 4.0 / (9.0 * (2.0 ** (1.0 / 3.0) - 1.0))
    f = _zeta_f(z)
    z4 = z * z * z * z
    return rho * (eP + aC * f * (1.0 - z4) / fpp + (eF - eP) * f * z4)


def lda_c_vwn_rpa_pol(ra, rb, saa, sab, sbb):
    """The RPA fit: plain f(zeta) interpolation between the paramagnetic and ferromagnetic fits."""
    rho = ra + rb
    z = (ra - rb) / rho
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    eP = _vwn_aux(rs, 0.0310907, -0.409286, 13.0720, 42.7198)
    eF = _vwn_aux(rs, 0.01554535, -0.743294, 20.1231, 101.578)
    f = _zeta_f(z)
    return rho * (eP * (1.0 - f) + eF * f)


def _b88_spin(r, s):
    beta = 0.0042
    cx = 1.5 * (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0)
    r43 = r ** (4.0 / 3.0)
    x = nsqrt(s) / r43
    return -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * nasinh(x))


def gga_x_b88_pol(ra, rb, saa, sab, sbb):
    return _b88_spin(ra, saa) + _b88_spin(rb, sbb)


def gga_c_lyp_pol(ra, rb, saa, sab, sbb):
    """Lee-Yang-Parr in the gradient-only form of Miehlich et al. (CPL 157, 200) for two spin densities."""
    a, b, c, d = 0.04918, 0.132, 0.2533, 0.349
    cf = 0.3 * (3.0 * math.pi ** 2) ** (2.0 / 3.0)
    rho = ra + rb
    rm13 = rho ** (-1.0 / 3.0)
    den = 1.0 + d * rm13
    omega = nexp(-c * rm13) / den * rho ** (-11.0 / 3.0)
    delta = c * rm13 + d * rm13 / den
    sig = saa + 2.0 * sab + sbb
    rab = ra * rb
    t1 = 2.0 ** (11.0 / 3.0) * cf * (ra ** (8.0 / 3.0) + rb ** (8.0 / 3.0))
    t2 = (47.0 / 18.0 - 7.0 * delta / 18.0) * sig
    t3 = (2.5 - delta / 18.0) * (saa + sbb)
    t4 = (delta - 11.0) / 9.0 * (ra * saa + rb * sbb) / rho
    br = rab * (t1 + t2 - t3 - t4) - (2.0 / 3.0) * rho * rho * sig \
        + ((2.0 / 3.0) * rho * rho - ra * ra) * sbb + ((2.0 / 3.0) * rho * rho - rb * rb) * saa
    return -a * 4.0 / den * rab / rho - a * b * omega * br


def _pbe_x_unpol(rho, sigma):
    kf = (3.0 * math.pi ** 2) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)
    s2 = sigma / (4.0 * kf * kf * rho * rho)
    fx = 1.0 + PBE_KAPPA - PBE_KAPPA / (1.0 + PBE_MU * s2 / PBE_KAPPA)
    return -0.75 * (3.0 / math.pi) ** (1.0 / 3.0) * rho ** (4.0 / 3.0) * fx


def gga_x_pbe_pol(ra, rb, saa, sab, sbb):
    """Exchange spin scaling: E[ra, rb] = (E[2 ra] + E[2 rb]) / 2."""
    return 0.5 * (_pbe_x_unpol(2.0 * ra, 4.0 * saa) + _pbe_x_unpol(2.0 * rb, 4.0 * sbb))


def _pw_mod_g(rs, A, a1, b1, b2, b3, b4):
    srs = nsqrt(rs)
    q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs)
    return -2.0 * A * (1.0 + a1 * rs) * nlog(1.0 + 1.0 / q)


PW_MOD_FZ20 = 1.709920934161365617563962776245


def _pw_mod_pol(rs, z):
    g0 = _pw_mod_g(rs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294)
    g1 = _pw_mod_g(rs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
    g2 = _pw_mod_g(rs, 0.0168869, 0.11125, 10.357, 3.6231, 0.88026, 0.49671)       # = -alpha_c
    f = _zeta_f(z)
    z4 = z * z * z * z
    return g0 - g2 * f * (1.0 - z4) / PW_MOD_FZ20 + (g1 - g0) * f * z4


def gga_c_pbe_pol(ra, rb, saa, sab, sbb):
    rho = ra + rb
    z = (ra - rb) / rho
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    ec = _pw_mod_pol(rs, z)
    phi = 0.5 * ((1.0 + z) ** (2.0 / 3.0) + (1.0 - z) ** (2.0 / 3.0))
    phi3 = phi * phi * phi
    kf = (3.0 * math.pi ** 2) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)
    ks2 = 4.0 * kf / math.pi
    sig = saa + 2.0 * sab + sbb
    t2 = sig / (4.0 * phi * phi * ks2 * rho * rho)
    A = (PBE_BETA / PBE_GAMMA) / (nexp(-ec / (PBE_GAMMA * phi3)) - 1.0)
    at2 = A * t2
    H = PBE_GAMMA * phi3 * nlog(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2))
    return rho * (ec + H)


POLARISED = {lda_x: lda_x_pol, lda_c_vwn: lda_c_vwn_pol, lda_c_vwn_rpa: lda_c_vwn_rpa_pol, gga_x_b88: gga_x_b88_pol,
             gga_c_lyp: gga_c_lyp_pol, gga_x_pbe: gga_x_pbe_pol, gga_c_pbe: gga_c_pbe_pol}
SPIN_FLOOR = 1.0e-30         # a spin density below this is held there: zeta stays inside (-1, 1)


def eval_functional_pol(name: str, ra, rb, saa, sab, sbb):
    """-> f per volume and (v_rho_a, v_rho_b, v_sigma_aa, v_sigma_ab, v_sigma_bb); zero where the TOTAL density is
    below the threshold (the layout of libxc's polarised calls, mqc_libcint_xc.F90:938-951)."""
    comps, _, _ = FUNCTIONALS[name.lower()]
    ok = (ra + rb) > DENS_THRESHOLD
    a = np.where(ok, np.maximum(ra, SPIN_FLOOR), 0.5)
    b = np.where(ok, np.maximum(rb, SPIN_FLOOR), 0.5)
    xs = [a, b, np.where(ok, np.maximum(saa, 1.0e-40), 1.0e-40), np.where(ok, sab, 0.0), np.where(ok, np.maximum(sbb, 1.0e-40), 1.0e-40)]
    V = [DualN.var(x, i, 5) for i, x in enumerate(xs)]
    f = 0.0
    dv = [0.0] * 5
    for wgt, fn in comps:
        d = POLARISED[fn](*V)
        f = f + wgt * d.v
        dv = [t + wgt * u for t, u in zip(dv, d.d)]
    z = np.zeros_like(ra)
    return np.where(ok, f, z), [np.where(ok, t, z) for t in dv]


# ------------------------------------------------------------------ meta-GGA: TPSS (unpolarised), variables (rho, sigma, tau)
# libxc 7.1.2 mgga_x_tpss / mgga_c_tpss, i.e. Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003) eqs. 5-14, with
# libxc's parameters (b 0.40, c 1.59096, e 1.537, kappa 0.804, mu 0.21951; d 2.8, C(0,0) 0.53, PBE beta and lda_c_pw_mod
# inside the correlation).  tau = 1/2 sum_i n_i |grad phi_i|^2 (mqc_libcint_ao.f90:374-417).
TAU_THRESHOLD = 1.0e-20


def _where(c, a, b):
    return DualN(np.where(c, a.v, b.v), [np.where(c, x, y) for x, y in zip(a.d, b.d)])


def mgga_x_tpss(rho, sigma, tau):
    b, c, e, kappa, mu = 0.40, 1.59096, 1.537, 0.804, 0.21951
    mu_ge = 10.0 / 81.0
    p = sigma / (4.0 * (3.0 * math.pi ** 2) ** (2.0 / 3.0) * rho ** (8.0 / 3.0))
    z = sigma / (8.0 * rho * tau)
    tau_unif = 0.3 * (3.0 * math.pi ** 2) ** (2.0 / 3.0) * rho ** (5.0 / 3.0)
    alpha = (tau - sigma / (8.0 * rho)) / tau_unif
    qb = 0.45 * (alpha - 1.0) / nsqrt(1.0 + b * alpha * (alpha - 1.0)) + (2.0 / 3.0) * p
    z2 = z * z
    num = ((mu_ge + c * z2 / ((1.0 + z2) * (1.0 + z2))) * p + (146.0 / 2025.0) * qb * qb
           - (73.0 / 405.0) * qb * nsqrt(0.5 * (0.36 * z2 + p * p)) + (mu_ge * mu_ge / kappa) * p * p
           + 2.0 * math.sqrt(e) * mu_ge * 0.36 * z2 + e * mu * p * p * p)
    den = (1.0 + math.sqrt(e) * p) * (1.0 + math.sqrt(e) * p)
    x = num / den
    fx = 1.0 + kappa - kappa / (1.0 + x / kappa)
    return -0.75 * (3.0 / math.pi) ** (1.0 / 3.0) * rho ** (4.0 / 3.0) * fx


def _pbe_c_eps_fixed_zeta(rho, sigma, ferro: bool):
    """PBE correlation energy PER PARTICLE at zeta = 0 or zeta = 1 (no zeta derivative is needed at either end)."""
    rs = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0) * rho ** (-1.0 / 3.0)
    if ferro:
        ec = _pw_mod_g(rs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
        phi = 2.0 ** (-1.0 / 3.0)
    else:
        ec = _pw_mod_g(rs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294)
        phi = 1.0
    phi3 = phi ** 3
    kf = (3.0 * math.pi ** 2) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)
    ks2 = 4.0 * kf / math.pi
    t2 = sigma / (4.0 * phi * phi * ks2 * rho * rho)
    A = (PBE_BETA / PBE_GAMMA) / (nexp(-ec / (PBE_GAMMA * phi3)) - 1.0)
    at2 = A * t2
    H = PBE_GAMMA * phi3 * nlog(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2))
    return ec + H


def mgga_c_tpss(rho, sigma, tau):
    d, C0 = 2.8, 0.53                      # C(zeta = 0, xi = 0)
    z = sigma / (8.0 * rho * tau)
    e_pbe = _pbe_c_eps_fixed_zeta(rho, sigma, False)
    e_one = _pbe_c_eps_fixed_zeta(0.5 * rho, 0.25 * sigma, True)       # one spin channel on its own, fully polarised
    e_til = _where(e_one.v > e_pbe.v, e_one, e_pbe)
    z2 = z * z
    e_rev = e_pbe * (1.0 + C0 * z2) - (1.0 + C0) * z2 * e_til
    return rho * e_rev * (1.0 + d * e_rev * z2 * z)


MGGA_FUNCTIONALS = {
    "tpss": ([(1.0, mgga_x_tpss), (1.0, mgga_c_tpss)], 0.0),
}


RESTRICTED_FUNCTIONALS = set(FUNCTIONALS) | set(MGGA_FUNCTIONALS)      # what the restricted Kohn-Sham leg covers


def eval_functional_mgga(name: str, rho, sigma, tau):
    """-> f per volume, v_rho, v_sigma, v_tau; zero where rho is below the threshold."""
    comps, _ = MGGA_FUNCTIONALS[name.lower()]
    ok = rho > DENS_THRESHOLD
    r = np.where(ok, rho, 1.0)
    s = np.where(ok, np.maximum(sigma, 1.0e-40), 1.0e-40)
    t = np.where(ok, np.maximum(tau, TAU_THRESHOLD), 1.0)
    V = [DualN.var(x, i, 3) for i, x in enumerate((r, s, t))]
    f = 0.0
    dv = [0.0] * 3
    for wgt, fn in comps:
        q = fn(*V)
        f = f + wgt * q.v
        dv = [a + wgt * b for a, b in zip(dv, q.d)]
    zz = np.zeros_like(rho)
    return np.where(ok, f, zz), np.where(ok, dv[0], zz), np.where(ok, dv[1], zz), np.where(ok, dv[2], zz)



# ---- spin-polarised TPSS: variables (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb, tau_a, tau_b)
def mgga_x_tpss_pol(ra, rb, saa, sab, sbb, ta, tb):
    """Exchange spin scaling: E[ra, rb] = (E[2 ra] + E[2 rb]) / 2 with sigma -> 4 sigma_ss, tau -> 2 tau_s."""
    return 0.5 * (mgga_x_tpss(2.0 * ra, 4.0 * saa, 2.0 * ta) + mgga_x_tpss(2.0 * rb, 4.0 * sbb, 2.0 * tb))


def mgga_c_tpss_pol(ra, rb, saa, sab, sbb, ta, tb):
    """revPKZB / TPSS correlation for two spin densities (PRL 91, 146401 eqs. 11-14; libxc mgga_c_tpss):
    C(zeta, xi) = C(zeta, 0) / {1 + xi^2 [(1+zeta)^(-4/3) + (1-zeta)^(-4/3)] / 2}^4,  xi = |grad zeta| / 2 (3 pi^2 rho)^(1/3)."""
    d = 2.8
    rho = ra + rb
    zeta = (ra - rb) / rho
    sig = saa + 2.0 * sab + sbb
    tau = ta + tb
    z = sig / (8.0 * rho * tau)
    z2 = z * z
    e_pbe = gga_c_pbe_pol(ra, rb, saa, sab, sbb) / rho
    e_a = _pbe_c_eps_fixed_zeta(ra, saa, True)
    e_b = _pbe_c_eps_fixed_zeta(rb, sbb, True)
    et_a = _where(e_a.v > e_pbe.v, e_a, e_pbe)
    et_b = _where(e_b.v > e_pbe.v, e_b, e_pbe)
    omz, opz = 1.0 - zeta, 1.0 + zeta
    gz2 = (omz * omz * saa - 2.0 * omz * opz * sab + opz * opz * sbb) / (rho * rho)          # |grad zeta|^2
    xi2 = gz2 / (4.0 * (3.0 * math.pi ** 2) ** (2.0 / 3.0) * rho ** (2.0 / 3.0))
    zeta2 = zeta * zeta
    c0 = 0.53 + 0.87 * zeta2 + 0.50 * zeta2 * zeta2 + 2.26 * zeta2 * zeta2 * zeta2
    den = 1.0 + 0.5 * xi2 * (opz ** (-4.0 / 3.0) + omz ** (-4.0 / 3.0))
    den2 = den * den
    C = c0 / (den2 * den2)
    e_rev = e_pbe * (1.0 + C * z2) - (1.0 + C) * z2 * (ra * et_a + rb * et_b) / rho
    return rho * e_rev * (1.0 + d * e_rev * z2 * z)


MGGA_POLARISED = {mgga_x_tpss: mgga_x_tpss_pol, mgga_c_tpss: mgga_c_tpss_pol}


def eval_functional_mgga_pol(name: str, ra, rb, saa, sab, sbb, ta, tb):
    """-> f per volume and the seven derivatives (v_rho_a, v_rho_b, v_sigma_aa, v_sigma_ab, v_sigma_bb, v_tau_a, v_tau_b)."""
    comps, _ = MGGA_FUNCTIONALS[name.lower()]
    ok = (ra + rb) > DENS_THRESHOLD
    xs = [np.where(ok, np.maximum(ra, SPIN_FLOOR), 0.5), np.where(ok, np.maximum(rb, SPIN_FLOOR), 0.5),
          np.where(ok, np.maximum(saa, 1.0e-40), 1.0e-40), np.where(ok, sab, 0.0), np.where(ok, np.maximum(sbb, 1.0e-40), 1.0e-40),
          np.where(ok, np.maximum(ta, TAU_THRESHOLD), 1.0), np.where(ok, np.maximum(tb, TAU_THRESHOLD), 1.0)]
    V = [DualN.var(x, i, 7) for i, x in enumerate(xs)]
    f = 0.0
    dv = [0.0] * 7
    for wgt, fn in comps:
        q = MGGA_POLARISED[fn](*V)
        f = f + wgt * q.v
        dv = [a + wgt * b for a, b in zip(dv, q.d)]
    zz = np.zeros_like(ra)
    return np.where(ok, f, zz), [np.where(ok, t, zz) for t in dv]


@dataclass
class XCOracle:
    """`xc` object for scf_oracle.run_rhf: .exx and .potential(D) -> (E_xc, V_xc)."""
    mol: scf_oracle.OracleMol
    name: str
    level: int = 3
    block: int = 4096          # AO_POINT_BLOCK, mqc_libcint_ao.f90:47

    def __post_init__(self):
        self.mgga = self.name.lower() in MGGA_FUNCTIONALS
        if self.mgga:
            self.comps, self.exx = MGGA_FUNCTIONALS[self.name.lower()]
            self.gga = True
        else:
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
            if self.mgga:
                # tau = 1/2 sum_d rowdot(d_d chi D, d_d chi); V += 1/2 sum_d (w v_tau d_d chi)^T d_d chi
                # (eval_rho, mqc_libcint_ao.f90:403-417; accumulate_xc_matrix, mqc_libcint_xc.F90:1436-1448)
                tau = 0.5 * sum(np.einsum("pi,pi->p", g[k] @ D, g[k]) for k in range(3))
                f, vr, vs, vt = eval_functional_mgga(self.name, rho, sigma, tau)
                for k in range(3):
                    V += 0.5 * (g[k] * (w * vt)[:, None]).T @ g[k]
            else:
                f, vr, vs = eval_functional(self.name, rho, sigma)
            exc += float(np.dot(w, f)); nel += float(np.dot(w, rho))
            V += (ao * (w * vr)[:, None]).T @ ao
            if self.gga:
                gc = np.einsum("p,dp,dpi->pi", 2.0 * w * vs, grho, g)
                A = gc.T @ ao
                V += A + A.T
        self.n_electrons = nel
        return exc, V

    def potential_uks(self, Da: np.ndarray, Db: np.ndarray):
        """E_xc and the two spin potentials (xc_add_potential_uks, mqc_libcint_xc.F90:929-1119): one AO evaluation,
        spin densities from C_s C_s^T (not doubled), dE/dgrad rho_a = 2 v_aa grad rho_a + v_ab grad rho_b."""
        n = self.mol.nao
        Va = np.zeros((n, n)); Vb = np.zeros((n, n))
        exc = 0.0; nel = 0.0
        for b0 in range(0, len(self.w), self.block):
            p = self.pts[b0:b0 + self.block]; w = self.w[b0:b0 + self.block]
            if self.gga:
                ao, g = scf_oracle.eval_ao(self.mol, p, deriv=True)
            else:
                ao, g = scf_oracle.eval_ao(self.mol, p), None
            Xa = ao @ Da; Xb = ao @ Db
            ra = np.einsum("pi,pi->p", Xa, ao); rb = np.einsum("pi,pi->p", Xb, ao)
            if self.gga:
                ga = 2.0 * np.einsum("pi,dpi->dp", Xa, g); gb = 2.0 * np.einsum("pi,dpi->dp", Xb, g)
                saa = np.einsum("dp,dp->p", ga, ga); sab = np.einsum("dp,dp->p", ga, gb); sbb = np.einsum("dp,dp->p", gb, gb)
            else:
                saa = sab = sbb = np.zeros_like(ra)
            if self.mgga:
                ta = 0.5 * sum(np.einsum("pi,pi->p", g[k] @ Da, g[k]) for k in range(3))
                tb = 0.5 * sum(np.einsum("pi,pi->p", g[k] @ Db, g[k]) for k in range(3))
                f, (vra, vrb, vaa, vab, vbb, vta, vtb) = eval_functional_mgga_pol(self.name, ra, rb, saa, sab, sbb, ta, tb)
                for k in range(3):
                    Va += 0.5 * (g[k] * (w * vta)[:, None]).T @ g[k]
                    Vb += 0.5 * (g[k] * (w * vtb)[:, None]).T @ g[k]
            else:
                f, (vra, vrb, vaa, vab, vbb) = eval_functional_pol(self.name, ra, rb, saa, sab, sbb)
            exc += float(np.dot(w, f)); nel += float(np.dot(w, ra + rb))
            Va += (ao * (w * vra)[:, None]).T @ ao
            Vb += (ao * (w * vrb)[:, None]).T @ ao
            if self.gga:
                ca = 2.0 * vaa * ga + vab * gb
                cb = 2.0 * vbb * gb + vab * ga
                A = np.einsum("p,dp,dpi->pi", w, ca, g).T @ ao
                B = np.einsum("p,dp,dpi->pi", w, cb, g).T @ ao
                Va += A + A.T; Vb += B + B.T
        self.n_electrons = nel
        return exc, Va, Vb

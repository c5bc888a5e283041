"""CPU oracle for the molecular quadrature grid.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's grid builder (all citations relative to /root/reference):
  level tables          src/methods/mqc_dft_grid.f90:52-81
  radial mesh           src/methods/mqc_dft_radial.f90:73-112 (Treutler-Ahlrichs M4, alpha = 0.6)
  xi / Bragg tables     src/methods/mqc_dft_radial_data.f90 (published values: Treutler & Ahlrichs
                        JCP 102, 346 (1995); Bragg-Slater radii in Angstrom / 0.52917721092)
  NWChem pruning        src/methods/mqc_dft_prune.f90:44-133
  Becke partition with Treutler size adjustment
                        src/methods/mqc_dft_partition.f90:82-179,324-375
  assembly              src/methods/mqc_dft_grid.f90:145-257 (w = 4 pi r^2 dr w_leb w_becke)
Lebedev points come from scipy.integrate.lebedev_rule (Lebedev-Laikov tables; weights there sum
to 4 pi, the reference's sum to 1).  Pinned by SURVEY.md section 9: the check_rhf water at level 3
has 33 698 points and sum(w) = 17 026.5356, and by the KS goldens of the manifest.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
from scipy.integrate import lebedev_rule

PERIOD_LAST_Z = [2, 10, 18, 36, 54, 86, 118]
RAD_GRIDS = [[10, 15, 20, 30, 35, 40, 50], [30, 40, 50, 60, 65, 70, 75], [40, 60, 65, 75, 80, 85, 90],
             [50, 75, 80, 90, 95, 100, 105], [60, 90, 95, 105, 110, 115, 120], [70, 105, 110, 120, 125, 130, 135],
             [80, 120, 125, 135, 140, 145, 150], [90, 135, 140, 150, 155, 160, 165],
             [100, 150, 155, 165, 170, 175, 180], [200, 200, 200, 200, 200, 200, 200]]
ANG_POINTS = [[50, 86, 110, 110, 110, 110, 110], [110, 194, 194, 194, 194, 194, 194],
              [194, 302, 302, 302, 302, 302, 302], [302, 302, 434, 434, 434, 434, 434],
              [434, 590, 590, 590, 590, 590, 590], [590, 770, 770, 770, 770, 770, 770],
              [770, 974, 974, 974, 974, 974, 974], [974, 1202, 1202, 1202, 1202, 1202, 1202],
              [1202, 1202, 1202, 1202, 1202, 1202, 1202], [1454, 1454, 1454, 1454, 1454, 1454, 1454]]

# Z = 0 (ghost) .. 36
TREUTLER_XI = [1.0, 0.8, 0.9, 1.8, 1.4, 1.3, 1.1, 0.9, 0.9, 0.9, 0.9, 1.4, 1.3, 1.3, 1.2, 1.1, 1.0, 1.0, 1.0,
               1.5, 1.4, 1.3, 1.2, 1.2, 1.2, 1.2, 1.2, 1.2, 1.1, 1.1, 1.1, 1.1, 1.0, 0.9, 0.9, 0.9, 0.9]
BRAGG_ANGSTROM = [2.0, 0.35, 1.40, 1.45, 1.05, 0.85, 0.70, 0.65, 0.60, 0.50, 1.50, 1.80, 1.50, 1.25, 1.10, 1.00,
                  1.00, 1.00, 1.80, 2.20, 1.80, 1.60, 1.40, 1.35, 1.40, 1.40, 1.40, 1.35, 1.35, 1.35, 1.35, 1.30,
                  1.25, 1.15, 1.15, 1.15, 1.90]
BRAGG_BOHR = [a / 0.52917721092 for a in BRAGG_ANGSTROM]

PRUNE_ORDERS = [38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590, 770, 974, 1202, 1454, 1730,
                2030, 2354, 2702, 3074, 3470, 3890, 4334, 4802, 5294, 5810]
ALPHAS = [[0.25, 0.5, 1.0, 4.5], [0.1667, 0.5, 0.9, 3.5], [0.1, 0.4, 0.8, 2.5]]
LEBEDEV_DEGREE = {6: 3, 14: 5, 26: 7, 38: 9, 50: 11, 74: 13, 86: 15, 110: 17, 146: 19, 170: 21, 194: 23, 230: 25,
                  266: 27, 302: 29, 350: 31, 434: 35, 590: 41, 770: 47, 974: 53, 1202: 59, 1454: 65}
M4_ALPHA = 0.6
MAX_ADJUST = 0.5


def element_period(z):
    for i, last in enumerate(PERIOD_LAST_Z):
        if z <= last:
            return i
    return 6


@lru_cache(maxsize=None)
def lebedev(npts):
    x, w = lebedev_rule(LEBEDEV_DEGREE[npts])
    assert x.shape[1] == npts
    return x.T.copy(), w / (4.0 * math.pi)


def radial(n, z):
    xi = TREUTLER_XI[z]
    step = math.pi / (n + 1)
    scale = xi / math.log(2.0)
    r, dr = np.zeros(n), np.zeros(n)
    for i in range(1, n + 1):
        x = math.cos(i * step); s = math.sin(i * step)
        lt = math.log((1.0 - x) / 2.0); mt = (1.0 + x) ** M4_ALPHA
        j = n - i
        r[j] = -scale * mt * lt
        dr[j] = step * s * scale * mt * (-M4_ALPHA / (1.0 + x) * lt + 1.0 / (1.0 - x))
    return r, dr


def prune_orders(z, r, n_ang):
    if n_ang < 50:
        return [n_ang] * len(r)
    if n_ang == 50:
        zone = [PRUNE_ORDERS[1], PRUNE_ORDERS[2], PRUNE_ORDERS[2], PRUNE_ORDERS[2], PRUNE_ORDERS[1]]
    else:
        t = PRUNE_ORDERS.index(n_ang)
        zone = [PRUNE_ORDERS[1], PRUNE_ORDERS[3], PRUNE_ORDERS[t - 1], PRUNE_ORDERS[t], PRUNE_ORDERS[t - 1]]
    cls = 0 if z <= 2 else (1 if z <= 10 else 2)
    out = []
    for ri in r:
        scaled = ri / (BRAGG_BOHR[z] + 1e-200)
        out.append(zone[sum(1 for a in ALPHAS[cls] if scaled > a)])
    return out


def atom_template(z, level=3):
    """Points relative to the nucleus, product weights 4 pi r^2 dr w_leb (no partition yet)."""
    p = element_period(z)
    nr, na = RAD_GRIDS[level][p], ANG_POINTS[level][p]
    r, dr = radial(nr, z)
    orders = prune_orders(z, r, na)
    pts, wts = [], []
    for ri, dri, o in zip(r, dr, orders):
        sph, w = lebedev(o)
        pts.append(ri * sph)
        wts.append(4.0 * math.pi * ri * ri * dri * w)
    return np.vstack(pts), np.concatenate(wts)


def becke_cutoff(nu):
    f = nu
    for _ in range(3):
        f = 0.5 * f * (3.0 - f * f)
    return 0.5 * (1.0 - f)


def becke_weights(points, owner, atom_xyz, numbers):
    n = len(numbers)
    if n == 1:
        return np.ones(len(points))
    radius = np.sqrt(np.array([BRAGG_BOHR[z] for z in numbers])) + 1e-200      # Treutler adjustment
    shift = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i != j:
                chi = radius[i] / radius[j]
                shift[i, j] = max(-MAX_ADJUST, min(MAX_ADJUST, 0.25 * (1.0 / chi - chi)))
    d = np.linalg.norm(points[:, None, :] - atom_xyz[None, :, :], axis=2)        # (P, n)
    cell = np.ones((len(points), n))
    for i in range(n):
        for j in range(i + 1, n):
            mu = (d[:, i] - d[:, j]) / np.linalg.norm(atom_xyz[i] - atom_xyz[j])
            nu = mu + shift[i, j] * (1.0 - mu * mu)
            s = becke_cutoff(nu)
            cell[:, i] *= s
            cell[:, j] *= (1.0 - s)
    tot = cell.sum(axis=1)
    own = cell[np.arange(len(points)), owner]
    return np.where(tot > 0.0, own / np.where(tot > 0.0, tot, 1.0), 0.0)


def build_grid(numbers, atom_xyz_bohr, level=3):
    """-> points (P,3), weights (P,), owner (P,)"""
    atom_xyz = np.asarray(atom_xyz_bohr, dtype=float).reshape(-1, 3)
    pts, wts, own = [], [], []
    for ia, z in enumerate(numbers):
        p, w = atom_template(int(z), level)
        pts.append(p + atom_xyz[ia]); wts.append(w); own.append(np.full(len(w), ia))
    pts = np.vstack(pts); wts = np.concatenate(wts); own = np.concatenate(own)
    return pts, wts * becke_weights(pts, own, atom_xyz, [int(z) for z in numbers]), own

import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from metalquicha_amd import methods
from metalquicha_amd.basis import ANGSTROM_TO_BOHR
from oracle import scf_oracle as so, xc_oracle
from tests.helpers import oracle_mol
rcc, rch = 1.397, 1.084
sym, xyz = [], []
for k in range(6):
    a = np.pi/3*k
    sym.append('C'); xyz.append([rcc*np.cos(a), rcc*np.sin(a), 0.0])
for k in range(6):
    a = np.pi/3*k
    sym.append('H'); xyz.append([(rcc+rch)*np.cos(a), (rcc+rch)*np.sin(a), 0.0])
frag = methods.PhysicalFragment.from_angstrom(sym, xyz)
AUX = "mqc-even-tempered-jkfit"
for fn, df in (("", False), ("b3lyp", True)):
    st = methods.ScfSettings(basis_set="cc-pvdz", functional=fn, density_fitting=df, aux_basis_set=AUX, energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    t=time.time(); r = methods.run_hip_scf(st, frag); t1=time.time()-t
    t=time.time(); r = methods.run_hip_scf(st, frag); t2=time.time()-t
    print("benzene", fn or "rhf", "df" if df else "exact", r.has_error, r.error_message, "E=%.10f iters=%d  first %.2fs second %.2fs" % (r.energy.scf, r.scf_iterations, t1, t2), flush=True)
    st_ = methods.get_stats()
    print("   stats: eri %.3f fock %.3f xc %.3f total %.3f" % (st_.t_eri, st_.t_fock, st_.xc_kernel_seconds, st_.t_total), flush=True)
if len(sys.argv) > 1:
    mol = oracle_mol("cc-pvdz", frag); aux = oracle_mol(AUX, frag)
    t=time.time()
    o = so.run_rhf(mol, 42, 100, 1e-10, 1e-8, aux=aux, xc=xc_oracle.XCOracle(mol, "b3lyp", 3))
    print("oracle DF-B3LYP E=%.10f iters=%d %.1fs  diff %.2e" % (o.energy, o.iterations, time.time()-t, r.energy.scf-o.energy))

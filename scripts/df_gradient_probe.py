import numpy as np, sys
sys.path.insert(0, "/root/repo")
from metalquicha_amd import methods
from metalquicha_amd.methods import ScfSettings
from tests import workload_cases as wc
from tests.helpers import fragment_bohr, recorded_oracle
frag = fragment_bohr([8, 1, 1], wc.DF_GRAD_XYZ)
st = ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set=wc.AUX, energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
r = methods.HFMethod(st).calc_gradient(frag)
o = recorded_oracle("df_rhf_gradient_water", frag, wc.DF_GRAD_KEY, wc.df_gradient_oracle)
fd = np.array(o["gradient"])
print("engine\n", r.gradient, "\noracle FD\n", fd, "\nmax dev", np.abs(r.gradient-fd).max())
st2 = ScfSettings(basis_set="cc-pvdz", energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
r2 = methods.HFMethod(st2).calc_gradient(frag)
print("exact - DF gradient max", np.abs(r2.gradient - r.gradient).max())

import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import *
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
from tests.helpers import GLIDER, oracle_step_hessian
gpu = torch.device("cuda", 0)
for act in [[1, 0, 1, 0], [0, 0, 1, 1], [0, 1, 0, 0], [0, 0, 0, 0], [1, 1, 1, 1]]:
    base = MlpData.synthetic((48, 24, 40), seed=5)
    md = MlpData(base.weights, base.biases, act, base.input_mean, base.input_std, base.output_mean, base.output_std)
    ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=md, aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
    ac.normalise = True
    X, U = synthetic_units(150, seed=31, flaps=True)
    Xn, A, Bm, c = ac.step_sens(dev(X, gpu), dev(U, gpu), 0.01)
    orc = make_oracle(ac)
    Xr, Ar, Br, cr = orc.step_sens(X, U, 0.01)
    lam = f32_exact(np.random.default_rng(3).standard_normal((13, 150)))
    Hd = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy()
    Hr = oracle_step_hessian(orc, X, U, 0.01, lam)
    print(act, "state %.2e (1e-5)  A %.2e B %.2e c %.2e (1e-4)  H %.2e (2e-3)" % (block_rel_err(Xn.cpu().numpy(), Xr), rel_fro(A.cpu().numpy(), Ar), rel_fro(Bm.cpu().numpy(), Br), rel_fro(c.cpu().numpy(), cr), rel_fro(Hd, Hr)))

"""Repeats the all-linear (fully folded, single-layer) and the mixed activation patterns many times in one process and
prints the worst parity figures: a guard against run-to-run variation in the folded-net path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import block_rel_err, dev, f32_exact, make_oracle, rel_fro, synthetic_units
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
from tests.helpers import GLIDER, oracle_step_hessian
gpu = torch.device("cuda", 0)
X, U = synthetic_units(150, seed=31, flaps=True)
lam = f32_exact(np.random.default_rng(3).standard_normal((13, 150)))
for act in [[0, 0, 0, 0], [1, 0, 1, 0]]:
    base = MlpData.synthetic((48, 24, 40), seed=5)
    md = MlpData(base.weights, base.biases, act, base.input_mean, base.input_std, base.output_mean, base.output_std)
    worst = np.zeros(4)
    ref = None
    for it in range(25):
        ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=md, aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
        ac.normalise = True
        if ref is None:
            orc = make_oracle(ac)
            Xr, Ar, Br, cr = orc.step_sens(X, U, 0.01)
            Hr = oracle_step_hessian(orc, X, U, 0.01, lam)
            ref = True
        Xn, A, Bm, c = ac.step_sens(dev(X, gpu), dev(U, gpu), 0.01)
        Xf = ac.state_update(dev(X, gpu), dev(U, gpu), 0.01)
        Hd = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy()
        worst = np.maximum(worst, [block_rel_err(Xn.cpu().numpy(), Xr), block_rel_err(Xf.cpu().numpy(), Xr), rel_fro(A.cpu().numpy(), Ar), rel_fro(Hd, Hr)])
        del ac
    print(act, "worst over 25 fresh handles: state(sens) %.2e state(fwd) %.2e A %.2e H %.2e" % tuple(worst))

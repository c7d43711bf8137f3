import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from tests.helpers import block_rel_err, f32_exact, make_aircraft, make_oracle, oracle_step_hessian, rel_fro, synthetic_units
from tests.test_gpu_fuzz import MODELS, dev
gpu = torch.device("cuda", 0)
SEEDS = [int(a) for a in sys.argv[1:]] or [32, 44, 67, 69, 80, 146, 157, 164, 188, 206]
for seed in SEEDS:
    rng = np.random.default_rng(1000 + seed)
    model, hidden, use_mfma = MODELS[seed % len(MODELS)]
    substeps = int(rng.choice([1, 1, 2, 3])); normalise = bool(rng.integers(2)); stall = bool(rng.integers(2))
    n = int(rng.choice([1, 15, 17, 63, 65, 250, 1025])); dt = float(rng.choice([0.005, 0.01, 0.02]))
    ac = make_aircraft(model, hidden=hidden, substeps=substeps, normalise=normalise, stall_scaling=stall, use_mfma=use_mfma)
    orc = make_oracle(ac)
    X, U = synthetic_units(n, seed=seed, flaps=bool(rng.integers(2)))
    Xd, Ud = dev(X, gpu), dev(U, gpu)
    xd = ac.state_derivative(Xd, Ud).cpu().numpy(); want = orc.state_derivative(X, U)
    e_der = np.abs(xd - want).max() / max(np.abs(want).max(), 1.0)
    per_unit = bool(rng.integers(2))
    dts = f32_exact(rng.uniform(0.5 * dt, 1.5 * dt, n)) if per_unit else dt
    xn = ac.state_update(Xd, Ud, dev(dts, gpu) if per_unit else dt).cpu().numpy()
    e_step = block_rel_err(xn, orc.state_update(X, U, dts))
    Xn, A, Bm, c = ac.step_sens(Xd, Ud, dev(dts, gpu) if per_unit else dt)
    Xr, Ar, Br, cr = orc.step_sens(X, U, dts)
    eA, eB, ec = rel_fro(A.cpu().numpy(), Ar), rel_fro(Bm.cpu().numpy(), Br), rel_fro(c.cpu().numpy(), cr)
    eH = None
    if substeps == 1 and n <= 250:
        lam = f32_exact(rng.normal(size=(13, n)))
        Hm = ac.step_hess(Xd, Ud, dev(dts, gpu) if per_unit else dt, dev(lam, gpu)).cpu().numpy().astype(np.float64)
        Hr = oracle_step_hessian(orc, X, U, dts, lam)
        num = np.sqrt(((Hm - Hr) ** 2).sum(axis=(0, 1))); den = np.sqrt((Hr ** 2).sum(axis=(0, 1)))
        rel = num / np.maximum(den, 1e-30)
        eH = rel.max()
        if eH > 1e-3:  # which units, and how close are they to the |.| kinks of the stall scaling (alpha, beta -> 0)?
            a = orc.aero(X, U)
            order = np.argsort(-rel)[:3]
            print("   worst second-order units:", [(int(i), f"{rel[i]:.2e}", f"alpha {a[4][i]:+.2e}", f"beta {a[5][i]:+.2e}") for i in order],
                  "units above 1e-3:", int((rel > 1e-3).sum()), "of", n)
            # the checker's step: central differences of exact Jacobians with h = 1e-5; a kink inside the step spoils it — repeat with h = 1e-7
            Hr2 = oracle_step_hessian(orc, X, U, dts, lam, h=1e-7)
            num2 = np.sqrt(((Hm - Hr2) ** 2).sum(axis=(0, 1))); den2 = np.sqrt((Hr2 ** 2).sum(axis=(0, 1)))
            rel2 = num2 / np.maximum(den2, 1e-30)
            print("   the same units against a checker step of 1e-7:", [f"{rel2[i]:.2e}" for i in order])
    Uh = f32_exact(np.tile(U[None], (6, 1, 1)))
    traj = ac.rollout(Xd, dev(Uh, gpu), dt).cpu().numpy()
    e_roll = block_rel_err(traj, orc.rollout(X, Uh, dt))
    print(f"seed {seed} {model}{hidden} sub={substeps} n={n} dt={dt}: der {e_der:.2e} (5e-6) step {e_step:.2e} (5e-6) A {eA:.2e} (2e-5) B {eB:.2e} (1e-4) c {ec:.2e} (1e-4) H {eH if eH is None else f'{eH:.2e}'} (1e-3) roll {e_roll:.2e} (2e-5)", flush=True)

"""Augmented-Lagrangian envelope: violation of the alpha bound over the outer iterations (diagnostic)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ilqr import setup, dev
from aircraft_amd.control import ILQR, QuadraticCost
gpu = torch.device("cuda", 0)
ac, il0, cost, X0, U = setup(gpu, "poly", None, B=48, H=40)
cost = QuadraticCost.goal((24.0, 0.0), w_goal=1.0, height=-185.0, w_height=40.0, w_lateral_speed=0.1, r=0.02, reg=1.0)
lim = np.deg2rad(4)
zmax = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if zmax < 0: lim = np.deg2rad(20)
bounds = ((20.0 ** 2, 100.0 ** 2), (-np.deg2rad(10), np.deg2rad(10)), (-lim, lim), (-1e30, zmax))
w = float(sys.argv[1]) if len(sys.argv) > 1 else 2e5
inner = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for mode in ("penalty", "al"):
    il = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.25, 0.1, 0.03), envelope_weight=w, envelope_bounds=bounds, envelope=mode)
    x0 = dev(X0, gpu); Uc = dev(np.zeros_like(U), gpu)
    X = il.rollout(x0, Uc)
    for outer in range(12):
        imp = 0
        for _ in range(inner):
            J, improved = il.iterate(x0, X, Uc); imp += int(improved.sum())
        rows, _ = il.envelope(X)
        a = rows[:, 2].abs().amax(dim=0).cpu().numpy() / lim
        zex = (rows[:, 3].amax(dim=0) - zmax).cpu().numpy()
        msg = f"{mode} w={w:g} outer {outer}: alpha/limit median {np.median(a):.3f} max {a.max():.3f} z excess median {np.median(zex):.4f} max {zex.max():.4f} J median {float(J.median()):.4g} improved {imp}"
        if mode == "al":
            il.update_multipliers(X)
            lam = il._ws["lam"]
            msg += f"  lam alpha rows max {float(lam[:, [2, 6]].max()):.3g} mean {float(lam[:, [2,6]].mean()):.3g}"
        print(msg, flush=True)
    if mode == "al":
        ws = il._ws
        J, improved = il.iterate(x0, X, Uc)
        na = len(il.alphas); B = 48
        Jc = ws["Jc"].view(na, B).cpu().numpy(); J0 = ws["J0"].cpu().numpy(); dV = ws["dV"].cpu().numpy()
        print("dV[0] (gradient term) median", np.median(dV[0]), "dV[1]", np.median(dV[1]))
        for a, al in enumerate(il.alphas):
            print(f"  alpha {al}: median Jc - J0 = {np.median(Jc[a] - J0):.4g}; predicted {np.median(al * dV[0] + al * al * dV[1]):.4g}; finite {np.isfinite(Jc[a]).mean():.2f}")
        kff = ws["kff"].cpu().numpy(); print("  |kff| max per control", np.abs(kff).max(axis=(0, 2)))
        Un = Uc.cpu().numpy(); print("  |U| max per control", np.abs(Un).max(axis=(0, 2)))

#!/usr/bin/env python3
"""The reference's goal-acquisition problem on the batched sweep (needs an MI355X).

What /root/reference/main/control/control.py:157-213 sets up — glider from data/glider/problem_definition.json, the
polynomial coefficient model every reference driver uses (:166), the driver's centre-of-mass override (:172), trim state at
80 m/s and 200 m, N = 400 nodes of dt = 0.01 s, goal (150, 0), vel_param = +1, Controller.loss (:44-68) with the final
velocity constraint v_x(N) < -2 — solved here for B random restarts at once (perturbed initial controls) by the iLQR sweep
on the exact loss, where the reference hands ONE instance to IPOPT.  Prints the loss history (mean / best over the batch).

    python examples/goal_acquisition.py [--batch 64] [--iters 12] [--time variable]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--nodes", type=int, default=400)
    ap.add_argument("--time", choices=("fixed", "variable"), default="fixed")
    args = ap.parse_args()
    import torch
    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
    from aircraft_amd.control import GoalAcquisition
    from aircraft_amd.synthetic import GLIDER

    poly = os.path.join(ROOT, "tests", "golden", "poly_coef.npz")  # the reference's fitted_models_casadi.pkl, decoded
    ac = Aircraft(AircraftOpts(coeff_model_type="poly", coeff_model_path=poly, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                               physical_integration_substeps=1))
    ac.com = np.array([0.0131991, -1.78875e-08, 0.00313384])       # control.py:169-172
    dev = torch.device("cuda", 0)
    B, H = args.batch, args.nodes
    x0 = torch.tensor([0, 0, -200, 80, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device=dev)[:, None].expand(13, B).contiguous()
    rng = np.random.default_rng(0)
    U0 = np.zeros((H, 7, B), dtype=np.float32)
    U0[:, :3] = rng.normal(0, 0.3, (1, 3, B))                      # random restarts: a constant offset on the three surfaces
    il = GoalAcquisition(system=ac, goal=(150.0, 0.0), dt=0.01, num_nodes=H, vel_param=1.0, time=args.time,
                         alphas=(1.0, 0.5, 0.25, 0.1, 0.03), reg=1.0)
    X, U, hist = il.solve(x0, torch.from_numpy(U0).to(dev), iters=args.iters, al_every=4)
    h = hist.cpu().numpy()
    for i, row in enumerate(h):
        print(f"sweep {i:2d}  loss mean {row.mean():14.1f}  best {row.min():14.1f}")
    xN = X[-1].cpu().numpy()
    b = int(h[-1].argmin())
    print(f"best instance {b}: final p = ({xN[0, b]:.1f}, {xN[1, b]:.1f}, {xN[2, b]:.1f}) m, v = ({xN[3, b]:.1f}, {xN[4, b]:.1f}, {xN[5, b]:.1f}) m/s, "
          f"goal (150, 0); v_x(N) < -2 violated by {float(il.update_goal_multiplier(X)[b]):.2f} m/s")


if __name__ == "__main__":
    main()

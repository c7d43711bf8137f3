#!/usr/bin/env python3
"""BASELINE configs[3]: H=100, B_total random-restart MPC instances sharded across the GPUs of one node, ONE RCCL
all-gather of each rank's best trajectories at the end.

    python examples/random_restart_mpc.py --batch 16384 --horizon 100            # 1 GPU: the whole batch
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           examples/random_restart_mpc.py --batch 16384 --horizon 100           # 8 GPUs: 2048 instances each

Every restart starts from the same aircraft state with a different random control sequence; each rank improves its shard
with a few batched iLQR iterations (no communication), then the ranks exchange their top-k records [cost, X, U].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--topk", type=int, default=4)
    ap.add_argument("--hidden", type=str, default="128,128,128,128")
    ap.add_argument("--model", type=str, default="nn", choices=["nn", "poly", "default"])
    ap.add_argument("--poly-path", type=str, default=os.path.join(ROOT, "tests", "golden", "poly_coef.npz"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
    from aircraft_amd.control import ILQR, QuadraticCost
    from aircraft_amd.distributed import gather_best, shard_bounds
    from aircraft_amd.synthetic import GLIDER, TRIM_STATE

    if args.model == "nn":
        path = MlpData.synthetic(tuple(int(h) for h in args.hidden.split(",")), seed=42)
    else:
        path = args.poly_path if args.model == "poly" else ""
    ac = Aircraft(AircraftOpts(coeff_model_type=args.model, coeff_model_path=path,
                               aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
    H = args.horizon
    T = H * 0.01
    cost = QuadraticCost.goal((50.0 * T, 2.0), w_goal=1.0, height=-200.0, w_height=1.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
    solver = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))

    lo, hi = shard_bounds(args.batch, rank, world)
    B = hi - lo
    rng = np.random.default_rng(1234)              # the same stream on every rank; each rank takes its slice
    U_all = np.zeros((H, 7, args.batch), dtype=np.float32)
    U_all[:, :3] = np.clip(np.cumsum(rng.normal(0, 0.3, (H, 3, args.batch)), axis=0), -5, 5)
    x0 = torch.from_numpy(np.repeat(TRIM_STATE[:, None], B, axis=1).astype(np.float32)).to(dev)
    U0 = torch.from_numpy(np.ascontiguousarray(U_all[:, :, lo:hi])).to(dev)

    # untimed warm-up on the real shapes: device workspaces (several GB at this size) are allocated on first use
    solver.solve(x0, U0, iters=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    X, U, hist = solver.solve(x0, U0, iters=args.iters)
    c, Xb, Ub = gather_best(X, U, None, k=args.topk, cost=hist[-1].contiguous(), system=ac)   # the one collective
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"world": world, "batch_total": args.batch, "batch_per_rank": B, "horizon": H, "iters": args.iters,
                          "seconds": el, "restart_solves_per_s": args.batch / el,
                          "initial_cost_median": float(hist[0].nanmedian()), "final_cost_median": float(hist[-1].nanmedian()),
                          "best_costs": [float(v) for v in c[: args.topk]],
                          "gathered_records": int(c.numel()), "record_bytes": int((1 + Xb[0].numel() + Ub[0].numel()) * 4)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — headline benchmark of the MPC hot path on MI355X.

Metric (BASELINE.json): MPC horizon-steps/sec (6-DoF + NN-surrogate RK4) at H=50, batch=4096.
One "step" of this bench = one pass of the hot path over one batch: the fused
`state_update + A,B sensitivities` kernel evaluated on every (instance, node) pair of the
multiple-shooting transcription, B x H = 4096 x 50 = 204 800 units, through the 5-128-128-128-128-6
tanh surrogate in fp32 (BASELINE configs[2]).  value = units / second (whole job, all ranks).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Weak scaling: every rank owns its own B = 4096 instances (independent MPC instances / random restarts shard
with no data-path collective); after the K timed sweeps — once per solve, inside the timed region — the ranks
all-gather their best trajectory record over RCCL (the one exchange the path has).  Inputs are synthetic (seed 42)
and resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="MPC instances per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--hidden", type=str, default="128,128,128,128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the second-order / closed-loop figures reported alongside")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-mfma", action="store_true", help="VALU matmul path (validation baseline)")
    return ap.parse_args()


def extras(ac, ms, X, U, dev):
    """Reported alongside the headline, N=1 only, a few seconds: the second-order blocks of the same units and the
    receding-horizon closed loop of BASELINE configs[4] (B=1024, H=50, hipGraph replay).  Never fails the bench."""
    import torch

    out = {}
    try:
        H, _, B = U.shape
        Lam = torch.randn((H, 13, B), device=dev)
        Hz = torch.empty((H, 21, 21, B), device=dev)
        ms.hessian(X, U, Lam, out=Hz); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ms.hessian(X, U, Lam, out=Hz)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 3
        out["second_order_blocks_units_per_s"] = B * H / t * 1e3
        out["second_order_blocks_ms"] = t
        del Hz, Lam
    except Exception as e:  # noqa: BLE001
        out["second_order_blocks_error"] = repr(e)
    try:
        from aircraft_amd.control import ILQR, QuadraticCost, RecedingHorizon

        Bc, Hc = 1024, 50
        cost = QuadraticCost.goal((30.0, 0.5), w_goal=1.0, height=-200.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
        il = ILQR(system=ac, dt=ms.dt, num_nodes=Hc, cost=cost, alphas=(1.0, 0.5, 0.1))
        x0 = X[0, :, :Bc].contiguous()
        U0 = U[:Hc, :, :Bc].contiguous()
        # eager on purpose (the captured loop runs the same 6.7 ms): rocprofv3 on this image crashes tracing a process
        # that captures and replays hipGraphs, and bench.py must stay profilable
        loop = RecedingHorizon(il, overlap=30, iterations=2).allocate(x0, U0)
        loop.run(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop.run(300); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 300
        out["closed_loop_ms_per_solve"] = t
        out["closed_loop"] = "B=1024 x H=50, 2 iLQR iterations per solve, overlap 30, 300 solves (eager launches; hipGraph replay: tools/bench_modes.py cfg5)"
    except Exception as e:  # noqa: BLE001
        out["closed_loop_error"] = repr(e)
    return out


def cpu_baseline(ac, X, U, dt, seconds):
    """Time the float64 oracle (the CPU port of the reference arithmetic) on a bounded sample of the same units."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from tests.helpers import make_oracle

    o = make_oracle(ac)
    # threads actually usable by this process (affinity mask), not every core the host shows
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    if usable < orc.num_threads():
        orc.set_num_threads(usable)
    cores = orc.num_threads()
    n_probe = min(X.shape[1], 64 * cores)
    t0 = time.perf_counter()
    o.step_sens(X[:, :n_probe], U[:, :n_probe], dt)
    rate = n_probe / (time.perf_counter() - t0)
    n = int(min(X.shape[1], max(n_probe, rate * seconds)))
    reps = max(1, int(np.ceil(rate * seconds / n)))  # repeat the sample until ~`seconds` of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        o.step_sens(X[:, :n], U[:, :n], dt)
    el = time.perf_counter() - t0
    return {"value": n * reps / el, "unit": "horizon-steps/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {n} of the same (x_k,u_k) units, step+A,B,c sensitivities, float64 C++ oracle "
                      f"(g++ -O3 -mavx2, OpenMP {cores} threads), {el:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on cuda:0, collectives over gloo
    backend = os.environ.get("AIRCRAFT_BENCH_BACKEND", "nccl")
    if os.environ.get("AIRCRAFT_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
    from aircraft_amd.control import MultipleShooting
    from aircraft_amd.distributed import gather_best
    from aircraft_amd.synthetic import GLIDER, synthetic_controls, synthetic_states

    hidden = tuple(int(h) for h in args.hidden.split(","))
    mlp = MlpData.synthetic(hidden, seed=42)
    opts = AircraftOpts(coeff_model_type="nn", coeff_model_path=mlp, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                        physical_integration_substeps=1, use_mfma=not args.no_mfma)
    ac = Aircraft(opts)
    B, H, dt = args.batch, args.horizon, 0.01
    ms = MultipleShooting(system=ac, dt=dt, num_nodes=H, opts={"quaternion": "integration"})  # normalise on (mhtt.py:60)

    # synthetic, in-envelope shooting nodes: every (instance, node) pair gets an independent state/control
    rng = np.random.default_rng(42 + rank)
    Xh = synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2)  # (H+1, 13, B)
    Uh = synthetic_controls(H, B, rng)
    X = torch.from_numpy(np.ascontiguousarray(Xh, dtype=np.float32)).to(dev)
    U = torch.from_numpy(np.ascontiguousarray(Uh, dtype=np.float32)).to(dev)
    F = torch.empty((H, 13, B), device=dev)
    A = torch.empty((H, 13, 13, B), device=dev)
    Bm = torch.empty((H, 13, 7, B), device=dev)
    out = (F, A, Bm, None)
    goal = torch.tensor([150.0, 10.0, -190.0], device=dev)

    ms.linearise(X, U, out=out)
    name, grid, block, lds = ac.last_launch()  # the dominant kernel of a step
    for _ in range(args.warmup):
        ms.linearise(X, U, out=out)
    if world > 1:
        # warm the communicator: the path's one exchange is an all-gather of every rank's best trajectory record
        # (cost, X[H+1,13], U[H,7]) over RCCL, once per solve — here: once after the K timed sweeps, inside the timing
        gather_best(X, U, goal, k=1, system=ac)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # HIP events on the stream the kernel is launched on (torch's current stream)
        ms.linearise(X, U, out=out)
        ev[i][1].record()
    if world > 1:
        best = gather_best(X, U, goal, k=1, system=ac)
        assert best[0].numel() == world
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # reported alongside the headline: forward-only passes of the same units, untimed by the driver
    def _time(fn, iters=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    traj = torch.empty((H + 1, 13, B), device=dev)
    x0 = X[0].contiguous()
    fwd_ms = _time(lambda: ms.propagate(X, U, out=F))
    roll_ms = _time(lambda: ms.rollout(x0, U, out=traj), 5)

    # HBM traffic of the dominant kernel from the PMC passes (FETCH_SIZE x2 per the gfx950 guide + WRITE_SIZE, calibrated
    # on a known-byte-count kernel of the same access pattern): collected separately with rocprofv3 --pmc and committed
    # under profiles/; it applies to the default workload only.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath) and (B, H, hidden) == (4096, 50, (128, 128, 128, 128)) and not args.no_mfma:
        traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
    units_per_step = B * H * world
    value = units_per_step * args.steps / elapsed
    F_mlp = mlp.flops_forward()
    flops_unit = 24 * F_mlp + 30000          # SURVEY.md §8d contract figure (4 stages x (1 value + 5 tangents))
    bytes_unit = 1172                         # read x,u (80 B) + write x+, A, B (1092 B)
    achieved_tflops = flops_unit * B * H / (kern_ms * 1e-3) / 1e12
    achieved_gbs = bytes_unit * B * H / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        res = {
            "metric": "MPC horizon-steps/sec (6-DoF+NN-surrogate RK4, step + A,B sensitivities) at H=50, batch=4096",
            "value": value, "unit": "horizon-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3: multiple-shooting defect+Jacobian pass, B={B}/GPU x H={H} units, "
                                   f"MLP 5-{'-'.join(map(str, hidden))}-6 tanh fp32, dt=0.01, 1 RK4 sub-step, q normalised",
                       "batch_per_gpu": B, "horizon": H, "units_per_step": units_per_step,
                       "mfma": not args.no_mfma, "parallelism": f"instances sharded x{world}, one all-gather of best records per solve"},
            "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": bytes_unit * B * H,
                         "kernel": name, "kernel_ms": kern_ms, "grid": grid, "block": block, "lds_bytes": lds,
                         "kernel_note": "kernel_ms spans the whole step: k_nn_step_sens on the units that fill whole rounds of "
                                        "64-unit workgroups over the CUs, then k_nn_step_sens_pair on a remainder of at most "
                                        "half a round (grid/block/lds are the first kernel's)",
                         "flops_per_unit": flops_unit, "hbm_bytes_per_unit": bytes_unit,
                         "hbm_achieved_GBs": achieved_gbs, "hbm_frac": achieved_gbs / PEAK_HBM_GBS},
        }
        res["alongside"] = {
            "forward_shooting_steps_per_s_per_gpu": B * H / fwd_ms * 1e3, "forward_shooting_ms": fwd_ms,
            "forward_flops_per_unit": 4 * F_mlp + 1500,
            "forward_tflops": (4 * F_mlp + 1500) * B * H / (fwd_ms * 1e-3) / 1e12,
            "rollout_steps_per_s_per_gpu": B * H / roll_ms * 1e3, "rollout_ms": roll_ms,
            "note": "same units, x+ only: ac_shoot_step_f32 (all nodes independent) and ac_rollout_f32 (sequential in k)"}
        if world == 1 and not args.no_extras:
            res["alongside"].update(extras(ac, ms, X, U, dev))
        if not args.no_cpu_baseline and world == 1:
            Xs = Xh[:H].transpose(1, 0, 2).reshape(13, H * B)
            Us = Uh.transpose(1, 0, 2).reshape(7, H * B)
            res["cpu_baseline"] = cpu_baseline(ac, np.ascontiguousarray(Xs), np.ascontiguousarray(Us), dt,
                                               args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

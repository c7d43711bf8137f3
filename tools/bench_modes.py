#!/usr/bin/env python3
"""Secondary measurements (not the headline): every mode of the hot path on one MI355X.
Prints one JSON object per line; run on the GPU box."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states
from tests.helpers import make_aircraft

dev = torch.device("cuda", 0)

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters  # ms

def timeit_graph(fn, launches=10, replays=6):
    """The same launches captured in one hipGraph and replayed: what the kernel costs when no host call sits between two
    launches (the Python -> ctypes -> hipLaunch path of one call is 20-40 us, a tenth of the shorter kernels here)."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(launches): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (replays * launches)  # ms

def problem(B, H, seed=42):
    rng = np.random.default_rng(seed)
    Xh = synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2)
    Uh = synthetic_controls(H, B, rng)
    return (torch.from_numpy(np.ascontiguousarray(Xh, dtype=np.float32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(Uh, dtype=np.float32)).to(dev))

def run(name, ac, B, H, iters=10):
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    X, U = problem(B, H)
    F = torch.empty((H, 13, B), device=dev); A = torch.empty((H, 13, 13, B), device=dev); Bm = torch.empty((H, 13, 7, B), device=dev)
    traj = torch.empty((H + 1, 13, B), device=dev)
    x0 = X[0].contiguous()
    t_sens_host = timeit(lambda: ms.linearise(X, U, out=(F, A, Bm, None)), iters)
    k_sens = ac.last_launch()
    try:
        t_sens = timeit_graph(lambda: ms.linearise(X, U, out=(F, A, Bm, None)))
        how = "10 launches captured in one hipGraph, 6 replays"
    except Exception as e:  # (a flavour that cannot be captured: keep the host-launched figure)
        t_sens, how = t_sens_host, f"host-launched ({type(e).__name__})"
    t_fwd = timeit(lambda: ms.propagate(X, U, out=F), iters)
    k_fwd = ac.last_launch()
    t_roll = timeit(lambda: ms.rollout(x0, U, out=traj), max(2, iters // 2))
    k_roll = ac.last_launch()
    n = B * H
    # SURVEY §8d contract figures: 1 172 B and 24 F_mlp + 30 000 flop per unit with sensitivities; the governing roofline is
    # HBM without a network (25.6 flop/B is at the ridge) and fp32 compute with one
    f_mlp = ac.coefficient_model.data.flops_forward() if getattr(ac.coefficient_model, "data", None) is not None else 0
    flops = 24 * f_mlp + 30000
    if f_mlp:
        ach = flops * n / (t_sens * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "flops_per_unit": flops}
    else:
        ach = 1172 * n / (t_sens * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "bytes_per_unit": 1172,
                "fp32_frac": flops * n / (t_sens * 1e-3) / 1e12 / 157.3}
    print(json.dumps({"case": name, "B": B, "H": H, "roofline": roof,
                      "sens_steps_per_s": n / t_sens * 1e3, "sens_ms": t_sens, "sens_timing": how, "sens_ms_host_launched": t_sens_host,
                      "sens_kernel": k_sens[0],
                      "fwd_steps_per_s": n / t_fwd * 1e3, "fwd_ms": t_fwd,
                      "rollout_steps_per_s": n / t_roll * 1e3, "rollout_ms": t_roll, "rollout_grid": k_roll[1]}), flush=True)

# usage: bench_modes.py [cases...]   or   bench_modes.py --model poly   (one model's line with its roofline object)
args = sys.argv[1:]
if "--model" in args:
    args = [args[args.index("--model") + 1]]
which = args or ["cfg3", "cfg2", "cfg2_valu", "real", "poly", "default", "linear", "cfg5"]
if "cfg3" in which: run("cfg3 4x128 mfma", make_aircraft("nn", hidden=(128,) * 4), 4096, 50)
if "cfg2" in which:
    run("cfg2 3x64 mfma B=256", make_aircraft("nn", hidden=(64,) * 3), 256, 50)
    run("cfg2 3x64 mfma B=4096", make_aircraft("nn", hidden=(64,) * 3), 4096, 50)
if "cfg2_valu" in which:
    run("cfg2 3x64 VALU (mfma off) B=256", make_aircraft("nn", hidden=(64,) * 3, use_mfma=False), 256, 50, iters=4)
    run("cfg2 3x64 VALU (mfma off) B=4096", make_aircraft("nn", hidden=(64,) * 3, use_mfma=False), 4096, 50, iters=3)
if "real" in which: run("real 5-16-32-6", make_aircraft("nn"), 4096, 50)
for m in ("poly", "default", "linear"):
    if m in which: run(m, make_aircraft(m), 4096, 50)
if "cfg4" in which: run("cfg4 per-GPU shard 4x128 B=2048 H=100", make_aircraft("nn", hidden=(128,) * 4), 2048, 100)
if "cfg5" in which:
    # receding-horizon closed loop (main/mhe/mhtt.py:79-124): sequential MPC solves; each solve = 2 iLQR iterations
    # (linearise + Riccati + closed-loop line-search rollouts + cost) and a shift by N - overlap.
    from aircraft_amd.control import ILQR, QuadraticCost
    ac = make_aircraft("nn", hidden=(128,) * 4)
    B, H = 1024, 50
    cost = QuadraticCost.goal((30.0, 0.5), w_goal=1.0, height=-200.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
    il = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))
    X, U = problem(B, H)
    from aircraft_amd.control import RecedingHorizon
    x0 = X[0].contiguous().clone(); U = U.contiguous()
    def loop_ms(capture, solves):
        loop = RecedingHorizon(il, overlap=30, iterations=2).allocate(x0, U)
        if capture:
            loop.capture()
        loop.run(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop.run(solves); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / solves
    t_eager = loop_ms(False, 50)
    t_graph = loop_ms(True, 1000)  # BASELINE configs[4]: 1000 sequential solves, one captured cycle replayed
    print(json.dumps({"case": "cfg5 closed loop B=1024 H=50: RecedingHorizon (2 iLQR iterations + shift + tail rollout), per solve over 1000 solves",
                      "eager_ms": t_eager, "graph_ms": t_graph, "solves_per_s_graph": 1e3 / t_graph,
                      "instance_solves_per_s": B * 1e3 / t_graph}), flush=True)

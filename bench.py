#!/usr/bin/env python3
"""bench.py — headline benchmark of the MPC hot path on MI355X.

Metric (BASELINE.json): MPC horizon-steps/sec (6-DoF + NN-surrogate RK4) at H=50, batch=4096; 1/2/4/8 GPU.
One "step" of this bench = one pass of the hot path over one batch: the fused
`state_update + A,B sensitivities` kernel evaluated on every (instance, node) pair of the
multiple-shooting transcription, B x H = 4096 x 50 = 204 800 units, through the 5-128-128-128-128-6
tanh surrogate in fp32 (BASELINE configs[2]).  value = units / second (whole job, all ranks).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launching: under a launcher (RANK / WORLD_SIZE set) this process is one rank.  WITHOUT a launcher and with
--gpus N > 1, main() starts N fresh rank processes itself — as its very first action, before anything touches the
GPU (no re-exec of a process that initialised HIP) — waits for them and relays rank 0's JSON line.  A run whose
ranks do not add up to --gpus exits non-zero: it never prints a single-GPU number under a multi-GPU label.

Scaling: the top-level numbers are WEAK-scaled (every rank owns its own B = 4096 instances; independent MPC
instances / random restarts shard with no data-path collective).  For N > 1 the same invocation also times the
STRONG-scaled job the metric's wording names (B = 4096 in total, sharded with shard_bounds) and reports it in the
"strong" object of the same JSON line.  After the K timed sweeps — once per solve, inside the timed region — the
ranks all-gather their best trajectory record over RCCL (the one exchange the path has).  Inputs are synthetic
(seed 42) and resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, = fp32 vector peak
PEAK_HBM_GBS = 8000.0
METRIC = "MPC horizon-steps/sec (6-DoF+NN-surrogate RK4, step + A,B sensitivities) at H=50, batch=4096"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="MPC instances per GPU (weak) / in total (strong)")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--hidden", type=str, default="128,128,128,128")
    ap.add_argument("--scaling", choices=("both", "weak", "strong"), default="both",
                    help="N > 1: which job(s) to time; the top-level value is the weak one unless 'strong' is chosen")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the second-order / closed-loop figures reported alongside")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample (both legs together)")
    ap.add_argument("--no-mfma", action="store_true", help="VALU matmul path (BASELINE cfg2 'MFMA off')")
    return ap.parse_args(argv)


# ---- self-launch (no launcher present) ---------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, base_env=None, port=None):
    """Environment of each of the n rank processes main() starts when no launcher did (one process per GPU)."""
    port = port or _free_port()
    envs = []
    for r in range(n):
        e = dict(base_env if base_env is not None else os.environ)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        envs.append(e)
    return envs


def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script, relay rank 0's stdout, return the exit code.
    The parent never imports torch and never touches the GPU."""
    log(f"[bench] no launcher (RANK/WORLD_SIZE unset): starting {n} rank processes, rendezvous on 127.0.0.1")
    procs = []
    for r, env in enumerate(rank_environments(n)):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0 = ""
    rc = 0
    deadline = time.time() + float(os.environ.get("AIRCRAFT_BENCH_TIMEOUT", "900"))
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        for p in procs[1:]:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        log("[bench] rank processes timed out")
        rc = 124
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.kill()  # exactly the processes started above
            p.wait()
        if p.returncode != 0 and rc == 0:
            log(f"[bench] rank {r} exited with {p.returncode}")
            rc = p.returncode or 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if rc == 0 and not lines:
        log("[bench] rank 0 printed no JSON line")
        rc = 1
    if rc == 0:
        res = json.loads(lines[-1])
        if res.get("n_gpus") != n or res.get("ranks_seen") != n:
            log(f"[bench] asked for {n} GPUs but the job saw n_gpus={res.get('n_gpus')} ranks_seen={res.get('ranks_seen')}")
            rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


# ---- alongside figures (N = 1 only) ------------------------------------------------------------------------------
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
        # eager on purpose (the captured loop runs the same time): rocprofv3 on this image crashes tracing a process
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
    """Time the float64 oracle (the CPU port of the reference arithmetic) on a bounded sample of the same units:
    once on every host core this process may use and once on ONE core (SURVEY §8d asks for both)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc

    o = orc.for_aircraft(ac)
    # threads actually usable by this process (affinity mask), not every core the host shows
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    all_threads = min(usable, orc.num_threads()) if orc.num_threads() > 0 else usable

    kept = {}

    def leg(threads, budget, keep=False):
        orc.set_num_threads(threads)
        cores = orc.num_threads()
        n_probe = min(X.shape[1], 64 * cores)
        t0 = time.perf_counter()
        o.step_sens(X[:, :n_probe], U[:, :n_probe], dt)
        rate = n_probe / (time.perf_counter() - t0)
        n = int(min(X.shape[1], max(n_probe, rate * budget)))
        reps = max(1, int(np.ceil(rate * budget / n)))  # repeat the sample until ~`budget` seconds of CPU work
        t0 = time.perf_counter()
        for _ in range(reps):
            outs = o.step_sens(X[:, :n], U[:, :n], dt)
        el = time.perf_counter() - t0
        if keep:  # the oracle's results on the first n units: checked against the timed GPU outputs (parity_in_run)
            kept.update(n=n, Xn=outs[0], A=outs[1], B=outs[2])
        return {"value": n * reps / el, "unit": "horizon-steps/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x {n} of the same (x_k,u_k) units, step+A,B,c sensitivities, float64 C++ oracle "
                          f"(g++ -O3 -mavx2, OpenMP {cores} thread{'s' if cores > 1 else ''}), {el:.1f} s"}

    res = leg(all_threads, 0.5 * seconds, keep=True)
    res["one_core"] = leg(1, 0.5 * seconds)
    res["cfg1_per_call"] = cfg1_per_call(orc)
    orc.set_num_threads(all_threads)
    return res, kept


def cfg1_per_call(orc, seconds=1.0):
    """BASELINE configs[0] in the shape the reference actually runs it — ONE glider, H = 20, analytic
    (default) coefficients, one call of the step per node in a host loop (main/control/control.py:72-93), float64 oracle on
    one core.  The reference's own figure for this loop is ~4.5e5 steps/s in the CasADi VM."""
    import numpy as np
    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
    from aircraft_amd.synthetic import GLIDER

    ac1 = Aircraft(AircraftOpts(coeff_model_type="default", aircraft_config=AircraftConfiguration(dict(GLIDER)),
                                physical_integration_substeps=1))
    o = orc.for_aircraft(ac1)
    orc.set_num_threads(1)
    x0 = np.array([0, 0, -200, 50, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)[:, None]
    u = np.array([0, 3, 0, 0, 0, 0, 0], dtype=np.float64)[:, None]
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        x = x0
        for _ in range(20):
            x = o.state_update(x, u, 0.01)
        reps += 1
    el = time.perf_counter() - t0
    return {"value": 20 * reps / el, "unit": "horizon-steps/s", "cores": 1, "kind": "port",
            "sample": f"{reps} rollouts of B=1 x H=20 (trim state, elevator 3 deg, default model), one oracle call per node "
                      f"from a Python loop, {el:.1f} s", "us_per_call": el / (20 * reps) * 1e6}


def parity_in_run(kept, F, A, Bm, tol=1e-5):
    """Compare the outputs the TIMED run left in HBM (F, A, B of the last timed sweep) with the float64 oracle's results
    on the same units (the cpu_baseline sample: the first n units in node-major order).  Same metrics as the parity
    tests: block-relative state error (blocks p, v, q, omega; floors 1 m, 1 m/s, 1, 0.1 rad/s) and, per unit, the
    max-norm relative error of its A and B blocks."""
    import numpy as np

    n = kept["n"]
    H, _, B = F.shape
    Fg = F.permute(1, 0, 2).reshape(13, H * B)[:, :n].double().cpu().numpy()
    Ag = A.permute(1, 2, 0, 3).reshape(13, 13, H * B)[:, :, :n].double().cpu().numpy()
    Bg = Bm.permute(1, 2, 0, 3).reshape(13, 7, H * B)[:, :, :n].double().cpu().numpy()
    worst = 0.0
    for sl, floor in ((slice(0, 3), 1.0), (slice(3, 6), 1.0), (slice(6, 10), 1.0), (slice(10, 13), 0.1)):
        d = np.abs(Fg[sl] - kept["Xn"][sl]).max(axis=0)
        worst = max(worst, float((d / np.maximum(np.abs(kept["Xn"][sl]).max(axis=0), floor)).max()))

    def unit_rel(a, ref):
        d = np.abs(a - ref).reshape(-1, n).max(axis=0)
        return float((d / np.maximum(np.abs(ref).reshape(-1, n).max(axis=0), 1e-300)).max())

    ea, eb = unit_rel(Ag, kept["A"]), unit_rel(Bg, kept["B"])
    finite = bool(np.isfinite(Fg).all() and np.isfinite(Ag).all() and np.isfinite(Bg).all())
    return {"units": int(n), "state_block_rel_max": worst, "A_unit_rel_max": ea, "B_unit_rel_max": eb, "tol": tol,
            "ok": bool(finite and worst <= tol and ea <= tol and eb <= tol),
            "note": "F, A, B exactly as the last TIMED sweep left them in HBM (the alongside passes write other buffers) vs the "
                    "float64 oracle on the same first `units` (x_k,u_k) units; every unit on its own (max norm), nothing masked"}


def strong_model(batch, H, world, headline_kernel, cus=256):
    """What DESIGN.md §7 predicts for the strong-scaled job before launch and gather costs, so that a SCALE run explains
    itself: the headline kernel runs in rounds of 64 units per CU; a remainder of at most half a round goes to the
    wave-pair kernel at ~0.58 of a round, a larger one costs a whole round."""
    if not headline_kernel:
        return None
    per_round = 64 * cus

    def cost(units):
        full, rem = divmod(units, per_round)
        return full + (0.0 if rem == 0 else (0.58 if rem <= per_round // 2 else 1.0))

    one = cost(batch * H)
    out = {"round_units": per_round, "rounds_one_gpu": batch * H / per_round, "cost_one_gpu_rounds": one, "per_n": {}}
    for n in (1, 2, 4, 8):
        units = -(-batch // n) * H
        out["per_n"][str(n)] = {"rounds_per_rank": units / per_round, "cost_rounds": cost(units), "predicted_speedup": one / cost(units)}
    out["this_run"] = out["per_n"].get(str(world))
    out["note"] = ("predicted_speedup excludes launch and the one all-gather per solve (gather_ms); >= 6x at N = 8 has under "
                   "5 % margin (DESIGN.md §7)")
    return out


# ---- one rank ----------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] error: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to label a {world}-rank run as {args.gpus} GPUs")
        return 2
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on cuda:0, collectives over gloo
    backend = os.environ.get("AIRCRAFT_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("AIRCRAFT_BENCH_ONE_GPU") == "1"
    if not one_gpu and torch.cuda.device_count() < world:  # device_count() does not initialise HIP on this image
        log(f"[bench] error: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible "
            "(AIRCRAFT_BENCH_ONE_GPU=1 AIRCRAFT_BENCH_BACKEND=gloo rehearses on one)")
        return 3
    if one_gpu:
        local_rank = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
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
    from aircraft_amd.distributed import gather_best, shard_bounds
    from aircraft_amd.synthetic import GLIDER, synthetic_controls, synthetic_states

    hidden = tuple(int(h) for h in args.hidden.split(","))
    mlp = MlpData.synthetic(hidden, seed=42)
    opts = AircraftOpts(coeff_model_type="nn", coeff_model_path=mlp, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                        physical_integration_substeps=1, use_mfma=not args.no_mfma)
    ac = Aircraft(opts)
    H, dt = args.horizon, 0.01
    ms = MultipleShooting(system=ac, dt=dt, num_nodes=H, opts={"quaternion": "integration"})  # normalise on (mhtt.py:60)
    goal = torch.tensor([150.0, 10.0, -190.0], device=dev)
    red_dev = dev if backend == "nccl" else "cpu"

    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, device=red_dev, dtype=torch.int64)
        dist.all_reduce(t)
        ranks_seen = int(t.item())
        if ranks_seen != args.gpus:
            log(f"[bench] error: {ranks_seen} ranks joined, --gpus {args.gpus}")
            return 4

    def timed_pass(B, seed):
        """K timed sweeps of the step + A,B kernel over this rank's B instances x H nodes (+ the one all-gather)."""
        rng = np.random.default_rng(seed)
        # synthetic, in-envelope shooting nodes: every (instance, node) pair gets an independent state/control
        Xh = synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2)  # (H+1, 13, B)
        Uh = synthetic_controls(H, B, rng)
        X = torch.from_numpy(np.ascontiguousarray(Xh, dtype=np.float32)).to(dev)
        U = torch.from_numpy(np.ascontiguousarray(Uh, dtype=np.float32)).to(dev)
        F = torch.empty((H, 13, B), device=dev)
        A = torch.empty((H, 13, 13, B), device=dev)
        Bm = torch.empty((H, 13, 7, B), device=dev)
        out = (F, A, Bm, None)
        ms.linearise(X, U, out=out)
        launch = ac.last_launch()  # the dominant kernel of a step
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
        gather = {}
        if world > 1:
            best = gather_best(X, U, goal, k=1, system=ac, timing=gather)   # cost + select/pack kernels, ONE all-gather, merge
            assert best[0].numel() == world
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return {"elapsed": elapsed, "kern_ms": kern_ms, "launch": launch, "gather_ms": gather.get("gather_ms"), "X": X, "U": U, "F": F, "A": A, "Bm": Bm, "Xh": Xh, "Uh": Uh, "B": B}

    F_mlp = mlp.flops_forward()
    flops_unit = 24 * F_mlp + 30000          # SURVEY.md §8d contract figure (4 stages x (1 value + 5 tangents))
    bytes_unit = 1172                         # read x,u (80 B) + write x+, A, B (1092 B)

    modes = ["weak"] if world == 1 else (["weak", "strong"] if args.scaling == "both" else [args.scaling])
    results = {}
    for mode in modes:
        if mode == "weak":
            B_local, total_B = args.batch, args.batch * world
        else:
            lo, hi = shard_bounds(args.batch, rank, world)
            B_local, total_B = hi - lo, args.batch
        r = timed_pass(B_local, 42 + rank + (1000 if mode == "strong" else 0))
        r["units_per_step"] = total_B * H
        r["value"] = total_B * H * args.steps / r["elapsed"]
        results[mode] = r
    primary = results[modes[0]]
    B = primary["B"]
    kern_ms = primary["kern_ms"]
    name, grid, block, lds = primary["launch"]
    X, U, Fbuf = primary["X"], primary["U"], primary["F"]

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
    Falong = torch.empty_like(Fbuf)  # the alongside passes write HERE: F, A, B of the timed sweeps stay as the timed kernel left them
    fwd_ms = _time(lambda: ms.propagate(X, U, out=Falong))
    roll_ms = _time(lambda: ms.rollout(x0, U, out=traj), 5)

    # HBM traffic of the dominant kernel from the PMC passes (FETCH_SIZE x2 per the gfx950 guide + WRITE_SIZE, calibrated
    # on a known-byte-count kernel of the same access pattern): collected separately with rocprofv3 --pmc and committed
    # under profiles/; it applies to the default workload only.
    traffic, traffic_src = None, None
    from aircraft_amd.build import source_sha
    this_build = source_sha()
    if (B, H, hidden) == (4096, 50, (128, 128, 128, 128)) and not args.no_mfma and not os.environ.get("AIRCRAFT_HIP_LIB"):
        for cand in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", cand)
            if not os.path.exists(tpath):
                continue
            rec = json.load(open(tpath))
            if rec.get("source_sha") == this_build:   # measured on THIS build's kernels: quote it
                traffic = rec["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{cand}, source_sha {this_build}"
            else:                                      # a stale figure would mislabel this build: say so instead
                traffic_src = (f"none for this build (source_sha {this_build}); last recorded: profiles/{cand} = "
                               f"{rec['traffic_bytes_per_launch']:.4g} B at source_sha {rec.get('source_sha', 'unrecorded')}")
            break
    achieved_tflops = flops_unit * B * H / (kern_ms * 1e-3) / 1e12
    achieved_gbs = bytes_unit * B * H / (kern_ms * 1e-3) / 1e9

    parity_failed = False
    if rank == 0:
        scaling = modes[0]
        res = {
            "metric": METRIC,
            "value": primary["value"], "unit": "horizon-steps/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": primary["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3: multiple-shooting defect+Jacobian pass, B={B}/GPU x H={H} units, "
                                   f"MLP 5-{'-'.join(map(str, hidden))}-6 tanh fp32, dt=0.01, 1 RK4 sub-step, q normalised",
                       "batch_per_gpu": B, "horizon": H, "units_per_step": primary["units_per_step"],
                       "mfma": not args.no_mfma,
                       "parallelism": f"instances sharded x{world}, one all-gather of best records per solve"},
            "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_unit": f"HBM bytes per launch (PMC: {traffic_src})" if traffic_src else None,
                         "algorithmic_bytes_per_launch": bytes_unit * B * H,
                         "kernel": name, "kernel_ms": kern_ms, "grid": grid, "block": block, "lds_bytes": lds,
                         "kernel_note": "kernel_ms spans the whole step (HIP events around one ac_shoot_sens_f32 call, rank 0): "
                                        "the dispatcher may split the units over k_nn_step_sens (64-unit workgroups, whole "
                                        "rounds over the CUs) and k_nn_step_sens_pair (32-unit workgroups, the remainder); "
                                        "grid/block/lds are the first kernel's",
                         "flops_per_unit": flops_unit, "hbm_bytes_per_unit": bytes_unit,
                         "hbm_achieved_GBs": achieved_gbs, "hbm_frac": achieved_gbs / PEAK_HBM_GBS},
        }
        if world > 1:
            res["gather_ms"] = primary["gather_ms"]
            res["gather_note"] = ("rank 0, HIP events around the path's one exchange, once per solve inside the timed region: "
                                  "k_traj_cost + k_best_records (select + pack) + one all_gather_into_tensor of the "
                                  f"{(1 + (H + 1) * 13 + H * 7) * 4} B record per rank + k_merge_records")
        if world > 1:
            res["strong_model"] = strong_model(args.batch, H, world, hidden == (128, 128, 128, 128) and not args.no_mfma)
        if "strong" in results and scaling != "strong":
            s = results["strong"]
            res["strong"] = {"value": s["value"], "unit": "horizon-steps/s", "scaling": "strong",
                             "ms_per_step": s["elapsed"] / args.steps * 1e3, "batch_total": args.batch,
                             "batch_per_gpu": s["B"], "units_per_step": s["units_per_step"], "kernel_ms_rank0": s["kern_ms"], "gather_ms": s["gather_ms"],
                             "note": "the SAME B x H units as the 1-GPU run, sharded over the ranks (the metric's wording); "
                                     "speed-up over N=1 = this value / the N=1 run's value"}
        res["alongside"] = {
            "forward_shooting_steps_per_s_per_gpu": B * H / fwd_ms * 1e3, "forward_shooting_ms": fwd_ms,
            "forward_flops_per_unit": 4 * F_mlp + 1500,
            "forward_tflops": (4 * F_mlp + 1500) * B * H / (fwd_ms * 1e-3) / 1e12,
            "rollout_steps_per_s_per_gpu": B * H / roll_ms * 1e3, "rollout_ms": roll_ms,
            "note": "same units, x+ only: ac_shoot_step_f32 (all nodes independent) and ac_rollout_f32 (sequential in k)"}
        if world == 1 and not args.no_extras:
            res["alongside"].update(extras(ac, ms, X, U, dev))
        if not args.no_cpu_baseline and world == 1:
            # the SAME numbers the GPU got: rounded to fp32 once, handed to the float64 oracle
            Xh = primary["Xh"].astype(np.float32).astype(np.float64)
            Uh = primary["Uh"].astype(np.float32).astype(np.float64)
            Xs = Xh[:H].transpose(1, 0, 2).reshape(13, H * B)
            Us = Uh.transpose(1, 0, 2).reshape(7, H * B)
            # F, A, B are still what the LAST TIMED sweep wrote (nothing after the timed region writes them)
            res["cpu_baseline"], kept = cpu_baseline(ac, np.ascontiguousarray(Xs), np.ascontiguousarray(Us), dt,
                                                     args.cpu_seconds)
            res["parity_in_run"] = parity_in_run(kept, Fbuf, primary["A"], primary["Bm"])
            parity_failed = not res["parity_in_run"]["ok"]
        print(json.dumps(res), flush=True)
        if parity_failed:
            log(f"[bench] PARITY FAILURE: {res['parity_in_run']}")
    if world > 1:
        dist.destroy_process_group()
    return 5 if parity_failed else 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        log("[bench] --gpus must be >= 1")
        return 2
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, argv)  # first thing: nothing has touched the GPU in this process
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())

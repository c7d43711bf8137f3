"""GPU tests of the raw C ABI: status codes, model-data requirements, handles and streams."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import block_rel_err, f32_exact, make_aircraft, make_oracle, synthetic_units

pytestmark = pytest.mark.gpu


def test_status_codes(gpu):
    import torch

    from aircraft_amd import _lib

    lib = _lib.load()
    ac = make_aircraft("default")
    ac._sync()
    h = ac._handle
    X = torch.zeros((13, 8), device=gpu); U = torch.zeros((7, 8), device=gpu); out = torch.empty_like(X)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.ac_step_f32(h, X.data_ptr(), U.data_ptr(), C.c_float(0.01), None, 8, out.data_ptr(), st) == 0
    assert lib.ac_step_f32(h, None, U.data_ptr(), C.c_float(0.01), None, 8, out.data_ptr(), st) == -1
    assert lib.ac_step_f32(h, X.data_ptr(), U.data_ptr(), C.c_float(0.01), None, -1, out.data_ptr(), st) == -1
    assert lib.ac_step_f32(None, X.data_ptr(), U.data_ptr(), C.c_float(0.01), None, 8, out.data_ptr(), st) == -1
    assert lib.ac_step_f32(h, None, None, C.c_float(0.01), None, 0, None, st) == 0  # empty batch is fine
    assert lib.ac_step_sens_f32(h, X.data_ptr(), U.data_ptr(), C.c_float(0.01), None, 8, out.data_ptr(), None, None, None, st) == -1
    # a model kind whose data was never supplied
    p = ac._param_struct()
    for kind in (1, 2, 3):
        p.model_kind = kind
        h2 = C.c_void_p()
        assert lib.ac_create(C.byref(p), C.byref(h2)) == 0
        assert lib.ac_step_f32(h2, X.data_ptr(), U.data_ptr(), C.c_float(0.01), None, 8, out.data_ptr(), st) == -4
        assert lib.ac_destroy(h2) == 0
    # MLP topology checks
    h3 = C.c_void_p(); p.model_kind = 2
    assert lib.ac_create(C.byref(p), C.byref(h3)) == 0
    fp = C.POINTER(C.c_float)
    W = np.zeros((6, 4), dtype=np.float32); b = np.zeros(6, dtype=np.float32); sc = np.ones(6, dtype=np.float32)
    widths = (C.c_int * 2)(4, 6); act = (C.c_int * 1)(0)
    Wp = (fp * 1)(W.ctypes.data_as(fp)); bp = (fp * 1)(b.ctypes.data_as(fp)); s = sc.ctypes.data_as(fp)
    assert lib.ac_set_mlp(h3, 1, widths, act, Wp, bp, s, s, s, s, 1) == -1  # input width must be 5
    assert lib.ac_set_mlp(h3, 9, widths, act, Wp, bp, s, s, s, s, 1) == -1  # too many layers
    assert lib.ac_destroy(h3) == 0
    arch = C.create_string_buffer(64)
    assert lib.ac_device_arch(arch, 64) == 0 and arch.value.startswith(b"gfx950")


def test_two_handles_two_streams(gpu):
    """Handles share no state: two models driven from two streams, interleaved, give what they give alone."""
    import torch

    a, b = make_aircraft("poly", normalise=True), make_aircraft("nn", hidden=(64, 64, 64), normalise=True)
    X, U = synthetic_units(3000, seed=31); X = f32_exact(X); U = f32_exact(U)
    Xd = torch.from_numpy(X).float().to(gpu); Ud = torch.from_numpy(U).float().to(gpu)
    ra, rb = a.state_update(Xd, Ud, 0.01).clone(), b.state_update(Xd, Ud, 0.01).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(5):
        with torch.cuda.stream(s1):
            oa = a.state_update(Xd, Ud, 0.01)
        with torch.cuda.stream(s2):
            ob = b.state_update(Xd, Ud, 0.01)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa, ra) and torch.equal(ob, rb)
    assert block_rel_err(ra.cpu().numpy(), make_oracle(a).state_update(X, U, 0.01)) < 1e-5


def test_single_layer_and_odd_width_networks(gpu):
    """Engine shapes off the beaten path: a single Linear(5,6) 'network' and hidden widths that are not multiples of 16
    (zero-padded by the host)."""
    from aircraft_amd import MlpData

    rng = np.random.default_rng(5)
    sc = ([1745.4, 3.7e-3, 0.0, 0.0, 7.1e-2], [954.0, 0.116, 0.121, 1.755, 2.84], [-0.116, 0, -0.184, 0, -0.0176, 0],
          [0.0895, 0.0332, 0.6166, 0.0391, 0.2478, 0.0057])
    nets = [MlpData([rng.normal(0, 0.3, (6, 5))], [rng.normal(0, 0.1, 6)], [0], *sc),
            MlpData([rng.normal(0, 0.4, (20, 5)), rng.normal(0, 0.2, (37, 20)), rng.normal(0, 0.2, (6, 37))],
                    [rng.normal(0, 0.1, 20), rng.normal(0, 0.1, 37), rng.normal(0, 0.1, 6)], [1, 1, 0], *sc),
            MlpData([rng.normal(0, 0.4, (100, 5)), rng.normal(0, 0.1, (6, 100))], [rng.normal(0, 0.1, 100), rng.normal(0, 0.1, 6)],
                    [1, 0], *sc)]
    import torch
    from tests.helpers import rel_fro, unit_max_rel
    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
    from aircraft_amd.synthetic import GLIDER

    X, U = synthetic_units(150, seed=37); X = f32_exact(X); U = f32_exact(U)
    Xd = torch.from_numpy(X).float().to(gpu); Ud = torch.from_numpy(U).float().to(gpu)
    for net in nets:
        ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=net, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                                   physical_integration_substeps=1))
        ac.normalise = True
        orc = make_oracle(ac)
        Xn, A, Bm, c = ac.step_sens(Xd, Ud, 0.01)
        Xr, Ar, Br, cr = orc.step_sens(X, U, 0.01)
        assert block_rel_err(Xn.cpu().numpy(), Xr) < 1e-5
        assert rel_fro(A.cpu().numpy(), Ar) < 1e-4 and rel_fro(Bm.cpu().numpy(), Br) < 1e-4
        assert unit_max_rel(A.cpu().numpy(), Ar).max() < 1e-4 and unit_max_rel(Bm.cpu().numpy(), Br).max() < 1e-4  # per unit
        assert block_rel_err(ac.state_update(Xd, Ud, 0.01).cpu().numpy(), Xr) < 1e-5
        Ur = np.repeat(U[None], 6, axis=0)
        roll = ac.rollout(Xd, torch.from_numpy(Ur).float().to(gpu), 0.01).cpu().numpy()
        assert block_rel_err(roll[1], Xr) < 1e-5


def test_status_codes_of_the_widened_entry_points(gpu):
    """Track / MHTT / Hessian / Newton entry points: empty batches, NULL arguments, missing track, bad modes."""
    import torch

    from aircraft_amd import _lib

    lib = _lib.load()
    ac = make_aircraft("poly")
    ac._sync()
    h = ac._handle
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda *s: torch.zeros(s, device=gpu)  # noqa: E731
    B, H = 8, 3
    X, U, Lam, Hz = f(H + 1, 13, B), f(H, 7, B), f(H, 13, B), f(H, 21, 21, B)
    S, s0, J = f(H + 1, B), f(B), f(B)
    w = _lib.MhttWeights(10, 5, 2, 50, 20, 10, 100)
    cost = _lib.IlqrCost()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    nul = C.c_void_p(0)
    # no track installed yet
    assert lib.ac_track_progress_f32(h, C.byref(w), p(X), p(s0), C.c_float(0.01), B, H, 1, p(S), nul, nul, nul, nul, nul, st) == -4
    assert lib.ac_mhtt_loss_f32(h, C.byref(w), p(X), p(U), p(S), B, H, p(J), st) == -4
    assert lib.ac_track_eval_f32(h, p(s0), B, p(f(3, B)), p(f(3, B)), st) == -4
    fp = C.POINTER(C.c_float)
    coef = np.zeros((2, 3, 4), dtype=np.float32); coef[:, 0, 1] = 1.0
    assert lib.ac_set_track(h, 2, coef.ctypes.data_as(fp), C.c_float(0.0)) == -1      # length must be positive
    assert lib.ac_set_track(h, 0, coef.ctypes.data_as(fp), C.c_float(2.0)) == -1
    assert lib.ac_set_track(h, 2, coef.ctypes.data_as(fp), C.c_float(2.0)) == 0
    assert lib.ac_track_progress_f32(h, C.byref(w), p(X), p(s0), C.c_float(0.01), B, H, 1, p(S), nul, nul, nul, nul, nul, st) == 0
    assert lib.ac_track_progress_f32(h, C.byref(w), p(X), p(s0), C.c_float(0.01), B, H, 2, p(S), nul, nul, nul, nul, nul, st) == -1  # mode
    assert lib.ac_track_progress_f32(h, C.byref(w), p(X), p(s0), C.c_float(0.01), B, H, 1, p(S), nul, nul, p(X), nul, nul, st) == -1  # model arrays: all three or none
    assert lib.ac_track_progress_f32(h, None, nul, nul, C.c_float(0.01), 0, H, 1, nul, nul, nul, nul, nul, nul, st) == 0  # empty batch
    assert lib.ac_mhtt_loss_f32(h, C.byref(w), p(X), p(U), p(S), B, H, p(J), st) == 0
    assert lib.ac_mhtt_loss_f32(h, None, p(X), p(U), p(S), B, H, p(J), st) == -1
    # Hessians
    assert lib.ac_shoot_hess_f32(h, p(X), p(U), C.c_float(0.01), nul, p(Lam), B, H, p(Hz), st) == 0
    assert lib.ac_shoot_hess_f32(h, p(X), p(U), C.c_float(0.01), nul, nul, B, H, p(Hz), st) == -1
    assert lib.ac_shoot_hess_f32(h, nul, nul, C.c_float(0.01), nul, nul, 0, H, nul, st) == 0
    assert lib.ac_step_hess_f32(h, p(X), p(U), C.c_float(0.01), nul, p(Lam), -1, p(Hz), st) == -1
    # Newton sweep pieces
    A, Bm = f(H, 13, 13, B), f(H, 13, 7, B)
    K, kff, dV = f(H, 7, 13, B), f(H, 7, B), f(2, B)
    assert lib.ac_ilqr_costate_f32(h, C.byref(cost), nul, nul, nul, p(X), p(A), B, H, p(Lam), st) == 0
    assert lib.ac_ilqr_costate_f32(h, C.byref(cost), p(X), nul, nul, p(X), p(A), B, H, p(Lam), st) == -1
    assert lib.ac_ilqr_backward_newton_f32(h, C.byref(cost), nul, nul, nul, p(Hz), p(X), p(U), p(A), p(Bm), B, H, p(K), p(kff), p(dV), st) == 0
    assert lib.ac_ilqr_backward_newton_f32(h, C.byref(cost), nul, nul, nul, p(Hz), p(X), p(U), p(A), p(Bm), B, 0, p(K), p(kff), p(dV), st) == -1
    assert lib.ac_ilqr_cost_node_f32(h, C.byref(cost), p(X), p(X), p(X), 0, p(X), p(U), B, H, p(J), st) == -1  # Bn must be > 0
    torch.cuda.synchronize()


def test_plain_c_program_against_the_python_host_path(gpu, tmp_path):
    """examples/abi_demo.c drives the library from C with hipMalloc'd buffers (no Python, no torch types in the
    signatures); its results must be what the Python host layer gets for the same inputs."""
    import json
    import os
    import shutil
    import subprocess

    import torch

    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
    from tests.helpers import ROOT

    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no C toolchain / HIP headers on this box")
    exe = str(tmp_path / "abi_demo")
    lib_dir = os.path.join(ROOT, "aircraft_amd")
    subprocess.run(["gcc", "-O2", os.path.join(ROOT, "examples", "abi_demo.c"), "-I" + os.path.join(ROOT, "include"),
                    "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib_dir, "-laircraft_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    res = json.loads(out)
    assert res["arch"].startswith("gfx950") and res["n"] == 1000
    # the same problem through the Python host layer
    N, H = 1000, 20
    i = np.arange(N, dtype=np.float32); t = i / np.float32(N)
    phi = np.float32(0.2) * np.sin(np.float32(7.0) * t); th = np.float32(0.05) * np.cos(np.float32(5.0) * t)
    X = np.zeros((13, N), dtype=np.float32)
    X[0] = 10 * t; X[1] = -5 * t; X[2] = -200; X[3] = 40 + 20 * t; X[4] = 1 - 2 * t; X[5] = 0.5
    q = np.stack([0.5 * phi, 0.5 * th, 0.1 * t, np.ones(N, dtype=np.float32)]).astype(np.float32)
    X[6:10] = q / np.sqrt((q * q).sum(axis=0, dtype=np.float32))
    X[10] = 0.1 * phi; X[11] = 0.05; X[12] = -0.02 * t
    U = np.zeros((7, N), dtype=np.float32)
    U[0] = 2 * np.sin(np.float32(3.0) * t); U[1] = -1 + t; U[2] = 0.5 * t
    cfg = AircraftConfiguration({"mass": 3.3, "reference_area": 0.238, "span": 1.75, "chord": 0.1375, "Ixx": 0.155,
                                 "Iyy": 0.16, "Izz": 0.3, "Ixz": 0.01, "aero_centre_offset": [0.0, 0.0, 0.0]})
    ac = Aircraft(AircraftOpts(coeff_model_type="default", aircraft_config=cfg, physical_integration_substeps=1))
    ac.normalise = True
    Xd, Ud = torch.from_numpy(X).to(gpu), torch.from_numpy(U).to(gpu)
    Xn, A, Bm, c = ac.step_sens(Xd, Ud, 0.01)
    traj = ac.rollout(Xd, Ud[None].expand(H, -1, -1).contiguous(), 0.01)
    fro = lambda a: float(np.sqrt((a.cpu().numpy().astype(np.float64) ** 2).sum()))  # noqa: E731
    # sinf/cosf of libm vs numpy differ in the last bit of a few inputs: compare to 1e-5, not bit for bit
    assert np.allclose(res["x_next_unit0"], Xn[:, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    for key, val in (("frob_xn", fro(Xn)), ("frob_A", fro(A)), ("frob_B", fro(Bm)), ("frob_c", fro(c)),
                     ("frob_rollout_end", fro(traj[-1]))):
        assert abs(res[key] - val) <= 1e-5 * abs(val), key

"""CPU tests of the host layer: the C-ABI library loads and exports every declared symbol, the plugin surface
mirrors the reference's, the model-data loaders work.  No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.helpers import GOLDEN, ROOT, golden, make_aircraft


def test_library_exports_every_declared_symbol():
    from aircraft_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "aircraft_hip.h")).read()
    declared = set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", header))
    declared -= {"ac_handle", "ac_params"}
    assert declared, "no prototypes parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ac_version().startswith(b"aircraft_hip")


def test_struct_layout_matches_header():
    from aircraft_amd import _lib

    # 4 + 9 + 9 + 3 + 1 + 1 + 3 floats, 4 ints
    assert ctypes.sizeof(_lib.AcParams) == (4 + 9 + 9 + 3 + 1 + 1 + 3) * 4 + 4 * 4


def test_no_gpu_calls_fail_loudly():
    """Without a GPU the product path raises; it never falls back to the oracle or any CPU code."""
    import torch

    from aircraft_amd import AircraftHipError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ac = make_aircraft("default")
    with pytest.raises(AircraftHipError):
        ac.state_update(np.zeros((13, 4)), np.zeros((7, 4)), 0.01)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aircraft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libaircraft_oracle" not in txt, f


def test_plugin_surface_mirrors_reference():
    from aircraft_amd import COEFF_MODEL_REGISTRY, Aircraft, AircraftConfiguration, AircraftOpts

    assert set(COEFF_MODEL_REGISTRY) == {"linear", "poly", "nn", "default"}  # coefficient_models.py:32-37
    opts = AircraftOpts(coeff_model_type="no-such-model", aircraft_config=AircraftConfiguration({"mass": 4.0}))
    ac = Aircraft(opts)
    assert ac.model_kind == "default"  # unknown key falls back silently (aircraft.py:37)
    assert ac.mass == 4.0 and opts.mass == 4.0  # mass comes from the aircraft config (aircraft.py:34)
    assert ac.num_states == 13 and ac.num_controls == 7
    f = ac.state_update
    assert (f.size1_in(0), f.size1_in(1), f.size1_in(2)) == (13, 7, 1)  # control/base.py:188-189
    assert ac.state_derivative.size1_in(0) == 13
    assert ac.physical_integration_substeps == 10 and ac.normalise is False  # dynamics/base.py:12, 33
    for name in ("v_frd_rel", "airspeed", "alpha", "beta", "qbar", "coefficients", "forces_frd", "moments_frd",
                 "phi", "theta", "psi"):
        assert callable(getattr(ac, name))


def test_inertia_tensor_with_com_shift():
    ac = make_aircraft("default")
    I = ac.inertia_tensor
    x, y, z = ac.com
    m = ac.mass
    assert np.isclose(I[0, 0], 0.155 + m * (y * y + z * z))
    assert np.isclose(I[0, 2], 0.01 - m * x * z) and np.isclose(I[2, 0], I[0, 2])
    assert np.allclose(ac.inverse_inertia_tensor @ I, np.eye(3), atol=1e-12)
    p = ac._param_struct()
    assert np.allclose(np.array(p.inertia[:]).reshape(3, 3), I, rtol=1e-6)


def test_param_struct_tracks_attribute_changes():
    ac = make_aircraft("poly")
    a = bytes(ac._param_struct())
    ac.com = np.array([0.05, 0.0, 0.01]); ac.normalise = True; ac.physical_integration_substeps = 3
    p = ac._param_struct()
    assert bytes(p) != a and p.normalise == 1 and p.substeps == 3 and abs(p.com[0] - 0.05) < 1e-7


def test_controller_sets_normalise_like_reference():
    from aircraft_amd.control import MultipleShooting

    ac = make_aircraft("poly")
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=50, opts={"quaternion": "integration"})
    assert ac.normalise is True and ms.state_dim == 13 and ms.control_dim == 7
    MultipleShooting(system=ac, dt=0.01, num_nodes=50, opts={"quaternion": "constraint"})
    assert ac.normalise is False
    assert np.isclose(MultipleShooting.dt_from_progress(10.0), 0.01)  # dt = 1/progress^2


def test_model_loaders(tmp_path):
    from aircraft_amd import MlpData, load_linear, load_model, load_poly

    coef, ic = load_poly(os.path.join(GOLDEN, "poly_coef.npz"))
    assert coef.shape == (6, 34) and np.isclose(ic[0], -0.023165109898439548)
    assert load_linear(os.path.join(GOLDEN, "linearised.npz")).shape == (6, 6)
    csv = tmp_path / "lin.csv"
    W = golden("linearised.npz")["W"]
    csv.write_text("q,alpha,beta,aileron,elevator,intercept\n" + "\n".join(",".join(repr(float(v)) for v in r) for r in W))
    assert np.array_equal(load_linear(str(csv)), W)
    # a reference-format .pth checkpoint (keys of train_nn_surrogate.py:245-251) round-trips through load_model
    import torch

    w = golden("scaledmodel_weights.npz")
    sd = {"core_layers.0.weight": torch.tensor(w["W0"]), "core_layers.0.bias": torch.tensor(w["b0"]),
          "core_layers.1.weight": torch.tensor(w["W1"]), "core_layers.1.bias": torch.tensor(w["b1"]),
          "core_layers.3.weight": torch.tensor(w["W2"]), "core_layers.3.bias": torch.tensor(w["b2"])}
    ck = {"model_state_dict": sd, **{k: torch.tensor(w[k]) for k in ("input_mean", "input_std", "output_mean", "output_std")}}
    pth = tmp_path / "m.pth"
    torch.save(ck, pth)
    m = load_model(str(pth))
    assert m.widths == [5, 16, 32, 6] and m.act == [0, 1, 0] and m.flops_forward() == 1568
    syn = MlpData.synthetic((128, 128, 128, 128))
    assert syn.flops_forward() == 101120 and MlpData.synthetic((64, 64, 64)).flops_forward() == 17792
    assert np.array_equal(syn.weights[0], MlpData.synthetic((128, 128, 128, 128)).weights[0])  # seeded


def test_restricted_unpickler_blocks_code_execution(tmp_path):
    import pickle

    from aircraft_amd.utils import load_poly

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))

    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps({"fitted_models": Evil()}))
    with pytest.raises(pickle.UnpicklingError):
        load_poly(str(p))


def test_restricted_unpickler_blocks_numpy_gadgets(tmp_path):
    """Only the six numpy reconstructors are allowed, not the numpy package: numpy.testing's runstring and
    numpy.load(allow_pickle=True) would otherwise execute code from a crafted .pkl (round-1 advisor finding)."""
    import pickle

    import numpy.testing._private.utils as ntu

    from aircraft_amd.utils import load_poly

    marker = tmp_path / "pwned"

    class RunString:
        def __reduce__(self):
            return (ntu.runstring, (f"open({str(marker)!r}, 'w').write('x')", {}))

    class NpLoad:
        def __reduce__(self):
            return (np.load, (str(tmp_path / "inner.npy"),), {"allow_pickle": True})

    for evil in (RunString(), NpLoad()):
        p = tmp_path / "evil.pkl"
        p.write_bytes(pickle.dumps({"fitted_models": evil}))
        with pytest.raises(pickle.UnpicklingError):
            load_poly(str(p))
    assert not marker.exists()
    # a legitimate payload (dict of numpy arrays / scalars) still loads
    good = {"fitted_models": {k: {"coef": np.arange(34.0) + i, "intercept": np.float64(i)} for i, k in
                              enumerate(["CX", "CY", "CZ", "Cl", "Cm", "Cn"])}}
    p = tmp_path / "good.pkl"
    p.write_bytes(pickle.dumps(good))
    coef, ic = load_poly(str(p))
    assert coef.shape == (6, 34) and np.array_equal(ic, np.arange(6.0))


def test_synthetic_inputs_are_in_envelope_and_seeded():
    from aircraft_amd.synthetic import synthetic_problem, synthetic_units

    X, U = synthetic_units(500, seed=42)
    X2, _ = synthetic_units(500, seed=42)
    assert np.array_equal(X, X2)
    assert np.abs(np.linalg.norm(X[6:10], axis=0) - 1).max() < 1e-12
    V = np.linalg.norm(X[3:6], axis=0)
    assert V.min() >= 30 and V.max() <= 80 and (X[2] < 0).all() and np.abs(U[:3]).max() <= 5
    X0, Us = synthetic_problem(16, 50)
    assert X0.shape == (13, 16) and Us.shape == (50, 7, 16) and not Us[:, 3:].any()


def test_abi_status_codes_without_gpu():
    """Argument checking and the no-device status are observable without a GPU (no compute call is made)."""
    import ctypes as C

    import torch

    from aircraft_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ac_create(None, C.byref(h)) == -1  # AC_ERR_BAD_ARG
    p = make_aircraft("default")._param_struct()
    assert lib.ac_create(C.byref(p), None) == -1
    bad = make_aircraft("default")._param_struct(); bad.substeps = 0
    assert lib.ac_create(C.byref(bad), C.byref(h)) == -1
    bad.substeps = 1; bad.model_kind = 9
    assert lib.ac_create(C.byref(bad), C.byref(h)) == -1
    assert lib.ac_destroy(None) == -1
    if not torch.cuda.is_available():
        assert lib.ac_create(C.byref(p), C.byref(h)) == -5  # AC_ERR_NO_DEVICE
        assert b"no HIP device" in lib.ac_last_error()


# ---- trajectory files in the reference's HDF5 layout ---------------------------------------------
def _need_hdf5():
    from aircraft_amd import trajectory_io

    if not trajectory_io.hdf5_available():
        pytest.skip("no HDF5 C library in this image")
    return trajectory_io


def test_trajectory_io_reads_reference_file():
    """tests/golden/simulation.h5 is the reference's own stored rollout (main/dynamics/dynamics.py:134-145)."""
    tio = _need_hdf5()
    path = os.path.join(os.path.dirname(__file__), "golden", "simulation.h5")
    assert tio.list_iterations(path) == [0]
    t = tio.load_trajectory(path, 0)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "simulation_h5.npz"))
    assert np.array_equal(t.state, g["state"]) and np.array_equal(t.control, g["control"])
    assert np.array_equal(t.times, g["times"]) and t.iteration == 0 and t.lam is None
    missing = tio.load_trajectory(path, 7)  # plotting.py:86-94: unknown iteration -> all None
    assert missing.state is None and missing.iteration is None


def test_trajectory_io_round_trip(tmp_path):
    tio = _need_hdf5()
    rng = np.random.default_rng(0)
    path = str(tmp_path / "traj.h5")
    X, U, t = rng.normal(size=(13, 51)), rng.normal(size=(7, 50)), 0.01 * np.arange(51)
    tio.save_trajectory(path, 0, X, U, t)
    tio.save_trajectory(path, 3, X[:, :5], U[:, :4], t[:5], compress=False, extra={"lam": np.ones((20, 5))})
    tio.save_trajectory(path, -2, X * 2, U * 2, None)  # negative iteration numbers occur (control/base.py:80)
    assert tio.list_iterations(path) == [-2, 0, 3]
    a = tio.load_trajectory(path, 0)
    assert np.array_equal(a.state, X) and np.array_equal(a.control, U) and np.array_equal(a.times, t)
    b = tio.load_trajectory(path, 3)
    assert b.state.shape == (13, 5) and np.array_equal(b.lam, np.ones((20, 5)))
    assert tio.load_trajectory(path, -2).times is None
    # rewriting an iteration replaces its datasets (control/base.py:101-103)
    tio.save_trajectory(path, 0, X[:, :7], U[:, :6], t[:7])
    assert tio.load_trajectory(path, 0).state.shape == (13, 7)
    # float32 torch tensors are accepted and stored as float64
    import torch
    tio.save_trajectory(path, 9, torch.ones(13, 4), torch.zeros(7, 3), torch.arange(4.0), mode="w")
    assert tio.list_iterations(path) == [9] and tio.load_trajectory(path, 9).state.dtype == np.float64
    with pytest.raises(FileNotFoundError):
        tio.load_trajectory(str(tmp_path / "nope.h5"), 0)


def test_bench_self_launch_plan_and_cpu_refusal():
    """bench.py --gpus N without a launcher: N rank environments (one process per GPU, rendezvous on 127.0.0.1), built
    before anything touches the GPU; on a box without GPUs the ranks fail loudly and the parent exits non-zero."""
    import subprocess
    import sys

    import bench

    envs = bench.rank_environments(4, base_env={"PATH": "/usr/bin"}, port=29999)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in envs)
    assert all(e["LOCAL_RANK"] == e["RANK"] and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999"
               and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert bench.parse(["--gpus", "8"]).scaling == "both"
    import torch

    if torch.cuda.is_available():
        return  # the GPU suite covers the live launch (tests/test_gpu_multigpu.py)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "starting 2 rank processes" in r.stderr


def test_bench_in_run_parity_gate_catches_one_bad_unit():
    """bench.py's parity_in_run (the figure the driver's bench line carries, exit code 5 when it fails): exact copies pass, ONE
    unit off by 3e-5 in one entry of A fails, a NaN fails, and only the first `n` units (the cpu_baseline sample) are read."""
    import importlib.util
    import os

    import torch

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(9)
    H, B, n = 3, 5, 11  # the sample covers the first 11 of 15 units (node-major)
    F = rng.normal(size=(H, 13, B)); A = rng.normal(size=(H, 13, 13, B)); Bm = rng.normal(size=(H, 13, 7, B))
    kept = {"n": n, "Xn": F.transpose(1, 0, 2).reshape(13, H * B)[:, :n].copy(),
            "A": A.transpose(1, 2, 0, 3).reshape(13, 13, H * B)[:, :, :n].copy(),
            "B": Bm.transpose(1, 2, 0, 3).reshape(13, 7, H * B)[:, :, :n].copy()}
    t = lambda a: torch.from_numpy(a.copy())  # noqa: E731
    ok = bench.parity_in_run(kept, t(F), t(A), t(Bm))
    assert ok["ok"] and ok["units"] == n and ok["state_block_rel_max"] == 0 and ok["A_unit_rel_max"] == 0
    bad = A.copy()
    k, b = 2, 0  # flat unit 2 * 5 + 0 = 10: the last one inside the sample
    bad[k, 4, 7, b] += 3e-5 * np.abs(A[k, :, :, b]).max()
    r = bench.parity_in_run(kept, t(F), t(bad), t(Bm))
    assert not r["ok"] and 2e-5 < r["A_unit_rel_max"] < 4e-5 and r["B_unit_rel_max"] == 0
    outside = A.copy(); outside[2, 4, 7, 1] += 1.0  # flat unit 11: outside the sample, not read
    assert bench.parity_in_run(kept, t(F), t(outside), t(Bm))["ok"]
    nan = F.copy(); nan[0, 3, 2] = np.nan
    assert not bench.parity_in_run(kept, t(nan), t(A), t(Bm))["ok"]
    small = F.copy(); small[1, 12, 3] += 5e-6  # omega block: floor 0.1 rad/s -> 5e-5 relative to the floor at most
    r = bench.parity_in_run(kept, t(small), t(A), t(Bm))
    assert r["state_block_rel_max"] > 0 and (r["ok"] == (r["state_block_rel_max"] <= 1e-5))


def test_quiet_capture_pauses_the_garbage_collector_and_restores_it(monkeypatch):
    """control.moving_horizon.quiet_capture: collector run before, disabled inside, restored after — also when the body or the
    capture itself raises (a finaliser's hipFree during a capture invalidates the graph; DESIGN §5 / INTEGRATION)."""
    import gc
    import types

    from aircraft_amd.control import moving_horizon as mh

    events = []

    class FakeGraphCtx:
        def __init__(self, graph, stream=None):
            events.append(("ctx", graph, stream))

        def __enter__(self):
            events.append(("enter", gc.isenabled()))
            return self

        def __exit__(self, *exc):
            events.append(("exit", gc.isenabled(), exc[0]))
            return False

    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(graph=FakeGraphCtx))
    monkeypatch.setattr(mh, "_torch", lambda: fake_torch)
    collected = []
    monkeypatch.setattr(gc, "collect", lambda *a: collected.append(True) or 0)
    assert gc.isenabled()
    with mh.quiet_capture("g", "s"):
        assert not gc.isenabled()
    assert gc.isenabled() and collected and events[0] == ("ctx", "g", "s") and events[1] == ("enter", False)
    with pytest.raises(RuntimeError):
        with mh.quiet_capture("g", "s"):
            raise RuntimeError("body failed")
    assert gc.isenabled() and events[-1][2] is RuntimeError
    gc.disable()  # a caller that runs with the collector off stays off
    try:
        with mh.quiet_capture("g", "s"):
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_handle_release_is_parked_while_a_stream_is_capturing(monkeypatch):
    """SixDOF.close() during a capture must not call ac_destroy (hipFree invalidates the capture): the handle is parked
    and destroyed by the next call that finds no capture open.  Host logic only: the library is replaced by a recorder."""
    import ctypes as C
    from aircraft_amd.dynamics import base as dyn

    destroyed = []

    class FakeLib:
        def ac_destroy(self, h):
            destroyed.append(h.value if hasattr(h, "value") else h)
            return 0

    monkeypatch.setattr(dyn._lib, "load", lambda: FakeLib())
    state = {"capturing": True}
    monkeypatch.setattr(dyn, "_capturing", lambda: state["capturing"])
    ac = make_aircraft("default")
    ac._handle = C.c_void_p(0x1234)
    ac.close()
    assert destroyed == [] and [h.value for h in dyn._PARKED] == [0x1234] and not ac._handle
    other = make_aircraft("default")
    other._handle = C.c_void_p(0x5678)
    state["capturing"] = False
    other.close()                       # no capture open: destroys its own handle and drains the parked one
    assert sorted(destroyed) == [0x1234, 0x5678] and dyn._PARKED == []


def test_receding_horizon_refuses_a_variable_time_solver():
    """The loop shifts / zeroes whole control columns and re-rolls at the fixed dt: a solver that carries dt_k in a control
    row (ILQR(time='variable')) would be linearised at dt_k = 0 after a 'zero' warm start — refused, as MHTT refuses it."""
    from aircraft_amd.control import RecedingHorizon

    class Solver:
        num_nodes, time_row = 50, 3

    with pytest.raises(ValueError, match="fixed-time"):
        RecedingHorizon(Solver(), overlap=30)
    Solver.time_row = 0
    assert RecedingHorizon(Solver(), overlap=30).keep == 20

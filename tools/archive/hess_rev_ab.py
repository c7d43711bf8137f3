#!/usr/bin/env python3
"""A/B of the width-128 second-order path: run with AIRCRAFT_HIP_LIB unset (reverse sweep) and with the -DAC_NO_HESS_REV
flavour (tools/variant_lib.sh: UNITS="aircraft_hip" ... norev -DAC_NO_HESS_REV; rev6 -DAC_HESS_REV6: the six-slab reverse sweep); each run saves its stage tensors and blocks
for a small batch and times the full-size call.   usage: hess_rev_ab.py <tag> [hidden widths ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData, _lib
from aircraft_amd.synthetic import GLIDER, synthetic_controls, synthetic_states

tag = sys.argv[1]
hidden = tuple(int(v) for v in sys.argv[2:]) or (128, 128, 128, 128)
dev = torch.device("cuda", 0)
ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=MlpData.synthetic(hidden, seed=42),
                           aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
ac.normalise = True
rng = np.random.default_rng(5)
out = {}
for n in (1000, 204800):
    X = torch.from_numpy(np.ascontiguousarray(synthetic_states(n, rng), dtype=np.float32)).to(dev)
    U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(1, n, rng)[0], dtype=np.float32)).to(dev)
    Lam = torch.from_numpy(rng.normal(size=(13, n)).astype(np.float32)).to(dev)
    Hz = ac.step_hess(X, U, 0.01, Lam)
    torch.cuda.synchronize()
    if n == 1000:
        ptr, fl = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.load().ac_hess_workspace(ac._handle, C.byref(ptr), C.byref(fl)), "ac_hess_workspace")
        ws = torch.empty(n * 504, device=dev)
        import torch.cuda
        C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(ws.data_ptr()), ptr, C.c_size_t(n * 504 * 4), 3)
        out["stage"] = ws.cpu().numpy().reshape(4, 126, n)
        out["Hz"] = Hz.cpu().numpy()
    else:
        t = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ac.step_hess(X, U, 0.01, Lam, out=Hz)
            torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
        print(f"{tag}: hidden {hidden}  n = {n}: step_hess {min(t):.2f} ms (min of 5; {', '.join('%.2f' % v for v in t)})", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"hess_rev_{tag}.npz"), **out)
for ref in ("norev", "rev6"):  # compare with whatever flavours ran before in this call
    other = os.path.join(ROOT, "gpurun_out", f"hess_rev_{ref}.npz")
    if ref == tag or not os.path.exists(other):
        continue
    print(f" {tag} against {ref}:")
    o = np.load(other)
    for k in ("stage", "Hz"):
        a, b = out[k].astype(np.float64), o[k].astype(np.float64)
        if k == "stage":
            for nm, sl in (("y", slice(0, 6)), ("J", slice(6, 36)), ("T", slice(36, 126))):
                d = np.abs(a[:, sl] - b[:, sl]).max(axis=(0, 1)); sc = np.abs(b[:, sl]).max(axis=(0, 1)) + 1e-30
                print(f"  {nm}: worst unit max|diff|/max|ref| = {(d / sc).max():.3e}   finite {np.isfinite(a[:, sl]).all()}")
        else:
            d = np.abs(a - b).reshape(-1, a.shape[-1]).max(axis=0); sc = np.abs(b).reshape(-1, a.shape[-1]).max(axis=0) + 1e-30
            print(f"  Hz: worst unit max|diff|/max|ref| = {(d / sc).max():.3e}   finite {np.isfinite(a).all()}")

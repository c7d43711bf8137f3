"""Batched 6-DoF NED rigid body on MI355X — the `SixDOF` plugin surface of the reference
(src/aircraft/dynamics/base.py:16-480) with the CasADi graph replaced by HIP kernels.

What stays the same for a caller (control/base.py:187-190 is the consumer):
    system.state_update          callable  F(x, u, dt) -> x+          with .size1_in(0) == 13, .size1_in(1) == 7
    system.state_derivative      callable  f(x, u)     -> x_dot
    system.normalise, .physical_integration_substeps, .com, .mass   plain attributes, read at call time
    getters v_frd_rel / airspeed / alpha / beta / qbar / coefficients / forces_frd / moments_frd / phi / theta / psi
What changes: arguments are column batches `(13, n)` / `(7, n)` (the reference's column-mapped call
convention, main/control/control.py:63) living on the GPU as float32 torch tensors; numpy in -> numpy out.
New on this surface: `rollout`, `step_sens` (A = dF/dx, B = dF/du, c = dF/ddt).
"""
from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from .. import _lib

NUM_STATES = _lib.NUM_STATES


@dataclass
class SixDOFOpts:
    """reference dynamics/base.py:9-14"""
    epsilon: float = 1e-6
    physical_integration_substeps: int = 10
    gravity: Sequence[float] = field(default_factory=lambda: (0.0, 0.0, 9.81))
    mass: float = 1.0


class BatchedFunction:
    """Stand-in for the `ca.Function` objects the reference hands to its controllers."""

    def __init__(self, name, in_dims, out_dim, fn):
        self._name, self._in_dims, self._out_dim, self._fn = name, tuple(in_dims), out_dim, fn

    def name(self):
        return self._name

    def size1_in(self, i):
        return self._in_dims[i]

    def size1_out(self, i=0):
        return self._out_dim

    def n_in(self):
        return len(self._in_dims)

    def __call__(self, *args, **kw):
        return self._fn(*args, **kw)

    def __repr__(self):
        return f"BatchedFunction({self._name}: {self._in_dims} -> {self._out_dim})"


def _torch():
    import torch

    return torch


# Handles whose release was requested while a stream was capturing.  ac_destroy frees device memory (hipFree), which is
# not permitted while a capture is open and invalidates it ("operation failed due to a previous error during capture") —
# and a release can be requested at any point of a capture: a cyclic-GC pass or a refcount reaching zero runs
# SixDOF.__del__ wherever it happens to.  Such handles are parked here and destroyed by the next call that finds no
# capture open (_drain_parked: every _sync(), every close()).
_PARKED: list = []


def _capturing() -> bool:
    import sys

    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    except Exception:
        return False


def _drain_parked() -> None:
    if _PARKED and not _capturing():
        lib = _lib.load()
        while _PARKED:
            lib.ac_destroy(_PARKED.pop())


class SixDOF(ABC):
    """Base class for batched 6-DoF dynamics in a NED frame."""

    num_states: int = NUM_STATES
    num_controls: int

    def __init__(self, *, opts: Optional[SixDOFOpts] = None, device=None, **kwargs) -> None:
        if opts is None:
            opts = SixDOFOpts()
        self.opts = opts
        self.gravity = tuple(float(g) for g in opts.gravity)
        self.epsilon = opts.epsilon
        self.mass = opts.mass
        self.normalise = False  # controllers flip this (control/base.py:182-185)
        self.physical_integration_substeps: int = opts.physical_integration_substeps
        self.stall_scaling = False
        self._handle = C.c_void_p()
        self._installed_key = None
        self._device = device

    # ---- subclass hooks ---------------------------------------------------------------------
    @abstractmethod
    def _param_struct(self) -> "_lib.AcParams":
        """Current constants as an ac_params (rebuilt whenever an attribute changed)."""

    @abstractmethod
    def _install_model(self) -> None:
        """Push coefficient-model data into the handle."""

    # ---- handle management ------------------------------------------------------------------
    def _device_obj(self):
        torch = _torch()
        if self._device is None:
            if not torch.cuda.is_available():
                raise _lib.AircraftHipError("no GPU visible: aircraft_amd runs on MI355X only (no CPU fallback)")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self._device)

    def _sync(self):
        """Create the handle on first use; re-send constants if any attribute changed since the last call."""
        lib = _lib.load()
        _drain_parked()
        p = self._param_struct()
        key = bytes(p)
        torch = _torch()
        dev = self._device_obj()
        if not self._handle:
            with torch.cuda.device(dev):
                _lib.check(lib.ac_create(C.byref(p), C.byref(self._handle)), "ac_create")
                self._install_model()
            self._installed_key = key
        elif key != self._installed_key:
            _lib.check(lib.ac_set_params(self._handle, C.byref(p)), "ac_set_params")
            self._installed_key = key
        return lib

    def close(self):
        if getattr(self, "_handle", None):
            if _capturing():
                _PARKED.append(C.c_void_p(self._handle.value))  # destroyed after the capture (see _PARKED)
            else:
                _lib.load().ac_destroy(self._handle)
            self._handle = C.c_void_p()
            # the workspaces died with the handle: a re-created one must be reserved again
            self._hess_reserved = 0
            self._installed_key = None
        _drain_parked()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tensor hand-off --------------------------------------------------------------------
    def _in_u(self, u):
        """Controls -> the 7-row device layout.  A plugin with fewer controls (the quadrotor's four thrusts) passes
        (num_controls, n); rows beyond are zero and ignored by its kernels.  A ready 7-row buffer passes through."""
        torch = _torch()
        rows = self.num_controls
        shape = tuple(getattr(u, "shape", ()))
        if rows < _lib.NUM_CONTROLS and len(shape) >= 1 and shape[0] == _lib.NUM_CONTROLS:
            rows = _lib.NUM_CONTROLS
        U, npu, vec = self._in(u, rows, "u")
        if rows < _lib.NUM_CONTROLS:
            U = torch.cat([U, torch.zeros((_lib.NUM_CONTROLS - rows, U.shape[1]), device=U.device, dtype=U.dtype)])
        return U, npu, vec

    def _in(self, a, rows, name):
        """-> (float32 contiguous cuda tensor (rows, n), came_from_numpy, was_vector)"""
        torch = _torch()
        dev = self._device_obj()
        from_np = not isinstance(a, torch.Tensor)
        t = torch.as_tensor(np.asarray(a, dtype=np.float32) if from_np else a)
        vec = t.dim() == 1
        if vec:
            t = t.reshape(-1, 1)
        if t.dim() != 2 or t.shape[0] != rows:
            raise ValueError(f"{name}: expected ({rows}, n), got {tuple(t.shape)}")
        t = t.to(device=dev, dtype=torch.float32).contiguous()
        return t, from_np, vec

    @staticmethod
    def _out(t, from_np, vec):
        if vec:
            t = t.reshape(t.shape[0] if t.dim() == 2 else t.shape[:-1])
        return t.cpu().numpy().astype(np.float64) if from_np else t

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self._device_obj()).cuda_stream)

    def _dt_args(self, dt, n):
        """scalar dt -> (dt, NULL) ; per-unit dt (n,) -> (0, ptr)."""
        torch = _torch()
        if isinstance(dt, torch.Tensor) and dt.numel() > 1 or (not isinstance(dt, torch.Tensor) and np.ndim(dt) > 0 and np.size(dt) > 1):
            t = torch.as_tensor(np.asarray(dt, dtype=np.float32) if not isinstance(dt, torch.Tensor) else dt)
            t = t.to(device=self._device_obj(), dtype=torch.float32).contiguous().reshape(-1)
            if t.numel() != n:
                raise ValueError(f"dt: expected a scalar or {n} values, got {t.numel()}")
            return C.c_float(0.0), C.c_void_p(t.data_ptr()), t
        return C.c_float(float(dt)), C.c_void_p(0), None

    # ---- the plugin surface -----------------------------------------------------------------
    @property
    def state_derivative(self) -> BatchedFunction:
        """x_dot = f(x, u)   (reference dynamics/base.py:385-406)"""

        def f(x, u):
            lib = self._sync()
            torch = _torch()
            X, npx, vec = self._in(x, self.num_states, "x")
            U, _, _ = self._in_u(u)
            n = X.shape[1]
            if U.shape[1] != n:
                raise ValueError("x and u must have the same number of columns")
            out = torch.empty_like(X)
            _lib.check(lib.ac_state_derivative_f32(self._handle, X.data_ptr(), U.data_ptr(), n, out.data_ptr(),
                                                   self._stream()), "ac_state_derivative_f32")
            return self._out(out, npx, vec)

        return BatchedFunction("dynamics", (self.num_states, self.num_controls), self.num_states, f)

    @property
    def state_update(self) -> BatchedFunction:
        """x+ = F(x, u, dt): RK4 with `physical_integration_substeps` sub-steps (reference dynamics/base.py:450-480)"""

        def F(x, u, dt):
            lib = self._sync()
            torch = _torch()
            X, npx, vec = self._in(x, self.num_states, "x")
            U, _, _ = self._in_u(u)
            n = X.shape[1]
            if U.shape[1] != n:
                raise ValueError("x and u must have the same number of columns")
            dts, dtp, keep = self._dt_args(dt, n)
            out = torch.empty_like(X)
            _lib.check(lib.ac_step_f32(self._handle, X.data_ptr(), U.data_ptr(), dts, dtp, n, out.data_ptr(),
                                       self._stream()), "ac_step_f32")
            del keep
            return self._out(out, npx, vec)

        return BatchedFunction("state_update", (self.num_states, self.num_controls, 1), self.num_states, F)

    def rollout(self, x0, U, dt, out=None):
        """X[k+1] = F(X[k], U[k], dt).  x0 (13, B); U (H, 7, B) -> X (H+1, 13, B).
        The batched form of Controller.initialise (reference main/control/control.py:72-93)."""
        lib = self._sync()
        torch = _torch()
        X0, npx, vec = self._in(x0, self.num_states, "x0")
        B = X0.shape[1]
        from_np = not isinstance(U, torch.Tensor)
        Ut = torch.as_tensor(np.asarray(U, dtype=np.float32) if from_np else U)
        if vec and Ut.dim() == 2:
            Ut = Ut.unsqueeze(-1)
        if Ut.dim() != 3 or Ut.shape[1] not in (self.num_controls, _lib.NUM_CONTROLS) or Ut.shape[2] != B:
            raise ValueError(f"U: expected (H, {self.num_controls}, {B}), got {tuple(Ut.shape)}")
        Ut = Ut.to(device=X0.device, dtype=torch.float32)
        if Ut.shape[1] < _lib.NUM_CONTROLS:  # fewer controls than device rows: zero-fill (see _in_u)
            Ut = torch.cat([Ut, torch.zeros((Ut.shape[0], _lib.NUM_CONTROLS - Ut.shape[1], B), device=Ut.device)], dim=1)
        Ut = Ut.contiguous()
        H = Ut.shape[0]
        if out is None:
            out = torch.empty((H + 1, self.num_states, B), device=X0.device, dtype=torch.float32)
        else:
            assert out.shape == (H + 1, self.num_states, B) and out.is_contiguous() and out.dtype == torch.float32
        _lib.check(lib.ac_rollout_f32(self._handle, X0.data_ptr(), Ut.data_ptr() if H else None, C.c_float(float(dt)),
                                      B, H, out.data_ptr(), self._stream()), "ac_rollout_f32")
        if npx:
            res = out.cpu().numpy().astype(np.float64)
            return res[:, :, 0] if vec else res
        return out[:, :, 0] if vec else out

    def step_sens(self, x, u, dt, want_c=True, out=None):
        """(x+, A, B, c): the step and its Jacobians A = dF/dx (13,13,n), B = dF/du (13,7,n),
        c = dF/ddt (13,n) — what `ca.jacobian(state_update, .)` yields in the reference
        (control/aircraft.py:85-95; the defect rows of control/base.py:279-280)."""
        lib = self._sync()
        torch = _torch()
        X, npx, vec = self._in(x, self.num_states, "x")
        U, _, _ = self._in_u(u)
        n = X.shape[1]
        if U.shape[1] != n:
            raise ValueError("x and u must have the same number of columns")
        dts, dtp, keep = self._dt_args(dt, n)
        ns, nc = self.num_states, _lib.NUM_CONTROLS
        fresh = out is None
        if out is None:
            Xn = torch.empty_like(X)
            A = torch.empty((ns, ns, n), device=X.device, dtype=torch.float32)
            Bm = torch.empty((ns, nc, n), device=X.device, dtype=torch.float32)
            c = torch.empty((ns, n), device=X.device, dtype=torch.float32) if want_c else None
        else:
            Xn, A, Bm, c = out
        _lib.check(lib.ac_step_sens_f32(self._handle, X.data_ptr(), U.data_ptr(), dts, dtp, n, Xn.data_ptr(),
                                        A.data_ptr(), Bm.data_ptr(), c.data_ptr() if c is not None else None,
                                        self._stream()), "ac_step_sens_f32")
        del keep
        if npx:
            cv = (lambda t: None if t is None else t.cpu().numpy().astype(np.float64))
            Xn, A, Bm, c = cv(Xn), cv(A), cv(Bm), cv(c)
        if fresh and self.num_controls < nc:
            Bm = Bm[:, : self.num_controls]  # the plugin's own controls; the remaining columns are zero
        if vec:
            Xn, A, Bm = Xn[..., 0], A[..., 0], Bm[..., 0]
            c = None if c is None else c[..., 0]
        return Xn, A, Bm, c

    def state_derivative_sens(self, x, u, out=None):
        """(x_dot, Fx, Fu): f(x, u) with Fx = df/dx (13,13,n) and Fu = df/du (13,7,n) — `ca.jacobian(state_derivative, .)`,
        which the reference's implicit defect row and Baumgarte row (control/base.py:282-304) and its LQR wrapper
        (dynamics/base.py:51-52) differentiate."""
        lib = self._sync()
        torch = _torch()
        X, npx, vec = self._in(x, self.num_states, "x")
        U, _, _ = self._in_u(u)
        n = X.shape[1]
        if U.shape[1] != n:
            raise ValueError("x and u must have the same number of columns")
        ns, nc = self.num_states, _lib.NUM_CONTROLS
        fresh = out is None
        if out is None:
            out = (torch.empty_like(X), torch.empty((ns, ns, n), device=X.device, dtype=torch.float32),
                   torch.empty((ns, nc, n), device=X.device, dtype=torch.float32))
        Xd, Fx, Fu = out
        _lib.check(lib.ac_state_derivative_sens_f32(self._handle, X.data_ptr(), U.data_ptr(), n, Xd.data_ptr(),
                                                    Fx.data_ptr(), Fu.data_ptr(), self._stream()),
                   "ac_state_derivative_sens_f32")
        if npx:
            Xd, Fx, Fu = (t.cpu().numpy().astype(np.float64) for t in (Xd, Fx, Fu))
        if fresh and self.num_controls < nc:
            Fu = Fu[:, : self.num_controls]
        if vec:
            Xd, Fx, Fu = Xd[..., 0], Fx[..., 0], Fu[..., 0]
        return Xd, Fx, Fu

    def step_hess(self, x, u, dt, lam, out=None):
        """Hessian of lam . F(x, u, dt) over z = (x[13], u[7], dt): (21, 21, n) — the block the defect rows contribute to
        the Lagrangian Hessian IPOPT evaluates as `nlp_hess_l` (control/base.py:279-280; todo.md:102).
        lam (13, n): multipliers of the rows of F.  Any number of RK4 sub-steps (composed per sub-step on the device)."""
        lib = self._sync()
        torch = _torch()
        X, npx, vec = self._in(x, self.num_states, "x")
        U, _, _ = self._in_u(u)
        L, _, _ = self._in(lam, self.num_states, "lam")
        n = X.shape[1]
        if U.shape[1] != n or L.shape[1] != n:
            raise ValueError("x, u and lam must have the same number of columns")
        dts, dtp, keep = self._dt_args(dt, n)
        if out is None:
            out = torch.empty((21, 21, n), device=X.device, dtype=torch.float32)
        self._reserve_hess(n)
        _lib.check(lib.ac_step_hess_f32(self._handle, X.data_ptr(), U.data_ptr(), dts, dtp, L.data_ptr(), n,
                                        out.data_ptr(), self._stream()), "ac_step_hess_f32")
        del keep
        res = out.cpu().numpy().astype(np.float64) if npx else out
        return res[..., 0] if vec else res

    def _reserve_hess(self, n: int) -> None:
        """Size the handle's second-order workspace for n units.  Host-side and idempotent (the library only grows
        it); the compute calls themselves never allocate, so call this before capturing a hipGraph."""
        n = int(n)
        ns = int(self.physical_integration_substeps or 1)
        if ns != getattr(self, "_hess_reserved_substeps", ns):
            self._hess_reserved = 0  # the composition buffers are sized for the sub-step count
        self._hess_reserved_substeps = ns
        if n > getattr(self, "_hess_reserved", 0):
            torch = _torch()
            if torch.cuda.is_current_stream_capturing():
                raise _lib.AircraftHipError("second-order workspace must be reserved before stream capture: "
                                            "call system._reserve_hess(n) (ac_reserve_hess_workspace) first")
            self._sync()  # push the current sub-step count first: the reservation is sized for it
            _lib.check(_lib.load().ac_reserve_hess_workspace(self._handle, n), "ac_reserve_hess_workspace")
            self._hess_reserved = n

    # ---- getters (reference dynamics/base.py:147-278, aircraft.py:255-330) ------------------------
    def _aero(self, x, u):
        lib = self._sync()
        torch = _torch()
        X, npx, vec = self._in(x, self.num_states, "x")
        if u is None:
            U = torch.zeros((_lib.NUM_CONTROLS, X.shape[1]), device=X.device, dtype=torch.float32)
        else:
            U, _, _ = self._in_u(u)
        n = X.shape[1]
        out = torch.empty((_lib.AERO_ROWS, n), device=X.device, dtype=torch.float32)
        _lib.check(lib.ac_aero_f32(self._handle, X.data_ptr(), U.data_ptr(), n, out.data_ptr(), self._stream()),
                   "ac_aero_f32")
        return out, npx, vec

    def _getter(self, name, rows):
        def fn(x, u=None):
            out, npx, vec = self._aero(x, u)
            return self._out(out[rows], npx, vec)

        dim = (rows.stop - rows.start) if isinstance(rows, slice) else 1
        return BatchedFunction(name, (self.num_states, self.num_controls), dim, fn)

    v_frd_rel = property(lambda self: self._getter("v_frd_rel", slice(0, 3)))
    airspeed = property(lambda self: self._getter("airspeed", 3))
    alpha = property(lambda self: self._getter("alpha", 4))
    beta = property(lambda self: self._getter("beta", 5))
    qbar = property(lambda self: self._getter("qbar", 6))
    coefficients = property(lambda self: self._getter("coefficients", slice(7, 13)))
    forces_frd = property(lambda self: self._getter("forces_frd", slice(13, 16)))
    moments_frd = property(lambda self: self._getter("moments_frd", slice(16, 19)))

    # Euler angles of the attitude (reference base.py:179-195); state-only getters f(x)
    phi = property(lambda self: self._getter("phi", 19))
    theta = property(lambda self: self._getter("theta", 20))
    psi = property(lambda self: self._getter("psi", 21))

    def last_launch(self):
        """(kernel name, grid, block, dynamic LDS bytes) of the most recent dispatch."""
        name = C.create_string_buffer(64)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        _lib.load().ac_last_launch(self._handle, name, 64, C.byref(g), C.byref(b), C.byref(l))
        return name.value.decode(), g.value, b.value, l.value

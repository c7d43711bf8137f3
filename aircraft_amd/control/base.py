"""Multiple-shooting loop over the batched dynamics — the part of the reference's `ControlProblem`
(src/aircraft/control/base.py:123-443) that sits ON the hot path:

    self.dynamics = system.state_update ; state_dim / control_dim from size1_in      base.py:187-190
    system.normalise = (opts['quaternion'] == 'integration')                         base.py:182-185
    defect rows  x_{k+1} - F(x_k, u_k, dt_k) = 0  for k = 0..N-1                      base.py:275-286, 423-443
    dt_k = 1 / progress_k**2                                                          base.py:276
    rollout initial guess  guess[:, i+1] = dynamics(guess[:13, i], guess[13:, i], dt) main/control/control.py:72-93

The NLP itself (Opti/IPOPT) is out of scope; what is built here are the residual and Jacobian blocks
the solver's nlp_g / nlp_jac_g evaluate, for B independent instances at once.
Trajectory buffers are rollout-shaped device tensors X (N+1, 13, B), U (N, 7, B), used in place.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib
from ..dynamics.base import SixDOF


def _torch():
    import torch

    return torch


class MultipleShooting:
    def __init__(self, *, system: SixDOF, dt: float = 0.01, num_nodes: int, opts: Optional[dict] = None, **kwargs):
        self.opts = opts if opts else {}
        # same side effect as the reference constructor (control/base.py:182-185)
        system.normalise = self.opts.get("quaternion", None) == "integration"
        self.system = system
        self.dynamics = system.state_update
        self.state_dim: int = self.dynamics.size1_in(0)
        self.control_dim: int = self.dynamics.size1_in(1)
        self.x_dot = system.state_derivative
        self.num_nodes = num_nodes
        self.dt = dt

    # ---- time handling ---------------------------------------------------------------------------
    @staticmethod
    def dt_from_progress(progress):
        """dt_k = 1 / progress_k^2  (control/base.py:276); fixed time uses progress = sqrt(1/dt) (:350)."""
        return 1.0 / (progress * progress)

    # ---- rollout (HOT LOOP #1) -------------------------------------------------------------------
    def rollout(self, x0, U, out=None):
        """x0 (13, B), U (N, 7, B) -> X (N+1, 13, B) on the device."""
        return self.system.rollout(x0, U, self.dt, out=out)

    def initialise(self, initial_state, controls=None):
        """Reference-shaped initial guess for ONE instance: array (state_dim + control_dim, N+1) whose state rows
        are the rollout of the control rows (main/control/control.py:72-93: aileron guess 1, elevator 0)."""
        N = self.num_nodes
        guess = np.zeros((self.state_dim + self.control_dim, N + 1))
        if controls is None:
            guess[13, :] = 1
            guess[14, :] = 0
        else:
            guess[self.state_dim:, :] = np.asarray(controls, dtype=np.float64).reshape(self.control_dim, N + 1)
        x0 = np.asarray(initial_state, dtype=np.float64).reshape(self.state_dim)
        U = np.ascontiguousarray(guess[self.state_dim:, :N].T[:, :, None])  # (N, 7, 1)
        traj = self.system.rollout(x0[:, None], U, self.dt)  # numpy in -> numpy out, (N+1, 13, 1)
        guess[: self.state_dim, :] = traj[:, :, 0].T
        return guess

    # ---- trajectory files (the reference's SaveMixin, control/base.py:30-114) ----------------------
    def save_progress(self, filepath, iteration, X, U, instance: int = 0, times=None, mode: str = "a"):
        """Write one instance of a batch as `iteration_<k>/{state (13,N+1), control (7,N), times}` in the
        reference's HDF5 layout, so its plotter (plotting/plotting.py:72-95) can read what was solved here."""
        from ..trajectory_io import save_trajectory

        Xi, Ui = X[:, :, instance].T, U[:, :, instance].T  # node-major buffers -> (13, N+1), (7, N)
        if times is None:
            times = self.dt * np.arange(Xi.shape[1])
        save_trajectory(filepath, iteration, Xi, Ui, times, mode=mode)

    # ---- defects and their Jacobian blocks (HOT LOOPS #2/#3) ---------------------------------------
    def _shoot_args(self, X, U, dt):
        torch = _torch()
        H, B = U.shape[0], U.shape[2]
        assert X.is_cuda and U.is_cuda and X.dtype == torch.float32 and U.dtype == torch.float32
        assert X.is_contiguous() and U.is_contiguous()
        assert X.shape[0] >= H and X.shape[1] == self.state_dim and X.shape[2] == B and U.shape[1] == _lib.NUM_CONTROLS  # device control rows
        if dt is None:
            dt = self.dt
        if isinstance(dt, torch.Tensor) and dt.numel() > 1:
            dtt = dt.to(device=X.device, dtype=torch.float32).contiguous()
            assert dtt.shape == (H, B)
            return H, B, C.c_float(0.0), C.c_void_p(dtt.data_ptr()), dtt
        return H, B, C.c_float(float(dt)), C.c_void_p(0), None

    def propagate(self, X, U, dt=None, out=None):
        """F(x_k, u_k, dt_k) for every node k < N of every instance: (N, 13, B)."""
        torch = _torch()
        lib = self.system._sync()
        H, B, dts, dtp, keep = self._shoot_args(X, U, dt)
        if out is None:
            out = torch.empty((H, self.state_dim, B), device=X.device, dtype=torch.float32)
        _lib.check(lib.ac_shoot_step_f32(self.system._handle, X.data_ptr(), U.data_ptr(), dts, dtp, B, H,
                                         out.data_ptr(), self.system._stream()), "ac_shoot_step_f32")
        del keep
        return out

    def defects(self, X, U, dt=None, integration: Optional[str] = None):
        """Defect rows of every node, (N, 13, B) — ControlProblem.state_constraint (control/base.py:275-286):
            'explicit'  r_k = x_{k+1} - F(x_k, u_k, dt_k)                      (:279-280)
            'implicit'  r_k = x_{k+1} - (x_k + dt_k f(x_{k+1}, u_k))           (:282-284)
        `integration` defaults to opts['integration'] (default 'explicit', as in the reference)."""
        mode = integration or self.opts.get("integration", "explicit")
        if mode not in ("explicit", "implicit"):
            raise NotImplementedError("Must choose integration mode from ['implicit', 'explicit']")  # base.py:286
        torch = _torch()
        lib = self.system._sync()
        H, B, dts, dtp, keep = self._shoot_args(X, U, dt)
        assert X.shape[0] == H + 1, "the defect rows pair every node with the next: X must hold N + 1 nodes"
        out = torch.empty((H, self.state_dim, B), device=X.device, dtype=torch.float32)
        fn = lib.ac_shoot_defect_f32 if mode == "explicit" else lib.ac_shoot_implicit_defect_f32
        # one launch: the step (explicit) or derivative (implicit) kernel stores the row in place of its result
        _lib.check(fn(self.system._handle, X.data_ptr(), U.data_ptr(), dts, dtp, B, H, out.data_ptr(), self.system._stream()),
                   fn.__name__)
        del keep
        return out

    def derivative(self, Xn, Un, out=None):
        """f(x, u) at the H (state, control) pairs Xn (H, 13, B) [a view such as X[1:] is used in place], Un (H, 7, B)."""
        torch = _torch()
        lib = self.system._sync()
        H, B = Un.shape[0], Un.shape[2]
        assert Xn.is_cuda and Xn.is_contiguous() and Un.is_contiguous() and Xn.shape == (H, self.state_dim, B)
        assert Un.shape[1] == _lib.NUM_CONTROLS and Xn.dtype == torch.float32 and Un.dtype == torch.float32
        if out is None:
            out = torch.empty((H, self.state_dim, B), device=Xn.device, dtype=torch.float32)
        _lib.check(lib.ac_shoot_derivative_f32(self.system._handle, Xn.data_ptr(), Un.data_ptr(), B, H, out.data_ptr(),
                                               self.system._stream()), "ac_shoot_derivative_f32")
        return out

    def derivative_sens(self, Xn, Un, out=None):
        """f, Fx = df/dx, Fu = df/du at the H (state, control) pairs Xn (H, 13, B) [a view such as X[1:] is used in
        place], Un (H, 7, B): (H,13,B), (H,13,13,B), (H,13,7,B)."""
        torch = _torch()
        lib = self.system._sync()
        H, B = Un.shape[0], Un.shape[2]
        assert Xn.is_cuda and Xn.is_contiguous() and Un.is_contiguous() and Xn.shape == (H, self.state_dim, B)
        assert Un.shape[1] == _lib.NUM_CONTROLS and Xn.dtype == torch.float32 and Un.dtype == torch.float32
        ns, nc = self.state_dim, _lib.NUM_CONTROLS
        if out is None:
            out = (torch.empty((H, ns, B), device=Xn.device), torch.empty((H, ns, ns, B), device=Xn.device),
                   torch.empty((H, ns, nc, B), device=Xn.device))
        f, Fx, Fu = out
        _lib.check(lib.ac_shoot_derivative_sens_f32(self.system._handle, Xn.data_ptr(), Un.data_ptr(), B, H, f.data_ptr(),
                                                    Fx.data_ptr(), Fu.data_ptr(), self.system._stream()),
                   "ac_shoot_derivative_sens_f32")
        return f, Fx, Fu

    def linearise_implicit(self, X, U, dt=None):
        """Jacobian blocks of the implicit defect rows r_k = x_{k+1} - x_k - dt_k f(x_{k+1}, u_k):
            d r_k / d x_k = -I,   d r_k / d x_{k+1} = I - dt_k Fx,   d r_k / d u_k = -dt_k Fu,   d r_k / d dt_k = -f
        Returns (r, Jnext (N,13,13,B), Ju (N,13,7,B), jdt (N,13,B)), all four written by ONE launch of the derivative-
        sensitivity kernel on the next nodes (ac_shoot_implicit_rows_f32)."""
        torch = _torch()
        lib = self.system._sync()
        H, B, dts, dtp, keep = self._shoot_args(X, U, dt)
        assert X.shape[0] == H + 1
        ns, nc = self.state_dim, _lib.NUM_CONTROLS
        r = torch.empty((H, ns, B), device=X.device)
        Jn = torch.empty((H, ns, ns, B), device=X.device)
        Ju = torch.empty((H, ns, nc, B), device=X.device)
        jdt = torch.empty((H, ns, B), device=X.device)
        _lib.check(lib.ac_shoot_implicit_rows_f32(self.system._handle, X.data_ptr(), U.data_ptr(), dts, dtp, B, H, r.data_ptr(),
                                                  Jn.data_ptr(), Ju.data_ptr(), jdt.data_ptr(), self.system._stream()),
                   "ac_shoot_implicit_rows_f32")
        del keep
        return r, Jn, Ju, jdt

    def quaternion_rows(self, Xn, Un=None, mode: Optional[str] = None):
        """Quaternion rows of ControlProblem.state_constraint on the nodes Xn (H, 13, B) (the NEXT nodes x_{k+1}):
            'constraint'  q.q - 1                                          (control/base.py:285-286)
            'baumgarte'   2 a phi_dot + b^2 phi, phi = q.q - 1, phi_dot = 2 q.q_dot(x_{k+1}, u_{k+1}), a = b = 2   (:288-304)
        Returns (row (H,B), Jx (H,13,B), Ju (H,7,B)).  'baumgarte' needs the controls Un (H, 7, B) paired with Xn."""
        torch = _torch()
        lib = self.system._sync()
        mode = mode or self.opts.get("quaternion", None)
        if mode not in ("constraint", "baumgarte"):
            raise ValueError("quaternion rows exist for opts['quaternion'] in ('constraint', 'baumgarte')")
        H, B = Xn.shape[0], Xn.shape[2]
        assert Xn.is_cuda and Xn.is_contiguous() and Xn.dtype == torch.float32
        row = torch.empty((H, B), device=Xn.device)
        Jx = torch.empty((H, 13, B), device=Xn.device)
        Ju = torch.empty((H, 7, B), device=Xn.device)
        keep = None
        ptrs = [C.c_void_p(0)] * 3
        if mode == "baumgarte":
            if Un is None:
                raise ValueError("'baumgarte' differentiates q_dot = f(x_{k+1}, u_{k+1}): pass the controls Un")
            keep = self.derivative_sens(Xn, Un)
            ptrs = [C.c_void_p(t.data_ptr()) for t in keep]
        _lib.check(lib.ac_quat_rows_f32(self.system._handle, 1 if mode == "baumgarte" else 0, Xn.data_ptr(), *ptrs, B, H,
                                        row.data_ptr(), Jx.data_ptr(), Ju.data_ptr(), self.system._stream()),
                   "ac_quat_rows_f32")
        del keep
        return row, Jx, Ju

    def envelope(self, X, want_jacobian: bool = True):
        """Envelope rows of every node of X (H', 13, B) (control/aircraft.py:44-59): rows (H', 4, B) =
        (|v_rel|^2, beta, alpha, z) and Jx (H', 4, 13, B)."""
        torch = _torch()
        lib = self.system._sync()
        Hn, B = X.shape[0], X.shape[2]
        assert X.is_cuda and X.is_contiguous() and X.dtype == torch.float32 and X.shape[1] == self.state_dim
        rows = torch.empty((Hn, 4, B), device=X.device)
        Jx = torch.empty((Hn, 4, 13, B), device=X.device) if want_jacobian else None
        _lib.check(lib.ac_shoot_envelope_f32(self.system._handle, X.data_ptr(), B, Hn, rows.data_ptr(),
                                             Jx.data_ptr() if Jx is not None else None, self.system._stream()),
                   "ac_shoot_envelope_f32")
        return rows, Jx

    def hessian(self, X, U, Lam, dt=None, out=None):
        """Second-order blocks of every node of every instance: out (N, 21, 21, B) = sum_i Lam[k, i, b] d2F_i/dz dz at
        (x_k, u_k, dt_k), z = (x, u, dt) — the defect rows' part of the Lagrangian Hessian (`nlp_hess_l`).
        Lam (N, 13, B): multipliers of the defect rows (or the costate of a DDP sweep)."""
        torch = _torch()
        lib = self.system._sync()
        H, B, dts, dtp, keep = self._shoot_args(X, U, dt)
        assert Lam.shape == (H, 13, B) and Lam.is_contiguous() and Lam.dtype == torch.float32
        if out is None:
            out = torch.empty((H, 21, 21, B), device=X.device, dtype=torch.float32)
        self.system._reserve_hess(H * B)
        _lib.check(lib.ac_shoot_hess_f32(self.system._handle, X.data_ptr(), U.data_ptr(), dts, dtp, Lam.data_ptr(), B, H,
                                         out.data_ptr(), self.system._stream()), "ac_shoot_hess_f32")
        del keep
        return out

    def linearise(self, X, U, dt=None, want_c=True, out=None):
        """(F, A, B, c) per node: F (N,13,B), A = dF/dx (N,13,13,B), B = dF/du (N,13,7,B), c = dF/ddt (N,13,B).
        The defect Jacobian rows are [-A_k, -B_k, I] (and -c_k * d(dt_k)/d(progress_k) in progress time)."""
        torch = _torch()
        lib = self.system._sync()
        H, B, dts, dtp, keep = self._shoot_args(X, U, dt)
        ns, nc = self.state_dim, _lib.NUM_CONTROLS
        if out is None:
            F = torch.empty((H, ns, B), device=X.device, dtype=torch.float32)
            A = torch.empty((H, ns, ns, B), device=X.device, dtype=torch.float32)
            Bm = torch.empty((H, ns, nc, B), device=X.device, dtype=torch.float32)
            c = torch.empty((H, ns, B), device=X.device, dtype=torch.float32) if want_c else None
        else:
            F, A, Bm, c = out
        _lib.check(lib.ac_shoot_sens_f32(self.system._handle, X.data_ptr(), U.data_ptr(), dts, dtp, B, H,
                                         F.data_ptr(), A.data_ptr(), Bm.data_ptr(),
                                         c.data_ptr() if c is not None else None, self.system._stream()),
                   "ac_shoot_sens_f32")
        del keep
        return F, A, Bm, c

"""Batched iLQR / Gauss-Newton sweep on top of the multiple-shooting linearisation.

The reference formulates its MPC as an NLP and hands it to IPOPT, one instance at a time
(src/aircraft/control/base.py:455-477).  This module is the build-side solver that turns the hot-path kernels
(rollout, step sensitivities) into B simultaneous MPC solves — random restarts / independent instances — playing the
`loss` and control-limit roles of the reference's controllers (control/base.py:323-337, control/aircraft.py:29-41,
main/control/control.py:35-70: goal term on the final position, actuation penalty, surface limits).

One iteration = linearise (ac_shoot_sens_f32) -> backward Riccati pass (ac_ilqr_backward_f32) -> closed-loop rollouts for
every line-search step alpha in ONE launch (ac_rollout_policy_f32) -> cost (ac_ilqr_cost_f32) -> per-instance argmin.
Everything stays on the device; torch supplies the buffers only (the acceptance is a HIP kernel too: ac_ilqr_accept_f32).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

from .. import _lib
from .base import MultipleShooting


def _torch():
    import torch

    return torch


@dataclass
class QuadraticCost:
    """J = sum_k 1/2 (x_k - x_ref)' diag(q) (x_k - x_ref) + 1/2 u_k' diag(r) u_k + 1/2 (x_N - x_goal)' diag(qf) (x_N - x_goal)."""
    q: Sequence[float] = field(default_factory=lambda: [0.0] * 13)
    qf: Sequence[float] = field(default_factory=lambda: [0.0] * 13)
    r: Sequence[float] = field(default_factory=lambda: [1e-2] * 7)
    x_ref: Sequence[float] = field(default_factory=lambda: [0.0] * 13)
    x_goal: Sequence[float] = field(default_factory=lambda: [0.0] * 13)
    # control box: aileron/elevator/rudder +-5 deg (control/aircraft.py:26), thrust disabled, flaps in [0, 1]
    u_min: Sequence[float] = field(default_factory=lambda: [-5, -5, -5, 0, 0, 0, 0])
    u_max: Sequence[float] = field(default_factory=lambda: [5, 5, 5, 0, 0, 0, 1])
    reg: float = 1e-3
    u_lin: Sequence[float] = field(default_factory=lambda: [0.0] * 7)   # linear control cost (the time term)
    dt_row: int = 0                                                      # > 0: control row carrying dt_k (ILQR(time="variable"))

    @staticmethod
    def goal(goal_xy, w_goal=1000.0, w_height=1.0, height=None, w_lateral_speed=1000.0, r=1e-2, reg=1e-3):
        """The shape of Controller.loss (main/control/control.py:35-70): reach goal (x, y) at the final node, keep the
        final height, damp final lateral/vertical speed, penalise actuation."""
        c = QuadraticCost(r=[r] * 7, reg=reg)
        qf = [0.0] * 13; xg = [0.0] * 13
        qf[0] = qf[1] = 2.0 * w_goal; xg[0], xg[1] = float(goal_xy[0]), float(goal_xy[1])
        if height is not None:
            qf[2] = 2.0 * w_height; xg[2] = float(height)
        qf[4] = qf[5] = 2.0 * w_lateral_speed
        c.qf, c.x_goal = qf, xg
        return c

    @staticmethod
    def cruise(w_lateral_speed=0.5, w_attitude=5.0, w_rate=0.2, r=0.5, reg=1.0):
        """A regulator without a goal point, for closed loops that run for minutes: hold the heading north (v_east -> 0,
        q -> identity, i.e. wings level, nose on the horizon, yaw 0), damp the body rates, penalise actuation; height and
        along-track position are free (a glider sinks)."""
        c = QuadraticCost(r=[r] * 7, reg=reg)
        q = [0.0] * 13
        q[4] = w_lateral_speed
        q[6] = q[7] = q[8] = w_attitude   # vector part of q (xyzw) -> 0
        q[10] = q[11] = q[12] = w_rate
        xr = [0.0] * 13
        xr[9] = 1.0
        c.q, c.qf, c.x_ref, c.x_goal = q, list(q), xr, list(xr)
        return c

    def struct(self) -> "_lib.IlqrCost":
        s = _lib.IlqrCost()
        for name, n in (("q", 13), ("qf", 13), ("r", 7), ("x_ref", 13), ("x_goal", 13), ("u_min", 7), ("u_max", 7), ("u_lin", 7)):
            v = [float(x) for x in getattr(self, name)]
            assert len(v) == n, name
            getattr(s, name)[:] = v
        s.reg = float(self.reg)
        s.dt_row = int(self.dt_row)
        return s


class ILQR(MultipleShooting):
    def __init__(self, *, system, dt: float = 0.01, num_nodes: int, cost: QuadraticCost, opts: Optional[dict] = None,
                 alphas: Sequence[float] = (1.0, 0.5, 0.25, 0.1, 0.03), hessian: str = "gauss-newton",
                 envelope_weight: float = 0.0, envelope_bounds=None, envelope: str = "penalty",
                 time: str = "fixed", dt_bounds=(0.005, 0.02), w_time: float = 0.0, r_time: float = 0.0):
        """hessian: 'gauss-newton' (first-order dynamics in the backward pass: iLQR) or 'exact' (adds the second-order
        terms  sum_i lambda_i d2F_i/dz dz  of every node — what IPOPT gets from `nlp_hess_l`).
        envelope_weight > 0: the flight envelope of AircraftControl.state_constraint (control/aircraft.py:44-59:
        20^2 <= |v_rel|^2 <= 100^2, |beta| <= 10 deg, |alpha| <= 20 deg, z < 0) as a soft constraint — the squared violation
        of its four rows, times the weight, at every node, with its Gauss-Newton model in the backward pass (IPOPT
        enforces the rows as hard constraints; the control box stays a hard clip).  envelope_bounds: ((lo, hi),) * 4,
        default `system.ENVELOPE_BOUNDS`.
        envelope: 'penalty' (the soft form above) or 'al' — the HARD treatment: augmented-Lagrangian multipliers per node, row
        and instance (ac_envelope_al_*), updated by `update_multipliers()` between sweeps (`solve(al_every=)`); the penalty
        alone leaves a violation of about multiplier / (2 weight) at an active bound, the multipliers remove it."""
        super().__init__(system=system, dt=dt, num_nodes=num_nodes, opts=opts or {"quaternion": "integration"})
        assert 1 <= len(alphas) <= 8 and hessian in ("gauss-newton", "exact") and envelope in ("penalty", "al")
        self.envelope_mode = envelope
        # time as a decision variable per node (the reference's opts['time'] in ('progress', 'variable'),
        # control/base.py:276, 339-385; its loss adds the total time, main/control/control.py:44, 66-67): a control row the
        # force model ignores carries dt_k, boxed by dt_bounds, costed by w_time * dt_k (+ 1/2 r_time dt_k^2); the column of B
        # for it is c = dF/d(dt) from the sensitivity kernels.
        assert time in ("fixed", "variable")
        self.time_row = 0
        if time == "variable":
            assert hessian == "gauss-newton", "the exact-Hessian sweep keeps the fixed step"
            import copy

            self.time_row = 4 if getattr(system, "num_controls", 7) == 4 else 3
            cost = copy.deepcopy(cost)
            r = self.time_row
            cost.u_min, cost.u_max, cost.u_lin, cost.r = list(cost.u_min), list(cost.u_max), list(cost.u_lin), list(cost.r)
            cost.u_min[r], cost.u_max[r] = float(dt_bounds[0]), float(dt_bounds[1])
            cost.u_lin[r], cost.r[r], cost.dt_row = float(w_time), float(r_time), r
            self.cost = cost
        self.cost = cost
        self.alphas = [float(a) for a in alphas]
        self.hessian_mode = hessian
        self.envelope_weight = float(envelope_weight)
        self.envelope_bounds = envelope_bounds
        self._ws = None

    # ---- device workspace (allocated once per (B, H)) ------------------------------------------------
    def _workspace(self, B, dev):
        torch = _torch()
        H, na = self.num_nodes, len(self.alphas)
        key = (B, H, na, str(dev))
        if self._ws is None or self._ws["key"] != key:
            f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)  # noqa: E731
            self._ws = dict(key=key, F=f(H, 13, B), A=f(H, 13, 13, B), Bm=f(H, 13, 7, B), K=f(H, 7, 13, B), kff=f(H, 7, B),
                            dV=f(2, B), Xc=f(H + 1, 13, na * B), Uc=f(H, 7, na * B), Jc=f(na * B), J0=f(B), Ja=f(B),
                            improved=torch.empty((B,), device=dev, dtype=torch.bool))
            if self.time_row > 0:
                self._ws.update(dt=f(H, B), c=f(H, 13, B))
            if self.hessian_mode == "exact":
                self._ws.update(Lam=f(H, 13, B), Hz=f(H, 21, 21, B))
                self.system._sync()
                self.system._reserve_hess(H * B)  # host-side, once: the solve itself stays allocation-free (capturable)
            if self.envelope_weight > 0:
                # constant-cost rows as per-node arrays (the penalty gradient is added to glin every iteration)
                c = self.cost
                nq = torch.tensor([list(c.q)] * H + [list(c.qf)], device=dev, dtype=torch.float32)        # (H+1, 13)
                nx = torch.tensor([list(c.x_ref)] * H + [list(c.x_goal)], device=dev, dtype=torch.float32)
                self._ws.update(env_q=nq[:, :, None].expand(H + 1, 13, B).contiguous(),
                                env_xref=nx[:, :, None].expand(H + 1, 13, B).contiguous(), env_glin=f(H + 1, 13, B))
                if "Hz" not in self._ws:
                    self._ws["Hz"] = f(H, 21, 21, B)
                if self.envelope_mode == "al":
                    self._ws.update(lam=torch.zeros((H + 1, 8, B), device=dev, dtype=torch.float32), viol=f(B))
        return self._ws

    def _envelope_struct(self):
        b = self.envelope_bounds or self.system.ENVELOPE_BOUNDS
        big = 3.0e38
        clamp = lambda v: max(-big, min(big, float(v)))  # noqa: E731  (inf does not survive some float paths: use +-FLT_MAX)
        p = _lib.EnvelopePenalty()
        p.lo[:] = [clamp(r[0]) for r in b]
        p.hi[:] = [clamp(r[1]) for r in b]
        p.weight = self.envelope_weight
        return p

    def _lam(self):
        """The multipliers of the augmented-Lagrangian envelope (H+1, 8, B), or None for the plain penalty."""
        return self._ws.get("lam") if (self._ws is not None and self.envelope_mode == "al") else None

    def envelope_cost(self, X, cost_inout):
        """cost_inout[b] += the envelope term of every node of X (in place, on the device): the squared violation times the
        weight, or its augmented-Lagrangian form with the current multipliers."""
        lib = self.system._sync()
        H, B = X.shape[0] - 1, X.shape[2]
        p = self._envelope_struct()
        lam = self._lam()
        _lib.check(lib.ac_envelope_al_cost_f32(self.system._handle, C.byref(p), lam.data_ptr() if lam is not None else None,
                                               lam.shape[2] if lam is not None else 1, X.data_ptr(), B, H,
                                               cost_inout.data_ptr(), self.system._stream()), "ac_envelope_al_cost_f32")
        return cost_inout

    def update_multipliers(self, X):
        """First-order multiplier update of the augmented-Lagrangian envelope at the iterate X (in place).  Returns each
        instance's largest bound excess before the update (B,), relative to the row's range."""
        lib = self.system._sync()
        assert self.envelope_mode == "al" and self.envelope_weight > 0
        H, B = X.shape[0] - 1, X.shape[2]
        ws = self._workspace(B, X.device)
        p = self._envelope_struct()
        ws["viol"].zero_()
        _lib.check(lib.ac_envelope_al_update_f32(self.system._handle, C.byref(p), X.data_ptr(), B, H, ws["lam"].data_ptr(),
                                                 ws["viol"].data_ptr(), self.system._stream()), "ac_envelope_al_update_f32")
        return ws["viol"]

    def _envelope_model(self, X, glin=None, Hz=None):
        """Adds the penalty's gradient to glin (N+1, 13, B) and / or its Gauss-Newton curvature to the (x, x) block of
        Hz (N, 21, 21, B), in place."""
        lib = self.system._sync()
        H, B = X.shape[0] - 1, X.shape[2]
        p = self._envelope_struct()
        lam = self._lam()
        _lib.check(lib.ac_envelope_al_model_f32(self.system._handle, C.byref(p), lam.data_ptr() if lam is not None else None,
                                                X.data_ptr(), B, H, glin.data_ptr() if glin is not None else None,
                                                Hz.data_ptr() if Hz is not None else None, self.system._stream()),
                   "ac_envelope_al_model_f32")

    def _cstruct(self):
        return C.byref(self.cost.struct())

    def trajectory_cost(self, X, U, out=None):
        torch = _torch()
        lib = self.system._sync()
        H, B = U.shape[0], U.shape[2]
        if out is None:
            out = torch.empty((B,), device=X.device, dtype=torch.float32)
        _lib.check(lib.ac_ilqr_cost_f32(self.system._handle, self._cstruct(), X.data_ptr(), U.data_ptr(), B, H,
                                        out.data_ptr(), self.system._stream()), "ac_ilqr_cost_f32")
        return out

    def _node_cost(self, X, U):
        """Hook: per-node state-cost arrays (node_q, node_xref, node_glin), each (N+1, 13, B), or None for the constant
        quadratic cost of `self.cost` (MHTT overrides this)."""
        return None

    @staticmethod
    def _ptrs(node):
        return [C.c_void_p(t.data_ptr()) for t in node] if node is not None else [C.c_void_p(0)] * 3

    def costate(self, X, A, node=None, out=None):
        """Multipliers of the defect rows at the current iterate, (N, 13, B): Lam[N-1] = grad l_N(x_N),
        Lam[k-1] = grad l_k(x_k) + A_k' Lam[k]."""
        torch = _torch()
        lib = self.system._sync()
        H, B = A.shape[0], A.shape[3]
        if out is None:
            out = torch.empty((H, 13, B), device=X.device, dtype=torch.float32)
        _lib.check(lib.ac_ilqr_costate_f32(self.system._handle, self._cstruct(), *self._ptrs(node), X.data_ptr(),
                                           A.data_ptr(), B, H, out.data_ptr(), self.system._stream()),
                   "ac_ilqr_costate_f32")
        return out

    _goal_model = None  # GoalAcquisition sets it (see iterate)

    def backward(self, X, U, A, Bm, out=None, Hz=None, node="auto", uglin=None):
        """Riccati pass -> K (N, 7, 13, B), kff (N, 7, B), dV (2, B).  Hz (N, 21, 21, B): optional second-order
        dynamics blocks from `hessian()` (exact-Hessian / Newton step).  uglin (N, 7, B): optional per-node control
        gradient (needs node arrays and Hz: ac_ilqr_backward_goal_f32)."""
        torch = _torch()
        lib = self.system._sync()
        H, B = U.shape[0], U.shape[2]
        if isinstance(node, str):
            node = self._node_cost(X, U)
        if out is None:
            out = (torch.empty((H, 7, 13, B), device=X.device), torch.empty((H, 7, B), device=X.device),
                   torch.empty((2, B), device=X.device))
        K, kff, dV = out
        if uglin is not None:
            assert node is not None and Hz is not None
            _lib.check(lib.ac_ilqr_backward_goal_f32(self.system._handle, self._cstruct(), *self._ptrs(node),
                                                     C.c_void_p(uglin.data_ptr()), C.c_void_p(Hz.data_ptr()), X.data_ptr(),
                                                     U.data_ptr(), A.data_ptr(), Bm.data_ptr(), B, H, K.data_ptr(),
                                                     kff.data_ptr(), dV.data_ptr(), self.system._stream()),
                       "ac_ilqr_backward_goal_f32")
            return K, kff, dV
        _lib.check(lib.ac_ilqr_backward_newton_f32(self.system._handle, self._cstruct(), *self._ptrs(node),
                                                   C.c_void_p(Hz.data_ptr() if Hz is not None else 0), X.data_ptr(),
                                                   U.data_ptr(), A.data_ptr(), Bm.data_ptr(), B, H, K.data_ptr(),
                                                   kff.data_ptr(), dV.data_ptr(), self.system._stream()),
                   "ac_ilqr_backward_newton_f32")
        return K, kff, dV

    def forward(self, x0, Xnom, U, K, kff, alphas=None, out=None):
        """Closed-loop rollouts for every alpha: Xc (H+1, 13, n_alpha*B), Uc (H, 7, n_alpha*B); column a*B + b."""
        torch = _torch()
        lib = self.system._sync()
        alphas = self.alphas if alphas is None else [float(a) for a in alphas]
        na = len(alphas)
        H, B = U.shape[0], U.shape[2]
        if out is None:
            out = (torch.empty((H + 1, 13, na * B), device=U.device), torch.empty((H, 7, na * B), device=U.device))
        Xc, Uc = out
        arr = (C.c_float * na)(*alphas)
        _lib.check(lib.ac_rollout_policy_f32(self.system._handle, self._cstruct(), x0.data_ptr(), Xnom.data_ptr(),
                                             U.data_ptr(), K.data_ptr(), kff.data_ptr(), arr, na, C.c_float(self.dt),
                                             B, H, Xc.data_ptr(), Uc.data_ptr(), self.system._stream()),
                   "ac_rollout_policy_f32")
        return Xc, Uc

    def iterate(self, x0, X, U):
        """One iLQR iteration in place on (X, U).  Returns (cost (B,), improved (B,) bool)."""
        torch = _torch()
        B = U.shape[2]
        ws = self._workspace(B, U.device)
        na = len(self.alphas)
        if self.time_row > 0:
            # linearise at the nodes' own steps dt_k = U[k, time_row]; its derivative c = dF/d(dt) IS the column of B for that row
            ws["dt"].copy_(U[:, self.time_row, :])
            self.linearise(X, U, dt=ws["dt"], want_c=True, out=(ws["F"], ws["A"], ws["Bm"], ws["c"]))
            ws["Bm"][:, :, self.time_row, :].copy_(ws["c"])
        else:
            self.linearise(X, U, want_c=False, out=(ws["F"], ws["A"], ws["Bm"], None))
        node = self._node_cost(X, U)  # None, or a subclass's per-node arrays (rewritten by it every iteration)
        Hz = None
        env = self.envelope_weight > 0
        goal = self._goal_model is not None  # GoalAcquisition: node arrays + control gradient + (u, u) curvature
        if self.hessian_mode != "exact" and (env or goal):
            Hz = ws["Hz"].zero_()
        # (first: it WRITES the node arrays the envelope below adds to; adds the rate curvature to Hz, returns the control gradient)
        uglin = self._goal_model(X, U, Hz) if goal else None
        if env:
            if node is None:  # the constant cost as per-node arrays, so that the penalty gradient has a place to go
                ws["env_glin"].zero_()
                node = (ws["env_q"], ws["env_xref"], ws["env_glin"])
            self._envelope_model(X, glin=node[2])  # before the costate: the multipliers see the penalty too
        if self.hessian_mode == "exact":
            self.costate(X, ws["A"], node, out=ws["Lam"])
            Hz = self.hessian(X, U, ws["Lam"], out=ws["Hz"])
        if env:
            self._envelope_model(X, Hz=Hz)
        self.backward(X, U, ws["A"], ws["Bm"], out=(ws["K"], ws["kff"], ws["dV"]), Hz=Hz, node=node, uglin=uglin)
        self.forward(x0, X, U, ws["K"], ws["kff"], out=(ws["Xc"], ws["Uc"]))
        self.trajectory_cost(ws["Xc"], ws["Uc"], out=ws["Jc"])
        self.trajectory_cost(X, U, out=ws["J0"])
        if env:
            self.envelope_cost(ws["Xc"], ws["Jc"])
            self.envelope_cost(X, ws["J0"])
        return self.accept(ws["Jc"], ws["J0"], ws["Xc"], ws["Uc"], X, U)

    def accept(self, Jc, J0, Xc, Uc, X, U):
        """Line-search acceptance in ONE launch (`ac_ilqr_accept_f32`): per instance the cheapest finite candidate replaces
        the iterate (X, U) in place if it beats J0.  Returns (accepted cost (B,), improved (B,) bool)."""
        torch = _torch()
        lib = self.system._sync()
        H, B = U.shape[0], U.shape[2]
        na = Jc.numel() // B
        ws = self._workspace(B, U.device)
        Ja, imp = ws["Ja"], ws["improved"]
        _lib.check(lib.ac_ilqr_accept_f32(self.system._handle, Jc.data_ptr(), J0.data_ptr(), Xc.data_ptr(), Uc.data_ptr(),
                                          na, B, H, X.data_ptr(), U.data_ptr(), Ja.data_ptr(), imp.data_ptr(),
                                          self.system._stream()), "ac_ilqr_accept_f32")
        return Ja, imp

    def solve(self, x0, U0, iters: int = 10, save_to: Optional[str] = None, save_instance: int = 0, al_every: int = 0):
        """Rollout from x0 with U0, then `iters` iLQR iterations.  Returns (X, U, cost history (iters+1, B)).
        `save_to` writes instance `save_instance` after every iteration in the reference's trajectory format
        (iteration_0 = the initial rollout), the role of the IPOPT callback in control/base.py:60-86.
        al_every > 0 (envelope = 'al'): a multiplier update after every al_every-th sweep."""
        torch = _torch()
        U = U0.clone()
        if self.time_row > 0:
            U[:, self.time_row, :] = self.dt  # every node starts at the nominal step (the rollout below uses it)
        X = self.rollout(x0, U)
        J = self.trajectory_cost(X, U).clone()
        if self.envelope_weight > 0:  # the same objective as every later entry (iterate() adds the penalty)
            self.envelope_cost(X, J)
        hist = [J]
        if save_to:
            self.save_progress(save_to, 0, X, U, save_instance, mode="w")
        for it in range(iters):
            J, _ = self.iterate(x0, X, U)
            hist.append(J.clone())
            if al_every and self.envelope_mode == "al" and (it + 1) % al_every == 0 and it + 1 < iters:
                self.update_multipliers(X)  # (the objective changes with the multipliers: the history is monotone between updates)
            if save_to:
                self.save_progress(save_to, it + 1, X, U, save_instance)
        return X, U, torch.stack(hist)


class GoalAcquisition(ILQR):
    """The reference's goal-acquisition problem (Controller, main/control/control.py:25-70) on the batched sweep: reach
    goal (x, y) at the final node while the loss of Controller.loss is minimised —

        1000 |p_xy(N) - goal|^2 + 100 sum l0_smooth(u_{k+1} - u_k, 1e-2) + (z_N - z_0)^2 - sum v_rel.v_rel / 100 / N
        + vel_param 1000 v_x(N) + 1000 (v_y(N)^2 + v_z(N)^2) [+ 10000 T],     subject to  v_x(N) < -2

    — for B instances at once (independent goals / initial states / random restarts).  The EXACT loss drives the line
    search and the history (ac_goal_cost_f32); the backward pass sees its convex quadratic model (ac_goal_model_f32:
    csrc/ac_goal.hpp states what is modelled how); the terminal inequality is an augmented-Lagrangian term with one
    multiplier per instance, updated by `update_goal_multiplier()` (`solve(al_every=)`).  time='variable' adds the
    reference's time term 10000 T through w_time (opts['time'] = 'progress' in the reference's driver, control.py:182)."""

    def __init__(self, *, system, goal, dt: float = 0.01, num_nodes: int = 400, vel_param: float = 1.0, w_goal: float = 1000.0,
                 w_rate: float = 100.0, eps_rate: float = 1e-2, w_height: float = 1.0, w_speed: float = 0.01,
                 w_final_velocity: float = 1000.0, vx_max: float = -2.0, w_al: float = 10.0, r: float = 0.0, reg: float = 1e-2,
                 time: str = "fixed", w_time: float = 10000.0, **kw):
        cost = QuadraticCost(q=[0.0] * 13, qf=[0.0] * 13, r=[float(r)] * 7, reg=float(reg))  # every state term lives in the node arrays
        super().__init__(system=system, dt=dt, num_nodes=num_nodes, cost=cost, time=time, w_time=w_time if time == "variable" else 0.0,
                         **kw)
        assert self.hessian_mode == "gauss-newton"
        self.goal = goal
        self.loss = _lib.GoalLoss()
        ls = self.loss
        ls.w_goal, ls.w_rate, ls.eps_rate, ls.w_height, ls.w_speed = w_goal, w_rate, eps_rate, w_height, w_speed
        ls.w_vx, ls.w_vyz, ls.vx_max, ls.w_al, ls.time_row = vel_param * w_final_velocity, w_final_velocity, vx_max, w_al, self.time_row
        self._gws = None

    def _goal_ws(self, B, dev):
        torch = _torch()
        H = self.num_nodes
        if self._gws is None or self._gws["key"] != (B, H, str(dev)):
            f = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)  # noqa: E731
            g = torch.as_tensor(self.goal, dtype=torch.float32, device=dev).reshape(2, -1)
            if g.shape[1] == 1:
                g = g.expand(2, B)
            assert g.shape == (2, B), "goal: (2,) for every instance or (2, B)"
            self._gws = dict(key=(B, H, str(dev)), goal=g.contiguous(), lam=f(B), viol=f(B), nq=f(H + 1, 13, B),
                             nx=f(H + 1, 13, B), ng=f(H + 1, 13, B), ug=f(H, 7, B))
            ws = self._workspace(B, dev)
            if "Hz" not in ws:
                ws["Hz"] = f(H, 21, 21, B)
        return self._gws

    # the node arrays are produced together with the control gradient (one launch) in _goal_model; _node_cost hands them over
    def _node_cost(self, X, U):
        g = self._goal_ws(U.shape[2], U.device)
        return (g["nq"], g["nx"], g["ng"])

    def _goal_model(self, X, U, Hz):
        lib = self.system._sync()
        H, B = U.shape[0], U.shape[2]
        g = self._goal_ws(B, U.device)
        _lib.check(lib.ac_goal_model_f32(self.system._handle, C.byref(self.loss), g["goal"].data_ptr(), g["lam"].data_ptr(),
                                         X.data_ptr(), U.data_ptr(), B, H, g["nq"].data_ptr(), g["nx"].data_ptr(),
                                         g["ng"].data_ptr(), g["ug"].data_ptr(), Hz.data_ptr(), self.system._stream()),
                   "ac_goal_model_f32")
        return g["ug"]

    def trajectory_cost(self, X, U, out=None):
        """The exact loss of every column of (X, U) (a candidate batch reads instance b % B's goal and multiplier)."""
        lib = self.system._sync()
        out = super().trajectory_cost(X, U, out=out)  # control terms of the cost struct (actuation, time)
        Bc, H = U.shape[2], U.shape[0]
        Bn = self._gws["key"][0] if self._gws is not None else Bc
        g = self._goal_ws(Bn, U.device)
        _lib.check(lib.ac_goal_cost_f32(self.system._handle, C.byref(self.loss), g["goal"].data_ptr(), g["lam"].data_ptr(), Bn,
                                        X.data_ptr(), U.data_ptr(), Bc, H, out.data_ptr(), self.system._stream()),
                   "ac_goal_cost_f32")
        return out

    def update_goal_multiplier(self, X):
        """First-order update of the terminal inequality's multipliers at the iterate; returns max(0, v_x(N) - vx_max) (B,)."""
        lib = self.system._sync()
        H, B = X.shape[0] - 1, X.shape[2]
        g = self._goal_ws(B, X.device)
        _lib.check(lib.ac_goal_multiplier_f32(self.system._handle, C.byref(self.loss), X.data_ptr(), B, H, g["lam"].data_ptr(),
                                              g["viol"].data_ptr(), self.system._stream()), "ac_goal_multiplier_f32")
        return g["viol"]

    def solve(self, x0, U0, iters: int = 10, al_every: int = 0, **kw):
        torch = _torch()
        self._goal_ws(U0.shape[2], U0.device)
        if not al_every:
            return super().solve(x0, U0, iters=iters, **kw)
        # multiplier updates between blocks of sweeps (the objective changes with them: monotone within a block)
        X, U, hist = super().solve(x0, U0, iters=min(al_every, iters), **kw)
        hists, done = [hist], min(al_every, iters)
        while done < iters:
            self.update_goal_multiplier(X)
            n = min(al_every, iters - done)
            X, U, h = super().solve(x0, U, iters=n, **kw)
            hists.append(h)
            done += n
        return X, U, torch.cat(hists)

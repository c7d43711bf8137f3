"""Receding-horizon closed loop — the loop of the reference's moving-horizon driver (main/mhe/mhtt.py:79-124) around
the batched solver:

    while not done:
        sol = solve()                                  # here: a few iLQR iterations on the device
        keep the first N - overlap nodes               # mhtt.py:86-88
        x0 = state[:, N - overlap]                     # mhtt.py:108 (last kept node)
        guess = initialise(x0)  (rollout, zero / shifted controls)   # control/moving_horizon.py:203-213
        re-parameterise, next solve                    # mhtt.py:110-114

for B independent instances at once.  One cycle is a fixed sequence of kernel launches on fixed buffers, so it can be
captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed — BASELINE configs[4].
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .. import _lib
from .ilqr import ILQR, QuadraticCost
from .track import Track


def _torch():
    import torch

    return torch


class quiet_capture:
    """torch.cuda.graph(g, stream=...) with Python's cyclic garbage collector run first and paused inside.  A collection
    that happens to start during the capture may finalise an `Aircraft` of some earlier computation; its close() frees
    device memory (ac_destroy: hipFree), which is not permitted while a stream of the process is capturing and invalidates
    the graph ("operation failed due to a previous error during capture").  torch >= 2.9 no longer collects on entry."""

    def __init__(self, graph, stream):
        self.graph, self.stream, self._ctx, self._was = graph, stream, None, False

    def __enter__(self):
        import gc

        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        self._ctx = _torch().cuda.graph(self.graph, stream=self.stream)
        try:
            return self._ctx.__enter__()
        except BaseException:
            if self._was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc

        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._was:
                gc.enable()


class RecedingHorizon:
    def __init__(self, solver: ILQR, overlap: int = 30, iterations: int = 2, warm_start: str = "shift"):
        """overlap: nodes of each solve that are discarded (mhtt.py:77); iterations: iLQR iterations per solve;
        warm_start: 'shift' re-uses the tail of the previous controls, 'zero' restarts from zero controls as the
        reference's MHTT.initialise does (control/moving_horizon.py:204)."""
        assert 0 <= overlap < solver.num_nodes and iterations >= 0
        if getattr(solver, "time_row", 0) != 0:
            # The loop re-rolls the trajectory at the fixed solver.dt and shifts / zeroes whole control columns; a solver
            # that carries dt_k in a control row would be linearised at dt_k = 0 after a 'zero' warm start and scored
            # against a rollout that is not its own (MHTT makes the same restriction: time == 'fixed').
            raise ValueError("RecedingHorizon needs a fixed-time solver: ILQR(time='fixed')")
        self.solver, self.overlap, self.iterations, self.warm_start = solver, overlap, iterations, warm_start
        self.keep = solver.num_nodes - overlap
        self._graph = None

    def allocate(self, x0, U0):
        torch = _torch()
        H, B = self.solver.num_nodes, x0.shape[1]
        self.x0 = x0.clone().contiguous()
        self.U = U0.clone().contiguous()
        self.X = torch.empty((H + 1, 13, B), device=x0.device, dtype=torch.float32)
        self._Xhead = torch.empty((self.overlap + 1, 13, B), device=x0.device, dtype=torch.float32)
        self.executed = torch.empty((self.keep, 13, B), device=x0.device, dtype=torch.float32)  # states 1..keep of the last solve
        self.cost = torch.empty((B,), device=x0.device, dtype=torch.float32)
        self.solver._workspace(B, x0.device)
        self.solver.rollout(self.x0, self.U, out=self.X)  # every cycle ends with X prepared for the next one
        return self

    def cycle(self):
        """One solve + shift on the allocated buffers (capturable: no allocation, no host sync)."""
        torch = _torch()
        s = self.solver
        J = None
        for _ in range(self.iterations):
            J, _ = s.iterate(self.x0, self.X, self.U)
        if J is None:  # iterations == 0: the loop alone (keep / shift / re-rollout), no solve
            J = s.trajectory_cost(self.X, self.U)
            if getattr(s, "envelope_weight", 0.0) > 0:  # the objective iterate() reports includes the penalty
                s.envelope_cost(self.X, J)
        self.cost.copy_(J)
        self.executed.copy_(self.X[1 : self.keep + 1])
        # advance: the state reached after the kept nodes becomes the next initial state
        self.x0.copy_(self.X[self.keep])
        if hasattr(s, "advance_progress"):  # MHTT: the progress reached at the last kept node (mhtt.py:105)
            s.advance_progress(self.X, self.keep)
        if self.warm_start == "shift":
            tail = self.U[self.keep:].clone()
            self.U[: self.overlap].copy_(tail)
            self.U[self.overlap:].copy_(tail[-1:].expand(self.keep, -1, -1))
            # the accepted X is the rollout of U, so the shifted window already has its first `overlap` nodes; only the
            # `keep` new ones at the end are integrated, from the old final state
            self._Xhead.copy_(self.X[self.keep:])
            self.X[: self.overlap + 1].copy_(self._Xhead)
            if self.keep > 0:
                s.rollout(self.X[self.overlap], self.U[self.overlap:], out=self.X[self.overlap:])
        else:
            self.U.zero_()
            s.rollout(self.x0, self.U, out=self.X)

    def capture(self):
        """Capture one cycle into a hipGraph; afterwards step() replays it."""
        torch = _torch()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x0, U, X = self.x0.clone(), self.U.clone(), self.X.clone()
            s0 = getattr(self.solver, "s0", None)  # MHTT carries the progress at node 0 across cycles
            s0_saved = s0.clone() if s0 is not None else None
            self.cycle()  # warm-up launch outside capture (lazy initialisation, LDS attribute calls)
            torch.cuda.synchronize()
            self.x0.copy_(x0); self.U.copy_(U); self.X.copy_(X)
            if s0 is not None:
                s0.copy_(s0_saved)
            g = torch.cuda.CUDAGraph()
            with quiet_capture(g, side):
                self.cycle()
            self.x0.copy_(x0); self.U.copy_(U); self.X.copy_(X)
            if s0 is not None:
                s0.copy_(s0_saved)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = g
        return self

    def step(self):
        if self._graph is not None:
            self._graph.replay()
        else:
            self.cycle()

    def run(self, cycles: int, record: bool = False):
        """`cycles` solves; returns the executed state history (cycles*keep + 1, 13, B) when record=True."""
        torch = _torch()
        hist = [self.x0.clone()[None]] if record else None
        for _ in range(cycles):
            x_before = self.x0.clone() if record else None
            self.step()
            if record:
                hist.append(self.executed.clone())  # the first `keep` steps of this solve were executed
                del x_before
        return torch.cat(hist) if record else None


@dataclass
class MHTTWeights:
    """Loss weights of the reference's MHTT.loss (control/moving_horizon.py:47-55)."""
    w_tracking: float = 10.0
    w_progress: float = 5.0
    w_progress_rate: float = 2.0
    w_backward: float = 50.0
    w_terminal_align: float = 20.0
    w_low_velocity: float = 10.0
    w_control: float = 100.0

    def struct(self) -> "_lib.MhttWeights":
        return _lib.MhttWeights(self.w_tracking, self.w_progress, self.w_progress_rate, self.w_backward,
                                self.w_terminal_align, self.w_low_velocity, self.w_control)


class MHTT(ILQR):
    """Moving-horizon track tracking for B instances — the track/progress part of the reference's `MHTT`
    (control/moving_horizon.py:33-247) on the batched solver:

        progress variable s_k per node, 0 <= s <= 1                                    :131-137
        s_{k+1} <= s_k + (v_k . t^)/L dt + 0.05 (p_k - track(s_k)) . t^ / L            :147-168
        loss = 10 tracking - 5 sum s - 2 sum s_dot + 50 backward + 10 slow + 20 |p_N - track(1)| + 100 |u|^2   :44-105
        initialise(x0, s0): rollout + velocity-projection progress guess               :203-239

    The NLP (IPOPT) is replaced by iLQR: progress follows its constraint at the bound (the loss rewards progress, so
    the bound is active), each iteration linearises the dynamics (ac_shoot_sens_f32), freezes the progress sequence
    to build a per-node diagonal-quadratic model of the loss (ac_track_progress_f32 -> ac_ilqr_backward_node_f32),
    and accepts line-search candidates on the TRUE loss with their own progress recursion (ac_mhtt_loss_f32).
    """

    def __init__(self, *, system, track: Track, dt: float, num_nodes: int, opts: Optional[dict] = None,
                 weights: Optional[MHTTWeights] = None, reg: float = 1.0,
                 alphas: Sequence[float] = (1.0, 0.5, 0.25, 0.1, 0.03), hessian: str = "gauss-newton", **kwargs):
        opts = opts if opts else {"time": "fixed", "quaternion": "integration", "integration": "explicit"}
        assert opts.get("time", "fixed") == "fixed", "can only run mhtt with fixed time"  # moving_horizon.py:35
        self.weights = weights or MHTTWeights()
        cost = QuadraticCost(r=[2.0 * self.weights.w_control] * 7, reg=reg)  # w_control * |u|^2 = 1/2 u' (2 w) u
        super().__init__(system=system, dt=dt, num_nodes=num_nodes, cost=cost, opts=opts, alphas=alphas, hessian=hessian)
        self.track = track
        self.track_length = track.length()
        assert self.track_length > 1e-6  # moving_horizon.py:155
        track.install(system)
        self.s0 = None
        self._mws = None

    # ---- workspace --------------------------------------------------------------------------------------
    def _mhtt_workspace(self, B, dev):
        torch = _torch()
        H, na = self.num_nodes, len(self.alphas)
        key = (B, H, na, str(dev))
        if self._mws is None or self._mws["key"] != key:
            f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)  # noqa: E731
            self._mws = dict(key=key, S=f(H + 1, B), Sc=f(H + 1, na * B), s0c=f(na * B), nq=f(H + 1, 13, B),
                             nx=f(H + 1, 13, B), ng=f(H + 1, 13, B))
        return self._mws

    def set_progress(self, s0):
        """Progress of every instance at node 0 (the `progress_guess_parameter[0]` pin, moving_horizon.py:139)."""
        torch = _torch()
        dev = self.system._device_obj()
        s0 = torch.as_tensor(np.asarray(s0, dtype=np.float32) if not isinstance(s0, torch.Tensor) else s0)
        s0 = s0.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if self.s0 is not None and self.s0.shape == s0.shape:
            self.s0.copy_(s0)  # keep the buffer: a captured graph reads it
        else:
            self.s0 = s0.clone()
        return self.s0

    # ---- progress recursion, loss ------------------------------------------------------------------------
    def progress(self, X, s0=None, mode: int = 1, want_terms: bool = False, model=None, out=None):
        """S (N+1, B) along X (N+1, 13, B) from s0 (B,).  mode 0 = initial guess, 1 = constraint at its bound.
        want_terms adds (s_dot (N, B), tracking_error (N, B)); model = (node_q, node_xref, node_glin) to fill."""
        torch = _torch()
        lib = self.system._sync()
        s0 = self.s0 if s0 is None else s0
        H, B = X.shape[0] - 1, X.shape[2]
        assert s0 is not None and s0.numel() == B, "set_progress() first: one s0 per instance"
        S = out if out is not None else torch.empty((H + 1, B), device=X.device, dtype=torch.float32)
        sd = torch.empty((H, B), device=X.device, dtype=torch.float32) if want_terms else None
        te = torch.empty((H, B), device=X.device, dtype=torch.float32) if want_terms else None
        w = self.weights.struct()
        ptr = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)  # noqa: E731
        nq, nx, ng = model if model is not None else (None, None, None)
        _lib.check(lib.ac_track_progress_f32(self.system._handle, C.byref(w), X.data_ptr(), s0.data_ptr(),
                                             C.c_float(self.dt), B, H, int(mode), S.data_ptr(), ptr(sd), ptr(te),
                                             ptr(nq), ptr(nx), ptr(ng), self.system._stream()),
                   "ac_track_progress_f32")
        return (S, sd, te) if want_terms else S

    def loss(self, X, U, S, out=None):
        """MHTT.loss (moving_horizon.py:44-105) of every instance -> (B,)."""
        torch = _torch()
        lib = self.system._sync()
        H, B = U.shape[0], U.shape[2]
        if out is None:
            out = torch.empty((B,), device=X.device, dtype=torch.float32)
        w = self.weights.struct()
        _lib.check(lib.ac_mhtt_loss_f32(self.system._handle, C.byref(w), X.data_ptr(), U.data_ptr(), S.data_ptr(), B, H,
                                        out.data_ptr(), self.system._stream()), "ac_mhtt_loss_f32")
        return out

    def track_eval(self, s):
        """Device evaluation of the track: s (n,) -> pos (3, n), tangent (3, n)."""
        torch = _torch()
        lib = self.system._sync()
        s = s.contiguous()
        n = s.numel()
        pos = torch.empty((3, n), device=s.device, dtype=torch.float32)
        tan = torch.empty((3, n), device=s.device, dtype=torch.float32)
        _lib.check(lib.ac_track_eval_f32(self.system._handle, s.data_ptr(), n, pos.data_ptr(), tan.data_ptr(),
                                         self.system._stream()), "ac_track_eval_f32")
        return pos, tan

    # ---- the reference's initial guess ----------------------------------------------------------------------
    def initialise(self, initial_state, current_progress, controls=None):
        """One instance, reference-shaped: array (13 + 7 + 1, N+1) = rollout of zero controls from `initial_state`
        with the velocity-projection progress guess in the last row (moving_horizon.py:203-239)."""
        torch = _torch()
        N = self.num_nodes
        guess = np.zeros((self.state_dim + self.control_dim + 1, N + 1))
        if controls is not None:
            guess[self.state_dim:-1, :] = np.asarray(controls, dtype=np.float64).reshape(self.control_dim, N + 1)
        x0 = np.asarray(initial_state, dtype=np.float64).reshape(self.state_dim)
        dev = self.system._device_obj()
        U = torch.as_tensor(np.ascontiguousarray(guess[self.state_dim:-1, :N].T[:, :, None]), dtype=torch.float32, device=dev)
        X = self.rollout(torch.as_tensor(x0[:, None], dtype=torch.float32, device=dev), U)
        s0 = torch.full((1,), float(current_progress), dtype=torch.float32, device=dev)
        S = self.progress(X, s0, mode=0)
        guess[: self.state_dim, :] = X[:, :, 0].T.cpu().numpy()
        guess[-1, :] = S[:, 0].cpu().numpy()
        return guess

    # ---- ILQR hooks: true loss for acceptance, frozen-progress model for the backward pass ------------------------
    def trajectory_cost(self, X, U, out=None):
        B = self.s0.numel()
        ws = self._mhtt_workspace(B, U.device)
        if X.shape[2] == B:
            S = self.progress(X, self.s0, mode=1, out=ws["S"])
        else:  # line-search candidates: column a*B + b starts from s0[b]
            na = X.shape[2] // B
            ws["s0c"].view(na, B).copy_(self.s0[None, :].expand(na, B))
            S = self.progress(X, ws["s0c"], mode=1, out=ws["Sc"])
        return self.loss(X, U, S, out=out)

    def _node_cost(self, X, U):
        """The frozen-progress quadratic model of the MHTT loss around (X, U), written by the progress kernel."""
        ws = self._mhtt_workspace(U.shape[2], U.device)
        self.progress(X, self.s0, mode=1, model=(ws["nq"], ws["nx"], ws["ng"]), out=ws["S"])
        return ws["nq"], ws["nx"], ws["ng"]

    def advance_progress(self, X, keep: int):
        """Receding-horizon shift: s0 <- progress reached at node `keep` of the accepted trajectory (mhtt.py:87, 105)."""
        ws = self._mhtt_workspace(self.s0.numel(), X.device)
        S = self.progress(X, self.s0, mode=1, out=ws["S"])
        self.s0.copy_(S[keep])

    def solve(self, x0, s0, U0, iters: int = 10, **kw):
        """(X, U, S, loss history (iters+1, B)) from initial states x0 (13, B), progress s0 (B,), controls U0 (N, 7, B)."""
        self.set_progress(s0)
        X, U, hist = super().solve(x0, U0, iters=iters, **kw)
        return X, U, self.progress(X, self.s0, mode=1).clone(), hist

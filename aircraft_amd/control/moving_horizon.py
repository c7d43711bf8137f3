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

from .ilqr import ILQR


def _torch():
    import torch

    return torch


class RecedingHorizon:
    def __init__(self, solver: ILQR, overlap: int = 30, iterations: int = 2, warm_start: str = "shift"):
        """overlap: nodes of each solve that are discarded (mhtt.py:77); iterations: iLQR iterations per solve;
        warm_start: 'shift' re-uses the tail of the previous controls, 'zero' restarts from zero controls as the
        reference's MHTT.initialise does (control/moving_horizon.py:204)."""
        assert 0 <= overlap < solver.num_nodes
        self.solver, self.overlap, self.iterations, self.warm_start = solver, overlap, iterations, warm_start
        self.keep = solver.num_nodes - overlap
        self._graph = None

    def allocate(self, x0, U0):
        torch = _torch()
        H, B = self.solver.num_nodes, x0.shape[1]
        self.x0 = x0.clone().contiguous()
        self.U = U0.clone().contiguous()
        self.X = torch.empty((H + 1, 13, B), device=x0.device, dtype=torch.float32)
        self.cost = torch.empty((B,), device=x0.device, dtype=torch.float32)
        self.solver._workspace(B, x0.device)
        return self

    def cycle(self):
        """One solve + shift on the allocated buffers (capturable: no allocation, no host sync)."""
        torch = _torch()
        s = self.solver
        s.rollout(self.x0, self.U, out=self.X)
        for _ in range(self.iterations):
            J, _ = s.iterate(self.x0, self.X, self.U)
        self.cost.copy_(J)
        # advance: the state reached after the kept nodes becomes the next initial state
        self.x0.copy_(self.X[self.keep])
        if self.warm_start == "shift":
            tail = self.U[self.keep:].clone()
            self.U[: self.overlap].copy_(tail)
            self.U[self.overlap:].copy_(tail[-1:].expand(self.keep, -1, -1))
        else:
            self.U.zero_()

    def capture(self):
        """Capture one cycle into a hipGraph; afterwards step() replays it."""
        torch = _torch()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x0, U = self.x0.clone(), self.U.clone()
            self.cycle()  # warm-up launch outside capture (lazy initialisation, LDS attribute calls)
            torch.cuda.synchronize()
            self.x0.copy_(x0); self.U.copy_(U)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self.cycle()
            self.x0.copy_(x0); self.U.copy_(U)
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
                # X still holds the trajectory of this solve: its first `keep` steps were executed
                hist.append(self.X[1 : self.keep + 1].clone())
                del x_before
        return torch.cat(hist) if record else None

"""The track of the moving-horizon track tracker: a piecewise cubic Hermite curve over s in [0, 1] through the
sampled Dubins path — the `eval` / `eval_tangent` / `length` trio `MHTT` asks of its `track` argument
(reference control/initialisation.py:782-851, 738-758; consumer control/moving_horizon.py:34-39, 147-152).

Host side is float64 numpy and keeps the reference's closed-interval segment test (a value exactly on an interior
knot is counted by both neighbours, initialisation.py:818-819) so `length()` returns the reference's number.
`install()` hands the segment cubics to the HIP handle for the device kernels (csrc/ac_track.hpp), which reproduce the
double count at the knots an fp32 progress value can hit exactly.
The Dubins path construction itself (initialisation.py:94-226, 594) is out of scope: pass its sampled points.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib


class Track:
    def __init__(self, points):
        P = np.asarray(points, dtype=np.float64)
        if P.ndim != 2 or P.shape[1] != 3 or P.shape[0] < 2:
            raise ValueError(f"track points must be (n >= 2, 3), got {P.shape}")
        self.points = P
        n = P.shape[0]
        self.s_vals = np.linspace(0.0, 1.0, num=n)  # initialisation.py:830
        h = np.diff(self.s_vals)
        secant = np.diff(P, axis=0) / h[:, None]
        d = np.empty_like(P)  # knot derivatives: central average inside, one-sided at the ends (:796-808)
        d[1:-1] = 0.5 * (secant[:-1] + secant[1:])
        d[0], d[-1] = secant[0], secant[-1]
        self.h, self.d = h, d
        self._length = None

    @property
    def n_segments(self) -> int:
        return self.points.shape[0] - 1

    # ---- host evaluation (float64) ------------------------------------------------------------------
    def _segment(self, i, s, deriv):
        """Hermite segment i (array of indices) at s -> (3, m)."""
        y0, y1, d0, d1, h = self.points[i].T, self.points[i + 1].T, self.d[i].T, self.d[i + 1].T, self.h[i]
        t = (s - self.s_vals[i]) / h
        if not deriv:
            h00, h10 = (1 + 2 * t) * (1 - t) ** 2, t * (1 - t) ** 2
            h01, h11 = t ** 2 * (3 - 2 * t), t ** 2 * (t - 1)
            return h00 * y0 + h10 * h * d0 + h01 * y1 + h11 * h * d1
        g00, g10 = 6 * t * t - 6 * t, 3 * t * t - 4 * t + 1
        g01, g11 = -6 * t * t + 6 * t, 3 * t * t - 2 * t
        return (g00 * y0 + g10 * h * d0 + g01 * y1 + g11 * h * d1) / h

    def _eval(self, s, deriv):
        s_in = np.asarray(s, dtype=np.float64)
        sv = np.atleast_1d(s_in).ravel()
        n = self.points.shape[0]
        inside = (sv >= self.s_vals[0]) & (sv <= self.s_vals[-1])
        i = np.clip(np.searchsorted(self.s_vals, sv, side="right") - 1, 0, n - 2)
        out = np.where(inside, self._segment(i, sv, deriv), 0.0)
        # closed intervals: a point on interior knot i also satisfies segment i-1's test
        twice = inside & (sv == self.s_vals[i]) & (i > 0)
        if twice.any():
            out = out + np.where(twice, self._segment(np.maximum(i - 1, 0), sv, deriv), 0.0)
        if not deriv:  # constant extrapolation (:821-823); its derivative is zero
            out = out + np.where(sv < self.s_vals[0], 1.0, 0.0) * self.points[0][:, None]
            out = out + np.where(sv > self.s_vals[-1], 1.0, 0.0) * self.points[-1][:, None]
        return out[:, 0] if s_in.ndim == 0 else out.reshape((3,) + s_in.shape)

    def eval(self, s):
        """Position on the track at progress s -> (3,) or (3, ...)."""
        return self._eval(s, False)

    def eval_tangent(self, s):
        """d position / d s."""
        return self._eval(s, True)

    def length(self, N: int = 100) -> float:
        """Trapezoid rule on |d pos / d s| over N grid points (initialisation.py:738-758)."""
        if self._length is None or N != 100:
            grid = np.linspace(0.0, 1.0, N)
            speed = np.linalg.norm(self.eval_tangent(grid), axis=0)
            val = float(np.sum(0.5 * (1.0 / (N - 1)) * (speed[:-1] + speed[1:])))
            if N != 100:
                return val
            self._length = val
        return self._length

    # ---- device hand-off ------------------------------------------------------------------------------
    def segment_cubics(self) -> np.ndarray:
        """(n_segments, 3, 4) float32: per segment and axis c0 + c1 t + c2 t^2 + c3 t^3, t in [0, 1]."""
        y0, y1 = self.points[:-1], self.points[1:]
        m0, m1 = self.h[:, None] * self.d[:-1], self.h[:, None] * self.d[1:]
        c = np.stack([y0, m0, -3 * y0 - 2 * m0 + 3 * y1 - m1, 2 * y0 + m0 - 2 * y1 + m1], axis=-1)
        return np.ascontiguousarray(c, dtype=np.float32)

    def install(self, system) -> None:
        """Copy the track into `system`'s HIP handle (ac_set_track)."""
        lib = system._sync()
        coef = self.segment_cubics()
        _lib.check(lib.ac_set_track(system._handle, self.n_segments, coef.ctypes.data_as(C.POINTER(C.c_float)),
                                    C.c_float(self.length())), "ac_set_track")

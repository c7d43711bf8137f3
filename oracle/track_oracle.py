"""Float64 restatement of the reference's track and MHTT progress terms (TEST INFRASTRUCTURE, NOT PRODUCT).

Follows the reference line by line, scalar loops and all:
  hermite()/TrackOracle.eval, eval_tangent   control/initialisation.py:796-851 (sum of if_else over closed segments)
  TrackOracle.length                         control/initialisation.py:738-758
  progress_initial                           control/moving_horizon.py:216-233
  progress_tight / step_terms                control/moving_horizon.py:147-175
  mhtt_loss                                  control/moving_horizon.py:44-105

PARITY UNPINNED for the evaluation: the reference holds no stored progress data or test for these functions, and it needs
casadi (not in this image) to run, so this restatement is checked against hand-computed cases only
(tests/test_track_oracle.py).  Its INPUT is pinned: tests/golden/dubins_track.npz is the sampled Dubins path of the
reference's own problem definition, produced by the reference's aircraft.dubins (tests/golden/make_fixtures.py).
Only tests/ may import this module.
"""
import numpy as np


class TrackOracle:
    def __init__(self, points):
        self.P = [tuple(float(c) for c in p) for p in points]
        n = len(self.P)
        self.s_vals = np.linspace(0, 1, num=n)
        self.d = []
        for axis in range(3):
            y = np.array([p[axis] for p in self.P])
            h = np.diff(self.s_vals)
            slopes = np.diff(y) / h
            d = np.zeros_like(y)
            d[1:-1] = (slopes[:-1] + slopes[1:]) / 2
            d[0] = slopes[0]
            d[-1] = slopes[-1]
            self.d.append(d)

    def _interp(self, axis, s, deriv):
        s_vals = self.s_vals
        y_vals = [p[axis] for p in self.P]
        d = self.d[axis]
        expr = 0.0
        for i in range(len(s_vals) - 1):
            s0, s1 = s_vals[i], s_vals[i + 1]
            y0, y1 = y_vals[i], y_vals[i + 1]
            d0, d1 = d[i], d[i + 1]
            h_i = s1 - s0
            t = (s - s0) / h_i
            if not (s >= s0 and s <= s1):
                continue
            if not deriv:
                h00 = (1 + 2 * t) * (1 - t) ** 2
                h10 = t * (1 - t) ** 2
                h01 = t ** 2 * (3 - 2 * t)
                h11 = t ** 2 * (t - 1)
                expr += h00 * y0 + h10 * h_i * d0 + h01 * y1 + h11 * h_i * d1
            else:  # d/ds of the same polynomial (what ca.jacobian(pos, s) yields inside the if_else)
                dt = 1.0 / h_i
                g00 = (2 * (1 - t) ** 2 - 2 * (1 + 2 * t) * (1 - t)) * dt
                g10 = ((1 - t) ** 2 - 2 * t * (1 - t)) * dt
                g01 = (2 * t * (3 - 2 * t) - 2 * t ** 2) * dt
                g11 = (2 * t * (t - 1) + t ** 2) * dt
                expr += g00 * y0 + g10 * h_i * d0 + g01 * y1 + g11 * h_i * d1
        if not deriv:
            if s < s_vals[0]:
                expr += y_vals[0]
            if s > s_vals[-1]:
                expr += y_vals[-1]
        return expr

    def eval(self, s):
        return np.array([self._interp(a, float(s), False) for a in range(3)])

    def eval_tangent(self, s):
        return np.array([self._interp(a, float(s), True) for a in range(3)])

    def length(self, N=100):
        s_grid = np.linspace(0, 1, N)
        ds = 1 / (N - 1)
        length = 0.0
        for i in range(N - 1):
            vi = np.linalg.norm(self.eval_tangent(s_grid[i]))
            vi1 = np.linalg.norm(self.eval_tangent(s_grid[i + 1]))
            length += 0.5 * ds * (vi + vi1)
        return length


def progress_initial(track, track_length, X, s0, dt):
    """X (H+1,13,B), s0 (B,) -> S (H+1,B): the initial progress guess."""
    H, B = X.shape[0] - 1, X.shape[2]
    S = np.zeros((H + 1, B))
    S[0] = s0
    for b in range(B):
        for i in range(1, H + 1):
            vel = X[i - 1, 3:6, b]
            s_current = S[i - 1, b]
            tangent = track.eval_tangent(s_current)
            with np.errstate(invalid="ignore", divide="ignore"):
                tangent_norm = tangent / np.linalg.norm(tangent)
            s_dot = np.dot(vel, tangent_norm) / track_length
            S[i, b] = np.clip(s_current + s_dot * dt, 0, 1.0)
    return S


def step_terms(track, track_length, x, s):
    """(s_dot, delta_s_correction, tracking_error) of one node."""
    tangent = track.eval_tangent(s)
    norm = np.linalg.norm(tangent)
    norm_safe = norm if norm > 1e-3 else 1.0
    tangent_norm = tangent / norm_safe
    track_pos = track.eval(s)
    pos, vel = x[:3], x[3:6]
    s_dot = np.dot(vel, tangent_norm) / track_length
    pos_err = pos - track_pos
    delta_s_correction = np.dot(pos_err, tangent_norm) / track_length
    return s_dot, delta_s_correction, float(np.sum(pos_err ** 2))


def progress_tight(track, track_length, X, s0, dt):
    """Progress with the constraint row  s_{k+1} <= s_k + s_dot dt + 0.05 delta_s  held at its bound and the
    [0, 1] box applied.  Returns S (H+1,B), s_dot (H,B), tracking_error (H,B)."""
    H, B = X.shape[0] - 1, X.shape[2]
    S = np.zeros((H + 1, B)); sd = np.zeros((H, B)); te = np.zeros((H, B))
    S[0] = s0
    for b in range(B):
        for k in range(H):
            s_dot, corr, err = step_terms(track, track_length, X[k, :, b], S[k, b])
            sd[k, b], te[k, b] = s_dot, err
            S[k + 1, b] = min(max(S[k, b] + s_dot * dt + 0.05 * corr, 0.0), 1.0)
    return S, sd, te


DEFAULT_WEIGHTS = dict(w_tracking=10.0, w_progress=5.0, w_progress_rate=2.0, w_backward=50.0, w_terminal_align=20.0,
                       w_low_velocity=10.0, w_control=100.0)


def mhtt_loss(track, track_length, X, U, S, w=None):
    """X (H+1,13,B), U (H,7,B), S (H+1,B) -> (B,).  Node i >= 1 carries the tracking error and progress rate of
    node i-1; node H's control is a free variable outside U and zero at any optimum."""
    w = dict(DEFAULT_WEIGHTS, **(w or {}))
    H, B = U.shape[0], U.shape[2]
    out = np.zeros(B)
    for b in range(B):
        tracking_loss = progress_reward = progress_rate_reward = backward_penalty = 0.0
        low_velocity_penalty = control_effort = 0.0
        for i in range(1, H + 1):
            s_dot, _, err = step_terms(track, track_length, X[i - 1, :, b], S[i - 1, b])
            tracking_loss += err
            progress_reward += S[i, b]
            progress_rate_reward += s_dot
            backward_penalty += max(0.0, -s_dot) ** 2
            velocity = np.linalg.norm(X[i, 3:6, b])
            low_velocity_penalty += max(0.1 - velocity, 0.0) ** 2
            if i < H:
                control_effort += float(np.sum(U[i, :, b] ** 2))
        terminal = np.linalg.norm(X[H, :3, b] - track.eval(1.0))
        out[b] = (w["w_tracking"] * tracking_loss - w["w_progress"] * progress_reward
                  - w["w_progress_rate"] * progress_rate_reward + w["w_backward"] * backward_penalty
                  + w["w_low_velocity"] * low_velocity_penalty + w["w_terminal_align"] * terminal
                  + w["w_control"] * control_effort)
    return out

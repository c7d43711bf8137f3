"""NumPy float64 restatement of the batched iLQR sweep (TEST INFRASTRUCTURE, NOT PRODUCT).

Checks aircraft_amd/csrc/ac_ilqr.hpp: the backward Riccati pass, the closed-loop (feedback policy) rollout and the
quadratic cost.  The dynamics inside the policy rollout are the C++ oracle's state_update.  Plain loops over
instances: meant for small cases."""
import numpy as np


def cost(c, X, U, node=None):
    """X (H+1,13,B), U (H,7,B) -> (B,).  node = (q, xref, glin), each (H+1,13,Bn): per-node state cost replacing
    q/qf/x_ref/x_goal (include/aircraft_hip.h, ac_ilqr_cost_node_f32); instance b reads column b % Bn."""
    if node is not None:
        nq, nx, ng = (np.tile(a, (1, 1, X.shape[2] // a.shape[2])) for a in node)
        d = X - nx
        return (0.5 * nq * d * d + ng * X).sum(axis=(0, 1)) + 0.5 * (np.asarray(c.r)[None, :, None] * U * U).sum(axis=(0, 1)) + \
            (_u_lin(c)[None, :, None] * U).sum(axis=(0, 1))
    q, qf, r = np.asarray(c.q), np.asarray(c.qf), np.asarray(c.r)
    dx = X[:-1] - np.asarray(c.x_ref)[None, :, None]
    dg = X[-1] - np.asarray(c.x_goal)[:, None]
    return 0.5 * (q[None, :, None] * dx * dx).sum(axis=(0, 1)) + 0.5 * (r[None, :, None] * U * U).sum(axis=(0, 1)) + \
        0.5 * (qf[:, None] * dg * dg).sum(axis=0) + (_u_lin(c)[None, :, None] * U).sum(axis=(0, 1))


def _u_lin(c):
    """linear control cost (the time term w_time * dt_k on the time row); absent on older cost objects"""
    return np.asarray(getattr(c, "u_lin", [0.0] * 7), dtype=np.float64)


def costate(c, X, A, node=None):
    """Multipliers of the defect rows: Lam[H-1] = grad l_N(x_N), Lam[k-1] = grad l_k(x_k) + A_k' Lam[k] -> (H,13,B)"""
    H, B = A.shape[0], A.shape[3]
    q, qf = np.asarray(c.q, float), np.asarray(c.qf, float)
    Lam = np.zeros((H, 13, B))
    for b in range(B):
        if node is None:
            lam = qf * (X[H, :, b] - np.asarray(c.x_goal))
        else:
            lam = node[0][H, :, b] * (X[H, :, b] - node[1][H, :, b]) + node[2][H, :, b]
        for k in range(H - 1, -1, -1):
            Lam[k, :, b] = lam
            if k == 0:
                break
            if node is None:
                lx = q * (X[k, :, b] - np.asarray(c.x_ref))
            else:
                lx = node[0][k, :, b] * (X[k, :, b] - node[1][k, :, b]) + node[2][k, :, b]
            lam = lx + A[k, :, :, b].T @ lam
    return Lam


def backward(c, X, U, A, Bm, node=None, Hz=None, uglin=None):
    """A (H,13,13,B), Bm (H,13,7,B) -> K (H,7,13,B), kff (H,7,B), dV (2,B).  Hz (H,21,21,B): optional second-order
    dynamics blocks added to Qxx (rows/cols 0-12), Qux (rows 13-19, cols 0-12) and Quu (13-19, 13-19).  uglin (H,7,B):
    optional per-node control gradient added to Q_u (ac_ilqr_backward_goal_f32)."""
    H, _, B = U.shape
    q, qf, r = np.asarray(c.q, float), np.asarray(c.qf, float), np.asarray(c.r, float)
    K = np.zeros((H, 7, 13, B)); kff = np.zeros((H, 7, B)); dV = np.zeros((2, B))
    for b in range(B):
        if node is None:
            Vx = qf * (X[H, :, b] - np.asarray(c.x_goal)); Vxx = np.diag(qf)
        else:
            Vx = node[0][H, :, b] * (X[H, :, b] - node[1][H, :, b]) + node[2][H, :, b]; Vxx = np.diag(node[0][H, :, b])
        for k in range(H - 1, -1, -1):
            Ak, Bk = A[k, :, :, b], Bm[k, :, :, b]
            lx = q * (X[k, :, b] - np.asarray(c.x_ref)); lu = r * U[k, :, b] + _u_lin(c)
            if node is not None:
                q = node[0][k, :, b]
                lx = q * (X[k, :, b] - node[1][k, :, b]) + node[2][k, :, b]
            if uglin is not None:
                lu = lu + uglin[k, :, b]
            Qx = lx + Ak.T @ Vx; Qu = lu + Bk.T @ Vx
            Qxx = np.diag(q) + Ak.T @ Vxx @ Ak
            Qux = Bk.T @ Vxx @ Ak
            Quu = np.diag(r + c.reg) + Bk.T @ Vxx @ Bk
            if Hz is not None:
                Qxx = Qxx + Hz[k, :13, :13, b]; Qux = Qux + Hz[k, 13:20, :13, b]; Quu = Quu + Hz[k, 13:20, 13:20, b]
            Quu = 0.5 * (Quu + Quu.T)
            Kk = -np.linalg.solve(Quu, Qux); kk = -np.linalg.solve(Quu, Qu)
            K[k, :, :, b] = Kk; kff[k, :, b] = kk
            dV[0, b] += kk @ Qu; dV[1, b] += 0.5 * kk @ Quu @ kk
            Vx = Qx + Kk.T @ Quu @ kk + Kk.T @ Qu + Qux.T @ kk
            Vxx = Qxx + Kk.T @ Quu @ Kk + Kk.T @ Qux + Qux.T @ Kk
            Vxx = 0.5 * (Vxx + Vxx.T)
    return K, kff, dV


def forward(orc, c, x0, Xnom, U, K, kff, alphas, dt):
    """Closed-loop rollouts: returns Xc (H+1,13,na*B), Uc (H,7,na*B), column a*B+b."""
    H, _, B = U.shape
    na = len(alphas)
    Xc = np.zeros((H + 1, 13, na * B)); Uc = np.zeros((H, 7, na * B))
    umin, umax = np.asarray(c.u_min, float)[:, None], np.asarray(c.u_max, float)[:, None]
    for a, al in enumerate(alphas):
        x = x0.copy(); sl = slice(a * B, (a + 1) * B)
        Xc[0, :, sl] = x
        for k in range(H):
            dx = x - Xnom[k]
            u = U[k] + al * kff[k] + np.einsum("imb,mb->ib", K[k], dx)
            u = np.clip(u, umin, umax)
            Uc[k, :, sl] = u
            row = getattr(c, "dt_row", 0)
            x = orc.state_update(x, u, u[row] if row > 0 else dt)  # time as a decision variable: node k's own (clipped) step
            Xc[k + 1, :, sl] = x
    return Xc, Uc


# ---- the flight envelope as an augmented-Lagrangian term (ac_kernels_analytic.hpp: k_envelope_cost / _model / _multipliers) ----
def envelope_al(orc, X, lo, hi, w, lam=None):
    """X (Hn, 13, B); bounds lo, hi (4,); multipliers lam (Hn, 8, B) (rows 0-3 upper, 4-7 lower; None = zeros: the plain
    penalty).  Returns (cost (B,), gradient (Hn, 13, B), Gauss-Newton curvature (Hn, 13, 13, B), shifted violation (Hn, 4, B),
    rows (Hn, 4, B)) of  L_A = w sum_r max(0, g - hi + lam_hi/2w)^2 - (lam_hi/2w)^2 + max(0, lo - g + lam_lo/2w)^2 - (lam_lo/2w)^2."""
    Hn, _, B = X.shape
    lam = np.zeros((Hn, 8, B)) if lam is None else np.asarray(lam, dtype=np.float64)
    cost = np.zeros(B); grad = np.zeros((Hn, 13, B)); curv = np.zeros((Hn, 13, 13, B))
    sv = np.zeros((Hn, 4, B)); rows_all = np.zeros((Hn, 4, B))
    for k in range(Hn):
        rows, Jx = orc.envelope(X[k])
        sh, sl = lam[k, :4] / (2 * w), lam[k, 4:] / (2 * w)
        up = rows - hi[:, None] + sh
        dn = lo[:, None] - rows + sl
        # straight from the formula: the two max() terms are independent (both can be positive once the shifts overlap)
        vh, vl = np.maximum(0.0, up), np.maximum(0.0, dn)
        v = vh - vl
        cost += w * ((vh ** 2).sum(axis=0) + (vl ** 2).sum(axis=0) - (sh ** 2).sum(axis=0) - (sl ** 2).sum(axis=0))
        grad[k] = 2 * w * np.einsum("rb,rjb->jb", v, Jx)
        curv[k] = 2 * w * np.einsum("rb,rib,rjb->ijb", (vh > 0).astype(float) + (vl > 0).astype(float), Jx, Jx)
        sv[k], rows_all[k] = v, rows
    return cost, grad, curv, sv, rows_all


def envelope_al_update(rows, lo, hi, w, lam):
    """lam_hi <- max(0, lam_hi + 2w (g - hi)), lam_lo <- max(0, lam_lo + 2w (lo - g)); rows (Hn, 4, B) -> new lam (Hn, 8, B)"""
    up = rows - hi[None, :, None]
    dn = lo[None, :, None] - rows
    return np.concatenate([np.maximum(0.0, lam[:, :4] + 2 * w * up), np.maximum(0.0, lam[:, 4:] + 2 * w * dn)], axis=1)


# ---- the goal-acquisition loss of the reference's MPC driver (Controller.loss, /root/reference/main/control/control.py:44-68) ----
# Written from the reference's formulas (not from ac_goal.hpp): l0_smooth (:22-23), goal_loss (:60), control_loss (:49-50),
# height_loss (:67), speed_loss (:68-69), final_velocity_loss (:62-64), the constraint state[3, -1] < -2 (:55) as an
# augmented-Lagrangian term.  Differences to the reference's shapes are stated in ac_goal.hpp (N controls, not N + 1).
class GoalLoss:
    def __init__(self, w_goal=1000.0, w_rate=100.0, eps_rate=1e-2, w_height=1.0, w_speed=0.01, w_vx=1000.0, w_vyz=1000.0,
                 vx_max=-2.0, w_al=0.0, time_row=0):
        self.w_goal, self.w_rate, self.eps_rate, self.w_height, self.w_speed = w_goal, w_rate, eps_rate, w_height, w_speed
        self.w_vx, self.w_vyz, self.vx_max, self.w_al, self.time_row = w_vx, w_vyz, vx_max, w_al, time_row


def l0_smooth(x, epsilon):
    return 1.0 - np.exp(-x ** 2 / epsilon)


def _rate_rows(g):
    return [i for i in range(7) if not (g.time_row > 0 and i == g.time_row)]


def goal_cost(orc, g, goal, X, U, lam=None):
    """X (H+1,13,Bc), U (H,7,Bc), goal (2,Bn), lam (Bn,) -> (Bc,); column o belongs to instance o % Bn."""
    H, _, Bc = U.shape
    Bn = goal.shape[1]
    gl = np.tile(goal, (1, Bc // Bn))
    J = g.w_goal * ((X[H, 0] - gl[0]) ** 2 + (X[H, 1] - gl[1]) ** 2)
    du = U[1:] - U[:-1]
    J = J + g.w_rate * l0_smooth(du[:, _rate_rows(g)], g.eps_rate).sum(axis=(0, 1))
    J = J + g.w_height * (X[H, 2] - X[0, 2]) ** 2
    speed = np.zeros(Bc)
    for k in range(H):
        vr = orc.aero(X[k], np.zeros((7, Bc)))[:3]
        speed += (vr * vr).sum(axis=0)
    J = J - g.w_speed * speed / H
    J = J + g.w_vx * X[H, 3] + g.w_vyz * (X[H, 4] ** 2 + X[H, 5] ** 2)
    if g.w_al > 0:
        s = (np.zeros(Bn) if lam is None else np.asarray(lam, float)) / (2 * g.w_al)
        s = np.tile(s, Bc // Bn)
        J = J + g.w_al * (np.maximum(0.0, X[H, 3] - g.vx_max + s) ** 2 - s ** 2)
    return J


def goal_model(orc, g, goal, X, U, lam=None):
    """The quadratic model of ac_goal_model_f32 around (X, U): node_q, node_xref, node_glin (H+1,13,B), uglin (H,7,B) and
    the rate curvature on the (u,u) diagonal (H,7,B)."""
    H, _, B = U.shape
    nq = np.zeros((H + 1, 13, B)); nx = np.zeros((H + 1, 13, B)); ng = np.zeros((H + 1, 13, B))
    for k in range(H):
        rows, Jx = orc.envelope(X[k])           # row 0 = v_rel . v_rel and its exact state Jacobian
        ng[k] = -(g.w_speed / H) * Jx[0]
    nq[H, 0] = nq[H, 1] = 2 * g.w_goal; nx[H, 0], nx[H, 1] = goal[0], goal[1]
    nq[H, 2] = 2 * g.w_height; nx[H, 2] = X[0, 2]
    nq[H, 4] = nq[H, 5] = 2 * g.w_vyz
    ng[H, 3] = g.w_vx
    if g.w_al > 0:
        s = (np.zeros(B) if lam is None else np.asarray(lam, float)) / (2 * g.w_al)
        act = X[H, 3] - g.vx_max + s > 0
        nq[H, 3] = np.where(act, 2 * g.w_al, 0.0); nx[H, 3] = np.where(act, g.vx_max - s, 0.0)
    eps = g.eps_rate
    d = U[1:] - U[:-1]                          # (H-1, 7, B): d[k] = u_{k+1} - u_k
    e = np.exp(-d ** 2 / eps)
    gp = (2 * d / eps) * e                      # l0'
    l0 = 1.0 - e
    with np.errstate(divide="ignore", invalid="ignore"):
        hp = np.where(l0 > 1e-12, gp ** 2 / (2 * np.maximum(l0, 1e-300)), (2 / eps) * e)   # Gauss-Newton curvature, -> 2/eps
    ug = np.zeros((H, 7, B)); uh = np.zeros((H, 7, B))
    ug[1:] += gp; ug[:-1] -= gp
    uh[1:] += hp; uh[:-1] += hp
    mask = np.zeros(7); mask[_rate_rows(g)] = 1.0
    return nq, nx, ng, g.w_rate * ug * mask[None, :, None], g.w_rate * uh * mask[None, :, None]


def goal_multiplier(g, X, lam):
    exc = X[-1, 3] - g.vx_max
    return np.maximum(0.0, lam + 2 * g.w_al * exc), np.maximum(0.0, exc)

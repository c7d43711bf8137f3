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


def backward(c, X, U, A, Bm, node=None, Hz=None):
    """A (H,13,13,B), Bm (H,13,7,B) -> K (H,7,13,B), kff (H,7,B), dV (2,B).  Hz (H,21,21,B): optional second-order
    dynamics blocks added to Qxx (rows/cols 0-12), Qux (rows 13-19, cols 0-12) and Quu (13-19, 13-19)."""
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
        v = np.where(up > 0, up, np.where(dn > 0, -dn, 0.0))
        cost += w * ((v ** 2).sum(axis=0) - (sh ** 2).sum(axis=0) - (sl ** 2).sum(axis=0))
        grad[k] = 2 * w * np.einsum("rb,rjb->jb", v, Jx)
        curv[k] = 2 * w * np.einsum("rb,rib,rjb->ijb", (v != 0).astype(float), Jx, Jx)
        sv[k], rows_all[k] = v, rows
    return cost, grad, curv, sv, rows_all


def envelope_al_update(rows, lo, hi, w, lam):
    """lam_hi <- max(0, lam_hi + 2w (g - hi)), lam_lo <- max(0, lam_lo + 2w (lo - g)); rows (Hn, 4, B) -> new lam (Hn, 8, B)"""
    up = rows - hi[None, :, None]
    dn = lo[None, :, None] - rows
    return np.concatenate([np.maximum(0.0, lam[:, :4] + 2 * w * up), np.maximum(0.0, lam[:, 4:] + 2 * w * dn)], axis=1)

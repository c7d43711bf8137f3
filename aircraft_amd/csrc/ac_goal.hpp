// ac_goal.hpp — the goal-acquisition loss of the reference's MPC driver (Controller.loss, main/control/control.py:44-68) for
// the batched sweep (the `loss` role of ControlProblem):
//
//   J = w_goal |p_xy(N) - goal|^2                                    goal_loss            (:60,  1000)
//     + w_rate sum_{k<N-1} sum_i l0(u_{k+1,i} - u_{k,i}; eps)        control_loss         (:49-50, 100, eps 1e-2),  l0(d) = 1 - exp(-d^2/eps)
//     + w_height (z_N - z_0)^2                                       height_loss          (:67)
//     - (w_speed / N) sum_{k<N} v_rel(x_k) . v_rel(x_k)              speed_loss           (:68-69, 1/100)
//     + w_vx v_x(N) + w_vyz (v_y(N)^2 + v_z(N)^2)                    final_velocity_loss  (:62-64, vel_param 1000, 1000)
//     [ + w_time sum dt_k: the linear control cost of the time row,  IlqrCost::u_lin ]     time_loss (:46, :71-72, 10000)
//   subject to  v_x(N) < vx_max                                     "final velocity constraint" (:55, -2)
//
// IPOPT sees the exact NLP; the sweep sees
//   * the EXACT value (k_goal_cost) wherever it compares trajectories (line search, acceptance, history), with the
//     inequality as an augmented-Lagrangian term  w_al max(0, v_x(N) - vx_max + lam / 2 w_al)^2 - (lam / 2 w_al)^2  and a
//     first-order multiplier update per instance (k_goal_multiplier);
//   * a convex quadratic MODEL around the iterate in the backward pass (k_goal_model): the terminal quadratics as they are,
//     the linear terms and the gradient of the (concave) speed reward as gradients, the control-rate term by its exact
//     gradient with respect to u_k — neighbours u_{k-1}, u_{k+1} held at the iterate — and the Gauss-Newton curvature of
//     both differences u_k takes part in, (l0')^2 / (2 l0) -> 2 / eps, on the diagonal of Q_uu.  (The cross terms between
//     consecutive controls are left to the line search; an exact treatment carries u_{k-1} as seven more states through
//     the Riccati pass.)
// The reference's control matrix has N + 1 columns (N differences); the sweep's has N (N - 1 differences).
#pragma once
#include "ac_ilqr.hpp"

namespace ac {

struct GoalLoss {
    float w_goal, w_rate, eps_rate, w_height, w_speed, w_vx, w_vyz;
    float vx_max, w_al;  // terminal inequality v_x(N) < vx_max; w_al = 0: not enforced
    int time_row;        // control row that carries dt_k (excluded from the rate term); <= 0: none
};

// (1 - exp(-t) as -expm1(-t): the differences of consecutive controls are small, and 1 - expf() keeps three digits of them)
AC_DI float l0_smooth(float d, float eps) { return -expm1f(-(d * d) / eps); }
// derivative and Gauss-Newton curvature (r = sqrt(2 l0): r'^2 = l0'^2 / (2 l0), -> 2 / eps at d = 0)
AC_DI void l0_model(float d, float eps, float& g, float& h) {
    const float t = (d * d) / eps;
    const float e = expf(-t);
    const float l = -expm1f(-t);
    g = (2.0f * d / eps) * e;
    h = l > 1e-12f ? (g * g) / (2.0f * l) : (2.0f / eps) * e;
}

// cost[o] += J(X[:, :, o], U[:, :, o]) for every column o of a candidate batch (instance b = o % Bn owns the goal, the
// multiplier and, through column o's own node 0, z_0).  X [H+1][13][B], U [H][7][B]; goal [2][Bn]; lam [Bn] or NULL.
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_goal_cost(const DevParams P, const GoalLoss G, const float* __restrict__ goal,
                                                      const float* __restrict__ lam, long Bn, const float* __restrict__ X,
                                                      const float* __restrict__ U, long B, long H, float* __restrict__ cost) {
    const long o = (long)blockIdx.x * kBlock + threadIdx.x;
    if (o >= B) return;
    const long b = o % Bn;
    float acc = 0.f;
    float speed = 0.f;
    for (long k = 0; k < H; ++k) {
        float x[13];
        load_rows<13>(X + k * 13 * B, B, o, x);
        AeroPre<float> a;
        aero_pre(P, x, a);
        speed += a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2];
    }
    acc -= (G.w_speed / (float)H) * speed;
    float rate = 0.f;
    for (long k = 0; k + 1 < H; ++k) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i == G.time_row && G.time_row > 0) continue;
            const float d = U[((k + 1) * 7 + i) * B + o] - U[(k * 7 + i) * B + o];
            rate += l0_smooth(d, G.eps_rate);
        }
    }
    acc = fmaf(G.w_rate, rate, acc);
    const float* xN = X + H * 13 * B + o;
    const float dx = xN[0] - goal[b], dy = xN[B] - goal[Bn + b];
    acc = fmaf(G.w_goal, dx * dx + dy * dy, acc);
    const float dz = xN[2 * B] - X[2 * B + o];
    acc = fmaf(G.w_height, dz * dz, acc);
    const float vx = xN[3 * B], vy = xN[4 * B], vz = xN[5 * B];
    acc = fmaf(G.w_vx, vx, acc);
    acc = fmaf(G.w_vyz, vy * vy + vz * vz, acc);
    if (G.w_al > 0.f) {
        const float s = (lam ? lam[b] : 0.f) * (0.5f / G.w_al);
        const float v = fmaxf(0.f, vx - G.vx_max + s);
        acc += G.w_al * (v * v - s * s);
    }
    cost[o] += acc;
}

// The quadratic model around the iterate, in the form the backward pass consumes: node_q / node_xref / node_glin
// [H+1][13][B] (WRITTEN, every entry), node_uglin [H][7][B] (written), and the rate curvature ADDED to the diagonal of the
// (u, u) block of Hz [H][21][21][B].  One lane per (node, instance).
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_goal_model(const DevParams P, const GoalLoss G, const float* __restrict__ goal,
                                                       const float* __restrict__ lam, const float* __restrict__ X,
                                                       const float* __restrict__ U, long B, long H,
                                                       float* __restrict__ node_q, float* __restrict__ node_xref,
                                                       float* __restrict__ node_glin, float* __restrict__ node_uglin,
                                                       float* __restrict__ Hz) {
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (H + 1) * B) return;
    const long k = t / B, b = t % B;
    float q[13], xr[13], gl[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) { q[j] = 0.f; xr[j] = 0.f; gl[j] = 0.f; }
    float xv[13];
    load_rows<13>(X + k * 13 * B, B, b, xv);
    if (k < H) {
        // speed reward: - (w_speed / N) grad (v_rel . v_rel); the rows depend on (v, q) only
        typedef Dual<7> T;
        T x[13];
#pragma unroll
        for (int r = 0; r < 13; ++r) { x[r] = T(xv[r]); if (r >= 3 && r < 10) x[r].d[r - 3] = 1.f; }
        AeroPre<T> a;
        aero_pre(P, x, a);
        const T vv = a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2];
        const float w = -(G.w_speed / (float)H);
#pragma unroll
        for (int j = 0; j < 7; ++j) gl[3 + j] = w * vv.d[j];
        // control-rate term: gradient with respect to u_k, Gauss-Newton curvature of the two differences u_k takes part in
        float* hz = Hz + k * 441 * B + b;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            float g = 0.f, h = 0.f;
            if (!(i == G.time_row && G.time_row > 0)) {
                const float u = U[(k * 7 + i) * B + b];
                if (k > 0) {
                    float g1, h1;
                    l0_model(u - U[((k - 1) * 7 + i) * B + b], G.eps_rate, g1, h1);
                    g += g1; h += h1;
                }
                if (k + 1 < H) {
                    float g1, h1;
                    l0_model(U[((k + 1) * 7 + i) * B + b] - u, G.eps_rate, g1, h1);
                    g -= g1; h += h1;
                }
            }
            node_uglin[(k * 7 + i) * B + b] = G.w_rate * g;
            if (Hz && h != 0.f) hz[(long)((13 + i) * 21 + 13 + i) * B] += G.w_rate * h;
        }
    } else {
        q[0] = q[1] = 2.0f * G.w_goal; xr[0] = goal[b]; xr[1] = goal[B + b];
        q[2] = 2.0f * G.w_height; xr[2] = X[2 * B + b];
        q[4] = q[5] = 2.0f * G.w_vyz;
        gl[3] = G.w_vx;
        if (G.w_al > 0.f) {
            const float s = (lam ? lam[b] : 0.f) * (0.5f / G.w_al);
            if (xv[3] - G.vx_max + s > 0.f) { q[3] = 2.0f * G.w_al; xr[3] = G.vx_max - s; }
        }
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        const long o = (k * 13 + j) * B + b;
        node_q[o] = q[j]; node_xref[o] = xr[j]; node_glin[o] = gl[j];
    }
}

// lam <- max(0, lam + 2 w_al (v_x(N) - vx_max)); viol [B] (may be NULL) = max(0, v_x(N) - vx_max) before the update
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_goal_multiplier(const GoalLoss G, const float* __restrict__ X, long B, long H,
                                                            float* __restrict__ lam, float* __restrict__ viol) {
    const long b = (long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float exc = X[(H * 13 + 3) * B + b] - G.vx_max;
    lam[b] = fmaxf(0.f, fmaf(2.0f * G.w_al, exc, lam[b]));
    if (viol) viol[b] = fmaxf(0.f, exc);
}

}  // namespace ac

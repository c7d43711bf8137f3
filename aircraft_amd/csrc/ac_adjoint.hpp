// ac_adjoint.hpp — second-order blocks by FORWARD-OVER-REVERSE: the gradient of  phi(z) = lambda . F(x, u, dt)  over
// z = (x, u, dt) by one reverse sweep through the RK4 step, evaluated in first-order duals, so that the tangent parts of the
// gradient are N columns of the Hessian  sum_i lambda_i d2F_i / dz dz  (the `nlp_hess_l` blocks of the
// reference's NLP, todo.md:102).
//
// Why: k_step_hess (ac_hess.hpp) pushes second-order jets (value, one outer and N inner directions, N mixed terms) through
// the step once per PAIR of directions — 71 lanes per unit at N = 2, each with its own copy of the primal.  A reverse sweep
// yields the whole gradient (21 entries) for about three evaluations of f, and its dual costs (1 + N) times that: a unit
// needs 16 / N lanes — 8 at N = 2 — and about a seventh of the instructions.
//
// Everything here is templated on the scalar T (float: the plain gradient A' lambda, B' lambda, c . lambda — used by the host
// build of these headers to pin the adjoint against the oracle's exact Jacobians; Dual<N>: the Hessian columns).
//   f_vjp      x_bar += (df/dx)' w,  u_bar += (df/du)' w   at (x, u), with f itself recomputed inside (no tape)
//   rk4_vjp    the same for one RK4 step (stage states kept, stage derivatives recomputed by f_vjp) incl. d/d(dt)
// Coefficient providers implement  vjp(P, a, x, u, Cbar, abar, wbar, ubar)  next to operator(): the adjoint of their map
// (qbar, alpha, beta, omega, controls) -> C[6].
#pragma once
#include "ac_dynamics.hpp"

namespace ac {

template <class T> AC_DI Q4<T> qconj(const Q4<T>& q) { return Q4<T>{-q.x, -q.y, -q.z, q.w}; }
template <class T> AC_DI void cross3(const T a[3], const T b[3], T o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <class T> AC_DI T zero_like(const T&) { return T(0.f); }

// adjoint of the aerodynamic inputs a coefficient model reads
template <class T> struct AeroBar { T qbar, alpha, beta, vr[3]; };  // (vr: the cubic fits' effective angles read the relative velocity itself)

// ---- adjoints of the coefficient models ------------------------------------------------------------------------------------
// Adjoint of the cubic-fit model given `grad(k, f, g)`: the gradient of fit k at the point f, in T arithmetic.
template <class T, class G>
AC_DI void poly_vjp(const DevParams& P, const AeroPre<T>& a, const T x[13], const T u[7], const T Cb[6], AeroBar<T>& ab, T wb[3],
                    T ub[7], G&& grad) {
    // PolynomialModel (coefficient_models.py:106-133) with the effective angles of aircraft.py:189-233:
    //   C_0..3 = P_0..3(alpha, beta, da, de),  C_3 += b/8 (P_CZ(alpha_r) - P_CZ(alpha_l)),
    //   C_4 = P_4(alpha_e, beta, da, de),  C_5 = P_5(alpha, beta_r, da, de) + 0.01 * 6 deg * dr
    const T* w = &x[10];
    const float eps = P.p.epsilon, arm = P.p.rudder_moment_arm, b4 = P.p.b * 0.25f;
    const T ux = a.vr[0] + eps;
    const T ye = a.vr[2] + arm * w[1], yl = a.vr[2] - b4 * w[0], yr = a.vr[2] + b4 * w[0];
    const T alpha_e = m_atan2(ye, ux), alpha_l = m_atan2(yl, ux), alpha_r = m_atan2(yr, ux);
    const T vy = a.vr[1] - arm * w[2];
    const T nb = m_sqrt(a.vr[0] * a.vr[0] + vy * vy + a.vr[2] * a.vr[2] + eps);
    const T tb = vy / nb;
    const T beta_r = m_asin(tb);
    const PolyTab tab(P);
    T fb_main[4] = {T(0.f), T(0.f), T(0.f), T(0.f)};  // adjoints of (alpha, beta, da, de) at the main point
    const T fm[4] = {a.alpha, a.beta, u[0], u[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T g[4];
        grad(k, fm, g);
#pragma unroll
        for (int v = 0; v < 4; ++v) fb_main[v] = fb_main[v] + Cb[k] * g[v];
    }
    T aeb, brb;  // adjoints of alpha_e, beta_r
    {
        const T fe[4] = {alpha_e, a.beta, u[0], u[1]};
        T g[4];
        grad(4, fe, g);
        aeb = Cb[4] * g[0];
        fb_main[1] = fb_main[1] + Cb[4] * g[1]; fb_main[2] = fb_main[2] + Cb[4] * g[2]; fb_main[3] = fb_main[3] + Cb[4] * g[3];
    }
    {
        const T fr[4] = {a.alpha, beta_r, u[0], u[1]};
        T g[4];
        grad(5, fr, g);
        fb_main[0] = fb_main[0] + Cb[5] * g[0];
        brb = Cb[5] * g[1];
        fb_main[2] = fb_main[2] + Cb[5] * g[2]; fb_main[3] = fb_main[3] + Cb[5] * g[3];
    }
    ab.alpha = ab.alpha + fb_main[0]; ab.beta = ab.beta + fb_main[1];
    ub[0] = ub[0] + fb_main[2]; ub[1] = ub[1] + fb_main[3];
    ub[2] = ub[2] + (0.01f * 6.0f * kDeg) * Cb[5];
    // P_CZ(alpha_x, 0, 0, 0) = z0 + z1 a + z2 a^2 + z3 a^3
    const float z1 = tab.coef(2, 0), z2 = tab.coef(2, 4), z3 = tab.coef(2, 14);
    const float hb = b4 * 0.5f;
    const T arb = (hb * Cb[3]) * (z1 + (2.0f * z2) * alpha_r + (3.0f * z3) * (alpha_r * alpha_r));
    const T alb = -((hb * Cb[3]) * (z1 + (2.0f * z2) * alpha_l + (3.0f * z3) * (alpha_l * alpha_l)));
    // the three atan2(y_x, ux)
    T uxb = T(0.f);
    auto atan2_bar = [&](const T& y, const T& bar, T& yb) {
        const T iden = 1.0f / (ux * ux + y * y);
        yb = bar * ux * iden;
        uxb = uxb - bar * y * iden;
    };
    T yeb, ylb, yrb;
    atan2_bar(ye, aeb, yeb); atan2_bar(yl, alb, ylb); atan2_bar(yr, arb, yrb);
    ab.vr[0] = ab.vr[0] + uxb;
    ab.vr[2] = ab.vr[2] + yeb + ylb + yrb;
    wb[1] = wb[1] + arm * yeb;
    wb[0] = wb[0] + b4 * (yrb - ylb);
    // beta_r = asin(vy / |(vr0, vy, vr2)|_eps)
    const T tbar = brb / m_sqrt(1.0f - tb * tb);
    T vyb = tbar / nb;
    const T sb2 = (-(tbar * tb) / nb) * (0.5f / nb);   // adjoint of the sum of squares
    ab.vr[0] = ab.vr[0] + 2.0f * a.vr[0] * sb2;
    vyb = vyb + 2.0f * vy * sb2;
    ab.vr[2] = ab.vr[2] + 2.0f * a.vr[2] * sb2;
    ab.vr[1] = ab.vr[1] + vyb;
    wb[2] = wb[2] - arm * vyb;
}

template <int MODEL> struct AdjAnalyticCoeffs : AnalyticCoeffs<MODEL> {
    static constexpr bool kFusedTangent = false;  // (the reverse sweep evaluates f through the generic operator forms)
    AC_DI AdjAnalyticCoeffs() {}
    template <class A, class B> AC_DI AdjAnalyticCoeffs(const A&, const B&) {}
    AC_DI void set_stage(int) {}
    template <class T>
    AC_DI void vjp(const DevParams& P, const AeroPre<T>& a, const T x[13], const T u[7], const T Cb[6], AeroBar<T>& ab, T wb[3],
                   T ub[7]) const {
        (void)x; (void)u;
        if constexpr (MODEL == AC_MODEL_LINEAR) {
            T* in[3] = {&ab.qbar, &ab.alpha, &ab.beta};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
#pragma unroll
                for (int j = 0; j < 3; ++j) *in[j] = *in[j] + P.linear_W[k * 6 + j] * Cb[k];
                ub[0] = ub[0] + P.linear_W[k * 6 + 3] * Cb[k];
                ub[1] = ub[1] + P.linear_W[k * 6 + 4] * Cb[k];
            }
            ub[2] = ub[2] + (-0.1f * 6.0f * kDeg) * Cb[5];
        } else if constexpr (MODEL == AC_MODEL_POLY) {
            // gradients of the fits in T arithmetic (the quadratic tables of DevParams::poly_tab)
            const PolyTab tab(P);
            auto grad = [&](int k, const T f[4], T g[4]) {
                T m2[10];
                int t = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = i; j < 4; ++j) m2[t++] = f[i] * f[j];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    T ga = T(tab.grad(k, v, 0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) ga = ga + tab.grad(k, v, 1 + q) * f[q];
#pragma unroll
                    for (int q = 0; q < 10; ++q) ga = ga + tab.grad(k, v, 5 + q) * m2[q];
                    g[v] = ga;
                }
            };
            poly_vjp(P, a, x, u, Cb, ab, wb, ub, grad);
        } else {  // DefaultModel (coefficient_models.py:41-78)
            static_assert(MODEL == AC_MODEL_DEFAULT, "adjoint: default, linear and cubic-fit models");
            ab.alpha = ab.alpha + (-0.6f * a.alpha) * Cb[0] + (-5.0f) * Cb[2];
            ab.beta = ab.beta + (-0.98f) * Cb[1];
            ub[0] = ub[0] + (0.08f * 4.0f * kDeg) * Cb[3]; wb[0] = wb[0] + (-0.05f) * Cb[3];
            ub[1] = ub[1] + (-1.2f * 5.0f * kDeg) * Cb[4]; wb[1] = wb[1] + (-0.5f) * Cb[4];
            ub[2] = ub[2] + (-0.1f * 6.0f * kDeg) * Cb[5]; wb[2] = wb[2] + (-0.05f) * Cb[5];
        }
    }
};

// ---- x_dot = f(x, u) and the adjoint of its Jacobians applied to w ---------------------------------------------------------
// xd: f itself (recomputed here); xb [13] += (df/dx)' w, ub [7] += (df/du)' w.  (p never enters f: xb[0..2] untouched.)
template <class T, class Coeffs>
AC_DI void f_vjp(const DevParams& P, Coeffs& coeffs, const T x[13], const T u[7], const T w[13], T xd[13], T xb[13], T ub[7]) {
    const float eps = P.p.epsilon;
    const float im = 1.0f / P.p.mass;
    const float* I = P.p.inertia;
    const float* Ii = P.p.inertia_inv;
    const Q4<T> q{x[6], x[7], x[8], x[9]};
    const T om[3] = {x[10], x[11], x[12]};
    // ---- forward ----
    AeroPost<T> o;
    AeroPre<T> a;
    Q4<T> qi = qinv(q);
    T r[3], C[6], C2raw, C4raw, sa = T(1.f), sb = T(1.f), ea = T(0.f), ebv = T(0.f), qS = T(0.f), tb = T(0.f), ux = T(0.f);
    float sg = 1.f;
    if constexpr (Coeffs::kModel == AC_MODEL_QUAD) {
        quad_forces(P, u, o);
    } else {
        const Q4<T> rq = qmul(qmul_vec(qi, x[3], x[4], x[5]), q);
        r[0] = rq.x; r[1] = rq.y; r[2] = rq.z;
        a.vr[0] = r[0] + eps; a.vr[1] = r[1] + eps; a.vr[2] = r[2] + eps;
        const T vv = a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2];
        a.V = m_sqrt(vv + eps);
        ux = a.vr[0] + eps;
        a.alpha = m_atan2(a.vr[2], ux);
        tb = a.vr[1] / a.V;
        a.beta = m_asin(tb);
        a.qbar = (0.5f * 1.225f) * vv;
        coeffs(P, a, x, u, C);
        C2raw = C[2]; C4raw = C[4];
        if (P.p.stall_scaling) {
            const float lim = 30.0f * kDeg, steep = 10.0f;
            ea = m_exp(steep * (m_fabs(a.alpha) - lim));
            ebv = m_exp(steep * (m_fabs(a.beta) - lim));
            sa = 1.0f / (1.0f + ea);
            sb = 1.0f / (1.0f + ebv);
            C[2] = C[2] * sa; C[2] = C[2] * sb; C[4] = C[4] * sa;
        }
        C[0] = C[0] + (-0.1f) * u[6];
        C[2] = C[2] + (-0.6f) * u[6];
        qS = a.qbar * P.p.S;
        sg = sign_of(value_of(a.vr[0]));
#pragma unroll
        for (int k = 0; k < 3; ++k) o.F[k] = C[k] * qS;
        o.F[0] = o.F[0] * sg;
        const T Ma0 = C[3] * qS * P.p.b, Ma1 = C[4] * qS * P.p.c, Ma2 = C[5] * qS * P.p.b;
        o.M[0] = Ma0 + (P.p.com[1] * o.F[2] - P.p.com[2] * o.F[1]);
        o.M[1] = Ma1 + (P.p.com[2] * o.F[0] - P.p.com[0] * o.F[2]);
        o.M[2] = Ma2 + (P.p.com[0] * o.F[1] - P.p.com[1] * o.F[0]);
    }
    const Q4<T> Fnq = qmul(qmul_vec(q, o.F[0], o.F[1], o.F[2]), qi);
    const T Fn[3] = {Fnq.x, Fnq.y, Fnq.z};
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
#pragma unroll
    for (int k = 0; k < 3; ++k) xd[3 + k] = Fn[k] * im + P.p.gravity[k];
    const Q4<T> hq{0.5f * q.x, 0.5f * q.y, 0.5f * q.z, 0.5f * q.w};
    const Q4<T> qd = qmul_vec(hq, om[0], om[1], om[2]);
    xd[6] = qd.x; xd[7] = qd.y; xd[8] = qd.z; xd[9] = qd.w;
    T Iw[3], rhs[3], yv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Iw[i] = I[3 * i] * om[0] + I[3 * i + 1] * om[1] + I[3 * i + 2] * om[2];
    cross3(om, Iw, yv);
#pragma unroll
    for (int i = 0; i < 3; ++i) rhs[i] = o.M[i] - yv[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[10 + i] = Ii[3 * i] * rhs[0] + Ii[3 * i + 1] * rhs[1] + Ii[3 * i + 2] * rhs[2];

    // ---- reverse ----
    // p_dot = v
    xb[3] = xb[3] + w[0]; xb[4] = xb[4] + w[1]; xb[5] = xb[5] + w[2];
    // q_dot = 1/2 q (x) (omega, 0):  q_bar += 1/2 wq (x) conj(omega, 0),  omega_bar += 1/2 vec(conj(q) (x) wq)
    {
        const Q4<T> wq{w[6], w[7], w[8], w[9]};
        const Q4<T> qb = qmul_vec(wq, -om[0], -om[1], -om[2]);
        xb[6] = xb[6] + 0.5f * qb.x; xb[7] = xb[7] + 0.5f * qb.y; xb[8] = xb[8] + 0.5f * qb.z; xb[9] = xb[9] + 0.5f * qb.w;
        const Q4<T> ob = qmul(qconj(q), wq);
        xb[10] = xb[10] + 0.5f * ob.x; xb[11] = xb[11] + 0.5f * ob.y; xb[12] = xb[12] + 0.5f * ob.z;
    }
    // omega_dot = I^-1 (M - omega x I omega)
    T Mb[3];
    {
        T rb[3], yb[3], c1[3], c2[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) rb[j] = Ii[j] * w[10] + Ii[3 + j] * w[11] + Ii[6 + j] * w[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) { Mb[j] = rb[j]; yb[j] = -rb[j]; }
        // y = omega x (I omega):  omega_bar += (I omega) x y_bar + I' (y_bar x omega)
        cross3(Iw, yb, c1);
        cross3(yb, om, c2);
#pragma unroll
        for (int j = 0; j < 3; ++j) xb[10 + j] = xb[10 + j] + c1[j] + (I[j] * c2[0] + I[3 + j] * c2[1] + I[6 + j] * c2[2]);
    }
    // v_dot = Fn / m + g,  Fn = q (F, 0) q^-1
    T Fb[3];
    {
        const T Fnb[3] = {w[3] * im, w[4] * im, w[5] * im};
        const Q4<T> fb = qmul(qmul_vec(qi, Fnb[0], Fnb[1], Fnb[2]), q);   // R' Fn_bar
        Fb[0] = fb.x; Fb[1] = fb.y; Fb[2] = fb.z;
        T e[3];
        cross3(Fn, Fnb, e);                                              // e'_bar / 2 = Fn x Fn_bar
        const Q4<T> qb = qmul(Q4<T>{2.0f * e[0], 2.0f * e[1], 2.0f * e[2], zero_like(e[0])}, qconj(qi));
        xb[6] = xb[6] + qb.x; xb[7] = xb[7] + qb.y; xb[8] = xb[8] + qb.z; xb[9] = xb[9] + qb.w;
    }
    if constexpr (Coeffs::kModel == AC_MODEL_QUAD) {
        // M = Ma(u) + com x F,  F = (0, 0, u0 + u1 + u2 + u3)
        T c[3];
        const T comv[3] = {T(P.p.com[0]), T(P.p.com[1]), T(P.p.com[2])};
        cross3(Mb, comv, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) Fb[k] = Fb[k] + c[k];
        const float s0[4] = {1.f, -1.f, -1.f, 1.f}, s1[4] = {-1.f, -1.f, 1.f, 1.f}, s2[4] = {0.5f, -0.5f, 0.5f, -0.5f};
#pragma unroll
        for (int i = 0; i < 4; ++i) ub[i] = ub[i] + Fb[2] + s0[i] * Mb[0] + s1[i] * Mb[1] + s2[i] * Mb[2];
        return;
    } else {
        // M = Ma + com x F
        {
            T c[3];
            const T comv[3] = {T(P.p.com[0]), T(P.p.com[1]), T(P.p.com[2])};
            cross3(Mb, comv, c);
#pragma unroll
            for (int k = 0; k < 3; ++k) Fb[k] = Fb[k] + c[k];
        }
        // F_k = sign_k C_k qS,  Ma_k = len_k C_{3+k} qS
        const float len[3] = {P.p.b, P.p.c, P.p.b};
        T Cb[6];
        T qSb = T(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sc = k == 0 ? sg : 1.0f;
            Cb[k] = (sc * qS) * Fb[k];
            Cb[3 + k] = (len[k] * qS) * Mb[k];
            qSb = qSb + (sc * C[k]) * Fb[k] + (len[k] * C[3 + k]) * Mb[k];
        }
        AeroBar<T> ab;
        ab.qbar = P.p.S * qSb; ab.alpha = T(0.f); ab.beta = T(0.f);
        ab.vr[0] = T(0.f); ab.vr[1] = T(0.f); ab.vr[2] = T(0.f);
        ub[6] = ub[6] + (-0.1f) * Cb[0] + (-0.6f) * Cb[2];
        if (P.p.stall_scaling) {
            const float steep = 10.0f;
            const T sab = Cb[2] * C2raw * sb + Cb[4] * C4raw;
            const T sbb = Cb[2] * C2raw * sa;
            Cb[2] = Cb[2] * sa * sb;
            Cb[4] = Cb[4] * sa;
            const float sga = value_of(a.alpha) < 0.f ? -steep : steep, sgb = value_of(a.beta) < 0.f ? -steep : steep;
            ab.alpha = ab.alpha + sab * (-(sa * sa) * ea * sga);
            ab.beta = ab.beta + sbb * (-(sb * sb) * ebv * sgb);
        }
        T wb[3] = {T(0.f), T(0.f), T(0.f)};
        coeffs.vjp(P, a, x, u, Cb, ab, wb, ub);
#pragma unroll
        for (int k = 0; k < 3; ++k) xb[10 + k] = xb[10 + k] + wb[k];
        // airspeed, alpha, beta, qbar
        T vrb[3] = {ab.vr[0], ab.vr[1], ab.vr[2]};
        T vvb = (0.5f * 1.225f) * ab.qbar;
        const T tbar = ab.beta / m_sqrt(1.0f - tb * tb);          // beta = asin(t), t = vr1 / V
        vrb[1] = vrb[1] + tbar / a.V;
        const T Vb = -(tbar * tb) / a.V;
        const T iden = 1.0f / (ux * ux + a.vr[2] * a.vr[2]);    // alpha = atan2(vr2, ux)
        vrb[2] = vrb[2] + ab.alpha * ux * iden;
        vrb[0] = vrb[0] - ab.alpha * a.vr[2] * iden;
        vvb = vvb + Vb * (0.5f / a.V);                           // V = sqrt(vv + eps)
#pragma unroll
        for (int k = 0; k < 3; ++k) vrb[k] = vrb[k] + 2.0f * a.vr[k] * vvb;
        // r = q^-1 (v, 0) q:  v_bar += R r_bar,  q_bar += conj(q^-1) (x) (2 r_bar x r, 0)
        const Q4<T> vb = qmul(qmul_vec(q, vrb[0], vrb[1], vrb[2]), qi);
        xb[3] = xb[3] + vb.x; xb[4] = xb[4] + vb.y; xb[5] = xb[5] + vb.z;
        T e[3];
        cross3(vrb, r, e);
        const Q4<T> qb = qmul_vec(qconj(qi), 2.0f * e[0], 2.0f * e[1], 2.0f * e[2]);
        xb[6] = xb[6] + qb.x; xb[7] = xb[7] + qb.y; xb[8] = xb[8] + qb.z; xb[9] = xb[9] + qb.w;
    }
}

// ---- one RK4 step: gradient of lambda . F over (x, u, h) --------------------------------------------------------------------
// lam [13] (plain numbers: the multipliers do not depend on z).  Outputs: xo = F(x, u, h) and g_x [13], g_u [7], g_h.
// Where the three intermediate stage states (rows 3..12: the p rows never enter f) wait for the reverse sweep: registers
// (host build; small N) or this lane's column of an LDS array (StageLds: 30 (1 + N) floats per lane less at the point of
// highest pressure — what lets two directions per lane fit the register file).
template <class T> struct StageRegs {
    T xs[3][10];
    AC_DI void put(int s, const T* row3) {
        // (explicit slots: an array indexed by the loop counter would live in scratch memory)
        if (s == 0) { for (int i = 0; i < 10; ++i) xs[0][i] = row3[i]; }
        else if (s == 1) { for (int i = 0; i < 10; ++i) xs[1][i] = row3[i]; }
        else if (s == 2) { for (int i = 0; i < 10; ++i) xs[2][i] = row3[i]; }
    }
    AC_DI void get(int s, T* row3) const {
        if (s == 0) { for (int i = 0; i < 10; ++i) row3[i] = xs[0][i]; }
        else if (s == 1) { for (int i = 0; i < 10; ++i) row3[i] = xs[1][i]; }
        else { for (int i = 0; i < 10; ++i) row3[i] = xs[2][i]; }
    }
};
#ifndef AC_HOST_CHECK
template <int N> struct StageLds {
    float* base;  // this lane's float of word 0; words are `stride` floats apart
    int stride;
    AC_DI StageLds(float* lane_word, int stride_) : base(lane_word), stride(stride_) {}
    AC_DI void put(int s, const Dual<N>* row3) {
        float* p = base + (long)s * 10 * (N + 1) * stride;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            p[(i * (N + 1)) * stride] = row3[i].v;
#pragma unroll
            for (int j = 0; j < N; ++j) p[(i * (N + 1) + 1 + j) * stride] = row3[i].d[j];
        }
    }
    AC_DI void get(int s, Dual<N>* row3) const {
        const float* p = base + (long)s * 10 * (N + 1) * stride;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            row3[i].v = p[(i * (N + 1)) * stride];
#pragma unroll
            for (int j = 0; j < N; ++j) row3[i].d[j] = p[(i * (N + 1) + 1 + j) * stride];
        }
    }
};
#endif

template <class T, class Coeffs, class Store>
AC_DI void rk4_vjp(const DevParams& P, Coeffs& coeffs, const T x[13], const T u[7], const T& h, const float lam[13], T xo[13],
                   T gx[13], T gu[7], T& gh, Store& store) {
    // forward: stage states
    T acc[13], k[13];
    T xcur[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { xcur[i] = x[i]; acc[i] = T(0.f); }
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        coeffs.set_stage(s);
        state_derivative(P, coeffs, xcur, u, k);
        const float wsum = (s == 1 || s == 2) ? 2.0f : 1.0f;
        const float cnext = (s == 2) ? 1.0f : 0.5f;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            acc[i] = acc[i] + wsum * k[i];
            xcur[i] = x[i] + (cnext * h) * k[i];
        }
        if (s < 3) store.put(s, &xcur[3]);
    }
    const T h6 = h * (1.0f / 6.0f);
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = x[i] + h6 * acc[i];
    // adjoint of the final normalisation q <- q / |q|:  q_bar_in = (q_bar - qn (qn . q_bar)) / |q|
    T xb_out[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) xb_out[i] = T(lam[i]);
    if (P.p.normalise) {
        const T n = m_sqrt(xo[6] * xo[6] + xo[7] * xo[7] + xo[8] * xo[8] + xo[9] * xo[9]);
        const T inv = 1.0f / n;
        T qn[4], dot = T(0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) { qn[i] = xo[6 + i] * inv; dot = dot + qn[i] * lam[6 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { xb_out[6 + i] = (lam[6 + i] - qn[i] * dot) * inv; xo[6 + i] = qn[i]; }
    }
    // x+ = x + h/6 acc
#pragma unroll
    for (int i = 0; i < 13; ++i) gx[i] = xb_out[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) gu[i] = T(0.f);
    gh = T(0.f);
#pragma unroll
    for (int i = 0; i < 13; ++i) gh = gh + xb_out[i] * acc[i] * (1.0f / 6.0f);
    // reverse through the stages: kb = adjoint of the stage derivative k_s
    T kb[13], pend[13];  // pend = c xs_bar of the stage just left: h_bar += pend . k_{s-1}, available one iteration later
#pragma unroll
    for (int i = 0; i < 13; ++i) { kb[i] = h6 * xb_out[i]; pend[i] = T(0.f); }   // stage 4 enters the sum with weight 1
#pragma nounroll
    for (int s = 3; s >= 0; --s) {
        coeffs.set_stage(s);
        T xst[13], xsb[13], kk[13];
#pragma unroll
        for (int i = 0; i < 3; ++i) xst[i] = x[i];
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 10; ++i) xst[3 + i] = x[3 + i];
        } else {
            store.get(s - 1, &xst[3]);
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) xsb[i] = T(0.f);
        f_vjp(P, coeffs, xst, u, kb, kk, xsb, gu);   // kk = k_s (recomputed), xsb = (df/dx)' kb, gu += (df/du)' kb
#pragma unroll
        for (int i = 0; i < 13; ++i) { gh = gh + pend[i] * kk[i]; gx[i] = gx[i] + xsb[i]; }
        if (s > 0) {
            // xs_s = x + c h k_{s-1}:  k_{s-1}_bar = w_{s-1} h/6 x+_bar + c h xs_bar,  h_bar += c xs_bar . k_{s-1}
            const float c = (s == 3) ? 1.0f : 0.5f;                // x + h k3, x + h/2 k2, x + h/2 k1
            const float wprev = (s == 3 || s == 2) ? 2.0f : 1.0f;  // weights of k3, k2, k1 in the sum
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                pend[i] = c * xsb[i];
                kb[i] = (wprev * h6) * xb_out[i] + (c * h) * xsb[i];
            }
        }
    }
}

template <class T, class Coeffs>
AC_DI void rk4_vjp(const DevParams& P, Coeffs& coeffs, const T x[13], const T u[7], const T& h, const float lam[13], T xo[13],
                   T gx[13], T gu[7], T& gh) {
    StageRegs<T> store;
    rk4_vjp<T>(P, coeffs, x, u, h, lam, xo, gx, gu, gh, store);
}

}  // namespace ac

// ac_dynamics.hpp — 6-DoF rigid-body derivative and RK4 update, templated on the scalar type
// (float or Dual<N>) and on a coefficient provider (analytic models here; the MLP provider lives
// in ac_mlp.hpp).  fp32 restatement of the reference arithmetic:
//   v_frd_rel/airspeed/alpha/beta/qbar   dynamics/base.py:147-177, 229-241
//   effective surface angles (poly)      dynamics/aircraft.py:189-233
//   coefficients + flaps/stall           dynamics/aircraft.py:255-307, coefficient_models.py:41-133
//   forces/moments                       dynamics/aircraft.py:309-330, dynamics/base.py:253-288
//   state_derivative / RK4 / sub-steps   dynamics/base.py:290-480
#pragma once
#include <type_traits>
#include <utility>

#include "ac_math.hpp"
#include "../../include/aircraft_hip.h"

namespace ac {

// Everything a kernel needs, passed BY VALUE as the kernel argument (kernarg segment -> scalar
// loads; all indices into it are compile-time after unrolling, so no VGPR is spent on constants).
struct DevParams {
    ac_params p;
    float linear_W[36];
    float poly_coef[6 * 34];
    float poly_intercept[6];
    float mlp_in_mean[5], mlp_in_std[5], mlp_out_mean[6], mlp_out_std[6];
    // mlp_out_std[k] / mlp_in_std[j], rounded once on the host (IEEE single division, what the device computes too): the
    // chain rule dC_k = sum_j J[k][j] * jscale[k][j] * d(in_j) reads them as scalar operands instead of holding thirty
    // wave-uniform quotients in vector registers for the life of the kernel
    float mlp_jscale[6][5];
};

constexpr float kDeg = 0.017453292519943295f;  // pi/180

template <class T> struct AeroPre {
    T vr[3], V, alpha, beta, qbar;
};

template <class T> AC_DI void aero_pre(const DevParams& P, const T x[13], AeroPre<T>& a) {
    const float eps = P.p.epsilon;
    const Q4<T> q{x[6], x[7], x[8], x[9]};
    const Q4<T> r = qmul(qmul_vec(qinv(q), x[3], x[4], x[5]), q);
    a.vr[0] = r.x + eps; a.vr[1] = r.y + eps; a.vr[2] = r.z + eps;
    const T vv = a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2];
    a.V = m_sqrt(vv + eps);
    a.alpha = m_atan2(a.vr[2], a.vr[0] + eps);
    a.beta = m_asin(a.vr[1] / a.V);
    a.qbar = (0.5f * 1.225f) * vv;
}

// ---- analytic coefficient models -----------------------------------------------------------
// Cubic fits over f[0..3] = (alpha, beta, aileron, elevator) with the 34 monomials of sklearn
// PolynomialFeatures(3, include_bias=False): combinations_with_replacement(range(4), d), d = 1, 2, 3.
// The monomials are STREAMED, never stored: nested loops i <= j <= k visit the degree-2 terms (index 4 + ...) and the
// degree-3 terms (index 14 + ...) in exactly sklearn's lexicographic order, each degree-2 product is built once and
// extended to its degree-3 terms, and every monomial is accumulated into the NOUT requested fits at once.  (Holding
// the 34 monomials as duals costs 170 registers per evaluation point and made the sensitivity kernel spill 1 KB/lane.)
// (For plain floats the monomials are cheap to hold — 34 registers — and building them first leaves the compiler a
// shorter dependent chain: the forward kernels keep that form.)
template <int NOUT>
AC_DI void poly_eval(const DevParams& P, const int (&ks)[NOUT], const float f[4], float out[NOUT]) {
    float m[34];
    int t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) m[t++] = f[i];
    float m2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) { m2[i][j] = f[i] * f[j]; m[t++] = m2[i][j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j)
#pragma unroll
            for (int k = j; k < 4; ++k) m[t++] = m2[i][j] * f[k];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float acc = P.poly_intercept[ks[o]];
#pragma unroll
        for (int q = 0; q < 34; ++q) acc = acc + P.poly_coef[ks[o] * 34 + q] * m[q];
        out[o] = acc;
    }
}
template <int NOUT, class T>
AC_DI void poly_eval(const DevParams& P, const int (&ks)[NOUT], const T f[4], T out[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) out[o] = T(P.poly_intercept[ks[o]]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[o] = out[o] + P.poly_coef[ks[o] * 34 + i] * f[i];
    int t2 = 4, t3 = 14;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            const T m2 = f[i] * f[j];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) out[o] = out[o] + P.poly_coef[ks[o] * 34 + t2] * m2;
            ++t2;
#pragma unroll
            for (int k = j; k < 4; ++k) {
                const T m3 = m2 * f[k];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) out[o] = out[o] + P.poly_coef[ks[o] * 34 + t3] * m3;
                ++t3;
            }
        }
}
// P_CZ(alpha, 0, 0, 0): only the pure-alpha monomials survive (terms 0, 4, 14)
template <class T> AC_DI T poly_cz_alpha_only(const DevParams& P, const T& al) {
    const T a2 = al * al;
    return T(P.poly_intercept[2]) + P.poly_coef[2 * 34 + 0] * al + P.poly_coef[2 * 34 + 4] * a2 +
           P.poly_coef[2 * 34 + 14] * (a2 * al);
}

// Coefficient-provider protocol:
//   prefetch(P, x, uv)      called on the stage state BEFORE anything else of the stage is computed; the MLP
//                           provider runs the whole network here from primal inputs, so that nothing but the
//                           RK4 carry is live across the (register-hungry) network evaluation
//   operator()(P, a, x, u, C)   turns the aerodynamic inputs (with tangents) into the six coefficients
template <int MODEL> struct AnalyticCoeffs {
    static constexpr int kModel = MODEL;
    template <class T> AC_DI void prefetch(const DevParams&, const T*, const float*) {}
    template <class T>
    AC_DI void operator()(const DevParams& P, const AeroPre<T>& a, const T x[13], const T u[7], T C[6]) const {
        const T* w = &x[10];
        const T da = u[0], de = u[1], dr = u[2];
        if constexpr (MODEL == AC_MODEL_LINEAR) {
            const T in[5] = {a.qbar, a.alpha, a.beta, da, de};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                T s = P.linear_W[k * 6 + 0] * in[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) s = s + P.linear_W[k * 6 + j] * in[j];
                C[k] = s + P.linear_W[k * 6 + 5];
            }
            C[5] = C[5] + (-0.1f * 6.0f * kDeg) * dr;
        } else if constexpr (MODEL == AC_MODEL_POLY) {
            const float eps = P.p.epsilon, arm = P.p.rudder_moment_arm, b4 = P.p.b * 0.25f;
            const T ux = a.vr[0] + eps;
            const T alpha_e = m_atan2(a.vr[2] + arm * w[1], ux);
            const T alpha_l = m_atan2(a.vr[2] - b4 * w[0], ux);
            const T alpha_r = m_atan2(a.vr[2] + b4 * w[0], ux);
            const T vy = a.vr[1] - arm * w[2];
            const T beta_r = m_asin(vy / m_sqrt(a.vr[0] * a.vr[0] + vy * vy + a.vr[2] * a.vr[2] + eps));
            {
                const T f[4] = {a.alpha, a.beta, da, de};
                const int ks[4] = {0, 1, 2, 3};
                T r4[4];
                poly_eval<4>(P, ks, f, r4);
#pragma unroll
                for (int k = 0; k < 4; ++k) C[k] = r4[k];
            }
            C[3] = C[3] + (b4 * 0.5f) * (poly_cz_alpha_only(P, alpha_r) - poly_cz_alpha_only(P, alpha_l));
            {
                const T f[4] = {alpha_e, a.beta, da, de};
                const int ks[1] = {4};
                T r1[1];
                poly_eval<1>(P, ks, f, r1);
                C[4] = r1[0];
            }
            {
                const T f[4] = {a.alpha, beta_r, da, de};
                const int ks[1] = {5};
                T r1[1];
                poly_eval<1>(P, ks, f, r1);
                C[5] = r1[0] + (0.01f * 6.0f * kDeg) * dr;
            }
        } else {  // DefaultModel
            C[0] = -(0.02f + 0.3f * (a.alpha * a.alpha));
            C[1] = -0.98f * a.beta;
            C[2] = -(5.0f * a.alpha);
            C[3] = (0.08f * 4.0f * kDeg) * da + (-0.05f) * w[0];
            C[4] = (-1.2f * 5.0f * kDeg) * de + (-0.5f) * w[1];
            C[5] = (-0.1f * 6.0f * kDeg) * dr + (-0.05f) * w[2];
        }
    }
};

// ---- forces, moments and the state derivative -----------------------------------------------
template <class T> struct AeroPost {
    T C[6], F[3], M[3];
};

template <class T>
AC_DI void aero_post(const DevParams& P, const AeroPre<T>& a, const T u[7], T C[6], AeroPost<T>& o) {
    if (P.p.stall_scaling) {  // uniform branch; dynamics/aircraft.py:280-294
        const float lim = 30.0f * kDeg, steep = 10.0f;
        const T sa = 1.0f / (1.0f + m_exp(steep * (m_fabs(a.alpha) - lim)));
        const T sb = 1.0f / (1.0f + m_exp(steep * (m_fabs(a.beta) - lim)));
        C[2] = C[2] * sa; C[2] = C[2] * sb; C[4] = C[4] * sa;
    }
    C[0] = C[0] + (-0.1f) * u[6];
    C[2] = C[2] + (-0.6f) * u[6];
    const T qS = a.qbar * P.p.S;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.C[k] = C[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o.F[k] = C[k] * qS;
    o.F[0] = o.F[0] * sign_of(value_of(a.vr[0]));
    const T Ma0 = C[3] * qS * P.p.b, Ma1 = C[4] * qS * P.p.c, Ma2 = C[5] * qS * P.p.b;
    o.M[0] = Ma0 + (P.p.com[1] * o.F[2] - P.p.com[2] * o.F[1]);
    o.M[1] = Ma1 + (P.p.com[2] * o.F[0] - P.p.com[0] * o.F[2]);
    o.M[2] = Ma2 + (P.p.com[0] * o.F[1] - P.p.com[1] * o.F[0]);
}

// Quadrotor plugin (dynamics/quadrotor.py:43-54): forces and moments straight from the four rotor thrusts u[0..3];
// the moment about the reference point picks up com x F like every SixDOF (dynamics/base.py:268-278).
template <class T> AC_DI void quad_forces(const DevParams& P, const T u[7], AeroPost<T>& o) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.C[k] = T(0.f);
    o.F[0] = T(0.f); o.F[1] = T(0.f);
    o.F[2] = u[0] + u[1] + u[2] + u[3];
    const T Ma0 = u[0] - u[1] - u[2] + u[3];
    const T Ma1 = u[2] + u[3] - u[0] - u[1];
    const T Ma2 = 0.5f * (u[0] - u[1] + u[2] - u[3]);
    o.M[0] = Ma0 + (P.p.com[1] * o.F[2] - P.p.com[2] * o.F[1]);
    o.M[1] = Ma1 + (P.p.com[2] * o.F[0] - P.p.com[0] * o.F[2]);
    o.M[2] = Ma2 + (P.p.com[0] * o.F[1] - P.p.com[1] * o.F[0]);
}

// Euler angles of the attitude quaternion (dynamics/base.py:179-195)
AC_DI void euler_angles(const float x[13], float& phi, float& theta, float& psi) {
    const float qx = x[6], qy = x[7], qz = x[8], qw = x[9];
    phi = atan2f(2.f * (qw * qx + qy * qz), 1.f - 2.f * (qx * qx + qy * qy));
    theta = asinf(2.f * (qw * qy - qz * qx));
    psi = atan2f(2.f * (qw * qz + qx * qy), 1.f - 2.f * (qy * qy + qz * qz));
}

template <class T> AC_DI void rigid_body(const DevParams& P, const T x[13], const AeroPost<T>& o, T xd[13]) {
    const Q4<T> q{x[6], x[7], x[8], x[9]};
    const T* w = &x[10];
    const Q4<T> Fn = qmul(qmul_vec(q, o.F[0], o.F[1], o.F[2]), qinv(q));
    const float im = 1.0f / P.p.mass;
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    xd[3] = Fn.x * im + P.p.gravity[0];
    xd[4] = Fn.y * im + P.p.gravity[1];
    xd[5] = Fn.z * im + P.p.gravity[2];
    const Q4<T> hq{0.5f * q.x, 0.5f * q.y, 0.5f * q.z, 0.5f * q.w};
    const Q4<T> qd = qmul_vec(hq, w[0], w[1], w[2]);
    xd[6] = qd.x; xd[7] = qd.y; xd[8] = qd.z; xd[9] = qd.w;
    const float* I = P.p.inertia;
    const float* Ii = P.p.inertia_inv;
    T Iw[3], rhs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Iw[i] = I[3 * i] * w[0] + I[3 * i + 1] * w[1] + I[3 * i + 2] * w[2];
    rhs[0] = o.M[0] - (w[1] * Iw[2] - w[2] * Iw[1]);
    rhs[1] = o.M[1] - (w[2] * Iw[0] - w[0] * Iw[2]);
    rhs[2] = o.M[2] - (w[0] * Iw[1] - w[1] * Iw[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[10 + i] = Ii[3 * i] * rhs[0] + Ii[3 * i + 1] * rhs[1] + Ii[3 * i + 2] * rhs[2];
}

// A coefficient provider whose prefetch() has already formed the aerodynamic quantities of this x hands them over through
// kept_aero(a) (MlpLazyCoeffs, ac_mlp_valu.hpp); every other provider has no such member and f forms them itself.
template <class C, class A, class = void> struct keeps_aero : std::false_type {};
template <class C, class A>
struct keeps_aero<C, A, std::void_t<decltype(std::declval<C&>().kept_aero(std::declval<A&>()))>> : std::true_type {};

// x_dot = f(x, u).  coeffs.prefetch(P, x, u-values) must have been called for this x.
template <class T, class Coeffs>
AC_DI void state_derivative(const DevParams& P, Coeffs& coeffs, const T x[13], const T u[7], T xd[13]) {
    AeroPost<T> o;
    if constexpr (Coeffs::kModel == AC_MODEL_QUAD) {
        quad_forces(P, u, o);
    } else {
        AeroPre<T> a;
        if constexpr (keeps_aero<Coeffs, AeroPre<T>>::value) coeffs.kept_aero(a);  // formed by prefetch() already
        else aero_pre(P, x, a);
        T C[6];
        coeffs(P, a, x, u, C);
        aero_post(P, a, u, C, o);
    }
    rigid_body(P, x, o, xd);
}

template <class T> AC_DI void normalise_q(T x[13]) {
    const T n = m_sqrt(x[6] * x[6] + x[7] * x[7] + x[8] * x[8] + x[9] * x[9]);
    const T inv = 1.0f / n;
#pragma unroll
    for (int i = 6; i < 10; ++i) x[i] = x[i] * inv;
}

// One classic RK4 step, control held (dynamics/base.py:408-446): dx = h/6 (k1 + 2 k2 + 2 k3 + k4).
// The four stages run as a rolled loop so the (possibly very large) coefficient-provider code is
// instantiated once.
template <class Coeffs>
AC_DI void rk4_increment(const DevParams& P, Coeffs& coeffs, const float x[13], const float u[7], float h,
                         float dx[13]) {
    float acc[13], xs[13], k[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { xs[i] = x[i]; acc[i] = 0.f; }
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        coeffs.prefetch(P, xs, u);
        state_derivative(P, coeffs, xs, u, k);
        const float wsum = (s == 1 || s == 2) ? 2.0f : 1.0f;  // k1 + 2 k2 + 2 k3 + k4
        const float cnext = (s == 2) ? 1.0f : 0.5f;           // x + h/2 k1, x + h/2 k2, x + h k3
        const float hs = h * cnext;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            acc[i] = fmaf(wsum, k[i], acc[i]);
            xs[i] = fmaf(hs, k[i], x[i]);
        }
    }
    const float h6 = h * (1.0f / 6.0f);
#pragma unroll
    for (int i = 0; i < 13; ++i) dx[i] = h6 * acc[i];
}

// state_update on a float64 CARRY: `substeps` RK4 steps of dt/substeps, quaternion normalised once at the end
// (dynamics/base.py:450-480).  All arithmetic of f is fp32; only  x <- x + dx  and the final normalisation run in
// float64.  For one step this is bit-identical to fp32 (x + dx rounds once either way); over chained steps
// (sub-steps, rollouts) it removes the 0.5-ulp-per-step random walk of re-rounding the state, which is what
// limits an all-fp32 50-step rollout to ~4e-5 relative against the float64 reference.
template <class Coeffs>
AC_DI void state_update_carry(const DevParams& P, Coeffs& coeffs, double xa[13], const float u[7], float dt) {
    const int ns = P.p.substeps < 1 ? 1 : P.p.substeps;
    const float h = (ns == 1) ? dt : dt / (float)ns;
#pragma nounroll
    for (int s = 0; s < ns; ++s) {
        float xf[13], dx[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xf[i] = (float)xa[i];
        rk4_increment(P, coeffs, xf, u, h, dx);
#pragma unroll
        for (int i = 0; i < 13; ++i) xa[i] += (double)dx[i];
    }
    if (P.p.normalise) {
        const double n2 = xa[6] * xa[6] + xa[7] * xa[7] + xa[8] * xa[8] + xa[9] * xa[9];
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int i = 6; i < 10; ++i) xa[i] *= inv;
    }
}

template <class Coeffs>
AC_DI void state_update(const DevParams& P, Coeffs& coeffs, float x[13], const float u[7], float dt) {
    double xa[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) xa[i] = (double)x[i];
    state_update_carry(P, coeffs, xa, u, dt);
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = (float)xa[i];
}

// ---- sensitivities: four lanes per unit, Dual<4> ------------------------------------------------
// Direction d = 4*g + j (g = lane>>4 within the wave, j = 0..3):
//   0-2 v   3-6 q   7-9 omega   10 aileron  11 elevator  12 rudder  13 flaps  14 dt  15 (unused)
// dF/dp = [I;0] and dF/dthrust = 0 exactly (the reference's force model ignores both), so those six
// columns are constants, not propagated.
//
// The inputs' tangents are 0/1 SEEDS, so they are never stored: the step keeps only the primal x0, u
// (20 registers) and rebuilds the seed pattern from the lane's group index where it is needed.  That is
// 80 registers per lane less than carrying Dual x0 and u through the four stages.
// N = tangent directions per lane (16 / N lanes per unit): N = 4 for the MLP kernels (the four lanes are the four row
// groups of the unit's MFMA column), N = 2 (eight lanes per unit) for the analytic models, whose dual arithmetic then
// fits half the registers and leaves room for two waves per SIMD.
template <int N> struct SeedsT {
    static AC_DI Dual<N> state(int g, int i, float v) {  // x0[i] as a dual
        Dual<N> r; r.v = v;
#pragma unroll
        for (int j = 0; j < N; ++j) r.d[j] = (i >= 3 && (i - 3) == N * g + j) ? 1.f : 0.f;
        return r;
    }
    // control directions 10..13: aircraft = aileron, elevator, rudder, flaps (rows 0, 1, 2, 6; thrust rows have no
    // effect); quadrotor = its four thrusts (rows 0..3)
    template <bool QUAD> static AC_DI void controls(int g, const float uv[7], Dual<N> u[7]) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            u[i].v = uv[i];
            const int dir = QUAD ? (i < 4 ? 10 + i : -1) : ((i < 3) ? 10 + i : (i == 6 ? 13 : -1));
#pragma unroll
            for (int j = 0; j < N; ++j) u[i].d[j] = (dir == N * g + j) ? 1.f : 0.f;
        }
    }
    static AC_DI Dual<N> step(int g, float hv, float dh_ddt) {  // h = dt/substeps as a dual in the dt direction
        Dual<N> r; r.v = hv;
#pragma unroll
        for (int j = 0; j < N; ++j) r.d[j] = (14 == N * g + j) ? dh_ddt : 0.f;
        return r;
    }
};
typedef SeedsT<4> Seeds;

// One RK4 step from primal inputs; xo = F(x0, u, h) with tangents w.r.t. this lane's N directions.
template <int N, class Coeffs>
AC_DI void rk4_step_seeded(const DevParams& P, Coeffs& coeffs, int g, const float xv[13], const float uv[7],
                           float hv, float dh_ddt, Dual<N> xo[13]) {
    typedef Dual<N> T;
    typedef SeedsT<N> Seeds;
    T acc[13], xs[13], k[13];
    {
        int g0 = g;
        asm volatile("" : "+v"(g0));  // (likewise: not hoisted out of an enclosing sub-step / unit-group loop)
#pragma unroll
        for (int i = 0; i < 13; ++i) { xs[i] = Seeds::state(g0, i, xv[i]); acc[i] = T(0.f); }
    }
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        coeffs.prefetch(P, xs, uv);
        int gg = g;
        asm volatile("" : "+v"(gg));  // keep the seed patterns out of loop-invariant registers
        {
            T u[7];
            Seeds::template controls<Coeffs::kModel == AC_MODEL_QUAD>(gg, uv, u);
            state_derivative(P, coeffs, xs, u, k);
        }
        const float wsum = (s == 1 || s == 2) ? 2.0f : 1.0f;
        const float cnext = (s == 2) ? 1.0f : 0.5f;
        const T hs = Seeds::step(gg, hv * cnext, dh_ddt * cnext);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            acc[i] = acc[i] + wsum * k[i];
            xs[i] = Seeds::state(gg, i, xv[i]) + hs * k[i];
        }
    }
    int ge = g;
    asm volatile("" : "+v"(ge));  // the seeds of the final combination are rebuilt here, not carried across the four stages
    const T h6 = Seeds::step(ge, hv * (1.0f / 6.0f), dh_ddt * (1.0f / 6.0f));
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = Seeds::state(ge, i, xv[i]) + h6 * acc[i];
}

}  // namespace ac

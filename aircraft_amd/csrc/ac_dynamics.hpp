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
    // d(fit k)/d(f_v) as a quadratic in f: [k][v][15] over the basis 1, f_0..f_3, f_a f_b (a <= b, sklearn order), derived
    // from poly_coef on the host (poly_gradient_tables): the sensitivity kernels evaluate value and gradient of the fits on
    // the primal and chain the tangents through the gradient instead of pushing duals through 34 monomials
    float poly_grad[6 * 4 * 15];
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

// ---- structured first-order tangents of the frame changes ------------------------------------------------------------
// A dual pushed through the two quaternion products of a frame change costs ~3 instructions per primal multiply AND
// direction (~95 per direction for v_frd_rel, as many again for forces_ned).  The products are rotations, so their
// tangents have closed forms whose coefficients are PRIMAL values, shared by every direction a lane carries:
//     r  = q^-1 (v,0) q       dr  = M dv  + 2 r x (q^-1 dq)_vec          M  = matrix of  v -> q^-1 v q
//     Fn = q (F,0) q^-1       dFn = M' dF + 2 (dq q^-1)_vec x Fn
// exact for non-unit q too (q^-1 = conj(q)/|q|^2 is the true inverse, d(q^-1) = -q^-1 dq q^-1): 27 / 30 instructions per
// direction.  The primal values are formed by the same float expressions as the forward kernels.
struct Rot3 { float m[3][3]; };
// M of  r = q^-1 (v,0) q = s [ (w^2 - a.a) v + 2 a (a.v) - 2 w (a x v) ],  a = q_vec, s = 1/|q|^2  (s a = -qi_vec, s w = qi.w)
AC_DI Rot3 rot_inverse(const Q4<float>& q, const Q4<float>& qi) {
    const float sx = -qi.x, sy = -qi.y, sz = -qi.z, sw = qi.w;
    const float d = sw * q.w - (sx * q.x + sy * q.y + sz * q.z);
    const float sx2 = sx + sx, sy2 = sy + sy, sz2 = sz + sz, sw2 = sw + sw;
    Rot3 R;
    R.m[0][0] = fmaf(sx2, q.x, d);
    R.m[1][1] = fmaf(sy2, q.y, d);
    R.m[2][2] = fmaf(sz2, q.z, d);
    R.m[0][1] = fmaf(sx2, q.y, sw2 * q.z);  R.m[1][0] = fmaf(sx2, q.y, -(sw2 * q.z));
    R.m[0][2] = fmaf(sx2, q.z, -(sw2 * q.y)); R.m[2][0] = fmaf(sx2, q.z, sw2 * q.y);
    R.m[1][2] = fmaf(sy2, q.z, sw2 * q.x);  R.m[2][1] = fmaf(sy2, q.z, -(sw2 * q.x));
    return R;
}
// vector parts of  qi (x) dq  (sign = +1) and of  dq (x) qi  (sign = -1): bw dq_v + dq_w b +- b x dq_v
template <int SIGN>
AC_DI void qinv_times_dq(const Q4<float>& qi, float dx, float dy, float dz, float dw, float e[3]) {
    const float tx = fmaf(qi.w, dx, dw * qi.x), ty = fmaf(qi.w, dy, dw * qi.y), tz = fmaf(qi.w, dz, dw * qi.z);
    const float cx = fmaf(qi.y, dz, -(qi.z * dy)), cy = fmaf(qi.z, dx, -(qi.x * dz)), cz = fmaf(qi.x, dy, -(qi.y * dx));
    if (SIGN > 0) { e[0] = tx + cx; e[1] = ty + cy; e[2] = tz + cz; }
    else { e[0] = tx - cx; e[1] = ty - cy; e[2] = tz - cz; }
}

template <int N> AC_DI void aero_pre(const DevParams& P, const Dual<N> x[13], AeroPre<Dual<N>>& a) {
    const float eps = P.p.epsilon;
    const Q4<float> q{x[6].v, x[7].v, x[8].v, x[9].v};
    const Q4<float> qi = qinv(q);
    const Q4<float> r = qmul(qmul_vec(qi, x[3].v, x[4].v, x[5].v), q);
    const Rot3 R = rot_inverse(q, qi);
    const float r2[3] = {r.x + r.x, r.y + r.y, r.z + r.z};
    a.vr[0].v = r.x + eps; a.vr[1].v = r.y + eps; a.vr[2].v = r.z + eps;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float e[3];
        qinv_times_dq<+1>(qi, x[6].d[j], x[7].d[j], x[8].d[j], x[9].d[j], e);
        const float dv0 = x[3].d[j], dv1 = x[4].d[j], dv2 = x[5].d[j];
        a.vr[0].d[j] = fmaf(R.m[0][0], dv0, fmaf(R.m[0][1], dv1, fmaf(R.m[0][2], dv2, fmaf(r2[1], e[2], -(r2[2] * e[1])))));
        a.vr[1].d[j] = fmaf(R.m[1][0], dv0, fmaf(R.m[1][1], dv1, fmaf(R.m[1][2], dv2, fmaf(r2[2], e[0], -(r2[0] * e[2])))));
        a.vr[2].d[j] = fmaf(R.m[2][0], dv0, fmaf(R.m[2][1], dv1, fmaf(R.m[2][2], dv2, fmaf(r2[0], e[1], -(r2[1] * e[0])))));
    }
    // airspeed, alpha, beta, qbar: tangent coefficients formed once on the primal, two or three instructions per direction
    const float v0 = a.vr[0].v, v1 = a.vr[1].v, v2 = a.vr[2].v;
    const float vv = v0 * v0 + v1 * v1 + v2 * v2;
    const float V = sqrtf(vv + eps);
    const float ux = v0 + eps;
    a.V.v = V;
    a.alpha.v = atan2f(v2, ux);
    const float t = v1 / V;
    a.beta.v = asinf(t);
    a.qbar.v = (0.5f * 1.225f) * vv;
    const float den = 1.0f / fmaf(ux, ux, v2 * v2);
    const float ca_y = ux * den, ca_x = -(v2 * den);            // d alpha = ca_y d v2 + ca_x d v0
    const float gV = 0.5f / V;                                   // d V = gV d vv
    const float gb = (1.0f / V) / sqrtf(fmaf(-t, t, 1.0f));      // d beta = gb (d v1 - t d V)
    const float gbt = -(gb * t);
    const float w0 = v0 + v0, w1 = v1 + v1, w2 = v2 + v2;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float d0 = a.vr[0].d[j], d1 = a.vr[1].d[j], d2 = a.vr[2].d[j];
        const float dvv = fmaf(w0, d0, fmaf(w1, d1, w2 * d2));
        const float dV = gV * dvv;
        a.V.d[j] = dV;
        a.alpha.d[j] = fmaf(ca_y, d2, ca_x * d0);
        a.beta.d[j] = fmaf(gb, d1, gbt * dV);
        a.qbar.d[j] = (0.5f * 1.225f) * dvv;
    }
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

// Host side of DevParams::poly_grad (ac_set_poly; tests/host_dyn): every monomial of sklearn's degree-<=3 basis over four
// features, differentiated with respect to f_v, lands on one element of the degree-<=2 basis with its multiplicity.
inline void poly_gradient_tables(const float* coef /*[6][34]*/, float* grad /*[6][4][15]*/) {
    auto basis2 = [](int a, int b) {  // index of f_a f_b (a <= b) in [1, f_0..f_3, pairs...]
        int idx = 5;
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) { if (i == a && j == b) return idx; ++idx; }
        return -1;
    };
    for (int k = 0; k < 6; ++k) {
        double g[4][15] = {};
        int t = 0;
        for (int i = 0; i < 4; ++i) g[i][0] += (double)coef[k * 34 + t++];
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) {
                const double c = coef[k * 34 + t++];
                g[i][1 + j] += c;  // d(f_i f_j)/df_i = f_j  (twice when i == j)
                g[j][1 + i] += c;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j)
                for (int l = j; l < 4; ++l) {
                    const double c = coef[k * 34 + t++];
                    g[i][basis2(j, l)] += c;
                    g[j][basis2(i, l)] += c;
                    g[l][basis2(i, j)] += c;
                }
        for (int v = 0; v < 4; ++v)
            for (int q = 0; q < 15; ++q) grad[(k * 4 + v) * 15 + q] = (float)g[v][q];
    }
}

// Value and gradient of the fits ks[0..NOUT) at the primal point f: 30 shared monomial products, 34 + 4 x 14 fused
// multiply-adds per fit, every coefficient a scalar operand.
template <int NOUT>
AC_DI void poly_value_grad(const DevParams& P, const int (&ks)[NOUT], const float f[4], float val[NOUT], float grad[NOUT][4]) {
    float m2[10], m3[20];
    int t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) m2[t++] = f[i] * f[j];
    t = 0;
    int t2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
#pragma unroll
            for (int k = j; k < 4; ++k) m3[t++] = m2[t2] * f[k];
            ++t2;
        }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float* c = &P.poly_coef[ks[o] * 34];
        float acc = P.poly_intercept[ks[o]];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(c[q], f[q], acc);
#pragma unroll
        for (int q = 0; q < 10; ++q) acc = fmaf(c[4 + q], m2[q], acc);
#pragma unroll
        for (int q = 0; q < 20; ++q) acc = fmaf(c[14 + q], m3[q], acc);
        val[o] = acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float* g = &P.poly_grad[(ks[o] * 4 + v) * 15];
            float ga = g[0];
#pragma unroll
            for (int q = 0; q < 4; ++q) ga = fmaf(g[1 + q], f[q], ga);
#pragma unroll
            for (int q = 0; q < 10; ++q) ga = fmaf(g[5 + q], m2[q], ga);
            grad[o][v] = ga;
        }
    }
}

// Coefficient-provider protocol:
//   prefetch(P, x, uv)      called on the stage state BEFORE anything else of the stage is computed; the MLP
//                           provider runs the whole network here from primal inputs, so that nothing but the
//                           RK4 carry is live across the (register-hungry) network evaluation
//   operator()(P, a, x, u, C)   turns the aerodynamic inputs (with tangents) into the six coefficients
template <int MODEL> struct AnalyticCoeffs {
    static constexpr int kModel = MODEL;
    template <class T> AC_DI void prefetch(const DevParams&, const T*, const float*) {}
    // First-order duals: the model's partial derivatives are formed once on the primal (closed forms; for the cubic fits
    // value and gradient from the host-derived tables) and every direction is one short chain-rule row.
    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7],
                          Dual<N> C[6]) const {
        typedef Dual<N> T;
        const T* w = &x[10];
        const T &da = u[0], &de = u[1], &dr = u[2];
        if constexpr (MODEL == AC_MODEL_LINEAR) {
            const T* in[5] = {&a.qbar, &a.alpha, &a.beta, &da, &de};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                T s = P.linear_W[k * 6 + 0] * (*in[0]);
#pragma unroll
                for (int j = 1; j < 5; ++j) s = dual_axpy(P.linear_W[k * 6 + j], *in[j], s);
                C[k] = s + P.linear_W[k * 6 + 5];
            }
            C[5] = dual_axpy(-0.1f * 6.0f * kDeg, dr, C[5]);
        } else if constexpr (MODEL == AC_MODEL_POLY) {
            const float eps = P.p.epsilon, arm = P.p.rudder_moment_arm, b4 = P.p.b * 0.25f;
            const float v0 = a.vr[0].v, v1 = a.vr[1].v, v2 = a.vr[2].v;
            const float ux = v0 + eps;
            // effective angles (aircraft.py:189-233) with the coefficients of their differentials
            const float ye = fmaf(arm, w[1].v, v2), yl = fmaf(-b4, w[0].v, v2), yr = fmaf(b4, w[0].v, v2);
            const float alpha_e = atan2f(ye, ux), alpha_l = atan2f(yl, ux), alpha_r = atan2f(yr, ux);
            const float de_ = 1.0f / fmaf(ux, ux, ye * ye), dl_ = 1.0f / fmaf(ux, ux, yl * yl), dr_ = 1.0f / fmaf(ux, ux, yr * yr);
            const float vy = fmaf(-arm, w[2].v, v1);
            const float nb = sqrtf(v0 * v0 + vy * vy + v2 * v2 + eps);
            const float tb = vy / nb;
            const float beta_r = asinf(tb);
            const float gb = (1.0f / nb) / sqrtf(fmaf(-tb, tb, 1.0f));
            const float kb = gb * tb / nb;  // d beta_r = gb d vy - kb (v0 d v0 + vy d vy + v2 d v2)
            float val4[4], g4[4][4], vale[1], ge[1][4], valr[1], gr[1][4];
            {
                const float f[4] = {a.alpha.v, a.beta.v, da.v, de.v};
                const int ks[4] = {0, 1, 2, 3};
                poly_value_grad<4>(P, ks, f, val4, g4);
            }
            {
                const float f[4] = {alpha_e, a.beta.v, da.v, de.v};
                const int ks[1] = {4};
                poly_value_grad<1>(P, ks, f, vale, ge);
            }
            {
                const float f[4] = {a.alpha.v, beta_r, da.v, de.v};
                const int ks[1] = {5};
                poly_value_grad<1>(P, ks, f, valr, gr);
            }
            // P_CZ(alpha_x, 0, 0, 0) and its slope
            const float z0 = P.poly_intercept[2], z1 = P.poly_coef[2 * 34 + 0], z2 = P.poly_coef[2 * 34 + 4], z3 = P.poly_coef[2 * 34 + 14];
            const float czr = fmaf(fmaf(fmaf(z3, alpha_r, z2), alpha_r, z1), alpha_r, z0);
            const float czl = fmaf(fmaf(fmaf(z3, alpha_l, z2), alpha_l, z1), alpha_l, z0);
            const float hb = b4 * 0.5f;
            const float sr = hb * fmaf(fmaf(3.0f * z3, alpha_r, z2 + z2), alpha_r, z1);
            const float sl = -(hb * fmaf(fmaf(3.0f * z3, alpha_l, z2 + z2), alpha_l, z1));
#pragma unroll
            for (int k = 0; k < 4; ++k) C[k].v = val4[k];
            C[3].v = fmaf(hb, czr - czl, C[3].v);
            C[4].v = vale[0];
            C[5].v = fmaf(0.01f * 6.0f * kDeg, dr.v, valr[0]);
            // d alpha_x = cx_y d y_x + cx_x d v0
            const float cey = ux * de_, cex = -(ye * de_), cly = ux * dl_, clx = -(yl * dl_), cry = ux * dr_, crx = -(yr * dr_);
            const float kb0 = -(kb * v0), kb2 = -(kb * v2), kby = fmaf(-kb, vy, gb);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float d0 = a.vr[0].d[j], d1 = a.vr[1].d[j], d2 = a.vr[2].d[j];
                const float dal = a.alpha.d[j], dbe = a.beta.d[j], dda = da.d[j], dde = de.d[j];
                const float dae = fmaf(cey, fmaf(arm, w[1].d[j], d2), cex * d0);
                const float dalp = fmaf(cly, fmaf(-b4, w[0].d[j], d2), clx * d0);
                const float darp = fmaf(cry, fmaf(b4, w[0].d[j], d2), crx * d0);
                const float dbr = fmaf(kby, fmaf(-arm, w[2].d[j], d1), fmaf(kb0, d0, kb2 * d2));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    C[k].d[j] = fmaf(g4[k][0], dal, fmaf(g4[k][1], dbe, fmaf(g4[k][2], dda, g4[k][3] * dde)));
                C[3].d[j] = fmaf(sr, darp, fmaf(sl, dalp, C[3].d[j]));
                C[4].d[j] = fmaf(ge[0][0], dae, fmaf(ge[0][1], dbe, fmaf(ge[0][2], dda, ge[0][3] * dde)));
                C[5].d[j] = fmaf(gr[0][0], dal, fmaf(gr[0][1], dbr, fmaf(gr[0][2], dda, fmaf(gr[0][3], dde, (0.01f * 6.0f * kDeg) * dr.d[j]))));
            }
        } else {  // DefaultModel
            const float al = a.alpha.v;
            C[0].v = -(0.02f + 0.3f * (al * al));
            C[1].v = -0.98f * a.beta.v;
            C[2].v = -(5.0f * al);
            C[3].v = (0.08f * 4.0f * kDeg) * da.v + (-0.05f) * w[0].v;
            C[4].v = (-1.2f * 5.0f * kDeg) * de.v + (-0.5f) * w[1].v;
            C[5].v = (-0.1f * 6.0f * kDeg) * dr.v + (-0.05f) * w[2].v;
            const float c0 = -0.6f * al;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                C[0].d[j] = c0 * a.alpha.d[j];
                C[1].d[j] = -0.98f * a.beta.d[j];
                C[2].d[j] = -5.0f * a.alpha.d[j];
                C[3].d[j] = fmaf(0.08f * 4.0f * kDeg, da.d[j], -0.05f * w[0].d[j]);
                C[4].d[j] = fmaf(-1.2f * 5.0f * kDeg, de.d[j], -0.5f * w[1].d[j]);
                C[5].d[j] = fmaf(-0.1f * 6.0f * kDeg, dr.d[j], -0.05f * w[2].d[j]);
            }
        }
    }
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

// the same on first-order duals with every update fused (one instruction per tangent and term)
template <int N>
AC_DI void aero_post(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> u[7], Dual<N> C[6], AeroPost<Dual<N>>& o) {
    typedef Dual<N> T;
    if (P.p.stall_scaling) {  // uniform branch; dynamics/aircraft.py:280-294
        const float lim = 30.0f * kDeg, steep = 10.0f;
        const T sa = 1.0f / (1.0f + m_exp(steep * (m_fabs(a.alpha) - lim)));
        const T sb = 1.0f / (1.0f + m_exp(steep * (m_fabs(a.beta) - lim)));
        C[2] = C[2] * sa; C[2] = C[2] * sb; C[4] = C[4] * sa;
    }
    C[0] = dual_axpy(-0.1f, u[6], C[0]);
    C[2] = dual_axpy(-0.6f, u[6], C[2]);
    const T qS = a.qbar * P.p.S;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.C[k] = C[k];
    const float sg = sign_of(a.vr[0].v);
    const float sc[3] = {sg, 1.0f, 1.0f};
    const float len[3] = {P.p.b, P.p.c, P.p.b};
    T Ma[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float f = C[k].v * qS.v, m = C[3 + k].v * qS.v;
        o.F[k].v = k == 0 ? f * sg : f;
        Ma[k].v = m * len[k];
        const float cq = C[k].v * sc[k], qs = qS.v * sc[k];          // d F_k = sign (C_k d qS + qS d C_k)
        const float cm = C[3 + k].v * len[k], qm = qS.v * len[k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            o.F[k].d[j] = fmaf(cq, qS.d[j], qs * C[k].d[j]);
            Ma[k].d[j] = fmaf(cm, qS.d[j], qm * C[3 + k].d[j]);
        }
    }
    const float* com = P.p.com;
    o.M[0].v = Ma[0].v + (com[1] * o.F[2].v - com[2] * o.F[1].v);
    o.M[1].v = Ma[1].v + (com[2] * o.F[0].v - com[0] * o.F[2].v);
    o.M[2].v = Ma[2].v + (com[0] * o.F[1].v - com[1] * o.F[0].v);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        o.M[0].d[j] = fmaf(com[1], o.F[2].d[j], fmaf(-com[2], o.F[1].d[j], Ma[0].d[j]));
        o.M[1].d[j] = fmaf(com[2], o.F[0].d[j], fmaf(-com[0], o.F[2].d[j], Ma[1].d[j]));
        o.M[2].d[j] = fmaf(com[0], o.F[1].d[j], fmaf(-com[1], o.F[0].d[j], Ma[2].d[j]));
    }
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

// The same with tangents in closed form (see rot_inverse above): forces_ned by  dFn = M' dF + 2 (dq q^-1)_vec x Fn,  the
// quaternion kinematics and Euler's equation as bilinear forms with primal coefficients.
template <int N> AC_DI void rigid_body(const DevParams& P, const Dual<N> x[13], const AeroPost<Dual<N>>& o, Dual<N> xd[13]) {
    const Q4<float> q{x[6].v, x[7].v, x[8].v, x[9].v};
    const Q4<float> qi = qinv(q);
    const float w0 = x[10].v, w1 = x[11].v, w2 = x[12].v;
    const Q4<float> Fn = qmul(qmul_vec(q, o.F[0].v, o.F[1].v, o.F[2].v), qi);
    const Rot3 R = rot_inverse(q, qi);
    const float im = 1.0f / P.p.mass;
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    xd[3].v = Fn.x * im + P.p.gravity[0];
    xd[4].v = Fn.y * im + P.p.gravity[1];
    xd[5].v = Fn.z * im + P.p.gravity[2];
    const Q4<float> hq{0.5f * q.x, 0.5f * q.y, 0.5f * q.z, 0.5f * q.w};
    const Q4<float> qd = qmul_vec(hq, w0, w1, w2);
    xd[6].v = qd.x; xd[7].v = qd.y; xd[8].v = qd.z; xd[9].v = qd.w;
    const float* I = P.p.inertia;
    const float* Ii = P.p.inertia_inv;
    float Iw[3], rhs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Iw[i] = I[3 * i] * w0 + I[3 * i + 1] * w1 + I[3 * i + 2] * w2;
    rhs[0] = o.M[0].v - (w1 * Iw[2] - w2 * Iw[1]);
    rhs[1] = o.M[1].v - (w2 * Iw[0] - w0 * Iw[2]);
    rhs[2] = o.M[2].v - (w0 * Iw[1] - w1 * Iw[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[10 + i].v = Ii[3 * i] * rhs[0] + Ii[3 * i + 1] * rhs[1] + Ii[3 * i + 2] * rhs[2];
    // tangent coefficients (primal): rows of M' scaled by 1/m, 2 Fn / m, omega / 2
    float Rm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) Rm[i][k] = R.m[k][i] * im;
    const float im2 = im + im;
    const float f2[3] = {Fn.x * im2, Fn.y * im2, Fn.z * im2};
    const float hw0 = 0.5f * w0, hw1 = 0.5f * w1, hw2 = 0.5f * w2;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float dqx = x[6].d[j], dqy = x[7].d[j], dqz = x[8].d[j], dqw = x[9].d[j];
        const float dw0 = x[10].d[j], dw1 = x[11].d[j], dw2 = x[12].d[j];
        float e[3];
        qinv_times_dq<-1>(qi, dqx, dqy, dqz, dqw, e);
        const float dF0 = o.F[0].d[j], dF1 = o.F[1].d[j], dF2 = o.F[2].d[j];
        xd[3].d[j] = fmaf(Rm[0][0], dF0, fmaf(Rm[0][1], dF1, fmaf(Rm[0][2], dF2, fmaf(e[1], f2[2], -(e[2] * f2[1])))));
        xd[4].d[j] = fmaf(Rm[1][0], dF0, fmaf(Rm[1][1], dF1, fmaf(Rm[1][2], dF2, fmaf(e[2], f2[0], -(e[0] * f2[2])))));
        xd[5].d[j] = fmaf(Rm[2][0], dF0, fmaf(Rm[2][1], dF1, fmaf(Rm[2][2], dF2, fmaf(e[0], f2[1], -(e[1] * f2[0])))));
        // q_dot = 1/2 q (x) (omega, 0):  d = dq (x) (omega/2, 0) + (q/2) (x) (d omega, 0)
        xd[6].d[j] = fmaf(dqw, hw0, fmaf(dqy, hw2, fmaf(-dqz, hw1, fmaf(hq.w, dw0, fmaf(hq.y, dw2, -(hq.z * dw1))))));
        xd[7].d[j] = fmaf(dqw, hw1, fmaf(dqz, hw0, fmaf(-dqx, hw2, fmaf(hq.w, dw1, fmaf(hq.z, dw0, -(hq.x * dw2))))));
        xd[8].d[j] = fmaf(dqw, hw2, fmaf(dqx, hw1, fmaf(-dqy, hw0, fmaf(hq.w, dw2, fmaf(hq.x, dw1, -(hq.y * dw0))))));
        xd[9].d[j] = -fmaf(dqx, hw0, fmaf(dqy, hw1, fmaf(dqz, hw2, fmaf(hq.x, dw0, fmaf(hq.y, dw1, hq.z * dw2)))));
        // omega_dot = I^-1 (M - omega x I omega)
        float dIw[3], dr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dIw[i] = fmaf(I[3 * i], dw0, fmaf(I[3 * i + 1], dw1, I[3 * i + 2] * dw2));
        dr[0] = o.M[0].d[j] - fmaf(dw1, Iw[2], fmaf(w1, dIw[2], -fmaf(dw2, Iw[1], w2 * dIw[1])));
        dr[1] = o.M[1].d[j] - fmaf(dw2, Iw[0], fmaf(w2, dIw[0], -fmaf(dw0, Iw[2], w0 * dIw[2])));
        dr[2] = o.M[2].d[j] - fmaf(dw0, Iw[1], fmaf(w0, dIw[1], -fmaf(dw1, Iw[0], w1 * dIw[0])));
#pragma unroll
        for (int i = 0; i < 3; ++i) xd[10 + i].d[j] = fmaf(Ii[3 * i], dr[0], fmaf(Ii[3 * i + 1], dr[1], Ii[3 * i + 2] * dr[2]));
    }
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
        AC_OPAQUE_V(g0);  // (likewise: not hoisted out of an enclosing sub-step / unit-group loop)
#pragma unroll
        for (int i = 0; i < 13; ++i) { xs[i] = Seeds::state(g0, i, xv[i]); acc[i] = T(0.f); }
    }
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        coeffs.prefetch(P, xs, uv);
        int gg = g;
        AC_OPAQUE_V(gg);  // keep the seed patterns out of loop-invariant registers
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
            acc[i] = dual_axpy(wsum, k[i], acc[i]);
            xs[i] = dual_mul_add(hs, k[i], Seeds::state(gg, i, xv[i]));
        }
    }
    int ge = g;
    AC_OPAQUE_V(ge);  // the seeds of the final combination are rebuilt here, not carried across the four stages
    const T h6 = Seeds::step(ge, hv * (1.0f / 6.0f), dh_ddt * (1.0f / 6.0f));
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = dual_mul_add(h6, acc[i], Seeds::state(ge, i, xv[i]));
}

}  // namespace ac

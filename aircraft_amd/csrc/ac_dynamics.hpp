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
    // Tables of the cubic fits in DEVICE memory (the handle's; host memory in the host build of these headers), in rows of
    // 16 floats = one s_load_dwordx16 each (PolyTab below):
    //   rows  3 k .. 3 k + 2     fit k: intercept, coef[34], padding
    //   rows 18 + 4 k + v        d(fit k)/d(f_v) as a quadratic in f over the basis 1, f_0..f_3, f_a f_b (a <= b, sklearn
    //                            order) — derived from coef on the host (poly_gradient_tables): the sensitivity kernels
    //                            evaluate value and gradient of the fits on the primal and chain the tangents through the
    //                            gradient instead of pushing duals through 34 monomials
    // NOT kernel-argument arrays: 570 wave-uniform floats read inside the RK4 stage loop are loop-invariant scalar
    // loads, which hipcc hoists out of the loop into ~570 scalar registers it does not have and spills lane by lane
    // into vector registers (v_writelane / v_readlane: a third of the vector instructions of the round-3 poly kernels).
    // PolyTab reads them behind a pointer the optimiser cannot see through, taken anew wherever a fit is evaluated, and
    // the evaluations stream the rows two at a time (the next pair in flight while the current one is consumed).
    const float* poly_tab;
    // What the forward / derivative kernels store in place of their result: the multiple-shooting rows built from it
    // (ControlProblem.state_constraint, control/base.py:275-286).  Set by the ac_shoot_defect / ac_shoot_implicit_* entry
    // points on the copy of this struct that a launch takes; AC_ROWS_PLAIN everywhere else.
    int rows;                       // AC_ROWS_*
    float rows_dt;                  // dt_k of the implicit rows (the derivative kernels have no dt argument)
    const float* rows_dt_per_unit;  // [n] or NULL
    float* rows_aux;                // implicit rows with Jacobians: d r / d dt = -f, [13][n]
    float mlp_in_mean[5], mlp_in_std[5], mlp_out_mean[6], mlp_out_std[6];
    // mlp_out_std[k] / mlp_in_std[j], rounded once on the host (IEEE single division, what the device computes too): the
    // chain rule dC_k = sum_j J[k][j] * jscale[k][j] * d(in_j) reads them as scalar operands instead of holding thirty
    // wave-uniform quotients in vector registers for the life of the kernel
    float mlp_jscale[6][5];
};

constexpr float kDeg = 0.017453292519943295f;  // pi/180
enum { AC_ROWS_PLAIN = 0, AC_ROWS_DEFECT = 1, AC_ROWS_IMPLICIT = 2 };

constexpr int kPolyTabRows = 66, kPolyTabFloats = kPolyTabRows * 16;  // 18 value + 24 gradient + 24 second-derivative rows
struct Row16 { float c[16]; };
struct PolyTab {
    const AC_CONSTANT float* t;
    AC_DI explicit PolyTab(const DevParams& P) {
        const float* p = P.poly_tab;
        AC_OPAQUE_S(p);
        t = (const AC_CONSTANT float*)__builtin_assume_aligned((const AC_CONSTANT float*)p, 64);
    }
    AC_DI float coef(int k, int q) const { return t[k * 48 + 1 + q]; }
    AC_DI float intercept(int k) const { return t[k * 48]; }
    AC_DI float grad(int k, int v, int q) const { return t[(18 + k * 4 + v) * 16 + q]; }
    // Request row r: ONE s_load_dwordx16, as a volatile asm statement — plain loads from the constant address space are
    // speculatable, and hipcc gathers every row of a stage into the first basic block that dominates their uses (ahead of
    // the branches inside atan2f / asinf), 40 rows = 640 scalar registers at once.  arrive() is the matching wait; it takes
    // the row as an in/out operand so that every use is ordered behind it.
    AC_DI Row16 row(int r) const {
        Row16 o;
#ifdef AC_HOST_CHECK
        for (int i = 0; i < 16; ++i) o.c[i] = t[r * 16 + i];
#else
        typedef float v16f __attribute__((ext_vector_type(16)));
        v16f v;
        const int off = r * 64;
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(t), "s"(off));
#pragma unroll
        for (int i = 0; i < 16; ++i) o.c[i] = v[i];
#endif
        return o;
    }
    static AC_DI void arrive(Row16& a, Row16& b) {
#ifndef AC_HOST_CHECK
        typedef float v16f __attribute__((ext_vector_type(16)));
        v16f va, vb;
#pragma unroll
        for (int i = 0; i < 16; ++i) { va[i] = a.c[i]; vb[i] = b.c[i]; }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(va), "+s"(vb));
#pragma unroll
        for (int i = 0; i < 16; ++i) { a.c[i] = va[i]; b.c[i] = vb[i]; }
#else
        (void)a; (void)b;
#endif
    }
    static constexpr int value_row(int k, int part) { return 3 * k + part; }
    static constexpr int grad_row(int k, int v) { return 18 + 4 * k + v; }
    // second derivatives of fit k: entries e = 0..9 = (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3), each linear in f
    // ([c, c_f0, c_f1, c_f2, c_f3]); row r holds the entries 3 r .. 3 r + 2
    static constexpr int hess_row(int k, int r) { return 42 + 4 * k + r; }
};
// host side of the layout
inline void poly_pack_tables(const float* coef /*[6][34]*/, const float* intercept /*[6]*/, const float* grad /*[6][4][15]*/,
                             const float* hess /*[6][10][5]*/, float* tab /*[kPolyTabFloats]*/) {
    for (int i = 0; i < kPolyTabFloats; ++i) tab[i] = 0.f;
    for (int k = 0; k < 6; ++k) {
        tab[k * 48] = intercept[k];
        for (int q = 0; q < 34; ++q) tab[k * 48 + 1 + q] = coef[k * 34 + q];
        for (int v = 0; v < 4; ++v)
            for (int q = 0; q < 15; ++q) tab[(18 + k * 4 + v) * 16 + q] = grad[(k * 4 + v) * 15 + q];
        for (int e = 0; e < 10; ++e)
            for (int t = 0; t < 5; ++t) tab[(42 + k * 4 + e / 3) * 16 + 5 * (e % 3) + t] = hess[(k * 10 + e) * 5 + t];
    }
}

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
    const float den = AC_RCP(fmaf(ux, ux, v2 * v2));
    const float ca_y = ux * den, ca_x = -(v2 * den);            // d alpha = ca_y d v2 + ca_x d v0
    const float rV = AC_RCP(V);
    const float gV = 0.5f * rV;                                  // d V = gV d vv
    const float gb = rV * AC_RSQ(fmaf(-t, t, 1.0f));             // d beta = gb (d v1 - t d V)
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
// fit k over the 34 monomials mseq (f, pairs, triples in sklearn order): part 0 consumes the intercept and the first 31
// coefficients (two table rows), part 1 the last three (one row)
AC_DI float poly_value_part(int part, const Row16& r0, const Row16& r1, const float mseq[34], float acc) {
    if (part == 0) {
        acc = r0.c[0];
#pragma unroll
        for (int q = 0; q < 15; ++q) acc = fmaf(r0.c[1 + q], mseq[q], acc);
#pragma unroll
        for (int q = 15; q < 31; ++q) acc = fmaf(r1.c[q - 15], mseq[q], acc);
    } else {
#pragma unroll
        for (int q = 31; q < 34; ++q) acc = fmaf(r0.c[q - 31], mseq[q], acc);
    }
    return acc;
}
AC_DI void poly_monomials(const float f[4], float m[34]) {
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
}
template <int NOUT>
AC_DI void poly_eval(const DevParams& P, const int (&ks)[NOUT], const float f[4], float out[NOUT]) {
    const PolyTab tab(P);
    float m[34];
    poly_monomials(f, m);
    // items: (fit o, part p), o-major; rows of item i + 1 are requested before item i is consumed
    Row16 c0 = tab.row(PolyTab::value_row(ks[0], 0)), c1 = tab.row(PolyTab::value_row(ks[0], 1));
    PolyTab::arrive(c0, c1);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NOUT; ++i) {
        const int o = i >> 1, part = i & 1;
        Row16 n0 = c0, n1 = c1;
        if (i + 1 < 2 * NOUT) {
            const int on = (i + 1) >> 1, pn = (i + 1) & 1;
            n0 = tab.row(PolyTab::value_row(ks[on], pn == 0 ? 0 : 2));
            if (pn == 0) n1 = tab.row(PolyTab::value_row(ks[on], 1));
        }
        acc = poly_value_part(part, c0, c1, m, acc);
        AC_OPAQUE_V(acc);  // (computed HERE: otherwise the chain is sunk to its first use, and the rows wait for it in vector lanes)
        if (part == 1) out[o] = acc;
        AC_SCHED_FENCE();  // the chain above stays between the request and the wait of the next rows
        if (i + 1 < 2 * NOUT) PolyTab::arrive(n0, n1);
        c0 = n0; c1 = n1;
    }
}
template <int NOUT, class T>
AC_DI void poly_eval(const DevParams& P, const int (&ks)[NOUT], const T f[4], T out[NOUT]) {
    const PolyTab tab(P);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) out[o] = T(tab.intercept(ks[o]));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[o] = out[o] + tab.coef(ks[o], i) * f[i];
    int t2 = 4, t3 = 14;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            const T m2 = f[i] * f[j];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) out[o] = out[o] + tab.coef(ks[o], t2) * m2;
            ++t2;
#pragma unroll
            for (int k = j; k < 4; ++k) {
                const T m3 = m2 * f[k];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) out[o] = out[o] + tab.coef(ks[o], t3) * m3;
                ++t3;
            }
        }
}
// P_CZ(alpha, 0, 0, 0): only the pure-alpha monomials survive (terms 0, 4, 14)
template <class T> AC_DI T poly_cz_alpha_only(const DevParams& P, const T& al) {
    const PolyTab tab(P);
    const T a2 = al * al;
    return T(tab.intercept(2)) + tab.coef(2, 0) * al + tab.coef(2, 4) * a2 + tab.coef(2, 14) * (a2 * al);
}

// Host side of the grad part of DevParams::poly_tab (ac_set_poly; tests/host_dyn): every monomial of sklearn's degree-<=3 basis over four
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

// Host side of the second-derivative part of DevParams::poly_tab: the gradient tables differentiated once more.  Entry (v, q),
// v <= q, of fit k is linear in f: hess[k][e] = [c, c_f0 .. c_f3].
inline void poly_hessian_tables(const float* grad /*[6][4][15]*/, float* hess /*[6][10][5]*/) {
    for (int k = 0; k < 6; ++k) {
        int e = 0;
        for (int v = 0; v < 4; ++v)
            for (int q = v; q < 4; ++q, ++e) {
                const float* g = grad + (k * 4 + v) * 15;  // d P_k / d f_v over [1, f_0..f_3, f_a f_b (a <= b)]
                double h[5] = {(double)g[1 + q], 0, 0, 0, 0};
                int idx = 5;
                for (int a = 0; a < 4; ++a)
                    for (int b = a; b < 4; ++b, ++idx) {  // d (f_a f_b) / d f_q = [a == q] f_b + [b == q] f_a
                        if (a == q) h[1 + b] += (double)g[idx];
                        if (b == q) h[1 + a] += (double)g[idx];
                    }
                for (int t = 0; t < 5; ++t) hess[(k * 10 + e) * 5 + t] = (float)h[t];
            }
    }
}

// Second derivatives of the fits ks[0..NOUT) at the primal point f (entry order: PolyTab::hess_row), rows streamed two at a time
// like poly_value_grad's.
template <int NOUT>
AC_DI void poly_hess(const DevParams& P, const int (&ks)[NOUT], const float f[4], float h[NOUT][10]) {
    const PolyTab tab(P);
    constexpr int NI = 2 * NOUT;  // item = (fit, row pair)
    Row16 c0 = tab.row(PolyTab::hess_row(ks[0], 0)), c1 = tab.row(PolyTab::hess_row(ks[0], 1));
    PolyTab::arrive(c0, c1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        Row16 n0 = c0, n1 = c1;
        if (i + 1 < NI) {
            n0 = tab.row(PolyTab::hess_row(ks[(i + 1) >> 1], 2 * ((i + 1) & 1)));
            n1 = tab.row(PolyTab::hess_row(ks[(i + 1) >> 1], 2 * ((i + 1) & 1) + 1));
        }
        const int o = i >> 1, e0 = 6 * (i & 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = fmaf(c0.c[5 * j + 4], f[3], fmaf(c0.c[5 * j + 3], f[2], fmaf(c0.c[5 * j + 2], f[1], fmaf(c0.c[5 * j + 1], f[0], c0.c[5 * j]))));
            AC_OPAQUE_V(a);
            h[o][e0 + j] = a;
            if (e0 + 3 + j < 10) {
                float b = fmaf(c1.c[5 * j + 4], f[3], fmaf(c1.c[5 * j + 3], f[2], fmaf(c1.c[5 * j + 2], f[1], fmaf(c1.c[5 * j + 1], f[0], c1.c[5 * j]))));
                AC_OPAQUE_V(b);
                h[o][e0 + 3 + j] = b;
            }
        }
        AC_SCHED_FENCE();
        if (i + 1 < NI) PolyTab::arrive(n0, n1);
        c0 = n0; c1 = n1;
    }
}

// Value and gradient of the fits ks[0..NOUT) at the primal point f: 30 shared monomial products, 34 + 4 x 14 fused
// multiply-adds per fit, every coefficient a scalar operand.
template <int NOUT>
AC_DI void poly_value_grad(const DevParams& P, const int (&ks)[NOUT], const float f[4], float val[NOUT], float grad[NOUT][4]) {
    const PolyTab tab(P);
    float m[34];
    poly_monomials(f, m);
    // items 0 .. 2 NOUT - 1: two gradient chains each (fit o, variables 2 h and 2 h + 1: 15 fused multiply-adds over 1, f, pairs);
    // items 2 NOUT .. 4 NOUT - 1: the value chain of fit o in two parts.  Two table rows per item, the next item's rows are
    // requested before the current item is consumed and waited for after it: at most four rows
    // (64 scalar registers) are ever live — left to itself the compiler requests every row of an evaluation point at
    // once and spills the scalar registers into vector lanes.
    constexpr int NI = 4 * NOUT;
    Row16 c0 = tab.row(PolyTab::grad_row(ks[0], 0)), c1 = tab.row(PolyTab::grad_row(ks[0], 1));
    PolyTab::arrive(c0, c1);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        Row16 n0 = c0, n1 = c1;
        if (i + 1 < NI) {
            const int in = i + 1;
            if (in < 2 * NOUT) {
                n0 = tab.row(PolyTab::grad_row(ks[in >> 1], 2 * (in & 1)));
                n1 = tab.row(PolyTab::grad_row(ks[in >> 1], 2 * (in & 1) + 1));
            } else {
                const int on = (in - 2 * NOUT) >> 1, pn = (in - 2 * NOUT) & 1;
                n0 = tab.row(PolyTab::value_row(ks[on], pn == 0 ? 0 : 2));
                if (pn == 0) n1 = tab.row(PolyTab::value_row(ks[on], 1));
            }
        }
        if (i < 2 * NOUT) {
            const int o = i >> 1, v = 2 * (i & 1);
            float g0 = c0.c[0], g1 = c1.c[0];
#pragma unroll
            for (int q = 0; q < 14; ++q) { g0 = fmaf(c0.c[1 + q], m[q], g0); g1 = fmaf(c1.c[1 + q], m[q], g1); }
            AC_OPAQUE_V(g0); AC_OPAQUE_V(g1);  // (computed HERE: see poly_eval)
            grad[o][v] = g0; grad[o][v + 1] = g1;
        } else {
            const int o = (i - 2 * NOUT) >> 1, part = (i - 2 * NOUT) & 1;
            acc = poly_value_part(part, c0, c1, m, acc);
            AC_OPAQUE_V(acc);
            if (part == 1) val[o] = acc;
        }
        AC_SCHED_FENCE();  // the chains above stay between the request and the wait of the next rows
        if (i + 1 < NI) PolyTab::arrive(n0, n1);
        c0 = n0; c1 = n1;
    }
}

// differentials of the aerodynamic inputs along one tangent direction (what a coefficient model's tangent() consumes)
struct AeroD { float vr[3], alpha, beta, qbar, w[3], da, de, dr; };

// Coefficient-provider protocol:
//   prefetch(P, x, uv)      called on the stage state BEFORE anything else of the stage is computed; the MLP
//                           provider runs the whole network here from primal inputs, so that nothing but the
//                           RK4 carry is live across the (register-hungry) network evaluation
//   operator()(P, a, x, u, C)   turns the aerodynamic inputs (with tangents) into the six coefficients
// SHARED (cubic fits, the kernels whose four waves are the four direction groups of the SAME 64 units): the primal part of
// linearise() — six fits with their gradients, 564 multiply-adds and four inverse trigonometric functions per stage — is the
// same in all four waves.  Each wave evaluates a quarter of it and the quarters are exchanged through LDS (`xch`, kPolyXchFloats
// floats, [2][34][64]: double-buffered, ONE workgroup barrier per call).  Same chains, same operations: bit-identical results.
constexpr int kPolyXchRows = 34, kPolyXchFloats = 2 * kPolyXchRows * 64;
template <int MODEL, bool SHARED = false> struct AnalyticCoeffs {
    static constexpr int kModel = MODEL;
    static_assert(!SHARED || MODEL == AC_MODEL_POLY, "only the cubic fits are worth sharing");
    float* xch = nullptr;  // SHARED: the workgroup's exchange buffer
    int xg = 0;            // SHARED: this wave's quarter (wave-uniform)
    int xph = 0;           // SHARED: call counter (which half of the buffer)
    template <class T> AC_DI void prefetch(const DevParams&, const T*, const float*) {}
    // First-order tangents, one direction at a time (state_derivative for duals, below): linearise() evaluates the model on
    // the primal and keeps its partial derivatives (closed forms; for the cubic fits value and gradient from the
    // host-derived tables), tangent() is the chain-rule row of one direction.
    static constexpr bool kFusedTangent = true;
    struct NoLin {};
    struct PolyLin { float g4[4][4], ge[4], gr[4], cey, cex, cly, clx, cry, crx, kb0, kb2, kby, sr, sl, arm, b4; };
    struct DefaultLin { float c0; };
    typename std::conditional<MODEL == AC_MODEL_POLY, PolyLin,
                              typename std::conditional<MODEL == AC_MODEL_DEFAULT, DefaultLin, NoLin>::type>::type lin;

    AC_DI void linearise(const DevParams& P, const AeroPre<float>& a, const float x[13], const float u[7], float C[6]) {
        if constexpr (MODEL == AC_MODEL_POLY) {
            const float* w = &x[10];
            const float eps = P.p.epsilon, arm = P.p.rudder_moment_arm, b4 = P.p.b * 0.25f;
            const float v0 = a.vr[0], v1 = a.vr[1], v2 = a.vr[2];
            const float ux = v0 + eps;
            // effective angles (aircraft.py:189-233) with the coefficients of their differentials
            const float ye = v2 + arm * w[1], yl = v2 - b4 * w[0], yr = v2 + b4 * w[0];
            const float de_ = AC_RCP(fmaf(ux, ux, ye * ye)), dl_ = AC_RCP(fmaf(ux, ux, yl * yl)), dr_ = AC_RCP(fmaf(ux, ux, yr * yr));
            const float vy = v1 - arm * w[2];
            const float nb2 = v0 * v0 + vy * vy + v2 * v2 + eps;
            auto sin_beta_r = [&]() { return vy / sqrtf(nb2); };  // the primal argument of asin: the forward kernels' expression
            const float rnb = AC_RSQ(nb2), tbt = vy * rnb;
            const float gb = rnb * AC_RSQ(fmaf(-tbt, tbt, 1.0f));
            const float kb = gb * tbt * rnb;  // d beta_r = gb d vy - kb (v0 d v0 + vy d vy + v2 d v2)
            float val4[4], vale[1], valr[1], ge[1][4], gr[1][4];
            float czr, czl;
            const float hb = b4 * 0.5f;
            if constexpr (!SHARED) {
                const float alpha_e = atan2f(ye, ux), alpha_l = atan2f(yl, ux), alpha_r = atan2f(yr, ux);
                const float beta_r = asinf(sin_beta_r());
                {
                    const float f[4] = {a.alpha, a.beta, u[0], u[1]};
                    const int ks[4] = {0, 1, 2, 3};
                    poly_value_grad<4>(P, ks, f, val4, lin.g4);
                }
                {
                    const float f[4] = {alpha_e, a.beta, u[0], u[1]};
                    const int ks[1] = {4};
                    poly_value_grad<1>(P, ks, f, vale, ge);
                }
                {
                    const float f[4] = {a.alpha, beta_r, u[0], u[1]};
                    const int ks[1] = {5};
                    poly_value_grad<1>(P, ks, f, valr, gr);
                }
                // P_CZ(alpha_x, 0, 0, 0) and its slope
                const PolyTab tab(P);
                const float z0 = tab.intercept(2), z1 = tab.coef(2, 0), z2 = tab.coef(2, 4), z3 = tab.coef(2, 14);
                czr = fmaf(fmaf(fmaf(z3, alpha_r, z2), alpha_r, z1), alpha_r, z0);
                czl = fmaf(fmaf(fmaf(z3, alpha_l, z2), alpha_l, z1), alpha_l, z0);
                lin.sr = hb * fmaf(fmaf(3.0f * z3, alpha_r, z2 + z2), alpha_r, z1);
                lin.sl = -(hb * fmaf(fmaf(3.0f * z3, alpha_l, z2 + z2), alpha_l, z1));
            } else {
#ifndef AC_HOST_CHECK
                // rows of the exchange buffer: 0..5 the values, 6 + 4 k + v the gradients, 30 / 31 cz and slope term of the left
                // wing station, 32 / 33 of the right one
                float* xb = xch + (xph & 1) * (kPolyXchRows * 64) + (threadIdx.x & 63);
                ++xph;
                if (xg < 2) {  // waves 0, 1: fits (0, 1), (2, 3) at (alpha, beta)
                    const float f[4] = {a.alpha, a.beta, u[0], u[1]};
                    const int ks[2] = {2 * xg, 2 * xg + 1};
                    float v2[2], g2[2][4];
                    poly_value_grad<2>(P, ks, f, v2, g2);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        xb[(2 * xg + o) * 64] = v2[o];
#pragma unroll
                        for (int v = 0; v < 4; ++v) xb[(6 + 4 * (2 * xg + o) + v) * 64] = g2[o][v];
                    }
                } else {  // wave 2: the elevator fit at alpha_e + the left wing station; wave 3: the rudder fit at beta_r + the right one
                    const bool el = xg == 2;
                    const float ang = el ? atan2f(ye, ux) : asinf(sin_beta_r());
                    const float f[4] = {el ? ang : a.alpha, el ? a.beta : ang, u[0], u[1]};
                    const int ks[1] = {el ? 4 : 5};
                    float v1[1], g1[1][4];
                    poly_value_grad<1>(P, ks, f, v1, g1);
                    xb[ks[0] * 64] = v1[0];
#pragma unroll
                    for (int v = 0; v < 4; ++v) xb[(6 + 4 * ks[0] + v) * 64] = g1[0][v];
                    const float alpha_w = atan2f(el ? yl : yr, ux);
                    const PolyTab tab(P);
                    const float z0 = tab.intercept(2), z1 = tab.coef(2, 0), z2 = tab.coef(2, 4), z3 = tab.coef(2, 14);
                    const float cz = fmaf(fmaf(fmaf(z3, alpha_w, z2), alpha_w, z1), alpha_w, z0);
                    const float sw = hb * fmaf(fmaf(3.0f * z3, alpha_w, z2 + z2), alpha_w, z1);
                    xb[(el ? 30 : 32) * 64] = cz;
                    xb[(el ? 31 : 33) * 64] = el ? -sw : sw;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    val4[k] = xb[k * 64];
#pragma unroll
                    for (int v = 0; v < 4; ++v) lin.g4[k][v] = xb[(6 + 4 * k + v) * 64];
                }
                vale[0] = xb[4 * 64]; valr[0] = xb[5 * 64];
#pragma unroll
                for (int v = 0; v < 4; ++v) { ge[0][v] = xb[(22 + v) * 64]; gr[0][v] = xb[(26 + v) * 64]; }
                czl = xb[30 * 64]; lin.sl = xb[31 * 64]; czr = xb[32 * 64]; lin.sr = xb[33 * 64];
#else
                czr = czl = 0.f;
#endif
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) { lin.ge[v] = ge[0][v]; lin.gr[v] = gr[0][v]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) C[k] = val4[k];
            C[3] = fmaf(hb, czr - czl, C[3]);
            C[4] = vale[0];
            C[5] = fmaf(0.01f * 6.0f * kDeg, u[2], valr[0]);
            // d alpha_x = cx_y d y_x + cx_x d v0
            lin.cey = ux * de_; lin.cex = -(ye * de_); lin.cly = ux * dl_; lin.clx = -(yl * dl_);
            lin.cry = ux * dr_; lin.crx = -(yr * dr_);
            lin.kb0 = -(kb * v0); lin.kb2 = -(kb * v2); lin.kby = fmaf(-kb, vy, gb);
            lin.arm = arm; lin.b4 = b4;
        } else {
            (*this)(P, a, x, u, C);  // the forward kernels' expressions
            if constexpr (MODEL == AC_MODEL_DEFAULT) lin.c0 = -0.6f * a.alpha;
        }
    }
    AC_DI void tangent(const DevParams& P, const AeroD& d, float dC[6]) const {
        if constexpr (MODEL == AC_MODEL_LINEAR) {
            const float in[5] = {d.qbar, d.alpha, d.beta, d.da, d.de};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float t = P.linear_W[k * 6 + 0] * in[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) t = fmaf(P.linear_W[k * 6 + j], in[j], t);
                dC[k] = t;
            }
            dC[5] = fmaf(-0.1f * 6.0f * kDeg, d.dr, dC[5]);
        } else if constexpr (MODEL == AC_MODEL_POLY) {
            const float dae = fmaf(lin.cey, fmaf(lin.arm, d.w[1], d.vr[2]), lin.cex * d.vr[0]);
            const float dal = fmaf(lin.cly, fmaf(-lin.b4, d.w[0], d.vr[2]), lin.clx * d.vr[0]);
            const float dar = fmaf(lin.cry, fmaf(lin.b4, d.w[0], d.vr[2]), lin.crx * d.vr[0]);
            const float dbr = fmaf(lin.kby, fmaf(-lin.arm, d.w[2], d.vr[1]), fmaf(lin.kb0, d.vr[0], lin.kb2 * d.vr[2]));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dC[k] = fmaf(lin.g4[k][0], d.alpha, fmaf(lin.g4[k][1], d.beta, fmaf(lin.g4[k][2], d.da, lin.g4[k][3] * d.de)));
            dC[3] = fmaf(lin.sr, dar, fmaf(lin.sl, dal, dC[3]));
            dC[4] = fmaf(lin.ge[0], dae, fmaf(lin.ge[1], d.beta, fmaf(lin.ge[2], d.da, lin.ge[3] * d.de)));
            dC[5] = fmaf(lin.gr[0], d.alpha, fmaf(lin.gr[1], dbr, fmaf(lin.gr[2], d.da, fmaf(lin.gr[3], d.de, (0.01f * 6.0f * kDeg) * d.dr))));
        } else {  // DefaultModel
            dC[0] = lin.c0 * d.alpha;
            dC[1] = -0.98f * d.beta;
            dC[2] = -5.0f * d.alpha;
            dC[3] = fmaf(0.08f * 4.0f * kDeg, d.da, -0.05f * d.w[0]);
            dC[4] = fmaf(-1.2f * 5.0f * kDeg, d.de, -0.5f * d.w[1]);
            dC[5] = fmaf(-0.1f * 6.0f * kDeg, d.dr, -0.05f * d.w[2]);
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

// stall: the factors (sa, sb) formed by the caller with the expressions below (state_derivative for duals shares them with the
// coefficients of its tangents), or nullptr
template <class T>
AC_DI void aero_post(const DevParams& P, const AeroPre<T>& a, const T u[7], T C[6], AeroPost<T>& o, const T* stall = nullptr) {
    if (P.p.stall_scaling) {  // uniform branch; dynamics/aircraft.py:280-294
        const float lim = 30.0f * kDeg, steep = 10.0f;
        const T sa = stall ? stall[0] : 1.0f / (1.0f + m_exp(steep * (m_fabs(a.alpha) - lim)));
        const T sb = stall ? stall[1] : 1.0f / (1.0f + m_exp(steep * (m_fabs(a.beta) - lim)));
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

// x_dot with first-order tangents.  Providers that implement linearise() / tangent() (the analytic models) take the FUSED
// form: everything that depends on the primal only — the forward kernels' own expressions for the values, and the
// coefficients of every differential — is formed once; then each direction runs from (dv, dq, d omega, du) to d x_dot on
// its own, so no N-wide intermediate (relative velocity, angles, coefficients, forces, moments: 88 registers at N = 4)
// is live between phases.  Other providers keep the phase structure above (overloads of aero_pre, aero_post, rigid_body).
template <class C, class = void> struct fused_tangent : std::false_type {};
template <class C> struct fused_tangent<C, typename std::enable_if<C::kFusedTangent>::type> : std::true_type {};

template <int N, class Coeffs>
AC_DI void state_derivative(const DevParams& P, Coeffs& coeffs, const Dual<N> x[13], const Dual<N> u[7], Dual<N> xd[13]) {
    if constexpr (!fused_tangent<Coeffs>::value || Coeffs::kModel == AC_MODEL_QUAD) {
        state_derivative<Dual<N>, Coeffs>(P, coeffs, x, u, xd);
    } else {
        const float eps = P.p.epsilon;
        float xv[13], uv[7];
#pragma unroll
        for (int i = 0; i < 13; ++i) xv[i] = x[i].v;
#pragma unroll
        for (int i = 0; i < 7; ++i) uv[i] = u[i].v;
        // ---- values: the float path ----
        AeroPre<float> a;
        aero_pre(P, xv, a);
        float C[6];
        coeffs.linearise(P, a, xv, uv, C);
        // stall factors (dynamics/aircraft.py:280-294) as coefficients of (dC, d alpha, d beta)
        float s2 = 1.f, k2a = 0.f, k2b = 0.f, s4 = 1.f, k4a = 0.f;
        float stall[2] = {1.f, 1.f};
        if (P.p.stall_scaling) {
            const float lim = 30.0f * kDeg, steep = 10.0f;
            const float ea = m_exp(steep * (m_fabs(a.alpha) - lim)), eb = m_exp(steep * (m_fabs(a.beta) - lim));
            const float sa = 1.0f / (1.0f + ea), sb = 1.0f / (1.0f + eb);  // (aero_post's expressions: it takes them from here)
            const float dsa = -(sa * sa) * ea * (a.alpha < 0.f ? -steep : steep);
            const float dsb = -(sb * sb) * eb * (a.beta < 0.f ? -steep : steep);
            s2 = sa * sb; s4 = sa;
            k2a = C[2] * sb * dsa; k2b = C[2] * sa * dsb; k4a = C[4] * dsa;
            stall[0] = sa; stall[1] = sb;
        }
        AeroPost<float> o;
        aero_post(P, a, uv, C, o, stall);  // (applies stall and flaps to C)
        float xdv[13];
        rigid_body(P, xv, o, xdv);
#pragma unroll
        for (int i = 0; i < 13; ++i) xd[i].v = xdv[i];
        // ---- tangents: per direction  (dv, dq, d omega, du) -> differentials of the aerodynamic inputs -> dC -> (dF, dM)
        // -> d x_dot.  Light models run a direction through all of it at once (no N-wide intermediate at all); the cubic
        // fits, whose 39 gradient coefficients would be live beside every other coefficient set (about 110 registers held
        // across the whole loop), run it as four passes separated by scheduling fences, each pass forming its own
        // coefficients behind the fence so that they are live for that pass only.
        constexpr bool kPasses = Coeffs::kModel == AC_MODEL_POLY;
        const Q4<float> q{xv[6], xv[7], xv[8], xv[9]};
        const Q4<float> qi = qinv(q);
        AeroD dd[kPasses ? N : 1];
        float dCC[kPasses ? N : 1][6], dFM[kPasses ? N : 1][6];
        if constexpr (kPasses) AC_SCHED_FENCE();
        {
            // (a) differentials of the aerodynamic inputs
            const Q4<float> r = qmul(qmul_vec(qi, xv[3], xv[4], xv[5]), q);
            const Rot3 R = rot_inverse(q, qi);
            const float r2[3] = {r.x + r.x, r.y + r.y, r.z + r.z};
            const float v0 = a.vr[0], v1 = a.vr[1], v2 = a.vr[2];
            const float ux = v0 + eps;
            const float den = AC_RCP(fmaf(ux, ux, v2 * v2));
            const float ca_y = ux * den, ca_x = -(v2 * den);        // d alpha = ca_y d v2 + ca_x d v0
            const float rV = AC_RCP(a.V);
            const float tb = v1 * rV;
            const float gb = rV * AC_RSQ(fmaf(-tb, tb, 1.0f));
            const float gbv = -(gb * tb) * (0.5f * rV);             // d beta = gb d v1 + gbv d(v.v)
            const float w0 = v0 + v0, w1 = v1 + v1, w2 = v2 + v2;   // d(v.v) = 2 v . dv
            auto aero_d = [&](int j, AeroD& d) {
                const float dv0 = x[3].d[j], dv1 = x[4].d[j], dv2 = x[5].d[j];
                const float dqx = x[6].d[j], dqy = x[7].d[j], dqz = x[8].d[j], dqw = x[9].d[j];
                // vector part of q^-1 dq:  bw dq_v + dq_w b + b x dq_v
                const float tx = fmaf(qi.w, dqx, dqw * qi.x), ty = fmaf(qi.w, dqy, dqw * qi.y), tz = fmaf(qi.w, dqz, dqw * qi.z);
                const float cx = fmaf(qi.y, dqz, -(qi.z * dqy)), cy = fmaf(qi.z, dqx, -(qi.x * dqz)), cz = fmaf(qi.x, dqy, -(qi.y * dqx));
                const float e0 = tx + cx, e1 = ty + cy, e2 = tz + cz;
                d.vr[0] = fmaf(R.m[0][0], dv0, fmaf(R.m[0][1], dv1, fmaf(R.m[0][2], dv2, fmaf(r2[1], e2, -(r2[2] * e1)))));
                d.vr[1] = fmaf(R.m[1][0], dv0, fmaf(R.m[1][1], dv1, fmaf(R.m[1][2], dv2, fmaf(r2[2], e0, -(r2[0] * e2)))));
                d.vr[2] = fmaf(R.m[2][0], dv0, fmaf(R.m[2][1], dv1, fmaf(R.m[2][2], dv2, fmaf(r2[0], e1, -(r2[1] * e0)))));
                const float dvv = fmaf(w0, d.vr[0], fmaf(w1, d.vr[1], w2 * d.vr[2]));
                d.alpha = fmaf(ca_y, d.vr[2], ca_x * d.vr[0]);
                d.beta = fmaf(gb, d.vr[1], gbv * dvv);
                d.qbar = (0.5f * 1.225f) * dvv;
                d.w[0] = x[10].d[j]; d.w[1] = x[11].d[j]; d.w[2] = x[12].d[j];
                d.da = u[0].d[j]; d.de = u[1].d[j]; d.dr = u[2].d[j];
            };
            // (b) the model's chain-rule row, stall factors, flaps
            auto coeff_d = [&](int j, const AeroD& d, float dC[6]) {
                coeffs.tangent(P, d, dC);
                if (P.p.stall_scaling) {
                    dC[2] = fmaf(s2, dC[2], fmaf(k2a, d.alpha, k2b * d.beta));
                    dC[4] = fmaf(s4, dC[4], k4a * d.alpha);
                }
                const float dfl = u[6].d[j];
                dC[0] = fmaf(-0.1f, dfl, dC[0]);
                dC[2] = fmaf(-0.6f, dfl, dC[2]);
            };
            // (c) d F_k = sign (C_k S d qbar + qS d C_k),  d Ma_k = len_k (C_{3+k} S d qbar + qS d C_{3+k}),  d M = d Ma + com x d F
            auto post_d = [&](float dqbar, const float dC[6], float fm[6]) {
                const float S = P.p.S;
                const float qS = a.qbar * S;
                const float sg = sign_of(a.vr[0]);
                const float len[3] = {P.p.b, P.p.c, P.p.b};
                const float* com = P.p.com;
                float dM[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float sc = k == 0 ? sg : 1.0f;
                    fm[k] = fmaf(C[k] * S * sc, dqbar, (qS * sc) * dC[k]);
                    dM[k] = fmaf(C[3 + k] * S * len[k], dqbar, (qS * len[k]) * dC[3 + k]);
                }
                fm[3] = fmaf(com[1], fm[2], fmaf(-com[2], fm[1], dM[0]));
                fm[4] = fmaf(com[2], fm[0], fmaf(-com[0], fm[2], dM[1]));
                fm[5] = fmaf(com[0], fm[1], fmaf(-com[1], fm[0], dM[2]));
            };
            if constexpr (kPasses) {
#pragma unroll
                for (int j = 0; j < N; ++j) aero_d(j, dd[j]);
                AC_SCHED_FENCE();
#pragma unroll
                for (int j = 0; j < N; ++j) coeff_d(j, dd[j], dCC[j]);
                AC_SCHED_FENCE();
#pragma unroll
                for (int j = 0; j < N; ++j) post_d(dd[j].qbar, dCC[j], dFM[j]);
                AC_SCHED_FENCE();
            }
            // (d) rigid body: forces_ned by the closed form of the rotation, quaternion kinematics, Euler's equation
            const float im = 1.0f / P.p.mass;
            const Q4<float> Fn = qmul(qmul_vec(q, o.F[0], o.F[1], o.F[2]), qi);
            float Rm[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) Rm[i][k] = R.m[k][i] * im;
            const float im2 = im + im;
            const float f2[3] = {Fn.x * im2, Fn.y * im2, Fn.z * im2};
            const float om0 = xv[10], om1 = xv[11], om2 = xv[12];
            const float hw0 = 0.5f * om0, hw1 = 0.5f * om1, hw2 = 0.5f * om2;
            const Q4<float> hq{0.5f * q.x, 0.5f * q.y, 0.5f * q.z, 0.5f * q.w};
            const float* I = P.p.inertia;
            const float* Ii = P.p.inertia_inv;
            float Iw[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Iw[i] = I[3 * i] * om0 + I[3 * i + 1] * om1 + I[3 * i + 2] * om2;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float fm[6];
                if constexpr (kPasses) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) fm[k] = dFM[j][k];
                } else {
                    AeroD d;
                    float dC[6];
                    aero_d(j, d);
                    coeff_d(j, d, dC);
                    post_d(d.qbar, dC, fm);
                }
                const float dqx = x[6].d[j], dqy = x[7].d[j], dqz = x[8].d[j], dqw = x[9].d[j];
                const float dw0 = x[10].d[j], dw1 = x[11].d[j], dw2 = x[12].d[j];
                // vector part of dq q^-1:  bw dq_v + dq_w b - b x dq_v
                const float tx = fmaf(qi.w, dqx, dqw * qi.x), ty = fmaf(qi.w, dqy, dqw * qi.y), tz = fmaf(qi.w, dqz, dqw * qi.z);
                const float cx = fmaf(qi.y, dqz, -(qi.z * dqy)), cy = fmaf(qi.z, dqx, -(qi.x * dqz)), cz = fmaf(qi.x, dqy, -(qi.y * dqx));
                const float p0 = tx - cx, p1 = ty - cy, p2 = tz - cz;
                xd[0].d[j] = x[3].d[j]; xd[1].d[j] = x[4].d[j]; xd[2].d[j] = x[5].d[j];
                xd[3].d[j] = fmaf(Rm[0][0], fm[0], fmaf(Rm[0][1], fm[1], fmaf(Rm[0][2], fm[2], fmaf(p1, f2[2], -(p2 * f2[1])))));
                xd[4].d[j] = fmaf(Rm[1][0], fm[0], fmaf(Rm[1][1], fm[1], fmaf(Rm[1][2], fm[2], fmaf(p2, f2[0], -(p0 * f2[2])))));
                xd[5].d[j] = fmaf(Rm[2][0], fm[0], fmaf(Rm[2][1], fm[1], fmaf(Rm[2][2], fm[2], fmaf(p0, f2[1], -(p1 * f2[0])))));
                xd[6].d[j] = fmaf(dqw, hw0, fmaf(dqy, hw2, fmaf(-dqz, hw1, fmaf(hq.w, dw0, fmaf(hq.y, dw2, -(hq.z * dw1))))));
                xd[7].d[j] = fmaf(dqw, hw1, fmaf(dqz, hw0, fmaf(-dqx, hw2, fmaf(hq.w, dw1, fmaf(hq.z, dw0, -(hq.x * dw2))))));
                xd[8].d[j] = fmaf(dqw, hw2, fmaf(dqx, hw1, fmaf(-dqy, hw0, fmaf(hq.w, dw2, fmaf(hq.x, dw1, -(hq.y * dw0))))));
                xd[9].d[j] = -fmaf(dqx, hw0, fmaf(dqy, hw1, fmaf(dqz, hw2, fmaf(hq.x, dw0, fmaf(hq.y, dw1, hq.z * dw2)))));
                float dIw[3], dr[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) dIw[i] = fmaf(I[3 * i], dw0, fmaf(I[3 * i + 1], dw1, I[3 * i + 2] * dw2));
                dr[0] = fm[3] - fmaf(dw1, Iw[2], fmaf(om1, dIw[2], -fmaf(dw2, Iw[1], om2 * dIw[1])));
                dr[1] = fm[4] - fmaf(dw2, Iw[0], fmaf(om2, dIw[0], -fmaf(dw0, Iw[2], om0 * dIw[2])));
                dr[2] = fm[5] - fmaf(dw0, Iw[1], fmaf(om0, dIw[1], -fmaf(dw1, Iw[0], om1 * dIw[0])));
#pragma unroll
                for (int i = 0; i < 3; ++i) xd[10 + i].d[j] = fmaf(Ii[3 * i], dr[0], fmaf(Ii[3 * i + 1], dr[1], Ii[3 * i + 2] * dr[2]));
            }
        }
    }
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

// Where the RK4 sum  k1 + 2 k2 + 2 k3 + k4  of a seeded step lives.  It is touched once per stage, so it need not occupy
// registers across the stage evaluation:
//   RegAcc<N>   registers (the MLP kernels: one wave per SIMD owns the whole register file anyway)
//   LdsAcc4     N = 4: the 13 x 4 tangents as one 16-byte LDS word per row and lane (conflict-free b128 accesses, one
//               read-modify-write per stage), the 13 values in registers — 52 registers less at the point of highest
//               pressure, which is what lets the analytic kernels run two waves per SIMD without scratch
template <int N> struct RegAcc {
    Dual<N> a[13];
    AC_DI void zero() {
#pragma unroll
        for (int i = 0; i < 13; ++i) a[i] = Dual<N>(0.f);
    }
    AC_DI void add(int i, float w, const Dual<N>& k) { a[i] = dual_axpy(w, k, a[i]); }
    AC_DI Dual<N> get(int i) const { return a[i]; }
};
#ifndef AC_HOST_CHECK
struct LdsAcc4 {
    float v[13];
    float4* base;  // this lane's word of row 0; rows are `stride` words apart
    int stride;
    AC_DI LdsAcc4(float4* lane_word, int stride_) : base(lane_word), stride(stride_) {}
    AC_DI void zero() {
#pragma unroll
        for (int i = 0; i < 13; ++i) { v[i] = 0.f; base[i * stride] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    AC_DI void add(int i, float w, const Dual<4>& k) {
        v[i] = fmaf(w, k.v, v[i]);
        float4 t = base[i * stride];
        t.x = fmaf(w, k.d[0], t.x); t.y = fmaf(w, k.d[1], t.y); t.z = fmaf(w, k.d[2], t.z); t.w = fmaf(w, k.d[3], t.w);
        base[i * stride] = t;
    }
    AC_DI Dual<4> get(int i) const {
        Dual<4> r; r.v = v[i];
        const float4 t = base[i * stride];
        r.d[0] = t.x; r.d[1] = t.y; r.d[2] = t.z; r.d[3] = t.w;
        return r;
    }
};
#endif

// (diagnostic flavour: a phase stamp through the coefficient provider's engine, where it has one)
template <class C, class = void> struct has_engine_stamper : std::false_type {};
template <class C> struct has_engine_stamper<C, std::void_t<decltype(std::declval<C&>().eng.st)>> : std::true_type {};
template <class C> AC_DI void provider_mark(C& coeffs, int id) {
#ifdef AC_STAMPS
    if constexpr (has_engine_stamper<C>::value) coeffs.eng.st.mark(id);
#else
    (void)coeffs; (void)id;
#endif
}

// One RK4 step from primal inputs; xo = F(x0, u, h) with tangents w.r.t. this lane's N directions.
template <int N, class Coeffs, class Acc>
AC_DI void rk4_step_seeded(const DevParams& P, Coeffs& coeffs, int g, const float xv[13], const float uv[7],
                           float hv, float dh_ddt, Dual<N> xo[13], Acc& acc) {
    typedef Dual<N> T;
    typedef SeedsT<N> Seeds;
    T xs[13], k[13];
    {
        int g0 = g;
        AC_OPAQUE_V(g0);  // (likewise: not hoisted out of an enclosing sub-step / unit-group loop)
#pragma unroll
        for (int i = 0; i < 13; ++i) xs[i] = Seeds::state(g0, i, xv[i]);
        acc.zero();
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
        if (s == 3) provider_mark(coeffs, 8);  // [8] the last stage's dual rigid body
        const float wsum = (s == 1 || s == 2) ? 2.0f : 1.0f;
        const float cnext = (s == 2) ? 1.0f : 0.5f;
        const T hs = Seeds::step(gg, hv * cnext, dh_ddt * cnext);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            acc.add(i, wsum, k[i]);
            xs[i] = dual_mul_add(hs, k[i], Seeds::state(gg, i, xv[i]));
        }
        if (s == 3) provider_mark(coeffs, 9);  // [9] its RK4 accumulation
    }
    int ge = g;
    AC_OPAQUE_V(ge);  // the seeds of the final combination are rebuilt here, not carried across the four stages
    const T h6 = Seeds::step(ge, hv * (1.0f / 6.0f), dh_ddt * (1.0f / 6.0f));
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = dual_mul_add(h6, acc.get(i), Seeds::state(ge, i, xv[i]));
}
template <int N, class Coeffs>
AC_DI void rk4_step_seeded(const DevParams& P, Coeffs& coeffs, int g, const float xv[13], const float uv[7],
                           float hv, float dh_ddt, Dual<N> xo[13]) {
    RegAcc<N> acc;
    rk4_step_seeded<N>(P, coeffs, g, xv, uv, hv, dh_ddt, xo, acc);
}

}  // namespace ac

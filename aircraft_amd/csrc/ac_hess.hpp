// ac_hess.hpp — second-order step sensitivities: the Hessian of  lambda . F(x, u, dt)  per unit.
//
// What IPOPT asks of the reference's NLP as `nlp_hess_l` (the defect rows x_{k+1} - F(x_k, u_k, dt_k) of
// control/base.py:279-280 contribute  -lambda_k' d2F/dz2  per node; todo.md:102 names it the largest time sink).
// Exact second-order forward mode of the same arithmetic as the step kernels: every scalar carries
//   v, a = d/d(alpha), d[j] = d/d(beta_j), h[j] = d2/d(alpha)d(beta_j)
// for one "outer" direction alpha and N "inner" directions beta_j.  A unit's direction pairs (upper triangle of
// 16 x 16, by blocks of N) are spread over N (16/N)(16/N + 1)/2 - 1 lanes (direction layout as in SeedsT: 0-9 = v, q, omega; 10-13 = active controls; 14 = dt; 15 unused);
// the primal is recomputed by every lane.  Analytic force models only (default / linear / poly / quadrotor); one RK4
// sub-step (what every MPC driver of the reference uses).  d2F/dp.. = 0 and dead controls have no rows: the caller's
// output is zero-filled and only the active 15 x 15 block is written.
#pragma once
#include "ac_kernels_analytic.hpp"

namespace ac {

template <int N> struct Jet2 {
    float v, a, d[N], h[N];
    AC_DI Jet2() {}
    AC_DI Jet2(float x) : v(x), a(0.f) {  // NOLINT (implicit on purpose)
#pragma unroll
        for (int j = 0; j < N; ++j) { d[j] = 0.f; h[j] = 0.f; }
    }
};

template <int N> AC_DI float value_of(const Jet2<N>& x) { return x.v; }

// r = f(x) given f, f', f'' at x.v
template <int N> AC_DI Jet2<N> jet_unary(const Jet2<N>& x, float f0, float f1, float f2) {
    Jet2<N> r; r.v = f0; r.a = f1 * x.a;
    const float f2a = f2 * x.a;
#pragma unroll
    for (int j = 0; j < N; ++j) { r.d[j] = f1 * x.d[j]; r.h[j] = fmaf(f2a, x.d[j], f1 * x.h[j]); }
    return r;
}

template <int N> AC_DI Jet2<N> operator+(const Jet2<N>& x, const Jet2<N>& y) {
    Jet2<N> r; r.v = x.v + y.v; r.a = x.a + y.a;
#pragma unroll
    for (int j = 0; j < N; ++j) { r.d[j] = x.d[j] + y.d[j]; r.h[j] = x.h[j] + y.h[j]; }
    return r;
}
template <int N> AC_DI Jet2<N> operator-(const Jet2<N>& x, const Jet2<N>& y) {
    Jet2<N> r; r.v = x.v - y.v; r.a = x.a - y.a;
#pragma unroll
    for (int j = 0; j < N; ++j) { r.d[j] = x.d[j] - y.d[j]; r.h[j] = x.h[j] - y.h[j]; }
    return r;
}
template <int N> AC_DI Jet2<N> operator-(const Jet2<N>& x) {
    Jet2<N> r; r.v = -x.v; r.a = -x.a;
#pragma unroll
    for (int j = 0; j < N; ++j) { r.d[j] = -x.d[j]; r.h[j] = -x.h[j]; }
    return r;
}
template <int N> AC_DI Jet2<N> operator*(const Jet2<N>& x, const Jet2<N>& y) {
    Jet2<N> r; r.v = x.v * y.v; r.a = fmaf(x.a, y.v, x.v * y.a);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        r.d[j] = fmaf(x.d[j], y.v, x.v * y.d[j]);
        r.h[j] = fmaf(x.h[j], y.v, fmaf(x.a, y.d[j], fmaf(x.d[j], y.a, x.v * y.h[j])));
    }
    return r;
}
template <int N> AC_DI Jet2<N> operator*(const Jet2<N>& x, float s) {
    Jet2<N> r; r.v = x.v * s; r.a = x.a * s;
#pragma unroll
    for (int j = 0; j < N; ++j) { r.d[j] = x.d[j] * s; r.h[j] = x.h[j] * s; }
    return r;
}
template <int N> AC_DI Jet2<N> operator*(float s, const Jet2<N>& x) { return x * s; }
template <int N> AC_DI Jet2<N> operator+(const Jet2<N>& x, float s) { Jet2<N> r = x; r.v += s; return r; }
template <int N> AC_DI Jet2<N> operator+(float s, const Jet2<N>& x) { Jet2<N> r = x; r.v += s; return r; }
template <int N> AC_DI Jet2<N> operator-(const Jet2<N>& x, float s) { Jet2<N> r = x; r.v -= s; return r; }
template <int N> AC_DI Jet2<N> operator-(float s, const Jet2<N>& x) { Jet2<N> r = -x; r.v += s; return r; }
template <int N> AC_DI Jet2<N> jet_recip(const Jet2<N>& y) {
    const float i1 = 1.0f / y.v, i2 = i1 * i1;
    return jet_unary(y, i1, -i2, 2.0f * i2 * i1);
}
template <int N> AC_DI Jet2<N> operator/(const Jet2<N>& x, const Jet2<N>& y) { return x * jet_recip(y); }
template <int N> AC_DI Jet2<N> operator/(const Jet2<N>& x, float s) { return x * (1.0f / s); }
template <int N> AC_DI Jet2<N> operator/(float s, const Jet2<N>& y) { return jet_recip(y) * s; }

template <int N> AC_DI Jet2<N> m_sqrt(const Jet2<N>& x) {
    const float r = sqrtf(x.v), g = 0.5f / r;
    return jet_unary(x, r, g, -0.5f * g / x.v);
}
template <int N> AC_DI Jet2<N> m_asin(const Jet2<N>& x) {
    const float c2 = fmaf(-x.v, x.v, 1.0f), g = 1.0f / sqrtf(c2);
    return jet_unary(x, asinf(x.v), g, x.v * g / c2);
}
template <int N> AC_DI Jet2<N> m_exp(const Jet2<N>& x) {
    const float e = expf(x.v);
    return jet_unary(x, e, e, e);
}
template <int N> AC_DI Jet2<N> m_fabs(const Jet2<N>& x) { return x.v < 0.f ? -x : x; }
template <int N> AC_DI Jet2<N> m_atan2(const Jet2<N>& y, const Jet2<N>& x) {
    const float i2 = 1.0f / fmaf(x.v, x.v, y.v * y.v);
    const float ty = x.v * i2, tx = -y.v * i2;                 // d theta / dy, d theta / dx
    const float tyy = -2.0f * x.v * y.v * i2 * i2, txx = -tyy;  // second derivatives
    const float txy = (y.v * y.v - x.v * x.v) * i2 * i2;
    Jet2<N> r; r.v = atan2f(y.v, x.v); r.a = fmaf(ty, y.a, tx * x.a);
    const float ca = fmaf(tyy, y.a, txy * x.a);  // coefficient of y.d[j]
    const float cb = fmaf(txy, y.a, txx * x.a);  // coefficient of x.d[j]
#pragma unroll
    for (int j = 0; j < N; ++j) {
        r.d[j] = fmaf(ty, y.d[j], tx * x.d[j]);
        r.h[j] = fmaf(ty, y.h[j], fmaf(tx, x.h[j], fmaf(ca, y.d[j], cb * x.d[j])));
    }
    return r;
}

// ---- coefficient providers of the Hessian kernel ------------------------------------------------------------------
template <int MODEL> struct HessAnalyticCoeffs : AnalyticCoeffs<MODEL> {
    AC_DI HessAnalyticCoeffs(const float*, const UnitAddr&) {}
    AC_DI void set_stage(int) {}
};

// MLP surrogate: chain rule through (y, J, T) of the stage, stored by k_nn_stage_tensors (ac_hess_nn.hpp) as
// [4][126] floats per unit: y[6], J[6][5], T[6][15] (symmetric pairs p <= q).
struct HessTensorCoeffs {
    static constexpr int kModel = AC_MODEL_NN;
    const float* __restrict__ base;
    long blk;
    int stage;
    AC_DI HessTensorCoeffs(const float* tensors, const UnitAddr& ua) : base(tensors + ua.off(4 * 126)), blk(ua.blk), stage(0) {}
    AC_DI void set_stage(int s) { stage = s; }
    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Jet2<N>>& a, const Jet2<N>*, const Jet2<N> u[7],
                          Jet2<N> C[6]) const {
        const float* t = base + (long)stage * 126 * blk;
        const Jet2<N> in[5] = {a.qbar, a.alpha, a.beta, u[0], u[1]};
        Jet2<N> z[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) * (1.0f / P.mlp_in_std[j]);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float os = P.mlp_out_std[k];
            Jet2<N> r(fmaf(t[(long)k * blk], os, P.mlp_out_mean[k]));
            float ta[5];  // (T z.a)_q = sum_p T[p][q] z_p.a
#pragma unroll
            for (int q = 0; q < 5; ++q) ta[q] = 0.f;
#pragma unroll
            for (int p = 0; p < 5; ++p)
#pragma unroll
                for (int q = p; q < 5; ++q) {
                    const float tpq = t[(long)(36 + k * 15 + (p * 5 - p * (p - 1) / 2 + (q - p))) * blk];
                    ta[q] = fmaf(tpq, z[p].a, ta[q]);
                    if (q != p) ta[p] = fmaf(tpq, z[q].a, ta[p]);
                }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float Jkj = t[(long)(6 + k * 5 + j) * blk];
                r.a = fmaf(Jkj, z[j].a, r.a);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    r.d[i] = fmaf(Jkj, z[j].d[i], r.d[i]);
                    r.h[i] = fmaf(Jkj, z[j].h[i], fmaf(ta[j], z[j].d[i], r.h[i]));
                }
            }
            r.a *= os;
#pragma unroll
            for (int i = 0; i < N; ++i) { r.d[i] *= os; r.h[i] *= os; }
            C[k] = r;
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
};
template <int MODEL> struct HessProvider { typedef HessAnalyticCoeffs<MODEL> type; };
template <> struct HessProvider<AC_MODEL_NN> { typedef HessTensorCoeffs type; };

// direction -> row/column of the 21 x 21 matrix over z = (x[13], u[7], dt); -1 = no such row
template <bool QUAD> AC_DI int hess_index(int dir) {
    if (dir < 10) return 3 + dir;
    if (dir < 13) return 13 + (dir - 10);
    if (dir == 13) return QUAD ? 16 : 19;
    if (dir == 14) return 20;
    return -1;
}

template <int N, bool QUAD> AC_DI Jet2<N> hess_seed(int a, int g, int dir, float v, float scale = 1.f) {
    Jet2<N> r(v);
    r.a = (dir >= 0 && dir == a) ? scale : 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) r.d[j] = (dir >= 0 && dir == N * g + j) ? scale : 0.f;
    return r;
}

template <int N> struct HessTasks { static constexpr int value = N * (16 / N) * (16 / N + 1) / 2 - 1; };  // lanes per unit

// H[za][zb][unit] = sum_i lambda_i d2F_i / dz_a dz_b, active block only (the ABI zero-fills the rest beforehand).
template <int MODEL, int N>
__global__ __launch_bounds__(kBlock) void k_step_hess(const DevParams P, const float* __restrict__ X,
                                                      const float* __restrict__ U, float dt,
                                                      const float* __restrict__ dt_per_unit,
                                                      const float* __restrict__ Lam,
                                                      const float* __restrict__ stage_tensors, long n, long blk,
                                                      float* __restrict__ Hout) {
    constexpr bool QUAD = MODEL == AC_MODEL_QUAD;
    constexpr int G = 16 / N;  // inner groups
    typedef Jet2<N> T;
    // The Hessian is symmetric: only the tasks (a, g) whose inner block reaches the diagonal (N g + N - 1 >= a) run, and
    // an off-diagonal block is written to both triangles.  Tasks are ordered by the block A = a / N of the outer
    // direction, then r = a % N, then g = A .. G-1; the unused direction 15 is the last task and is dropped.
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long unit_raw = gt / HessTasks<N>::value;
    int rem = (int)(gt % HessTasks<N>::value), A = 0;
    while (rem >= N * (G - A)) { rem -= N * (G - A); ++A; }
    const int a = N * A + rem / (G - A), g = A + rem % (G - A);
    const bool live = unit_raw < n;
    const long unit = live ? unit_raw : n - 1;
    const UnitAddr ua(unit, blk);
    float xv[13], uv[7], lam[13];
    load_rows<13>(X, ua, xv);
    load_rows<7>(U, ua, uv);
    load_rows<13>(Lam, ua, lam);
    const float hv = dt_per_unit ? dt_per_unit[unit] : dt;

    // The seeds are 0/1 patterns of (a, g): they are rebuilt where they are used instead of being carried through the
    // RK4 loop (13 + 7 jets of registers otherwise), with the lane ids hidden from loop-invariant hoisting.
    auto seed_x = [&](int aa, int gg, int i) { return hess_seed<N, QUAD>(aa, gg, i >= 3 ? i - 3 : -1, xv[i]); };
    typename HessProvider<MODEL>::type coeffs(stage_tensors, ua);
    T acc[13], xs[13], k[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { xs[i] = seed_x(a, g, i); acc[i] = T(0.f); }
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        int aa = a, gg = g;
        asm volatile("" : "+v"(aa), "+v"(gg));
        {
            T u[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int dir = QUAD ? (i < 4 ? 10 + i : -1) : ((i < 3) ? 10 + i : (i == 6 ? 13 : -1));
                u[i] = hess_seed<N, QUAD>(aa, gg, dir, uv[i]);
            }
            coeffs.set_stage(s);
            state_derivative(P, coeffs, xs, u, k);
        }
        const float wsum = (s == 1 || s == 2) ? 2.0f : 1.0f;
        const float cnext = (s == 2) ? 1.0f : 0.5f;
        const T hs = hess_seed<N, QUAD>(aa, gg, 14, hv * cnext, cnext);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            acc[i] = acc[i] + wsum * k[i];
            xs[i] = seed_x(aa, gg, i) + hs * k[i];
        }
    }
    T xo[13];
    {
        int aa = a, gg = g;
        asm volatile("" : "+v"(aa), "+v"(gg));
        const T h6 = hess_seed<N, QUAD>(aa, gg, 14, hv * (1.0f / 6.0f), 1.0f / 6.0f);
#pragma unroll
        for (int i = 0; i < 13; ++i) xo[i] = seed_x(aa, gg, i) + h6 * acc[i];
    }
    if (P.p.normalise) normalise_q(xo);

    const int za = hess_index<QUAD>(a);
    if (!live || za < 0) return;
    float* Hu = Hout + ua.off(441);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int zb = hess_index<QUAD>(N * g + j);
        if (zb < 0) continue;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 13; ++i) s = fmaf(lam[i], xo[i].h[j], s);
        Hu[((long)za * 21 + zb) * blk] = s;
        if (g != A) Hu[((long)zb * 21 + za) * blk] = s;  // the mirrored task does not run
    }
}

// ---- composition across RK4 sub-steps (physical_integration_substeps > 1: dynamics/base.py:463-474, default 10) ----------
// F = N o S_ns o ... o S_1 with S_s : w_s = (x_{s-1}, u, h = dt / ns) -> x_s.  For phi = lambda . F:
//   d2 phi / dz2 = sum_s T_s' H_s T_s,   H_s = sum_i mu_{s,i} d2 S_i / dw2 at w_s   (the single-step kernel above),
//   T_s = d w_s / dz = [ X_{s-1} ; (0 I 0) ; (0 0 1/ns) ],   X_s = d x_s / dz = A_s X_{s-1} + (0 | B_s | c_s / ns),
//   mu_{s-1} = A_s' mu_s   (mu_ns = lambda; the final normalisation belongs to the last sub-step's kernels).
// All arrays in the blocked component-major layout of the inputs (UnitAddr), one lane per (unit, column).

// XZn [13][21] = A [13][13] . XZ [13][21] + (0 | B [13][7] | c [13] * inv_ns); XZ == nullptr: X_0 = (I 0 0)
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_hess_chain(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const float* __restrict__ c, const float* __restrict__ XZ,
                                                       float inv_ns, long n, long blk, float* __restrict__ XZn) {
    const long unit = (long)blockIdx.x * kBlock + threadIdx.x;
    const int b = blockIdx.y;  // column of z: 0..20
    if (unit >= n) return;
    const UnitAddr ua(unit, blk);
    const float* Au = A + ua.off(169);
    float col[13];
    if (XZ) {
        const float* Xu = XZ + ua.off(273);
#pragma unroll
        for (int k = 0; k < 13; ++k) col[k] = Xu[(long)(k * 21 + b) * blk];
    } else {
#pragma unroll
        for (int k = 0; k < 13; ++k) col[k] = (k == b) ? 1.f : 0.f;
    }
    float* out = XZn + ua.off(273);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 13; ++k) s = fmaf(Au[(long)(i * 13 + k) * blk], col[k], s);
        if (b >= 13 && b < 20) s += Bm[ua.off(91) + (long)(i * 7 + (b - 13)) * blk];
        if (b == 20) s = fmaf(c[ua.off(13) + (long)i * blk], inv_ns, s);
        out[(long)(i * 21 + b) * blk] = s;
    }
}

// mu_out [13] = A' mu [13]
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_hess_adjoint(const float* __restrict__ A, const float* __restrict__ mu, long n,
                                                         long blk, float* __restrict__ mu_out) {
    const long unit = (long)blockIdx.x * kBlock + threadIdx.x;
    if (unit >= n) return;
    const UnitAddr ua(unit, blk);
    const float* Au = A + ua.off(169);
    float m[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) m[i] = mu[ua.off(13) + (long)i * blk];
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 13; ++i) s = fmaf(Au[(long)(i * 13 + j) * blk], m[i], s);
        mu_out[ua.off(13) + (long)j * blk] = s;
    }
}

// Hout [21][21] += T' Hs T  with T = [XZ ; (0 I 0) ; (0 0 inv_ns)];  XZ == nullptr: X_0 = (I 0 0).  Lane = (unit, column b).
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_hess_accum(const float* __restrict__ Hs, const float* __restrict__ XZ,
                                                       float inv_ns, long n, long blk, float* __restrict__ Hout) {
    const long unit = (long)blockIdx.x * kBlock + threadIdx.x;
    const int b = blockIdx.y;
    if (unit >= n) return;
    const UnitAddr ua(unit, blk);
    const float* Hu = Hs + ua.off(441);
    const float* Xu = XZ ? XZ + ua.off(273) : nullptr;
    // column b of T
    float t[21];
#pragma unroll
    for (int k = 0; k < 13; ++k) t[k] = Xu ? Xu[(long)(k * 21 + b) * blk] : ((k == b) ? 1.f : 0.f);
#pragma unroll
    for (int k = 13; k < 20; ++k) t[k] = (k == b) ? 1.f : 0.f;
    t[20] = (b == 20) ? inv_ns : 0.f;
    // m = Hs t
    float m[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 21; ++k) s = fmaf(Hu[(long)(i * 21 + k) * blk], t[k], s);
        m[i] = s;
    }
    // (T' m)_a
    float* out = Hout + ua.off(441);
#pragma unroll
    for (int a = 0; a < 21; ++a) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 13; ++k) s = fmaf(Xu ? Xu[(long)(k * 21 + a) * blk] : ((k == a) ? 1.f : 0.f), m[k], s);
        if (a >= 13 && a < 20) s += m[a];
        if (a == 20) s = fmaf(m[20], inv_ns, s);
        out[(long)(a * 21 + b) * blk] += s;
    }
}

// dt_out = dt_in * inv_ns (per-unit step of the sub-steps)
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_scale_rows(const float* __restrict__ in, float scale, long count,
                                                       float* __restrict__ out) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) out[i] = in[i] * scale;
}

}  // namespace ac

// dyn_host.cpp — the kernels' own arithmetic headers (ac_math.hpp, ac_dynamics.hpp) compiled for the HOST (g++,
// -DAC_HOST_CHECK) behind a small C API, so that `pytest -m "not gpu"` can check the device math (fp32 forward-mode
// tangents of the RK4 step, lane group by lane group) against the float64 oracle without a GPU.  TEST INFRASTRUCTURE:
// nothing in aircraft_amd loads this.
#define AC_HOST_CHECK 1
#include "../../aircraft_amd/csrc/ac_adjoint.hpp"

using namespace ac;

namespace {

template <int N, bool QUAD> void scatter(int g, const Dual<N> x[13], float* A, float* B, float* c) {
    for (int j = 0; j < N; ++j) {
        const int d = N * g + j;
        for (int i = 0; i < 13; ++i) {
            if (d < 10) A[i * 13 + 3 + d] = x[i].d[j];
            else if (d < 13) B[i * 7 + (d - 10)] = x[i].d[j];
            else if (d == 13) B[i * 7 + (QUAD ? 3 : 6)] = x[i].d[j];
            else if (d == 14) c[i] = x[i].d[j];
        }
    }
}

template <int MODEL, int N> void unit_step_sens(const DevParams& P, const float xv[13], const float uv[7], float dt,
                                                float xn[13], float A[169], float B[91], float c[13]) {
    for (int i = 0; i < 169; ++i) A[i] = 0.f;
    for (int i = 0; i < 91; ++i) B[i] = 0.f;
    for (int i = 0; i < 3; ++i) A[i * 13 + i] = 1.f;  // dF/dp = [I; 0]
    AnalyticCoeffs<MODEL> coeffs;
    for (int g = 0; g < 16 / N; ++g) {
        Dual<N> x[13];
        rk4_step_seeded<N>(P, coeffs, g, xv, uv, dt, 1.0f, x);
        if (P.p.normalise) normalise_q(x);
        scatter<N, MODEL == AC_MODEL_QUAD>(g, x, A, B, c);
        for (int i = 0; i < 13; ++i) xn[i] = x[i].v;
    }
}

template <int MODEL, int N> void unit_deriv_sens(const DevParams& P, const float xv[13], const float uv[7],
                                                 float xd[13], float Fx[169], float Fu[91]) {
    for (int i = 0; i < 169; ++i) Fx[i] = 0.f;
    for (int i = 0; i < 91; ++i) Fu[i] = 0.f;
    float cdummy[13];
    AnalyticCoeffs<MODEL> coeffs;
    for (int g = 0; g < 16 / N; ++g) {
        Dual<N> xs[13], u[7], k[13];
        for (int i = 0; i < 13; ++i) xs[i] = SeedsT<N>::state(g, i, xv[i]);
        SeedsT<N>::template controls<MODEL == AC_MODEL_QUAD>(g, uv, u);
        state_derivative(P, coeffs, xs, u, k);
        scatter<N, MODEL == AC_MODEL_QUAD>(g, k, Fx, Fu, cdummy);
        for (int i = 0; i < 13; ++i) xd[i] = k[i].v;
    }
}

template <int MODEL, int N>
void run(const DevParams& P, int what, const float* X, const float* U, float dt, long n, float* Xn, float* A, float* B,
         float* c) {
    for (long u = 0; u < n; ++u) {
        float xv[13], uv[7], xn[13], a[169], b[91], cc[13] = {0};
        for (int i = 0; i < 13; ++i) xv[i] = X[i * n + u];
        for (int i = 0; i < 7; ++i) uv[i] = U[i * n + u];
        if (what == 0) unit_step_sens<MODEL, N>(P, xv, uv, dt, xn, a, b, cc);
        else unit_deriv_sens<MODEL, N>(P, xv, uv, xn, a, b);
        for (int i = 0; i < 13; ++i) Xn[i * n + u] = xn[i];
        for (int i = 0; i < 169; ++i) A[i * n + u] = a[i];
        for (int i = 0; i < 91; ++i) B[i * n + u] = b[i];
        if (c) for (int i = 0; i < 13; ++i) c[i * n + u] = cc[i];
    }
}

// gradient of lam . F over z = (x[13], u[7], dt) by the reverse sweep in plain floats, and its Hessian by the same sweep in
// duals (N directions at a time) — ac_adjoint.hpp
template <int MODEL, int N> void unit_adjoint(const DevParams& P, const float xv[13], const float uv[7], float dt, const float lam[13],
                                              float grad[21], float Hm[441]) {
    AdjAnalyticCoeffs<MODEL> coeffs;
    {
        float xo[13], gx[13], gu[7], gh;
        rk4_vjp<float>(P, coeffs, xv, uv, dt, lam, xo, gx, gu, gh);
        for (int i = 0; i < 13; ++i) grad[i] = gx[i];
        for (int i = 0; i < 7; ++i) grad[13 + i] = gu[i];
        grad[20] = gh;
    }
    for (int i = 0; i < 441; ++i) Hm[i] = 0.f;
    for (int g = 0; g < (21 + N - 1) / N; ++g) {
        typedef Dual<N> T;
        T x[13], u[7], h(dt);
        for (int i = 0; i < 13; ++i) x[i] = T(xv[i]);
        for (int i = 0; i < 7; ++i) u[i] = T(uv[i]);
        for (int j = 0; j < N; ++j) {
            const int z = N * g + j;
            if (z < 13) x[z].d[j] = 1.f;
            else if (z < 20) u[z - 13].d[j] = 1.f;
            else if (z == 20) h.d[j] = 1.f;
        }
        T xo[13], gx[13], gu[7], gh;
        rk4_vjp<T>(P, coeffs, x, u, h, lam, xo, gx, gu, gh);
        for (int j = 0; j < N; ++j) {
            const int z = N * g + j;
            if (z > 20) continue;
            for (int i = 0; i < 13; ++i) Hm[i * 21 + z] = gx[i].d[j];
            for (int i = 0; i < 7; ++i) Hm[(13 + i) * 21 + z] = gu[i].d[j];
            Hm[20 * 21 + z] = gh.d[j];
        }
    }
}

}  // namespace

extern "C" int host_dyn_adjoint(const ac_params* p, const float* linear_W, const float* poly_coef, const float* poly_intercept, int N,
                                const float* X, const float* U, float dt, const float* Lam, long n, float* grad /*[21][n]*/,
                                float* Hm /*[21][21][n]*/) {
    DevParams P{};
    P.p = *p;
    if (linear_W) for (int i = 0; i < 36; ++i) P.linear_W[i] = linear_W[i];
    alignas(64) static thread_local float tab[kPolyTabFloats];
    if (poly_coef && poly_intercept) {
        float gradt[6 * 4 * 15], hesst[6 * 10 * 5];
        poly_gradient_tables(poly_coef, gradt);
        poly_hessian_tables(gradt, hesst);
        poly_pack_tables(poly_coef, poly_intercept, gradt, hesst, tab);
        P.poly_tab = tab;
    }
    if (P.p.substeps > 1) return -1;
    for (long u = 0; u < n; ++u) {
        float xv[13], uv[7], lam[13], g[21], H[441];
        for (int i = 0; i < 13; ++i) { xv[i] = X[i * n + u]; lam[i] = Lam[i * n + u]; }
        for (int i = 0; i < 7; ++i) uv[i] = U[i * n + u];
        bool done = false;
#define AC_CASE(M_, N_) if (!done && P.p.model_kind == M_ && N == N_) { unit_adjoint<M_, N_>(P, xv, uv, dt, lam, g, H); done = true; }
        AC_CASE(AC_MODEL_DEFAULT, 1) AC_CASE(AC_MODEL_DEFAULT, 2) AC_CASE(AC_MODEL_DEFAULT, 4)
        AC_CASE(AC_MODEL_LINEAR, 1) AC_CASE(AC_MODEL_LINEAR, 2) AC_CASE(AC_MODEL_QUAD, 2)
        AC_CASE(AC_MODEL_POLY, 1) AC_CASE(AC_MODEL_POLY, 2)
#undef AC_CASE
        if (!done) return -2;
        for (int i = 0; i < 21; ++i) grad[i * n + u] = g[i];
        for (int i = 0; i < 441; ++i) Hm[i * n + u] = H[i];
    }
    return 0;
}

// what: 0 = one RK4 step with A, B, c (substeps must be 1); 1 = f with df/dx, df/du.  N = tangent directions per lane
// group (2, 4 or 8).  Arrays component-major like the device ABI: X [13][n], A [13][13][n], ...
extern "C" int host_dyn_sens(const ac_params* p, const float* linear_W, const float* poly_coef, const float* poly_intercept,
                             int what, int N, const float* X, const float* U, float dt, long n, float* Xn, float* A,
                             float* B, float* c) {
    DevParams P{};
    P.p = *p;
    if (linear_W) for (int i = 0; i < 36; ++i) P.linear_W[i] = linear_W[i];
    alignas(64) static thread_local float tab[kPolyTabFloats];
    if (poly_coef && poly_intercept) {
        float grad[6 * 4 * 15], hess[6 * 10 * 5];
        poly_gradient_tables(poly_coef, grad);
        poly_hessian_tables(grad, hess);
        poly_pack_tables(poly_coef, poly_intercept, grad, hess, tab);
        P.poly_tab = tab;
    }
    if (what == 0 && P.p.substeps > 1) return -1;
#define AC_CASE(M_, N_) if (P.p.model_kind == M_ && N == N_) { run<M_, N_>(P, what, X, U, dt, n, Xn, A, B, c); return 0; }
    AC_CASE(AC_MODEL_DEFAULT, 2) AC_CASE(AC_MODEL_DEFAULT, 4) AC_CASE(AC_MODEL_DEFAULT, 8)
    AC_CASE(AC_MODEL_LINEAR, 2) AC_CASE(AC_MODEL_LINEAR, 4) AC_CASE(AC_MODEL_LINEAR, 8)
    AC_CASE(AC_MODEL_POLY, 2) AC_CASE(AC_MODEL_POLY, 4) AC_CASE(AC_MODEL_POLY, 8)
    AC_CASE(AC_MODEL_QUAD, 2) AC_CASE(AC_MODEL_QUAD, 4) AC_CASE(AC_MODEL_QUAD, 8)
#undef AC_CASE
    return -2;
}

// Value, gradient and second derivatives of fit k at the points F [4][n] through the PACKED tables and the device's own
// streaming evaluators (poly_value_grad, poly_hess): out [15][n] = val, g[4], h[10] (entry order: PolyTab::hess_row).
extern "C" int host_poly_point(const float* poly_coef, const float* poly_intercept, int k, const float* F, long n, float* out) {
    DevParams P{};
    alignas(64) static thread_local float tab[kPolyTabFloats];
    float grad[6 * 4 * 15], hess[6 * 10 * 5];
    poly_gradient_tables(poly_coef, grad);
    poly_hessian_tables(grad, hess);
    poly_pack_tables(poly_coef, poly_intercept, grad, hess, tab);
    P.poly_tab = tab;
    if (k < 0 || k > 5) return -1;
    for (long u = 0; u < n; ++u) {
        const float f[4] = {F[u], F[n + u], F[2 * n + u], F[3 * n + u]};
        const int ks[1] = {k};
        float val[1], g[1][4], h[1][10];
        poly_value_grad<1>(P, ks, f, val, g);
        poly_hess<1>(P, ks, f, h);
        out[u] = val[0];
        for (int v = 0; v < 4; ++v) out[(1 + v) * n + u] = g[0][v];
        for (int e = 0; e < 10; ++e) out[(5 + e) * n + u] = h[0][e];
    }
    return 0;
}

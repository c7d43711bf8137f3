// aircraft_oracle.cpp — float64 CPU oracle for the AIrcraft MPC-rollout hot path.
//
// TEST INFRASTRUCTURE, NOT PRODUCT (see aircraft_oracle.h).  A restatement — written
// from the reference's arithmetic, not copied from its CasADi graph code — of
//   src/aircraft/dynamics/base.py            (6-DoF kinematics, RK4, sub-stepping)
//   src/aircraft/dynamics/aircraft.py        (coefficients -> forces/moments, inertia)
//   src/aircraft/dynamics/coefficient_models.py  (default / linear / nn / poly)
//   src/aircraft/surrogates/models.py        (ScaledModel MLP)
// Third-party conventions that are not in the reference tree (liecasadi 0.0.6
// Quaternion: xyzw storage, Hamilton product, inverse = conj/|q|^2; casadi 3.6.7
// sign(0)=0, atan2, asin) are fixed by the simulation.h5 replay.
//
// Every function is templated on the scalar type so the SAME arithmetic runs on
// double (values) and on Dual<N> (forward-mode AD) — the latter is what
// ca.jacobian(state_update, .) computes in the reference (control/aircraft.py:85-95).
//
// Build: see oracle/Makefile (g++ -O2, no fast-math, OpenMP over the batch).

#include "aircraft_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------------------
// Forward-mode dual numbers
// ----------------------------------------------------------------------------------------
template <int N>
struct Dual {
    double v;
    double d[N];
    Dual() : v(0.0) {
        for (int i = 0; i < N; ++i) d[i] = 0.0;
    }
    Dual(double x) : v(x) {  // NOLINT (implicit on purpose)
        for (int i = 0; i < N; ++i) d[i] = 0.0;
    }
};

template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r;
}
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r;
}
template <int N> inline Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r;
}
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r;
}
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v / b.v;  // the value part is computed exactly as the plain-double path computes it
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { return (-a) + b; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) {
    Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r;
}
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) {
    Dual<N> r; r.v = a.v / b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r;
}
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }

inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Dual<N>& x) { return x.v; }

inline double m_sqrt(double x) { return std::sqrt(x); }
template <int N> inline Dual<N> m_sqrt(const Dual<N>& a) {
    Dual<N> r; r.v = std::sqrt(a.v); const double g = 0.5 / r.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r;
}
inline double m_atan2(double y, double x) { return std::atan2(y, x); }
template <int N> inline Dual<N> m_atan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; r.v = std::atan2(y.v, x.v); const double den = 1.0 / (x.v * x.v + y.v * y.v);
    for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
    return r;
}
inline double m_asin(double x) { return std::asin(x); }
template <int N> inline Dual<N> m_asin(const Dual<N>& a) {
    Dual<N> r; r.v = std::asin(a.v); const double g = 1.0 / std::sqrt(1.0 - a.v * a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r;
}
inline double m_exp(double x) { return std::exp(x); }
template <int N> inline Dual<N> m_exp(const Dual<N>& a) {
    Dual<N> r; r.v = std::exp(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v; return r;
}
inline double m_fabs(double x) { return std::fabs(x); }
template <int N> inline Dual<N> m_fabs(const Dual<N>& a) { return a.v < 0.0 ? -a : a; }
// casadi sign(): -1, 0, +1; its derivative is identically zero
inline double sign_of(double x) { return (x > 0.0) - (x < 0.0); }

// ----------------------------------------------------------------------------------------
// Quaternions, xyzw (liecasadi convention; dynamics/base.py:88, 149-159, 284-295)
// ----------------------------------------------------------------------------------------
template <class T> struct Q4 { T x, y, z, w; };

// Hamilton product a (x) b = [a_w b_v + b_w a_v + a_v x b_v , a_w b_w - a_v . b_v]
template <class T> inline Q4<T> qmul(const Q4<T>& a, const Q4<T>& b) {
    Q4<T> r;
    r.x = a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y);
    r.y = a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z);
    r.z = a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x);
    r.w = a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z);
    return r;
}
// inverse = conjugate / |q|^2  (the simulation.h5 replay discriminates this from a bare conjugate)
template <class T> inline Q4<T> qinv(const Q4<T>& q) {
    const T n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    Q4<T> r; r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; r.w = q.w / n2;
    return r;
}

// ----------------------------------------------------------------------------------------
// Constants derived once per parameter set
// ----------------------------------------------------------------------------------------
struct Derived {
    double I[3][3];     // inertia about the shifted CoM, dynamics/aircraft.py:168-187
    double Iinv[3][3];  // dynamics/base.py:139-144
    int poly_terms[34][3];  // variable indices of each monomial (-1 = unused slot)
    int poly_deg[34];
    // MLP weights transposed to [in][out] for a k-outer / n-inner evaluation order
    std::vector<double> Wt[ORACLE_MAX_LAYERS];
};

void make_derived(const oracle_params& P, Derived& D) {
    const double x = P.com[0], y = P.com[1], z = P.com[2], m = P.mass;
    const double I0[3][3] = {{P.Ixx, 0.0, P.Ixz}, {0.0, P.Iyy, 0.0}, {P.Ixz, 0.0, P.Izz}};
    const double K[3][3] = {{y * y + z * z, -x * y, -x * z}, {-y * x, x * x + z * z, -y * z}, {-z * x, -z * y, x * x + y * y}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)  // the parallel-axis shift is Aircraft's (aircraft.py:168-187); Quadrotor's tensor is fixed
            D.I[i][j] = I0[i][j] + (P.model_kind == ORACLE_MODEL_QUAD ? 0.0 : m * K[i][j]);
    const double (*a)[3] = D.I;
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    D.Iinv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det;
    D.Iinv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    D.Iinv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    D.Iinv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
    D.Iinv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
    D.Iinv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    D.Iinv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det;
    D.Iinv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    D.Iinv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    // sklearn PolynomialFeatures(degree=3, include_bias=False) over 4 features:
    // itertools.combinations_with_replacement(range(4), d) for d = 1, 2, 3  
    int t = 0;
    for (int i = 0; i < 4; ++i) { D.poly_terms[t][0] = i; D.poly_terms[t][1] = D.poly_terms[t][2] = -1; D.poly_deg[t++] = 1; }
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) { D.poly_terms[t][0] = i; D.poly_terms[t][1] = j; D.poly_terms[t][2] = -1; D.poly_deg[t++] = 2; }
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j)
            for (int k = j; k < 4; ++k) { D.poly_terms[t][0] = i; D.poly_terms[t][1] = j; D.poly_terms[t][2] = k; D.poly_deg[t++] = 3; }
    if (P.model_kind == ORACLE_MODEL_NN) {
        for (int l = 0; l < P.mlp_n_layers; ++l) {
            const int nin = P.mlp_widths[l], nout = P.mlp_widths[l + 1];
            D.Wt[l].resize((size_t)nin * nout);
            for (int o = 0; o < nout; ++o)
                for (int i = 0; i < nin; ++i) D.Wt[l][(size_t)i * nout + o] = P.mlp_W[l][(size_t)o * nin + i];
        }
    }
}

// ----------------------------------------------------------------------------------------
// Coefficient models (dynamics/coefficient_models.py)
// ----------------------------------------------------------------------------------------
constexpr double kDeg = M_PI / 180.0;

// One cubic fit P_k(alpha, beta, aileron, elevator); coefficient_models.py:112-114
template <class T> inline T poly_eval(const oracle_params& P, const Derived& D, int k, const T f[4]) {
    T acc = T(P.poly_intercept[k]);
    for (int t = 0; t < 34; ++t) {
        T term = f[D.poly_terms[t][0]];
        for (int s = 1; s < D.poly_deg[t]; ++s) term = term * f[D.poly_terms[t][s]];
        acc = acc + P.poly_coef[k * 34 + t] * term;
    }
    return acc;
}

// MLP value and input-Jacobian in plain double (k-outer, n-inner; ascending-k summation per output).
// ScaledModel.forward, surrogates/models.py:143-155.  jac may be null.
void mlp_eval(const oracle_params& P, const Derived& D, const double in[5], double out[6], double jac[30]) {
    const int L = P.mlp_n_layers;
    int wmax = 6;
    for (int l = 0; l <= L; ++l) wmax = P.mlp_widths[l] > wmax ? P.mlp_widths[l] : wmax;
    const int R = jac ? 6 : 1;  // row 0 = value, rows 1..5 = d/d(input j)
    std::vector<double> a((size_t)R * wmax, 0.0), bnext((size_t)R * wmax, 0.0);
    for (int j = 0; j < 5; ++j) a[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
    if (jac)
        for (int j = 0; j < 5; ++j) a[(size_t)(1 + j) * wmax + j] = 1.0 / P.mlp_in_std[j];
    for (int l = 0; l < L; ++l) {
        const int nin = P.mlp_widths[l], nout = P.mlp_widths[l + 1];
        const double* Wt = D.Wt[l].data();
        for (int r = 0; r < R; ++r) {
            double* o = &bnext[(size_t)r * wmax];
            const double* i_ = &a[(size_t)r * wmax];
            for (int n = 0; n < nout; ++n) o[n] = (r == 0) ? P.mlp_b[l][n] : 0.0;
            for (int k = 0; k < nin; ++k) {
                const double xk = i_[k];
                const double* w = Wt + (size_t)k * nout;
                for (int n = 0; n < nout; ++n) o[n] += w[n] * xk;
            }
        }
        if (P.mlp_act[l] == 1) {
            double* h = &bnext[0];
            for (int n = 0; n < nout; ++n) h[n] = std::tanh(h[n]);
            for (int r = 1; r < R; ++r) {
                double* o = &bnext[(size_t)r * wmax];
                for (int n = 0; n < nout; ++n) o[n] *= (1.0 - h[n] * h[n]);
            }
        }
        a.swap(bnext);
    }
    for (int c = 0; c < 6; ++c) out[c] = a[c] * P.mlp_out_std[c] + P.mlp_out_mean[c];
    if (jac)
        for (int c = 0; c < 6; ++c)
            for (int j = 0; j < 5; ++j) jac[c * 5 + j] = a[(size_t)(1 + j) * wmax + c] * P.mlp_out_std[c];
}

inline void mlp_call(const oracle_params& P, const Derived& D, const double in[5], double out[6]) {
    mlp_eval(P, D, in, out, nullptr);
}
// AD rule for the MLP node: value + J * d(inputs)  (what l4casadi's Jacobian callback supplies)
template <int N> inline void mlp_call(const oracle_params& P, const Derived& D, const Dual<N> in[5], Dual<N> out[6]) {
    double x[5], y[6], J[30];
    for (int j = 0; j < 5; ++j) x[j] = in[j].v;
    mlp_eval(P, D, x, y, J);
    for (int c = 0; c < 6; ++c) {
        out[c].v = y[c];
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
            for (int j = 0; j < 5; ++j) s += J[c * 5 + j] * in[j].d[i];
            out[c].d[i] = s;
        }
    }
}

// ----------------------------------------------------------------------------------------
// Aerodynamics + rigid body
// ----------------------------------------------------------------------------------------
template <class T> struct Aero {
    T vr[3], V, alpha, beta, qbar, C[6], F[3], M[3];
};

template <class T>
void aero(const oracle_params& P, const Derived& D, const T x[13], const T u[7], Aero<T>& a) {
    const double eps = P.epsilon;
    const Q4<T> q{x[6], x[7], x[8], x[9]};
    const T* w = &x[10];
    // v_frd_rel = (q^-1 (x) (v,0) (x) q)[:3] + eps            dynamics/base.py:147-162
    const Q4<T> vq{x[3], x[4], x[5], T(0.0)};
    const Q4<T> r = qmul(qmul(qinv(q), vq), q);
    a.vr[0] = r.x + eps; a.vr[1] = r.y + eps; a.vr[2] = r.z + eps;
    const T vv = a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2];
    a.V = m_sqrt(vv + eps);                         // base.py:167
    a.alpha = m_atan2(a.vr[2], a.vr[0] + eps);      // base.py:176
    a.beta = m_asin(a.vr[1] / a.V);                 // base.py:234
    a.qbar = 0.5 * 1.225 * vv;                      // base.py:240
    if (P.model_kind == ORACLE_MODEL_QUAD) {  // dynamics/quadrotor.py:43-54; moments_frd adds com x F (base.py:253-278)
        for (int k = 0; k < 6; ++k) a.C[k] = T(0.0);
        a.F[0] = T(0.0); a.F[1] = T(0.0);
        a.F[2] = u[0] + u[1] + u[2] + u[3];
        const T Mq[3] = {u[0] - u[1] - u[2] + u[3], -u[0] - u[1] + u[2] + u[3], 0.5 * (u[0] - u[1] + u[2] - u[3])};
        a.M[0] = Mq[0] + (P.com[1] * a.F[2] - P.com[2] * a.F[1]);
        a.M[1] = Mq[1] + (P.com[2] * a.F[0] - P.com[0] * a.F[2]);
        a.M[2] = Mq[2] + (P.com[0] * a.F[1] - P.com[1] * a.F[0]);
        return;
    }
    const T da = u[0], de = u[1], dr = u[2], flaps = u[6];

    T C[6];
    switch (P.model_kind) {
        case ORACLE_MODEL_LINEAR: {  // coefficient_models.py:80-89
            const T in[5] = {a.qbar, a.alpha, a.beta, da, de};
            for (int k = 0; k < 6; ++k) {
                T s = T(0.0);
                for (int j = 0; j < 5; ++j) s = s + P.linear_W[k * 6 + j] * in[j];
                C[k] = s + P.linear_W[k * 6 + 5];
            }
            C[5] = C[5] + (-0.1) * 6.0 * dr * kDeg;
            break;
        }
        case ORACLE_MODEL_NN: {  // coefficient_models.py:91-104
            const T in[5] = {a.qbar, a.alpha, a.beta, da, de};
            mlp_call(P, D, in, C);
            C[5] = C[5] + (-0.1) * 6.0 * dr * kDeg;
            break;
        }
        case ORACLE_MODEL_POLY: {  // coefficient_models.py:106-133, aircraft.py:189-233
            const double arm = P.rudder_moment_arm;
            const T ux = a.vr[0] + eps;
            const T alpha_e = m_atan2(a.vr[2] + arm * w[1], ux);          // aircraft.py:198
            const T alpha_l = m_atan2(a.vr[2] - P.b * w[0] / 4.0, ux);    // aircraft.py:210
            const T alpha_r = m_atan2(a.vr[2] + P.b * w[0] / 4.0, ux);    // aircraft.py:221
            const T vy = a.vr[1] - arm * w[2];                            // aircraft.py:229
            const T beta_r = m_asin(vy / m_sqrt(a.vr[0] * a.vr[0] + vy * vy + a.vr[2] * a.vr[2] + eps));
            const T fm[4] = {a.alpha, a.beta, da, de};
            for (int k = 0; k < 6; ++k) C[k] = poly_eval(P, D, k, fm);
            const T fl[4] = {alpha_l, T(0.0), T(0.0), T(0.0)};
            const T fr[4] = {alpha_r, T(0.0), T(0.0), T(0.0)};
            const T czl = poly_eval(P, D, 2, fl), czr = poly_eval(P, D, 2, fr);
            C[3] = C[3] + P.b / 4.0 * (czr / 2.0 - czl / 2.0);            // coefficient_models.py:122
            const T fe[4] = {alpha_e, a.beta, da, de};
            C[4] = poly_eval(P, D, 4, fe);                                // :124-125
            const T fb[4] = {a.alpha, beta_r, da, de};
            C[5] = poly_eval(P, D, 5, fb) + 0.01 * 6.0 * dr * kDeg;       // :127-132 (+0.01, sic)
            break;
        }
        default: {  // DefaultModel, coefficient_models.py:41-78
            const T CD = 0.02 + 0.3 * (a.alpha * a.alpha);
            const T CL = 0.0 + 5.0 * a.alpha;
            C[0] = -CD;
            C[1] = -0.98 * a.beta;
            C[2] = -CL;
            C[3] = 0.08 * 4.0 * da * kDeg + (-0.05) * w[0];
            C[4] = -1.2 * 5.0 * de * kDeg + (-0.5) * w[1];
            C[5] = -0.1 * 6.0 * dr * kDeg + (-0.05) * w[2];
            break;
        }
    }
    if (P.stall_scaling) {  // aircraft.py:280-294
        const double lim = 30.0 * kDeg, steep = 10.0;
        const T sa = 1.0 / (1.0 + m_exp(steep * (m_fabs(a.alpha) - lim)));
        const T sb = 1.0 / (1.0 + m_exp(steep * (m_fabs(a.beta) - lim)));
        C[2] = C[2] * sa; C[2] = C[2] * sb; C[4] = C[4] * sa;
    }
    C[0] = C[0] + (-0.1) * flaps;  // aircraft.py:297-300
    C[2] = C[2] + (-0.6) * flaps;
    for (int k = 0; k < 6; ++k) a.C[k] = C[k];
    // forces / moments                               aircraft.py:309-330, base.py:253-278
    for (int k = 0; k < 3; ++k) a.F[k] = C[k] * a.qbar * P.S;
    a.F[0] = a.F[0] * sign_of(value_of(a.vr[0]));
    const double lever[3] = {P.b, P.c, P.b};
    T Ma[3];
    for (int k = 0; k < 3; ++k) Ma[k] = C[3 + k] * a.qbar * P.S * lever[k];
    a.M[0] = Ma[0] + (P.com[1] * a.F[2] - P.com[2] * a.F[1]);
    a.M[1] = Ma[1] + (P.com[2] * a.F[0] - P.com[0] * a.F[2]);
    a.M[2] = Ma[2] + (P.com[0] * a.F[1] - P.com[1] * a.F[0]);
}

// x_dot = [v ; F_ned/m + g ; 0.5 q (x) (w,0) ; I^-1 (M - w x I w)]     dynamics/base.py:290-406
template <class T>
void state_derivative(const oracle_params& P, const Derived& D, const T x[13], const T u[7], T xd[13]) {
    Aero<T> a;
    aero(P, D, x, u, a);
    const Q4<T> q{x[6], x[7], x[8], x[9]};
    const T* w = &x[10];
    const Q4<T> Fq{a.F[0], a.F[1], a.F[2], T(0.0)};
    const Q4<T> Fn = qmul(qmul(q, Fq), qinv(q));  // forces_ned, base.py:280-288
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    xd[3] = Fn.x / P.mass + P.gravity[0];
    xd[4] = Fn.y / P.mass + P.gravity[1];
    xd[5] = Fn.z / P.mass + P.gravity[2];
    const Q4<T> hq{0.5 * q.x, 0.5 * q.y, 0.5 * q.z, 0.5 * q.w};
    const Q4<T> wq{w[0], w[1], w[2], T(0.0)};
    const Q4<T> qd = qmul(hq, wq);  // base.py:295
    xd[6] = qd.x; xd[7] = qd.y; xd[8] = qd.z; xd[9] = qd.w;
    T Iw[3], rhs[3];
    for (int i = 0; i < 3; ++i) Iw[i] = D.I[i][0] * w[0] + D.I[i][1] * w[1] + D.I[i][2] * w[2];
    rhs[0] = a.M[0] - (w[1] * Iw[2] - w[2] * Iw[1]);
    rhs[1] = a.M[1] - (w[2] * Iw[0] - w[0] * Iw[2]);
    rhs[2] = a.M[2] - (w[0] * Iw[1] - w[1] * Iw[0]);
    for (int i = 0; i < 3; ++i) xd[10 + i] = D.Iinv[i][0] * rhs[0] + D.Iinv[i][1] * rhs[1] + D.Iinv[i][2] * rhs[2];
}

template <class T> inline void normalise_q(T x[13]) {  // Quaternion.normalize, base.py:443-444, 473-474
    const T n = m_sqrt(x[6] * x[6] + x[7] * x[7] + x[8] * x[8] + x[9] * x[9]);
    for (int i = 6; i < 10; ++i) x[i] = x[i] / n;
}

// classic RK4 with the control held, dynamics/base.py:408-446
template <class T>
void state_step(const oracle_params& P, const Derived& D, const T x[13], const T u[7], const T& h, T xn[13]) {
    T k1[13], k2[13], k3[13], k4[13], xs[13];
    const T half = h / 2.0, sixth = h / 6.0;
    state_derivative(P, D, x, u, k1);
    for (int i = 0; i < 13; ++i) xs[i] = x[i] + half * k1[i];
    state_derivative(P, D, xs, u, k2);
    for (int i = 0; i < 13; ++i) xs[i] = x[i] + half * k2[i];
    state_derivative(P, D, xs, u, k3);
    for (int i = 0; i < 13; ++i) xs[i] = x[i] + h * k3[i];
    state_derivative(P, D, xs, u, k4);
    for (int i = 0; i < 13; ++i) xn[i] = x[i] + sixth * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

// sub-stepped update, quaternion normalised once after the last sub-step; base.py:450-480
template <class T>
void state_update(const oracle_params& P, const Derived& D, const T x[13], const T u[7], const T& dt, T xn[13]) {
    const int ns = P.substeps < 1 ? 1 : P.substeps;
    T cur[13], nxt[13];
    for (int i = 0; i < 13; ++i) cur[i] = x[i];
    const T h = (ns == 1) ? dt : dt / (double)ns;
    for (int s = 0; s < ns; ++s) {
        state_step(P, D, cur, u, h, nxt);
        for (int i = 0; i < 13; ++i) cur[i] = nxt[i];
    }
    if (P.normalise) normalise_q(cur);
    for (int i = 0; i < 13; ++i) xn[i] = cur[i];
}

inline void gather(const double* A, long n, long i, int rows, double* out) {
    for (int r = 0; r < rows; ++r) out[r] = A[(long)r * n + i];
}

}  // namespace

extern "C" {

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_state_derivative_f64(const oracle_params* p, const double* X, const double* U, long n, double* Xdot) {
    if (!p || !X || !U || !Xdot || n < 0) return -1;
    Derived D; make_derived(*p, D);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double x[13], u[7], xd[13];
        gather(X, n, i, 13, x); gather(U, n, i, 7, u);
        state_derivative<double>(*p, D, x, u, xd);
        for (int r = 0; r < 13; ++r) Xdot[(long)r * n + i] = xd[r];
    }
    return 0;
}

int oracle_step_f64(const oracle_params* p, const double* X, const double* U, const double* dt, int dt_is_scalar,
                    long n, double* Xn) {
    if (!p || !X || !U || !dt || !Xn || n < 0) return -1;
    Derived D; make_derived(*p, D);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double x[13], u[7], xn[13];
        gather(X, n, i, 13, x); gather(U, n, i, 7, u);
        const double h = dt_is_scalar ? dt[0] : dt[i];
        state_update<double>(*p, D, x, u, h, xn);
        for (int r = 0; r < 13; ++r) Xn[(long)r * n + i] = xn[r];
    }
    return 0;
}

int oracle_rollout_f64(const oracle_params* p, const double* X0, const double* U, double dt, long B, long H,
                       double* Xout) {
    if (!p || !X0 || !U || !Xout || B < 0 || H < 0) return -1;
    Derived D; make_derived(*p, D);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < B; ++i) {
        double x[13], u[7], xn[13];
        gather(X0, B, i, 13, x);
        for (int r = 0; r < 13; ++r) Xout[(long)r * B + i] = x[r];
        for (long k = 0; k < H; ++k) {
            gather(U + k * 7 * B, B, i, 7, u);
            state_update<double>(*p, D, x, u, dt, xn);
            for (int r = 0; r < 13; ++r) { x[r] = xn[r]; Xout[((k + 1) * 13 + r) * B + i] = xn[r]; }
        }
    }
    return 0;
}

int oracle_step_sens_f64(const oracle_params* p, const double* X, const double* U, const double* dt,
                         int dt_is_scalar, long n, double* Xn, double* A, double* Bm, double* c) {
    if (!p || !X || !U || !dt || !Xn || n < 0) return -1;
    Derived D; make_derived(*p, D);
    typedef Dual<21> T;  // 13 state + 7 control + dt directions
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double xv[13], uv[7];
        gather(X, n, i, 13, xv); gather(U, n, i, 7, uv);
        T x[13], u[7], xn[13];
        for (int r = 0; r < 13; ++r) { x[r] = T(xv[r]); x[r].d[r] = 1.0; }
        for (int r = 0; r < 7; ++r) { u[r] = T(uv[r]); u[r].d[13 + r] = 1.0; }
        T h(dt_is_scalar ? dt[0] : dt[i]); h.d[20] = 1.0;
        state_update<T>(*p, D, x, u, h, xn);
        for (int r = 0; r < 13; ++r) {
            Xn[(long)r * n + i] = xn[r].v;
            if (A) for (int j = 0; j < 13; ++j) A[((long)r * 13 + j) * n + i] = xn[r].d[j];
            if (Bm) for (int j = 0; j < 7; ++j) Bm[((long)r * 7 + j) * n + i] = xn[r].d[13 + j];
            if (c) c[(long)r * n + i] = xn[r].d[20];
        }
    }
    return 0;
}

// x_dot = f(x, u) with its Jacobians Fx = df/dx (13x13), Fu = df/du (13x7): what ca.jacobian(state_derivative, .) gives
// the reference — the implicit defect row x_k + dt f(x_{k+1}, u_k) (control/base.py:282-284), the Baumgarte row
// (:288-304) and the LQR wrapper (dynamics/base.py:51-52) differentiate f, not the step.
int oracle_state_derivative_sens_f64(const oracle_params* p, const double* X, const double* U, long n, double* Xdot,
                                     double* Fx, double* Fu) {
    if (!p || !X || !U || !Xdot || n < 0) return -1;
    Derived D; make_derived(*p, D);
    typedef Dual<20> T;  // 13 state + 7 control directions
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double xv[13], uv[7];
        gather(X, n, i, 13, xv); gather(U, n, i, 7, uv);
        T x[13], u[7], xd[13];
        for (int r = 0; r < 13; ++r) { x[r] = T(xv[r]); x[r].d[r] = 1.0; }
        for (int r = 0; r < 7; ++r) { u[r] = T(uv[r]); u[r].d[13 + r] = 1.0; }
        state_derivative<T>(*p, D, x, u, xd);
        for (int r = 0; r < 13; ++r) {
            Xdot[(long)r * n + i] = xd[r].v;
            if (Fx) for (int j = 0; j < 13; ++j) Fx[((long)r * 13 + j) * n + i] = xd[r].d[j];
            if (Fu) for (int j = 0; j < 7; ++j) Fu[((long)r * 7 + j) * n + i] = xd[r].d[13 + j];
        }
    }
    return 0;
}

// Envelope rows of AircraftControl.state_constraint (control/aircraft.py:44-59) and their state Jacobian:
//   rows[0] = v_rel . v_rel   (bounded 20^2 .. 100^2)      rows[1] = beta   (|.| <= 10 deg)
//   rows[2] = alpha           (|.| <= 20 deg)              rows[3] = z = x[2]  (< 0)
// rows [4][n], Jx [4][13][n] (may be NULL).  The rows do not depend on the control.
int oracle_envelope_f64(const oracle_params* p, const double* X, long n, double* rows, double* Jx) {
    if (!p || !X || !rows || n < 0) return -1;
    Derived D; make_derived(*p, D);
    typedef Dual<13> T;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double xv[13];
        gather(X, n, i, 13, xv);
        T x[13], u[7];
        for (int r = 0; r < 13; ++r) { x[r] = T(xv[r]); x[r].d[r] = 1.0; }
        for (int r = 0; r < 7; ++r) u[r] = T(0.0);
        Aero<T> a;
        aero<T>(*p, D, x, u, a);
        const T row[4] = {a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2], a.beta, a.alpha, x[2]};
        for (int r = 0; r < 4; ++r) {
            rows[(long)r * n + i] = row[r].v;
            if (Jx) for (int j = 0; j < 13; ++j) Jx[((long)r * 13 + j) * n + i] = row[r].d[j];
        }
    }
    return 0;
}

int oracle_aero_f64(const oracle_params* p, const double* X, const double* U, long n, double* out) {
    if (!p || !X || !U || !out || n < 0) return -1;
    Derived D; make_derived(*p, D);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double x[13], u[7];
        gather(X, n, i, 13, x); gather(U, n, i, 7, u);
        Aero<double> a;
        aero<double>(*p, D, x, u, a);
        const double qx = x[6], qy = x[7], qz = x[8], qw = x[9];  // Euler getters, base.py:179-195
        const double phi = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
        const double theta = std::asin(2 * (qw * qy - qz * qx));
        const double psi = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
        double o[22] = {a.vr[0], a.vr[1], a.vr[2], a.V, a.alpha, a.beta, a.qbar, a.C[0], a.C[1], a.C[2],
                        a.C[3], a.C[4], a.C[5], a.F[0], a.F[1], a.F[2], a.M[0], a.M[1], a.M[2], phi, theta, psi};
        for (int r = 0; r < 22; ++r) out[(long)r * n + i] = o[r];
    }
    return 0;
}

int oracle_mlp_f64(const oracle_params* p, const double* inputs, long n, double* outputs, double* jac) {
    if (!p || !inputs || !outputs || n < 0 || p->model_kind != ORACLE_MODEL_NN) return -1;
    Derived D; make_derived(*p, D);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) mlp_eval(*p, D, inputs + 5 * i, outputs + 6 * i, jac ? jac + 30 * i : nullptr);
    return 0;
}

}  // extern "C"

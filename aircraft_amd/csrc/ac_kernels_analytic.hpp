// ac_kernels_analytic.hpp — kernels for the analytic coefficient models (default / linear / poly).
//
// Layouts (all fp32, component-major): X [13][n], U [7][n], Xn [13][n], A [13][13][n], Bm [13][7][n],
// c [13][n]; rollout U [H][7][B], Xout [H+1][13][B].  Consecutive lanes own consecutive units, so
// every global access is a contiguous 256-B (forward) or 64-B (4-lanes-per-unit) segment per row.
#pragma once
#include "ac_dynamics.hpp"

namespace ac {

constexpr int kBlock = 256;

// Blocked component-major addressing.  Unit u = q*blk + r lives at [q][row][r] of an array
// [n/blk][ROWS][blk].  blk == n is the flat [ROWS][n] case; blk == B addresses rollout-shaped buffers
// ([H][ROWS][B]: q = node, r = instance) in place, so multiple-shooting callers need no transpose.
struct UnitAddr {
    long q, r, blk;
    AC_DI UnitAddr(long u, long blk_) : q(u / blk_), r(u % blk_), blk(blk_) {}
    AC_DI long off(int rows) const { return q * rows * blk + r; }
    // A copy the optimiser cannot see through.  Used right before the output stores of the long kernels: hipcc
    // otherwise hoists all ~80 output addresses (64-bit each) to kernel entry, spills them to scratch, and reloads them
    // one at a time behind a full s_waitcnt vmcnt(0) in front of every store.
    AC_DI UnitAddr late() const {
        UnitAddr u = *this;
        asm volatile("" : "+v"(u.q), "+v"(u.r));
        asm volatile("" : "+s"(u.blk));  // (wave-uniform: a kernel argument) — its row multiples r * blk are not hoisted either
        return u;
    }
};

template <int ROWS> AC_DI void load_rows(const float* __restrict__ src, const UnitAddr& ua, float out[ROWS]) {
    const float* p = src + ua.off(ROWS);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) out[r] = p[(long)r * ua.blk];
}
template <int ROWS> AC_DI void store_rows(float* __restrict__ dst, const UnitAddr& ua, const float v[ROWS]) {
    float* p = dst + ua.off(ROWS);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) p[(long)r * ua.blk] = v[r];
}
// What a forward (x+ = F) or derivative (x_dot = f) kernel stores for unit `unit` of the node-major buffer X: its result
// v, or the defect row built from it (control/base.py:275-286) — the neighbouring node of the same instance is one block
// of X away, so the rows need no pointer of their own:
//   AC_ROWS_DEFECT    x_{k+1} - F(x_k, u_k, dt_k)                      X = nodes 0 .. H, unit (k, b) reads node k + 1
//   AC_ROWS_IMPLICIT  x_{k+1} - x_k - dt_k f(x_{k+1}, u_k)            X = nodes 1 .. H (the caller's X + 13 B), reads node k
AC_DI void store_state_rows(const DevParams& P, const float* __restrict__ X, float* __restrict__ out, const UnitAddr& ua,
                            long unit, const float v[13]) {
    if (P.rows == AC_ROWS_PLAIN) { store_rows<13>(out, ua, v); return; }
    float r[13];
    if (P.rows == AC_ROWS_DEFECT) {
        UnitAddr un = ua;
        un.q += 1;
        float xn[13];
        load_rows<13>(X, un, xn);
#pragma unroll
        for (int i = 0; i < 13; ++i) r[i] = xn[i] - v[i];
    } else {
        UnitAddr up = ua;
        up.q -= 1;
        float x1[13], x0[13];
        load_rows<13>(X, ua, x1);
        load_rows<13>(X, up, x0);
        const float dtk = P.rows_dt_per_unit ? P.rows_dt_per_unit[unit] : P.rows_dt;
#pragma unroll
        for (int i = 0; i < 13; ++i) r[i] = fmaf(-dtk, v[i], x1[i] - x0[i]);
    }
    store_rows<13>(out, ua, r);
}
// flat [ROWS][n] helper used by the rollout kernels
template <int ROWS> AC_DI void load_rows(const float* __restrict__ src, long n, long i, float out[ROWS]) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) out[r] = src[(long)r * n + i];
}

// ---- forward kernels: one lane per unit ------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_state_derivative(const DevParams P, const float* __restrict__ X,
                                                             const float* __restrict__ U, long n, long blk,
                                                             float* __restrict__ Xdot) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const UnitAddr ua(i, blk);
    float x[13], u[7], xd[13];
    load_rows<13>(X, ua, x);
    load_rows<7>(U, ua, u);
    AnalyticCoeffs<MODEL> coeffs;
    coeffs.prefetch(P, x, u);
    state_derivative<float>(P, coeffs, x, u, xd);
    store_state_rows(P, X, Xdot, ua, i, xd);
}

template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_step(const DevParams P, const float* __restrict__ X,
                                                 const float* __restrict__ U, float dt,
                                                 const float* __restrict__ dt_per_unit, long n, long blk,
                                                 float* __restrict__ Xn) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const UnitAddr ua(i, blk);
    float x[13], u[7];
    load_rows<13>(X, ua, x);
    load_rows<7>(U, ua, u);
    const float h = dt_per_unit ? dt_per_unit[i] : dt;
    AnalyticCoeffs<MODEL> coeffs;
    state_update(P, coeffs, x, u, h);
    store_state_rows(P, X, Xn, ua, i, x);
}

// Sequential-in-k rollout: the state stays in registers, u_k streams in, x_{k+1} streams out
// (28 B read + 52 B written per horizon step).
template <int MODEL>
__global__ __launch_bounds__(64) void k_rollout(const DevParams P, const float* __restrict__ X0,
                                                const float* __restrict__ U, float dt, long B, long H,
                                                float* __restrict__ Xout) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= B) return;
    float x[13], u[7], un[7];
    load_rows<13>(X0, B, i, x);
    double xa[13];  // float64 carry of the state across the horizon (see state_update_carry)
#pragma unroll
    for (int r = 0; r < 13; ++r) { Xout[(long)r * B + i] = x[r]; xa[r] = (double)x[r]; }
    if (H > 0) load_rows<7>(U, B, i, u);
    AnalyticCoeffs<MODEL> coeffs;
    for (long k = 0; k < H; ++k) {
        if (k + 1 < H) load_rows<7>(U + (k + 1) * 7 * B, B, i, un);  // prefetch next control
        state_update_carry(P, coeffs, xa, u, dt);
        float* o = Xout + (k + 1) * 13 * B;
#pragma unroll
        for (int r = 0; r < 13; ++r) o[(long)r * B + i] = (float)xa[r];
#pragma unroll
        for (int r = 0; r < 7; ++r) u[r] = un[r];
    }
}

// Aerodynamic getters (v_frd_rel, airspeed, alpha, beta, qbar, coefficients, forces_frd, moments_frd)
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_aero(const DevParams P, const float* __restrict__ X,
                                                 const float* __restrict__ U, long n, long blk,
                                                 float* __restrict__ out) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const UnitAddr ua(i, blk);
    float x[13], u[7];
    load_rows<13>(X, ua, x);
    load_rows<7>(U, ua, u);
    AeroPre<float> a;
    aero_pre(P, x, a);
    AeroPost<float> o;
    if constexpr (MODEL == AC_MODEL_QUAD) {
        quad_forces(P, u, o);
    } else {
        float C[6];
        AnalyticCoeffs<MODEL> coeffs;
        coeffs.prefetch(P, x, u);
        coeffs(P, a, x, u, C);
        aero_post(P, a, u, C, o);
    }
    float eu[3];
        euler_angles(x, eu[0], eu[1], eu[2]);
        const float v[22] = {a.vr[0], a.vr[1], a.vr[2], a.V, a.alpha, a.beta, a.qbar, o.C[0], o.C[1], o.C[2],
                         o.C[3], o.C[4], o.C[5], o.F[0], o.F[1], o.F[2], o.M[0], o.M[1], o.M[2], eu[0], eu[1], eu[2]};
    store_rows<22>(out, ua, v);
}

// ---- sensitivities (direction layout: ac_dynamics.hpp, struct SeedsT) -----------------------------
// QUAD: the quadrotor's control directions 10..13 are B columns 0..3 and columns 4..6 are the constant zeros
// (aircraft: columns 0, 1, 2, 6 and zeros in the thrust columns 3..5).
template <int N, bool QUAD = false> struct SensIOT {
    // column of direction d in the output arrays: base pointer + row stride (null = no column)
    static AC_DI void column(int d, long n, float* Au, float* Bu, float* cu, float*& base, long& stride) {
        if (d < 10) { base = Au + (long)(3 + d) * n; stride = 13 * n; }
        else if (d < 13) { base = Bu + (long)(d - 10) * n; stride = 7 * n; }
        else if (d == 13) { base = Bu + (QUAD ? 3L : 6L) * n; stride = 7 * n; }
        else if (d == 14) { base = cu; stride = n; }
        else { base = nullptr; stride = 0; }
    }

    // Store this lane's N tangent columns of the 13 outputs (+ its share of the constant columns).
    // p_diag: the diagonal of the constant position block — 1 for the step (dF/dp = [I; 0]), 0 for f itself (df/dp = 0)
    // scale, diag: the stored Jacobian is  scale * J + diag * I  (the implicit defect rows store I - dt Fx and -dt Fu)
    static AC_DI void store(int g, const UnitAddr& ua, const Dual<N> x[13], float* __restrict__ A,
                            float* __restrict__ Bm, float* __restrict__ c, bool constants, float p_diag = 1.f,
                            float scale = 1.f, float diag = 0.f) {
        const long n = ua.blk;  // row stride
        float* Au = A + ua.off(169);
        float* Bu = Bm + ua.off(91);
        float* cu = c ? c + ua.off(13) : nullptr;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float* base; long stride;
            column(N * g + j, n, Au, Bu, cu, base, stride);
            if (base) {
                const int d = N * g + j;
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    float v = x[i].d[j];
                    if (scale != 1.f || diag != 0.f) v = fmaf(scale, v, (d < 10 && i == 3 + d) ? diag : 0.f);
                    base[(long)i * stride] = v;
                }
            }
        }
        if (constants && g < 3) {
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                Au[((long)i * 13 + g) * n] = (i == g) ? p_diag : 0.f;  // dF/dp_g
                Bu[((long)i * 7 + (QUAD ? 4 : 3) + g) * n] = 0.f;  // dF/d(control without effect)
            }
        }
    }

    // Read back the columns this lane stored earlier (sub-step composition only).
    static AC_DI void load(int g, const UnitAddr& ua, Dual<N> x[13], const float* A, const float* Bm,
                           const float* c) {
        const long n = ua.blk;
        float* Au = const_cast<float*>(A) + ua.off(169);
        float* Bu = const_cast<float*>(Bm) + ua.off(91);
        float* cu = c ? const_cast<float*>(c) + ua.off(13) : nullptr;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float* base; long stride;
            column(N * g + j, n, Au, Bu, cu, base, stride);
#pragma unroll
            for (int i = 0; i < 13; ++i) x[i].d[j] = base ? base[(long)i * stride] : 0.f;
        }
    }
};
typedef SensIOT<4> SensIO;

// The whole sensitivity step for one unit spread over 16/N lanes (g = 0 .. 16/N - 1; `col` = unit within the wave,
// units per wave UPW = 4 N).  Wave-collective when the coefficient provider is (MLP engine) and always for
// substeps > 1 (cross-lane composition).
//
// substeps == 1 (every MPC driver of the reference): one seeded RK4 step, tangents stay in registers.
// substeps  > 1: each sub-step is seeded on its own (local Jacobian T_s w.r.t. its inputs) and composed with
// the running total  T <- dF_s/dx . T + dF_s/d(u, dt)  kept in the OUTPUT arrays between sub-steps, so the
// register footprint of the hot path is not paid for by the rare one.
// SUB: 0 = the kernel handles any sub-step count (MLP kernels), 1 = one sub-step only (no composition code in the
// kernel: its ~130 registers of told / tnew would otherwise set the allocation of the whole kernel), 2 = sub-stepped only.
template <int N, int SUB = 0, class Coeffs, class Acc>
AC_DI void sens_update(const DevParams& P, Coeffs& coeffs, int g, int col, const UnitAddr& ua, float xv[13],
                       const float uv[7], float dt, Dual<N> x[13], float* __restrict__ A, float* __restrict__ Bm,
                       float* __restrict__ c, bool live, Acc& acc) {
    constexpr int UPW = 4 * N;  // units per wave
    typedef SensIOT<N, Coeffs::kModel == AC_MODEL_QUAD> IO;
    if constexpr (SUB == 1) {
        rk4_step_seeded<N>(P, coeffs, g, xv, uv, dt, 1.0f, x, acc);
        if (P.p.normalise) normalise_q(x);
        return;
    }
    const int ns = P.p.substeps < 1 ? 1 : P.p.substeps;
    const float hv = (ns == 1) ? dt : dt / (float)ns;
    const float dh = 1.0f / (float)ns;
#pragma nounroll
    for (int s = 0; s < ns; ++s) {
        rk4_step_seeded<N>(P, coeffs, g, xv, uv, hv, dh, x, acc);
        if (ns > 1) {
            const UnitAddr ul = ua.late();  // addresses computed here, not hoisted to kernel entry
            if (s > 0) {
                Dual<N> told[13], tnew[13];
                if (live) IO::load(g, ul, told, A, Bm, c);
                else {
#pragma unroll
                    for (int i = 0; i < 13; ++i) told[i] = Dual<N>(0.f);
                }
#pragma unroll
                for (int i = 0; i < 13; ++i) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        // direct dependence of this sub-step on (u, dt); state directions chain only
                        float t = (N * g + j >= 10) ? x[i].d[j] : 0.f;
                        if (i < 3) t += told[i].d[j];  // dF_s/dp = [I; 0]
                        tnew[i].d[j] = t;
                    }
                }
#pragma unroll
                for (int k = 0; k < 10; ++k) {
#pragma unroll
                    for (int i = 0; i < 13; ++i) {
                        const float a_ik = __shfl(x[i].d[k % N], col + UPW * (k / N), 64);  // dF_s[i]/dx[3+k]
#pragma unroll
                        for (int j = 0; j < N; ++j) tnew[i].d[j] = fmaf(a_ik, told[3 + k].d[j], tnew[i].d[j]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 13; ++i) {
#pragma unroll
                    for (int j = 0; j < N; ++j) x[i].d[j] = tnew[i].d[j];
                }
            }
            if (live) IO::store(g, ul, x, A, Bm, c, false);
            __builtin_amdgcn_s_waitcnt(0);  // own stores retired before the next sub-step reads them back
#pragma unroll
            for (int i = 0; i < 13; ++i) xv[i] = x[i].v;
        }
    }
    if (P.p.normalise) normalise_q(x);
}

template <int N, class Coeffs>
AC_DI void sens_update(const DevParams& P, Coeffs& coeffs, int g, int col, const UnitAddr& ua, float xv[13],
                       const float uv[7], float dt, Dual<N> x[13], float* __restrict__ A, float* __restrict__ Bm,
                       float* __restrict__ c, bool live) {
    RegAcc<N> acc;
    sens_update<N>(P, coeffs, g, col, ua, xv, uv, dt, x, A, Bm, c, live, acc);
}

// Directions per lane for the analytic models: four (four lanes per unit, 16 units per wave).  The kernels are bound by
// vector-ALU issue (profiles/r04_analytic_pmc_before.json: 79 % of wave cycles in VALU at one wave per SIMD, HBM writes
// 1.00x the algorithmic bytes), the primal is recomputed by every lane of a unit, so fewer lanes per unit is less work;
// the structured tangents (ac_dynamics.hpp) keep four directions within 256 registers = two waves per SIMD, which is
// what lets the SIMD issue a vector instruction every ~2 cycles instead of every 4.
template <int MODEL> struct AnalyticSensN { static constexpr int value = 4; };
#ifndef AC_SENS_WPS
#define AC_SENS_WPS 2
#endif
constexpr int kSensWavesPerSimd = AC_SENS_WPS;
// ---- x_dot = f(x, u) with its Jacobians (the implicit defect row and the Baumgarte row differentiate f, not the step:
// control/base.py:282-304; the LQR wrapper: dynamics/base.py:51-52) -------------------------------------------------
// One evaluation of f seeded like the first RK4 stage: k[i].d[j] = df_i / d(direction N g + j).
template <int N, class Coeffs>
AC_DI void deriv_seeded(const DevParams& P, Coeffs& coeffs, int g, const float xv[13], const float uv[7], Dual<N> k[13]) {
    typedef SeedsT<N> Seeds;
    Dual<N> xs[13], u[7];
#pragma unroll
    for (int i = 0; i < 13; ++i) xs[i] = Seeds::state(g, i, xv[i]);
    coeffs.prefetch(P, xs, uv);
    Seeds::template controls<Coeffs::kModel == AC_MODEL_QUAD>(g, uv, u);
    state_derivative(P, coeffs, xs, u, k);
}

// X: the kernel's node-major input (for the implicit rows: nodes 1 .. H, see store_state_rows)
template <int N, bool QUAD>
AC_DI void deriv_store(const DevParams& P, const float* __restrict__ X, long unit, int g, const UnitAddr& ua,
                       const Dual<N> k[13], float* __restrict__ Xdot, float* __restrict__ Fx, float* __restrict__ Fu) {
    const UnitAddr uo = ua.late();
    if (P.rows == AC_ROWS_IMPLICIT) {
        // r = x_{k+1} - x_k - dt f,  d r / d x_{k+1} = I - dt Fx,  d r / d u = -dt Fu,  d r / d dt = -f   (control/base.py:282-284)
        const float dtk = P.rows_dt_per_unit ? P.rows_dt_per_unit[unit] : P.rows_dt;
        if (g == 0) {
            float f[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) f[i] = k[i].v;
            store_state_rows(P, X, Xdot, uo, unit, f);
            float* p = P.rows_aux + uo.off(13);
#pragma unroll
            for (int i = 0; i < 13; ++i) p[(long)i * uo.blk] = -f[i];
        }
        SensIOT<N, QUAD>::store(g, uo, k, Fx, Fu, nullptr, true, 1.f, -dtk, 1.f);  // (d/dp: I - dt 0)
        return;
    }
    if (g == 0) {
        float* p = Xdot + uo.off(13);
#pragma unroll
        for (int i = 0; i < 13; ++i) p[(long)i * uo.blk] = k[i].v;
    }
    SensIOT<N, QUAD>::store(g, uo, k, Fx, Fu, nullptr, true, 0.f);  // df/dp = 0, df/d(dead controls) = 0; no dt column
}

template <int MODEL>
__global__ __launch_bounds__(kBlock, kSensWavesPerSimd) void k_deriv_sens(const DevParams P, const float* __restrict__ X,
                                                       const float* __restrict__ U, long n, long blk,
                                                       float* __restrict__ Xdot, float* __restrict__ Fx,
                                                       float* __restrict__ Fu) {
    constexpr int kAnN = AnalyticSensN<MODEL>::value;
    static_assert(kBlock / 64 == 16 / kAnN, "one wave per direction group");
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave = direction group, lane = unit (see k_step_sens)
    const long unit_raw = (long)blockIdx.x * 64 + lane;
    const bool live = unit_raw < n;
    const long unit = live ? unit_raw : n - 1;
    const UnitAddr ua(unit, blk);
    float xv[13], uv[7];
    load_rows<13>(X, ua, xv);
    load_rows<7>(U, ua, uv);
    Dual<kAnN> k[13];
    constexpr bool kShared = MODEL == AC_MODEL_POLY;  // the four waves split the primal part of the cubic fits (AnalyticCoeffs)
    AnalyticCoeffs<MODEL, kShared> coeffs;
    if constexpr (kShared) {
        __shared__ float poly_xch[kPolyXchFloats];
        coeffs.xch = poly_xch;
        coeffs.xg = g;
    }
    deriv_seeded<kAnN>(P, coeffs, g, xv, uv, k);
    if (live) deriv_store<kAnN, MODEL == AC_MODEL_QUAD>(P, X, unit, g, ua, k, Xdot, Fx, Fu);
}

// ---- envelope rows of AircraftControl.state_constraint (control/aircraft.py:44-59) and their state Jacobian ---------
//   rows[0] = v_rel . v_rel (20^2 .. 100^2)   rows[1] = beta (+-10 deg)   rows[2] = alpha (+-20 deg)   rows[3] = z (< 0)
// One lane per unit; the rows depend on (v, q) only: Dual<7>.  rows [4][n]; Jx [4][13][n] (may be NULL).
template <int INST = 0>  // a template only so that the header may be included by several translation units
__global__ __launch_bounds__(kBlock) void k_envelope(const DevParams P, const float* __restrict__ X, long n, long blk,
                                                     float* __restrict__ rows, float* __restrict__ Jx) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const UnitAddr ua(i, blk);
    float xv[13];
    load_rows<13>(X, ua, xv);
    typedef Dual<7> T;
    T x[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) {
        x[r] = T(xv[r]);
        if (r >= 3 && r < 10) x[r].d[r - 3] = 1.f;
    }
    AeroPre<T> a;
    aero_pre(P, x, a);
    const T row[3] = {a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2], a.beta, a.alpha};
    float* ro = rows + ua.off(4);
#pragma unroll
    for (int r = 0; r < 3; ++r) ro[(long)r * blk] = row[r].v;
    ro[3L * blk] = xv[2];
    if (Jx) {
        float* jo = Jx + ua.off(52);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                float v = 0.f;
                if (r < 3 && j >= 3 && j < 10) v = row[r].d[j - 3];
                if (r == 3 && j == 2) v = 1.f;
                jo[(long)(r * 13 + j) * blk] = v;
            }
    }
}

// ---- the envelope as a soft constraint of the batched solver (the state_constraint role,
// control/aircraft.py:44-59; IPOPT enforces the rows, the iLQR sweep penalises their violation) -----------------------
//   penalty_k(x) = w sum_r max(0, g_r(x) - hi_r)^2 + max(0, lo_r - g_r(x))^2
struct EnvelopePenalty {
    float lo[4], hi[4];  // bounds of (|v_rel|^2, beta, alpha, z): Aircraft.ENVELOPE_BOUNDS
    float weight;
};

// signed violation of row r (0 inside the bounds)
AC_DI float envelope_violation(const EnvelopePenalty& E, int r, float g) {
    return g > E.hi[r] ? g - E.hi[r] : (g < E.lo[r] ? g - E.lo[r] : 0.f);
}

// Augmented-Lagrangian form (the HARD treatment of the rows IPOPT enforces, control/aircraft.py:44-59): with multipliers
// lam_hi, lam_lo >= 0 per node, row and instance,
//   L_A = w [ max(0, g - hi + lam_hi / 2w)^2 - (lam_hi / 2w)^2 + max(0, lo - g + lam_lo / 2w)^2 - (lam_lo / 2w)^2 ]
// — the same quadratic penalty on bounds SHIFTED inwards by lam / 2w, so cost, gradient and Gauss-Newton curvature are the
// penalty kernels' with a shifted violation; the first-order multiplier update lam <- max(0, lam + 2w (g - hi)) is
// lam <- 2w x (that shifted violation).  lam [H+1][8][B]: rows 0-3 upper, 4-7 lower bounds; NULL = plain penalty.
// Both sides are evaluated on their own (with large multipliers on a narrow row both shifted bounds can be violated at once):
//   v = max(0, up) - max(0, dn)  (gradient 2 w v grad g),  sq = max(0, up)^2 + max(0, dn)^2  (cost),  active = sides violated
//   (Gauss-Newton curvature 2 w active grad g grad g').
struct ShiftedViolation { float v, sq, active, const_term; };
AC_DI ShiftedViolation envelope_violation_al(const EnvelopePenalty& E, int r, float g, float lam_hi, float lam_lo) {
    const float i2w = E.weight > 0.f ? 0.5f / E.weight : 0.f;
    const float sh = lam_hi * i2w, sl = lam_lo * i2w;
    const float up = fmaxf(0.f, g - E.hi[r] + sh), dn = fmaxf(0.f, E.lo[r] - g + sl);
    ShiftedViolation o;
    o.v = up - dn;
    o.sq = fmaf(up, up, dn * dn);
    o.active = (up > 0.f ? 1.f : 0.f) + (dn > 0.f ? 1.f : 0.f);
    o.const_term = sh * sh + sl * sl;
    return o;
}
AC_DI void load_multipliers(const float* __restrict__ lam, long k, long b, long B, float lh[4], float ll[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lh[r] = lam ? lam[(k * 8 + r) * B + b] : 0.f;
        ll[r] = lam ? lam[(k * 8 + 4 + r) * B + b] : 0.f;
    }
}

// cost[b] += sum_k penalty_k(x_k), k = 0..H: one lane per instance (X [H+1][13][B])
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_envelope_cost(const DevParams P, const EnvelopePenalty E,
                                                          const float* __restrict__ lam, long Bl,
                                                          const float* __restrict__ X, long B, long H,
                                                          float* __restrict__ cost) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    float acc = 0.f;
    for (long k = 0; k <= H; ++k) {
        float x[13];
        load_rows<13>(X + k * 13 * B, B, i, x);
        AeroPre<float> a;
        aero_pre(P, x, a);
        const float g[4] = {a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2], a.beta, a.alpha, x[2]};
        float lh[4], ll[4];
        load_multipliers(lam, k, i % Bl, Bl, lh, ll);  // (a batch of line-search candidates reads instance b % Bl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const ShiftedViolation sv = envelope_violation_al(E, r, g[r], lh[r], ll[r]);
            acc = (acc + sv.sq) - sv.const_term;
        }
    }
    cost[i] += E.weight * acc;
}

// First-order multiplier update at the iterate X [H+1][13][B]: lam <- 2 w x (shifted violation), one lane per (node,
// instance); viol_max [B] (may be NULL; zero it first): the largest |g - bound| excess of the instance, unshifted, each row
// scaled by `scale[r]` so that the four rows compare (atomic max on the bits of a non-negative float).
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_envelope_multipliers(const DevParams P, const EnvelopePenalty E,
                                                                 const float* __restrict__ X, long B, long H,
                                                                 float* __restrict__ lam, float* __restrict__ viol_max) {
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (H + 1) * B) return;
    const long k = t / B, b = t % B;
    float x[13];
    load_rows<13>(X + k * 13 * B, B, b, x);
    AeroPre<float> a;
    aero_pre(P, x, a);
    const float g[4] = {a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2], a.beta, a.alpha, x[2]};
    float lh[4], ll[4];
    load_multipliers(lam, k, b, B, lh, ll);
    const float w2 = 2.0f * E.weight;
    float worst = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float up = g[r] - E.hi[r], dn = E.lo[r] - g[r];
        lam[(k * 8 + r) * B + b] = fmaxf(0.f, fmaf(w2, up, lh[r]));
        lam[(k * 8 + 4 + r) * B + b] = fmaxf(0.f, fmaf(w2, dn, ll[r]));
        const float span = E.hi[r] - E.lo[r];
        const float sc = (span > 0.f && span < 1e30f) ? 1.0f / span : 1.0f;  // relative to the row's range where it has one
        worst = fmaxf(worst, fmaxf(up, dn) * sc);
    }
    if (viol_max && worst > 0.f && worst == worst) atomicMax(reinterpret_cast<int*>(viol_max) + b, __float_as_int(worst));
}

// Quadratic model of the penalty around the iterate, in the form the backward pass consumes: the gradient
// 2 w sum_r viol_r grad g_r is ADDED to node_glin [H+1][13][B]; the Gauss-Newton curvature 2 w sum_{r violated} grad g_r grad g_r'
// is ADDED to the (x, x) block of Hz [H][21][21][B] (nodes k < H; the terminal node keeps the gradient only).
// One lane per (node, instance).
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_envelope_model(const DevParams P, const EnvelopePenalty E,
                                                           const float* __restrict__ lam,
                                                           const float* __restrict__ X, long B, long H,
                                                           float* __restrict__ node_glin, float* __restrict__ Hz) {
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (H + 1) * B) return;
    const long k = t / B, b = t % B;
    float xv[13];
    load_rows<13>(X + k * 13 * B, B, b, xv);
    typedef Dual<7> T;
    T x[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) { x[r] = T(xv[r]); if (r >= 3 && r < 10) x[r].d[r - 3] = 1.f; }
    AeroPre<T> a;
    aero_pre(P, x, a);
    const T row[3] = {a.vr[0] * a.vr[0] + a.vr[1] * a.vr[1] + a.vr[2] * a.vr[2], a.beta, a.alpha};
    float viol[4], active[4], grad[4][13], lh[4], ll[4];
    load_multipliers(lam, k, b, B, lh, ll);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const ShiftedViolation sv = envelope_violation_al(E, r, r < 3 ? row[r].v : xv[2], lh[r], ll[r]);
        viol[r] = sv.v; active[r] = sv.active;
#pragma unroll
        for (int j = 0; j < 13; ++j) grad[r][j] = r < 3 ? ((j >= 3 && j < 10) ? row[r].d[j - 3] : 0.f) : (j == 2 ? 1.f : 0.f);
    }
    const float w2 = 2.0f * E.weight;
    if (node_glin) {
        float* gl = node_glin + k * 13 * B + b;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(viol[r], grad[r][j], s);
            if (s != 0.f) gl[(long)j * B] += w2 * s;
        }
    }
    if (k < H && Hz) {
        float* hz = Hz + k * 441 * B + b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (active[r] == 0.f) continue;
#pragma unroll
            for (int i = 0; i < 13; ++i)
#pragma unroll
                for (int j = 0; j < 13; ++j) {
                    const float v = grad[r][i] * grad[r][j];
                    if (v != 0.f) hz[(long)(i * 21 + j) * B] += (w2 * active[r]) * v;
                }
        }
    }
}

// ---- quaternion rows of ControlProblem.state_constraint (control/base.py:285-304) on the NEXT node ------------------
//   mode 0 ('constraint'):  row = q . q - 1
//   mode 1 ('baumgarte'):   row = 2 a phi_dot + b^2 phi,  phi = q . q - 1,  phi_dot = 2 q . q_dot,  a = b = 2   (:296-301)
// with Jx = d row / dx [13][n] and Ju = d row / du [7][n] from x_dot, Fx = df/dx, Fu = df/du of the same (x, u).
template <int INST = 0>
__global__ __launch_bounds__(kBlock) void k_quat_rows(const float* __restrict__ X, const float* __restrict__ Xdot,
                                                      const float* __restrict__ Fx, const float* __restrict__ Fu, long n,
                                                      long blk, int mode, float* __restrict__ row,
                                                      float* __restrict__ Jx, float* __restrict__ Ju) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const UnitAddr ua(i, blk);
    float q[4], qd[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xp = X + ua.off(13);
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = xp[(long)(6 + c) * blk];
    const float phi = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) - 1.0f;
    float* jx = Jx + ua.off(13);
    float* ju = Ju + ua.off(7);
    if (mode == 0) {
        row[ua.off(1)] = phi;
#pragma unroll
        for (int j = 0; j < 13; ++j) jx[(long)j * blk] = (j >= 6 && j < 10) ? 2.0f * q[j - 6] : 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) ju[(long)j * blk] = 0.f;
        return;
    }
    const float* dp = Xdot + ua.off(13);
#pragma unroll
    for (int c = 0; c < 4; ++c) qd[c] = dp[(long)(6 + c) * blk];
    const float alpha = 2.0f, beta = 2.0f;
    const float phid = 2.0f * (q[0] * qd[0] + q[1] * qd[1] + q[2] * qd[2] + q[3] * qd[3]);
    row[ua.off(1)] = 2.0f * alpha * phid + beta * beta * phi;
    const float* fx = Fx + ua.off(169);
    const float* fu = Fu + ua.off(91);
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        float s = 0.f;  // q . d(q_dot)/dx_j
#pragma unroll
        for (int c = 0; c < 4; ++c) s = fmaf(q[c], fx[(long)((6 + c) * 13 + j) * blk], s);
        float dphid = 2.0f * s, dphi = 0.f;
        if (j >= 6 && j < 10) { dphid += 2.0f * qd[j - 6]; dphi = 2.0f * q[j - 6]; }
        jx[(long)j * blk] = 2.0f * alpha * dphid + beta * beta * dphi;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) s = fmaf(q[c], fu[(long)((6 + c) * 7 + j) * blk], s);
        ju[(long)j * blk] = 2.0f * alpha * 2.0f * s;
    }
}

template <int MODEL, bool SUBSTEPPED>
__global__ __launch_bounds__(kBlock, kSensWavesPerSimd) void k_step_sens(const DevParams P, const float* __restrict__ X,
                                                      const float* __restrict__ U, float dt,
                                                      const float* __restrict__ dt_per_unit, long n, long blk,
                                                      float* __restrict__ Xn, float* __restrict__ A,
                                                      float* __restrict__ Bm, float* __restrict__ c) {
    constexpr int kAnN = AnalyticSensN<MODEL>::value;
    constexpr int UPW = 4 * kAnN;  // units per wave (sub-stepped form)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One sub-step (every MPC driver): a WAVE is one direction group of 64 consecutive units (lane = unit), so every load and
    // store instruction of a wave covers 256 contiguous bytes and the 0/1 seed pattern is wave-uniform.  Sub-stepped: the
    // lanes of a unit share a wave (16 units x 4 groups), because the composition exchanges tangents across them.
    static_assert(kBlock / 64 == 16 / kAnN, "one wave per direction group");
    const int col = SUBSTEPPED ? lane % UPW : lane;
    const int g = SUBSTEPPED ? lane / UPW : __builtin_amdgcn_readfirstlane(wave);
    const long unit_raw = SUBSTEPPED ? ((long)blockIdx.x * (kBlock / 64) + wave) * UPW + col : (long)blockIdx.x * 64 + lane;
    const bool live = unit_raw < n;
    const long unit = live ? unit_raw : n - 1;  // clamp: dead lanes recompute the last unit, store nothing
    const UnitAddr ua(unit, blk);
    float xv[13], uv[7];
    load_rows<13>(X, ua, xv);
    load_rows<7>(U, ua, uv);
    const float hv = dt_per_unit ? dt_per_unit[unit] : dt;
    Dual<kAnN> x[13];
    constexpr bool kShared = MODEL == AC_MODEL_POLY && !SUBSTEPPED;  // (see k_deriv_sens)
    AnalyticCoeffs<MODEL, kShared> coeffs;
    if constexpr (kShared) {
        __shared__ float poly_xch[kPolyXchFloats];
        coeffs.xch = poly_xch;
        coeffs.xg = g;
    }
    static_assert(kAnN == 4, "LdsAcc4");
    __shared__ float4 acc_words[13 * kBlock];  // 53 KB: two workgroups per CU
    LdsAcc4 acc(&acc_words[threadIdx.x], kBlock);
    sens_update<kAnN, SUBSTEPPED ? 2 : 1>(P, coeffs, g, col, ua, xv, uv, hv, x, A, Bm, c, live, acc);
    if (!live) return;
    const UnitAddr uo = ua.late();
    if (g == 0) {
        float* p = Xn + uo.off(13);
#pragma unroll
        for (int i = 0; i < 13; ++i) p[(long)i * blk] = x[i].v;
    }
    SensIOT<kAnN, MODEL == AC_MODEL_QUAD>::store(g, uo, x, A, Bm, c, true);
}

// The step + sensitivity and derivative + sensitivity kernels of the analytic models are compiled in translation units of
// their own; every other unit only refers to them.  an_inst_sens_poly.hip is built with -fno-slp-vectorize: packing pairs
// of tangent chains into v_pk_fma_f32 costs the cubic-fit kernel more registers than two waves per SIMD leave it (300 B/lane
// of scratch); an_inst_sens.hip (default / linear / quadrotor) keeps the packing — those kernels fit either way, and at two
// waves per SIMD a packed instruction takes one issue slot like any other (profiles/r04_analytic_pmc_*.json).
#define AC_AN_SENS_ARGS const DevParams, const float*, const float*, float, const float*, long, long, float*, float*, float*, float*
#define AC_AN_DERIV_ARGS const DevParams, const float*, const float*, long, long, float*, float*, float*
#define AC_AN_MODEL(EXT, M)                                              \
    EXT template __global__ void k_step_sens<M, false>(AC_AN_SENS_ARGS); \
    EXT template __global__ void k_step_sens<M, true>(AC_AN_SENS_ARGS);  \
    EXT template __global__ void k_deriv_sens<M>(AC_AN_DERIV_ARGS);
#if defined(AC_AN_SENS_INSTANTIATE) && AC_AN_SENS_INSTANTIATE == 1
AC_AN_MODEL(, AC_MODEL_DEFAULT) AC_AN_MODEL(, AC_MODEL_LINEAR) AC_AN_MODEL(, AC_MODEL_QUAD)
#else
AC_AN_MODEL(extern, AC_MODEL_DEFAULT) AC_AN_MODEL(extern, AC_MODEL_LINEAR) AC_AN_MODEL(extern, AC_MODEL_QUAD)
#endif
#if defined(AC_AN_SENS_INSTANTIATE) && AC_AN_SENS_INSTANTIATE == 2
AC_AN_MODEL(, AC_MODEL_POLY)
#else
AC_AN_MODEL(extern, AC_MODEL_POLY)
#endif
#undef AC_AN_MODEL

}  // namespace ac

// ac_ilqr.hpp — batched iLQR / Gauss-Newton sweep on top of the step sensitivities.
//
// Build-side component: the reference solves its NLP with IPOPT on one instance (control/base.py:455-477); what is
// built here plays the `loss` / control-limit roles of ControlProblem (control/base.py:323-337,
// control/aircraft.py:29-41, main/control/control.py:35-70) for B independent instances at once:
//
//   cost      J = sum_k 1/2 (x_k - x_ref)' Q (x_k - x_ref) + 1/2 u_k' R u_k  +  1/2 (x_N - x_goal)' Qf (x_N - x_goal)
//   backward  Q-function expansion with the kernel-computed A_k = dF/dx, B_k = dF/du (no second-order dynamics terms),
//             Levenberg regularisation mu on Q_uu, gains  K_k = -Quu^-1 Qux,  k_k = -Quu^-1 Qu
//   forward   closed-loop rollout  u = clip(u_k + alpha k_k + K_k (x - x_k)),  several alpha per instance in one launch
//
// Layouts: X [H+1][13][B], U [H][7][B], A [H][13][13][B], Bm [H][13][7][B], K [H][7][13][B], kff [H][7][B].
#pragma once
#include "ac_kernels_analytic.hpp"

namespace ac {

// quadratic cost + box limits, passed by value
struct IlqrCost {
    float q[13], qf[13], r[7];
    float x_ref[13], x_goal[13];
    float u_min[7], u_max[7];
    float reg;  // Levenberg term added to the diagonal of Quu
};

// Optional per-node, per-instance state cost  l_k(x) = 1/2 sum_j q_k[j] (x_j - xref_k[j])^2 + glin_k . x  (k = 0..H,
// node H is the terminal one).  When q != nullptr these arrays [H+1][13][Bn] replace q / qf / x_ref / x_goal of
// IlqrCost; a candidate batch wider than Bn (line search) reads instance b % Bn.
struct NodeCost {
    const float* __restrict__ q;
    const float* __restrict__ xref;
    const float* __restrict__ glin;
    long Bn;
    AC_DI bool on() const { return q != nullptr; }
    // value and gradient pieces of row j at node k
    // NODE is a compile-time switch of the kernels (constant cost = the original straight-line code); bn = b % Bn
    template <bool NODE>
    AC_DI void row(const IlqrCost& C, long k, bool terminal, int j, long bn, float& qj, float& xr, float& gl) const {
        if constexpr (NODE) {
            const long o = (k * 13 + j) * Bn + bn;
            qj = q[o]; xr = xref[o]; gl = glin[o];
        } else {
            qj = terminal ? C.qf[j] : C.q[j];
            xr = terminal ? C.x_goal[j] : C.x_ref[j];
            gl = 0.f;
        }
    }
};

struct AlphaSet {
    int n;
    float a[8];
};

// ---- backward (Riccati) pass ------------------------------------------------------------------
// 16 lanes per instance (lane j owns column j of the 13-wide matrices / column j of the 7-wide ones), 4 instances per
// 64-thread workgroup, all matrices of an instance in LDS.  Sequential in k; negligible flops (~15 kflop per node)
// next to the linearisation that feeds it, so it is written for clarity, not for the roofline.
constexpr int kIlqrFloats = 1024;  // LDS floats per instance (layout below)

template <bool NODE, bool NEWTON>  // per-node cost arrays or the constant cost; second-order dynamics blocks or none
__global__ __launch_bounds__(64) void k_ilqr_backward(const IlqrCost C, const NodeCost N, const float* __restrict__ X,
                                                      const float* __restrict__ U, const float* __restrict__ A,
                                                      const float* __restrict__ Bm, const float* __restrict__ Hz, long B, long H,
                                                      float* __restrict__ K, float* __restrict__ kff,
                                                      float* __restrict__ dV) {
    __shared__ float smem[4 * kIlqrFloats];
    const int sub = threadIdx.x >> 4, j = threadIdx.x & 15;
    const long b_raw = (long)blockIdx.x * 4 + sub;
    const bool live = b_raw < B;
    const long b = live ? b_raw : B - 1;
    float* S = smem + sub * kIlqrFloats;
    float* sA = S;            // [13][13]
    float* sB = S + 169;      // [13][7]
    float* sV = S + 260;      // [13][13]
    float* sVA = S + 429;     // [13][13]
    float* sVB = S + 598;     // [13][7]
    float* sQux = S + 689;    // [7][13]
    float* sQuu = S + 780;    // [7][7]
    float* sK = S + 829;      // [7][13]
    float* sQxx = S + 920;    // [13] scratch column exchange (unused rows of the block are padding)
    float* svx = S + 940;     // [13]
    float* sqx = S + 953;     // [13]
    float* squ = S + 966;     // [7]
    float* skf = S + 973;     // [7]
    (void)sQxx;

    // terminal condition
    if (j < 13) {
        const float xn = X[(H * 13 + j) * B + b];
        float qj, xr, gl;
        N.template row<NODE>(C, H, true, j, b, qj, xr, gl);
        svx[j] = fmaf(qj, xn - xr, gl);
        for (int i = 0; i < 13; ++i) sV[i * 13 + j] = (i == j) ? qj : 0.f;
    }
    float dv1 = 0.f, dv2 = 0.f;
    __syncthreads();

    for (long k = H - 1; k >= 0; --k) {
        // load A_k, B_k (column j), stage gradients
        float qjj = 0.f;  // this lane's diagonal entry of the stage Hessian
        if (j < 13) {
            for (int i = 0; i < 13; ++i) sA[i * 13 + j] = A[((k * 13 + i) * 13 + j) * B + b];
            const float xk = X[(k * 13 + j) * B + b];
            float xr, gl;
            N.template row<NODE>(C, k, false, j, b, qjj, xr, gl);
            sqx[j] = fmaf(qjj, xk - xr, gl);
        }
        if (j < 7) {
            for (int i = 0; i < 13; ++i) sB[i * 7 + j] = Bm[((k * 13 + i) * 7 + j) * B + b];
            squ[j] = C.r[j] * U[(k * 7 + j) * B + b];
        }
        __syncthreads();
        // VA = V A, VB = V B  (column j)
        if (j < 13) {
            for (int i = 0; i < 13; ++i) {
                float s = 0.f;
                for (int m = 0; m < 13; ++m) s = fmaf(sV[i * 13 + m], sA[m * 13 + j], s);
                sVA[i * 13 + j] = s;
            }
        }
        if (j < 7) {
            for (int i = 0; i < 13; ++i) {
                float s = 0.f;
                for (int m = 0; m < 13; ++m) s = fmaf(sV[i * 13 + m], sB[m * 7 + j], s);
                sVB[i * 7 + j] = s;
            }
        }
        __syncthreads();
        // Qx, Qu, Qxx (column j, in registers), Qux (column j), Quu (column j)
        float qxx[13];
        float qx = 0.f;
        if (j < 13) {
            qx = sqx[j];
            for (int m = 0; m < 13; ++m) qx = fmaf(sA[m * 13 + j], svx[m], qx);
            // Hz (optional): second-order dynamics terms  sum_i lambda_i d2F_i/dz dz  of this node, z = (x, u, dt)
            const float* hz = NEWTON ? Hz + (k * 441) * B + b : nullptr;
            for (int i = 0; i < 13; ++i) {
                float s = (i == j) ? qjj : 0.f;
                if constexpr (NEWTON) s += hz[((long)i * 21 + j) * B];
                for (int m = 0; m < 13; ++m) s = fmaf(sA[m * 13 + i], sVA[m * 13 + j], s);
                qxx[i] = s;
            }
            for (int i = 0; i < 7; ++i) {
                float s = NEWTON ? hz[((long)(13 + i) * 21 + j) * B] : 0.f;
                for (int m = 0; m < 13; ++m) s = fmaf(sB[m * 7 + i], sVA[m * 13 + j], s);
                sQux[i * 13 + j] = s;
            }
        }
        if (j < 7) {
            float qu = squ[j];
            for (int m = 0; m < 13; ++m) qu = fmaf(sB[m * 7 + j], svx[m], qu);
            squ[j] = qu;  // now holds Qu
            for (int i = 0; i < 7; ++i) {
                float s = (i == j) ? C.r[j] + C.reg : 0.f;
                if constexpr (NEWTON) s += Hz[((k * 21 + 13 + i) * 21 + 13 + j) * B + b];
                for (int m = 0; m < 13; ++m) s = fmaf(sB[m * 7 + i], sVB[m * 7 + j], s);
                sQuu[i * 7 + j] = s;
            }
        }
        __syncthreads();
        // Cholesky Quu = L L' (every lane, in registers), then K(:, j) = -Quu^-1 Qux(:, j), kff = -Quu^-1 Qu
        float L[7][7];
        for (int i = 0; i < 7; ++i)
            for (int m = 0; m <= i; ++m) {
                float s = 0.5f * (sQuu[i * 7 + m] + sQuu[m * 7 + i]);
                for (int p = 0; p < m; ++p) s -= L[i][p] * L[m][p];
                L[i][m] = (i == m) ? sqrtf(fmaxf(s, 1e-12f)) : s / L[m][m];
            }
        auto solve = [&](float rhs[7]) {  // in place: rhs <- Quu^-1 rhs
            for (int i = 0; i < 7; ++i) {
                float s = rhs[i];
                for (int p = 0; p < i; ++p) s -= L[i][p] * rhs[p];
                rhs[i] = s / L[i][i];
            }
            for (int i = 6; i >= 0; --i) {
                float s = rhs[i];
                for (int p = i + 1; p < 7; ++p) s -= L[p][i] * rhs[p];
                rhs[i] = s / L[i][i];
            }
        };
        float kf[7];
        for (int i = 0; i < 7; ++i) kf[i] = squ[i];
        solve(kf);
        for (int i = 0; i < 7; ++i) kf[i] = -kf[i];
        float kcol[7];
        if (j < 13) {
            for (int i = 0; i < 7; ++i) kcol[i] = sQux[i * 13 + j];
            solve(kcol);
            for (int i = 0; i < 7; ++i) { kcol[i] = -kcol[i]; sK[i * 13 + j] = kcol[i]; }
            if (live)
                for (int i = 0; i < 7; ++i) K[((k * 7 + i) * 13 + j) * B + b] = kcol[i];
        }
        if (j == 0) {
            for (int i = 0; i < 7; ++i) {
                skf[i] = kf[i];
                if (live) kff[(k * 7 + i) * B + b] = kf[i];
                dv1 += kf[i] * squ[i];
                float s = 0.f;
                for (int m = 0; m < 7; ++m) s += 0.5f * (sQuu[i * 7 + m] + sQuu[m * 7 + i]) * kf[m];
                dv2 += 0.5f * kf[i] * s;
            }
        }
        __syncthreads();
        // V_x(j) = Qx + K' Quu kff + K' Qu + Qux' kff ;  V_xx(:, j) = Qxx + K' Quu K + K' Qux + Qux' K
        float vx = 0.f, vcol[13];
        if (j < 13) {
            float quuk[7], quukf[7];  // Quu K(:, j), Quu kff
            for (int i = 0; i < 7; ++i) {
                float s = 0.f, t = 0.f;
                for (int m = 0; m < 7; ++m) {
                    const float qs = 0.5f * (sQuu[i * 7 + m] + sQuu[m * 7 + i]);
                    s = fmaf(qs, kcol[m], s);
                    t = fmaf(qs, kf[m], t);
                }
                quuk[i] = s; quukf[i] = t;
            }
            vx = qx;
            for (int i = 0; i < 7; ++i) vx += kcol[i] * (quukf[i] + squ[i]) + sQux[i * 13 + j] * kf[i];
            for (int i = 0; i < 13; ++i) {
                float s = qxx[i];
                for (int m = 0; m < 7; ++m) s += sK[m * 13 + i] * (quuk[m] + sQux[m * 13 + j]) + sQux[m * 13 + i] * kcol[m];
                vcol[i] = s;
            }
        }
        __syncthreads();  // every lane has read the old V / Qu before they are overwritten
        if (j < 13) {
            svx[j] = vx;
            for (int i = 0; i < 13; ++i) sV[i * 13 + j] = vcol[i];
        }
        __syncthreads();
        // symmetrise V (each lane fixes its column against the transpose)
        if (j < 13)
            for (int i = 0; i < 13; ++i) vcol[i] = 0.5f * (sV[i * 13 + j] + sV[j * 13 + i]);
        __syncthreads();
        if (j < 13)
            for (int i = 0; i < 13; ++i) sV[i * 13 + j] = vcol[i];
        __syncthreads();
    }
    if (live && j == 0) { dV[b] = dv1; dV[B + b] = dv2; }
}

// ---- costate sweep ------------------------------------------------------------------------------------
// Multipliers of the defect rows x_{k+1} = F(x_k, u_k) at the current iterate (the NLP's lambda estimate):
//   Lam[H-1] = grad l_N(x_N),   Lam[k-1] = grad l_k(x_k) + A_k' Lam[k]
// One lane per instance, sequential in k; feeds ac_shoot_hess_f32 for the exact-Hessian (Newton) backward pass.
template <bool NODE>
__global__ __launch_bounds__(kBlock) void k_ilqr_costate(const IlqrCost C, const NodeCost N, const float* __restrict__ X,
                                                         const float* __restrict__ A, long B, long H,
                                                         float* __restrict__ Lam) {
    const long b = (long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    float lam[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        float qj, xr, gl;
        N.template row<NODE>(C, H, true, j, b, qj, xr, gl);
        lam[j] = fmaf(qj, X[(H * 13 + j) * B + b] - xr, gl);
    }
    for (long k = H - 1;; --k) {
#pragma unroll
        for (int j = 0; j < 13; ++j) Lam[(k * 13 + j) * B + b] = lam[j];
        if (k == 0) break;
        float nxt[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            float qj, xr, gl;
            N.template row<NODE>(C, k, false, j, b, qj, xr, gl);
            float s = fmaf(qj, X[(k * 13 + j) * B + b] - xr, gl);
#pragma unroll
            for (int m = 0; m < 13; ++m) s = fmaf(A[((k * 13 + m) * 13 + j) * B + b], lam[m], s);
            nxt[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) lam[j] = nxt[j];
    }
}

// ---- quadratic trajectory cost ---------------------------------------------------------------------
template <bool NODE>
__global__ __launch_bounds__(kBlock) void k_ilqr_cost(const IlqrCost C, const NodeCost N, const float* __restrict__ X,
                                                      const float* __restrict__ U, long B, long H,
                                                      float* __restrict__ cost) {
    const long b = (long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    float acc = 0.f;
    const long bn = NODE ? b % N.Bn : b;
    for (long k = 0; k <= H; ++k) {
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            float qi, xr, gl;
            N.template row<NODE>(C, k, k == H, i, bn, qi, xr, gl);
            const float x = X[(k * 13 + i) * B + b], d = x - xr;
            acc = fmaf(0.5f * qi * d, d, acc);
            if constexpr (NODE) acc = fmaf(gl, x, acc);
        }
        if (k < H) {
#pragma unroll
            for (int i = 0; i < 7; ++i) { const float u = U[(k * 7 + i) * B + b]; acc = fmaf(0.5f * C.r[i] * u, u, acc); }
        }
    }
    cost[b] = acc;
}

// ---- feedback policy evaluated inside the rollout kernels ---------------------------------------------
// Output instance o = a * B + b (a = line-search index): u = clip(U_k[b] + alpha_a kff_k[b] + K_k[b] (x - Xnom_k[b])).
struct Policy {
    const float* __restrict__ Xnom;  // [H+1][13][B]
    const float* __restrict__ U;     // [H][7][B]
    const float* __restrict__ K;     // [H][7][13][B]
    const float* __restrict__ kff;   // [H][7][B]
    long B;                          // nominal batch
    AlphaSet alphas;
    float u_min[7], u_max[7];

    AC_DI void control(long k, long o, const float x[13], float u[7]) const {
        const long b = o % B;
        const float alpha = alphas.a[(int)(o / B)];
        float dx[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) dx[i] = x[i] - Xnom[(k * 13 + i) * B + b];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            float s = fmaf(alpha, kff[(k * 7 + i) * B + b], U[(k * 7 + i) * B + b]);
#pragma unroll
            for (int m = 0; m < 13; ++m) s = fmaf(K[((k * 7 + i) * 13 + m) * B + b], dx[m], s);
            u[i] = fminf(fmaxf(s, u_min[i]), u_max[i]);
        }
    }
};

// closed-loop rollout, analytic models: one lane per output instance
template <int MODEL>
__global__ __launch_bounds__(64) void k_rollout_policy(const DevParams P, const Policy pol, const float* __restrict__ X0,
                                                       float dt, long Bout, long H, float* __restrict__ Xout,
                                                       float* __restrict__ Uout) {
    const long o = (long)blockIdx.x * 64 + threadIdx.x;
    if (o >= Bout) return;
    float x[13], u[7];
    load_rows<13>(X0, pol.B, o % pol.B, x);
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) { Xout[(long)r * Bout + o] = x[r]; xa[r] = (double)x[r]; }
    AnalyticCoeffs<MODEL> coeffs;
    for (long k = 0; k < H; ++k) {
#pragma unroll
        for (int r = 0; r < 13; ++r) x[r] = (float)xa[r];
        pol.control(k, o, x, u);
#pragma unroll
        for (int r = 0; r < 7; ++r) Uout[(k * 7 + r) * Bout + o] = u[r];
        state_update_carry(P, coeffs, xa, u, dt);
        float* out = Xout + (k + 1) * 13 * Bout;
#pragma unroll
        for (int r = 0; r < 13; ++r) out[(long)r * Bout + o] = (float)xa[r];
    }
}

}  // namespace ac

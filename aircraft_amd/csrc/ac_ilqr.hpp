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
    float u_lin[7];  // linear control cost  u_lin . u_k  per node (the time term: w_time * dt_k on the time row)
    int dt_row;      // <= 0: fixed step (a zeroed struct means that); r > 0: control row r carries dt_k as a decision variable (Policy::step)
};

// Optional per-node, per-instance state cost  l_k(x) = 1/2 sum_j q_k[j] (x_j - xref_k[j])^2 + glin_k . x  (k = 0..H,
// node H is the terminal one).  When q != nullptr these arrays [H+1][13][Bn] replace q / qf / x_ref / x_goal of
// IlqrCost; a candidate batch wider than Bn (line search) reads instance b % Bn.
struct NodeCost {
    const float* __restrict__ q;
    const float* __restrict__ xref;
    const float* __restrict__ glin;
    long Bn;
    // optional per-node, per-instance CONTROL gradient [H][7][Bn] added to Q_u (the backward pass with second-order blocks
    // only: k_ilqr_backward<true, true>): the control-rate term of the goal-acquisition loss (ac_goal.hpp), whose curvature
    // arrives in the (u, u) block of Hz
    const float* __restrict__ uglin = nullptr;
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
// One wave per instance: lane (j, rb) = (t & 15, t >> 4) owns rows rb, rb+4, rb+8, (rb+12) of column j of the 13-wide
// matrices (and of the 7-wide ones), all matrices of the instance in LDS.  The 7x7 Cholesky and the triangular solves
// run redundantly in registers on every lane.  Sequential in k; ~15 kflop per node, so what matters is latency:
//  * the inputs of a node (A_k, B_k, x_k, u_k, its cost rows, its second-order block) are 260-760 floats that sit
//    B floats apart in HBM (the arrays are instance-minor for the kernels that produce them), i.e. one cache line and
//    one page each — fetched on demand they cost ~10 us per node.  They are gathered by LDS-DMA (per-lane source
//    address, lane-linear LDS image) into a ring kRing nodes deep, several nodes ahead of the one being processed,
//    with a counted s_waitcnt vmcnt(N): nothing in the loop waits on an ordinary global load;
//  * one wave = one workgroup, so LDS hand-offs need only lgkmcnt(0), never a cross-wave barrier.
constexpr int kIlqrWork = 736;  // LDS floats of the per-instance work area (layout below)

template <bool NODE, bool NEWTON> struct IlqrRing {
    static constexpr bool kUglin = NODE && NEWTON;               // the control-gradient row (NodeCost::uglin) travels in the ring
    static constexpr int kInstr = 6 + (NEWTON ? 7 : 0) + (kUglin ? 1 : 0);  // LDS-DMA wave-instructions per node
    static constexpr int kNodeFloats = NEWTON ? 768 : 320;      // A 0..168, B 169..259, x 260, u 273, q 280, xref 293, glin 306, Hz 320..760, uglin 761..767
    static constexpr int kDepth = NEWTON ? 5 : 8;                // nodes in flight (kInstr * (kDepth - 1) <= 63)
    static_assert(kInstr * (kDepth - 1) <= 63, "vmcnt is a 6-bit counter");
};

AC_DI void ilqr_glds(const float* g, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_dst, 4, 0, 0);
}
// wave-level hand-off through LDS (single-wave workgroup): LDS traffic retired, no reordering across this point
AC_DI void ilqr_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

template <bool NODE, bool NEWTON>  // per-node cost arrays or the constant cost; second-order dynamics blocks or none
__global__ __launch_bounds__(64) void k_ilqr_backward(const IlqrCost C, const NodeCost N, const float* __restrict__ X,
                                                      const float* __restrict__ U, const float* __restrict__ A,
                                                      const float* __restrict__ Bm, const float* __restrict__ Hz, long B, long H,
                                                      float* __restrict__ K, float* __restrict__ kff,
                                                      float* __restrict__ dV) {
    typedef IlqrRing<NODE, NEWTON> R;
    __shared__ float smem[kIlqrWork + R::kDepth * R::kNodeFloats];  // ONE array: work area, then the node ring
    float* S = smem;
    float* ring = smem + kIlqrWork;
    const int t = threadIdx.x, rb = t >> 4, j = t & 15;
    const long b = blockIdx.x;  // grid = B
    float* sV = S;            // [13][13]
    float* sVA = S + 169;     // [13][13]
    float* sVB = S + 338;     // [13][7]
    float* sQux = S + 429;    // [7][13]
    float* sQuu = S + 520;    // [7][7]
    float* sK = S + 569;      // [7][13]
    float* svx = S + 660;     // [13]
    float* sqx = S + 673;     // [13]
    float* squ = S + 686;     // [7]

    // gather node k into ring slot `slot` (wave-uniform); element e of an array lives B floats after element e-1
    auto issue = [&](long k, int slot) {
        float* dst = ring + slot * R::kNodeFloats;
        const float* a = A + (k * 169) * B + b;
        ilqr_glds(a + (long)t * B, dst);
        ilqr_glds(a + (long)(t + 64) * B, dst + 64);
        if (t < 169 - 128) ilqr_glds(a + (long)(t + 128) * B, dst + 128);
        const float* bm = Bm + (k * 91) * B + b;
        ilqr_glds(bm + (long)t * B, dst + 169);
        if (t < 91 - 64) ilqr_glds(bm + (long)(t + 64) * B, dst + 169 + 64);
        {   // lanes 0-12 x_k, 13-19 u_k, 20-32 q_k, 33-45 xref_k, 46-58 glin_k
            const float* src = X + (k * 13 + t) * B + b;
            if (t >= 13) src = U + (k * 7 + (t - 13)) * B + b;
            if constexpr (NODE) {
                if (t >= 20) src = N.q + (k * 13 + (t - 20)) * B + b;
                if (t >= 33) src = N.xref + (k * 13 + (t - 33)) * B + b;
                if (t >= 46) src = N.glin + (k * 13 + (t - 46)) * B + b;
            }
            if (t < (NODE ? 59 : 20)) ilqr_glds(src, dst + 260);
        }
        if constexpr (NEWTON) {
            const float* hz = Hz + (k * 441) * B + b;
#pragma unroll
            for (int c = 0; c < 7; ++c)
                if (c * 64 + t < 441) ilqr_glds(hz + (long)(c * 64 + t) * B, dst + 320 + c * 64);
        }
        if constexpr (R::kUglin) {  // (always issued, so that the vmcnt bookkeeping is static; without the array: u_k again, unused)
            const float* ug = N.uglin ? N.uglin + (k * 7) * B + b : U + (k * 7) * B + b;
            if (t < 7) ilqr_glds(ug + (long)t * B, dst + 761);
        }
    };
    const float ug_on = (R::kUglin && N.uglin) ? 1.f : 0.f;

    // terminal condition (ordinary loads, before any LDS-DMA is in flight)
    float qterm = 0.f;
    if (j < 13) {
        const float xn = X[(H * 13 + j) * B + b];
        float xr, gl;
        N.template row<NODE>(C, H, true, j, b, qterm, xr, gl);
        if (rb == 0) svx[j] = fmaf(qterm, xn - xr, gl);
        for (int i = rb; i < 13; i += 4) sV[i * 13 + j] = (i == j) ? qterm : 0.f;
    }
    float dv1 = 0.f, dv2 = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int d = 0; d < R::kDepth; ++d)
        if (H - 1 - d >= 0) issue(H - 1 - d, d);
    ilqr_sync();

    for (long it = 0; it < H; ++it) {
        const long k = H - 1 - it;
        const int slot = (int)(it % R::kDepth);
        // node k's gather has landed once at most (kDepth - 1) younger ones are outstanding; in the tail fewer were issued
        if (it + R::kDepth - 1 < H) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R::kInstr * (R::kDepth - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float* nd = ring + slot * R::kNodeFloats;
        const float* sA = nd;         // [13][13]
        const float* sB = nd + 169;   // [13][7]
        // stage gradients
        float qjj = 0.f;  // diagonal entry j of the stage Hessian
        if (j < 13) {
            float xr = C.x_ref[j], gl = 0.f;
            qjj = C.q[j];
            if constexpr (NODE) { qjj = nd[280 + j]; xr = nd[293 + j]; gl = nd[306 + j]; }
            if (rb == 0) sqx[j] = fmaf(qjj, nd[260 + j] - xr, gl);
        }
        if (j < 7 && rb == 0) {
            float g = fmaf(C.r[j], nd[273 + j], C.u_lin[j]);
            if constexpr (R::kUglin) g = fmaf(ug_on, nd[761 + j], g);
            squ[j] = g;
        }
        ilqr_sync();
        // This lane's columns of A_k and B_k stay in registers for the node (read once from the ring slot).
        float acol[13], bcol[13];
#pragma unroll
        for (int m = 0; m < 13; ++m) {
            acol[m] = (j < 13) ? sA[m * 13 + j] : 0.f;
            bcol[m] = (j < 7) ? sB[m * 7 + j] : 0.f;
        }
        // VA = V A, VB = V B: one read of row i of V serves both products
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int i = rb + 4 * n;
            if (i < 13) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int m = 0; m < 13; ++m) {
                    const float v = sV[i * 13 + m];
                    sa = fmaf(v, acol[m], sa);
                    sb = fmaf(v, bcol[m], sb);
                }
                if (j < 13) sVA[i * 13 + j] = sa;
                if (j < 7) sVB[i * 7 + j] = sb;
            }
        }
        ilqr_sync();
        // Qx, Qu (per column, every row group), Qxx rows of column j (registers), Qux, Quu
        float qxx[4] = {0.f, 0.f, 0.f, 0.f};
        float qx = 0.f, qu = 0.f;
        const float* hz = nd + 320;  // [21][21] (NEWTON only)
        {
            float vacol[13], vbcol[13], vxs[13];  // column j of VA / VB, and V_x
#pragma unroll
            for (int m = 0; m < 13; ++m) {
                vacol[m] = (j < 13) ? sVA[m * 13 + j] : 0.f;
                vbcol[m] = (j < 7) ? sVB[m * 7 + j] : 0.f;
                vxs[m] = svx[m];
            }
            qx = (j < 13) ? sqx[j] : 0.f;
            qu = (j < 7) ? squ[j] : 0.f;
#pragma unroll
            for (int m = 0; m < 13; ++m) { qx = fmaf(acol[m], vxs[m], qx); qu = fmaf(bcol[m], vxs[m], qu); }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int i = rb + 4 * n;
                if (i < 13 && j < 13) {
                    float s = (i == j) ? qjj : 0.f;
                    if constexpr (NEWTON) s += hz[i * 21 + j];
#pragma unroll
                    for (int m = 0; m < 13; ++m) s = fmaf(sA[m * 13 + i], vacol[m], s);
                    qxx[n] = s;
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i = rb + 4 * n;
                if (i < 7) {  // column i of B_k serves row i of Qux and of Quu
                    float sx = NEWTON ? ((j < 13) ? hz[(13 + i) * 21 + j] : 0.f) : 0.f;
                    float su = (i == j) ? C.r[j < 7 ? j : 0] + C.reg : 0.f;
                    if constexpr (NEWTON) su += (j < 7) ? hz[(13 + i) * 21 + 13 + j] : 0.f;
#pragma unroll
                    for (int m = 0; m < 13; ++m) {
                        const float bi = sB[m * 7 + i];
                        sx = fmaf(bi, vacol[m], sx);
                        su = fmaf(bi, vbcol[m], su);
                    }
                    if (j < 13) sQux[i * 13 + j] = sx;
                    if (j < 7) sQuu[i * 7 + j] = su;
                }
            }
        }
        ilqr_sync();  // Qux, Quu complete; sqx / squ / svx fully read
        if (j < 7 && rb == 0) squ[j] = qu;  // now holds Qu
        ilqr_sync();
        // Cholesky Quu = L L' (every lane, in registers; reciprocal diagonal via rsq, no divisions), then
        // K(:, j) = -Quu^-1 Qux(:, j), kff = -Quu^-1 Qu
        float Qs[7][7];  // symmetrised Quu
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int m = 0; m <= i; ++m) { Qs[i][m] = 0.5f * (sQuu[i * 7 + m] + sQuu[m * 7 + i]); Qs[m][i] = Qs[i][m]; }
        float L[7][7], rinv[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            float d = Qs[m][m];
#pragma unroll
            for (int p = 0; p < m; ++p) d = fmaf(-L[m][p], L[m][p], d);
            d = fmaxf(d, 1e-12f);
            rinv[m] = __builtin_amdgcn_rsqf(d);
            rinv[m] = rinv[m] * fmaf(-0.5f * d * rinv[m], rinv[m], 1.5f);  // one Newton step: full fp32 accuracy
            L[m][m] = d * rinv[m];
#pragma unroll
            for (int i = m + 1; i < 7; ++i) {
                float s = Qs[i][m];
#pragma unroll
                for (int p = 0; p < m; ++p) s = fmaf(-L[i][p], L[m][p], s);
                L[i][m] = s * rinv[m];
            }
        }
        auto solve = [&](float rhs[7]) {  // in place: rhs <- Quu^-1 rhs
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                float s = rhs[i];
#pragma unroll
                for (int p = 0; p < i; ++p) s = fmaf(-L[i][p], rhs[p], s);
                rhs[i] = s * rinv[i];
            }
#pragma unroll
            for (int i = 6; i >= 0; --i) {
                float s = rhs[i];
#pragma unroll
                for (int p = i + 1; p < 7; ++p) s = fmaf(-L[p][i], rhs[p], s);
                rhs[i] = s * rinv[i];
            }
        };
        float kf[7], quv[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) { quv[i] = squ[i]; kf[i] = quv[i]; }
        solve(kf);
#pragma unroll
        for (int i = 0; i < 7; ++i) kf[i] = -kf[i];
        float kcol[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j < 13) {
#pragma unroll
            for (int i = 0; i < 7; ++i) kcol[i] = sQux[i * 13 + j];
            solve(kcol);
#pragma unroll
            for (int i = 0; i < 7; ++i) kcol[i] = -kcol[i];
            if (rb == 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i) { sK[i * 13 + j] = kcol[i]; K[((k * 7 + i) * 13 + j) * B + b] = kcol[i]; }
            }
        }
        float quukf[7];  // Quu kff (every lane)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < 7; ++m) t = fmaf(Qs[i][m], kf[m], t);
            quukf[i] = t;
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                kff[(k * 7 + i) * B + b] = kf[i];
                dv1 += kf[i] * quv[i];
                dv2 += 0.5f * kf[i] * quukf[i];
            }
        }
        ilqr_sync();  // sK visible
        // V_x(j) = Qx + K' Quu kff + K' Qu + Qux' kff ;  V_xx(i, j) = Qxx + K' Quu K + K' Qux + Qux' K
        float vx = 0.f, vrow[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < 13) {
            float w[7], quxj[7];  // w = Quu K(:, j) + Qux(:, j)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                float sq = 0.f;
#pragma unroll
                for (int m = 0; m < 7; ++m) sq = fmaf(Qs[i][m], kcol[m], sq);
                quxj[i] = sQux[i * 13 + j];
                w[i] = sq + quxj[i];
            }
            vx = qx;
#pragma unroll
            for (int i = 0; i < 7; ++i) vx += kcol[i] * (quukf[i] + quv[i]) + quxj[i] * kf[i];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int i = rb + 4 * n;
                if (i < 13) {
                    float sv = qxx[n];
#pragma unroll
                    for (int m = 0; m < 7; ++m) sv += sK[m * 13 + i] * w[m] + sQux[m * 13 + i] * kcol[m];
                    vrow[n] = sv;
                }
            }
        }
        ilqr_sync();  // every lane has read the old V / Qux before they are overwritten
        if (j < 13) {
            if (rb == 0) svx[j] = vx;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (rb + 4 * n < 13) sV[(rb + 4 * n) * 13 + j] = vrow[n];
        }
        ilqr_sync();
        // symmetrise V: entry (i, j) <- mean with (j, i); both read before either is written
        float sym[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < 13) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (rb + 4 * n < 13) sym[n] = 0.5f * (vrow[n] + sV[j * 13 + rb + 4 * n]);
        }
        ilqr_sync();
        if (j < 13) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (rb + 4 * n < 13) sV[(rb + 4 * n) * 13 + j] = sym[n];
        }
        ilqr_sync();
        if (k - R::kDepth >= 0) issue(k - R::kDepth, slot);  // every read of this slot has retired (lgkmcnt(0) above)
    }
    if (threadIdx.x == 0) { dV[b] = dv1; dV[B + b] = dv2; }
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
            for (int i = 0; i < 7; ++i) {
                const float u = U[(k * 7 + i) * B + b];
                acc = fmaf(0.5f * C.r[i] * u, u, acc);
                acc = fmaf(C.u_lin[i], u, acc);
            }
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
    // Time as a decision variable (the reference's `dt_k` per node, control/base.py:276, 339-385: dt_k = 1 / progress_k^2
    // in 'progress' time, progress_k^2 in 'variable' time, bounded by dt_bounds): control row `dt_row` — a row the force model
    // ignores (a thrust row of the fixed-wing plugin) — carries dt_k itself, clipped to its box like every control; the
    // column of B for it is c = dF/d(dt) of the sensitivity kernels.  dt_row <= 0: the fixed step.
    int dt_row;
    AC_DI float step(const float u[7], float dt) const {
        float r = dt;
#pragma unroll
        for (int i = 1; i < 7; ++i) r = (i == dt_row) ? u[i] : r;
        return r;
    }

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

    // The same control law for kernels where the four lane groups g = 0..3 of a wave hold the same instance (MLP
    // rollouts): group g takes the columns m = g, g+4, g+8, g+12 of K_k, so a lane loads a quarter of the gains, and
    // the partial products are summed across the groups with two cross-lane adds.  load() can be issued a node ahead.
    struct Quarter {
        float kq[7][4], xn[4], ub[7];
    };
    AC_DI void load(long k, long o, int g, Quarter& q) const {
        const long b = o % B;
        const float alpha = alphas.a[(int)(o / B)];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = g + 4 * c;
            const bool on = m < 13;
            const int mm = on ? m : 12;
            q.xn[c] = Xnom[(k * 13 + mm) * B + b];
#pragma unroll
            for (int i = 0; i < 7; ++i) q.kq[i][c] = on ? K[((k * 7 + i) * 13 + mm) * B + b] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) q.ub[i] = fmaf(alpha, kff[(k * 7 + i) * B + b], U[(k * 7 + i) * B + b]);
    }
    AC_DI void control(const Quarter& q, int g, const float x[13], float u[7]) const {
        float dx[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // x[g + 4 c] without dynamic register indexing (c == 3 only exists for g == 0)
            const float x0 = x[4 * c], x1 = (4 * c + 1 < 13) ? x[(4 * c + 1 < 13) ? 4 * c + 1 : 0] : 0.f;
            const float x2 = (4 * c + 2 < 13) ? x[(4 * c + 2 < 13) ? 4 * c + 2 : 0] : 0.f;
            const float x3 = (4 * c + 3 < 13) ? x[(4 * c + 3 < 13) ? 4 * c + 3 : 0] : 0.f;
            const float xs = g == 0 ? x0 : (g == 1 ? x1 : (g == 2 ? x2 : x3));
            dx[c] = xs - q.xn[c];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) s = fmaf(q.kq[i][c], dx[c], s);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            u[i] = fminf(fmaxf(q.ub[i] + s, u_min[i]), u_max[i]);
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
        state_update_carry(P, coeffs, xa, u, pol.step(u, dt));
        float* out = Xout + (k + 1) * 13 * Bout;
#pragma unroll
        for (int r = 0; r < 13; ++r) out[(long)r * Bout + o] = (float)xa[r];
    }
}

}  // namespace ac

// ac_hess_adj.hpp — the second-order blocks kernel built on the reverse sweep of ac_adjoint.hpp (device side: the MLP provider
// over the stage tensors, the kernel).
#pragma once
#include "ac_hess.hpp"
#include "ac_adjoint.hpp"

namespace ac {

// MLP surrogate through the stage tensors (y, J, T) of k_nn_stage_tensors (layout: HessTensorCoeffs, ac_hess.hpp)
// Optional LDS tile [126][64] through which the four waves of a workgroup (four direction groups of the SAME 64 units) can
// share the tensors of the current stage, filled cooperatively at set_stage().  MEASURED AND NOT USED (the kernel passes no
// tile): a quarter of the global loads, but the four waves then meet at 14 barriers per task with every load burst fully
// exposed — 204 800 units of the 4 x 128 net: 15.1 ms against 14.0 (one direction per lane), 14.1 against 13.7 (two).
struct AdjTensorCoeffs : HessTensorCoeffs {
    const float* tp;  // row 0 of the current stage for this lane's unit
    long ts;          // distance between rows
    float* tile;      // LDS [126][64], or nullptr: read global memory directly
    int loaded;
    AC_DI AdjTensorCoeffs(const float* tensors, const UnitAddr& ua, float* lds_tile = nullptr)
        : HessTensorCoeffs(tensors, ua), tp(base), ts(ua.blk), tile(lds_tile), loaded(-1) {}
    AC_DI void set_stage(int s) {
        stage = s;
        if (!tile) { tp = base + (long)s * 126 * blk; ts = blk; return; }
        if (loaded == s) return;  // (the reverse sweep starts at the stage the forward sweep ended with)
        loaded = s;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        __syncthreads();  // every wave has finished reading the previous stage's tile
        const float* src = base + (long)s * 126 * blk;
        for (int idx = wave; idx < 126; idx += nw) tile[idx * 64 + lane] = src[(long)idx * blk];
        __syncthreads();
        tp = tile + lane; ts = 64;
    }
    // value: the second-order Taylor model of the net around the stage's primal inputs, in first-order duals:
    //   C_k = os_k (y_k + J_k . dz) + mean_k  with dz the tangent parts only (the primal IS the expansion point)
    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N>*, const Dual<N> u[7], Dual<N> C[6]) const {
        const Dual<N>* in[5] = {&a.qbar, &a.alpha, &a.beta, &u[0], &u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float os = P.mlp_out_std[k];
            C[k].v = fmaf(tp[(long)k * ts], os, P.mlp_out_mean[k]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) s = fmaf(tp[(long)(6 + k * 5 + j) * ts] * P.mlp_jscale[k][j], in[j]->d[i], s);
                C[k].d[i] = s;
            }
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
    AC_DI void operator()(const DevParams& P, const AeroPre<float>&, const float*, const float u[7], float C[6]) const {
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = fmaf(tp[(long)k * ts], P.mlp_out_std[k], P.mlp_out_mean[k]);
        C[5] += (-0.1f * 6.0f * kDeg) * u[2];
    }
    // adjoint: in_bar_j = sum_k (os_k / std_j) Cbar_k J_kj(z), with J_kj(z) = J_kj + sum_q T_kjq dz_q in duals
    template <int N>
    AC_DI void vjp(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N>*, const Dual<N> u[7], const Dual<N> Cb[6],
                   AeroBar<Dual<N>>& ab, Dual<N> wb[3], Dual<N> ub[7]) const {
        (void)wb;
        const Dual<N>* in[5] = {&a.qbar, &a.alpha, &a.beta, &u[0], &u[1]};
        float is[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) is[j] = 1.0f / P.mlp_in_std[j];
        // S_jq = sum_k os_k Cbar_k.v T_kjq  (symmetric 5 x 5), in z units
        float S[5][5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int q = 0; q < 5; ++q) S[j][q] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float ck = P.mlp_out_std[k] * Cb[k].v;
#pragma unroll
            for (int p = 0; p < 5; ++p)
#pragma unroll
                for (int q = p; q < 5; ++q) {
                    const float tpq = tp[(long)(36 + k * 15 + (p * 5 - p * (p - 1) / 2 + (q - p))) * ts];
                    S[p][q] = fmaf(ck, tpq, S[p][q]);
                }
        }
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int q = 0; q < p; ++q) S[p][q] = S[q][p];
        Dual<N> ib[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            ib[j] = Dual<N>(0.f);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float Jkj = tp[(long)(6 + k * 5 + j) * ts] * P.mlp_out_std[k];
                ib[j].v = fmaf(Jkj, Cb[k].v, ib[j].v);
#pragma unroll
                for (int i = 0; i < N; ++i) ib[j].d[i] = fmaf(Jkj, Cb[k].d[i], ib[j].d[i]);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int i = 0; i < N; ++i) ib[j].d[i] = fmaf(S[j][q] * is[q], in[q]->d[i], ib[j].d[i]);
            ib[j] = ib[j] * is[j];
        }
        ab.qbar = ab.qbar + ib[0]; ab.alpha = ab.alpha + ib[1]; ab.beta = ab.beta + ib[2];
        ub[0] = ub[0] + ib[3]; ub[1] = ub[1] + ib[4];
        ub[2] = ub[2] + (-0.1f * 6.0f * kDeg) * Cb[5];
    }
    AC_DI void vjp(const DevParams& P, const AeroPre<float>&, const float*, const float*, const float Cb[6], AeroBar<float>& ab,
                   float wb[3], float ub[7]) const {
        (void)wb;
        float ib[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) s = fmaf(tp[(long)(6 + k * 5 + j) * ts] * P.mlp_out_std[k], Cb[k], s);
            ib[j] = s / P.mlp_in_std[j];
        }
        ab.qbar += ib[0]; ab.alpha += ib[1]; ab.beta += ib[2];
        ub[0] += ib[3]; ub[1] += ib[4];
        ub[2] += (-0.1f * 6.0f * kDeg) * Cb[5];
    }
};
template <int MODEL> struct AdjProvider { typedef AdjAnalyticCoeffs<MODEL> type; };
template <> struct AdjProvider<AC_MODEL_NN> { typedef AdjTensorCoeffs type; };



// H[za][zb][unit] = sum_i lambda_i d2F_i / dz_a dz_b by the reverse sweep in duals: one WAVE per direction group (N
// directions, 16 / N groups: gridDim.y x 4 waves), lane = unit, so every load and store of a wave is 256 contiguous bytes and
// the seed pattern is wave-uniform.  A lane owns N COLUMNS zb of its unit's block and writes all active rows of them; the
// block is symmetric up to rounding (the jet kernel mirrors one triangle; here both come from their own sweeps).
template <int MODEL, int N>
__global__ __launch_bounds__(kBlock) void k_step_hess_rev(const DevParams P, const float* __restrict__ X,
                                                          const float* __restrict__ U, float dt,
                                                          const float* __restrict__ dt_per_unit,
                                                          const float* __restrict__ Lam,
                                                          const float* __restrict__ stage_tensors, long n, long blk,
                                                          float* __restrict__ Hout) {
    constexpr bool QUAD = MODEL == AC_MODEL_QUAD;
    typedef Dual<N> T;
    const int lane = threadIdx.x & 63;
    int g = blockIdx.y * (kBlock / 64) + (threadIdx.x >> 6);
    g = __builtin_amdgcn_readfirstlane(g);  // wave-uniform
    const long unit_raw = (long)blockIdx.x * 64 + lane;
    const bool live = unit_raw < n;
    const long unit = live ? unit_raw : n - 1;
    const UnitAddr ua(unit, blk);
    float xv[13], uv[7], lam[13];
    load_rows<13>(X, ua, xv);
    load_rows<7>(U, ua, uv);
    load_rows<13>(Lam, ua, lam);
    const float hv = dt_per_unit ? dt_per_unit[unit] : dt;
    T x[13], u[7], h(hv);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        x[i] = T(xv[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) x[i].d[j] = (i >= 3 && (i - 3) == N * g + j) ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        u[i] = T(uv[i]);
        const int dir = QUAD ? (i < 4 ? 10 + i : -1) : ((i < 3) ? 10 + i : (i == 6 ? 13 : -1));
#pragma unroll
        for (int j = 0; j < N; ++j) u[i].d[j] = (dir == N * g + j) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) h.d[j] = (14 == N * g + j) ? 1.f : 0.f;
    T xo[13], gx[13], gu[7], gh;
    __shared__ float stage_words[30 * (N + 1) * kBlock];  // N = 2: 92 KB (one workgroup per CU: the kernel owns the register file anyway)
    auto make_coeffs = [&]() {
        if constexpr (MODEL == AC_MODEL_NN) return AdjTensorCoeffs(stage_tensors, ua, nullptr);
        else return typename AdjProvider<MODEL>::type(stage_tensors, ua);
    };
    auto coeffs = make_coeffs();
    StageLds<N> store(&stage_words[threadIdx.x], kBlock);
    rk4_vjp<T>(P, coeffs, x, u, h, lam, xo, gx, gu, gh, store);
    if (!live) return;
    float* Hu = Hout + ua.late().off(441);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int zb = hess_index<QUAD>(N * g + j);
        if (zb < 0) continue;
#pragma unroll
        for (int i = 3; i < 13; ++i) Hu[((long)i * 21 + zb) * blk] = gx[i].d[j];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const bool active = QUAD ? (i < 4) : (i < 3 || i == 6);
            if (active) Hu[((long)(13 + i) * 21 + zb) * blk] = gu[i].d[j];
        }
        Hu[((long)20 * 21 + zb) * blk] = gh.d[j];
    }
}

}  // namespace ac

// ac_hess_adj.hpp — the second-order blocks kernel built on the reverse sweep of ac_adjoint.hpp (device side: the MLP provider
// over the stage tensors, the kernel).
#pragma once
#include "ac_hess.hpp"
#include "ac_adjoint.hpp"

namespace ac {

// MLP surrogate through the stage tensors (y, J, T) of k_nn_stage_tensors (layout: HessTensorCoeffs, ac_hess.hpp)
// LDS tile: the four waves of a workgroup are four direction groups of the SAME 64 units and read the same 126 rows of every
// stage.  Filled synchronously (cooperative loads, two barriers per stage) the tile was SLOWER than plain global loads — every
// load burst fully exposed: 15.1 ms against 14.0 per 204 800 units of the 4 x 128 net.  Here it is filled by LDS-DMA one stage
// AHEAD: two buffers [2][126][64]; the stage sequence of rk4_vjp is 0 1 2 3 | 3 2 1 0, stage s lives in buffer s & 1, and the
// rows of the next distinct stage are requested right after the barrier that opens the current one (the buffer they land in
// was last read in the step before: every wave is past it).  A wave requests rows wave, wave + 4, ...: one
// global_load_lds_dword per row (64 lanes x 4 B = the row), no vector register involved.
constexpr int kAdjTileFloats = 2 * 126 * 64;
template <bool TILE> struct AdjTensorCoeffsT : HessTensorCoeffs {
    const float* tp;  // !TILE: row 0 of the current stage for this lane's unit in global memory, rows `blk` apart
    float* tile;      // TILE: LDS [2][126][64]
    int cur;          // TILE: float offset of this lane's column of the current stage's buffer
    int step;         // calls of set_stage so far
    AC_DI AdjTensorCoeffsT(const float* tensors, const UnitAddr& ua, float* lds_tile = nullptr)
        : HessTensorCoeffs(tensors, ua), tp(base), tile(lds_tile), cur(0), step(0) {}
    // row r of the current stage
    AC_DI float row(int r) const {
        if constexpr (TILE) return tile[cur + r * 64];
        else return tp[(long)r * blk];
    }
#ifndef AC_HOST_CHECK
    AC_DI void request(int s) const {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        constexpr int nw = kBlock / 64;
        const float* src = base + (long)s * 126 * blk;  // this lane's unit
        float* dst = tile + (s & 1) * (126 * 64);
        for (int r = wave; r < 126; r += nw)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)r * blk),
                                             (__attribute__((address_space(3))) void*)(dst + r * 64), 4, 0, 0);
    }
#endif
    AC_DI void set_stage(int s) {
        stage = s;
        if constexpr (!TILE) { tp = base + (long)s * 126 * blk; return; }
#ifndef AC_HOST_CHECK
        const int i = step++;
        if (i == 4) return;  // the reverse sweep starts on the stage the forward sweep ended with
        if (i == 0) request(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's rows of stage s have landed; every wave is done with the other buffer
        const int nxt = i < 3 ? s + 1 : s - 1;
        if (nxt >= 0) request(nxt);
        cur = (s & 1) * (126 * 64) + (int)(threadIdx.x & 63);
#endif
    }
    // value: the second-order Taylor model of the net around the stage's primal inputs, in first-order duals:
    //   C_k = os_k (y_k + J_k . dz) + mean_k  with dz the tangent parts only (the primal IS the expansion point)
    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N>*, const Dual<N> u[7], Dual<N> C[6]) const {
        const Dual<N>* in[5] = {&a.qbar, &a.alpha, &a.beta, &u[0], &u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float os = P.mlp_out_std[k];
            C[k].v = fmaf(row(k), os, P.mlp_out_mean[k]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) s = fmaf(row(6 + k * 5 + j) * P.mlp_jscale[k][j], in[j]->d[i], s);
                C[k].d[i] = s;
            }
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
    AC_DI void operator()(const DevParams& P, const AeroPre<float>&, const float*, const float u[7], float C[6]) const {
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = fmaf(row(k), P.mlp_out_std[k], P.mlp_out_mean[k]);
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
                    const float tpq = row(36 + k * 15 + (p * 5 - p * (p - 1) / 2 + (q - p)));
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
                const float Jkj = row(6 + k * 5 + j) * P.mlp_out_std[k];
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
            for (int k = 0; k < 6; ++k) s = fmaf(row(6 + k * 5 + j) * P.mlp_out_std[k], Cb[k], s);
            ib[j] = s / P.mlp_in_std[j];
        }
        ab.qbar += ib[0]; ab.alpha += ib[1]; ab.beta += ib[2];
        ub[0] += ib[3]; ub[1] += ib[4];
        ub[2] += (-0.1f * 6.0f * kDeg) * Cb[5];
    }
};
// Cubic fits for the reverse sweep in duals, through PRIMAL tables.  A first-order dual of a fit is its value and gradient at
// the primal point, a dual of its gradient the gradient and the second derivatives there — none of which depends on the
// direction.  Pushing duals through the 34 monomials and the gradient tables instead (AdjAnalyticCoeffs<POLY>, what the host
// build checks) costs ~2 700 instructions per stage in each of the 16 direction waves of 64 units.  Here the four waves of a
// workgroup (four direction groups of the same units) evaluate value, gradient and second derivatives of the six fits ONCE per
// stage of the forward sweep, a quarter each (AnalyticCoeffs<POLY, SHARED>'s split), into LDS — [4 stages][90 rows][64 lanes],
// 92 KB beside the 61 KB of stage states — behind one barrier; every later evaluation of that stage (the forward value, the
// recomputation and the adjoint in the reverse sweep) reads rows.  Row 15 k + (0 | 1..4 | 5..14) = value | gradient | second
// derivatives (PolyTab::hess_row order) of fit k at its own point: (alpha, beta, da, de) for k < 4, (alpha_e, ..) for the
// elevator fit, (alpha, beta_r, ..) for the rudder fit.
constexpr int kAdjPolyRows = 90, kAdjPolyFloats = 4 * kAdjPolyRows * 64;
struct AdjPolyTabCoeffs {
    static constexpr int kModel = AC_MODEL_POLY;
    static constexpr bool kFusedTangent = false;
    float* tl;   // LDS base + lane
    int wave;    // this wave's quarter (wave-uniform)
    int stage, step;
    AC_DI AdjPolyTabCoeffs(float* lds, int wave_) : tl(lds + (threadIdx.x & 63)), wave(wave_), stage(0), step(0) {}
    AC_DI void set_stage(int s) { stage = s; ++step; }
    AC_DI bool forward_sweep() const { return step <= 4; }  // set_stage calls 1..4: the forward sweep (stages 0..3)
    AC_DI const float* rows() const { return tl + stage * (kAdjPolyRows * 64); }
    // write value, gradient and second derivatives of the fits ks[] at the float point f
    template <int NOUT> AC_DI void tabulate(const DevParams& P, const int (&ks)[NOUT], const float f[4]) {
        float val[NOUT], g[NOUT][4], h[NOUT][10];
        poly_value_grad<NOUT>(P, ks, f, val, g);
        poly_hess<NOUT>(P, ks, f, h);
        float* r = tl + stage * (kAdjPolyRows * 64);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float* rk = r + ks[o] * 15 * 64;
            rk[0] = val[o];
#pragma unroll
            for (int v = 0; v < 4; ++v) rk[(1 + v) * 64] = g[o][v];
#pragma unroll
            for (int e = 0; e < 10; ++e) rk[(5 + e) * 64] = h[o][e];
        }
    }
    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7], Dual<N> C[6]) {
        typedef Dual<N> T;
        const T* w = &x[10];
        const float eps = P.p.epsilon, arm = P.p.rudder_moment_arm, b4 = P.p.b * 0.25f;
        const T ux = a.vr[0] + eps;
        const T alpha_e = m_atan2(a.vr[2] + arm * w[1], ux);
        const T alpha_l = m_atan2(a.vr[2] - b4 * w[0], ux);
        const T alpha_r = m_atan2(a.vr[2] + b4 * w[0], ux);
        const T vy = a.vr[1] - arm * w[2];
        const T beta_r = m_asin(vy / m_sqrt(a.vr[0] * a.vr[0] + vy * vy + a.vr[2] * a.vr[2] + eps));
        if (forward_sweep()) {
            if (wave < 2) {
                const float f[4] = {a.alpha.v, a.beta.v, u[0].v, u[1].v};
                const int ks[2] = {2 * wave, 2 * wave + 1};
                tabulate<2>(P, ks, f);
            } else {
                const bool el = wave == 2;
                const float f[4] = {el ? alpha_e.v : a.alpha.v, el ? a.beta.v : beta_r.v, u[0].v, u[1].v};
                const int ks[1] = {el ? 4 : 5};
                tabulate<1>(P, ks, f);
            }
            __syncthreads();
        }
        const float* r = rows();
        const T* fm[4] = {&a.alpha, &a.beta, &u[0], &u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const T* f0 = k == 4 ? &alpha_e : fm[0];
            const T* f1 = k == 5 ? &beta_r : fm[1];
            const float* rk = r + k * 15 * 64;
            C[k].v = rk[0];
            const float g0 = rk[1 * 64], g1 = rk[2 * 64], g2 = rk[3 * 64], g3 = rk[4 * 64];
#pragma unroll
            for (int i = 0; i < N; ++i) C[k].d[i] = fmaf(g0, f0->d[i], fmaf(g1, f1->d[i], fmaf(g2, u[0].d[i], g3 * u[1].d[i])));
        }
        C[3] = C[3] + (b4 * 0.5f) * (poly_cz_alpha_only(P, alpha_r) - poly_cz_alpha_only(P, alpha_l));
        C[5] = C[5] + (0.01f * 6.0f * kDeg) * u[2];
    }
    template <int N>
    AC_DI void vjp(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7], const Dual<N> Cb[6],
                   AeroBar<Dual<N>>& ab, Dual<N> wb[3], Dual<N> ub[7]) const {
        typedef Dual<N> T;
        const float* r = rows();
        // the gradient of fit k at its point f as a dual: (g_v, sum_q h_vq df_q)
        auto grad = [&](int k, const T f[4], T g[4]) {
            const float* rk = r + k * 15 * 64;
            float h[4][4];
            int e = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int q = v; q < 4; ++q, ++e) { h[v][q] = rk[(5 + e) * 64]; h[q][v] = h[v][q]; }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                g[v].v = rk[(1 + v) * 64];
#pragma unroll
                for (int i = 0; i < N; ++i)
                    g[v].d[i] = fmaf(h[v][0], f[0].d[i], fmaf(h[v][1], f[1].d[i], fmaf(h[v][2], f[2].d[i], h[v][3] * f[3].d[i])));
            }
        };
        poly_vjp(P, a, x, u, Cb, ab, wb, ub, grad);
    }
};

template <int MODEL> struct AdjProvider { typedef AdjAnalyticCoeffs<MODEL> type; };
template <> struct AdjProvider<AC_MODEL_NN> { typedef AdjTensorCoeffsT<false> type; };
typedef AdjTensorCoeffsT<false> AdjTensorCoeffs;



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
        if constexpr (MODEL == AC_MODEL_NN) {
#ifdef AC_HESS_NO_TILE  // (A/B flavour: the stage tensors by plain global loads)
            return AdjTensorCoeffsT<false>(stage_tensors, ua, nullptr);
#else
            __shared__ float tensor_tile[kAdjTileFloats];
            return AdjTensorCoeffsT<true>(stage_tensors, ua, tensor_tile);
#endif
        }
        else if constexpr (MODEL == AC_MODEL_POLY) {
#ifdef AC_HESS_POLY_DUALS  // (A/B flavour: duals through the monomials and the gradient tables in every direction wave)
            return AdjAnalyticCoeffs<MODEL>(stage_tensors, ua);
#else
            static_assert(kBlock / 64 == 4, "four direction waves per workgroup");
            __shared__ float poly_points[kAdjPolyFloats];
            return AdjPolyTabCoeffs(poly_points, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
#endif
        }
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

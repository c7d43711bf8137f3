// ac_hess_nn.hpp — second-order step sensitivities with the MLP surrogate (NN part).
//
// Two kernels.  The second derivatives of the RK4 step only need the network's value, Jacobian J = dy/dz (6 x 5) and
// second-derivative tensor T = d2y/dz dz (6 x 15 symmetric pairs) at the four stage points, and those points depend on
// the PRIMAL trajectory of the step alone:
//   k_nn_stage_tensors   walks the primal RK4 stages of 16 units per wave and evaluates y, J, T at each stage with the
//                        MFMA engine in second-order mode: per pass the slabs are value, d/dz_p, d/dz_q, d2/dz_p2,
//                        d2/dz_q2, d2/dz_p dz_q for one input pair (p, q); ten passes cover all pairs of the five inputs.
//                        Output per unit: [4 stages][126 = 6 + 30 + 90] floats.
//   k_step_hess<NN>      (ac_hess.hpp) the same second-order forward-mode kernel as for the analytic models, with a
//                        coefficient provider that applies the chain rule through the stored (y, J, T).
#pragma once
#include "ac_kernels_nn.hpp"

namespace ac {

constexpr int kStageRows = 126;                // y[6], J[6][5], T[6][15]
constexpr int kStageFloats = 4 * kStageRows;   // per unit
AC_DI constexpr int pair_index(int p, int q) { return p * 5 - p * (p - 1) / 2 + (q - p); }  // p <= q < 5

// coefficients from a known network value (primal stage advance inside k_nn_stage_tensors)
struct GivenY {
    static constexpr int kModel = AC_MODEL_NN;
    float y[6];
    AC_DI void operator()(const DevParams& P, const AeroPre<float>&, const float*, const float u[7], float C[6]) const {
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
        C[5] += (-0.1f * 6.0f * kDeg) * u[2];
    }
};

template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_stage_tensors(const DevParams P, const MlpPlan plan,
                                                                const float* __restrict__ blob,
                                                                const float* __restrict__ X, const float* __restrict__ U,
                                                                float dt, const float* __restrict__ dt_per_unit, long n,
                                                                long blk, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngine<6, WT, USE_MFMA, false, true> eng(plan, blob, smem);
    eng.load_weights();
    const WaveUnit w(n, blk);
    float x0[13], u[7];
    load_rows<13>(X, w.ua, x0);
    load_rows<7>(U, w.ua, u);
    const float h = dt_per_unit ? dt_per_unit[w.unit] : dt;
    float xs[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) xs[i] = x0[i];
#pragma nounroll
    for (int s = 0; s < 4; ++s) {
        AeroPre<float> a;
        aero_pre(P, xs, a);
        const float in[5] = {a.qbar, a.alpha, a.beta, u[0], u[1]};
        float z[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
        GivenY prov;
        float* o = out + w.ua.off(kStageFloats) + (long)s * kStageRows * blk;
        int pass = 0;
#pragma nounroll
        for (int p = 0; p < 4; ++p) {
#pragma nounroll
            for (int q = p + 1; q < 5; ++q, ++pass) {
                eng.set_pair(p, q);
                float yy[6], D[6][5];
                eng.forward(z, yy, D);
                if (pass == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) prov.y[k] = yy[k];
                }
                if (w.live && w.g == (pass & 3)) {  // the four lane groups of a unit hold the same results
                    if (pass == 0) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) o[(long)k * blk] = yy[k];
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        o[(long)(6 + k * 5 + p) * blk] = D[k][0];
                        o[(long)(6 + k * 5 + q) * blk] = D[k][1];
                        o[(long)(36 + k * 15 + pair_index(p, p)) * blk] = D[k][2];
                        o[(long)(36 + k * 15 + pair_index(q, q)) * blk] = D[k][3];
                        o[(long)(36 + k * 15 + pair_index(p, q)) * blk] = D[k][4];
                    }
                }
            }
        }
        if (s < 3) {  // next primal stage point
            float k1[13];
            state_derivative<float>(P, prov, xs, u, k1);
            const float hs = h * ((s == 2) ? 1.0f : 0.5f);
#pragma unroll
            for (int i = 0; i < 13; ++i) xs[i] = fmaf(hs, k1[i], x0[i]);
        }
    }
    eng.drain();
}

}  // namespace ac

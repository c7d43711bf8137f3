// ac_hess_nn.hpp — second-order step sensitivities with the MLP surrogate (NN part).
//
// Two kernels.  The second derivatives of the RK4 step only need the network's value, Jacobian J = dy/dz (6 x 5) and
// second-derivative tensor T = d2y/dz dz (6 x 15 symmetric pairs) at the four stage points, and those points depend on
// the PRIMAL trajectory of the step alone:
//   k_nn_stage_tensors   walks the primal RK4 stages of 16 units per wave and evaluates y, J, T at each stage with the
//                        MFMA engine in second-order mode: per pass ten slabs — value, the three first and the six second
//                        derivatives for one input triple (p, q, r) — which this kernel's register file holds because it
//                        carries no dual RK4 state; four triples {0,1,2} {0,3,4} {1,3,4} {2,3,4} cover all fifteen pairs
//                        of the five inputs: 40 slab evaluations per stage (ten passes over input PAIRS with six slabs
//                        each needed 60).  Widths <= 64: one pass of 21 slabs over all five inputs.
//                        Width 128 runs as two launches: PART 0 = the triples {0,1,2} and {2,3,4} (ten slabs each),
//                        PART 1 = one bipartite pass (nine slabs: value, d/dz of 0, 1, 3, 4 and the four cross pairs
//                        03, 04, 13, 14): 29 slab evaluations per stage.
//                        Output per unit: [4 stages][126 = 6 + 30 + 90] floats.
//   k_step_hess<NN>      (ac_hess.hpp) the same second-order forward-mode kernel as for the analytic models, with a
//                        coefficient provider that applies the chain rule through the stored (y, J, T).
#pragma once
#include "ac_kernels_nn.hpp"
#ifdef AC_EXP_ZERO_REGS
#include "../../tools/experiments/ac_exp_zero_regs.inc"  // (experiment flavour only: not a product header)
#endif

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

template <int WT, bool USE_MFMA, int PART>
__global__ __launch_bounds__(kBlock, 1) void k_nn_stage_tensors(const DevParams P, const MlpPlan plan,
                                                                const float* __restrict__ blob,
                                                                const float* __restrict__ X, const float* __restrict__ U,
                                                                float dt, const float* __restrict__ dt_per_unit, long n,
                                                                long blk, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef AC_EXP_ZERO_REGS  // (experiment flavour only: tools/archive/bisect_exp_last2.sh)
    AC_ZERO_ALL_REGS();
    for (int i = threadIdx.x; i < plan.lds_total / 4; i += blockDim.x) reinterpret_cast<volatile float*>(smem)[i] = 0.f;  // and the LDS
    __syncthreads();
#endif
    // width <= 64: all five inputs in one pass (21 slabs of 4 WT registers); width 128: four passes over input triples
    constexpr bool kSingle = WT <= 4;
    // width 128: two triples here + the bipartite pass in a second launch (PART 1).  (Not instantiated for the VALU
    // validation flavour: tools/gen_nn_units.py.)
    constexpr bool kTwoPart = !kSingle;
    static_assert(PART == 0 || (PART == 1 && kTwoPart), "PART 1 (the bipartite pass) exists for width 128 only");
    MlpEngine<kSingle ? 21 : (PART == 1 ? 9 : 10), WT, USE_MFMA, false, true> eng(plan, blob, smem);
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
        if constexpr (kSingle) {
            float yy[6], D[6][20];  // D[k] = d/dz_0..4, d2/dz_0..4 ^2, then the mixed pairs (0,1) (0,2) .. (3,4)
            eng.forward(z, yy, D);
#pragma unroll
            for (int k = 0; k < 6; ++k) prov.y[k] = yy[k];
            if (w.live) {  // the four lane groups of a unit hold the same results: y and J from group 0, T split by output
                if (w.g == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        o[(long)k * blk] = yy[k];
#pragma unroll
                        for (int i = 0; i < 5; ++i) o[(long)(6 + k * 5 + i) * blk] = D[k][i];
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (w.g == 1 + (k >> 1)) {
                        int m = 10;
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            o[(long)(36 + k * 15 + pair_index(i, i)) * blk] = D[k][5 + i];
#pragma unroll
                            for (int j = i + 1; j < 5; ++j, ++m) o[(long)(36 + k * 15 + pair_index(i, j)) * blk] = D[k][m];
                        }
                    }
                }
            }
        } else if constexpr (PART == 1) {
            eng.set_quad(0, 1, 3, 4);
            float yy[6], D[6][8];  // D[k] = d/dz_0, 1, 3, 4 (PART 0 stores those), then the cross pairs 03, 04, 13, 14
            eng.forward(z, yy, D);
#pragma unroll
            for (int k = 0; k < 6; ++k) prov.y[k] = yy[k];
            if (w.live && w.g == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    o[(long)(36 + k * 15 + pair_index(0, 3)) * blk] = D[k][4];
                    o[(long)(36 + k * 15 + pair_index(0, 4)) * blk] = D[k][5];
                    o[(long)(36 + k * 15 + pair_index(1, 3)) * blk] = D[k][6];
                    o[(long)(36 + k * 15 + pair_index(1, 4)) * blk] = D[k][7];
                }
            }
        } else {
#pragma nounroll
        for (int pass = 0; pass < 2; ++pass) {
            // triples {0,1,2} {2,3,4}; the cross pairs 03 04 13 14 come from PART 1
            const int tp = pass == 0 ? 0 : 2, tq = pass == 0 ? 1 : 3, tr = pass == 0 ? 2 : 4;
            eng.set_triple(tp, tq, tr);
            float yy[6], D[6][9];
            eng.forward(z, yy, D);
            if (pass == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) prov.y[k] = yy[k];
            }
            if (w.live && w.g == pass) {  // the four lane groups of a unit hold the same results: one pass each
                if (pass == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[(long)k * blk] = yy[k];
                }
                const int t3[3] = {tp, tq, tr};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        o[(long)(6 + k * 5 + t3[i]) * blk] = D[k][i];
                        o[(long)(36 + k * 15 + pair_index(t3[i], t3[i])) * blk] = D[k][3 + i];
                    }
                    o[(long)(36 + k * 15 + pair_index(tp, tq)) * blk] = D[k][6];
                    o[(long)(36 + k * 15 + pair_index(tp, tr)) * blk] = D[k][7];
                    o[(long)(36 + k * 15 + pair_index(tq, tr)) * blk] = D[k][8];
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

// ac_hess_rev.hpp — stage tensors (y, J, T) of the width-128 surrogate by forward tangents + a reverse sweep
// (NN part; the second algorithm behind ac_step_hess_f32 / ac_shoot_hess_f32).
//
// k_nn_stage_tensors (ac_hess_nn.hpp) carries one slab per second derivative through the net: 21 slabs for the five inputs,
// which the register file holds up to width 64; at width 128 it needs three passes and 29 slab evaluations per stage.  The
// same tensor is a sum over the hidden layers,
//     T[k][ab] = d2 y_k / dz_a dz_b = sum_l sum_n  R_l[k][n] * act''(a_l[n]) * t_l^a[n] * t_l^b[n],
// with t_l^a = d a_l / dz_a the forward tangents (a_l: pre-activations) and R_l[k][n] = d y_k / d h_l[n] the six output rows
// pulled back to layer l:  R_top = W_last,  R_l = (R_{l+1} . act'(a_{l+1})) W_l.  So per stage:
//   forward   value + five tangents through the net, the six-slab engine of the step kernels (MlpEngine<6, WT, true>); the
//             state after every hidden layer but the top one goes to a per-wave scratch in global memory (h_1 alone for the
//             first hidden layer: its tangents are act'(a_1) times the columns of W0);
//   reverse   six slabs again — the rows R[k] — through the TRANSPOSED hidden blocks the host appends to the weight blob
//             (plan_rev: the ring streams L1, L2, L3, L3', L2', L1' cyclically), no bias, no activation;
//   contract  at every hidden layer, per unit and on the vector ALUs:  T[k][ab] += R[k][n] * P[n][ab] over the lane's 2 WT
//             neurons, P[n][ab] = -2 h s^a s^b / (1 - h^2) from the post-activation tangents s = act'(a) t the engine holds;
//             the four lanes of a unit add their parts at the end (permlane swaps).
// 12 slab-layer products per hidden layer instead of 29, + the contraction (~1.5 slab-layer products' worth of vector
// instructions per layer).  tanh on every hidden layer (ac_set_mlp's fold guarantees it); an activation on the LAST layer
// adds act'(p_k) to R_top and the term act''(p_k) J_a J_b.
#pragma once
#include "ac_hess_nn.hpp"

namespace ac {

template <int WT>
struct MlpEngineRev : MlpEngine<6, WT, true, true> {
    typedef MlpEngine<6, WT, true, true> Base;
    using Base::a;
    using Base::g;
    using Base::lane;
    static_assert(!Base::kSpread, "the reverse engine uses the burst form of the ring copy (build this unit with AC_CH = 4)");

    AC_DI MlpEngineRev(const MlpPlan& pl, const float* blob, char* lds_base) : Base(pl, blob, lds_base) {}

    // a[s] <- B a[s] for the six slabs: B the packed (transposed) block at wl, no bias, no activation.  Same block walk as
    // MlpEngine::gemm_chunk: four output tiles with independent accumulators, the next block's fragments fetched behind
    // k-step 0 of the current one.
    AC_DI void layer_raw(const char* wl) {
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        constexpr int C = WT < 4 ? WT : 4;
        f32x4 wcur[C], wnext[C];
#pragma unroll
        for (int i = 0; i < C; ++i) wcur[i] = wf[(i * WT + 0) * 64];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            f32x4 o[WT];
#pragma unroll
            for (int nc = 0; nc < WT; nc += C) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[C];
#pragma unroll
                for (int i = 0; i < C; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < WT; ++kt) {
                    const bool last = kt + 1 == WT;
                    const bool fetch = !last || !(s == 5 && nc + C >= WT);
#pragma unroll
                    for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<true>(wcur[i][0], a[s][kt][0], acc[i]);
                    if (fetch) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int nnc = last ? (nc + C) % WT : nc, nkt = last ? 0 : kt + 1;
#pragma unroll
                        for (int i = 0; i < C; ++i) wnext[i] = wf[((nnc + i) * WT + nkt) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 1; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<true>(wcur[i][r], a[s][kt][r], acc[i]);
                    if (fetch) {
#pragma unroll
                        for (int i = 0; i < C; ++i) wcur[i] = wnext[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < C; ++i) o[nc + i] = acc[i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[s][nt][r] = o[nt][r];
        }
    }

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // One neuron of one unit: acc[k][ab] += R[k] * c s^a s^b for the pairs a <= b in pair_index() order.  The accumulators are
    // pairs over the outputs, acc[kp][ab] = (k = 2 kp, k = 2 kp + 1), so one v_pk_fma_f32 with the product broadcast from either
    // half of a register pair serves two outputs: 45 packed FMAs per neuron.  (Written as instructions: left to itself the
    // compiler re-vectorises the 90 scalar chains across neurons and spills ~500 registers doing so.)
    AC_DI static void contract1(float c, const float (&s)[5], const f32x2 (&R)[3], f32x2 (&acc)[3][15]) {
        float u[5], p[16];
#pragma unroll
        for (int i = 0; i < 5; ++i) u[i] = c * s[i];
        int pi = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i; j < 5; ++j, ++pi) p[pi] = u[i] * s[j];
        p[15] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x2 pp = {p[2 * q], p[2 * q + 1]};
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) {
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[kp][2 * q]) : "v"(R[kp]), "v"(pp));
                if (2 * q + 1 < 15)
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[kp][2 * q + 1]) : "v"(R[kp]), "v"(pp));
            }
        }
    }
    // The accumulators must sit in architectural VGPRs while a contraction runs (every packed FMA reads and writes them) and the
    // 192 slab registers in accumulation registers (each is read once per contraction): left alone, the allocator keeps the
    // slabs — MFMA operands a moment ago — in the VGPRs and wraps every FMA in two v_accvgpr_read and two v_accvgpr_write.
    AC_DI void pin(f32x2 (&acc)[3][15]) {
#pragma unroll
        for (int kp = 0; kp < 3; ++kp)
#pragma unroll
            for (int i = 0; i < 15; ++i) asm volatile("" : "+v"(acc[kp][i]));
#pragma unroll
        for (int sl = 0; sl < 6; ++sl)
#pragma unroll
            for (int t = 0; t < WT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+a"(a[sl][t][r]));
    }
    // act''/act' of a tanh neuron with output h, as the second-order epilogue of MlpEngine forms it: -2 h / (1 - h^2),
    // 0 for a saturated neuron (its tangents are 0 as well)
    AC_DI static float curv_over_slope(float h, float sp) { return sp > 1e-30f ? -2.0f * h / sp : 0.f; }

    // Top hidden layer: its state (h, s^1..5) is in the registers; R = the rows of the last layer from the `wlt` image of
    // last_valu ([tile][lane group][6 float4] = {W(2kp, n), W(2kp+1, n), W(2kp, n+1), W(2kp+1, n+1)}, ac_set_mlp), times rs[k]
    // (act' of an activated last layer, else 1).  Leaves a[k] = R[k] . act'(a_top): the input of the first reverse product.
    AC_DI void contract_top(const char* wl_last, const float (&rs)[6], f32x2 (&acc)[3][15]) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl_last + 1024) + g * 6;
        pin(acc);
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            f32x4 w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = wv[t * 24 + i];
            __builtin_amdgcn_sched_barrier(0);  // (one tile at a time: the scheduler would otherwise hoist every tile's loads)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 R[3];
                float s[5];
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) {
                    const f32x4 q = w[2 * kp + (r >> 1)];
                    R[kp] = f32x2{q[2 * (r & 1)] * rs[2 * kp], q[2 * (r & 1) + 1] * rs[2 * kp + 1]};
                }
                const float h = a[0][t][r], sp = fmaf(-h, h, 1.0f);
#pragma unroll
                for (int i = 0; i < 5; ++i) s[i] = a[1 + i][t][r];
                contract1(curv_over_slope(h, sp), s, R, acc);
#pragma unroll
                for (int k = 0; k < 6; ++k) a[k][t][r] = R[k >> 1][k & 1] * sp;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // A hidden layer between: a[k] = R[k] (fresh from layer_raw), the layer's state from this wave's scratch
    // ([6 slabs][WT tiles][64 lanes] float4, as store_state wrote it).  Leaves a[k] = R[k] . act'.
    AC_DI void contract_mid(const f32x4* __restrict__ sl, f32x2 (&acc)[3][15]) {
        f32x4 cur[6], nxt[6];
        pin(acc);
#pragma unroll
        for (int i = 0; i < 6; ++i) cur[i] = sl[(i * WT + 0) * 64 + lane];
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            if (t + 1 < WT) {
#pragma unroll
                for (int i = 0; i < 6; ++i) nxt[i] = sl[(i * WT + t + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 R[3];
                float s[5];
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) R[kp] = f32x2{a[2 * kp][t][r], a[2 * kp + 1][t][r]};
                const float h = cur[0][r], sp = fmaf(-h, h, 1.0f);
#pragma unroll
                for (int i = 0; i < 5; ++i) s[i] = cur[1 + i][r];
                contract1(curv_over_slope(h, sp), s, R, acc);
#pragma unroll
                for (int k = 0; k < 6; ++k) a[k][t][r] = R[k >> 1][k & 1] * sp;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) cur[i] = nxt[i];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The first hidden layer under at least one hidden product: h_1 from the scratch ([WT][64] float4), the tangents are
    // s^a = (1 - h^2) W0[n][a] (the transposed W0 of first_valu's block), so c s^a s^b = -2 h (1 - h^2) W0[n][a] W0[n][b].
    AC_DI void contract_bottom(const f32x4* __restrict__ sl, const char* wl0, f32x2 (&acc)[3][15]) {
        const f32x4* w0t = reinterpret_cast<const f32x4*>(wl0 + 1024);
        pin(acc);
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            const f32x4 h4 = sl[t * 64 + lane];
            f32x4 w[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) w[j] = w0t[j * (WT * 4) + 4 * t + g];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 R[3];
                float s[5];
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) R[kp] = f32x2{a[2 * kp][t][r], a[2 * kp + 1][t][r]};
                const float h = h4[r], sp = fmaf(-h, h, 1.0f);
#pragma unroll
                for (int i = 0; i < 5; ++i) s[i] = w[i][r];
                contract1(-2.0f * h * sp, s, R, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // slabs [0, NS) of the state to the scratch
    template <int NS> AC_DI void store_state(f32x4* __restrict__ sl) const {
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int t = 0; t < WT; ++t) sl[(i * WT + t) * 64 + lane] = f32x4{a[i][t][0], a[i][t][1], a[i][t][2], a[i][t][3]};
    }
};

// float4 of scratch per wave: h_1 + the full state of the hidden layers 2 .. nh
constexpr long rev_scratch_f32x4(int wt, int nh) { return (long)(1 + 6 * (nh > 1 ? nh - 1 : 0)) * wt * 64; }

// Persistent workgroups (one per CU: the 131 KB weight plan), task = 64 units.  `plan` = plan_rev: entries 0 .. L-1 the net as
// in plan_sens, L .. L+nh-1 the transposed hidden blocks of the layers nh, nh-1, .., 1 (nh = L - 2 >= 1).
template <int WT>
__global__ __launch_bounds__(kBlock, 1) void k_nn_stage_tensors_rev(const DevParams P, const MlpPlan plan,
                                                                    const float* __restrict__ blob,
                                                                    const float* __restrict__ X, const float* __restrict__ U,
                                                                    float dt, const float* __restrict__ dt_per_unit, long n,
                                                                    long blk, int L, float* __restrict__ scratch,
                                                                    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngineRev<WT> eng(plan, blob, smem);
    eng.load_weights();
    const int nh = L - 2;
    f32x4* const mine = reinterpret_cast<f32x4*>(scratch) +
                        ((long)blockIdx.x * (kBlock >> 6) + (threadIdx.x >> 6)) * rev_scratch_f32x4(WT, nh);
    const long ntasks = (n + 63) / 64;
#pragma nounroll
    for (long task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const WaveUnit w((int)threadIdx.x, task, n, blk);
        float x0[13], u[7];
        load_rows<13>(X, w.ua, x0);
        load_rows<7>(U, w.ua, u);
        const float h = dt_per_unit ? dt_per_unit[w.unit] : dt;
        float xs[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xs[i] = x0[i];
#pragma nounroll
        for (int s = 0; s < 4; ++s) {
            AeroPre<float> ap;
            aero_pre(P, xs, ap);
            const float in[5] = {ap.qbar, ap.alpha, ap.beta, u[0], u[1]};
            float z[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
            // ---- forward: value + five tangents, the states below the top hidden layer to the scratch
            const char* wl0 = eng.acquire(0);
            eng.first_valu(wl0, z);
            eng.template store_state<1>(mine);
#pragma nounroll
            for (int l = 1; l <= nh; ++l) {
                const char* wl = eng.acquire(l);
                eng.template layer<WT, WT, 1>(wl, 1);
                if (l < nh) eng.template store_state<6>(mine + WT * 64 + (long)(l - 1) * 6 * WT * 64);
            }
            float y[6], J[6][5];
            const char* wll = eng.acquire(L - 1);
            const int act_last = plan.act[L - 1];
            eng.template last_valu<5>(wll, act_last, y, J);
            // ---- outputs y and J from lane group 0 (J is dead afterwards)
            float* o = out + w.ua.off(kStageFloats) + (long)s * kStageRows * blk;
            if (w.live && w.g == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    o[(long)k * blk] = y[k];
#pragma unroll
                    for (int i = 0; i < 5; ++i) o[(long)(6 + k * 5 + i) * blk] = J[k][i];
                }
            }
            // ---- reverse sweep with the contraction at every hidden layer
            float rs[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) rs[k] = act_last ? fmaf(-y[k], y[k], 1.0f) : 1.0f;
            typename MlpEngineRev<WT>::f32x2 acc[3][15];
            {   // y = tanh(p) on the last layer: + act''(p_k) dp/dz_a dp/dz_b = -2 y_k J_a J_b / act'(p_k), once per unit (the
                // four lanes' sums are added at the end); zero otherwise
                const bool fix = act_last && w.g == 0;
                int pi = 0;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = i; j < 5; ++j, ++pi)
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            acc[k >> 1][pi][k & 1] = fix ? MlpEngineRev<WT>::curv_over_slope(y[k], rs[k]) * J[k][i] * J[k][j] : 0.f;
            }
            eng.contract_top(wll, rs, acc);
#pragma nounroll
            for (int l = nh; l >= 1; --l) {
                eng.layer_raw(eng.acquire(L + nh - l));
                if (l > 1) eng.contract_mid(mine + WT * 64 + (long)(l - 2) * 6 * WT * 64, acc);
                else eng.contract_bottom(mine, wl0, acc);
            }
            // ---- the 90 sums of T reduce-scattered over the unit's four lanes
#pragma unroll
            for (int q = 0; q < 23; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * q + e;
                    v[e] = f < 90 ? acc[(f / 15) >> 1][f % 15][(f / 15) & 1] : 0.f;
                }
                const float tot = MlpEngineRev<WT>::unit_scatter4(v);  // lane group 0, 1, 2, 3: the total of v[0], v[2], v[1], v[3]
                const int f = 4 * q + (w.g == 0 ? 0 : w.g == 1 ? 2 : w.g == 2 ? 1 : 3);
                if (w.live && f < 90) o[(long)(36 + f) * blk] = tot;
            }
            if (s < 3) {  // next primal stage point
                GivenY prov;
#pragma unroll
                for (int k = 0; k < 6; ++k) prov.y[k] = y[k];
                float k1[13];
                state_derivative<float>(P, prov, xs, u, k1);
                const float hs = h * ((s == 2) ? 1.0f : 0.5f);
#pragma unroll
                for (int i = 0; i < 13; ++i) xs[i] = fmaf(hs, k1[i], x0[i]);
            }
        }
    }
    eng.drain();
}

}  // namespace ac

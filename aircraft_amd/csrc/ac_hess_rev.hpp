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
// Two kernels: k_nn_stage_tensors_rev (all six rows in one reverse sweep: 192 slab registers + 90 sums — hipcc's allocator
// leaves 1 200 B of scratch per lane in it; kept as the A/B flavour -DAC_HESS_REV6) and k_nn_stage_tensors_rev3, the product:
// the reverse sweep twice with three rows each (96 slab registers + 45 sums: 208 B of scratch), 20.1 -> 17.55 ms per call.
#pragma once
#include "ac_hess_nn.hpp"

namespace ac {

#ifdef AC_REV_CLOCKS  // (measurement flavour, tools/archive/hess_rev_phases.sh: shader-clock cycles per phase of one wave, printed)
#define AC_REV_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define AC_REV_TICK(i) do { } while (0)
#endif

template <int WT, bool MF = true>  // MF = false: the cross-lane validation form of the 16x16x4 product ("MFMA off")
struct MlpEngineRev : MlpEngine<6, WT, MF, true> {
    typedef MlpEngine<6, WT, MF, true> Base;
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
                    for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<MF>(wcur[i][0], a[s][kt][0], acc[i]);
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
                        for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<MF>(wcur[i][r], a[s][kt][r], acc[i]);
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
    // The 192 slab registers belong in the accumulation registers while a contraction runs (each is read once; the 90 sums are
    // read and written by every packed FMA): left alone, the allocator keeps the slabs — MFMA operands a moment ago — in the
    // VGPRs and wraps every FMA in two v_accvgpr_read and two v_accvgpr_write.
    AC_DI void pin_slabs() {
#pragma unroll
        for (int sl = 0; sl < 6; ++sl)
#pragma unroll
            for (int t = 0; t < WT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+a"(a[sl][t][r]));
    }
    // act''/act' of a tanh neuron with output h, as the second-order epilogue of MlpEngine forms it: -2 h / (1 - h^2),
    // 0 for a saturated neuron (its tangents are 0 as well)
    // (v_rcp_f32, 1 ulp: an IEEE division here is 12 instructions and a branch per neuron)
    AC_DI static float curv_over_slope(float h, float sp) { return sp > 1e-30f ? -2.0f * h * __builtin_amdgcn_rcpf(sp) : 0.f; }

    // Top hidden layer: its state (h, s^1..5) is in the registers; R = the rows of the last layer from the `wlt` image of
    // last_valu ([tile][lane group][6 float4] = {W(2kp, n), W(2kp+1, n), W(2kp, n+1), W(2kp+1, n+1)}, ac_set_mlp), times rs[k]
    // (act' of an activated last layer, else 1).  Leaves a[k] = R[k] . act'(a_top): the input of the first reverse product.
    // (Routed through the scratch like the layers between — state stored, rows loaded into the slabs, contract_mid — the same
    // code came out of the register allocator twice as slow at BOTH call sites: 24.5 ms against 19.9.)
    AC_DI void contract_top(const char* wl_last, const float (&rs)[6], f32x2 (&acc)[3][15]) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl_last + 1024) + g * 6;
        pin_slabs();
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
        pin_slabs();
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
                __builtin_amdgcn_sched_barrier(0);  // one neuron at a time: interleaved, four neurons' operands overflow the file
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
        pin_slabs();
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
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // A layer's 90 per-lane sums, reduce-scattered over the unit's four lanes and added to the lane's running totals: lane group
    // g keeps flat entry 4 q + (0, 2, 1, 3)[g] of quad q (flat = k * 15 + ab).  Only these 23 registers — not the 90 sums —
    // are live across the next reverse product.
    AC_DI static void fold(const f32x2 (&acc)[3][15], float (&tot)[23]) {
#pragma unroll
        for (int q = 0; q < 23; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * q + e;
                v[e] = f < 90 ? acc[(f / 15) >> 1][f % 15][(f / 15) & 1] : 0.f;
            }
            tot[q] += Base::unit_scatter4(v);
        }
    }
    // values that only cross a contraction (the running totals, the stage point): into the 64 accumulation registers the slabs
    // leave free, so the contraction's 90 sums and its operands have the vector file to themselves
    template <int N> AC_DI static void park(float (&v)[N]) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" : "+a"(v[i]));
    }
    AC_DI static void zero(f32x2 (&acc)[3][15]) {
#pragma unroll
        for (int kp = 0; kp < 3; ++kp)
#pragma unroll
            for (int i = 0; i < 15; ++i) acc[kp][i] = f32x2{0.f, 0.f};
    }
    // slabs [0, NS) of the state to the scratch
    template <int NS> AC_DI void store_state(f32x4* __restrict__ sl) const {
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int t = 0; t < WT; ++t) sl[(i * WT + t) * 64 + lane] = f32x4{a[i][t][0], a[i][t][1], a[i][t][2], a[i][t][3]};
    }
};

// ---- the reverse sweep in two halves over the outputs (k = 0..2, then 3..5) ------------------------------------------------
// Same algorithm; the reverse phase holds THREE slabs (96 registers) and 45 sums instead of six and 90, so it fits the
// vector file without scratch traffic.  Price: the transposed blocks stream twice per stage (the ring is sequenced by hand:
// acquire_seq), every layer state is read twice and the top one goes through the scratch as well, and the 15 products
// c s^a s^b of a neuron are formed in both halves.
template <int WT, bool MF = true>
struct MlpEngineRev3 : MlpEngineRev<WT, MF> {
    typedef MlpEngineRev<WT, MF> Rev;
    typedef typename Rev::Base Base;
    typedef typename Rev::f32x2 f32x2;
    using Base::a;
    using Base::g;
    using Base::lane;
    AC_DI MlpEngineRev3(const MlpPlan& pl, const float* blob, char* lds_base) : Rev(pl, blob, lds_base) {}

    // The streamed block the ring holds next (whatever block that is: the caller knows the order), and the copy of block `nl`
    // into the other slot started — MlpEngine::acquire with the successor named instead of read from plan.streamed[].
    AC_DI const char* acquire_seq(int nl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* wl = this->lds + this->plan.ring_off[this->ring_pos & 1];
        lds_dma_copy(this->gblob + this->plan.g_off[nl], this->lds + this->plan.ring_off[(this->ring_pos + 1) & 1], this->plan.bytes[nl],
                     this->wave, this->nwaves, lane);
        ++this->ring_pos;
        return wl;
    }
    // a[s] <- B a[s], slabs 0..2 (layer_raw of the six-slab sweep)
    AC_DI void layer_raw3(const char* wl) {
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        constexpr int C = WT < 4 ? WT : 4;
        f32x4 wcur[C], wnext[C];
#pragma unroll
        for (int i = 0; i < C; ++i) wcur[i] = wf[(i * WT + 0) * 64];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
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
                    const bool fetch = !last || !(s == 2 && nc + C >= WT);
#pragma unroll
                    for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<MF>(wcur[i][0], a[s][kt][0], acc[i]);
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
                        for (int i = 0; i < C; ++i) acc[i] = mma_16x16x4<MF>(wcur[i][r], a[s][kt][r], acc[i]);
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
    // a[j] = row k0 + j of the last layer (k0 = 3 half) times rs3[j], from the `wlt` image of last_valu
    AC_DI void load_top_rows3(const char* wl_last, int half, const float (&rs3)[3]) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl_last + 1024) + g * 6;
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            f32x4 w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = wv[t * 24 + i];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ka = j, kb = 3 + j;  // the two candidates: first and second half
                    const float va = w[2 * (ka >> 1) + (r >> 1)][(ka & 1) + 2 * (r & 1)];
                    const float vb = w[2 * (kb >> 1) + (r >> 1)][(kb & 1) + 2 * (r & 1)];
                    a[j][t][r] = (half ? vb : va) * rs3[j];
                }
        }
    }
    // one neuron: acc[j][q] (pairs over ab = 2q, 2q + 1; entry 15 is padding) += R[j] * c s^a s^b, j = 0..2
    AC_DI static void contract1_3(float c, const float (&s)[5], const float (&R)[3], f32x2 (&acc)[3][8]) {
        float u[5], p[16];
#pragma unroll
        for (int i = 0; i < 5; ++i) u[i] = c * s[i];
        int pi = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i; j < 5; ++j, ++pi) p[pi] = u[i] * s[j];
        p[15] = 0.f;
        const f32x2 r01 = {R[0], R[1]}, r2x = {R[2], R[2]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x2 pp = {p[2 * q], p[2 * q + 1]};
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0][q]) : "v"(r01), "v"(pp));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[1][q]) : "v"(r01), "v"(pp));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[2][q]) : "v"(r2x), "v"(pp));
        }
    }
    // any hidden layer but the first: state from the scratch, R in slabs 0..2; leaves a[j] = R[j] . act'
    AC_DI void contract3_mid(const f32x4* __restrict__ sl, f32x2 (&acc)[3][8]) {
        f32x4 cur[6], nxt[6];
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
                float R[3], s[5];
#pragma unroll
                for (int j = 0; j < 3; ++j) R[j] = a[j][t][r];
                const float h = cur[0][r], sp = fmaf(-h, h, 1.0f);
#pragma unroll
                for (int i = 0; i < 5; ++i) s[i] = cur[1 + i][r];
                contract1_3(Rev::curv_over_slope(h, sp), s, R, acc);
#pragma unroll
                for (int j = 0; j < 3; ++j) a[j][t][r] = R[j] * sp;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) cur[i] = nxt[i];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    AC_DI void contract3_bottom(const f32x4* __restrict__ sl, const char* wl0, f32x2 (&acc)[3][8]) {
        const f32x4* w0t = reinterpret_cast<const f32x4*>(wl0 + 1024);
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            const f32x4 h4 = sl[t * 64 + lane];
            f32x4 w[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) w[j] = w0t[j * (WT * 4) + 4 * t + g];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float R[3], s[5];
#pragma unroll
                for (int j = 0; j < 3; ++j) R[j] = a[j][t][r];
                const float h = h4[r], sp = fmaf(-h, h, 1.0f);
#pragma unroll
                for (int i = 0; i < 5; ++i) s[i] = w[i][r];
                contract1_3(-2.0f * h * sp, s, R, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the layer's 48 sums (flat = 16 j + ab) reduce-scattered over the unit's four lanes into 12 running totals per lane
    AC_DI static void fold3(const f32x2 (&acc)[3][8], float (&tot)[12]) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * q + e;
                v[e] = acc[f / 16][(f % 16) >> 1][f & 1];
            }
            tot[q] += Base::unit_scatter4(v);
        }
    }
};

// float4 of scratch per wave: h_1 + the full state of the hidden layers 2 .. nh + 1 (slot l - 2 for layer l; the six-slab
// kernel keeps the top one in registers and leaves its slot unused)
constexpr long rev_scratch_f32x4(int wt, int nh) { return (long)(1 + 6 * nh) * wt * 64; }

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
    // this wave's scratch: a wave-uniform base (scalar registers), the lane offset added by the accesses themselves
    f32x4* const mine0 = reinterpret_cast<f32x4*>(scratch) +
                         ((long)blockIdx.x * (kBlock >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) *
                             rev_scratch_f32x4(WT, nh);
    const long ntasks = (n + 63) / 64;
#ifdef AC_REV_CLOCKS
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (long task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // lane-dependent values are rebuilt per task, not hoisted into registers the body needs
        const WaveUnit w(tid, task, n, blk);
        float x0[13], u[7];
        load_rows<13>(X, w.ua, x0);
        load_rows<7>(U, w.ua, u);
        const float h = dt_per_unit ? dt_per_unit[w.unit] : dt;
        float xs[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xs[i] = x0[i];
#pragma nounroll
        for (int s = 0; s < 4; ++s) {
            // a copy of the scratch base the optimiser cannot see through: it otherwise hoists the ~150 tile addresses of the
            // stage (and the 126 output addresses below: UnitAddr::late) to kernel entry and spills them
            f32x4* mine = mine0;
            asm volatile("" : "+s"(mine));
            AeroPre<float> ap;
            aero_pre(P, xs, ap);
            const float in[5] = {ap.qbar, ap.alpha, ap.beta, u[0], u[1]};
            float z[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
            // ---- forward: value + five tangents, the states below the top hidden layer to the scratch
            const char* wl0 = eng.acquire(0);
            eng.first_valu(wl0, z);
#ifndef AC_REV_SKIP_STORE
            eng.template store_state<1>(mine);
#endif
#pragma nounroll
            for (int l = 1; l <= nh; ++l) {
                const char* wl = eng.acquire(l);
#ifndef AC_REV_SKIP_FWD
                eng.template layer<WT, WT, 1>(wl, 1);
#endif
#ifndef AC_REV_SKIP_STORE
                if (l < nh) eng.template store_state<6>(mine + WT * 64 + (long)(l - 1) * 6 * WT * 64);
#endif
            }
            AC_REV_TICK(0);  // [0] stage point, first layer, hidden products, state stores
            float y[6], J[6][5];
            const char* wll = eng.acquire(L - 1);
            const int act_last = plan.act[L - 1];
            eng.template last_valu<5>(wll, act_last, y, J);
            // ---- outputs y and J from lane group 0 (J is dead afterwards)
            if (w.live && w.g == 0) {
                float* o = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    o[(long)k * blk] = y[k];
#pragma unroll
                    for (int i = 0; i < 5; ++i) o[(long)(6 + k * 5 + i) * blk] = J[k][i];
                }
            }
            AC_REV_TICK(1);  // [1] last layer, y and J out
            // ---- reverse sweep with the contraction at every hidden layer
            float rs[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) rs[k] = act_last ? fmaf(-y[k], y[k], 1.0f) : 1.0f;
            typedef MlpEngineRev<WT> E;
            float tot[23];  // this lane's share of the 90 entries of T (E::fold)
#pragma unroll
            for (int q = 0; q < 23; ++q) tot[q] = 0.f;
            {
                typename E::f32x2 acc[3][15];
                // y = tanh(p) on the last layer: + act''(p_k) dp/dz_a dp/dz_b = -2 y_k J_a J_b / act'(p_k), once per unit (lane
                // group 0; the four lanes' sums are added by fold); zero otherwise
                const bool fix = act_last && w.g == 0;
                int pi = 0;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = i; j < 5; ++j, ++pi)
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            acc[k >> 1][pi][k & 1] = fix ? E::curv_over_slope(y[k], rs[k]) * J[k][i] * J[k][j] : 0.f;
                E::park(tot);
                E::park(xs);
#ifndef AC_REV_SKIP_CONTRACT  // (timing experiments only: tools/archive/hess_rev_phases.sh)
                eng.contract_top(wll, rs, acc);
#endif
                AC_REV_TICK(2);  // [2] top contraction
                E::fold(acc, tot);
                AC_REV_TICK(6);  // [6] folds
            }
#pragma nounroll
            for (int l = nh; l >= 1; --l) {
#ifndef AC_REV_SKIP_RAW
                eng.layer_raw(eng.acquire(L + nh - l));
#else
                (void)eng.acquire(L + nh - l);
#endif
                AC_REV_TICK(3);  // [3] reverse products
                typename E::f32x2 acc[3][15];
                E::zero(acc);
                E::park(tot);
                E::park(xs);
#ifndef AC_REV_SKIP_CONTRACT
                if (l > 1) eng.contract_mid(mine + WT * 64 + (long)(l - 2) * 6 * WT * 64, acc);
                else eng.contract_bottom(mine, wl0, acc);
#endif
                if (l > 1) AC_REV_TICK(4); else AC_REV_TICK(5);  // [4] contractions between, [5] bottom contraction
                E::fold(acc, tot);
                AC_REV_TICK(6);
            }
            // ---- T: every lane stores its 23 (lane group 0 and 1: 22 or 23) entries
            if (w.live) {
                float* o = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                for (int q = 0; q < 23; ++q) {
                    const int f = 4 * q + (w.g == 0 ? 0 : w.g == 1 ? 2 : w.g == 2 ? 1 : 3);
                    if (f < 90) o[(long)(36 + f) * blk] = tot[q];
                }
            }
            if (s < 3) {  // next primal stage point.  x0, u and this stage's y are read again (y from the rows lane group 0 of this
                          // very wave stored above): 32 registers fewer across the products, where the vector file is full
                load_rows<13>(X, w.ua, x0);
                load_rows<7>(U, w.ua, u);
                GivenY prov;
                const float* yo = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                for (int k = 0; k < 6; ++k) prov.y[k] = yo[(long)k * blk];
                float k1[13];
                state_derivative<float>(P, prov, xs, u, k1);
                const float hs = h * ((s == 2) ? 1.0f : 0.5f);
#pragma unroll
                for (int i = 0; i < 13; ++i) xs[i] = fmaf(hs, k1[i], x0[i]);
            }
            AC_REV_TICK(7);  // [7] T out, next stage point
        }
    }
    eng.drain();
#ifdef AC_REV_CLOCKS
    if ((threadIdx.x & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x < 128)
        printf("rev clocks block %d wave %d: fwd %llu last %llu top %llu raw %llu mid %llu bottom %llu fold %llu rest %llu\n", (int)blockIdx.x,
               (int)(threadIdx.x >> 6), clk[0], clk[1], clk[2], clk[3], clk[4], clk[5], clk[6], clk[7]);
#endif
}

// The same stage tensors with the reverse sweep in two halves (MlpEngineRev3).  Same arguments, same scratch layout.
template <int WT, bool MF>
__global__ __launch_bounds__(kBlock, 1) void k_nn_stage_tensors_rev3(const DevParams P, const MlpPlan plan,
                                                                     const float* __restrict__ blob,
                                                                     const float* __restrict__ X, const float* __restrict__ U,
                                                                     float dt, const float* __restrict__ dt_per_unit, long n,
                                                                     long blk, int L, float* __restrict__ scratch,
                                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineRev3<WT, MF> E;
    E eng(plan, blob, smem);
    eng.load_weights();  // resident blocks + block 1 in ring slot 0
    const int nh = L - 2;
    f32x4* const mine0 = reinterpret_cast<f32x4*>(scratch) +
                         ((long)blockIdx.x * (kBlock >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) *
                             rev_scratch_f32x4(WT, nh);
    const long ntasks = (n + 63) / 64;
#ifdef AC_REV_CLOCKS
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (long task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const WaveUnit w(tid, task, n, blk);
        float x0[13], u[7];
        load_rows<13>(X, w.ua, x0);
        load_rows<7>(U, w.ua, u);
        const float h = dt_per_unit ? dt_per_unit[w.unit] : dt;
        float xs[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xs[i] = x0[i];
#pragma nounroll
        for (int s = 0; s < 4; ++s) {
            f32x4* mine = mine0;
            asm volatile("" : "+s"(mine));
            AeroPre<float> ap;
            aero_pre(P, xs, ap);
            const float in[5] = {ap.qbar, ap.alpha, ap.beta, u[0], u[1]};
            float z[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
            // ---- forward: the state after EVERY hidden layer to the scratch
            const char* wl0 = eng.Base::acquire(0);
            eng.first_valu(wl0, z);
            eng.template store_state<1>(mine);
#pragma nounroll
            for (int l = 1; l <= nh; ++l) {
                const char* wl = eng.acquire_seq(l < nh ? l + 1 : L);
                eng.template layer<WT, WT, 1>(wl, 1);
                eng.template store_state<6>(mine + WT * 64 + (long)(l - 1) * 6 * WT * 64);
            }
            AC_REV_TICK(0);
            float y[6];
            const char* wll = eng.Base::acquire(L - 1);
            const int act_last = plan.act[L - 1];
            {
                float J[6][5];
                eng.template last_valu<5>(wll, act_last, y, J);
                if (w.live && w.g == 0) {
                    float* o = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        o[(long)k * blk] = y[k];
#pragma unroll
                        for (int i = 0; i < 5; ++i) o[(long)(6 + k * 5 + i) * blk] = J[k][i];
                    }
                }
            }
            AC_REV_TICK(1);
            // ---- reverse sweep, outputs 3 half .. 3 half + 2
            float tot[2][12];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int q = 0; q < 12; ++q) tot[hf][q] = 0.f;
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
                float rs3[3], y3[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    y3[j] = half ? y[3 + j] : y[j];
                    rs3[j] = act_last ? fmaf(-y3[j], y3[j], 1.0f) : 1.0f;
                }
                eng.load_top_rows3(wll, half, rs3);
                float th[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) th[q] = 0.f;
                AC_REV_TICK(6);  // (rows of the last layer into the slabs: with the folds)
                // hidden layers nh + 1 (the top one: R = the rows just loaded) .. 1; ONE instance of the contraction code for all of
                // them (a separate call for the top layer came out of the register allocator 2.3 times slower than the loop's)
#pragma nounroll
                for (int l = nh + 1; l >= 1; --l) {
                    if (l <= nh) {
                        const int i = nh - l;  // transposed block L + i; its successor: the next one, or L again / block 1 of the next stage
                        eng.layer_raw3(eng.acquire_seq(i + 1 < nh ? L + i + 1 : (half == 0 ? L : 1)));
                    }
                    AC_REV_TICK(3);
                    typename E::f32x2 acc[3][8];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[j][q] = typename E::f32x2{0.f, 0.f};
                    if (l > 1) {
                        eng.contract3_mid(mine + WT * 64 + (long)(l - 2) * 6 * WT * 64, acc);
                    } else {
                        eng.contract3_bottom(mine, wl0, acc);
                        if (act_last && w.g == 0) {  // y = tanh(p): + act''(p_k) dp/dz_a dp/dz_b = -2 y_k J_a J_b / act'(p_k), once per unit; J from
                                                     // the rows lane group 0 of this very wave stored above
                            const float* o = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                float Jr[5];
#pragma unroll
                                for (int a5 = 0; a5 < 5; ++a5) Jr[a5] = o[(long)(6 + ((half ? 3 : 0) + j) * 5 + a5) * blk];
                                const float c2 = E::Rev::curv_over_slope(y3[j], rs3[j]);
                                int pi = 0;
#pragma unroll
                                for (int a5 = 0; a5 < 5; ++a5)
#pragma unroll
                                    for (int b5 = a5; b5 < 5; ++b5, ++pi) acc[j][pi >> 1][pi & 1] += c2 * Jr[a5] * Jr[b5];
                            }
                        }
                    }
                    if (l > nh) AC_REV_TICK(2); else if (l > 1) AC_REV_TICK(4); else AC_REV_TICK(5);
                    E::fold3(acc, th);
                    AC_REV_TICK(6);
                }
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    if (half == 0) tot[0][q] = th[q]; else tot[1][q] = th[q];
                }
            }
            // ---- T: lane group g holds entry 4 q + (0, 2, 1, 3)[g] of each half's 48 (flat = 16 j + ab, ab = 15 padding)
            if (w.live) {
                float* o = out + w.ua.late().off(kStageFloats) + (long)s * kStageRows * blk;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int q = 0; q < 12; ++q) {
                        const int f = 4 * q + (w.g == 0 ? 0 : w.g == 1 ? 2 : w.g == 2 ? 1 : 3);
                        const int ab = f & 15, k = 3 * hf + (f >> 4);
                        if (ab < 15) o[(long)(36 + k * 15 + ab) * blk] = tot[hf][q];
                    }
            }
            if (s < 3) {
                load_rows<13>(X, w.ua, x0);
                load_rows<7>(U, w.ua, u);
                GivenY prov;
#pragma unroll
                for (int k = 0; k < 6; ++k) prov.y[k] = y[k];
                float k1[13];
                state_derivative<float>(P, prov, xs, u, k1);
                const float hs = h * ((s == 2) ? 1.0f : 0.5f);
#pragma unroll
                for (int i = 0; i < 13; ++i) xs[i] = fmaf(hs, k1[i], x0[i]);
            }
            AC_REV_TICK(7);
        }
    }
    eng.drain();
#ifdef AC_REV_CLOCKS
    if ((threadIdx.x & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x < 128)
        printf("rev3 clocks block %d wave %d: fwd %llu last %llu top %llu raw %llu mid %llu bottom %llu fold %llu rest %llu\n", (int)blockIdx.x,
               (int)(threadIdx.x >> 6), clk[0], clk[1], clk[2], clk[3], clk[4], clk[5], clk[6], clk[7]);
#endif
}

}  // namespace ac

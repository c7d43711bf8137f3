// ac_mlp.hpp — wave-level MLP surrogate engine for gfx950 (the NeuralModel of
// dynamics/coefficient_models.py:91-104 / ScaledModel of surrogates/models.py:101-155).
//
// Data layout (one wave = 16 units, 4 lanes per unit: col = lane & 15 is the unit, g = lane >> 4):
//   * An activation "slab" is a [width][16 units] fp32 matrix held ENTIRELY IN REGISTERS in the
//     C/D layout of v_mfma_f32_16x16x4_f32: register a[s][t][r] of lane (col, g) is row 16 t + 4 g + r,
//     column `col`.  A D tile in that layout is *directly* the B operand of the next layer's MFMA
//     (k-step r of k-tile t multiplies rows {16t+r, 16t+4+r, 16t+8+r, 16t+12+r}), so activations never
//     leave the register file between layers — no LDS round trip, no shuffles.
//   * The matching A operand for (out-tile nt, k-tile kt, k-step r) is W[16 nt + col][16 kt + 4 g + r]:
//     the host packs each layer in "fragment order" [nt][kt][lane][4] so that one conflict-free
//     ds_read_b128 at base + 16*lane fetches the A operands of all four k-steps.
//   * Slab 0 carries values.  In tangent mode slabs 1..5 carry d/d(input j) — forward-mode through
//     the net: t_out = act'(h_out) * (W t_in).  The 6 x 128 x 16 fp32 activations of a 4x128 net are
//     192 registers per lane; the output tile of the slab being computed adds 32.
//   * Weights: layers that fit stay resident in LDS; layers that do not (3 x 64 KB for 4x128) stream
//     through a two-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4) one layer ahead, so
//     the L2 -> LDS copy of layer l+1 overlaps the MFMAs of layer l.  One workgroup barrier per
//     streamed layer.
#pragma once
#include "ac_dynamics.hpp"

namespace ac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpPlan {
    int n_layers;
    int KT[AC_MAX_LAYERS];       // ceil(n_in / 16)
    int NT[AC_MAX_LAYERS];       // ceil(n_out / 16)
    int act[AC_MAX_LAYERS];      // 0 identity, 1 tanh; ac_set_mlp guarantees 1 on every layer but the last
    int g_off[AC_MAX_LAYERS];    // float offset of the packed layer block in the global blob
    int bytes[AC_MAX_LAYERS];    // block size: NT*KT*1024 (weights) + 1024 (bias piece)
    int lds_off[AC_MAX_LAYERS];  // byte offset if resident, -1 if streamed through the ring
    int ring_off[2];             // byte offsets of the two ring slots
    int n_streamed;              // number of streamed layers per forward pass
    int first_streamed;          // index of the first streamed layer (-1 if none)
    int streamed[AC_MAX_LAYERS]; // layer index of the i-th streamed layer, i < n_streamed
    int lds_total;               // dynamic LDS bytes to request
};

// ---- 16x16x4 fp32 matrix-multiply-accumulate on one wave -------------------------------------
// "MFMA off" validation path: the same contraction with cross-lane reads on the VALU.
// A[i][k] lives on lane i + 16k, B[k][j] on lane j + 16k; this lane owns D[4g + r][col].
// k ascending with one fmaf per product = the MFMA's documented k-ordered fmaf chain.
// Deliberately NOT inlined: it is a baseline, and one out-of-line copy keeps the build fast.
__device__ __attribute__((noinline)) inline f32x4 mma_16x16x4_valu(float a, float b, f32x4 c) {
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float bk = __shfl(b, col + 16 * k, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = fmaf(__shfl(a, 4 * g + r + 16 * k, 64), bk, c[r]);
    }
    return c;
}

template <bool USE_MFMA> AC_DI f32x4 mma_16x16x4(float a, float b, f32x4 c) {
    if constexpr (USE_MFMA) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    else return mma_16x16x4_valu(a, b, c);
}

// Branch-free tanh(x) = 1 - 2 / (1 + e^{2x}): one v_exp_f32 and one v_rcp_f32, so it can sit between MFMAs
// (ocml tanhf branches on |x|).  Saturates correctly (e^{2x} -> inf gives 1, -> 0 gives -1), NaN propagates;
// absolute error <= ~1.5e-7 (the 1 - 2/(...) cancellation near 0 costs relative, not absolute, accuracy).
AC_DI float act_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2^(2x log2 e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

// Copy `bytes` (a multiple of 1024: the host pads every layer block to whole pieces) from global to
// LDS with LDS-DMA.  Every wave of the workgroup takes 1-KiB pieces round-robin; a piece is one
// global_load_lds_dwordx4 wave-instruction (LDS destination = piece base + 16 * lane, all lanes active).
AC_DI void lds_dma_copy(const float* __restrict__ gsrc, char* lds_dst, int bytes, int wave, int nwaves,
                        int lane) {
    // the piece loop is wave-uniform: run it on the scalar unit (uniform base + one VGPR offset per lane) instead of as an
    // exec-masked vector loop with a 64-bit vector address and a readfirstlane for M0 per piece
    const int pieces = __builtin_amdgcn_readfirstlane(bytes >> 10);
    const int w0 = __builtin_amdgcn_readfirstlane(wave), step = __builtin_amdgcn_readfirstlane(nwaves);
    const unsigned voff = (unsigned)lane * 16u;
    for (int p = w0; p < pieces; p += step) {
        const char* base = (const char*)gsrc + ((long)p << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                         (__attribute__((address_space(3))) void*)(lds_dst + (p << 10)), 16, 0, 0);
    }
}

// NSLAB slabs of 16 columns each.  TANGENT: slab 0 carries values and slabs 1..5 the five input tangents of the SAME
// 16 units (NSLAB = 6).  Otherwise every slab is a value slab: NSLAB = 1 (16 units, the four lanes of a unit redundant)
// or NSLAB = 4 (64 units, lane = unit; slab s = units 16 s .. 16 s + 15).  WT: register tiles per slab = width / 16.
// SECOND (second-order mode): slabs = value, the K first and the K (K + 1) / 2 second derivatives for one input triple
// (p, q, r) set with set_triple() (NSLAB == 10) or for all five inputs (NSLAB == 21, widths <= 64); used by the Hessian
// path (ac_hess_nn.hpp), never by the step kernels.
// TOFF (tangent mode with fewer slabs): tangent slab s carries input TOFF + s - 1 — a wave pair splits the five
// tangents as value + {0, 1, 2} and value + {3, 4} (k_nn_step_sens_pair).
// The input triple of the second-order mode lives in a base class that is EMPTY for every other engine: a member the
// step kernels never read still moved hipcc's register allocation of k_nn_step_sens (scratch 228 -> 264 B, -1.2 %).
template <bool ON> struct TripleHolder {
    int tri_p = 0, tri_q = 1, tri_r = 2, tri_t = 3;
    AC_DI void set_triple(int p, int q, int r) { tri_p = p; tri_q = q; tri_r = r; }
    AC_DI void set_quad(int p, int q, int r, int t) { tri_p = p; tri_q = q; tri_r = r; tri_t = t; }
    AC_DI int tri(int i) const { return i == 0 ? tri_p : i == 1 ? tri_q : i == 2 ? tri_r : tri_t; }
};
template <> struct TripleHolder<false> {
    AC_DI int tri(int) const { return 0; }
};

// PAIR (k_nn_step_sens_pair: two waves share the six slabs of one unit group, three each): 1 = the wave with the value slab and
// tangents TOFF, TOFF + 1 — it publishes every hidden layer's value activations h through LDS (`hx`); 2 = the wave with three
// tangent slabs and NO value slab — its hidden-layer outputs stay unscaled until it has read h behind the next barrier
// (scale_from_hx).  Both roles meet at the same workgroup barriers: one at the head of every hidden layer, one after the second
// slab of it (between the reader's load of h and the writer's next store), one before the last layer.
template <int NSLAB, int WT, bool USE_MFMA, bool TANGENT = (NSLAB == 6), bool SECOND = false, int TOFF = 0, int PAIR = 0>
struct MlpEngine : TripleHolder<SECOND && NSLAB <= 10> {
    using TripleHolder<SECOND && NSLAB <= 10>::tri;
    static constexpr bool kNoValue = PAIR == 2;  // every slab is a tangent slab
    static constexpr int kFirstTangent = kNoValue ? 0 : 1;  // slab index of tangent TOFF
    static_assert(PAIR == 0 || (TANGENT && USE_MFMA && NSLAB == 3), "wave-pair roles: three slabs each, matrix-core flavour");
    static_assert(!TANGENT || (NSLAB >= 2 && NSLAB <= 6 && TOFF + NSLAB - kFirstTangent <= 5), "tangent mode = value + a range of the 5 input tangents");
    static constexpr int kTangents = TANGENT ? NSLAB - kFirstTangent : 0;
    // second-order mode over K inputs (a triple (p, q, r) set with set_triple(), NSLAB = 10; or all five, NSLAB = 21):
    // slab 0 value; 1..K d/dz_i; K+1..2K d2/dz_i2; then d2/dz_i dz_j for the pairs i < j in lexicographic order
    // (K = 3: 7-9 = pq, pr, qr).  NSLAB = 1 + K + K (K + 1) / 2.
    // Bipartite layout (NSLAB = 9, inputs (p, q | r, t) set with set_quad()): slab 0 value; 1-4 d/dz_p, q, r, t; 5-8 the four
    // cross derivatives pr, pt, qr, qt — the pairs between two input groups whose own pairs other passes cover.
    static_assert(!SECOND || ((NSLAB == 9 || NSLAB == 10 || NSLAB == 21) && !TANGENT), "second-order mode: 9 (bipartite), 10 (triple) or 21 (all five inputs) slabs");
    static constexpr int kFirstOrder = !SECOND ? 0 : (NSLAB == 10 ? 3 : NSLAB == 9 ? 4 : 5);  // slabs 1..K are first-order
    // first-order slabs (1-based input positions) whose product the second-order slab s differentiates
    AC_DI static constexpr int second_a(int s) {
        constexpr int K = kFirstOrder;
        if (NSLAB == 9) return s < 7 ? 1 : 2;
        if (s <= 2 * K) return s - K;
        int m = s - 2 * K - 1;
        for (int i = 1; i < K; ++i) { if (m < K - i) return i; m -= K - i; }
        return K;
    }
    AC_DI static constexpr int second_b(int s) {
        constexpr int K = kFirstOrder;
        if (NSLAB == 9) return (s & 1) ? 3 : 4;
        if (s <= 2 * K) return s - K;
        int m = s - 2 * K - 1;
        for (int i = 1; i < K; ++i) { if (m < K - i) return i + 1 + m; m -= K - i; }
        return K;
    }
    AC_DI int input_of(int i) const { if constexpr (NSLAB <= 10) return tri(i); else return i; }
    static constexpr bool kTangent = TANGENT;
    static constexpr int kWT = WT;
    static constexpr bool kDeriv = TANGENT || SECOND;  // slabs > 0 are derivative slabs (no bias, chain-rule epilogue)
    // The first and last layers (5 -> width, width -> 6) of the matrix-core engines run on the vector ALUs (first_valu / last_valu
    // below): as MFMA tiles they are mostly padding.  Such engines take the handle's `plan_sens`, whose edge blocks hold
    // [bias][W0 transposed] and [bias][wlt] only.  (The second-order engines keep the matrix form and `plan`.)
    static constexpr bool kVLast = USE_MFMA && !SECOND;  // (the value-only engines of the forward kernels as well: *_values below)
#ifndef AC_CH
#define AC_CH 4  // the two headline units are built with 2 (build.py UNIT_FLAGS): 84 % less spill, DESIGN.md §6
#endif
    static constexpr int CH = WT < AC_CH ? WT : AC_CH;  // output tiles computed together (independent accumulators)

    float a[NSLAB][WT][4];
    const MlpPlan& plan;
    const float* __restrict__ gblob;
    char* lds;
    int lane, g, wave, nwaves;
    int ring_pos;  // number of streamed layers consumed so far (slot = ring_pos & 1)
    int snext;     // position in plan.streamed[] of the next layer to fetch (wraps: the sequence is cyclic)
    Stamper st;    // diagnostic flavor only (empty otherwise)
    f32x4* hx = nullptr;  // PAIR: this pair's [WT][64 lanes] float4 exchange of the value activations
    // Spread mode (the six-slab sensitivity engine at width 128 with two-tile chunks: 24 chunks per hidden layer): the next
    // streamed layer's LDS-DMA is issued one 1-KiB piece per chunk of the current layer's matrix stream instead of as a burst
    // of 17 pieces behind the barrier — the burst delayed the first blocks' ds_reads (+0.4 % on the headline; with four-tile
    // chunks the same change cost 500-1000 B of scratch per lane).  All three fields are wave-uniform.
    static constexpr bool kSpread = TANGENT && NSLAB * (WT / CH) * 4 >= WT * WT + 1;
    const char* dma_src;
    unsigned dma_dst;
    int dma_last;  // last piece of the pending copy.  Until acquire() meets a streamed layer (all-resident nets: never) the
                   // pending copy is piece 0 of the resident first layer ONTO ITSELF: same bytes, so the issue points need no
                   // branch (a wave-uniform test around them costs 370 B of scratch per lane in the headline kernel)

    AC_DI MlpEngine(const MlpPlan& pl, const float* blob, char* lds_base)
        : plan(pl), gblob(blob), lds(lds_base), ring_pos(0), snext(pl.n_streamed > 1 ? 1 : 0),
          dma_src((const char*)(blob + pl.g_off[0])), dma_dst((unsigned)pl.lds_off[0]), dma_last(0) {
        lane = threadIdx.x & 63; g = lane >> 4; wave = threadIdx.x >> 6; nwaves = blockDim.x >> 6;
    }

    AC_DI int streamed_layer(int idx) const {  // idx-th streamed layer of the cyclic sequence
        int seen = 0;
        for (int l = 0; l < plan.n_layers; ++l)
            if (plan.lds_off[l] < 0) { if (seen == idx) return l; ++seen; }
        return -1;
    }

    // Prologue: resident layers + the first streamed layer into ring slot 0.
    AC_DI void load_weights() {
        for (int l = 0; l < plan.n_layers; ++l)
            if (plan.lds_off[l] >= 0) lds_dma_copy(gblob + plan.g_off[l], lds + plan.lds_off[l], plan.bytes[l], wave, nwaves, lane);
        if (plan.n_streamed > 0) {
            const int l = plan.first_streamed;
            lds_dma_copy(gblob + plan.g_off[l], lds + plan.ring_off[0], plan.bytes[l], wave, nwaves, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // drains vmcnt (LDS-DMA) and makes the image visible to every wave
    }

    // Must run before the wave exits: an LDS-DMA prefetch may still be in flight.
    AC_DI void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // Epilogue of one output tile: activation on the value slab, act'(h) scaling on a tangent slab.
    // ACT: 1 = tanh, 0 = identity, -1 = decided by the runtime flag (a select per element; the hidden layers, where the
    // epilogue is exposed VALU time, are dispatched on the flag once per layer instead)
    template <int NT, int ACT = -1>
    AC_DI void epilogue_tile(int s, int nt, const f32x4 (&o)[NT], int act_flag) {
        const bool act = ACT < 0 ? act_flag != 0 : ACT != 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (kNoValue) {
                a[s][nt][r] = o[nt][r];  // W t; act'(h) follows in scale_from_hx() once the partner wave has published h
            } else if (s == 0 || !kDeriv) {
                a[s][nt][r] = act ? act_tanh(o[nt][r]) : o[nt][r];
            } else if (!SECOND || s <= kFirstOrder) {
                const float h = a[0][nt][r];  // already the NEW value activation
                a[s][nt][r] = act ? o[nt][r] * fmaf(-h, h, 1.0f) : o[nt][r];
            } else {
                // h_ab = s'(z) z_ab + s''(z) z_a z_b with s'' = -2 h s' and z_a = h_a / s'  (h_a, h_b already NEW)
                const float h = a[0][nt][r], sp = fmaf(-h, h, 1.0f);
                const float ha = a[second_a(s)][nt][r], hb = a[second_b(s)][nt][r];
                const float inv = sp > 1e-30f ? 1.0f / sp : 0.f;  // saturated neuron: both terms vanish
                a[s][nt][r] = act ? fmaf(o[nt][r], sp, -2.0f * h * ha * hb * inv) : o[nt][r];
            }
        }
    }

    // CNT output tiles (independent accumulators) x KT k-tiles for slab s; straight-line code, the next
    // block's A fragments are fetched (ds_read_b128) while the current one's MFMAs issue.  The epilogue of
    // the PREVIOUS slab (VALU / transcendental work, independent of this slab's MFMAs) runs after the first block.
    template <int CNT, int KT, int NT, int ACT>
    AC_DI void gemm_chunk(const f32x4* __restrict__ wf, const f32x4* __restrict__ bias4, int s, int nc, f32x4 (&o)[NT],
                          const float (&in)[WT][4], const f32x4 (&oprev)[NT], int act, f32x4 (&wcur)[CNT],
                          f32x4 (&bcur)[CNT], bool prefetch_next_chunk) {
        // The chunks of a layer are mutually independent; without a fence the machine scheduler interleaves
        // them across the whole straight-line layer and the live accumulators no longer fit the register file.
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kSpread && KT == WT && NT == WT) {
            const int ord = s * (NT / CNT) + nc / CNT;  // compile-time ordinal of this chunk within the layer
            if (ord * 4 < WT * WT + 1 + 3) {
                int pc = __builtin_amdgcn_readfirstlane(wave) + 4 * ord;
                pc = pc < dma_last ? pc : dma_last;  // past the end: repeat the last piece (same bytes, harmless)
                const char* base = dma_src + ((long)pc << 10);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)lane * 16u),
                                                 (__attribute__((address_space(3))) void*)(lds + dma_dst + (pc << 10)), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 acc[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            if ((s == 0 && !kNoValue) || !kDeriv) acc[i] = bcur[i];  // a value slab starts from the bias (fetched during the previous chunk)
            else acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 wnext[CNT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = (kt + 1 == KT);
            const bool fetch = !last || prefetch_next_chunk;
            // k-step 0 first: its operands force the wait for THIS block's fragments while no newer LDS read is
            // outstanding (LDS returns in order; issued the other way round hipcc waits lgkmcnt(0) on the brand-new
            // reads as well, one exposed LDS latency per block with the matrix pipe idle).
#pragma unroll
            for (int i = 0; i < CNT; ++i) acc[i] = mma_16x16x4<USE_MFMA>(wcur[i][0], in[kt][0], acc[i]);
            if (fetch) {
                // then fetch the NEXT block's A fragments (next k-tile, or k-tile 0 of the next chunk — they depend on
                // the output tiles only, not on the slab); 12 MFMAs (384 cycles) cover their latency.
                __builtin_amdgcn_sched_barrier(0);
                const int nnc = last ? (nc + CNT) % NT : nc;
                const int nkt = last ? 0 : kt + 1;
#pragma unroll
                for (int i = 0; i < CNT; ++i) wnext[i] = wf[((nnc + i) * KT + nkt) * 64];
                // ... and, behind the last block, the bias the next chunk's accumulators start from (read at the head of the
                // chunk it was one exposed LDS latency per chunk of every value slab: 4 x ~130 cycles per hidden layer)
                if (last && !kNoValue && (kDeriv ? (s == 0 && nc + CNT < NT) : (nc + CNT < NT || s + 1 < NSLAB))) {
#pragma unroll
                    for (int i = 0; i < CNT; ++i) bcur[i] = bias4[(nnc + i) * 4 + g];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 1; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < CNT; ++i) acc[i] = mma_16x16x4<USE_MFMA>(wcur[i][r], in[kt][r], acc[i]);
            if (s > 0) {
                // The previous slab's whole epilogue, as ONE uninterrupted run of VALU work after this slab's first
                // block.  With one wave per SIMD nothing issues in the shadow of the matrix pipe: every switch
                // MFMA -> VALU -> MFMA idles it (profiles/r01_micro_mfma_valu_overlap.txt: one v_fma between two MFMAs
                // costs 15.5 cycles, eight in one run 72), so the epilogue is kept in one piece rather than spread
                // over the k-tile blocks (measured: 1 tile per 2 blocks 4.145 ms, 2 tiles 4.125, 4 tiles 4.12, all 8
                // 4.107; later blocks than the first are slower too).  Deferring it to the next slab still pays: the
                // accumulators it reads are long complete.
                // (The fences stay in every block of the slab: without them the scheduler reorders the later blocks
                // and the kernel spills 80 B/lane more.)
                __builtin_amdgcn_sched_barrier(0);
                if (nc == 0 && kt == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) epilogue_tile<NT, ACT>(s - 1, t, oprev, act);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (fetch) {
#pragma unroll
                for (int i = 0; i < CNT; ++i) wcur[i] = wnext[i];
            }
        }
#pragma unroll
        for (int i = 0; i < CNT; ++i) o[nc + i] = acc[i];
        __builtin_amdgcn_sched_barrier(0);
    }

    // One Linear(+tanh) layer of static shape KT x NT tiles on all slabs.  wl: LDS address of the packed block.
    // The host pads every hidden width to 16*WT, so only the shapes <1,WT> (first), <WT,WT> (hidden),
    // <WT,1> (last) and <1,1> (single-layer net) occur.
#ifdef AC_EXP_LAST2
    // EXPERIMENT (not in the product build): the last layer (KT = WT, NT = 1) with two slabs' accumulator chains
    // interleaved on shared A fragments — the shape of the round-1 experiment behind DESIGN §9.4.
    template <int KT, int ACT>
    AC_DI void layer_last2(const char* wl, int act) {
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl + KT * 1024);
        constexpr int NPAIR = NSLAB / 2;
#pragma unroll
        for (int sp = 0; sp < NPAIR; ++sp) {
            const int s0 = 2 * sp, s1 = 2 * sp + 1;
            f32x4 acc0 = (s0 == 0 || !kDeriv) ? bias4[g] : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acc1 = (!kDeriv) ? bias4[g] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const f32x4 w = wf[kt * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0 = mma_16x16x4<USE_MFMA>(w[r], a[s0][kt][r], acc0);
                    acc1 = mma_16x16x4<USE_MFMA>(w[r], a[s1][kt][r], acc1);
                }
            }
            f32x4 o0[1] = {acc0}, o1[1] = {acc1};
            epilogue_tile<1, ACT>(s0, 0, o0, act);
            epilogue_tile<1, ACT>(s1, 0, o1, act);
        }
        if constexpr (NSLAB % 2 == 1) {
            constexpr int s = NSLAB - 1;
            f32x4 acc = (s == 0 || !kDeriv) ? bias4[g] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const f32x4 w = wf[kt * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mma_16x16x4<USE_MFMA>(w[r], a[s][kt][r], acc);
            }
            f32x4 o0[1] = {acc};
            epilogue_tile<1, ACT>(s, 0, o0, act);
        }
    }
#endif

    template <int KT, int NT, int ACT = -1>
    AC_DI void layer(const char* wl, int act) {
#ifdef AC_EXP_LAST2
        if constexpr (NT == 1 && KT == WT && KT > 1) { layer_last2<KT, ACT>(wl, act); return; }
#endif
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl + NT * KT * 1024);
        constexpr int C = NT < CH ? NT : CH;
        static_assert(NT % C == 0, "tile count must be a multiple of the chunk");
        f32x4 o[2][NT];  // ping-pong: slab s accumulates into o[s&1] while slab s-1's epilogue drains o[(s-1)&1]
        f32x4 wcur[C];   // A fragments of the block about to run; carried across chunks
#pragma unroll
        for (int i = 0; i < C; ++i) wcur[i] = wf[(i * KT + 0) * 64];
        f32x4 bcur[C];   // bias of the value-slab chunk about to run
#pragma unroll
        for (int i = 0; i < C; ++i) bcur[i] = bias4[i * 4 + g];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
#pragma unroll
            for (int nc = 0; nc < NT; nc += C) {
                const bool more = !(s == NSLAB - 1 && nc + C >= NT);
                gemm_chunk<C, KT, NT, ACT>(wf, bias4, s, nc, o[s & 1], a[s], o[(s + 1) & 1], act, wcur, bcur, more);
            }
            if (KT == WT && NT == WT) { if (s == 0) AC_MARK(st, 9); else if (s == 1) AC_MARK(st, 10); else AC_MARK(st, 11); }
            if constexpr (PAIR != 0 && KT == WT && NT == WT) {
                if (s == 1) {
                    // the partner has read the previous layer's h by now (it did so before its first slab); this layer's
                    // value epilogue ran inside slab 1's first block, so a[0] is h of THIS layer
                    __syncthreads();
                    if constexpr (PAIR == 1) {
#pragma unroll
                        for (int nt = 0; nt < WT; ++nt) hx[nt * 64 + lane] = f32x4{a[0][nt][0], a[0][nt][1], a[0][nt][2], a[0][nt][3]};
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) epilogue_tile<NT, ACT>(NSLAB - 1, nt, o[(NSLAB - 1) & 1], act);
    }

    // First layer (5 -> width), tangent-aware: the value slab runs on the MFMA (one padded k-tile); the tangent
    // slabs of the first layer are just columns of W0 scaled by act'(h) — W0[n][j] (1 - h_n^2) — so they are read
    // from a transposed copy of W0 the host appends to the block, with no MFMA at all (saves 5/6 of this layer's
    // matrix work, 3 % of a stage).
    // PAIR == 2: act'(h) of the hidden layer just finished, h from the partner wave (published before the barrier the caller
    // has just passed), on this wave's three unscaled slabs — the same product o * (1 - h^2) the one-wave engine forms.
    AC_DI void scale_from_hx() {
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) {
            const f32x4 h = hx[nt * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sp = fmaf(-h[r], h[r], 1.0f);
#pragma unroll
                for (int sl = 0; sl < NSLAB; ++sl) a[sl][nt][r] = a[sl][nt][r] * sp;
            }
        }
    }

    // The first layer of the sensitivity engines on the vector ALUs as well: the tangent slabs already read the five columns of
    // W0 (transposed copy, see layer_first), so the value pre-activation is five more packed FMAs on the same registers
    // instead of a padded k-tile on the matrix core and a trip through the accumulators.  Four tiles at a time: eight
    // independent chains.  Products enter in the matrix form's order (z0, z4, z1, z2, z3).
    AC_DI void first_valu(const char* wl, const float z[5]) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl);          // plan_sens: the block is [bias 1 KiB][W0 transposed]
        const f32x4* w0t = reinterpret_cast<const f32x4*>(wl + 1024);     // [5][16*WT] floats
        const f32x2 z01 = {z[0], z[1]}, z23 = {z[2], z[3]}, z4x = {z[4], 0.f};
        constexpr int TB = WT < 4 ? WT : 4;
#pragma unroll
        for (int n0 = 0; n0 < WT; n0 += TB) {
            f32x4 w[TB][5], b[TB];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                b[i] = bias4[(n0 + i) * 4 + g];
#pragma unroll
                for (int j = 0; j < 5; ++j) w[i][j] = w0t[j * (WT * 4) + 4 * (n0 + i) + g];
            }
            f32x2 pre[TB][2];
#pragma unroll
            for (int i = 0; i < TB; ++i) { pre[i][0] = f32x2{b[i][0], b[i][1]}; pre[i][1] = f32x2{b[i][2], b[i][3]}; }
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                constexpr int order[5] = {0, 4, 1, 2, 3};
                const int j = order[step];
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x2 wp = {w[i][j][2 * hh], w[i][j][2 * hh + 1]};
                        const f32x2 zp = j < 2 ? z01 : (j < 4 ? z23 : z4x);
                        if (j & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(pre[i][hh]) : "v"(wp), "v"(zp));
                        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pre[i][hh]) : "v"(wp), "v"(zp));
                    }
            }
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = act_tanh(pre[i][r >> 1][r & 1]);
                    if constexpr (!kNoValue) a[0][n0 + i][r] = h;
                    const float sp = fmaf(-h, h, 1.0f);
#pragma unroll
                    for (int j = 0; j < kTangents; ++j) a[kFirstTangent + j][n0 + i][r] = w[i][TOFF + j][r] * sp;
                }
        }
    }

    // The edge layers of the VALUE-ONLY engines (forward kernels: every slab a value slab of its own units, NSLAB = 1 or 4) on
    // the vector ALUs, as first_valu / last_valu do for the sensitivity engines.  Slab s carries units 16 s + col (NSLAB = 4:
    // lane = unit, so z comes from lane col + 16 s) or the lane's own unit (NSLAB = 1).
    AC_DI void first_valu_values(const char* wl, const float z[5]) {
        static_assert(!kDeriv, "value slabs only");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl);
        const f32x4* w0t = reinterpret_cast<const f32x4*>(wl + 1024);
        const int col = lane & 15;
        f32x2 z01[NSLAB], z23[NSLAB], z4x[NSLAB];
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl) {
            float zz[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) zz[k] = NSLAB > 1 ? __shfl(z[k], col + 16 * sl, 64) : z[k];
            z01[sl] = f32x2{zz[0], zz[1]}; z23[sl] = f32x2{zz[2], zz[3]}; z4x[sl] = f32x2{zz[4], 0.f};
        }
        constexpr int TB = WT < 2 ? WT : 2;
#pragma unroll
        for (int n0 = 0; n0 < WT; n0 += TB) {
            f32x4 w[TB][5], b[TB];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                b[i] = bias4[(n0 + i) * 4 + g];
#pragma unroll
                for (int j = 0; j < 5; ++j) w[i][j] = w0t[j * (WT * 4) + 4 * (n0 + i) + g];
            }
            f32x2 pre[NSLAB][TB][2];
#pragma unroll
            for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
                for (int i = 0; i < TB; ++i) { pre[sl][i][0] = f32x2{b[i][0], b[i][1]}; pre[sl][i][1] = f32x2{b[i][2], b[i][3]}; }
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                constexpr int order[5] = {0, 4, 1, 2, 3};
                const int j = order[step];
#pragma unroll
                for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
                    for (int i = 0; i < TB; ++i)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const f32x2 wp = {w[i][j][2 * hh], w[i][j][2 * hh + 1]};
                            const f32x2 zp = j < 2 ? z01[sl] : (j < 4 ? z23[sl] : z4x[sl]);
                            if (j & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(pre[sl][i][hh]) : "v"(wp), "v"(zp));
                            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pre[sl][i][hh]) : "v"(wp), "v"(zp));
                        }
            }
#pragma unroll
            for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[sl][n0 + i][r] = act_tanh(pre[sl][i][r >> 1][r & 1]);
        }
    }

    AC_DI void last_valu_values(const char* wl, int act, float y[6]) {
        static_assert(!kDeriv && (NSLAB == 1 || NSLAB == 4), "value slabs only");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl + 1024) + g * 6;
        const float* bias = reinterpret_cast<const float*>(wl);
        constexpr int kPar = 4;  // one accumulator per row r in BOTH engines: the one-slab engine needs the twelve chains, and a
                                 // unit's result must not depend on which of the two evaluates it (same products, same order)
        f32x2 acc[NSLAB][3][kPar];
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
            for (int kp = 0; kp < 3; ++kp)
#pragma unroll
                for (int q = 0; q < kPar; ++q) acc[sl][kp][q] = f32x2{0.f, 0.f};
        f32x4 w[2][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[0][i] = wv[i];
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            if (t + 1 < WT) {
#pragma unroll
                for (int i = 0; i < 6; ++i) w[(t + 1) & 1][i] = wv[(t + 1) * 24 + i];
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 (&wc)[6] = w[t & 1];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int sl = 0; sl < NSLAB; ++sl) {
                    const f32x2 ap = {a[sl][t][r & 2], a[sl][t][(r & 2) + 1]};
#pragma unroll
                    for (int kp = 0; kp < 3; ++kp) {
                        const f32x4 wq = wc[2 * kp + (r >> 1)];
                        const f32x2 wp = (r & 1) ? f32x2{wq[2], wq[3]} : f32x2{wq[0], wq[1]};
                        f32x2& dst = acc[sl][kp][kPar == 4 ? r : 0];
                        if (r & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(dst) : "v"(ap), "v"(wp));
                        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(dst) : "v"(ap), "v"(wp));
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        float p[NSLAB][6];
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) {
                f32x2 sum = acc[sl][kp][0];
                if constexpr (kPar == 4) sum = (sum + acc[sl][kp][1]) + (acc[sl][kp][2] + acc[sl][kp][3]);
                p[sl][2 * kp] = sum[0]; p[sl][2 * kp + 1] = sum[1];
            }
        float tot[6];
        if constexpr (NSLAB == 4) {
            // lane (col, g) is unit 16 g + col = slab g: the reduce-scatter alone hands row g the total of slab g
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float v[4] = {p[0][k], p[2][k], p[1][k], p[3][k]};
                tot[k] = unit_scatter4(v);
            }
        } else {
            float q0[4] = {p[0][0], p[0][1], p[0][2], p[0][3]}, q1[4] = {p[0][4], p[0][5], 0.f, 0.f};
            unit_totals4(q0); unit_totals4(q1);
            tot[0] = q0[0]; tot[1] = q0[1]; tot[2] = q0[2]; tot[3] = q0[3]; tot[4] = q1[0]; tot[5] = q1[1];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float pre = tot[k] + bias[k];
            y[k] = act ? act_tanh(pre) : pre;
        }
    }

    AC_DI void layer_first(const char* wl) {
        constexpr bool act = true;  // not the last layer (see forward()): always tanh after the host-side fold
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl + WT * 1024);
        const f32x4* w0t = reinterpret_cast<const f32x4*>(wl + WT * 1024 + 1024);  // [5][16*WT] floats
#pragma unroll
        for (int sv = 0; sv < (kDeriv ? 1 : NSLAB); ++sv) {  // every value slab
            f32x4 o[WT];
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) {
                f32x4 acc = bias4[nt * 4 + g];
                const f32x4 w = wf[nt * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mma_16x16x4<USE_MFMA>(w[r], a[sv][0][r], acc);
                o[nt] = acc;
            }
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) epilogue_tile<WT, 1>(sv, nt, o, 1);
        }
        if constexpr (kTangent) {
            // tile by tile: the W0 columns of all tangents are requested together (one LDS wait per tile, not one per
            // read) and act'(h) is formed once per element
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) {
                f32x4 w[kTangents];
#pragma unroll
                for (int j = 0; j < kTangents; ++j) w[j] = w0t[(TOFF + j) * (WT * 4) + 4 * nt + g];
                float sp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = a[0][nt][r];
                    sp[r] = act ? fmaf(-h, h, 1.0f) : 1.0f;
                }
#pragma unroll
                for (int j = 0; j < kTangents; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[1 + j][nt][r] = w[j][r] * sp[r];
            }
        }
        if constexpr (SECOND) {
            // z = W0 in + b is linear in the inputs: h_a = s' W0[:, a], h_ab = s'' W0[:, a] W0[:, b] = -2 h h_a W0[:, b]
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) {
                f32x4 w[kFirstOrder];
#pragma unroll
                for (int i = 0; i < kFirstOrder; ++i) w[i] = w0t[input_of(i) * (WT * 4) + 4 * nt + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = a[0][nt][r], sp = fmaf(-h, h, 1.0f), m2h = -2.0f * h;
                    float ha[kFirstOrder];
#pragma unroll
                    for (int i = 0; i < kFirstOrder; ++i) { ha[i] = w[i][r] * sp; a[1 + i][nt][r] = ha[i]; }
#pragma unroll
                    for (int sl = kFirstOrder + 1; sl < NSLAB; ++sl)  // h_ab = s'' w_a w_b = -2 h (s' w_a) w_b
                        a[sl][nt][r] = m2h * ha[second_a(sl) - 1] * w[second_b(sl) - 1][r];
                }
            }
        }
    }

    // Last layer of a multi-layer net, width -> 6, on the vector ALUs.  A lane holds 4 WT of its unit's 16 WT neurons per slab
    // (rows 16 t + 4 g + r), so it forms the partial sums of the 6 outputs over those with v_pk_fma_f32 (accumulator pair =
    // outputs (2 kp, 2 kp + 1); the activation register is broadcast to both halves by op_sel), 12 WT per slab against the
    // 4 WT exposed MFMAs (of 32 cycles) of the matrix form; the four lanes of a unit are then summed, and the total handed to
    // all four, by unit_totals4 above.  wl: the block [bias 1 KiB]
    // [wlt: per (tile t, lane group g) six float4 {W[2kp][n], W[2kp+1][n], W[2kp][n+1], W[2kp+1][n+1]}, n = 16t+4g+2rp, index
    // (t*4+g)*6 + 2 kp + rp] the host packs (ac_set_mlp).  Outputs go straight to y / J: no output tile, no broadcast.
    // The reduce-scatter half on its own: row g of the result carries the total of v[0], v[2], v[1], v[3] for g = 0, 1, 2, 3.
    AC_DI static float unit_scatter4(const float (&v)[4]) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto sw32 = [](float p, float q, float& lo, float& hi) {
            const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(q), false, false);
            lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
        };
        auto sw16 = [](float p, float q, float& lo, float& hi) {
            const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
            lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
        };
        float l0, h0, l1, h1;
        sw32(v[0], v[1], l0, h0);  // rows (0, 1) now carry v0's halves, rows (2, 3) v1's
        sw32(v[2], v[3], l1, h1);
        const float w0 = l0 + h0, w1 = l1 + h1;
        float l2, h2;
        sw16(w0, w1, l2, h2);
        return l2 + h2;
    }
    // Totals of four per-lane values over the four lanes of a unit (lanes col, col + 16, col + 32, col + 48), handed to all
    // four: a reduce-scatter by v_permlane32_swap / v_permlane16_swap (two values per swap) and the mirror-image all-gather —
    // 12 swaps and 3 adds for four values, every total formed once (the four lanes hold the same bits).
    AC_DI static void unit_totals4(float (&v)[4]) {
        const float u = unit_scatter4(v);  // row 0: total of v0, row 1: v2, row 2: v1, row 3: v3
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto sw32 = [](float p, float q, float& lo, float& hi) {
            const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(q), false, false);
            lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
        };
        auto sw16 = [](float p, float q, float& lo, float& hi) {
            const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
            lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
        };
        float e, o;
        sw16(u, u, e, o);          // e = rows (0, 0, 2, 2) of u, o = rows (1, 1, 3, 3)
        sw32(e, e, v[0], v[1]);
        sw32(o, o, v[2], v[3]);
    }

    template <int JC> AC_DI void last_valu(const char* wl, int act, float y[6], float (*J)[JC]) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl + 1024) + g * 6;
        const float* bias = reinterpret_cast<const float*>(wl);
        float sp[6];
#ifndef AC_LV_GROUP
#define AC_LV_GROUP 6
#endif
#ifndef AC_LV_PARITY
#define AC_LV_PARITY 1
#endif
        // kGroup slabs at a time (the weight image is re-read per group), kPar accumulators per (slab, output pair) — 2 = even
        // and odd rows apart.  All slabs in one pass, one accumulator each: 18 independent chains in the six-slab engine (a
        // dependent v_pk_fma_f32 issues ~38 cycles after its producer, profiles/r02_micro_pkfma_banks.txt: ten chains cover
        // it) and one read of the weight image instead of three — the four waves of the workgroup read it in the same phase,
        // and it was the LDS, not the vector ALU, that set this layer's time: same-box A/B 3.849-3.862 ms against 3.875-3.884
        // for pairs of slabs with split rows, 3.865-3.872 for triples.  (36 accumulator registers next to the 192 activation
        // registers cost 100 B of scratch while the kernel still spilled elsewhere; none now.)
        constexpr int kGroup = NSLAB < AC_LV_GROUP ? NSLAB : AC_LV_GROUP, kPar = AC_LV_PARITY;
#pragma unroll
        for (int s0 = 0; s0 < NSLAB; s0 += kGroup) {
            f32x2 acc[kGroup][3][kPar];
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
#pragma unroll
                for (int kp = 0; kp < 3; ++kp)
#pragma unroll
                    for (int q = 0; q < kPar; ++q) acc[i][kp][q] = f32x2{0.f, 0.f};
            f32x4 w[2][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[0][i] = wv[i];
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if (t + 1 < WT) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) w[(t + 1) & 1][i] = wv[(t + 1) * 24 + i];
                }
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 (&wc)[6] = w[t & 1];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) {
                        if (s0 + i >= NSLAB) continue;
                        const f32x2 ap = {a[s0 + i][t][r & 2], a[s0 + i][t][(r & 2) + 1]};
#pragma unroll
                        for (int kp = 0; kp < 3; ++kp) {
                            const f32x4 wq = wc[2 * kp + (r >> 1)];
                            const f32x2 wp = (r & 1) ? f32x2{wq[2], wq[3]} : f32x2{wq[0], wq[1]};
                            f32x2& dst = acc[i][kp][kPar == 2 ? (r & 1) : 0];
                            if (r & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(dst) : "v"(ap), "v"(wp));
                            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(dst) : "v"(ap), "v"(wp));
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the group's partial sums, flat [slab][output]; unit totals four at a time
            constexpr int kVals = 6 * kGroup, kQuads = (kVals + 3) / 4;
            float p[kQuads * 4];
#pragma unroll
            for (int i = 0; i < kQuads * 4; ++i) p[i] = 0.f;
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) {
                    f32x2 sum = acc[i][kp][0];
                    if constexpr (kPar == 2) sum += acc[i][kp][1];
                    p[i * 6 + 2 * kp] = sum[0]; p[i * 6 + 2 * kp + 1] = sum[1];
                }
            const int live = (NSLAB - s0 < kGroup ? NSLAB - s0 : kGroup) * 6;  // compile-time after unrolling
#pragma unroll
            for (int q = 0; q < kQuads; ++q) {
                if (4 * q >= live) continue;
                float v[4] = {p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]};
                unit_totals4(v);
#pragma unroll
                for (int e = 0; e < 4; ++e) p[4 * q + e] = v[e];
            }
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int sl = s0 + i;
                if (sl >= NSLAB) continue;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const float tot = p[i * 6 + k];
                    if (sl == 0 && !kNoValue) {
                        const float pre = tot + bias[k];
                        y[k] = act ? act_tanh(pre) : pre;
                        sp[k] = act ? fmaf(-y[k], y[k], 1.0f) : 1.0f;
                    } else if constexpr (PAIR != 0) {
                        J[k][sl - kFirstTangent] = tot;  // unscaled: the pair applies act'(y) after its exchange (MlpPairCoeffs)
                    } else {
                        J[k][sl - 1] = sp[k] * tot;
                    }
                }
            }
        }
    }

    // LDS address of layer l's block; for a streamed layer: wait for its DMA, then (the barrier having
    // proven every wave is done with the other slot) request the next streamed layer into that slot.
    AC_DI const char* acquire(int l) {
        if (plan.lds_off[l] >= 0) return lds + plan.lds_off[l];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* wl = lds + plan.ring_off[ring_pos & 1];
        const int nl = plan.streamed[snext];  // one scalar load (was: a scan of lds_off[] and a modulo per call)
        snext = (snext + 1 == plan.n_streamed) ? 0 : snext + 1;
        if constexpr (kSpread) {
            dma_src = (const char*)(gblob + __builtin_amdgcn_readfirstlane(plan.g_off[nl]));
            dma_dst = (unsigned)__builtin_amdgcn_readfirstlane(plan.ring_off[(ring_pos + 1) & 1]);
            dma_last = __builtin_amdgcn_readfirstlane((plan.bytes[nl] >> 10) - 1);
        } else {
            lds_dma_copy(gblob + plan.g_off[nl], lds + plan.ring_off[(ring_pos + 1) & 1], plan.bytes[nl], wave, nwaves,
                         lane);
        }
        ++ring_pos;
        return wl;
    }

    // y[6] (and J[6][5] = dy/dz in tangent mode) of the raw network for normalised inputs z[5].
    // Every lane of a unit passes the same z and receives the same outputs.  Wave-collective and,
    // when layers are streamed, workgroup-collective (one barrier per streamed layer).
    template <int JC> AC_DI void forward(const float z[5], float y[6], float (*J)[JC]) {
        static_assert(!kDeriv || JC >= NSLAB - 1, "J holds one column per derivative slab");
        if constexpr (PAIR != 0) {
            // wave-pair roles (>= 2 layers: the dispatcher keeps single-layer nets off the pair kernel).  y is this role's only
            // for PAIR == 1; J holds this role's UNSCALED last-layer columns.
            const int Lp = plan.n_layers;
            first_valu(acquire(0), z);
#pragma nounroll
            for (int l = 1; l < Lp - 1; ++l) {
                const char* wl = acquire(l);
                if (plan.lds_off[l] >= 0) __syncthreads();  // a streamed layer's acquire() has just passed one
                if constexpr (kNoValue) { if (l > 1) scale_from_hx(); }  // (layer 0's act' this wave formed itself)
                layer<WT, WT, 1>(wl, 1);
            }
            if (Lp > 2) {
                __syncthreads();
                if constexpr (kNoValue) scale_from_hx();
            }
            last_valu<JC>(acquire(Lp - 1), plan.act[Lp - 1], y, J);
            return;
        }
        const int col = lane & 15;
        if constexpr (!kDeriv && NSLAB > 1) {
            // multi-value mode: lane = unit.  Slab s needs z of unit 16 s + col in rows 0..4 (row k on lane group k>>2).
#pragma unroll
            for (int sl = 0; sl < NSLAB; ++sl) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[sl][0][r] = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float v = __shfl(z[k], col + 16 * sl, 64);
                    if (g == (k >> 2)) a[sl][0][k & 3] = v;
                }
            }
        } else {
            // layer-0 input slab: rows 0..4 = z, rest 0 ; tangent slab j = unit vector e_j
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) v = (row == k) ? z[k] : v;
                a[0][0][r] = v;
                if constexpr (kTangent) {
#pragma unroll
                    for (int j = 0; j < kTangents; ++j) a[1 + j][0][r] = (row == TOFF + j) ? 1.f : 0.f;
                }
                if constexpr (SECOND) {
#pragma unroll
                    for (int i = 0; i < kFirstOrder; ++i) a[1 + i][0][r] = (row == input_of(i)) ? 1.f : 0.f;
#pragma unroll
                    for (int sl = kFirstOrder + 1; sl < NSLAB; ++sl) a[sl][0][r] = 0.f;
                }
            }
        }
        const int L = plan.n_layers;
        AC_MARK(st, 1);  // [1] primal aero + input slab
        if (L == 1) {
            // Every engine decides the activation of the last (here: only) layer ONCE — a wave-uniform branch around two
            // straight-line bodies — instead of by a runtime flag inside each slab epilogue.  The runtime form's
            // `act ? tanh(o) : o` selects come out as v_cndmask pairs that the SLP vectoriser packs into 64-bit registers, and
            // SIFoldOperands of the ROCm 7.2.0 backend folds the 32-bit copies of their halves into accumulation registers
            // into a malformed `agpr_32 = REG_SEQUENCE` that loses the subregister index (both halves get the same value):
            // NaN / 1e19 from k_nn_stage_tensors<2,true,0> on single-layer nets.  Found by opt-bisect + MIR diff —
            // profiles/r03_exp_last2_miscompile.txt, tools/archive/bisect_exp_last2.sh; tools/archive/check_sifold_regsequence.sh checks every
            // translation unit of the product for the malformed form.
#ifdef AC_EXP_RUNTIME_ACT  // (experiment flavour: the form that exposes the defect, tools/archive/bisect_exp_last2.sh)
            layer<1, 1>(acquire(0), plan.act[0]);
#else
            const char* wl1 = acquire(0);
            if (plan.act[0]) layer<1, 1, 1>(wl1, 1); else layer<1, 1, 0>(wl1, 0);
#endif
        } else {
            if constexpr (kVLast && kDeriv) first_valu(acquire(0), z);
            else if constexpr (kVLast) first_valu_values(acquire(0), z);
            else layer_first(acquire(0));
            AC_MARK(st, 2);  // [2] first layer
#pragma nounroll
            for (int l = 1; l < L - 1; ++l) {
                const char* wl = acquire(l);
                AC_MARK(st, 3);  // [3] acquire: DMA wait + barrier + DMA issue
                layer<WT, WT, 1>(wl, 1);  // tanh on every layer but the last: ac_set_mlp folds activation-free layers away
                AC_MARK(st, 4);  // [4] hidden layer GEMM + epilogues
            }
            if constexpr (kVLast && !kDeriv) {
                last_valu_values(acquire(L - 1), plan.act[L - 1], y);
                AC_MARK(st, 5);
                AC_MARK(st, 6);
                return;
            } else if constexpr (kVLast) {
                last_valu<JC>(acquire(L - 1), plan.act[L - 1], y, J);
                AC_MARK(st, 5);  // [5] last layer
                AC_MARK(st, 6);
                return;
            } else {  // as for L == 1: no runtime activation flag inside the epilogues
                const char* wll = acquire(L - 1);
                if (plan.act[L - 1]) layer<WT, 1, 1>(wll, 1); else layer<WT, 1, 0>(wll, 0);
            }
            AC_MARK(st, 5);  // [5] last layer
        }
        // outputs: rows 0..5 of tile 0 — row k sits in register k&3 of lane (col, k>>2)
        if constexpr (!kDeriv && NSLAB > 1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float v = 0.f;
#pragma unroll
                for (int sl = 0; sl < NSLAB; ++sl) {
                    const float t = __shfl(a[sl][0][k & 3], col + 16 * (k >> 2), 64);
                    v = (g == sl) ? t : v;  // this lane's unit lives in slab g
                }
                y[k] = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int src = col + 16 * (k >> 2);
                y[k] = __shfl(a[0][0][k & 3], src, 64);
                if constexpr (kDeriv) {  // SECOND: J[k][0..8] = d/dz_{p,q,r}, d2/dz_{p,q,r}2, d2/dz_{pq,pr,qr}
#pragma unroll
                    for (int j = 0; j < NSLAB - 1; ++j) J[k][j] = __shfl(a[1 + j][0][k & 3], src, 64);  // partial: local columns
                }
            }
        }
        AC_MARK(st, 6);  // [6] output broadcast
    }
};

// ---- cooperative forward engine (rollouts) -----------------------------------------------------
// A rollout is sequential in k, so only B instances are parallel: at B = 4096 one wave per 16 instances is
// 256 waves on 1024 SIMDs, and every lone wave has to stream its own 65 KiB of weights per layer.  Here the
// FOUR waves of a workgroup share one 16-instance slab: wave w computes output tiles [w*NT/4, (w+1)*NT/4) of
// every layer (all waves hold the full input slab in registers), the tiles are exchanged through an 8 KiB LDS
// buffer, and the weight ring is filled by all four waves.  4x the parallelism per instance, 1/4 of the MFMA
// chain per wave; the price is two workgroup barriers per layer.
template <int WT, bool USE_MFMA>
struct MlpEngineCoop {
    static constexpr bool kTangent = false;
    static constexpr int kWaves = 4;

    float a[WT][4];  // full input slab of the current layer (values only)
    const MlpPlan& plan;
    const float* __restrict__ gblob;
    char* lds;
    f32x4* xbuf;  // [tile][lane] exchange buffer, 1 KiB per tile
    int lane, g, wave;
    int ring_pos;

    AC_DI MlpEngineCoop(const MlpPlan& pl, const float* blob, char* lds_base)
        : plan(pl), gblob(blob), lds(lds_base), ring_pos(0) {
        lane = threadIdx.x & 63; g = lane >> 4; wave = threadIdx.x >> 6;
        xbuf = reinterpret_cast<f32x4*>(lds_base + pl.lds_total);  // the host reserves WT KiB behind the weights
    }

    AC_DI int streamed_layer(int idx) const {
        int seen = 0;
        for (int l = 0; l < plan.n_layers; ++l)
            if (plan.lds_off[l] < 0) { if (seen == idx) return l; ++seen; }
        return -1;
    }

    // resident layers + the first TWO streamed layers (one per ring slot)
    AC_DI void load_weights() {
        for (int l = 0; l < plan.n_layers; ++l)
            if (plan.lds_off[l] >= 0) lds_dma_copy(gblob + plan.g_off[l], lds + plan.lds_off[l], plan.bytes[l], wave, kWaves, lane);
        if (plan.n_streamed > 0) {
            for (int i = 0; i < 2; ++i) {  // slot i holds the (i mod n_streamed)-th streamed layer
                const int l = streamed_layer(i % plan.n_streamed);
                lds_dma_copy(gblob + plan.g_off[l], lds + plan.ring_off[i], plan.bytes[l], wave, kWaves, lane);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    AC_DI void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    template <int KT, int NT>
    AC_DI void layer(int l, int act) {
        const bool streamed = plan.lds_off[l] < 0;
        const char* wl = streamed ? lds + plan.ring_off[ring_pos & 1] : lds + plan.lds_off[l];
        const f32x4* wf = reinterpret_cast<const f32x4*>(wl) + lane;
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(wl + NT * KT * 1024);
        constexpr int T = NT >= kWaves ? NT / kWaves : 1;  // tiles per wave
        const int t0 = wave * T;
        if (t0 < NT) {  // wave-uniform
            f32x4 acc[T];
#pragma unroll
            for (int i = 0; i < T; ++i) acc[i] = bias4[(t0 + i) * 4 + g];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 w[T];
#pragma unroll
                for (int i = 0; i < T; ++i) w[i] = wf[((t0 + i) * KT + kt) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < T; ++i) acc[i] = mma_16x16x4<USE_MFMA>(w[i][r], a[kt][r], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < T; ++i) {
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = act ? act_tanh(acc[i][r]) : acc[i][r];
                xbuf[(t0 + i) * 64 + lane] = h;
            }
        }
        // one barrier: every tile of this layer is in xbuf, every wave is done with this layer's weights, and
        // (vmcnt) every wave's pieces of the ring slot that the NEXT streamed layer needs have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (streamed) {  // this layer's slot is free: request the streamed layer after next into it
            const int nl = streamed_layer((ring_pos + 2) % plan.n_streamed);
            lds_dma_copy(gblob + plan.g_off[nl], lds + plan.ring_off[ring_pos & 1], plan.bytes[nl], wave, kWaves, lane);
            ++ring_pos;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 h = xbuf[t * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[t][r] = h[r];
        }
        __syncthreads();  // xbuf may be overwritten by the next layer only after everyone has read it
    }

    AC_DI void forward(const float z[5], float y[6], float (*)[5]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) v = (row == k) ? z[k] : v;
            a[0][r] = v;
        }
        const int L = plan.n_layers;
        if (L == 1) {
            layer<1, 1>(0, plan.act[0]);
        } else {
            layer<1, WT>(0, plan.act[0]);
#pragma nounroll
            for (int l = 1; l < L - 1; ++l) layer<WT, WT>(l, plan.act[l]);
            layer<WT, 1>(L - 1, plan.act[L - 1]);
        }
        const int col = lane & 15;
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] = __shfl(a[0][k & 3], col + 16 * (k >> 2), 64);
    }
};

// ---- cooperative forward engine with REGISTER-RESIDENT weights (rollouts) ------------------------------------
// In the cooperative split each wave only ever multiplies by the fragments of its OWN output tiles: T = WT/4 tiles x WT
// k-tiles x 4 floats = 64 registers per 128x128 layer.  A rollout applies the same network H x 4 times, so the wave
// loads its fragments once and keeps them in registers for the whole horizon: no LDS weight image, no LDS-DMA issue,
// no ring; LDS only holds the double-buffered activation exchange (one barrier per layer).  NH = number of hidden
// (width x width) layers; the layer loop is unrolled so the fragment registers are statically indexed.
template <int WT, int NH, bool USE_MFMA>
struct MlpEngineCoopReg {
    static constexpr bool kTangent = false;
    static constexpr int kWaves = 4;
    static constexpr int T = WT >= kWaves ? WT / kWaves : 1;  // output tiles per wave

    float a[WT][4];
    f32x4 w0[T], b0[T];            // first layer: one k-tile
    f32x4 wh[NH][T][WT], bh[NH][T];
    static constexpr int NP = WT >= kWaves ? kWaves : WT;  // waves sharing the last layer's contraction
    static constexpr int KW = WT / NP;                     // k-tiles of the last layer per participating wave
    f32x4 wl[KW], bl;              // last layer (one output tile): this wave's k-tiles; the bias enters through wave 0
    const MlpPlan& plan;
    f32x4* xbuf;                   // [2][WT tiles][64 lanes]
    int lane, g, wave, t0, buf;
    bool active;                   // this wave owns tiles of the wide layers

    AC_DI MlpEngineCoopReg(const MlpPlan& pl, const float* blob, char* lds_base) : plan(pl), buf(0) {
        lane = threadIdx.x & 63; g = lane >> 4; wave = threadIdx.x >> 6;
        xbuf = reinterpret_cast<f32x4*>(lds_base);
        t0 = wave * T;
        active = t0 < WT;
        const int tt = active ? t0 : 0;  // idle waves (WT < 4) shadow tile 0 to stay in bounds
        const f32x4* g0 = reinterpret_cast<const f32x4*>(blob + pl.g_off[0]);
#pragma unroll
        for (int i = 0; i < T; ++i) { w0[i] = g0[(tt + i) * 64 + lane]; b0[i] = g0[WT * 64 + (tt + i) * 4 + g]; }
#pragma unroll
        for (int l = 0; l < NH; ++l) {
            const f32x4* gl = reinterpret_cast<const f32x4*>(blob + pl.g_off[1 + l]);
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int kt = 0; kt < WT; ++kt) wh[l][i][kt] = gl[((tt + i) * WT + kt) * 64 + lane];
                bh[l][i] = gl[WT * WT * 64 + (tt + i) * 4 + g];
            }
        }
        const f32x4* gL = reinterpret_cast<const f32x4*>(blob + pl.g_off[1 + NH]);
#pragma unroll
        for (int i = 0; i < KW; ++i) wl[i] = gL[((wave < NP ? wave : 0) * KW + i) * 64 + lane];
        bl = gL[WT * 64 + g];
    }
    AC_DI void load_weights() {}
    AC_DI void drain() {}

    AC_DI f32x4 activate(f32x4 v, int act) const {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = act ? act_tanh(v[r]) : v[r];
        return h;
    }
    // publish this wave's tiles, one barrier, then every wave pulls the whole slab
    template <int NTILES> AC_DI void exchange() {
        __syncthreads();
        const f32x4* src = xbuf + buf * (WT * 64);
#pragma unroll
        for (int t = 0; t < NTILES; ++t) {
            const f32x4 h = src[t * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[t][r] = h[r];
        }
        buf ^= 1;
    }

    AC_DI void forward(const float z[5], float y[6], float (*)[5]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) v = (row == k) ? z[k] : v;
            a[0][r] = v;
        }
        // first layer: 5 (padded 16) -> 16 WT
        if (active) {
            f32x4* dst = xbuf + buf * (WT * 64);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                f32x4 acc = b0[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mma_16x16x4<USE_MFMA>(w0[i][r], a[0][r], acc);
                dst[(t0 + i) * 64 + lane] = activate(acc, plan.act[0]);
            }
        }
        exchange<WT>();
        // hidden layers
#pragma unroll
        for (int l = 0; l < NH; ++l) {
            if (active) {
                f32x4 acc[T];
#pragma unroll
                for (int i = 0; i < T; ++i) acc[i] = bh[l][i];
#pragma unroll
                for (int kt = 0; kt < WT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < T; ++i) acc[i] = mma_16x16x4<USE_MFMA>(wh[l][i][kt][r], a[kt][r], acc[i]);
                f32x4* dst = xbuf + buf * (WT * 64);
#pragma unroll
                for (int i = 0; i < T; ++i) dst[(t0 + i) * 64 + lane] = activate(acc[i], plan.act[1 + l]);
            }
            exchange<WT>();
        }
        // last layer: ONE output tile, whose 4 WT dependent MFMAs would be a latency chain on one wave while three wait: the
        // contraction is split over the waves' k-tiles instead, the partial tiles are summed (fixed order) by every wave
        if (wave < NP) {
            f32x4 acc = wave == 0 ? bl : f32x4{0.f, 0.f, 0.f, 0.f};
            // (a wave-uniform branch per participant: the activation registers are indexed statically)
#pragma unroll
            for (int w = 0; w < NP; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int i = 0; i < KW; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = mma_16x16x4<USE_MFMA>(wl[i][r], a[w * KW + i][r], acc);
                }
            }
            xbuf[buf * (WT * 64) + wave * 64 + lane] = acc;
        }
        __syncthreads();
        {
            const f32x4* src = xbuf + buf * (WT * 64);
            f32x4 sum = src[lane];
#pragma unroll
            for (int w = 1; w < NP; ++w) sum += src[w * 64 + lane];
            const f32x4 h = activate(sum, plan.act[1 + NH]);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[0][r] = h[r];
            buf ^= 1;
        }
        const int col = lane & 15;
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] = __shfl(a[0][k & 3], col + 16 * (k >> 2), 64);
    }
};

// Coefficient provider that plugs the engine into state_derivative().  prefetch() runs the network on the
// primal aerodynamic inputs of the stage state and keeps only y[6] (+ J[6][5]); operator() then applies the
// output scaler and, for duals, the chain rule  dC = J . d(inputs)  — the custom-Jacobian rule l4casadi
// supplies to CasADi in the reference (coefficient_models.py:93-100).
// FUSED = true additionally offers linearise() / tangent() — the per-direction protocol of state_derivative for duals
// (ac_dynamics.hpp), which needs no N-wide intermediates: the kernels that run two waves per SIMD take it.
template <class Engine, bool FUSED = false> struct MlpCoeffs {
    static constexpr int kModel = AC_MODEL_NN;
    static constexpr bool kFusedTangent = FUSED && Engine::kTangent;
    Engine& eng;
    float y[6];
    float J[Engine::kTangent ? 6 : 1][5];
    AC_DI explicit MlpCoeffs(Engine& e) : eng(e) {}

    AC_DI void linearise(const DevParams& P, const AeroPre<float>& a, const float x[13], const float u[7], float C[6]) {
        (*this)(P, a, x, u, C);
        if constexpr (Engine::kTangent) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int j = 0; j < 5; ++j) J[k][j] = J[k][j] * P.mlp_jscale[k][j];  // (consumed by tangent() only, once per stage)
        }
    }
    AC_DI void tangent(const DevParams& P, const AeroD& d, float dC[6]) const {
        (void)P;
        const float in[5] = {d.qbar, d.alpha, d.beta, d.da, d.de};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) s = fmaf(J[k][j], in[j], s);
            dC[k] = s;
        }
        dC[5] = fmaf(-0.1f * 6.0f * kDeg, d.dr, dC[5]);
    }

    template <class T> AC_DI void prefetch(const DevParams& P, const T x[13], const float uv[7]) {
        float xf[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xf[i] = value_of(x[i]);
        AeroPre<float> a;
        aero_pre(P, xf, a);
        const float in[5] = {a.qbar, a.alpha, a.beta, uv[0], uv[1]};
        float z[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
        eng.forward(z, y, Engine::kTangent ? J : nullptr);
    }

    AC_DI void operator()(const DevParams& P, const AeroPre<float>& a, const float x[13], const float u[7],
                          float C[6]) const {
        (void)a; (void)x;
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
        C[5] += (-0.1f * 6.0f * kDeg) * u[2];
    }

    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7],
                          Dual<N> C[6]) const {
        (void)x;
        static_assert(Engine::kTangent, "dual coefficients need the tangent engine");
        const Dual<N> in[5] = {a.qbar, a.alpha, a.beta, u[0], u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            C[k].v = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) s = fmaf(J[k][j] * P.mlp_jscale[k][j], in[j].d[i], s);
                C[k].d[i] = s;
            }
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
};


// Provider for a wave PAIR that splits the six slabs (k_nn_step_sens_pair): this wave's engine delivers the unscaled
// last-layer columns TOFF .. TOFF + Engine::kTangents - 1 of the Jacobian (and, role 1, y); the pair completes y and J through
// one LDS exchange per network evaluation (`xch`: [16 units][36] floats of this pair; single-buffered — the barriers of the
// next evaluation's hidden layers lie between these reads and the next writes) and applies act'(y) of the last layer.
template <class Engine, int TOFF> struct MlpPairCoeffs {
    static constexpr int kModel = AC_MODEL_NN;
    Engine& eng;
    float* xch;
    float y[6];
    float J[6][5];
    AC_DI MlpPairCoeffs(Engine& e, float* exchange) : eng(e), xch(exchange) {}

    template <class T> AC_DI void prefetch(const DevParams& P, const T x[13], const float uv[7]) {
        float xf[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) xf[i] = value_of(x[i]);
        AeroPre<float> a;
        aero_pre(P, xf, a);
        const float in[5] = {a.qbar, a.alpha, a.beta, uv[0], uv[1]};
        float z[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
        float yl[6], Jl[6][5];
        eng.forward(z, yl, Jl);
        float* buf = xch + (eng.lane & 15) * 36;
        if (eng.g == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if constexpr (!Engine::kNoValue) buf[k] = yl[k];
#pragma unroll
                for (int j = 0; j < Engine::kTangents; ++j) buf[6 + k * 5 + TOFF + j] = Jl[k][j];
            }
        }
        __syncthreads();
        const int act = eng.plan.act[eng.plan.n_layers - 1];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            y[k] = buf[k];
            const float sp = act ? fmaf(-y[k], y[k], 1.0f) : 1.0f;
#pragma unroll
            for (int j = 0; j < 5; ++j) J[k][j] = sp * buf[6 + k * 5 + j];
        }
    }

    AC_DI void operator()(const DevParams& P, const AeroPre<float>& a, const float x[13], const float u[7],
                          float C[6]) const {
        (void)a; (void)x;
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
        C[5] += (-0.1f * 6.0f * kDeg) * u[2];
    }

    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7],
                          Dual<N> C[6]) const {
        (void)x;
        const Dual<N> in[5] = {a.qbar, a.alpha, a.beta, u[0], u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            C[k].v = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) s = fmaf(J[k][j] * P.mlp_jscale[k][j], in[j].d[i], s);
                C[k].d[i] = s;
            }
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
};

}  // namespace ac

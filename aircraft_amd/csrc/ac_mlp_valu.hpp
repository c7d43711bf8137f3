// ac_mlp_valu.hpp — the MLP surrogate on the VECTOR ALUs: "MFMA off", BASELINE configs[1] (cfg2: 3x64).
//
// A wave evaluates the network as a register-tiled fp32 GEMM per layer on v_pk_fma_f32, with no cross-lane operation in the
// inner loop: a lane owns a few rows (units x slabs) and width / 8 output neurons, reads its activation rows and weight
// columns from LDS and accumulates in registers; activations live in an LDS buffer of the wave's own, written by one layer's
// epilogue and read as the next layer's operand; the whole weight image (36 KB for 3 x 64) is resident in LDS.  A tile and
// not a lane-per-neuron or lane-per-unit layout because every weight that has to reach all 64 lanes costs an LDS read
// (~14 wave cycles per 4 weights) or a scalar load whose latency a wave cannot cover (tools/micro/valu_pkfma_sgpr.hip).
//   * MlpEngineTiled8 — the tangent tile of the sensitivity kernels (value + five input tangents per unit): 8 units per wave,
//     eight waves per workgroup = two per SIMD, persistent workgroups with a work queue; the rigid-body / RK4 / dual
//     arithmetic around it is the same code as every other sensitivity kernel's (sens_update<2>: eight lanes per unit).
//   * MlpEngineTiled — the value tile of the forward kernels (step, derivative, getters, rollout): 64 units per wave.
// Hidden widths <= 64 (the activation buffers of the waves and the weights must fit 160 KB of LDS) and at least two layers
// after the host-side fold; other nets keep the cross-lane validation path of ac_mlp.hpp.
#pragma once
#include "ac_kernels_nn.hpp"

namespace ac {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ValuPlan {
    int n_layers;                 // >= 2 (after the fold); layer 0: 8 (5 padded) -> W, hidden: W -> W, last: W -> 8 (6 padded)
    int act_last;                 // tanh on the last layer?
    int w_off[AC_MAX_LAYERS];     // float offset of the layer's weights in the image: [K][N] row-major (k-major) for all
                                  // layers but the last, which is stored transposed [8][K + 4] (padded rows)
    int b_off[AC_MAX_LAYERS];     // float offset of the bias (N floats, zero padded)
    int image_floats;             // padded to a multiple of 256 (whole 1-KiB LDS-DMA pieces)
};

// acc.xy += a.{lo|hi} (broadcast) * w.xy
AC_DI void pk_fma_alo(f32x2& acc, const f32x2& a, const f32x2& w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(w));
}
AC_DI void pk_fma_ahi(f32x2& acc, const f32x2& a, const f32x2& w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(w));
}
// acc.xy = a.lo (broadcast) * w.xy — the first k-step of a contraction starts the accumulators (no zero-fill, no add)
AC_DI void pk_mul_alo(f32x2& acc, const f32x2& a, const f32x2& w) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(a), "v"(w));
}
// acc.xy += a.xy * w.xy (two k-steps at once: the last layer's k-pair partial sums)
AC_DI void pk_fma_pair(f32x2& acc, const f32x2& a, const f32x2& w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}

// The VALUE tile (forward kernels: step, derivative, getters, rollout): 64 units per wave, value rows only, lane = unit for
// the rigid-body code.  Lane (i, j) owns 8 rows (units i + 8 r, r = 0..7) and WIDTH / 8 output neurons; a 4-deep k-step runs
// as two half steps of four rows, everything a half step consumes requested one half step earlier (two weight banks).
template <int WIDTH> struct MlpEngineTiled {
    static_assert(WIDTH == 32 || WIDTH == 64, "hidden width padded to 32 or 64");
    static constexpr bool kTangent = false;
    static constexpr int kTangents = 0;
    static constexpr int HR = 4;              // rows per half step
    static constexpr int R = 2 * HR;          // rows per lane
    static constexpr int S = WIDTH + 4;       // activation row stride in floats (the eight row groups land on distinct bank groups)
    static constexpr int kRows = 8 * R;       // 64 units
    static constexpr int kBufFloats = kRows * S;
    static constexpr int kWaves = 4;
    static constexpr int NB = WIDTH / 8;      // output neurons per lane (8 lane columns): 8 at width 64, 4 at width 32
    static constexpr int NP = NB / 2;         // ... as packed pairs
    static constexpr int NQ = NB / 4;         // ... as 16-byte LDS accesses

    const ValuPlan& plan;
    const float* wimg;   // LDS: weight image
    float* act;          // LDS: this wave's activation buffer [64][S]
    int lane, ti, tj;
    const float* gimg;
    char* lds0;

    static AC_DI int lds_bytes(const ValuPlan& pl) { return pl.image_floats * 4 + kWaves * kBufFloats * 4; }

    AC_DI MlpEngineTiled(const ValuPlan& pl, const float* blob, char* lds_base) : plan(pl) {
        lane = threadIdx.x & 63; ti = lane >> 3; tj = lane & 7;
        wimg = reinterpret_cast<const float*>(lds_base);
        act = reinterpret_cast<float*>(lds_base) + pl.image_floats + (threadIdx.x >> 6) * kBufFloats;
        gimg = blob;
        lds0 = lds_base;
    }

    AC_DI void load_weights() {
        lds_dma_copy(gimg, lds0, plan.image_floats * 4, threadIdx.x >> 6, blockDim.x >> 6, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    AC_DI void drain() {}

    // row of (half u in {0, 1} of this lane, s-th row of the half): unit ti + 8 (HR u + s); S mod 64 = 4, so the eight lane
    // rows read eight distinct bank groups
    AC_DI int row_of(int u, int s) const { return ti + 8 * (HR * u + s); }

    // LDS ordering inside the wave: the LDS executes one wave's operations in issue order; the fence keeps the compiler
    // from moving a read of another lane's data above the write that produced it.
    AC_DI static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    template <int N> AC_DI void load_w(f32x4 (&wv)[4][NQ], const float* __restrict__ w, int k0) const {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int h = 0; h < NQ; ++h)
                wv[kk][h] = *reinterpret_cast<const f32x4*>(w + (k0 + kk) * N + NB * tj + 4 * h);
    }
    AC_DI void load_a(f32x4 (&a)[HR], int u, int k0) const {
#pragma unroll
        for (int s = 0; s < HR; ++s) a[s] = *reinterpret_cast<const f32x4*>(act + row_of(u, s) * S + k0);
    }
    AC_DI void half_step(f32x2 (&acc)[R][NP], int u, const f32x4 (&a)[HR], const f32x4 (&wv)[4][NQ]) const {
#pragma unroll
        for (int s = 0; s < HR; ++s) {
            const f32x2 a01 = {a[s][0], a[s][1]}, a23 = {a[s][2], a[s][3]};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x2 wp = {wv[kk][p >> 1][2 * (p & 1)], wv[kk][p >> 1][2 * (p & 1) + 1]};
                    if (kk == 0) pk_fma_alo(acc[HR * u + s][p], a01, wp);
                    else if (kk == 1) pk_fma_ahi(acc[HR * u + s][p], a01, wp);
                    else if (kk == 2) pk_fma_alo(acc[HR * u + s][p], a23, wp);
                    else pk_fma_ahi(acc[HR * u + s][p], a23, wp);
                }
            }
        }
    }
    // steps k0 and k0 + 4 with the weights of k0 in wA on entry (and of `knext` on exit), half-0 rows of k0 in a0 on entry
    // (and of `knext` on exit)
    template <int N> AC_DI void two_steps(f32x2 (&acc)[R][NP], const float* __restrict__ w, int k0, int knext, f32x4 (&a0)[HR],
                                          f32x4 (&wA)[4][NQ]) const {
        f32x4 a1[HR], wB[4][NQ];
        load_a(a1, 1, k0); load_w<N>(wB, w, k0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        half_step(acc, 0, a0, wA);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a0, 0, k0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        half_step(acc, 1, a1, wA);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a1, 1, k0 + 4); load_w<N>(wA, w, knext);
        __builtin_amdgcn_sched_barrier(0);
        half_step(acc, 0, a0, wB);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a0, 0, knext);
        __builtin_amdgcn_sched_barrier(0);
        half_step(acc, 1, a1, wB);
        __builtin_amdgcn_sched_barrier(0);
    }

    // bias (+ tanh) on every row, then the 8 x NB results go back to the activation buffer as the next layer's operand
    template <bool TANH> AC_DI void epilogue_store(f32x2 (&acc)[R][NP], const float* __restrict__ bias) {
        float b[NB];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + NB * tj + 4 * h);
            b[4 * h] = bq[0]; b[4 * h + 1] = bq[1]; b[4 * h + 2] = bq[2]; b[4 * h + 3] = bq[3];
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = acc[r][p][e] + b[2 * p + e];
                    acc[r][p][e] = TANH ? act_tanh(v) : v;
                }
        wave_sync();  // every lane is done reading this layer's operand rows before they are overwritten
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float* dst = act + row_of(r / HR, r % HR) * S + NB * tj;
#pragma unroll
            for (int h = 0; h < NQ; ++h)
                *reinterpret_cast<f32x4*>(dst + 4 * h) = f32x4{acc[r][2 * h][0], acc[r][2 * h][1], acc[r][2 * h + 1][0], acc[r][2 * h + 1][1]};
        }
        wave_sync();
    }

    template <int K> AC_DI void dense_layer(int l) {
        f32x2 acc[R][NP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[r][p] = f32x2{0.f, 0.f};
        const float* w = wimg + plan.w_off[l];
        f32x4 wA[4][NQ], a0[HR];
        load_w<WIDTH>(wA, w, 0);
        load_a(a0, 0, 0);
#pragma nounroll
        for (int k0 = 0; k0 < K; k0 += 8)
            two_steps<WIDTH>(acc, w, k0, k0 + 8 < K ? k0 + 8 : 0, a0, wA);  // (the last prefetch is a harmless re-read of step 0)
        epilogue_store<true>(acc, wimg + plan.b_off[l]);  // tanh on every layer but the last (ac_set_mlp folds the others)
    }

    // Last layer, WIDTH -> 6 (padded 8): lane (i, j) computes output neuron j of its 8 rows; the packed FMA runs over
    // k-pairs (even / odd partial sums) against the transposed weights Wt[j][k]; 8-deep k-steps, software-pipelined like the
    // dense layers.
    AC_DI void last_layer(int l) {
        const float* wt = wimg + plan.w_off[l] + tj * (WIDTH + 4);  // rows padded by 4: the eight lane columns read eight distinct bank groups
        f32x2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x2{0.f, 0.f};
        f32x4 ra[2][R][2], rw[2][2];
        auto fetch = [&](int b, int k0) {
            rw[b][0] = *reinterpret_cast<const f32x4*>(wt + k0);
            rw[b][1] = *reinterpret_cast<const f32x4*>(wt + k0 + 4);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* ar = act + row_of(r / HR, r % HR) * S + k0;
                ra[b][r][0] = *reinterpret_cast<const f32x4*>(ar);
                ra[b][r][1] = *reinterpret_cast<const f32x4*>(ar + 4);
            }
        };
        auto compute = [&](int b) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                pk_fma_pair(acc[r], f32x2{ra[b][r][0][0], ra[b][r][0][1]}, f32x2{rw[b][0][0], rw[b][0][1]});
                pk_fma_pair(acc[r], f32x2{ra[b][r][0][2], ra[b][r][0][3]}, f32x2{rw[b][0][2], rw[b][0][3]});
                pk_fma_pair(acc[r], f32x2{ra[b][r][1][0], ra[b][r][1][1]}, f32x2{rw[b][1][0], rw[b][1][1]});
                pk_fma_pair(acc[r], f32x2{ra[b][r][1][2], ra[b][r][1][3]}, f32x2{rw[b][1][2], rw[b][1][3]});
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < WIDTH; k0 += 16) {
            fetch(1, k0 + 8);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(0, k0 + 16 < WIDTH ? k0 + 16 : 0);  // (the last prefetch is a harmless re-read of step 0)
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float bj = wimg[plan.b_off[l] + tj];
        float o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v = acc[r][0] + acc[r][1] + bj;
            o[r] = plan.act_last ? act_tanh(v) : v;
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) act[row_of(r / HR, r % HR) * S + tj] = o[r];
        wave_sync();
    }

    // y[6] of the raw network for normalised inputs z[5], lane = unit.  Wave-collective.
    template <int JC> AC_DI void forward(const float z[5], float y[6], float (*J)[JC]) {
        (void)J;
        wave_sync();
        {   // operand row of layer 0: (z, 0, 0, 0)
            float* dst = act + lane * S;
            *reinterpret_cast<f32x4*>(dst) = f32x4{z[0], z[1], z[2], z[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{z[4], 0.f, 0.f, 0.f};
        }
        wave_sync();
        dense_layer<8>(0);
#pragma nounroll
        for (int l = 1; l < plan.n_layers - 1; ++l) dense_layer<WIDTH>(l);
        last_layer(plan.n_layers - 1);
        const float* src = act + lane * S;
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(src), y1 = *reinterpret_cast<const f32x4*>(src + 4);
        y[0] = y0[0]; y[1] = y0[1]; y[2] = y0[2]; y[3] = y0[3]; y[4] = y1[0]; y[5] = y1[1];
        wave_sync();
    }
};

// ---- the tangent tile for TWO waves per SIMD --------------------------------------------------------------------------
// One wave of the vector ALUs issues a v_pk_fma_f32 every ~5.6 cycles and any other vector instruction every ~5; two waves
// on a SIMD issue them every ~4.6 / ~2.6 (tools/micro/valu_pkfma_peak.hip, profiles/r02_micro_valu_pkfma_peak.txt).  So the
// sensitivity kernels of this flavour run EIGHT units per wave, eight waves per workgroup (two per SIMD, 256 registers
// each), instead of sixteen units on one 512-register wave per SIMD:
//   * lane (unit, tj) — unit = lane & 7, tj = lane >> 3 — owns the 6 rows (value + five input tangents) of ONE unit and
//     WIDTH / 8 output neurons: 6 x 8 accumulators at width 64 (half the 16-unit tile's), epilogue still lane-local.  The
//     same eight lanes carry the unit through the rigid-body code, two tangent directions each (Dual<2>, sens_update<2>):
//     78 registers of RK4 carry across a network evaluation instead of 130;
//   * neurons of lane tj: {4 tj + 32 h + e}, e = 0..3, h < WIDTH / 32 — one ds_write_b128 per h covers 8 lanes x 4 banks
//     = all 32 store banks exactly once (the 16-unit tile's contiguous 8 tj + e collides two ways: 12.8 % of its LDS cycles);
//   * rows are slab-major, row = slab * 8 + unit, stride WIDTH + 4 floats: the eight units of a lane group sit 4 banks
//     apart for reads (64 banks) and stores (32 banks) alike;
//   * 2-deep k-steps (6 ds_read_b64 of activations + WIDTH / 16 ds_read_b128 of weights per 48 packed FMAs at width 64),
//     two operand banks, everything a step consumes requested one step earlier — with the sister wave of the SIMD
//     covering what is left of the LDS latency;
//   * LDS: the 36 KB weight image once per workgroup + 8 x 48 x 68 x 4 B = 138 KB, one 8-wave workgroup per CU.
// SL = 6: value + five input tangents per unit (the sensitivity kernels).  SL = 1: the value row alone — the sequential
// rollout of small batches, where what counts is the latency of one network evaluation: 8 instances per wave put a batch
// of 256 on 32 waves with a sixth of the 64-instances-per-wave tile's instructions per evaluation.
// KU: units per wave (8; 4 for the width-64 rollout: 16 lanes per instance halve the instructions of an evaluation again).
template <int WIDTH, int SL = 6, int KU = 8> struct MlpEngineTiled8 {
    static_assert(WIDTH == 32 || WIDTH == 64, "hidden width padded to 32 or 64");
    static_assert(SL == 6 || SL == 1, "value + five tangents, or the value row alone");
    static constexpr bool kTangent = SL > 1;
    static constexpr int kTangents = SL - 1;
    static_assert(KU == 8 || (KU == 4 && WIDTH == 64 && SL == 1), "a lane owns whole groups of four neurons");
    static constexpr int kUnits = KU;              // units per wave
    static constexpr int LPU = 64 / KU;            // lanes per unit
    static constexpr int S = WIDTH + 4;            // activation row stride in floats
    static constexpr int kRows = SL * kUnits;
    static constexpr int kBufFloats = kRows * S;
    static constexpr int kWaves = 8;
    static constexpr int NQ = WIDTH / (4 * LPU);   // 16-byte neuron groups per lane
    static constexpr int NP = 2 * NQ;              // ... as packed pairs
    static constexpr int NB = 4 * NQ;              // neurons per lane

    const ValuPlan& plan;
    const float* wimg;   // LDS: weight image
    float* act;          // LDS: this wave's activation buffer [SL slabs][8 units][S]
    const float* gimg;
    char* lds0;
    int lane, unit, tj;
    Stamper st;

    static AC_DI int lds_bytes(const ValuPlan& pl) { return pl.image_floats * 4 + kWaves * kBufFloats * 4; }

    AC_DI MlpEngineTiled8(const ValuPlan& pl, const float* blob, char* lds_base) : plan(pl) {
        lane = threadIdx.x & 63; unit = lane % KU; tj = lane / KU;
        wimg = reinterpret_cast<const float*>(lds_base);
        act = reinterpret_cast<float*>(lds_base) + pl.image_floats + (threadIdx.x >> 6) * kBufFloats;
        gimg = blob;
        lds0 = lds_base;
    }
    AC_DI void load_weights() {
        lds_dma_copy(gimg, lds0, plan.image_floats * 4, threadIdx.x >> 6, blockDim.x >> 6, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    AC_DI void drain() {}
    AC_DI static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    AC_DI int row_of(int s) const { return s * kUnits + unit; }
    AC_DI int ncol(int h) const { return 4 * tj + 4 * LPU * h; }  // first neuron of this lane's h-th group

    AC_DI void load_w(f32x4 (&wv)[2][NQ], const float* __restrict__ w, int k0) const {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int h = 0; h < NQ; ++h) wv[kk][h] = *reinterpret_cast<const f32x4*>(w + (k0 + kk) * WIDTH + ncol(h));
    }
    AC_DI void load_a(f32x2 (&a)[SL], int k0) const {
#pragma unroll
        for (int s = 0; s < SL; ++s) a[s] = *reinterpret_cast<const f32x2*>(act + row_of(s) * S + k0);
    }
    // acc[6 rows][NP neuron pairs] (+)= A[rows][k0, k0 + 1] * W[k0, k0 + 1][this lane's neurons]; FIRST: the step that starts
    // the accumulators (a multiply instead of 48 zero-fills and the first 48 adds: same values, 0 + a w = a w)
    template <bool FIRST = false>
    AC_DI void step(f32x2 (&acc)[SL][NP], const f32x2 (&a)[SL], const f32x4 (&wv)[2][NQ]) const {
#pragma unroll
        for (int s = 0; s < SL; ++s) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x2 wp = {wv[kk][p >> 1][2 * (p & 1)], wv[kk][p >> 1][2 * (p & 1) + 1]};
                    if (kk == 0) { if (FIRST) pk_mul_alo(acc[s][p], a[s], wp); else pk_fma_alo(acc[s][p], a[s], wp); }
                    else pk_fma_ahi(acc[s][p], a[s], wp);
                }
            }
        }
    }

    // bias + tanh on the value row, (1 - h^2) on the five tangent rows of the same unit — all in this lane — then the 6 x NB
    // results go back to the activation buffer as the next layer's operand
    template <bool TANH> AC_DI void epilogue_store(f32x2 (&acc)[SL][NP], const float* __restrict__ bias) {
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + ncol(h));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 2 * h + (e >> 1), x = e & 1;
                const float v = acc[0][p][x] + bq[e];
                if (TANH) {
                    const float hv = act_tanh(v), sp = fmaf(-hv, hv, 1.0f);
                    acc[0][p][x] = hv;
#pragma unroll
                    for (int s = 1; s < SL; ++s) acc[s][p][x] *= sp;
                } else {
                    acc[0][p][x] = v;
                }
            }
        }
        wave_sync();  // every lane is done reading this layer's operand rows before they are overwritten
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            float* dst = act + row_of(s) * S;
#pragma unroll
            for (int h = 0; h < NQ; ++h)
                *reinterpret_cast<f32x4*>(dst + ncol(h)) =
                    f32x4{acc[s][2 * h][0], acc[s][2 * h][1], acc[s][2 * h + 1][0], acc[s][2 * h + 1][1]};
        }
        wave_sync();
    }

    // SL = 1 (the rollout of small batches): one row per lane makes a 2-deep step four packed FMAs — far shorter than an LDS
    // round trip, so the pipeline below would stall at every step (measured: 4.7 us per network evaluation).  Here the lane's
    // whole operand row is fetched at once and the weights stream in 8-deep chunks, two in flight.
    AC_DI void dense_layer_row(int l) {
        static_assert(SL == 1 || SL == 6, "");
        const float* w = wimg + plan.w_off[l];
        f32x4 arow[WIDTH / 4];
#pragma unroll
        for (int i = 0; i < WIDTH / 4; ++i) arow[i] = *reinterpret_cast<const f32x4*>(act + row_of(0) * S + 4 * i);
        f32x4 wc[2][8][NQ];
        auto fetchw = [&](int b, int k0) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int h = 0; h < NQ; ++h) wc[b][kk][h] = *reinterpret_cast<const f32x4*>(w + (k0 + kk) * WIDTH + ncol(h));
        };
        f32x2 acc[1][NP];
        fetchw(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < WIDTH; k0 += 8) {
            const int b = (k0 >> 3) & 1;
            if (k0 + 8 < WIDTH) fetchw(b ^ 1, k0 + 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = k0 + kk;
                const f32x2 ap = {arow[k >> 2][2 * ((k >> 1) & 1)], arow[k >> 2][2 * ((k >> 1) & 1) + 1]};
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x2 wp = {wc[b][kk][p >> 1][2 * (p & 1)], wc[b][kk][p >> 1][2 * (p & 1) + 1]};
                    if (k == 0) pk_mul_alo(acc[0][p], ap, wp);
                    else if ((k & 1) == 0) pk_fma_alo(acc[0][p], ap, wp);
                    else pk_fma_ahi(acc[0][p], ap, wp);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue_store<true>(reinterpret_cast<f32x2 (&)[SL][NP]>(acc), wimg + plan.b_off[l]);
    }

    AC_DI void dense_layer(int l) {
        if constexpr (SL == 1) { dense_layer_row(l); return; }
        f32x2 acc[SL][NP];
        const float* w = wimg + plan.w_off[l];
        f32x4 wA[2][NQ], wB[2][NQ];
        f32x2 aA[SL], aB[SL];
        load_w(wA, w, 0); load_a(aA, 0);
        // steps 0 and 2 peeled: the first one starts the accumulators
        load_w(wB, w, 2); load_a(aB, 2);
        __builtin_amdgcn_sched_barrier(0);
        step<true>(acc, aA, wA);
        __builtin_amdgcn_sched_barrier(0);
        load_w(wA, w, 4); load_a(aA, 4);
        __builtin_amdgcn_sched_barrier(0);
        step(acc, aB, wB);
        __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
        for (int k0 = 4; k0 < WIDTH; k0 += 4) {
            load_w(wB, w, k0 + 2); load_a(aB, k0 + 2);
            __builtin_amdgcn_sched_barrier(0);
            step(acc, aA, wA);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = k0 + 4 < WIDTH ? k0 + 4 : 0;  // (the last prefetch is a harmless re-read of step 0)
            load_w(wA, w, kn); load_a(aA, kn);
            __builtin_amdgcn_sched_barrier(0);
            step(acc, aB, wB);
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue_store<true>(acc, wimg + plan.b_off[l]);  // tanh on every layer but the last (ac_set_mlp folds the others)
    }

    // First layer in closed form (the operand rows of layer 0 are (z, 0, 0, 0) and the unit vectors): value row = five FMAs
    // per neuron in the tile's own order, tangent row s = row s - 1 of W0.  z is this lane's own unit's: no shuffle.
    AC_DI void first_layer_direct(const float z[5]) {
        const float* w = wimg + plan.w_off[0];
        f32x2 acc[SL][NP];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            f32x4 wr[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) wr[k] = *reinterpret_cast<const f32x4*>(w + k * WIDTH + ncol(h));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 2 * h + (e >> 1), x = e & 1;
                float v = z[0] * wr[0][e];
#pragma unroll
                for (int k = 1; k < 5; ++k) v = fmaf(z[k], wr[k][e], v);
                acc[0][p][x] = v;
#pragma unroll
                for (int sl = 1; sl < SL; ++sl) acc[sl][p][x] = wr[sl - 1][e];
            }
        }
        epilogue_store<true>(acc, wimg + plan.b_off[0]);
    }

    // Last layer, WIDTH -> 6 (padded 8): lane (unit, j) forms output neuron j of its unit's six rows over the whole
    // contraction (packed over k-pairs: even / odd partial sums, against the transposed weights Wt[j][k]), 4-deep steps
    // software-pipelined like the dense layers; the 6 x 8 results then cross to all eight lanes of the unit through the
    // buffer (columns 0..7 of the unit's rows).
    AC_DI void last_layer(int l) {
        const float* wt = wimg + plan.w_off[l] + (tj & 7) * (WIDTH + 4);  // (16 lanes per unit: lanes 8..15 shadow 0..7)
        if constexpr (SL == 1) {  // one row: fetch it and the lane's weight row whole, then WIDTH / 2 packed FMAs (same order)
            f32x4 ra1[WIDTH / 4], rw1[WIDTH / 4];
#pragma unroll
            for (int i = 0; i < WIDTH / 4; ++i) {
                ra1[i] = *reinterpret_cast<const f32x4*>(act + row_of(0) * S + 4 * i);
                rw1[i] = *reinterpret_cast<const f32x4*>(wt + 4 * i);
            }
            f32x2 a1 = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < WIDTH / 4; ++i) {
                pk_fma_pair(a1, f32x2{ra1[i][0], ra1[i][1]}, f32x2{rw1[i][0], rw1[i][1]});
                pk_fma_pair(a1, f32x2{ra1[i][2], ra1[i][3]}, f32x2{rw1[i][2], rw1[i][3]});
            }
            const float v = a1[0] + a1[1] + wimg[plan.b_off[l] + (tj & 7)];
            const float o1 = plan.act_last ? act_tanh(v) : v;
            wave_sync();
            if (tj < 8) act[row_of(0) * S + tj] = o1;
            wave_sync();
            return;
        }
        f32x2 acc[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) acc[s] = f32x2{0.f, 0.f};
        f32x4 ra[2][SL], rw[2];
        auto fetch = [&](int b, int k0) {
            rw[b] = *reinterpret_cast<const f32x4*>(wt + k0);
#pragma unroll
            for (int s = 0; s < SL; ++s) ra[b][s] = *reinterpret_cast<const f32x4*>(act + row_of(s) * S + k0);
        };
        auto compute = [&](int b) {
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                pk_fma_pair(acc[s], f32x2{ra[b][s][0], ra[b][s][1]}, f32x2{rw[b][0], rw[b][1]});
                pk_fma_pair(acc[s], f32x2{ra[b][s][2], ra[b][s][3]}, f32x2{rw[b][2], rw[b][3]});
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < WIDTH; k0 += 8) {
            fetch(1, k0 + 4);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(0, k0 + 8 < WIDTH ? k0 + 8 : 0);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float bj = wimg[plan.b_off[l] + tj];
        float o[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) o[s] = acc[s][0] + acc[s][1];
        {
            const float v = o[0] + bj;
            if (plan.act_last) {
                const float hv = act_tanh(v), sp = fmaf(-hv, hv, 1.0f);
                o[0] = hv;
#pragma unroll
                for (int s = 1; s < SL; ++s) o[s] *= sp;
            } else {
                o[0] = v;
            }
        }
        wave_sync();
#pragma unroll
        for (int s = 0; s < SL; ++s) act[row_of(s) * S + tj] = o[s];
        wave_sync();
    }

    // The outputs of the last forward_keep(): y[6], J[6][5] of this lane's unit, from columns 0..7 of the unit's six rows.
    // They stay in the buffer until the next evaluation overwrites them, so the rigid-body code reads them where it uses
    // them instead of carrying 36 registers through the dual aerodynamic arithmetic in between.
    AC_DI void read_outputs(float y[6], float (*J)[5]) const {
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(act + row_of(0) * S), y1 = *reinterpret_cast<const f32x4*>(act + row_of(0) * S + 4);
        y[0] = y0[0]; y[1] = y0[1]; y[2] = y0[2]; y[3] = y0[3]; y[4] = y1[0]; y[5] = y1[1];
#pragma unroll
        for (int s = 1; s < SL; ++s) {
            const f32x4 j0 = *reinterpret_cast<const f32x4*>(act + row_of(s) * S), j1 = *reinterpret_cast<const f32x4*>(act + row_of(s) * S + 4);
            J[0][s - 1] = j0[0]; J[1][s - 1] = j0[1]; J[2][s - 1] = j0[2]; J[3][s - 1] = j0[3];
            J[4][s - 1] = j1[0]; J[5][s - 1] = j1[1];
        }
    }

    // The network for normalised inputs z[5] (the eight lanes of a unit pass the same z); outputs stay in LDS: read_outputs().
    // Wave-collective.
    AC_DI void forward_keep(const float z[5]) {
        AC_MARK(st, 1);
        wave_sync();
        AC_MARK(st, 3);
        first_layer_direct(z);
        AC_MARK(st, 2);
#pragma nounroll
        for (int l = 1; l < plan.n_layers - 1; ++l) dense_layer(l);
        AC_MARK(st, 4);
        last_layer(plan.n_layers - 1);
        AC_MARK(st, 5);
        AC_MARK(st, 6);
    }
    // the MlpCoeffs interface of the value-only engines: y now (J unused)
    template <int JC> AC_DI void forward(const float z[5], float y[6], float (*J)[JC]) {
        static_assert(SL == 1, "the tangent engine keeps its outputs in LDS: MlpLazyCoeffs");
        forward_keep(z);
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(act + row_of(0) * S), y1 = *reinterpret_cast<const f32x4*>(act + row_of(0) * S + 4);
        y[0] = y0[0]; y[1] = y0[1]; y[2] = y0[2]; y[3] = y0[3]; y[4] = y1[0]; y[5] = y1[1];
        (void)J;
    }
};

// Coefficient provider on an engine that keeps its outputs in LDS (MlpEngineTiled8): prefetch() runs the network on the
// primal aerodynamic inputs of the stage state, operator() reads y, J back and applies the output scaler and the chain rule
// dC = J . d(inputs) — the arithmetic of MlpCoeffs (ac_mlp.hpp), which see for the reference lines.
template <class Engine> struct MlpLazyCoeffs {
    static constexpr int kModel = AC_MODEL_NN;
    Engine& eng;
    AC_DI explicit MlpLazyCoeffs(Engine& e) : eng(e) {}

    // The dual aerodynamic quantities of the stage state are formed ONCE, before the network (their value parts are the
    // network's inputs), and kept across it for state_derivative() — 21 registers live over the evaluation instead of a
    // second, primal-only pass through the rotation, sqrt, atan2 and asin (the matrix-core kernels make the opposite choice:
    // their six activation slabs leave no room).
    AeroPre<Dual<2>> kept;
    AC_DI void prefetch(const DevParams& P, const Dual<2> x[13], const float uv[7]) {
        aero_pre(P, x, kept);
        const float in[5] = {kept.qbar.v, kept.alpha.v, kept.beta.v, uv[0], uv[1]};
        float z[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) z[j] = (in[j] - P.mlp_in_mean[j]) / P.mlp_in_std[j];
        eng.forward_keep(z);
    }
    AC_DI void kept_aero(AeroPre<Dual<2>>& a) const { a = kept; }

    template <int N>
    AC_DI void operator()(const DevParams& P, const AeroPre<Dual<N>>& a, const Dual<N> x[13], const Dual<N> u[7],
                          Dual<N> C[6]) const {
        (void)x;
        float y[6], J[6][5];
        eng.read_outputs(y, J);
        const Dual<N> in[5] = {a.qbar, a.alpha, a.beta, u[0], u[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            C[k].v = fmaf(y[k], P.mlp_out_std[k], P.mlp_out_mean[k]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float sacc = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) sacc = fmaf(J[k][j] * P.mlp_jscale[k][j], in[j].d[i], sacc);
                C[k].d[i] = sacc;
            }
        }
        C[5] = C[5] + (-0.1f * 6.0f * kDeg) * u[2];
    }
};

constexpr int kBlock8 = 512;  // eight waves: two per SIMD

// lane = col + 8 g: col = unit of the wave (0..7), g = tangent-direction pair (0..7) — the layout of sens_update<2>
struct WaveUnit8 {
    int lane, col, g;
    long unit;
    bool live;
    UnitAddr ua;
    // group `grp` of eight consecutive units; dead lanes shadow the last unit so that the wave-collective engine stays uniform
    // `tid`: the caller's (laundered) copy of threadIdx.x — see launder_tid()
    AC_DI WaveUnit8(int tid, long grp, long n, long blk)
        : lane(tid & 63), col(tid & 7), g((tid & 63) >> 3),
          unit(grp * 8 + (tid & 7) < n ? grp * 8 + (tid & 7) : n - 1), live(grp * 8 + (tid & 7) < n), ua(unit, blk) {}
};
// threadIdx.x behind an empty asm, taken anew in every iteration of the persistent loop: whatever depends on the lane
// (seed patterns, lane-group selects, addresses) is then rebuilt per group — a handful of instructions — instead of being
// hoisted out of the loop into registers that the loop body cannot spare (hipcc spilled them: 30 dwords at kernel entry).
AC_DI int launder_tid() {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    return tid;
}

// PERSISTENT workgroups with a work queue: the grid is at most one 8-wave workgroup per CU, and every wave draws unit groups
// (8 units) from one ticket counter in global memory until the tickets run out.  Why not a workgroup per 64 units, or a
// static share per wave: the two waves of a SIMD do not progress alike (the issue arbiter favours one of them; measured with
// -DAC_CLOCKS: mean wave lifetime 1017 us, longest 1223 us = the launch), so with equal static shares the favoured wave
// finishes early and its sister runs the rest alone, at one wave's issue rate; and with one 141 KB workgroup resident per CU,
// a workgroup per 64 units had every CU wait for the slowest of eight waves before the next eight could start.  Drawing
// tickets keeps every wave busy until the queue is empty.  The weight image is loaded once per CU.
//   * most of a wave's groups are its own static slots; the last ones come from the queue, and the wave holds the ticket of
//     its NEXT group while it computes the current one;
//   * every wave draws until it gets an invalid ticket, so a launch consumes a known number of tickets: the wave that draws
//     the last of them (all other draws are then complete) stores 0 for the next launch — launches of one handle that use
//     the queue must be ordered on one stream, like its second-order workspace;
//   * a unit's result does not depend on which wave computes it: same arithmetic, same bits;
//   * every wave leaves the loop after its first invalid ticket: the grid drains by itself.
struct GroupQueue {
    unsigned* counter;
    unsigned ngroups, waves, n_static, total;
    // static rounds: every wave owns the groups slot + k * waves, k < rounds; tickets hand out the rest (about the last two
    // and a half groups per wave — the favoured waves of the static phase run ahead by about one group in ten, and the pool
    // has to absorb that).  A launch with at most one group per wave (cfg2's own 12 800 units) draws no ticket at all.
    AC_DI GroupQueue(unsigned* c, long n) : counter(c), ngroups((unsigned)((n + 7) / 8)), waves(gridDim.x * (kBlock8 >> 6)) {
        const unsigned per_wave = ngroups / waves;
        const unsigned rounds = per_wave > 3 ? per_wave - 2 : (per_wave > 0 ? per_wave : 1);
        n_static = rounds * waves < ngroups ? rounds * waves : ngroups;
        total = n_static < ngroups ? (ngroups - n_static) + waves : 0;  // every wave ends on one invalid ticket
    }
    // wave-major slots: slot = wave * workgroups + workgroup (a batch smaller than the chip spreads over all CUs first)
    AC_DI unsigned first() const { return (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * gridDim.x + blockIdx.x; }
    // does the group AFTER `cur` come from the queue?  (then draw() its ticket while `cur` computes)
    AC_DI bool next_is_ticket(unsigned cur) const { return total != 0 && cur + waves >= n_static; }
    // lane 0's ticket (a vector register: the wait for the atomic sits where the value is first used)
    AC_DI unsigned draw() const {
        unsigned t = 0;
        if ((threadIdx.x & 63) == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    }
    // the group after `cur` (>= ngroups: this wave is done).  The launch consumes exactly `total` tickets; whoever holds the
    // last one (every other draw is then complete) resets the counter for the next launch.
    AC_DI unsigned next(unsigned cur, bool ticket, unsigned ticket_v) const {
        if (!ticket) return cur + waves < n_static ? cur + waves : ngroups;  // (no queue and the static share is done: stop)
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket_v);
        if (t == total - 1 && (threadIdx.x & 63) == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return n_static + t;
    }
};

// The fused step + sensitivities kernel of the "MFMA off" flavour (BASELINE configs[1]).
template <int WIDTH>
__global__ __launch_bounds__(kBlock8) void k_nn_step_sens_tiled8(const DevParams P, const ValuPlan plan,
                                                                 const float* __restrict__ blob,
                                                                 const float* __restrict__ X, const float* __restrict__ U,
                                                                 float dt, const float* __restrict__ dt_per_unit, long n,
                                                                 long blk, float* __restrict__ Xn, float* __restrict__ A,
                                                                 float* __restrict__ Bm, float* __restrict__ c,
                                                                 unsigned* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled8<WIDTH> Engine;
    Engine eng(plan, blob, smem);
    eng.st.start();
    WaveClock wc;
    wc.start();
    eng.load_weights();
    AC_MARK(eng.st, 0);  // [0] prologue: weight image into LDS
#if defined(AC_STAMPS) || defined(AC_CLOCKS)
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(c);  // the diagnostic flavours' buffer travels in `c`
    c = nullptr;
#endif
    const GroupQueue q(queue, n);
    MlpLazyCoeffs<Engine> coeffs(eng);
    unsigned grp = q.first();
#pragma nounroll
    while (grp < q.ngroups) {
        const WaveUnit8 w(launder_tid(), grp, n, blk);
        float xv[13], uv[7];
        load_rows<13>(X, w.ua, xv);
        load_rows<7>(U, w.ua, uv);
        const float hv = dt_per_unit ? dt_per_unit[w.unit] : dt;
        // the next ticket, if the next group comes from the queue: in flight while this group computes.  Issued BEHIND the
        // input loads: vector memory operations return in order, so a wait for x, u would also wait for the atomic.
        const bool ticket = q.next_is_ticket(grp);
        unsigned next_ticket = 0;
        if (ticket) next_ticket = q.draw();
        Dual<2> x[13];
        sens_update<2>(P, coeffs, w.g, w.col, w.ua, xv, uv, hv, x, A, Bm, c, w.live);
        AC_MARK(eng.st, 7);  // [7] dual rigid body + RK4 combine after the last network evaluation
        // the ticket is taken BEFORE the result stores are issued: its wait (vmcnt counts stores too on this chip) then covers
        // the long-complete atomic only, and the stores drain under the next group (or after the wave has gone)
        grp = q.next(grp, ticket, next_ticket);
        if (w.live) {
            const UnitAddr uo = w.ua.late();
            if (w.g == 0) {
                float* p = Xn + uo.off(13);
#pragma unroll
                for (int i = 0; i < 13; ++i) p[(long)i * blk] = x[i].v;
            }
            SensIOT<2>::store(w.g, uo, x, A, Bm, c, true);
        }
    }
#ifdef AC_STAMPS
    eng.st.flush(stamp_buf);
#endif
#ifdef AC_CLOCKS
    wc.stop(stamp_buf);
#endif
}

template <int WIDTH>
__global__ __launch_bounds__(kBlock8) void k_nn_deriv_sens_tiled8(const DevParams P, const ValuPlan plan,
                                                                  const float* __restrict__ blob,
                                                                  const float* __restrict__ X, const float* __restrict__ U,
                                                                  long n, long blk, float* __restrict__ Xdot,
                                                                  float* __restrict__ Fx, float* __restrict__ Fu,
                                                                  unsigned* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled8<WIDTH> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const GroupQueue q(queue, n);
    MlpLazyCoeffs<Engine> coeffs(eng);
    unsigned grp = q.first();
#pragma nounroll
    while (grp < q.ngroups) {
        const WaveUnit8 w(launder_tid(), grp, n, blk);
        float xv[13], uv[7];
        load_rows<13>(X, w.ua, xv);
        load_rows<7>(U, w.ua, uv);
        const bool ticket = q.next_is_ticket(grp);
        unsigned next_ticket = 0;
        if (ticket) next_ticket = q.draw();  // behind the input loads (see k_nn_step_sens_tiled8)
        Dual<2> k[13];
        deriv_seeded<2>(P, coeffs, w.g, xv, uv, k);
        grp = q.next(grp, ticket, next_ticket);  // before the stores (see k_nn_step_sens_tiled8)
        if (w.live) deriv_store<2, false>(P, X, w.unit, w.g, w.ua, k, Xdot, Fx, Fu);
    }
}

// ---- forward kernels on the vector ALUs: 64 units per wave, lane = unit (the value-only tile) ---------------------------
template <int WIDTH, int OP>
__global__ __launch_bounds__(kBlock, 1) void k_nn_fwd_tiled(const DevParams P, const ValuPlan plan,
                                                            const float* __restrict__ blob, const float* __restrict__ X,
                                                            const float* __restrict__ U, float dt,
                                                            const float* __restrict__ dt_per_unit, long n, long blk,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled<WIDTH> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const long raw = (long)blockIdx.x * kBlock + threadIdx.x;
    const bool live = raw < n;
    const long unit = live ? raw : n - 1;  // dead lanes shadow the last unit: the engine is wave-collective
    const UnitAddr ua(unit, blk);
    float x[13], u[7];
    load_rows<13>(X, ua, x);
    load_rows<7>(U, ua, u);
    MlpCoeffs<Engine> coeffs(eng);
    if constexpr (OP == OP_DERIV) {
        float xd[13];
        coeffs.prefetch(P, x, u);
        state_derivative<float>(P, coeffs, x, u, xd);
        if (live) store_state_rows(P, X, out, ua.late(), unit, xd);
    } else if constexpr (OP == OP_STEP) {
        const float h = dt_per_unit ? dt_per_unit[unit] : dt;
        state_update(P, coeffs, x, u, h);
        if (live) store_state_rows(P, X, out, ua.late(), unit, x);
    } else {
        coeffs.prefetch(P, x, u);
        AeroPre<float> a;
        aero_pre(P, x, a);
        float C[6];
        coeffs(P, a, x, u, C);
        AeroPost<float> o;
        aero_post(P, a, u, C, o);
        float eu[3];
        euler_angles(x, eu[0], eu[1], eu[2]);
        const float v[22] = {a.vr[0], a.vr[1], a.vr[2], a.V, a.alpha, a.beta, a.qbar, o.C[0], o.C[1], o.C[2],
                             o.C[3], o.C[4], o.C[5], o.F[0], o.F[1], o.F[2], o.M[0], o.M[1], o.M[2], eu[0], eu[1], eu[2]};
        if (live) store_rows<22>(out, ua.late(), v);
    }
}

// Sequential rollout, lane = instance (64 per wave), state carried in float64 like every rollout kernel.
template <int WIDTH>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout_tiled(const DevParams P, const ValuPlan plan,
                                                                const float* __restrict__ blob,
                                                                const float* __restrict__ X0, const float* __restrict__ U,
                                                                float dt, long B, long H, float* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled<WIDTH> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const long raw = (long)blockIdx.x * kBlock + threadIdx.x;
    const bool live = raw < B;
    const long i = live ? raw : B - 1;
    float x[13], u[7], un[7];
    load_rows<13>(X0, B, i, x);
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (live) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * B + i] = x[r];
    }
    if (H > 0) load_rows<7>(U, B, i, u);
    MlpCoeffs<Engine> coeffs(eng);
    for (long k = 0; k < H; ++k) {
        if (k + 1 < H) load_rows<7>(U + (k + 1) * 7 * B, B, i, un);
        state_update_carry(P, coeffs, xa, u, dt);
        if (live) {
            float* o = Xout + (k + 1) * 13 * B;
#pragma unroll
            for (int r = 0; r < 13; ++r) o[(long)r * B + i] = (float)xa[r];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) u[r] = un[r];
    }
}

// Sequential rollout of SMALL batches (cfg2's own B = 256): 8 instances per wave on the value-only 8-unit tile, the eight
// lanes of an instance compute the same rigid-body values (its share of the work is small; what counts here is the latency
// of the 4 H sequential network evaluations), lane tj = 0 writes.  State carried in float64 like every rollout kernel.
template <int WIDTH> constexpr int kRolloutUnits = WIDTH == 64 ? 4 : 8;  // instances per wave of k_nn_rollout_tiled8
template <int WIDTH>
__global__ __launch_bounds__(kBlock) void k_nn_rollout_tiled8(const DevParams P, const ValuPlan plan,
                                                              const float* __restrict__ blob,
                                                              const float* __restrict__ X0, const float* __restrict__ U,
                                                              float dt, long B, long H, float* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled8<WIDTH, 1, kRolloutUnits<WIDTH>> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const long raw = ((long)blockIdx.x * (kBlock >> 6) + (threadIdx.x >> 6)) * Engine::kUnits + eng.unit;
    const bool live = raw < B, writer = live && eng.tj == 0;
    const long i = live ? raw : B - 1;
    float x[13], u[7], un[7];
    load_rows<13>(X0, B, i, x);
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (writer) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * B + i] = x[r];
    }
    if (H > 0) load_rows<7>(U, B, i, u);
    MlpCoeffs<Engine> coeffs(eng);
    for (long k = 0; k < H; ++k) {
        if (k + 1 < H) load_rows<7>(U + (k + 1) * 7 * B, B, i, un);
        state_update_carry(P, coeffs, xa, u, dt);
        if (writer) {
            float* o = Xout + (k + 1) * 13 * B;
#pragma unroll
            for (int r = 0; r < 13; ++r) o[(long)r * B + i] = (float)xa[r];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) u[r] = un[r];
    }
}

// Closed-loop (feedback policy) rollout of this flavour: the same small-batch tile, the control of every node from the
// solver's gains (Policy::control; every lane of an instance computes the same 7 x 13 product from the same addresses, so a
// wave fetches each gain once).  Output instance o = a * B + b (a = line-search index), as the MFMA kernels.
template <int WIDTH>
__global__ __launch_bounds__(kBlock) void k_nn_rollout_policy_tiled8(const DevParams P, const ValuPlan plan,
                                                                     const float* __restrict__ blob, const Policy pol,
                                                                     const float* __restrict__ X0, float dt, long Bout,
                                                                     long H, float* __restrict__ Xout,
                                                                     float* __restrict__ Uout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled8<WIDTH, 1, kRolloutUnits<WIDTH>> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const long raw = ((long)blockIdx.x * (kBlock >> 6) + (threadIdx.x >> 6)) * Engine::kUnits + eng.unit;
    const bool live = raw < Bout, writer = live && eng.tj == 0;
    const long o = live ? raw : Bout - 1;
    float x[13], u[7];
    load_rows<13>(X0, pol.B, o % pol.B, x);
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (writer) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * Bout + o] = x[r];
    }
    MlpCoeffs<Engine> coeffs(eng);
    for (long k = 0; k < H; ++k) {
#pragma unroll
        for (int r = 0; r < 13; ++r) x[r] = (float)xa[r];
        pol.control(k, o, x, u);
        if (writer) {
#pragma unroll
            for (int r = 0; r < 7; ++r) Uout[(k * 7 + r) * Bout + o] = u[r];
        }
        state_update_carry(P, coeffs, xa, u, pol.step(u, dt));
        if (writer) {
            float* out = Xout + (k + 1) * 13 * Bout;
#pragma unroll
            for (int r = 0; r < 13; ++r) out[(long)r * Bout + o] = (float)xa[r];
        }
    }
}

}  // namespace ac

// ac_mlp_valu.hpp — the MLP surrogate on the VECTOR ALUs: "MFMA off", BASELINE configs[1] (cfg2: 3x64).
//
// A wave evaluates the network for 16 units x 6 slabs (value + five input tangents) = 96 rows as a register-tiled fp32
// GEMM per layer on v_pk_fma_f32, with no cross-lane operation in the inner loop:
//   * lane (i, j), i = lane >> 3, j = lane & 7, owns the 12 rows of units i and i + 8 (all six slabs of a unit sit in ONE
//     lane, so the tanh / (1 - h^2) epilogue is lane-local) and width / 8 of the layer's output neurons: 12 x 8
//     accumulators at width 64;
//   * per 4-deep k-step the lane reads its 12 activation rows (12 ds_read_b128) and its 8 weight columns (8 ds_read_b128)
//     from LDS and issues 192 v_pk_fma_f32 — one LDS read per 9.6 packed FMAs (tools/micro/valu_pkfma_sgpr.hip measured
//     why it is a tile and not a lane-per-neuron or lane-per-unit layout: every weight that has to reach all 64 lanes costs
//     an LDS read (~14 wave cycles per 4 weights) or a scalar load whose latency one 102-SGPR wave cannot cover);
//   * activations live in an LDS buffer of the wave's own, [96 rows][width + 4] floats — the +4 padding spreads the eight
//     row groups over distinct banks — written by the epilogue of one layer and read as the operand of the next;
//     the whole weight image (36 KB for 3 x 64) is resident in LDS;
//   * the rigid-body / RK4 / dual arithmetic around it is the SAME code as the matrix-core kernels' (16 units x 4 lanes,
//     Dual<4>): the engine presents the same forward(z, y, J) interface as MlpEngine.
// Hidden widths <= 64 (the activation buffers of four waves and the weights must fit 160 KB of LDS) and at least two
// layers after the host-side fold; other nets keep the cross-lane validation path of ac_mlp.hpp.
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
// acc.xy += a.xy * w.xy (two k-steps at once: the last layer's k-pair partial sums)
AC_DI void pk_fma_pair(f32x2& acc, const f32x2& a, const f32x2& w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}

// TANGENT = true: 16 units per wave x 6 slabs (value + five input tangents) = 96 rows; the rigid-body code around it runs
// 16 units x 4 lanes (the sensitivity kernels).  TANGENT = false: 64 units per wave, value rows only = 64 rows; lane = unit
// for the rigid-body code (the forward kernels: step, derivative, getters, rollout).  Same tile, same pipeline: a lane owns
// 2 x HR rows (HR = 6 | 4) and WIDTH / 8 output neurons.
template <int WIDTH, bool TANGENT = true> struct MlpEngineTiled {
    static_assert(WIDTH == 32 || WIDTH == 64, "hidden width padded to 32 or 64");
    static constexpr bool kTangent = TANGENT;
    static constexpr int kTangents = TANGENT ? 5 : 0;
    static constexpr int HR = TANGENT ? 6 : 4;    // rows per half step: one unit's six slabs | four units' value rows
    static constexpr int R = 2 * HR;              // rows per lane
    static constexpr int S = WIDTH + 4;       // activation row stride in floats (the eight row groups land on distinct bank groups)
    static constexpr int kRows = 8 * R;       // 96 = 16 units x 6 slabs | 64 units
    static constexpr int kBufFloats = kRows * S;
    static constexpr int kWaves = 4;
    static constexpr int NB = WIDTH / 8;      // output neurons per lane (8 lane columns): 8 at width 64, 4 at width 32
    static constexpr int NP = NB / 2;         // ... as packed pairs
    static constexpr int NQ = NB / 4;         // ... as 16-byte LDS accesses

    const ValuPlan& plan;
    const float* wimg;   // LDS: weight image
    float* act;          // LDS: this wave's activation buffer [96][S]
    int lane, g, col, ti, tj;
    Stamper st;

    static AC_DI int lds_bytes(const ValuPlan& pl) { return pl.image_floats * 4 + kWaves * kBufFloats * 4; }

    AC_DI MlpEngineTiled(const ValuPlan& pl, const float* blob, char* lds_base) : plan(pl) {
        lane = threadIdx.x & 63; g = lane >> 4; col = lane & 15; ti = lane >> 3; tj = lane & 7;
        wimg = reinterpret_cast<const float*>(lds_base);
        act = reinterpret_cast<float*>(lds_base) + pl.image_floats + (threadIdx.x >> 6) * kBufFloats;
        gimg = blob;
        lds0 = lds_base;
    }
    const float* gimg;
    char* lds0;

    AC_DI void load_weights() {
        lds_dma_copy(gimg, lds0, plan.image_floats * 4, threadIdx.x >> 6, blockDim.x >> 6, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    AC_DI void drain() {}

    // row of (half u in {0, 1} of this lane, s-th row of the half).  Tangent engine: units ti and ti + 8, row = unit * 6 + slab
    // (6 S mod 64 = 24: the eight lane rows read eight distinct bank groups).  Forward engine: row = unit = ti + 8 (HR u + s)
    // (S mod 64 = 4: likewise).
    AC_DI int row_of(int u, int s) const { return TANGENT ? (ti + 8 * u) * 6 + s : ti + 8 * (HR * u + s); }

    // LDS ordering inside the wave: the LDS executes one wave's operations in issue order; the fence keeps the compiler
    // from moving a read of another lane's data above the write that produced it.
    AC_DI static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // One 4-deep k-step:  acc[12 rows][NP neuron pairs] += A[rows][k0 .. k0+3] * W[k0 .. k0+3][NB tj .. NB tj + NB - 1],
    // run as two HALF steps of six rows (one unit's six slabs each).  Everything a half step consumes was requested one
    // half step (96 packed FMAs) earlier: while unit 0's rows compute, unit 1's rows and the next step's weight fragment
    // are in flight; while unit 1's rows compute, the next step's unit-0 rows are (PMC on the first version, which fetched
    // a step's rows inside the step: SQ_WAIT_ANY = 24 % of the wave's cycles).  Two weight banks, the k loop unrolled by two.
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
    // steps k0 and k0 + 4 with the weights of k0 in wA on entry (and of k0 + 8 on exit), unit-0 rows of k0 in a0 on entry
    // (and of k0 + 8 on exit); `more`: further steps follow
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

    // bias + tanh on the value row, (1 - h^2) scaling on the five tangent rows of the same unit, all in this lane; then
    // the 12 x 8 results go back to the activation buffer as the next layer's operand
    template <bool TANH>
    AC_DI void epilogue_store(f32x2 (&acc)[R][NP], const float* __restrict__ bias) {
        float b[NB];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + NB * tj + 4 * h);
            b[4 * h] = bq[0]; b[4 * h + 1] = bq[1]; b[4 * h + 2] = bq[2]; b[4 * h + 3] = bq[3];
        }
        if constexpr (TANGENT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = acc[6 * u][p][e] + b[2 * p + e];
                        if (TANH) {
                            const float h = act_tanh(v), sp = fmaf(-h, h, 1.0f);
                            acc[6 * u][p][e] = h;
#pragma unroll
                            for (int s = 1; s < 6; ++s) acc[6 * u + s][p][e] *= sp;
                        } else {
                            acc[6 * u][p][e] = v;
                        }
                    }
                }
            }
        } else {  // every row is a value row
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = acc[r][p][e] + b[2 * p + e];
                        acc[r][p][e] = TANH ? act_tanh(v) : v;
                    }
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

    // First layer of the tangent engine in closed form: the operand rows of layer 0 are (z, 0, 0, 0) and the unit vectors, so
    // the value row is five FMAs per neuron and tangent row s is row s - 1 of W0 (times act'(h) in the epilogue) — no operand
    // rows written, no generic K = 8 tile (it took 9.5 % of the wave for 2 % of the FMAs).  Same products in the same order
    // as the tile formed them.  A lane's two units are ti and ti + 8; their z sit on the lanes whose column is the unit.
    AC_DI void first_layer_direct(const float z[5]) {
        static_assert(TANGENT, "value + five tangent rows per unit");
        float zu[2][5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { zu[0][k] = __shfl(z[k], ti, 64); zu[1][k] = __shfl(z[k], ti + 8, 64); }
        const float* w = wimg + plan.w_off[0] + NB * tj;
        float wr[5][NB];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(w + k * WIDTH + 4 * h);
                wr[k][4 * h] = q[0]; wr[k][4 * h + 1] = q[1]; wr[k][4 * h + 2] = q[2]; wr[k][4 * h + 3] = q[3];
            }
        f32x2 acc[R][NP];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int n = 2 * p + e;
                    float v = zu[u][0] * wr[0][n];
#pragma unroll
                    for (int k = 1; k < 5; ++k) v = fmaf(zu[u][k], wr[k][n], v);
                    acc[6 * u][p][e] = v;
#pragma unroll
                    for (int sl = 1; sl < 6; ++sl) acc[6 * u + sl][p][e] = wr[sl - 1][n];
                }
        epilogue_store<true>(acc, wimg + plan.b_off[0]);
    }

    // Last layer, WIDTH -> 6 (padded 8): lane (i, j) computes output neuron j of its 12 rows; the packed FMA runs over
    // k-pairs (even / odd partial sums) against the transposed weights Wt[j][k].
    AC_DI void last_layer(int l) {
        const float* wt = wimg + plan.w_off[l] + tj * (WIDTH + 4);  // rows padded by 4: the eight lane columns read eight distinct bank groups
        f32x2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x2{0.f, 0.f};
        // 8-deep k-steps, software-pipelined like the dense layers: the 24 row reads and 2 weight reads of step k0 + 8 are in
        // flight while step k0 computes (stamps of the first version: this layer took 16 % of the wave for 5 % of the FMAs)
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
        for (int r = 0; r < R; ++r) o[r] = acc[r][0] + acc[r][1];
        if constexpr (TANGENT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float v = o[6 * u] + bj;
                if (plan.act_last) {
                    const float h = act_tanh(v), sp = fmaf(-h, h, 1.0f);
                    o[6 * u] = h;
#pragma unroll
                    for (int s = 1; s < 6; ++s) o[6 * u + s] *= sp;
                } else {
                    o[6 * u] = v;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = o[r] + bj; o[r] = plan.act_last ? act_tanh(v) : v; }
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) act[row_of(r / HR, r % HR) * S + tj] = o[r];
        wave_sync();
    }

    // y[6], J[6][5] of the raw network for normalised inputs z[5]; every lane of a unit (col, g = 0..3) passes the same z
    // and receives the same outputs.  Wave-collective.
    template <int JC> AC_DI void forward(const float z[5], float y[6], float (*J)[JC]) {
        static_assert(!TANGENT || JC >= 5, "J holds the five input tangents");
        AC_MARK(st, 1);
        wave_sync();
#ifndef AC_TILED_GENERIC_FIRST
        if constexpr (TANGENT) {
            AC_MARK(st, 3);
            first_layer_direct(z);
            AC_MARK(st, 2);
        } else
#endif
        {
        if constexpr (TANGENT) {
            // operand rows of layer 0: slab 0 = (z, 0, 0, 0), slab s = unit vector e_{s-1}; lane group g writes slabs 2g, 2g + 1
            if (g < 3) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int s = 2 * g + q;
                    float* dst = act + (col * 6 + s) * S;
                    f32x4 lo, hi;
                    if (s == 0) { lo = f32x4{z[0], z[1], z[2], z[3]}; hi = f32x4{z[4], 0.f, 0.f, 0.f}; }
                    else {
                        lo = f32x4{s == 1 ? 1.f : 0.f, s == 2 ? 1.f : 0.f, s == 3 ? 1.f : 0.f, s == 4 ? 1.f : 0.f};
                        hi = f32x4{s == 5 ? 1.f : 0.f, 0.f, 0.f, 0.f};
                    }
                    *reinterpret_cast<f32x4*>(dst) = lo;
                    *reinterpret_cast<f32x4*>(dst + 4) = hi;
                }
            }
        } else {  // lane = unit: row `lane` = (z, 0, 0, 0)
            float* dst = act + lane * S;
            *reinterpret_cast<f32x4*>(dst) = f32x4{z[0], z[1], z[2], z[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{z[4], 0.f, 0.f, 0.f};
        }
        wave_sync();
        AC_MARK(st, 3);  // [3] operand rows of layer 0 written
        dense_layer<8>(0);
        AC_MARK(st, 2);
        }
#pragma nounroll
        for (int l = 1; l < plan.n_layers - 1; ++l) dense_layer<WIDTH>(l);
        AC_MARK(st, 4);
        last_layer(plan.n_layers - 1);
        AC_MARK(st, 5);
        if constexpr (TANGENT) {
            // outputs: row 6 col + s, columns 0..5
            const float* src = act + col * 6 * S;
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(src), y1 = *reinterpret_cast<const f32x4*>(src + 4);
            y[0] = y0[0]; y[1] = y0[1]; y[2] = y0[2]; y[3] = y0[3]; y[4] = y1[0]; y[5] = y1[1];
#pragma unroll
            for (int s = 1; s < 6; ++s) {
                const f32x4 j0 = *reinterpret_cast<const f32x4*>(src + s * S), j1 = *reinterpret_cast<const f32x4*>(src + s * S + 4);
                J[0][s - 1] = j0[0]; J[1][s - 1] = j0[1]; J[2][s - 1] = j0[2]; J[3][s - 1] = j0[3];
                J[4][s - 1] = j1[0]; J[5][s - 1] = j1[1];
            }
        } else {
            (void)J;
            const float* src = act + lane * S;
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(src), y1 = *reinterpret_cast<const f32x4*>(src + 4);
            y[0] = y0[0]; y[1] = y0[1]; y[2] = y0[2]; y[3] = y0[3]; y[4] = y1[0]; y[5] = y1[1];
        }
        wave_sync();
        AC_MARK(st, 6);
    }
};

// The fused step + sensitivities kernel on the vector ALUs: k_nn_step_sens with the tiled engine.
template <int WIDTH>
__global__ __launch_bounds__(kBlock, 1) void k_nn_step_sens_tiled(const DevParams P, const ValuPlan plan,
                                                                  const float* __restrict__ blob,
                                                                  const float* __restrict__ X, const float* __restrict__ U,
                                                                  float dt, const float* __restrict__ dt_per_unit, long n,
                                                                  long blk, float* __restrict__ Xn, float* __restrict__ A,
                                                                  float* __restrict__ Bm, float* __restrict__ c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngineTiled<WIDTH> eng(plan, blob, smem);
    eng.st.start();
    eng.load_weights();
    AC_MARK(eng.st, 0);  // [0] prologue: weight image into LDS
#ifdef AC_STAMPS
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(c);  // the diagnostic flavor's stamp buffer travels in `c`
    c = nullptr;
#endif
    const WaveUnit w(n, blk);
    float xv[13], uv[7];
    load_rows<13>(X, w.ua, xv);
    load_rows<7>(U, w.ua, uv);
    const float hv = dt_per_unit ? dt_per_unit[w.unit] : dt;
    Dual<4> x[13];
    MlpCoeffs<MlpEngineTiled<WIDTH>> coeffs(eng);
    sens_update<4>(P, coeffs, w.g, w.col, w.ua, xv, uv, hv, x, A, Bm, c, w.live);
    AC_MARK(eng.st, 7);  // [7] dual rigid body + RK4 combine after the last network evaluation
#ifdef AC_STAMPS
    eng.st.flush(stamp_buf);
#endif
    if (w.live) {
        const UnitAddr uo = w.ua.late();
        if (w.g == 0) {
            float* p = Xn + uo.off(13);
#pragma unroll
            for (int i = 0; i < 13; ++i) p[(long)i * blk] = x[i].v;
        }
        SensIO::store(w.g, uo, x, A, Bm, c, true);
    }
}

template <int WIDTH>
__global__ __launch_bounds__(kBlock, 1) void k_nn_deriv_sens_tiled(const DevParams P, const ValuPlan plan,
                                                                   const float* __restrict__ blob,
                                                                   const float* __restrict__ X, const float* __restrict__ U,
                                                                   long n, long blk, float* __restrict__ Xdot,
                                                                   float* __restrict__ Fx, float* __restrict__ Fu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngineTiled<WIDTH> eng(plan, blob, smem);
    eng.load_weights();
    const WaveUnit w(n, blk);
    float xv[13], uv[7];
    load_rows<13>(X, w.ua, xv);
    load_rows<7>(U, w.ua, uv);
    Dual<4> k[13];
    MlpCoeffs<MlpEngineTiled<WIDTH>> coeffs(eng);
    deriv_seeded<4>(P, coeffs, w.g, xv, uv, k);
    if (w.live) deriv_store<4, false>(w.g, w.ua, k, Xdot, Fx, Fu);
}

// ---- forward kernels on the vector ALUs: 64 units per wave, lane = unit (the value-only tile) ---------------------------
template <int WIDTH, int OP>
__global__ __launch_bounds__(kBlock, 1) void k_nn_fwd_tiled(const DevParams P, const ValuPlan plan,
                                                            const float* __restrict__ blob, const float* __restrict__ X,
                                                            const float* __restrict__ U, float dt,
                                                            const float* __restrict__ dt_per_unit, long n, long blk,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngineTiled<WIDTH, false> Engine;
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
        if (live) store_rows<13>(out, ua.late(), xd);
    } else if constexpr (OP == OP_STEP) {
        const float h = dt_per_unit ? dt_per_unit[unit] : dt;
        state_update(P, coeffs, x, u, h);
        if (live) store_rows<13>(out, ua.late(), x);
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
    typedef MlpEngineTiled<WIDTH, false> Engine;
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

}  // namespace ac

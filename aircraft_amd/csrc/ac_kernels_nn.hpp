// ac_kernels_nn.hpp — kernels for the MLP surrogate ("nn") model, built on the wave-level engine of
// ac_mlp.hpp.  One wave = 16 units (col = lane & 15), four lanes per unit (g = lane >> 4).
//   * sensitivities: the four lanes carry four tangent directions each (Dual<4>), the engine runs
//     6 slabs (value + 5 input tangents) -> the fused step+Jacobian kernel, the headline hot path.
//   * forward (derivative / step / rollout / aero): the four lanes of a unit compute the same values
//     (the rigid-body part is <3 % of the work); the engine runs the value slab only.
#pragma once
#include "ac_kernels_analytic.hpp"
#include "ac_mlp.hpp"
#include "ac_ilqr.hpp"

namespace ac {

enum FwdOp { OP_DERIV = 0, OP_STEP = 1, OP_AERO = 2 };

struct WaveUnit {
    int lane, col, g;
    long unit;
    bool live;
    UnitAddr ua;
    AC_DI static long raw_unit(long unit0) {
        return unit0 + ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (threadIdx.x & 15);
    }
    // units [unit0, n) of the batch; dead lanes shadow the last unit so wave/workgroup collectives stay uniform
    AC_DI WaveUnit(long n, long blk, long unit0 = 0)
        : lane(threadIdx.x & 63), col(threadIdx.x & 15), g((threadIdx.x & 63) >> 4),
          unit(raw_unit(unit0) < n ? raw_unit(unit0) : n - 1), live(raw_unit(unit0) < n), ua(unit, blk) {}
    // task `task` (64 units: one workgroup's share) of a persistent grid; `tid` is the caller's laundered threadIdx.x
    AC_DI static long task_unit(int tid, long task) { return task * 64 + (tid >> 6) * 16 + (tid & 15); }
    AC_DI WaveUnit(int tid, long task, long n, long blk)
        : lane(tid & 63), col(tid & 15), g((tid & 63) >> 4),
          unit(task_unit(tid, task) < n ? task_unit(tid, task) : n - 1), live(task_unit(tid, task) < n), ua(unit, blk) {}
};

template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_step_sens(const DevParams P, const MlpPlan plan,
                                                            const float* __restrict__ blob,
                                                            const float* __restrict__ X, const float* __restrict__ U,
                                                            float dt, const float* __restrict__ dt_per_unit, long n, long blk,
                                                            float* __restrict__ Xn, float* __restrict__ A,
                                                            float* __restrict__ Bm, float* __restrict__ c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngine<6, WT, USE_MFMA> eng(plan, blob, smem);
    eng.st.start();
    WaveClock wc;
    wc.start();
    eng.load_weights();
    AC_MARK(eng.st, 0);  // [0] prologue: weights into LDS
#if defined(AC_STAMPS) || defined(AC_CLOCKS)
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(c);
    c = nullptr;
#endif
    MlpCoeffs<MlpEngine<6, WT, USE_MFMA>> coeffs(eng);
    // PERSISTENT workgroups: the grid is at most one workgroup per CU and workgroup b runs the tasks (64 units each)
    // b, b + gridDim.x, ...  With one 138 KB workgroup resident per CU, a workgroup per task had every CU idle between the
    // last wave of one task and the first of the next (dispatch + the weight prologue: -DAC_CLOCKS measured a wave lifetime
    // of 303 us inside rounds of 320); here the next task's input loads follow the previous task's stores directly and the
    // weights are loaded once.  The four waves stay in step through the barriers of the streamed layers, every wave runs the
    // same number of tasks, and the loop bounds are workgroup-uniform: the grid drains by itself.
    const long ntasks = (n + 63) / 64;
#pragma nounroll
    for (long task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // lane-dependent values are rebuilt per task, not hoisted into registers the body needs
        const WaveUnit w(tid, task, n, blk);
        float xv[13], uv[7];
        load_rows<13>(X, w.ua, xv);
        load_rows<7>(U, w.ua, uv);
        const float hv = dt_per_unit ? dt_per_unit[w.unit] : dt;
        Dual<4> x[13];
        sens_update<4>(P, coeffs, w.g, w.col, w.ua, xv, uv, hv, x, A, Bm, c, w.live);
        AC_MARK(eng.st, 7);  // [7] dual aero + rigid body + RK4 combine (everything outside forward())
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
    eng.drain();
    AC_MARK(eng.st, 8);  // [8] stores
#ifdef AC_STAMPS
    eng.st.flush(stamp_buf);
#endif
#ifdef AC_CLOCKS
    wc.stop(stamp_buf);
#endif
}

// The same for SMALL nets (WT = 2: hidden widths <= 32, e.g. the reference's own 5-16-32-6 checkpoint) at TWO waves per
// SIMD.  With so little matrix work the step is bound by vector-ALU issue of the dual rigid-body arithmetic
// (profiles/r04_analytic_pmc_before.json, "real": 80 % of the wave cycles in VALU instructions at one wave per SIMD, where
// a wave issues one every 4 cycles), so the lever is a second wave on every SIMD: fused per-direction tangents and the RK4
// sum in LDS bring the kernel under 256 registers, and the grid is two persistent workgroups per CU.
template <int WT>
__global__ __launch_bounds__(kBlock, 2) void k_nn_step_sens_w2(const DevParams P, const MlpPlan plan,
                                                               const float* __restrict__ blob,
                                                               const float* __restrict__ X, const float* __restrict__ U,
                                                               float dt, const float* __restrict__ dt_per_unit, long n, long blk,
                                                               float* __restrict__ Xn, float* __restrict__ A,
                                                               float* __restrict__ Bm, float* __restrict__ c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngine<6, WT, true> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    MlpCoeffs<Engine, true> coeffs(eng);
    float4* acc_words = reinterpret_cast<float4*>(smem + ((plan.lds_total + 15) & ~15));
    const long ntasks = (n + 63) / 64;
#pragma nounroll
    for (long task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const WaveUnit w(tid, task, n, blk);
        float xv[13], uv[7];
        load_rows<13>(X, w.ua, xv);
        load_rows<7>(U, w.ua, uv);
        const float hv = dt_per_unit ? dt_per_unit[w.unit] : dt;
        Dual<4> x[13];
        LdsAcc4 acc(&acc_words[tid], kBlock);
        sens_update<4, 1>(P, coeffs, w.g, w.col, w.ua, xv, uv, hv, x, A, Bm, c, w.live, acc);
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
    eng.drain();
}
constexpr int kSensW2AccBytes = 13 * kBlock * 16;

// x_dot = f(x, u) with Fx = df/dx, Fu = df/du through the surrogate: ONE evaluation of the 6-slab engine (value + five
// input tangents) and the dual rigid-body arithmetic on the same lanes — the first stage of k_nn_step_sens on its own.
template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_deriv_sens(const DevParams P, const MlpPlan plan,
                                                             const float* __restrict__ blob,
                                                             const float* __restrict__ X, const float* __restrict__ U,
                                                             long n, long blk, float* __restrict__ Xdot,
                                                             float* __restrict__ Fx, float* __restrict__ Fu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngine<6, WT, USE_MFMA> eng(plan, blob, smem);
    eng.load_weights();
    const WaveUnit w(n, blk);
    float xv[13], uv[7];
    load_rows<13>(X, w.ua, xv);
    load_rows<7>(U, w.ua, uv);
    Dual<4> k[13];
    MlpCoeffs<MlpEngine<6, WT, USE_MFMA>> coeffs(eng);
    deriv_seeded<4>(P, coeffs, w.g, xv, uv, k);
    eng.drain();
    if (w.live) deriv_store<4, false>(P, X, w.unit, w.g, w.ua, k, Xdot, Fx, Fu);
}

// The sensitivity step for a REMAINDER of units that would leave the last round of the grid half empty: a wave's time
// is set by its slab count, not by its live columns, so a unit group (16 units) is given to a PAIR of waves — one
// carries the value slab and tangents 0-1, the other tangents 2-4 and NO value slab: it takes the value activations of
// every hidden layer from its partner through LDS (MlpEngine PAIR roles) — and a workgroup holds two groups (32 units).
// Both waves run the whole dual RK4 arithmetic on identical inputs (so the pair's results are bit-identical to
// k_nn_step_sens'); they exchange y and their Jacobian columns through LDS after every network evaluation; the first
// wave of the pair stores.  Twice the workgroups, each a little over half the time of a full one.
template <class Engine, int TOFF>
AC_DI void sens_pair_body(const DevParams& P, const MlpPlan& plan, const float* __restrict__ blob, char* smem,
                          float* xch, bool store, const float* __restrict__ X, const float* __restrict__ U, float dt,
                          const float* __restrict__ dt_per_unit, long n, long blk, long unit0, int pair,
                          float* __restrict__ Xn, float* __restrict__ A, float* __restrict__ Bm,
                          float* __restrict__ c) {
    Engine eng(plan, blob, smem);
    eng.hx = reinterpret_cast<f32x4*>(smem + plan.lds_total) + pair * (Engine::kWT * 64);
    eng.load_weights();
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
    const long raw = unit0 + ((long)blockIdx.x * 2 + pair) * 16 + col;
    const bool live = raw < n;
    const long unit = live ? raw : n - 1;
    const UnitAddr ua(unit, blk);
    float xv[13], uv[7];
    load_rows<13>(X, ua, xv);
    load_rows<7>(U, ua, uv);
    const float hv = dt_per_unit ? dt_per_unit[unit] : dt;
    // The sixteen tangent directions of a unit are split over the pair as well: eight lanes per unit (four in each wave), two
    // directions per lane — each wave runs the dual RK4 arithmetic on Dual<2> instead of both on Dual<4> (the value parts are
    // computed twice, the derivative parts once).  Lane-local for one RK4 sub-step; the dispatcher keeps sub-stepped updates
    // (whose composition shuffles across the lanes of a unit inside one wave) on the one-wave kernel.
    (void)store;
    const int g8 = 4 * (Engine::kNoValue ? 1 : 0) + g;
    Dual<2> x[13];
    MlpPairCoeffs<Engine, TOFF> coeffs(eng, xch);
    sens_update<2>(P, coeffs, g8, col, ua, xv, uv, hv, x, A, Bm, c, live);
    eng.drain();
    if (live) {
        const UnitAddr uo = ua.late();
        if (g8 == 0) {
            float* p = Xn + uo.off(13);
#pragma unroll
            for (int i = 0; i < 13; ++i) p[(long)i * blk] = x[i].v;
        }
        SensIOT<2>::store(g8, uo, x, A, Bm, c, true);
    }
}

template <int WT>
__global__ __launch_bounds__(kBlock, 1) void k_nn_step_sens_pair(const DevParams P, const MlpPlan plan,
                                                                 const float* __restrict__ blob,
                                                                 const float* __restrict__ X, const float* __restrict__ U,
                                                                 float dt, const float* __restrict__ dt_per_unit, long n,
                                                                 long blk, float* __restrict__ Xn, float* __restrict__ A,
                                                                 float* __restrict__ Bm, float* __restrict__ c,
                                                                 long unit0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, role = wave & 1, pair = wave >> 1;
    // behind the plan's image: [2 pairs][WT KiB] value activations, then [2 pairs][16 units][36] outputs
    float* xch = reinterpret_cast<float*>(smem + plan.lds_total + 2 * WT * 1024) + pair * (16 * 36);
    if (role == 0)
        sens_pair_body<MlpEngine<3, WT, true, true, false, 0, 1>, 0>(P, plan, blob, smem, xch, true, X, U, dt, dt_per_unit, n,
                                                                      blk, unit0, pair, Xn, A, Bm, c);
    else
        sens_pair_body<MlpEngine<3, WT, true, true, false, 2, 2>, 2>(P, plan, blob, smem, xch, false, X, U, dt, dt_per_unit, n,
                                                                      blk, unit0, pair, Xn, A, Bm, c);
}

template <int WT, bool USE_MFMA, int OP>
__global__ __launch_bounds__(kBlock, 1) void k_nn_fwd(const DevParams P, const MlpPlan plan,
                                                      const float* __restrict__ blob, const float* __restrict__ X,
                                                      const float* __restrict__ U, float dt,
                                                      const float* __restrict__ dt_per_unit, long n, long blk,
                                                      float* __restrict__ out, long unit0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngine<1, WT, USE_MFMA> eng(plan, blob, smem);
    eng.load_weights();
    const WaveUnit w(n, blk, unit0);  // this launch covers units [unit0, n)
    float x[13], u[7];
    load_rows<13>(X, w.ua, x);
    load_rows<7>(U, w.ua, u);
    MlpCoeffs<MlpEngine<1, WT, USE_MFMA>> coeffs(eng);
    if constexpr (OP == OP_DERIV) {
        float xd[13];
        coeffs.prefetch(P, x, u);
    state_derivative<float>(P, coeffs, x, u, xd);
        eng.drain();
        if (w.live && w.g == 0) store_state_rows(P, X, out, w.ua, w.unit, xd);
    } else if constexpr (OP == OP_STEP) {
        const float h = dt_per_unit ? dt_per_unit[w.unit] : dt;
        state_update(P, coeffs, x, u, h);
        eng.drain();
        if (w.live && w.g == 0) store_state_rows(P, X, out, w.ua, w.unit, x);
    } else {
        coeffs.prefetch(P, x, u);
        AeroPre<float> a;
        aero_pre(P, x, a);
        float C[6];
        coeffs(P, a, x, u, C);
        AeroPost<float> o;
        aero_post(P, a, u, C, o);
        eng.drain();
        float eu[3];
        euler_angles(x, eu[0], eu[1], eu[2]);
        const float v[22] = {a.vr[0], a.vr[1], a.vr[2], a.V, a.alpha, a.beta, a.qbar, o.C[0], o.C[1], o.C[2],
                             o.C[3], o.C[4], o.C[5], o.F[0], o.F[1], o.F[2], o.M[0], o.M[1], o.M[2], eu[0], eu[1], eu[2]};
        if (w.live && w.g == 0) store_rows<22>(out, w.ua, v);
    }
}

// Forward kernels, 64 units per wave: four value slabs share every weight fragment and LDS-DMA piece, and each slab's
// tanh epilogue issues inside the next slab's MFMA stream (MFMA path only; lane = unit).
template <int WT, int OP>
__global__ __launch_bounds__(kBlock, 1) void k_nn_fwd4(const DevParams P, const MlpPlan plan,
                                                       const float* __restrict__ blob, const float* __restrict__ X,
                                                       const float* __restrict__ U, float dt,
                                                       const float* __restrict__ dt_per_unit, long n, long blk,
                                                       float* __restrict__ out, long unit0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MlpEngine<4, WT, true, false> Engine;
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const long raw = unit0 + (long)blockIdx.x * kBlock + threadIdx.x;  // this launch covers units [unit0, n)
    const bool live = raw < n;
    const long unit = live ? raw : n - 1;  // dead lanes shadow the last unit: the engine is wave/workgroup-collective
    const UnitAddr ua(unit, blk);
    float x[13], u[7];
    load_rows<13>(X, ua, x);
    load_rows<7>(U, ua, u);
    MlpCoeffs<Engine> coeffs(eng);
    if constexpr (OP == OP_DERIV) {
        float xd[13];
        coeffs.prefetch(P, x, u);
        state_derivative<float>(P, coeffs, x, u, xd);
        eng.drain();
        if (live) store_state_rows(P, X, out, ua.late(), unit, xd);
    } else if constexpr (OP == OP_STEP) {
        const float h = dt_per_unit ? dt_per_unit[unit] : dt;
        state_update(P, coeffs, x, u, h);
        eng.drain();
        if (live) store_state_rows(P, X, out, ua.late(), unit, x);
    } else {
        coeffs.prefetch(P, x, u);
        AeroPre<float> a;
        aero_pre(P, x, a);
        float C[6];
        coeffs(P, a, x, u, C);
        AeroPost<float> o;
        aero_post(P, a, u, C, o);
        eng.drain();
        float eu[3];
        euler_angles(x, eu[0], eu[1], eu[2]);
        const float v[22] = {a.vr[0], a.vr[1], a.vr[2], a.V, a.alpha, a.beta, a.qbar, o.C[0], o.C[1], o.C[2],
                             o.C[3], o.C[4], o.C[5], o.F[0], o.F[1], o.F[2], o.M[0], o.M[1], o.M[2], eu[0], eu[1], eu[2]};
        if (live) store_rows<22>(out, ua.late(), v);
    }
}

// Sequential rollout, one wave per 16 instances (launched with 4-wave workgroups).  Used for very large batches
// and for the VALU validation path; moderate batches take k_nn_rollout_coop below.
template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout(const DevParams P, const MlpPlan plan,
                                                      const float* __restrict__ blob, const float* __restrict__ X0,
                                                      const float* __restrict__ U, float dt, long B, long H,
                                                      float* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MlpEngine<1, WT, USE_MFMA> eng(plan, blob, smem);
    eng.load_weights();
    const WaveUnit w(B, B);
    float x[13], u[7], un[7];
    load_rows<13>(X0, B, w.unit, x);
    const bool writer = w.live && w.g == 0;
    double xa[13];  // float64 carry of the state across the horizon (see state_update_carry)
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (writer) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * B + w.unit] = x[r];
    }
    if (H > 0) load_rows<7>(U, B, w.unit, u);
    MlpCoeffs<MlpEngine<1, WT, USE_MFMA>> coeffs(eng);
    for (long k = 0; k < H; ++k) {
        if (k + 1 < H) load_rows<7>(U + (k + 1) * 7 * B, B, w.unit, un);
        state_update_carry(P, coeffs, xa, u, dt);
        if (writer) {
            float* o = Xout + (k + 1) * 13 * B;
#pragma unroll
            for (int r = 0; r < 13; ++r) o[(long)r * B + w.unit] = (float)xa[r];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) u[r] = un[r];
    }
    eng.drain();
}

// Cooperative rollout: one 4-wave workgroup per 16 instances (see MlpEngineCoop).  Every wave integrates the same
// 16 instances (the rigid-body part is tiny); wave 0 writes the trajectory.
template <class Engine>
AC_DI void rollout_coop_body(const DevParams& P, const MlpPlan& plan, const float* __restrict__ blob,
                             const float* __restrict__ X0, const float* __restrict__ U, float dt, long B, long H,
                             float* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const long raw = (long)blockIdx.x * 16 + col;
    const bool live = raw < B;
    const long unit = live ? raw : B - 1;
    float x[13], u[7], un[7];
    load_rows<13>(X0, B, unit, x);
    const bool writer = live && g == 0 && wave == 0;
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (writer) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * B + unit] = x[r];
    }
    if (H > 0) load_rows<7>(U, B, unit, u);
    MlpCoeffs<Engine> coeffs(eng);
    for (long k = 0; k < H; ++k) {
        if (k + 1 < H) load_rows<7>(U + (k + 1) * 7 * B, B, unit, un);
        state_update_carry(P, coeffs, xa, u, dt);
        if (writer) {
            float* o = Xout + (k + 1) * 13 * B;
#pragma unroll
            for (int r = 0; r < 13; ++r) o[(long)r * B + unit] = (float)xa[r];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) u[r] = un[r];
    }
    eng.drain();
}

template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout_coop(const DevParams P, const MlpPlan plan,
                                                               const float* __restrict__ blob,
                                                               const float* __restrict__ X0,
                                                               const float* __restrict__ U, float dt, long B, long H,
                                                               float* __restrict__ Xout) {
    rollout_coop_body<MlpEngineCoop<WT, USE_MFMA>>(P, plan, blob, X0, U, dt, B, H, Xout);
}

// Same, weights resident in registers for the whole horizon (NH hidden layers; see MlpEngineCoopReg).
template <int WT, int NH>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout_reg(const DevParams P, const MlpPlan plan,
                                                              const float* __restrict__ blob,
                                                              const float* __restrict__ X0,
                                                              const float* __restrict__ U, float dt, long B, long H,
                                                              float* __restrict__ Xout) {
    rollout_coop_body<MlpEngineCoopReg<WT, NH, true>>(P, plan, blob, X0, U, dt, B, H, Xout);
}

// Closed-loop (feedback policy) rollout through the MLP surrogate: the cooperative engine, with the control of every
// node computed from the iLQR gains.  Output instance o = a * B + b (a = line-search index).
template <class Engine>
AC_DI void rollout_policy_body(const DevParams& P, const MlpPlan& plan, const float* __restrict__ blob,
                               const Policy& pol, const float* __restrict__ X0, float dt, long Bout, long H,
                               float* __restrict__ Xout, float* __restrict__ Uout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Engine eng(plan, blob, smem);
    eng.load_weights();
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const long raw = (long)blockIdx.x * 16 + col;
    const bool live = raw < Bout;
    const long o = live ? raw : Bout - 1;
    float x[13], u[7];
    load_rows<13>(X0, pol.B, o % pol.B, x);
    const bool writer = live && g == 0 && wave == 0;
    double xa[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) xa[r] = (double)x[r];
    if (writer) {
#pragma unroll
        for (int r = 0; r < 13; ++r) Xout[(long)r * Bout + o] = x[r];
    }
    MlpCoeffs<Engine> coeffs(eng);
    Policy::Quarter q, qn;  // this lane group's quarter of the gains: node k, and node k+1 prefetched during the step
    if (H > 0) pol.load(0, o, g, q);
    for (long k = 0; k < H; ++k) {
#pragma unroll
        for (int r = 0; r < 13; ++r) x[r] = (float)xa[r];
        pol.control(q, g, x, u);
        if (k + 1 < H) pol.load(k + 1, o, g, qn);
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
        q = qn;
    }
    eng.drain();
}

template <int WT, bool USE_MFMA>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout_policy_coop(const DevParams P, const MlpPlan plan,
                                                                      const float* __restrict__ blob, const Policy pol,
                                                                      const float* __restrict__ X0, float dt,
                                                                      long Bout, long H, float* __restrict__ Xout,
                                                                      float* __restrict__ Uout) {
    rollout_policy_body<MlpEngineCoop<WT, USE_MFMA>>(P, plan, blob, pol, X0, dt, Bout, H, Xout, Uout);
}

template <int WT, int NH>
__global__ __launch_bounds__(kBlock, 1) void k_nn_rollout_policy_reg(const DevParams P, const MlpPlan plan,
                                                                     const float* __restrict__ blob, const Policy pol,
                                                                     const float* __restrict__ X0, float dt,
                                                                     long Bout, long H, float* __restrict__ Xout,
                                                                     float* __restrict__ Uout) {
    rollout_policy_body<MlpEngineCoopReg<WT, NH, true>>(P, plan, blob, pol, X0, dt, Bout, H, Xout, Uout);
}

}  // namespace ac

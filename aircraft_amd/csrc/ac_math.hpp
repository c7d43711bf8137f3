// ac_math.hpp — scalar / forward-mode-dual arithmetic used by every kernel (gfx950, fp32).
//
// The rigid-body code in ac_dynamics.hpp is templated on the scalar type T:
//   T = float      : values only (forward step / rollout)
//   T = Dual<N>    : value + N tangent directions (step sensitivities; N = 4, four lanes per unit)
// Tangent directions are independent given the primal, so a unit's 15 non-trivial directions
// (v, q, omega, 4 active controls, dt) are spread over 4 lanes x Dual<4>.
#pragma once
// -DAC_HOST_CHECK (tests/host_dyn, g++): the arithmetic headers (this one and ac_dynamics.hpp) compile as plain host C++
// so that the CPU test suite can run the kernels' own math against the float64 oracle without a GPU.  Never defined in
// the product build.
#ifdef AC_HOST_CHECK
#include <cmath>
#define AC_DI inline
#define AC_OPAQUE_V(x) ((void)0)
#define AC_OPAQUE_S(x) ((void)0)
#define AC_SCHED_FENCE() ((void)0)
#define AC_CONSTANT
#define AC_RCP(x) (1.0f / (x))
#define AC_RSQ(x) (1.0f / sqrtf(x))
#else
// 1-ulp reciprocal / reciprocal square root (v_rcp_f32, v_rsq_f32: one instruction where an IEEE division or square root
// is about ten).  ONLY for coefficients of tangents (Jacobian entries, bar 1e-5): every primal value keeps the forward
// kernels' IEEE expressions.
#define AC_RCP(x) __builtin_amdgcn_rcpf(x)
#define AC_RSQ(x) __builtin_amdgcn_rsqf(x)
// the same for a wave-uniform value (a pointer the loads behind it must not be hoisted past)
#define AC_OPAQUE_S(x) asm volatile("" : "+s"(x))
// constant address space: a uniform address in it is always read with scalar loads
#define AC_CONSTANT __attribute__((address_space(4)))
// the instruction scheduler moves nothing across this point (keeps a register-hungry primal phase from being interleaved
// with the tangent phase that follows it)
#define AC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#include <hip/hip_runtime.h>
#define AC_DI __device__ __forceinline__
// a copy of a per-lane value the optimiser cannot see through (keeps rebuilt 0/1 seed patterns out of long-lived registers)
#define AC_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

namespace ac {

template <int N>
struct Dual {
    float v;
    float d[N];
    AC_DI Dual() {}
    AC_DI Dual(float x) : v(x) {  // NOLINT (implicit on purpose)
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = 0.f;
    }
};

// ---- in-kernel phase stamps: DIAGNOSTIC build flavor only (-DAC_STAMPS; aircraft_amd/build.py --diag) -----------
// In the product build AC_MARK() expands to nothing.  In the diagnostic flavor every wave accumulates the shader-clock
// time between consecutive marks into per-phase sums (wave-uniform, SGPRs) and adds them to a global buffer at the end.
// The buffer travels through the kernel's optional `c` (dF/ddt) pointer, which the diagnostic kernel does not write.
#ifdef AC_STAMPS
struct Stamper {
    unsigned long long last;
    unsigned long long acc[12];
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0;
        last = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int id) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the time value has arrived
        acc[id] += t - last;
        last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void flush(unsigned long long* buf) {
        if ((threadIdx.x & 63) == 0 && buf) {
#pragma unroll
            for (int i = 0; i < 12; ++i) atomicAdd(&buf[i], acc[i]);
            atomicAdd(&buf[12], 1ull);
        }
    }
};
#define AC_MARK(st, id) (st).mark(id)
#else
struct Stamper {
    AC_DI void start() {}
    AC_DI void mark(int) {}
    AC_DI void flush(unsigned long long*) {}
};
#define AC_MARK(st, id) ((void)0)
#endif

// -DAC_CLOCKS (a measurement flavour, tools/diag_clock_ratio.py): every wave adds its lifetime in shader-clock cycles
// (s_memtime) and in constant 100 MHz ticks (s_memrealtime) to buf[0], buf[1] and counts itself in buf[2] — the clock the
// chip really holds under THIS kernel is 100 MHz x buf[0] / buf[1] (power management lowers it under heavy vector load).
struct WaveClock {
#ifdef AC_CLOCKS
    unsigned long long t0, r0;
    __device__ __forceinline__ void start() { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void stop(unsigned long long* buf) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && buf) {
            atomicAdd(&buf[0], t1 - t0); atomicAdd(&buf[1], r1 - r0); atomicAdd(&buf[2], 1ull);
            atomicMax(&buf[3], r1 - r0);          // longest wave lifetime (100 MHz ticks)
            atomicMin(&buf[4], r0); atomicMax(&buf[5], r1);  // first start / last end of any wave since the buffer was reset
            unsigned xcc;                          // per-XCD lifetime sums: does one die run slower than the others?
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicAdd(&buf[8 + (xcc & 7)], r1 - r0);
        }
    }
#else
    AC_DI void start() {}
    AC_DI void stop(unsigned long long*) {}
#endif
};

template <int N> AC_DI Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N> AC_DI Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N> AC_DI Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
template <int N> AC_DI Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = fmaf(a.d[i], b.v, a.v * b.d[i]);
    return r;
}
// fused forms of the RK4 combinations (one FMA per tangent where the operator forms take a multiply and an add)
template <int N> AC_DI Dual<N> dual_axpy(float a, const Dual<N>& x, const Dual<N>& y) {  // a x + y
    Dual<N> r; r.v = fmaf(a, x.v, y.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = fmaf(a, x.d[i], y.d[i]);
    return r;
}
template <int N> AC_DI Dual<N> dual_mul_add(const Dual<N>& a, const Dual<N>& x, const Dual<N>& y) {  // a x + y
    Dual<N> r; r.v = fmaf(a.v, x.v, y.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = fmaf(a.v, x.d[i], fmaf(a.d[i], x.v, y.d[i]));
    return r;
}
template <int N> AC_DI Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; const float inv = 1.0f / b.v; r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N> AC_DI Dual<N> operator+(const Dual<N>& a, float b) { Dual<N> r = a; r.v += b; return r; }
template <int N> AC_DI Dual<N> operator+(float b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N> AC_DI Dual<N> operator-(const Dual<N>& a, float b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> AC_DI Dual<N> operator-(float b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> AC_DI Dual<N> operator*(const Dual<N>& a, float b) {
    Dual<N> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int N> AC_DI Dual<N> operator*(float b, const Dual<N>& a) { return a * b; }
template <int N> AC_DI Dual<N> operator/(const Dual<N>& a, float b) { return a * (1.0f / b); }
template <int N> AC_DI Dual<N> operator/(float a, const Dual<N>& b) {
    Dual<N> r; const float inv = 1.0f / b.v; r.v = a * inv; const float g = -r.v * inv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = g * b.d[i];
    return r;
}

AC_DI float value_of(float x) { return x; }
template <int N> AC_DI float value_of(const Dual<N>& x) { return x.v; }

AC_DI float m_sqrt(float x) { return sqrtf(x); }
template <int N> AC_DI Dual<N> m_sqrt(const Dual<N>& a) {
    Dual<N> r; r.v = sqrtf(a.v); const float g = 0.5f / r.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r;
}
AC_DI float m_atan2(float y, float x) { return atan2f(y, x); }
template <int N> AC_DI Dual<N> m_atan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; r.v = atan2f(y.v, x.v); const float den = 1.0f / fmaf(x.v, x.v, y.v * y.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
    return r;
}
AC_DI float m_asin(float x) { return asinf(x); }
template <int N> AC_DI Dual<N> m_asin(const Dual<N>& a) {
    Dual<N> r; r.v = asinf(a.v); const float g = 1.0f / sqrtf(fmaf(-a.v, a.v, 1.0f));
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
    return r;
}
AC_DI float m_exp(float x) { return expf(x); }
template <int N> AC_DI Dual<N> m_exp(const Dual<N>& a) {
    Dual<N> r; r.v = expf(a.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
    return r;
}
AC_DI float m_fabs(float x) { return fabsf(x); }
template <int N> AC_DI Dual<N> m_fabs(const Dual<N>& a) { return a.v < 0.f ? -a : a; }
// casadi sign(): -1 / 0 / +1 (NaN propagates); derivative identically zero
AC_DI float sign_of(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : (x == 0.f ? 0.f : x)); }

// xyzw quaternion, Hamilton product (liecasadi convention fixed by the simulation.h5 replay)
template <class T> struct Q4 { T x, y, z, w; };

template <class T> AC_DI Q4<T> qmul(const Q4<T>& a, const Q4<T>& b) {
    Q4<T> r;
    r.x = a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y);
    r.y = a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z);
    r.z = a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x);
    r.w = a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z);
    return r;
}
// q (x) (v, 0): the pure-vector right factor saves the w-terms
template <class T> AC_DI Q4<T> qmul_vec(const Q4<T>& a, const T& bx, const T& by, const T& bz) {
    Q4<T> r;
    r.x = a.w * bx + (a.y * bz - a.z * by);
    r.y = a.w * by + (a.z * bx - a.x * bz);
    r.z = a.w * bz + (a.x * by - a.y * bx);
    r.w = -(a.x * bx + a.y * by + a.z * bz);
    return r;
}
// inverse = conjugate / |q|^2   (dynamics/base.py:158 via liecasadi Quaternion.inverse)
template <class T> AC_DI Q4<T> qinv(const Q4<T>& q) {
    const T n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const T inv = 1.0f / n2;
    Q4<T> r; r.x = -(q.x * inv); r.y = -(q.y * inv); r.z = -(q.z * inv); r.w = q.w * inv;
    return r;
}

}  // namespace ac

// Microbenchmark: the issue rate of v_pk_fma_f32 (and v_fma_f32) from registers alone, 1 and 2 waves per SIMD — the
// practical ceiling behind the "157.3 TFLOP/s fp32 vector" figure the K4 kernel is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/micro/valu_pkfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int THREADS> __global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    f32x2 acc[8], a = {1.0001f, 0.9999f}, w = {0.5f + threadIdx.x * 1e-6f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{0.f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(w));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(w));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(w.x));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[(long)blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int KIND, int THREADS> void run(float* out, const char* name) {
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * (THREADS / 64) * iters * 128;       // wave-instructions
    const double flop = instr * 64 * (KIND == 2 ? 2 : 4);
    printf("%-34s %d waves/SIMD  %.3f ms  %.2f ns per wave-instr per SIMD  %.1f TFLOP/s (%.0f %% of 157.3)\n", name, THREADS / 256,
           ms, ms * 1e6 / ((double)iters * 128 * (THREADS / 256)), flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1.573e12);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0, 256>(out, "v_pk_fma_f32");
    run<0, 512>(out, "v_pk_fma_f32");
    run<0, 1024>(out, "v_pk_fma_f32");
    run<1, 256>(out, "v_pk_fma_f32 op_sel broadcast");
    run<1, 512>(out, "v_pk_fma_f32 op_sel broadcast");
    run<2, 256>(out, "v_fma_f32");
    run<2, 512>(out, "v_fma_f32");
    run<2, 1024>(out, "v_fma_f32");
    return 0;
}

// Microbenchmark: TWO waves per SIMD (512-thread workgroups; waves w and w+4 of a workgroup share a SIMD).
//   split:  wave A issues only v_mfma_f32_16x16x4_f32 (8 independent accumulators per iteration), wave B only VALU work
//           (8 K instructions per iteration) — does B's VALU issue in the shadow of A's MFMAs?
//   mixed:  both waves run the same stream: 8 MFMAs then one run of 8 K VALU instructions per iteration.
// Prints s_memtime cycles per iteration / 8 for each role.   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/m2 <this file>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K> __device__ __forceinline__ void valu_run() {
#define VAL(j) if (K > j) { _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 v%0, v%0, v3, v4" ::"n"(10 + j)); }
    VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7)
#undef VAL
}
__device__ __forceinline__ void mfma_run() {
#define ONE(i) asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], v1, v2, a[%0:%1]" ::"n"(4 * i), "n"(4 * i + 3));
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7)
#undef ONE
}

template <int MODE, int K> __global__ __launch_bounds__(512) void k(long* out, int iters) {
    asm volatile("v_mov_b32 v1, 1.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 0.999\nv_mov_b32 v4, 0.001" ::: "v1", "v2", "v3", "v4");
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {  // split roles
        if (wave < 4) { for (int it = 0; it < iters; ++it) mfma_run(); }
        else { for (int it = 0; it < iters; ++it) valu_run<K>(); }
    } else {          // mixed: every wave does both
        for (int it = 0; it < iters; ++it) { mfma_run(); valu_run<K>(); }
    }
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
    long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    asm volatile("" ::: "v1","v2","v3","v4","v10","v11","v12","v13","v14","v15","v16","v17",
                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19",
                 "a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
}

template <int MODE, int K> void run(long* d, int iters) {
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(512), 0, 0, d, iters);
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<long> h(2048);
    hipMemcpy(h.data(), d, 2048 * sizeof(long), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 2048; ++i) ((i & 7) < 4 ? a : b) += (double)h[i];
    printf("  K=%d: waves 0-3 %7.1f   waves 4-7 %7.1f\n", K, a / 1024 / ((double)iters * 8), b / 1024 / ((double)iters * 8));
}

int main() {
    long* d; hipMalloc(&d, 2048 * sizeof(long));
    const int it = 2000;
    printf("cycles per iteration / 8 (one MFMA = 32 cycles alone; K v_fma per MFMA slot)\n");
    printf("split roles (waves 0-3: MFMA only, waves 4-7 on the same SIMDs: VALU only)\n");
    run<0, 0>(d, it); run<0, 1>(d, it); run<0, 2>(d, it); run<0, 4>(d, it); run<0, 6>(d, it); run<0, 8>(d, it);
    printf("mixed (each of the two waves per SIMD: 8 MFMAs then 8K v_fma per iteration; per-wave cycles, two waves share the SIMD)\n");
    run<1, 0>(d, it); run<1, 1>(d, it); run<1, 2>(d, it); run<1, 4>(d, it); run<1, 6>(d, it); run<1, 8>(d, it);
    return 0;
}

// Microbenchmark: how much independent VALU work issues in the shadow of v_mfma_f32_16x16x4_f32 within ONE wave per SIMD?
// Each loop iteration runs 8 independent MFMAs (8 accumulator tiles, as in MlpEngine's hidden layers) with K VALU
// instructions of one kind after every MFMA.  Prints s_memtime cycles per MFMA for K = 0..8.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/mfma_valu_overlap tools/micro/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define MFMA(i) "v_mfma_f32_16x16x4_f32 a[" #i "*4:" #i "*4+3], v1, v2, a[" #i "*4:" #i "*4+3]\n"

template <int KIND, int K> __device__ __forceinline__ void body() {
    // KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_accvgpr_read_b32 (AGPRs the MFMAs do not touch), 3: v_rcp_f32,
    //      4: v_accvgpr_read of the MFMA accumulators written 7 MFMAs earlier
#define VAL(j)                                                                                         \
    if (K > j) {                                                                                       \
        if (KIND == 0) asm volatile("v_fma_f32 v%0, v%0, v3, v4" ::"n"(10 + j));                         \
        if (KIND == 1) asm volatile("v_exp_f32 v%0, v%0" ::"n"(10 + j));                                \
        if (KIND == 2) asm volatile("v_accvgpr_read_b32 v%0, a%1" ::"n"(10 + j), "n"(40 + j));          \
        if (KIND == 3) asm volatile("v_rcp_f32 v%0, v%0" ::"n"(10 + j));                                \
        if (KIND == 4) asm volatile("ds_read_b128 v[%0:%1], v5" ::"n"(20 + 4 * j), "n"(23 + 4 * j));    \
        if (KIND == 5) asm volatile("s_mov_b32 s%0, 0" ::"n"(20 + j));                                  \
        if (KIND == 6) asm volatile("s_nop 0");                                                        \
        if (KIND == 7) asm volatile("v_pk_fma_f32 v[%0:%1], v[%0:%1], v[6:7], v[8:9]" ::"n"(20 + 2 * j), "n"(21 + 2 * j)); \
    }
#define ONE(i)                                                                                          \
    asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], v1, v2, a[%0:%1]" ::"n"(4 * i), "n"(4 * i + 3));       \
    VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7)
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7)
    if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
#undef ONE
#undef VAL
}

template <int KIND, int K> __device__ __forceinline__ void body_clump() {
#define ONE(i) asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], v1, v2, a[%0:%1]" ::"n"(4 * i), "n"(4 * i + 3));
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7)
#undef ONE
#define VAL(j) if (K > j) { _Pragma("unroll") for (int r = 0; r < 8; ++r) {                              \
        if (KIND == 0) asm volatile("v_fma_f32 v%0, v%0, v3, v4" ::"n"(10 + j));                         \
        if (KIND == 1) asm volatile("v_exp_f32 v%0, v%0" ::"n"(10 + j)); } }
    VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7)
#undef VAL
}

template <int KIND, int K> __global__ __launch_bounds__(256) void k(long* out, int iters) {
    __shared__ float lds_buf[4096]; lds_buf[threadIdx.x] = 0.f; __syncthreads();
    asm volatile("v_mov_b32 v1, 1.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 0.999\nv_mov_b32 v4, 0.001\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0.999\nv_mov_b32 v7, 0.999\nv_mov_b32 v8, 0.001\nv_mov_b32 v9, 0.001" ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9");
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND >= 10) body_clump<KIND - 10, K>(); else body<KIND, K>();
    }
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
    long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    asm volatile("" ::: "v1","v2","v3","v4","v5","v6","v7","v8","v9","s20","s21","s22","s23","s24","s25","s26","s27","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v10","v11","v12","v13","v14","v15","v16","v17",
                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19",
                 "a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a40","a41","a42","a43","a44","a45","a46","a47");
}

template <int KIND, int K> double run(long* d, int iters) {
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<long> h(1024);
    hipMemcpy(h.data(), d, 1024 * sizeof(long), hipMemcpyDeviceToHost);
    double s = 0; for (long v : h) s += (double)v;
    return s / 1024 / ((double)iters * 8);
}

template <int KIND> void sweep(long* d, const char* name) {
    const int it = 2000;
    printf("%-28s", name);
    printf(" %6.1f", run<KIND, 0>(d, it)); printf(" %6.1f", run<KIND, 1>(d, it)); printf(" %6.1f", run<KIND, 2>(d, it));
    printf(" %6.1f", run<KIND, 3>(d, it)); printf(" %6.1f", run<KIND, 4>(d, it)); printf(" %6.1f", run<KIND, 5>(d, it));
    printf(" %6.1f", run<KIND, 6>(d, it)); printf(" %6.1f", run<KIND, 7>(d, it)); printf(" %6.1f\n", run<KIND, 8>(d, it));
}

int main() {
    long* d; hipMalloc(&d, 1024 * sizeof(long));
    printf("cycles per MFMA (clock counter units) with K VALU ops after each MFMA; K = 0..8\n");
    sweep<0>(d, "v_fma_f32");
    sweep<1>(d, "v_exp_f32");
    sweep<3>(d, "v_rcp_f32");
    sweep<2>(d, "v_accvgpr_read_b32");
    sweep<7>(d, "v_pk_fma_f32");
    sweep<4>(d, "ds_read_b128 (+wait/8)");
    sweep<5>(d, "s_mov_b32");
    sweep<6>(d, "s_nop 0");
    printf("clumped: 8 MFMAs back to back, then 8K ops in one run\n");
    sweep<10>(d, "v_fma_f32 clumped");
    sweep<11>(d, "v_exp_f32 clumped");
    return 0;
}

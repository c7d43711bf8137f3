// Microbenchmark (round 2): does VALU work issue in the shadow of v_mfma_f32_32x32x2_f32 (64-cycle issue) within ONE wave
// per SIMD?  Round 1 measured that nothing hides under v_mfma_f32_16x16x4_f32 (32 cycles): profiles/r01_micro_mfma_valu_overlap.txt.
// Each loop iteration runs 4 independent 32x32x2 MFMAs (4 accumulator tiles of 16 AGPRs) with K VALU instructions of one
// kind after every MFMA.  Prints clock-counter cycles per MFMA for K = 0..12 (step 1..) — 64.0 = fully hidden.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/mfma32_valu_overlap tools/micro/mfma32_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int K> __device__ __forceinline__ void body() {
#define VAL(j)                                                                                         \
    if (K > j) {                                                                                       \
        if (KIND == 0) asm volatile("v_fma_f32 v%0, v%0, v3, v4" ::"n"(10 + j));                         \
        if (KIND == 1) asm volatile("v_exp_f32 v%0, v%0" ::"n"(10 + j));                                \
        if (KIND == 2) asm volatile("v_accvgpr_read_b32 v%0, a[80+" #j "]" ::"n"(10 + j));          \
        if (KIND == 4) asm volatile("ds_read_b128 v[24+4*" #j ":27+4*" #j "], v5");    \
        if (KIND == 7) asm volatile("v_pk_mul_f32 v[24+2*" #j ":25+2*" #j "], v[24+2*" #j ":25+2*" #j "], v[6:7]"); \
    }
#define ONE(i)                                                                                          \
    asm volatile("v_mfma_f32_32x32x2_f32 a[16*" #i ":16*" #i "+15], v1, v2, a[16*" #i ":16*" #i "+15]");    \
    VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7) VAL(8) VAL(9) VAL(10) VAL(11)
    ONE(0) ONE(1) ONE(2) ONE(3)
    if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
#undef ONE
#undef VAL
}

// clumped: 4 MFMAs back to back, then 4K ops in one run
template <int KIND, int K> __device__ __forceinline__ void body_clump() {
#define ONE(i) asm volatile("v_mfma_f32_32x32x2_f32 a[16*" #i ":16*" #i "+15], v1, v2, a[16*" #i ":16*" #i "+15]");
    ONE(0) ONE(1) ONE(2) ONE(3)
#undef ONE
#define VAL(j) if (K > j) { _Pragma("unroll") for (int r = 0; r < 4; ++r) {                              \
        if (KIND == 0) asm volatile("v_fma_f32 v%0, v%0, v3, v4" ::"n"(10 + j));                         \
        if (KIND == 1) asm volatile("v_exp_f32 v%0, v%0" ::"n"(10 + j)); } }
    VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7) VAL(8) VAL(9) VAL(10) VAL(11)
#undef VAL
}

template <int KIND, int K> __global__ __launch_bounds__(256) void k(long* out, int iters) {
    __shared__ float lds_buf[4096]; lds_buf[threadIdx.x] = 0.f; __syncthreads();
    asm volatile("v_mov_b32 v1, 1.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 0.999\nv_mov_b32 v4, 0.001\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0.999\nv_mov_b32 v7, 0.999" ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7");
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND >= 10) body_clump<KIND - 10, K>(); else body<KIND, K>();
    }
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
    long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    asm volatile("" ::: "v1","v2","v3","v4","v5","v6","v7","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21",
                 "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43",
                 "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63",
                 "v64","v65","v66","v67","v68","v69","v70","v71",
                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19",
                 "a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39",
                 "a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59",
                 "a60","a61","a62","a63","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91");
}

template <int KIND, int K> double run(long* d, int iters) {
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<long> h(1024);
    hipMemcpy(h.data(), d, 1024 * sizeof(long), hipMemcpyDeviceToHost);
    double s = 0; for (long v : h) s += (double)v;
    return s / 1024 / ((double)iters * 4);
}

template <int KIND> void sweep(long* d, const char* name) {
    const int it = 2000;
    printf("%-28s", name);
    printf(" %6.1f", run<KIND, 0>(d, it)); printf(" %6.1f", run<KIND, 1>(d, it)); printf(" %6.1f", run<KIND, 2>(d, it));
    printf(" %6.1f", run<KIND, 3>(d, it)); printf(" %6.1f", run<KIND, 4>(d, it)); printf(" %6.1f", run<KIND, 6>(d, it));
    printf(" %6.1f", run<KIND, 8>(d, it)); printf(" %6.1f", run<KIND, 10>(d, it)); printf(" %6.1f\n", run<KIND, 12>(d, it));
}

int main() {
    long* d; hipMalloc(&d, 1024 * sizeof(long));
    printf("cycles per v_mfma_f32_32x32x2_f32 with K ops after each MFMA; K = 0 1 2 3 4 6 8 10 12\n");
    sweep<0>(d, "v_fma_f32");
    sweep<1>(d, "v_exp_f32");
    sweep<2>(d, "v_accvgpr_read_b32");
    sweep<7>(d, "v_pk_mul_f32");
    sweep<4>(d, "ds_read_b128 (+wait/4)");
    printf("clumped: 4 MFMAs back to back, then 4K ops in one run\n");
    sweep<10>(d, "v_fma_f32 clumped");
    sweep<11>(d, "v_exp_f32 clumped");
    return 0;
}

// Microbenchmark (round 2, K4 design): one 64x64 layer per (lane = unit x slab pair) with v_pk_fma_f32, the weights
// wave-uniform and fed as SGPR operands (scalar loads through the constant cache), the layer's inputs in 64 VGPR pairs,
// outputs written to LDS.  Prints cycles per v_pk_fma_f32 per wave (4.0 = one per issue slot = 100 % of the fp32 VALU
// peak with one wave per SIMD) for 1 .. 4 waves per CU and the whole chip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_pkfma_sgpr tools/micro/valu_pkfma_sgpr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc.xy += in.xy * w: w is one half of a 64-bit SGPR pair (the packed instruction takes a 64-bit scalar source),
// broadcast to both halves with op_sel / op_sel_hi
__device__ __forceinline__ void pkfma_lo(f32x2& acc, const f32x2& in, const f32x2& wp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(in), "s"(wp));
}
__device__ __forceinline__ void pkfma_vlo(f32x2& acc, const f32x2& in, const f32x2& wp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(in), "v"(wp));
}
__device__ __forceinline__ void pkfma_vhi(f32x2& acc, const f32x2& in, const f32x2& wp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(in), "v"(wp));
}
__device__ __forceinline__ void pkfma_hi(f32x2& acc, const f32x2& in, const f32x2& wp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(in), "s"(wp));
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ W, float* __restrict__ out, long* cyc, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2* lds = reinterpret_cast<f32x2*>(smem) + (threadIdx.x >> 6) * (64 * 64) + (threadIdx.x & 63);
    if (MODE == 5 || MODE == 6 || MODE == 8) {  // weight image into LDS behind the activation buffers
        float* wl = reinterpret_cast<float*>(smem + 4 * 64 * 64 * 8);
        for (int i = threadIdx.x; i < 2 * 4096; i += 256) wl[i] = W[i];
        __syncthreads();
    }
    f32x2 in[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) in[k] = f32x2{0.001f * (threadIdx.x + k), 0.002f * k};
    const long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < layers; ++l) {
        const float* wl = W + (l & 1) * 4096;
        if (MODE == 8) {  // LDS weights, four chains, loop-carried software pipeline: bank c+1 is in flight while bank c is consumed
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4* wq = reinterpret_cast<const f32x4*>(smem + 4 * 64 * 64 * 8) + (l & 1) * 1024;
            f32x4 cur[8], nxt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) cur[i] = wq[i];
#pragma nounroll
            for (int np = 0; np < 32; ++np) {
                f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4* nq = wq + ((np * 32 + 8 * (c + 1)) & 1023);  // c == 3: the first bank of the next pair (one bank of padding at the end)
#pragma unroll
                    for (int i = 0; i < 8; ++i) nxt[i] = nq[i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int k = 16 * c + 2 * i;
                        const f32x2 p0 = {cur[i][0], cur[i][1]}, p1 = {cur[i][2], cur[i][3]};
                        pkfma_vlo(a0, in[k], p0); pkfma_vlo(a1, in[k], p1);
                        pkfma_vhi(a2, in[k + 1], p0); pkfma_vhi(a3, in[k + 1], p1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
                }
                a0 += a2; a1 += a3;
                lds[(2 * np) * 64] = a0;
                lds[(2 * np + 1) * 64] = a1;
            }
        } else
#pragma nounroll
        for (int np = 0; np < 32; ++np) {  // neuron pairs; weights stored [np][kc 4][2 neurons][16 k]
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
            if (MODE == 7) {  // no loads at all, weights in VGPR pairs: is the third 64-bit VGPR source what slows the LDS variant?
                f32x2 a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
                f32x2 wv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { wv[i] = f32x2{0.01f * i, 0.02f * i}; asm volatile("" : "+v"(wv[i])); }
#pragma unroll
                for (int k = 0; k < 64; k += 2) {
                    const f32x2 p0 = wv[(k / 2) & 7], p1 = wv[(k / 2 + 3) & 7];
                    pkfma_vlo(a0, in[k], p0); pkfma_vlo(a1, in[k], p1);
                    pkfma_vhi(a2, in[k + 1], p0); pkfma_vhi(a3, in[k + 1], p1);
                }
                a0 += a2; a1 += a3;
                lds[(2 * np) * 64] = a0;
                lds[(2 * np + 1) * 64] = a1;
                continue;
            }
            if (MODE == 3 || MODE == 4) {  // no scalar loads in the loop: the issue rate of v_pk_fma_f32 with SGPR operands alone
                const f32x16* wf = reinterpret_cast<const f32x16*>(W);
                const f32x16 w0 = wf[0], w1 = wf[1], w2 = wf[2], w3 = wf[3];
                f32x2 a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
                    for (int j = 0; j < 16; j += 2) {
                        const f32x2 p0 = {w0[j], w0[j + 1]}, p1 = {w1[j], w1[j + 1]}, p2 = {w2[j], w2[j + 1]}, p3 = {w3[j], w3[j + 1]};
                        if (MODE == 3) {
                            pkfma_lo(a0, in[kc * 16 + j], p0); pkfma_lo(a1, in[kc * 16 + j], p1);
                            pkfma_hi(a0, in[kc * 16 + j + 1], p0); pkfma_hi(a1, in[kc * 16 + j + 1], p1);
                        } else {
                            pkfma_lo(a0, in[kc * 16 + j], p0); pkfma_lo(a1, in[kc * 16 + j], p1);
                            pkfma_lo(a2, in[kc * 16 + j], p2); pkfma_lo(a3, in[kc * 16 + j], p3);
                        }
                    }
                }
                if (MODE == 4) { a0 += a2; a1 += a3; }
                lds[(2 * np) * 64] = a0;
                lds[(2 * np + 1) * 64] = a1;
                continue;
            }
            if (MODE == 5 || MODE == 6) {  // weights from LDS: wave-uniform ds_read_b128 (4 weights), VGPR operands
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4* wq = reinterpret_cast<const f32x4*>(smem + 4 * 64 * 64 * 8) + (l & 1) * 1024 + np * 32;
                f32x2 a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
                // explicit two-bank software pipeline: the next 16 k (8 x ds_read_b128) are requested, pinned by a scheduling
                // fence, before the current 16 k are consumed (the LDS returns in order: counted lgkmcnt waits)
                f32x4 cur[8], nxt[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = wq[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) nxt[i] = wq[8 * (c + 1) + i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int k = 16 * c + 2 * i;
                        const f32x2 p0 = {cur[i][0], cur[i][1]}, p1 = {cur[i][2], cur[i][3]};
                        if (MODE == 5) {
                            pkfma_vlo(a0, in[k], p0); pkfma_vlo(a1, in[k], p1);
                            pkfma_vhi(a0, in[k + 1], p0); pkfma_vhi(a1, in[k + 1], p1);
                        } else {
                            pkfma_vlo(a0, in[k], p0); pkfma_vlo(a1, in[k], p1);
                            pkfma_vhi(a2, in[k + 1], p0); pkfma_vhi(a3, in[k + 1], p1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
                }
                if (MODE == 6) { a0 += a2; a1 += a3; }
                lds[(2 * np) * 64] = a0;
                lds[(2 * np + 1) * 64] = a1;
                continue;
            }
            if (MODE == 2) {  // explicit double buffering across neuron pairs: the next pair's first half is requested early
                const f32x16* wq = reinterpret_cast<const f32x16*>(wl + np * 128);
                f32x16 c0 = wq[0], c1 = wq[1], c2 = wq[2], c3 = wq[3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x16 n0 = wq[4 + 4 * h], n1 = wq[5 + 4 * h], n2 = wq[6 + 4 * h], n3 = wq[7 + 4 * h];  // prefetch (runs past the end by one half: padded)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x16 w0 = q ? c2 : c0, w1 = q ? c3 : c1;
#pragma unroll
                        for (int j = 0; j < 16; j += 2) {
                            const f32x2 p0 = {w0[j], w0[j + 1]}, p1 = {w1[j], w1[j + 1]};
                            pkfma_lo(a0, in[(2 * h + q) * 16 + j], p0); pkfma_lo(a1, in[(2 * h + q) * 16 + j], p1);
                            pkfma_hi(a0, in[(2 * h + q) * 16 + j + 1], p0); pkfma_hi(a1, in[(2 * h + q) * 16 + j + 1], p1);
                        }
                    }
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                }
                lds[(2 * np) * 64] = a0;
                lds[(2 * np + 1) * 64] = a1;
                continue;
            }
            const f32x16* w16 = reinterpret_cast<const f32x16*>(wl + np * 128);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const f32x16 w0 = w16[kc * 2], w1 = w16[kc * 2 + 1];
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const f32x2 p0 = {w0[j], w0[j + 1]}, p1 = {w1[j], w1[j + 1]};
                    pkfma_lo(a0, in[kc * 16 + j], p0);
                    pkfma_lo(a1, in[kc * 16 + j], p1);
                    pkfma_hi(a0, in[kc * 16 + j + 1], p0);
                    pkfma_hi(a1, in[kc * 16 + j + 1], p1);
                }
            }
            if (MODE == 1) {  // epilogue stand-in: tanh on .x, (1 - h^2) scaling on .y
                const float e0 = __builtin_amdgcn_exp2f(a0.x * 2.885f), h0 = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e0);
                const float e1 = __builtin_amdgcn_exp2f(a1.x * 2.885f), h1 = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e1);
                a0 = f32x2{h0, a0.y * (1.f - h0 * h0)};
                a1 = f32x2{h1, a1.y * (1.f - h1 * h1)};
            }
            lds[(2 * np) * 64] = a0;
            lds[(2 * np + 1) * 64] = a1;
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) in[k] = lds[k * 64];
    }
    const long t1 = __builtin_readcyclecounter();
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 64; ++k) s += in[k];
    out[(long)blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const float* W, float* out, long* cyc, int grid, int layers, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 64 * 8 + (MODE >= 5 ? 32768 : 0));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 4 * 64 * 64 * 8 + (MODE >= 5 ? 32768 : 0), 0, W, out, cyc, layers);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 4 * 64 * 64 * 8 + (MODE >= 5 ? 32768 : 0), 0, W, out, cyc, layers);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * sizeof(long), hipMemcpyDeviceToHost);
    double s = 0; for (long v : h) s += (double)v;
    const double per = s / (grid * 4) / ((double)layers * 4096);
    const double tflops = (double)grid * 4 * layers * 4096 * 256 / (ms * 1e-3) / 1e12;  // 64 lanes x 2 x 2 flop per pk_fma
    printf("%-22s grid %4d  cycles/pk_fma %.2f   %.3f ms   %.1f TFLOP/s (%.0f %% of 157.3)\n", name, grid, per, ms, tflops, tflops / 1.573);
}

int main() {
    float *W, *out; long* cyc;
    hipMalloc(&W, (2 * 4096 + 256) * sizeof(float)); hipMalloc(&out, 2048 * 256 * sizeof(float)); hipMalloc(&cyc, 2048 * 4 * sizeof(long));
    std::vector<float> hw(2 * 4096 + 256);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 7919) % 23 - 11) / 11.f;
    hipMemcpy(W, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
    run<0>(W, out, cyc, 1, 64, "bare, 1 workgroup");
    run<0>(W, out, cyc, 256, 64, "bare, 256 (1/CU)");
    run<0>(W, out, cyc, 1024, 64, "bare, 1024 (4 rounds)");
    run<1>(W, out, cyc, 256, 64, "+epilogue, 256");
    run<1>(W, out, cyc, 1024, 64, "+epilogue, 1024");
    run<3>(W, out, cyc, 256, 64, "no s_load, 2 acc");
    run<4>(W, out, cyc, 256, 64, "no s_load, 4 acc");
    run<2>(W, out, cyc, 256, 64, "half-pair prefetch");
    run<7>(W, out, cyc, 256, 64, "no load, VGPR weights");
    run<5>(W, out, cyc, 256, 64, "LDS weights, 2 acc");
    run<6>(W, out, cyc, 256, 64, "LDS weights, 4 acc");
    run<6>(W, out, cyc, 1024, 64, "LDS weights, 4 acc");
    run<8>(W, out, cyc, 256, 64, "LDS weights, pipelined");
    run<8>(W, out, cyc, 1024, 64, "LDS weights, pipelined");
    return 0;
}

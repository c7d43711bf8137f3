// ac_select.hpp — selection + packing kernels of the sharded random-restart driver and of the solver's line search
// (build-side; the reference solves ONE instance, so there is no counterpart to compare with):
//   k_best_records   per-rank best-K trajectories by cost (NaN -> +inf, ties -> lowest index) packed as
//                    rows [cost, X(H+1,13), U(H,7)] straight from the rollout-shaped buffers — what each rank
//                    contributes to the ONE all-gather of the path
//   k_merge_records  the K*world gathered rows sorted by cost (rank sort: every row counts the rows that beat it)
//   k_ilqr_accept    per-instance argmin over the line-search candidates + conditional copy into the iterate
// All three are HBM/latency-bound gathers (a record is 8 KB at H = 100); nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include "ac_math.hpp"

namespace ac {

constexpr int kSelBlock = 1024;  // 16 waves: one workgroup scans a rank's costs (16 384 at cfg4: 16 per lane)
constexpr int kMaxBestK = 8;

struct CostKey {
    float c;
    long i;
};
// strict lexicographic order on (cost, index); costs are already sanitised (no NaN)
AC_DI bool key_less(const CostKey& a, const CostKey& b) { return a.c < b.c || (a.c == b.c && a.i < b.i); }
AC_DI float sanitise_cost(float c) { return (c != c) ? __builtin_huge_valf() : c; }  // NaN -> +inf (a crashed rollout loses)

AC_DI CostKey wave_min_key(CostKey k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        CostKey o;
        o.c = __shfl_xor(k.c, off);
        const int lo = __shfl_xor((int)(k.i & 0xffffffffl), off), hi = __shfl_xor((int)(k.i >> 32), off);
        o.i = ((long)hi << 32) | (unsigned int)lo;
        if (key_less(o, k)) k = o;
    }
    return k;
}

// grid = K workgroups of kSelBlock lanes.  Every workgroup runs the same K selection rounds (K * B cost reads from L2: a
// rank's costs are 64 KB) and workgroup r packs record r — no inter-workgroup dependency, one launch.
// cost [B]; X [H+1][13][B]; U [H][7][B]; rec [K][1 + (H+1)*13 + H*7].
__global__ __launch_bounds__(kSelBlock) void k_best_records(const float* __restrict__ cost, const float* __restrict__ X,
                                                            const float* __restrict__ U, long B, long H, int K,
                                                            float* __restrict__ rec) {
    __shared__ CostKey s_part[kSelBlock / 64];
    __shared__ CostKey s_sel;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    CostKey prev;
    prev.c = -__builtin_huge_valf(); prev.i = -1;  // nothing selected yet: every (c, i) is greater
    const int mine = blockIdx.x;                   // this workgroup packs the `mine`-th best
    for (int round = 0; round <= mine; ++round) {
        CostKey best;
        best.c = __builtin_huge_valf(); best.i = B;  // sentinel: greater than every real key
        for (long i = tid; i < B; i += kSelBlock) {
            CostKey k;
            k.c = sanitise_cost(cost[i]); k.i = i;
            if (key_less(prev, k) && key_less(k, best)) best = k;
        }
        best = wave_min_key(best);
        if (lane == 0) s_part[wave] = best;
        __syncthreads();
        if (wave == 0) {
            CostKey k = lane < kSelBlock / 64 ? s_part[lane] : best;
            k = wave_min_key(k);
            if (lane == 0) s_sel = k;
        }
        __syncthreads();
        prev = s_sel;
        __syncthreads();  // s_sel / s_part are rewritten by the next round
    }
    const long nx = (H + 1) * 13, nu = H * 7, R = 1 + nx + nu;
    float* out = rec + (long)mine * R;
    const long idx = prev.i;  // < B because K <= B (checked on the host)
    if (tid == 0) out[0] = prev.c;
    for (long j = tid; j < nx; j += kSelBlock) out[1 + j] = X[j * B + idx];
    for (long j = tid; j < nu; j += kSelBlock) out[1 + nx + j] = U[j * B + idx];
    (void)K;
}

// rec_in [n][R] -> rec_out [n][R] sorted by column 0 ascending (NaN -> +inf, ties -> lower input row first).
// grid = n workgroups (n <= 1024: K * world rows), each ranks its own row and copies it.
__global__ __launch_bounds__(256) void k_merge_records(const float* __restrict__ rec_in, long n, long R,
                                                       float* __restrict__ rec_out) {
    __shared__ int s_cnt[4];
    const long me = blockIdx.x;
    CostKey mine;
    mine.c = sanitise_cost(rec_in[me * R]); mine.i = me;
    int cnt = 0;
    for (long j = threadIdx.x; j < n; j += 256) {
        CostKey o;
        o.c = sanitise_cost(rec_in[j * R]); o.i = j;
        cnt += key_less(o, mine) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    const long rank = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const float* src = rec_in + me * R;
    float* dst = rec_out + rank * R;
    for (long j = threadIdx.x; j < R; j += 256) dst[j] = (j == 0) ? mine.c : src[j];
}

// Line-search acceptance of the batched solver sweep: candidates a = 0..na-1 of instance b sit in column a*B + b of
// Xc [H+1][13][na*B], Uc [H][7][na*B] with costs Jc [na*B]; J0 [B] is the cost of the current iterate X [H+1][13][B],
// U [H][7][B].  best = min_a Jc (NaN / inf never win; ties -> lowest a); if best < J0 the candidate replaces the iterate.
// Jout [B] = accepted cost, improved [B] (bytes 0/1, may be NULL).  grid.x over instances, grid.y over row chunks of the
// (H+1)*13 + H*7 rows; every lane redoes the na-way argmin of its instance (na <= 8 reads, L2-resident).
constexpr int kAcceptRows = 64;
__global__ __launch_bounds__(256) void k_ilqr_accept(const float* __restrict__ Jc, const float* __restrict__ J0,
                                                     const float* __restrict__ Xc, const float* __restrict__ Uc, int na,
                                                     long B, long H, float* __restrict__ X, float* __restrict__ U,
                                                     float* __restrict__ Jout, unsigned char* __restrict__ improved) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float best = __builtin_huge_valf();
    int arg = 0;
    for (int a = 0; a < na; ++a) {
        const float c = Jc[(long)a * B + b];
        if (c < best && c > -__builtin_huge_valf()) { best = c; arg = a; }  // non-finite costs (NaN, +-inf) never win
    }
    const float j0 = J0[b];
    const bool imp = best < j0;
    if (blockIdx.y == 0) {
        Jout[b] = imp ? best : j0;
        if (improved) improved[b] = imp ? 1 : 0;
    }
    if (!imp) return;
    const long nx = (H + 1) * 13, nrows = nx + H * 7, Bc = (long)na * B, col = (long)arg * B + b;
    const long r0 = (long)blockIdx.y * kAcceptRows, r1 = r0 + kAcceptRows < nrows ? r0 + kAcceptRows : nrows;
    for (long r = r0; r < r1; ++r) {
        if (r < nx) X[r * B + b] = Xc[r * Bc + col];
        else U[(r - nx) * B + b] = Uc[(r - nx) * Bc + col];
    }
}

}  // namespace ac

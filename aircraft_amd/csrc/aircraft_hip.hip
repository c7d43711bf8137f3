// aircraft_hip.hip — C ABI (include/aircraft_hip.h) + kernel dispatch for libaircraft_hip.so.
// gfx950 only.  No allocation or synchronisation inside the compute entry points (graph-capturable).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ac_kernels_analytic.hpp"
#include "ac_nn_decl.hpp"
#include "ac_ilqr.hpp"
#include "ac_goal.hpp"
#include "ac_track.hpp"
#include "ac_hess.hpp"
#include "ac_hess_adj.hpp"
#include "ac_hess_nn.hpp"
#include "ac_hess_rev.hpp"
#include "ac_select.hpp"

using namespace ac;

namespace ac {
// Trajectory cost for the random-restart driver: X [H+1][13][B] -> cost [B]
__global__ __launch_bounds__(kBlock) void k_traj_cost(const float* __restrict__ X, long B, long H, float gx, float gy,
                                                      float gz, float w_track, float w_goal,
                                                      float* __restrict__ cost) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    float acc = 0.f, last = 0.f;
    for (long k = 0; k <= H; ++k) {
        const float* xk = X + k * 13 * B;
        const float dx = xk[i] - gx, dy = xk[B + i] - gy, dz = xk[2 * B + i] - gz;
        last = dx * dx + dy * dy + dz * dz;
        acc += last;
    }
    cost[i] = w_track * acc + w_goal * last;
}

}  // namespace ac

namespace {

thread_local char g_err[512] = "";

int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return AC_ERR_HIP;
}
// every non-HIP error return that carries text goes through here, so ac_last_error() never shows a stale message
int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
#define AC_HIP(call)                                           \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hip_fail(e_, #call);      \
    } while (0)

constexpr int kLdsBudget = 160 * 1024;  // gfx950 LDS per workgroup

}  // namespace

struct ac_handle {
    DevParams dp;
    int device;
    int num_cus;
    // measurement aids, diagnostic flavours only (-DAC_DIAG_ENV; always false in the product library):
    bool no_pair;  // AIRCRAFT_HIP_NO_PAIR=1: never route a remainder to k_nn_step_sens_pair
    bool all_pair; // AIRCRAFT_HIP_ALL_PAIR=1: every unit through k_nn_step_sens_pair
    bool has_linear, has_poly, has_mlp;
    MlpPlan plan;
    MlpPlan plan_sens;  // the MFMA sensitivity engines' plan: last layer = [bias][wlt] for MlpEngine::last_valu (ac_set_mlp)
    MlpPlan plan_rev;   // width 128 on the matrix cores, <= 3 hidden products: plan_sens + the TRANSPOSED hidden blocks for the
    bool has_rev;       // reverse sweep of k_nn_stage_tensors_rev (ac_hess_rev.hpp); rev_layers = layers of the net itself
    int rev_layers;
    float* d_rev_scratch;  // per-wave layer states of that kernel (ac_reserve_hess_workspace)
    size_t rev_scratch_floats;
    int wt;         // register tiles per slab the plan needs (2, 4 or 8)
    int use_mfma;
    float* d_blob;  // packed MLP weights + biases (device)
    size_t blob_floats;
    // "MFMA off" flavour on the tiled v_pk_fma_f32 engine (ac_mlp_valu.hpp): hidden widths <= 64, >= 2 layers
    bool has_vplan;
    ValuPlan vplan;
    int vwidth;        // 32 or 64
    float* d_vblob;    // weight image [layer][K][N] (+ biases), device
    unsigned* d_queue; // ticket counter of the persistent tiled kernels (GroupQueue, ac_mlp_valu.hpp); 0 between launches
    float* d_hess_ws;  // [n][4][126] stage tensors of the MLP Hessian path (ac_reserve_hess_workspace)
    size_t hess_ws_floats;
    float* d_hess_ws2;  // sub-step composition of the second-order blocks (substeps > 1)
    size_t hess_ws2_floats;
    float* d_poly_tab;  // coefficient, intercept and gradient tables of the cubic fits (DevParams::poly_tab)
    float* d_track;  // [nseg][3][4] segment cubics (device)
    TrackDev track;
    // kernels whose dynamic-LDS limit was already raised on this handle's device (hipFuncSetAttribute is not free)
    const void* lds_fn[48];
    int lds_bytes_set[48];
    int n_lds_fn;
    // last launch (profiling aid)
    char last_name[64];
    int last_grid, last_block, last_lds;
};

namespace {

void note_launch(ac_handle* h, const char* name, int grid, int block, int lds) {
    snprintf(h->last_name, sizeof(h->last_name), "%s", name);
    h->last_grid = grid; h->last_block = block; h->last_lds = lds;
}

int check_params(const ac_params* p) {
    if (!p) return AC_ERR_BAD_ARG;
    if (p->substeps < 1) return AC_ERR_BAD_ARG;
    if (p->model_kind < AC_MODEL_DEFAULT || p->model_kind > AC_MODEL_QUAD) return AC_ERR_BAD_ARG;
    return AC_OK;
}

int model_ready(const ac_handle* h) {
    switch (h->dp.p.model_kind) {
        case AC_MODEL_LINEAR: return h->has_linear ? AC_OK : AC_ERR_NO_MODEL;
        case AC_MODEL_POLY: return h->has_poly ? AC_OK : AC_ERR_NO_MODEL;
        case AC_MODEL_NN: return h->has_mlp ? AC_OK : AC_ERR_NO_MODEL;
        default: return AC_OK;
    }
}

// Raise a kernel's dynamic-LDS limit once per (handle, kernel, size): the attribute is sticky per device, and the call
// costs a driver round trip that does not belong in front of every launch.
template <class K> int set_lds_limit(ac_handle* h, K kernel, int bytes) {
    if (bytes <= 64 * 1024) return AC_OK;
    const void* fn = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < h->n_lds_fn; ++i)
        if (h->lds_fn[i] == fn) {
            if (h->lds_bytes_set[i] >= bytes) return AC_OK;
            AC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            h->lds_bytes_set[i] = bytes;
            return AC_OK;
        }
    AC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (h->n_lds_fn < 48) { h->lds_fn[h->n_lds_fn] = fn; h->lds_bytes_set[h->n_lds_fn] = bytes; ++h->n_lds_fn; }
    return AC_OK;
}

// Launch on the handle's device whatever the caller's current device is (a process may hold handles on several GPUs);
// restores the caller's device on scope exit.  hipSetDevice is host state only: legal during stream capture.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const ac_handle* h) {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define AC_ENTER(h) g_err[0] = 0; DeviceGuard dev_guard_(h)

#ifndef AC_HESS_N_POLY
#define AC_HESS_N_POLY 1
#endif
#ifndef AC_HESS_N_OTHER
#define AC_HESS_N_OTHER 2
#endif
template <int MODEL> struct HessN { static constexpr int value = (MODEL == AC_MODEL_POLY) ? AC_HESS_N_POLY : AC_HESS_N_OTHER; };

// The reverse sweep in duals (ac_hess_adj.hpp), one wave per direction group and 64 units; -DAC_HESS_JETS keeps the
// second-order-jet kernel of rounds 1-3 (k_step_hess) as the A/B flavour.
// directions per lane: two where the sweep fits the register file with them (no scratch); the MLP provider's extra state
// (stage tensors, the 5 x 5 contraction) leaves room for one (AC_HESS_REV_N_NN: measured both)
#ifndef AC_HESS_REV_N
#define AC_HESS_REV_N 2
#endif
#ifndef AC_HESS_REV_N_POLY
#define AC_HESS_REV_N_POLY 1
#endif
#ifndef AC_HESS_REV_N_NN
#define AC_HESS_REV_N_NN 2
#endif
template <int MODEL>
void launch_hess(ac_handle* h, hipStream_t st, const float* X, const float* U, float dt, const float* dt_per_unit,
                        const float* Lam, long n, long blk, float* Hout, int* grid_out) {
#ifndef AC_HESS_JETS
    {
        constexpr int NR = MODEL == AC_MODEL_NN ? AC_HESS_REV_N_NN : (MODEL == AC_MODEL_POLY ? AC_HESS_REV_N_POLY : AC_HESS_REV_N);
        static_assert((16 / NR) % (kBlock / 64) == 0, "direction groups per workgroup");
        const dim3 grid((unsigned)((n + 63) / 64), (16 / NR) / (kBlock / 64));
        hipLaunchKernelGGL((k_step_hess_rev<MODEL, NR>), grid, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, Lam,
                           (const float*)h->d_hess_ws, n, blk, Hout);
        *grid_out = (int)(grid.x * grid.y);
        return;
    }
#endif
    constexpr int N = HessN<MODEL>::value;
    const long lanes = n * HessTasks<N>::value;  // upper-triangle tasks of every unit, packed across workgroups
    const int grid = (int)((lanes + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_step_hess<MODEL, N>), grid, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, Lam,
                       (const float*)h->d_hess_ws, n, blk, Hout);
    *grid_out = grid;
}

// ---- dispatch helpers --------------------------------------------------------------------------
#define AC_LAUNCH_ANALYTIC(KERNEL, GRID, BLOCK, ...)                                                        \
    do {                                                                                                    \
        switch (h->dp.p.model_kind) {                                                                       \
            case AC_MODEL_LINEAR: hipLaunchKernelGGL(KERNEL<AC_MODEL_LINEAR>, GRID, BLOCK, 0, st, h->dp, __VA_ARGS__); break; \
            case AC_MODEL_POLY: hipLaunchKernelGGL(KERNEL<AC_MODEL_POLY>, GRID, BLOCK, 0, st, h->dp, __VA_ARGS__); break;     \
            case AC_MODEL_QUAD: hipLaunchKernelGGL(KERNEL<AC_MODEL_QUAD>, GRID, BLOCK, 0, st, h->dp, __VA_ARGS__); break;     \
            default: hipLaunchKernelGGL(KERNEL<AC_MODEL_DEFAULT>, GRID, BLOCK, 0, st, h->dp, __VA_ARGS__); break;             \
        }                                                                                                   \
    } while (0)

// Instantiate an NN kernel template for the (WT, MFMA) the handle needs.
#define AC_NN_CASE_PLAN(PLAN_, WT_, MF_, KERNEL_EXPR, GRID, BLOCK, ...)                                \
    if (h->wt == WT_ && (h->use_mfma != 0) == MF_) {                                                 \
        auto kern = KERNEL_EXPR;                                                                     \
        int rc_ = set_lds_limit(h, kern, (PLAN_).lds_total);                                         \
        if (rc_ != AC_OK) return rc_;                                                                \
        hipLaunchKernelGGL(kern, GRID, BLOCK, (PLAN_).lds_total, st, h->dp, PLAN_, h->d_blob, __VA_ARGS__); \
        launched = true;                                                                             \
    }
#define AC_NN_CASE(WT_, MF_, KERNEL_EXPR, GRID, BLOCK, ...) AC_NN_CASE_PLAN(h->plan, WT_, MF_, KERNEL_EXPR, GRID, BLOCK, __VA_ARGS__)
// kernels on MlpEngine (sensitivity and forward): the matrix-core flavour runs its edge layers on the vector ALUs and takes
// plan_sens; the cross-lane validation flavour keeps plan.  (The second-order and the cooperative rollout engines: plan.)
#define AC_NN_CASE_SENS(WT_, MF_, KERNEL_EXPR, GRID, BLOCK, ...)                                      \
    if (MF_) { AC_NN_CASE_PLAN(h->plan_sens, WT_, MF_, KERNEL_EXPR, GRID, BLOCK, __VA_ARGS__) }       \
    else { AC_NN_CASE_PLAN(h->plan, WT_, MF_, KERNEL_EXPR, GRID, BLOCK, __VA_ARGS__) }

}  // namespace

extern "C" {

const char* ac_last_error(void) { return g_err; }
const char* ac_version(void) { return "aircraft_hip 0.1 (gfx950)"; }

int ac_device_arch(char* buf, size_t len) {
    if (!buf || len == 0) return AC_ERR_BAD_ARG;
    int dev = 0;
    AC_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    AC_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, len, "%s", prop.gcnArchName);
    return AC_OK;
}

int ac_create(const ac_params* params, ac_handle** out) {
    if (!out) return AC_ERR_BAD_ARG;
    int rc = check_params(params);
    if (rc != AC_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible");
        return AC_ERR_NO_DEVICE;
    }
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ERR_BAD_ARG;
    memset(h, 0, sizeof(*h));
    h->dp.p = *params;
    {
        hipError_t e = hipGetDevice(&h->device);
        if (e != hipSuccess) { delete h; return hip_fail(e, "hipGetDevice"); }
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess) h->num_cus = cus;
    }
#ifdef AC_DIAG_ENV
    {   // measurement switches exist in the diagnostic flavours only (build.py --diag / tools/archive/variant_lib.sh -DAC_DIAG_ENV):
        // the product library's dispatch never depends on the environment
        const char* e = getenv("AIRCRAFT_HIP_NO_PAIR");
        h->no_pair = e && e[0] == '1';
        const char* ea = getenv("AIRCRAFT_HIP_ALL_PAIR");
        h->all_pair = ea && ea[0] == '1';
    }
#endif
#if defined(AC_STAMPS) || defined(AC_CLOCKS)
    h->no_pair = true;  // the diagnostic flavours pass their buffer through `c`; only k_nn_step_sens knows that
#endif
    {   // ticket counter of the persistent kernels' work queue: zero now, and left at zero by every launch that uses it
        hipError_t e = hipMalloc(&h->d_queue, 256);
        if (e == hipSuccess) e = hipMemset(h->d_queue, 0, 256);
        if (e != hipSuccess) { if (h->d_queue) (void)hipFree(h->d_queue); delete h; return hip_fail(e, "hipMalloc(work queue)"); }
    }
    *out = h;
    return AC_OK;
}

int ac_destroy(ac_handle* h) {
    AC_ENTER(h);
    if (!h) return AC_ERR_BAD_ARG;
    if (h->d_blob) (void)hipFree(h->d_blob);
    if (h->d_vblob) (void)hipFree(h->d_vblob);
    if (h->d_queue) (void)hipFree(h->d_queue);
    if (h->d_track) (void)hipFree(h->d_track);
    if (h->d_poly_tab) (void)hipFree(h->d_poly_tab);
    if (h->d_hess_ws) (void)hipFree(h->d_hess_ws);
    if (h->d_hess_ws2) (void)hipFree(h->d_hess_ws2);
    if (h->d_rev_scratch) (void)hipFree(h->d_rev_scratch);
    delete h;
    return AC_OK;
}

int ac_set_params(ac_handle* h, const ac_params* params) {
    if (!h) return AC_ERR_BAD_ARG;
    int rc = check_params(params);
    if (rc != AC_OK) return rc;
    h->dp.p = *params;
    return AC_OK;
}

int ac_set_linear(ac_handle* h, const float* W) {
    if (!h || !W) return AC_ERR_BAD_ARG;
    memcpy(h->dp.linear_W, W, sizeof(float) * 36);
    h->has_linear = true;
    return AC_OK;
}

int ac_set_poly(ac_handle* h, const float* coef, const float* intercept) {
    if (!h || !coef || !intercept) return AC_ERR_BAD_ARG;
    AC_ENTER(h);
    float tab[kPolyTabFloats], grad[6 * 4 * 15], hess[6 * 10 * 5];
    poly_gradient_tables(coef, grad);
    poly_hessian_tables(grad, hess);
    poly_pack_tables(coef, intercept, grad, hess, tab);
    if (!h->d_poly_tab) AC_HIP(hipMalloc((void**)&h->d_poly_tab, sizeof(tab)));
    AC_HIP(hipMemcpy(h->d_poly_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    h->dp.poly_tab = h->d_poly_tab;
    h->has_poly = true;
    return AC_OK;
}

int ac_set_mlp(ac_handle* h, int n_layers, const int* widths, const int* act, const float* const* W,
               const float* const* b, const float* in_mean, const float* in_std, const float* out_mean,
               const float* out_std, int use_mfma) {
    AC_ENTER(h);
    if (!h || !widths || !act || !W || !b || !in_mean || !in_std || !out_mean || !out_std) return AC_ERR_BAD_ARG;
    if (n_layers < 1 || n_layers > AC_MAX_LAYERS) return AC_ERR_BAD_ARG;
    if (widths[0] != 5 || widths[n_layers] != 6) return AC_ERR_BAD_ARG;
    for (int l = 0; l <= n_layers; ++l) {
        if (widths[l] < 1) return AC_ERR_BAD_ARG;
        if (widths[l] > AC_MAX_WIDTH) {
            snprintf(g_err, sizeof(g_err), "MLP width %d > AC_MAX_WIDTH %d", widths[l], AC_MAX_WIDTH);
            return AC_ERR_UNSUPPORTED;
        }
    }
    // Fold every activation-free layer that is not the last into its successor (in float64):
    //   W2 (W1 x + b1) + b2 = (W2 W1) x + (W2 b1 + b2).
    // The device engines then see tanh on every layer but the last (a compile-time fact in the hidden-layer epilogues,
    // which are exposed VALU time) and the reference checkpoint's Linear-Linear-Tanh-Linear net (surrogates/models.py:
    // 114-123) runs as 5-32-6 instead of 5-16-32-6.
    struct HostLayer { int nin, nout, act; std::vector<double> W, b; };
    std::vector<HostLayer> net((size_t)n_layers);
    for (int l = 0; l < n_layers; ++l) {
        if (!W[l] || !b[l]) return AC_ERR_BAD_ARG;
        HostLayer& L = net[(size_t)l];
        L.nin = widths[l]; L.nout = widths[l + 1]; L.act = act[l] ? 1 : 0;
        L.W.assign(W[l], W[l] + (size_t)L.nin * L.nout);
        L.b.assign(b[l], b[l] + L.nout);
    }
    for (size_t l = 0; l + 1 < net.size();) {
        if (net[l].act) { ++l; continue; }
        const HostLayer &A = net[l], &B = net[l + 1];
        HostLayer M;
        M.nin = A.nin; M.nout = B.nout; M.act = B.act;
        M.W.assign((size_t)M.nin * M.nout, 0.0);
        M.b = B.b;
        for (int i = 0; i < B.nout; ++i)
            for (int k = 0; k < B.nin; ++k) {
                const double w = B.W[(size_t)i * B.nin + k];
                M.b[(size_t)i] += w * A.b[(size_t)k];
                for (int j = 0; j < A.nin; ++j) M.W[(size_t)i * M.nin + j] += w * A.W[(size_t)k * A.nin + j];
            }
        net[l] = std::move(M);
        net.erase(net.begin() + (long)l + 1);
    }
    n_layers = (int)net.size();
    std::vector<int> fwidths((size_t)n_layers + 1);
    std::vector<std::vector<float>> fW((size_t)n_layers), fb((size_t)n_layers);
    fwidths[0] = net[0].nin;
    for (int l = 0; l < n_layers; ++l) {
        fwidths[(size_t)l + 1] = net[(size_t)l].nout;
        fW[(size_t)l].assign(net[(size_t)l].W.begin(), net[(size_t)l].W.end());
        fb[(size_t)l].assign(net[(size_t)l].b.begin(), net[(size_t)l].b.end());
    }
    widths = fwidths.data();
    MlpPlan pl;
    memset(&pl, 0, sizeof(pl));
    pl.n_layers = n_layers;
    // Hidden widths are zero-padded to one common multiple of 16 (wt tiles): padded neurons get zero weights
    // and zero bias, so they output act(0) = 0 and feed nothing forward.  Input (5) pads to one tile, output (6)
    // to one tile.  The engine then only meets the static shapes <1,wt>, <wt,wt>, <wt,1>, <1,1>.
    int maxh = 0;
    for (int l = 1; l < n_layers; ++l) maxh = std::max(maxh, widths[l]);
    const int maxt = (maxh + 15) / 16;
    const int wt = maxt <= 2 ? 2 : (maxt <= 4 ? 4 : 8);
    size_t total_floats = 0;
    const int wlt_bytes = ((wt * 384 + 1023) / 1024) * 1024;  // [wt tiles][4 lane groups][6 float4], whole LDS-DMA pieces
    for (int l = 0; l < n_layers; ++l) {
        pl.KT[l] = (l == 0) ? 1 : wt;
        pl.NT[l] = (l == n_layers - 1) ? 1 : wt;
        pl.act[l] = net[(size_t)l].act;  // 1 for every l < n_layers - 1 after the fold
        pl.bytes[l] = pl.NT[l] * pl.KT[l] * 1024 + 1024;  // weights + one 1-KiB bias piece (whole LDS-DMA pieces only)
        // first layer of a multi-layer net: + W0 transposed [5][16*wt] for the MFMA-free tangent slabs
        if (l == 0 && n_layers > 1) pl.bytes[l] += ((5 * wt * 64 + 1023) / 1024) * 1024;
        pl.g_off[l] = (int)total_floats;
        total_floats += (size_t)pl.bytes[l] / 4;
        // last layer of a multi-layer net: + wlt, the per-lane weight pairs of MlpEngine::last_valu — in the global blob only;
        // plan_sens (below) copies [bias][wlt] to LDS, `pl` the fragments and the bias as before
        if (l == n_layers - 1 && n_layers > 1) total_floats += (size_t)wlt_bytes / 4;
    }
    // Width 128 on the matrix cores with one to three hidden products: the transposed hidden blocks of the reverse sweep
    // (k_nn_stage_tensors_rev), top hidden layer first, behind the net's own blocks.
    const int n_hid = n_layers - 2;
    const bool want_rev = wt == 8 && n_hid >= 1 && n_layers + n_hid <= AC_MAX_LAYERS;  // (both flavours of the matrix product)
    size_t rev_off[AC_MAX_LAYERS] = {0};
    const size_t rev_block_floats = (size_t)wt * wt * 256 + 256;  // fragments + a (zero) bias piece: the hidden blocks' size class
    if (want_rev)
        for (int i = 0; i < n_hid; ++i) { rev_off[i] = total_floats; total_floats += rev_block_floats; }
    // Pack: [nt][kt][lane][4] with lane = col + 16 g -> W[16 nt + col][16 kt + 4 g + j]; then the padded bias.
    std::vector<float> blob(total_floats, 0.f);
    if (want_rev)
        for (int i = 0; i < n_hid; ++i) {
            const int l = n_hid - i;  // forward layer l: h_l (nin) -> h_{l+1} (nout); the block multiplies by its transpose
            const int nin = widths[l], nout = widths[l + 1];
            float* dst = blob.data() + rev_off[i];
            for (int nt = 0; nt < wt; ++nt)
                for (int kt = 0; kt < wt; ++kt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int row = 16 * nt + (lane & 15), k = 16 * kt + 4 * (lane >> 4) + j;
                            dst[((size_t)(nt * wt + kt) * 64 + lane) * 4 + j] =
                                (row < nin && k < nout) ? fW[(size_t)l][(size_t)k * nin + row] : 0.f;
                        }
        }
    for (int l = 0; l < n_layers; ++l) {
        const int nin = widths[l], nout = widths[l + 1], KT = pl.KT[l], NT = pl.NT[l];
        float* dst = blob.data() + pl.g_off[l];
        for (int nt = 0; nt < NT; ++nt)
            for (int kt = 0; kt < KT; ++kt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int row = 16 * nt + (lane & 15), k = 16 * kt + 4 * (lane >> 4) + j;
                        dst[((size_t)(nt * KT + kt) * 64 + lane) * 4 + j] =
                            (row < nout && k < nin) ? fW[(size_t)l][(size_t)row * nin + k] : 0.f;
                    }
        float* bd = dst + (size_t)NT * KT * 256;
        for (int i = 0; i < NT * 16; ++i) bd[i] = i < nout ? fb[(size_t)l][(size_t)i] : 0.f;
        if (l == 0 && n_layers > 1) {
            float* wt0 = bd + 256;  // after the 1-KiB bias piece
            for (int j = 0; j < 5; ++j)
                for (int n = 0; n < wt * 16; ++n) wt0[j * wt * 16 + n] = n < nout ? fW[(size_t)l][(size_t)n * nin + j] : 0.f;
        }
        if (l == n_layers - 1 && n_layers > 1) {
            float* wl = bd + 256;  // after the 1-KiB bias piece
            auto wv = [&](int k, int n) { return (k < nout && n < nin) ? fW[(size_t)l][(size_t)k * nin + n] : 0.f; };
            for (int t = 0; t < wt; ++t)
                for (int g = 0; g < 4; ++g)
                    for (int kp = 0; kp < 3; ++kp)
                        for (int rp = 0; rp < 2; ++rp) {
                            float* q = wl + (size_t)(((t * 4 + g) * 6) + 2 * kp + rp) * 4;
                            const int n0 = 16 * t + 4 * g + 2 * rp;
                            q[0] = wv(2 * kp, n0); q[1] = wv(2 * kp + 1, n0); q[2] = wv(2 * kp, n0 + 1); q[3] = wv(2 * kp + 1, n0 + 1);
                        }
        }
    }
    // LDS plan: everything resident if it fits; otherwise the largest layers stream through a 2-slot ring.
    // force_stream: the largest size class goes through the ring even when everything would fit (plan_rev of a net with ONE
    // hidden product: the reverse-sweep kernel sequences its hidden and transposed blocks through the ring by hand)
    auto plan_lds = [&](MlpPlan& pl, bool force_stream = false) -> int {
        const int n_layers = pl.n_layers;  // (plan_rev carries more blocks than the net has layers)
        pl.n_streamed = 0; pl.first_streamed = -1;
        int total = 0;
        for (int l = 0; l < n_layers; ++l) total += pl.bytes[l];
        for (int l = 0; l < n_layers; ++l) pl.lds_off[l] = 0;
        if (total <= kLdsBudget && !force_stream) {
            int off = 0;
            for (int l = 0; l < n_layers; ++l) { pl.lds_off[l] = off; off += pl.bytes[l]; }
            pl.n_streamed = 0; pl.first_streamed = -1; pl.lds_total = off;
        } else {
            int big = 0;
            for (int l = 0; l < n_layers; ++l) big = std::max(big, pl.bytes[l]);
            // stream every layer of the maximal size class; keep the rest resident
            int off = 0, resident = 0;
            for (int l = 0; l < n_layers; ++l) if (pl.bytes[l] < big) resident += pl.bytes[l];
            if (resident + 2 * big > kLdsBudget) {
                snprintf(g_err, sizeof(g_err), "MLP does not fit the LDS plan (%d resident + 2 x %d ring)", resident, big);
                return (int)AC_ERR_UNSUPPORTED;
            }
            pl.first_streamed = -1;
            for (int l = 0; l < n_layers; ++l) {
                if (pl.bytes[l] < big) { pl.lds_off[l] = off; off += pl.bytes[l]; }
                else { pl.lds_off[l] = -1; pl.n_streamed++; if (pl.first_streamed < 0) pl.first_streamed = l; }
            }
            pl.ring_off[0] = off; pl.ring_off[1] = off + big;
            pl.lds_total = off + 2 * big;
        }
        for (int l = 0, i = 0; l < n_layers; ++l)
            if (pl.lds_off[l] < 0) pl.streamed[i++] = l;
        return AC_OK;
    };
    { const int rc = plan_lds(pl); if (rc != AC_OK) return rc; }
    // plan_sens: the first and last layers run on the vector ALUs there (MlpEngine::first_valu / last_valu), so their LDS
    // copies leave the MFMA fragments out: [bias 1 KiB][W0 transposed] and [bias 1 KiB][wlt] (the blob keeps the fragments
    // in front of them for the other kernels)
    MlpPlan ps = pl;
    if (n_layers > 1) {
        const int last = n_layers - 1;
        ps.g_off[0] = pl.g_off[0] + pl.NT[0] * pl.KT[0] * 256;
        ps.bytes[0] = pl.bytes[0] - pl.NT[0] * pl.KT[0] * 1024;
        ps.g_off[last] = pl.g_off[last] + pl.NT[last] * pl.KT[last] * 256;
        ps.bytes[last] = 1024 + wlt_bytes;
        const int rc = plan_lds(ps);
        if (rc != AC_OK) return rc;
    }
    MlpPlan pr = ps;
    bool rev_ok = false;
    if (want_rev) {
        for (int i = 0; i < n_hid; ++i) {
            const int e = n_layers + i;
            pr.KT[e] = wt; pr.NT[e] = wt; pr.act[e] = 0;
            pr.g_off[e] = (int)rev_off[i];
            pr.bytes[e] = (int)(rev_block_floats * 4);
        }
        pr.n_layers = n_layers + n_hid;
        // the kernel expects the hidden and the transposed blocks to stream (one size class) and the edge blocks to stay
        rev_ok = plan_lds(pr, /*force_stream=*/true) == AC_OK && pr.n_streamed == 2 * n_hid && pr.lds_off[0] >= 0 && pr.lds_off[n_layers - 1] >= 0;
    }
    // "MFMA off": weight image of the tiled vector-ALU engine, k-major [K][N] per layer (the last layer transposed [8][K]),
    // hidden widths zero-padded to 32 or 64.  Nets it does not cover (wider than 64, or a single layer after the fold)
    // keep the cross-lane validation path.
    ValuPlan vp;
    memset(&vp, 0, sizeof(vp));
    std::vector<float> vimg;
    bool vok = !use_mfma && n_layers >= 2 && maxh <= 64;
    const int vw = maxh <= 32 ? 32 : 64;
    if (vok) {
        vp.n_layers = n_layers;
        vp.act_last = net[(size_t)n_layers - 1].act;
        size_t off = 0;
        for (int l = 0; l < n_layers; ++l) {
            const bool last = l == n_layers - 1;
            const int K = l == 0 ? 8 : vw, N = last ? 8 : vw;
            vp.w_off[l] = (int)off; off += last ? (size_t)N * (K + 4) : (size_t)K * N;  // last layer: transposed, rows padded
            vp.b_off[l] = (int)off; off += (size_t)N;
        }
        vp.image_floats = (int)((off + 255) / 256 * 256);
        vimg.assign((size_t)vp.image_floats, 0.f);
        for (int l = 0; l < n_layers; ++l) {
            const bool last = l == n_layers - 1;
            const int nin = widths[l], nout = widths[l + 1], K = l == 0 ? 8 : vw, N = last ? 8 : vw;
            float* wd = vimg.data() + vp.w_off[l];
            for (int k = 0; k < nin; ++k)
                for (int nn = 0; nn < nout; ++nn) {
                    const float wv = fW[(size_t)l][(size_t)nn * nin + k];
                    if (last) wd[(size_t)nn * (K + 4) + k] = wv;   // Wt[j][k], row stride K + 4
                    else wd[(size_t)k * N + nn] = wv;        // W[k][n]
                }
            for (int nn = 0; nn < nout; ++nn) vimg[(size_t)vp.b_off[l] + nn] = fb[(size_t)l][(size_t)nn];
        }
        const int lds_need = vp.image_floats * 4 + 4 * 96 * (vw + 4) * 4;
        if (lds_need > kLdsBudget) vok = false;
    }
    float* dv = nullptr;
    if (vok) {
        AC_HIP(hipMalloc(&dv, vimg.size() * sizeof(float)));
        hipError_t ev = hipMemcpy(dv, vimg.data(), vimg.size() * sizeof(float), hipMemcpyHostToDevice);
        if (ev != hipSuccess) { (void)hipFree(dv); return hip_fail(ev, "hipMemcpy(valu image)"); }
    }
    float* d = nullptr;
    {
        hipError_t em = hipMalloc(&d, total_floats * sizeof(float));
        if (em != hipSuccess) { if (dv) (void)hipFree(dv); return hip_fail(em, "hipMalloc(mlp blob)"); }
    }
    hipError_t e = hipMemcpy(d, blob.data(), total_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); if (dv) (void)hipFree(dv); return hip_fail(e, "hipMemcpy(mlp blob)"); }
    if (h->d_vblob) (void)hipFree(h->d_vblob);
    h->d_vblob = dv;
    h->has_vplan = vok;
    h->vplan = vp;
    h->vwidth = vw;
    if (h->d_blob) (void)hipFree(h->d_blob);
    h->d_blob = d;
    h->blob_floats = total_floats;
    h->plan = pl;
    h->plan_sens = ps;
    h->plan_rev = pr;
    h->has_rev = rev_ok;
    h->rev_layers = n_layers;
    h->wt = wt;
    h->use_mfma = use_mfma ? 1 : 0;
    memcpy(h->dp.mlp_in_mean, in_mean, 5 * sizeof(float));
    memcpy(h->dp.mlp_in_std, in_std, 5 * sizeof(float));
    memcpy(h->dp.mlp_out_mean, out_mean, 6 * sizeof(float));
    memcpy(h->dp.mlp_out_std, out_std, 6 * sizeof(float));
    for (int k = 0; k < 6; ++k)
        for (int j = 0; j < 5; ++j) h->dp.mlp_jscale[k][j] = out_std[k] / in_std[j];  // float / float: one IEEE rounding
    h->has_mlp = true;
    return AC_OK;
}

// ---- compute entry points ------------------------------------------------------------------------
static int launch_nn_fwd(ac_handle* h, int op, const float* X, const float* U, float dt, const float* dtp, long n,
                         long blk, float* out, hipStream_t st) {
    bool launched = false;
    if (!h->use_mfma && h->has_vplan) {
        // "MFMA off": the value-only tile of ac_mlp_valu.hpp, 64 units per wave, 256 per workgroup
        const int grid = (int)((n + kBlock - 1) / kBlock);
        const int lds = h->vplan.image_floats * 4 + 4 * 64 * (h->vwidth + 4) * 4;
#define AC_TILED_FWD(W_, OP_)                                                                                      \
        if (h->vwidth == W_ && op == OP_) {                                                                        \
            auto kern = k_nn_fwd_tiled<W_, OP_>;                                                                   \
            int rc_ = set_lds_limit(h, kern, lds);                                                                 \
            if (rc_ != AC_OK) return rc_;                                                                          \
            hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->vplan, h->d_vblob, X, U, dt, dtp, n, blk, out); \
            launched = true;                                                                                       \
        }
        AC_TILED_FWD(32, OP_DERIV) AC_TILED_FWD(32, OP_STEP) AC_TILED_FWD(32, OP_AERO)
        AC_TILED_FWD(64, OP_DERIV) AC_TILED_FWD(64, OP_STEP) AC_TILED_FWD(64, OP_AERO)
#undef AC_TILED_FWD
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
        note_launch(h, op == OP_DERIV ? "k_nn_fwd_tiled<deriv>" : (op == OP_STEP ? "k_nn_fwd_tiled<step>" : "k_nn_fwd_tiled<aero>"), grid,
                    kBlock, lds);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    // Two kernels: k_nn_fwd (16 units per wave, 64 per workgroup) and k_nn_fwd4 (four value slabs per wave, 256 units per
    // workgroup: the per-layer fixed costs are shared, measured 3.4x the time of a 64-unit workgroup for 4x the units).
    // One workgroup per CU is resident, so time goes in whole ROUNDS over the CUs and the last, partly filled round costs
    // as much as a full one.  Split the batch: the units that fill whole rounds of k_nn_fwd4 go there, the remainder
    // (less than one such round) runs in the cheaper rounds of k_nn_fwd; take whichever of {all fwd, all fwd4, split}
    // has the lowest modelled cost (in k_nn_fwd rounds).
    const long cus = h->num_cus > 0 ? h->num_cus : 256;
    const double kFwd4Round = 3.4;
    auto rounds = [&](long units, long per_wg) { return (double)(((units + per_wg - 1) / per_wg + cus - 1) / cus); };
    const long full4 = h->use_mfma ? (n / (256 * cus)) * (256 * cus) : 0;
    const double cost_fwd = rounds(n, 64), cost_fwd4 = h->use_mfma ? kFwd4Round * rounds(n, 256) : 1e30;
    const double cost_split = (full4 > 0 && full4 < n) ? kFwd4Round * (double)(full4 / (256 * cus)) + rounds(n - full4, 64) + 0.05
                                                       : 1e30;
    long n4 = 0;  // units [0, n4) take k_nn_fwd4, [n4, n) take k_nn_fwd
    if (cost_fwd4 <= cost_fwd && cost_fwd4 <= cost_split) n4 = n;
    else if (cost_split < cost_fwd) n4 = full4;
    if (n4 > 0) {
        const int grid4 = (int)((n4 + kBlock - 1) / kBlock);
        const long zero = 0;
#define AC_FWD4_CASE(WT_)                                                                                        \
        if (h->wt == WT_) {                                                                                      \
            if (op == OP_DERIV) { AC_NN_CASE_SENS(WT_, true, (k_nn_fwd4<WT_, OP_DERIV>), grid4, kBlock, X, U, dt, dtp, n4, blk, out, zero) } \
            else if (op == OP_STEP) { AC_NN_CASE_SENS(WT_, true, (k_nn_fwd4<WT_, OP_STEP>), grid4, kBlock, X, U, dt, dtp, n4, blk, out, zero) } \
            else { AC_NN_CASE_SENS(WT_, true, (k_nn_fwd4<WT_, OP_AERO>), grid4, kBlock, X, U, dt, dtp, n4, blk, out, zero) }    \
        }
        AC_FWD4_CASE(2) AC_FWD4_CASE(4) AC_FWD4_CASE(8)
#undef AC_FWD4_CASE
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
        note_launch(h, op == OP_DERIV ? "k_nn_fwd4<deriv>" : (op == OP_STEP ? "k_nn_fwd4<step>" : "k_nn_fwd4<aero>"), grid4,
                    kBlock, (h->use_mfma ? h->plan_sens : h->plan).lds_total);
        AC_HIP(hipGetLastError());
        if (n4 == n) return AC_OK;
        launched = false;
    }
    const int grid = (int)((n - n4 + 63) / 64);
#define AC_FWD_OPS(WT_, MF_)                                                                                   \
    if (op == OP_DERIV) { AC_NN_CASE_SENS(WT_, MF_, (k_nn_fwd<WT_, MF_, OP_DERIV>), grid, kBlock, X, U, dt, dtp, n, blk, out, n4) } \
    else if (op == OP_STEP) { AC_NN_CASE_SENS(WT_, MF_, (k_nn_fwd<WT_, MF_, OP_STEP>), grid, kBlock, X, U, dt, dtp, n, blk, out, n4) } \
    else { AC_NN_CASE_SENS(WT_, MF_, (k_nn_fwd<WT_, MF_, OP_AERO>), grid, kBlock, X, U, dt, dtp, n, blk, out, n4) }
    AC_FWD_OPS(2, true) AC_FWD_OPS(4, true) AC_FWD_OPS(8, true)
    AC_FWD_OPS(2, false) AC_FWD_OPS(4, false) AC_FWD_OPS(8, false)
#undef AC_FWD_OPS
    if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
    if (n4 == 0)
        note_launch(h, op == OP_DERIV ? "k_nn_fwd<deriv>" : (op == OP_STEP ? "k_nn_fwd<step>" : "k_nn_fwd<aero>"), grid,
                    kBlock, (h->use_mfma ? h->plan_sens : h->plan).lds_total);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

static int derivative_impl(ac_handle* h, const float* X, const float* U, long n, long blk, float* Xdot, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;  // empty batch: nothing to do (pointers may be NULL)
    if (!h || !X || !U || !Xdot || n < 0 || blk <= 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    if (n == 0) return AC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN) return launch_nn_fwd(h, OP_DERIV, X, U, 0.f, nullptr, n, blk, Xdot, st);
    const int grid = (int)((n + kBlock - 1) / kBlock);
    AC_LAUNCH_ANALYTIC(k_state_derivative, grid, kBlock, X, U, n, blk, Xdot);
    note_launch(h, "k_state_derivative", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_state_derivative_f32(ac_handle* h, const float* X, const float* U, long n, float* Xdot, void* stream) {
    return derivative_impl(h, X, U, n, n > 0 ? n : 1, Xdot, stream);
}

int ac_shoot_derivative_f32(ac_handle* h, const float* X, const float* U, long B, long H, float* Xdot, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return derivative_impl(h, X, U, B * H, B > 0 ? B : 1, Xdot, stream);
}

static int step_impl(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n, long blk,
                     float* Xn, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !X || !U || !Xn || n < 0 || blk < 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    if (n == 0) return AC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN) return launch_nn_fwd(h, OP_STEP, X, U, dt, dt_per_unit, n, blk, Xn, st);
    const int grid = (int)((n + kBlock - 1) / kBlock);
    AC_LAUNCH_ANALYTIC(k_step, grid, kBlock, X, U, dt, dt_per_unit, n, blk, Xn);
    note_launch(h, "k_step", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_step_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n, float* Xn,
                void* stream) {
    return step_impl(h, X, U, dt, dt_per_unit, n, n, Xn, stream);
}

int ac_shoot_step_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B, long H,
                      float* Xn, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return step_impl(h, X, U, dt, dt_per_unit, B * H, B, Xn, stream);
}

static int deriv_sens_impl(ac_handle* h, const float* X, const float* U, long n, long blk, float* Xdot, float* Fx,
                           float* Fu, void* stream);

// ---- the defect rows of multiple shooting, written by the step / derivative kernels themselves (control/base.py:275-286) ----
namespace {
struct RowsScope {  // the launch takes a copy of h->dp: set for the launches in scope, plain again afterwards
    ac_handle* h;
    RowsScope(ac_handle* h_, int rows, float dt, const float* dtp, float* aux) : h(h_) {
        h->dp.rows = rows; h->dp.rows_dt = dt; h->dp.rows_dt_per_unit = dtp; h->dp.rows_aux = aux;
    }
    ~RowsScope() { h->dp.rows = AC_ROWS_PLAIN; h->dp.rows_dt = 0.f; h->dp.rows_dt_per_unit = nullptr; h->dp.rows_aux = nullptr; }
};
}  // namespace

int ac_shoot_defect_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B, long H,
                        float* R, void* stream) {
    if (!h || B < 0 || H < 0) return AC_ERR_BAD_ARG;
    RowsScope scope(h, AC_ROWS_DEFECT, 0.f, nullptr, nullptr);
    return step_impl(h, X, U, dt, dt_per_unit, B * H, B, R, stream);
}

int ac_shoot_implicit_defect_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                                 long H, float* R, void* stream) {
    if (!h || B < 0 || H < 0) return AC_ERR_BAD_ARG;
    if (h && B * H == 0) return AC_OK;
    if (!X) return AC_ERR_BAD_ARG;
    RowsScope scope(h, AC_ROWS_IMPLICIT, dt, dt_per_unit, nullptr);
    return derivative_impl(h, X + 13 * B, U, B * H, B > 0 ? B : 1, R, stream);  // f at the NEXT nodes, paired with u_k
}

int ac_shoot_implicit_rows_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                               long H, float* R, float* Jnext, float* Ju, float* Jdt, void* stream) {
    if (!h || B < 0 || H < 0) return AC_ERR_BAD_ARG;
    if (h && B * H == 0) return AC_OK;
    if (!X || !Jdt) return AC_ERR_BAD_ARG;
    RowsScope scope(h, AC_ROWS_IMPLICIT, dt, dt_per_unit, Jdt);
    return deriv_sens_impl(h, X + 13 * B, U, B * H, B > 0 ? B : 1, R, Jnext, Ju, stream);
}

int ac_aero_f32(ac_handle* h, const float* X, const float* U, long n, float* out, void* stream) {
    AC_ENTER(h);
    const long blk = n;
    if (h && n == 0) return AC_OK;
    if (!h || !X || !U || !out || n < 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    if (n == 0) return AC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN) return launch_nn_fwd(h, OP_AERO, X, U, 0.f, nullptr, n, blk, out, st);
    const int grid = (int)((n + kBlock - 1) / kBlock);
    AC_LAUNCH_ANALYTIC(k_aero, grid, kBlock, X, U, n, blk, out);
    note_launch(h, "k_aero", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_rollout_f32(ac_handle* h, const float* X0, const float* U, float dt, long B, long H, float* Xout,
                   void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !X0 || !Xout || B < 0 || H < 0 || (H > 0 && !U)) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    if (B == 0) return AC_OK;
    hipStream_t st = (hipStream_t)stream;
    const long roll_units = h->vwidth == 64 ? kRolloutUnits<64> : kRolloutUnits<32>;  // instances per wave of k_nn_rollout_tiled8
    if (h->dp.p.model_kind == AC_MODEL_NN && !h->use_mfma && h->has_vplan && B <= roll_units * 4 * (long)(h->num_cus > 0 ? h->num_cus : 256)) {
        // small batches (at most one wave per SIMD): the latency of the 4 H sequential network evaluations is what counts —
        // the value-only tile with 4 or 8 instances per wave, four waves per workgroup
        const int grid = (int)((B + 4 * roll_units - 1) / (4 * roll_units));
        const int lds = h->vplan.image_floats * 4 + 4 * (int)roll_units * (h->vwidth + 4) * 4;
        bool launched = false;
#define AC_TILED_ROLL8(W_)                                                                                         \
        if (h->vwidth == W_) {                                                                                     \
            auto kern = k_nn_rollout_tiled8<W_>;                                                                   \
            int rc_ = set_lds_limit(h, kern, lds);                                                                 \
            if (rc_ != AC_OK) return rc_;                                                                          \
            hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->vplan, h->d_vblob, X0, U, dt, B, H, Xout);   \
            launched = true;                                                                                       \
        }
        AC_TILED_ROLL8(32) AC_TILED_ROLL8(64)
#undef AC_TILED_ROLL8
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
        note_launch(h, "k_nn_rollout_tiled8", grid, kBlock, lds);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    if (h->dp.p.model_kind == AC_MODEL_NN && !h->use_mfma && h->has_vplan) {
        const int grid = (int)((B + kBlock - 1) / kBlock);
        const int lds = h->vplan.image_floats * 4 + 4 * 64 * (h->vwidth + 4) * 4;
        bool launched = false;
#define AC_TILED_ROLL(W_)                                                                                          \
        if (h->vwidth == W_) {                                                                                     \
            auto kern = k_nn_rollout_tiled<W_>;                                                                    \
            int rc_ = set_lds_limit(h, kern, lds);                                                                 \
            if (rc_ != AC_OK) return rc_;                                                                          \
            hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->vplan, h->d_vblob, X0, U, dt, B, H, Xout);   \
            launched = true;                                                                                       \
        }
        AC_TILED_ROLL(32) AC_TILED_ROLL(64)
#undef AC_TILED_ROLL
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
        note_launch(h, "k_nn_rollout_tiled", grid, kBlock, lds);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    if (h->dp.p.model_kind == AC_MODEL_NN) {
        const long groups = (B + 15) / 16;  // 16 instances per wave-slab
        bool launched = false;
        const int nh = h->plan.n_layers - 2;  // hidden (width x width) layers
        if (h->use_mfma && groups < 4096 && nh >= 1 && nh <= 3) {
            // cooperative with register-resident weights: each wave keeps the fragments of its own output tiles
            const int grid = (int)groups;
            const int lds = 2 * h->wt * 1024;  // double-buffered activation exchange only
#define AC_REG_CASE(WT_, NH_)                                                                                \
            if (h->wt == WT_ && nh == NH_) {                                                                \
                hipLaunchKernelGGL((k_nn_rollout_reg<WT_, NH_>), grid, kBlock, lds, st, h->dp, h->plan, h->d_blob, X0, U, dt, B, H, Xout); \
                launched = true;                                                                            \
            }
            AC_REG_CASE(2, 1) AC_REG_CASE(2, 2) AC_REG_CASE(2, 3) AC_REG_CASE(4, 1) AC_REG_CASE(4, 2) AC_REG_CASE(4, 3)
            AC_REG_CASE(8, 1) AC_REG_CASE(8, 2) AC_REG_CASE(8, 3)
#undef AC_REG_CASE
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
            note_launch(h, "k_nn_rollout_reg", grid, kBlock, lds);
            AC_HIP(hipGetLastError());
            return AC_OK;
        }
        if (h->use_mfma && groups < 4096) {
            // cooperative: one 4-wave workgroup per 16 instances (4x the parallelism per instance)
            const int grid = (int)groups;
            const int lds = h->plan.lds_total + h->wt * 1024;  // + the activation exchange buffer
#define AC_COOP_CASE(WT_)                                                                                   \
            if (h->wt == WT_) {                                                                             \
                auto kern = k_nn_rollout_coop<WT_, true>;                                                   \
                int rc_ = set_lds_limit(h, kern, lds);                                                         \
                if (rc_ != AC_OK) return rc_;                                                               \
                hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->plan, h->d_blob, X0, U, dt, B, H, Xout); \
                launched = true;                                                                            \
            }
            AC_COOP_CASE(2) AC_COOP_CASE(4) AC_COOP_CASE(8)
#undef AC_COOP_CASE
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
            note_launch(h, "k_nn_rollout_coop", grid, kBlock, lds);
            AC_HIP(hipGetLastError());
            return AC_OK;
        }
        const int grid = (int)((groups + 3) / 4);
        AC_NN_CASE_SENS(2, true, (k_nn_rollout<2, true>), grid, kBlock, X0, U, dt, B, H, Xout)
        AC_NN_CASE_SENS(4, true, (k_nn_rollout<4, true>), grid, kBlock, X0, U, dt, B, H, Xout)
        AC_NN_CASE_SENS(8, true, (k_nn_rollout<8, true>), grid, kBlock, X0, U, dt, B, H, Xout)
        AC_NN_CASE_SENS(2, false, (k_nn_rollout<2, false>), grid, kBlock, X0, U, dt, B, H, Xout)
        AC_NN_CASE_SENS(4, false, (k_nn_rollout<4, false>), grid, kBlock, X0, U, dt, B, H, Xout)
        AC_NN_CASE_SENS(8, false, (k_nn_rollout<8, false>), grid, kBlock, X0, U, dt, B, H, Xout)
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
        note_launch(h, "k_nn_rollout", grid, kBlock, (h->use_mfma ? h->plan_sens : h->plan).lds_total);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    const int grid = (int)((B + 63) / 64);
    AC_LAUNCH_ANALYTIC(k_rollout, grid, 64, X0, U, dt, B, H, Xout);
    note_launch(h, "k_rollout", grid, 64, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

static int sens_impl(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n, long blk,
                     float* Xn, float* A, float* Bm, float* c, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !X || !U || !Xn || !A || !Bm || n < 0 || blk < 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    if (n == 0) return AC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN) {
        // MLP: 16 units per wave, 4 waves (64 units) per workgroup, one workgroup resident per CU — time goes in whole
        // rounds over the CUs.  A remainder of at most half a round is given to k_nn_step_sens_pair (two waves per 16
        // units, 32 units per workgroup, ~0.73 of a full workgroup's time) instead of paying a full round for it.
        if (!h->use_mfma && h->has_vplan) {
            // "MFMA off": the tiled v_pk_fma_f32 engine (ac_mlp_valu.hpp)
            bool launched = false;
            // 8 units per wave, eight waves per workgroup (two per SIMD), persistent: at most one workgroup per CU
            const long cus_t = h->num_cus > 0 ? h->num_cus : 256;
            // (a batch smaller than the chip fills whole workgroups: 200 full ones ran cfg2's 12 800 units in 0.120 ms, 256 of
            // six or seven working waves in 0.135)
            const int grid = (int)std::min<long>((n + 63) / 64, cus_t);
            const int lds = h->vplan.image_floats * 4 + 8 * 48 * (h->vwidth + 4) * 4;
#define AC_TILED_SENS(W_)                                                                                          \
            if (h->vwidth == W_) {                                                                                 \
                auto kern = k_nn_step_sens_tiled8<W_>;                                                             \
                int rc_ = set_lds_limit(h, kern, lds);                                                             \
                if (rc_ != AC_OK) return rc_;                                                                      \
                hipLaunchKernelGGL(kern, grid, kBlock8, lds, st, h->dp, h->vplan, h->d_vblob, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c, h->d_queue); \
                launched = true;                                                                                   \
            }
            AC_TILED_SENS(32) AC_TILED_SENS(64)
#undef AC_TILED_SENS
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
            note_launch(h, "k_nn_step_sens_tiled8", grid, kBlock8, lds);
            AC_HIP(hipGetLastError());
            return AC_OK;
        }
        const long cus = h->num_cus > 0 ? h->num_cus : 256;
        const long per_round = 64 * cus;
        long n_main = n, n_pair = 0;
        // (the pair kernel wants >= 1 hidden layer: its no-value role forms the first layer's act' itself, and the barriers of
        // the hidden layers separate the reads of one evaluation's output exchange from the next one's writes)
        if (h->use_mfma && !h->no_pair && h->plan.n_layers >= 3 && h->dp.p.substeps <= 1) {  // (sub-stepped updates: one-wave kernel)
            const long rem = n % per_round;
            if (rem > 0 && rem <= 32 * cus) { n_main = n - rem; n_pair = rem; }
            if (h->all_pair) { n_main = 0; n_pair = n; }
        }
        if (h->use_mfma && h->wt == 2 && h->dp.p.substeps <= 1) {
            // small nets: two persistent workgroups per CU = two waves per SIMD (k_nn_step_sens_w2)
            const int lds = ((h->plan_sens.lds_total + 15) & ~15) + kSensW2AccBytes;
            if (lds <= 80 * 1024) {
                const int grid = (int)std::min<long>((n + 63) / 64, 2 * cus);
                auto kern = k_nn_step_sens_w2<2>;
                int rc_ = set_lds_limit(h, kern, lds);
                if (rc_ != AC_OK) return rc_;
                hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->plan_sens, h->d_blob, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c);
                note_launch(h, "k_nn_step_sens_w2", grid, kBlock, lds);
                AC_HIP(hipGetLastError());
                return AC_OK;
            }
        }
        if (n_main > 0) {
            // persistent: at most one workgroup per CU, each walks the 64-unit tasks b, b + grid, ... (k_nn_step_sens)
#ifndef AC_NO_PERSIST
            const int grid = (int)std::min<long>((n_main + 63) / 64, cus);
#else
            const int grid = (int)((n_main + 63) / 64);  // (A/B flavour: a workgroup per task, as in round 2)
#endif
            bool launched = false;
            AC_NN_CASE_SENS(2, true, (k_nn_step_sens<2, true>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            AC_NN_CASE_SENS(4, true, (k_nn_step_sens<4, true>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            AC_NN_CASE_SENS(8, true, (k_nn_step_sens<8, true>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            AC_NN_CASE_SENS(2, false, (k_nn_step_sens<2, false>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            AC_NN_CASE_SENS(4, false, (k_nn_step_sens<4, false>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            AC_NN_CASE_SENS(8, false, (k_nn_step_sens<8, false>), grid, kBlock, X, U, dt, dt_per_unit, n_main, blk, Xn, A, Bm, c)
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
            note_launch(h, "k_nn_step_sens", grid, kBlock, (h->use_mfma ? h->plan_sens : h->plan).lds_total);
            AC_HIP(hipGetLastError());
        }
        if (n_pair > 0) {
            const int grid_p = (int)((n_pair + 31) / 32);
            const int lds_p = h->plan_sens.lds_total + 2 * h->wt * 1024 + 2 * 16 * 36 * (int)sizeof(float);  // + the pairs' activation and output exchanges
            bool launched = false;
#define AC_PAIR_CASE(WT_)                                                                                              \
            if (h->wt == WT_) {                                                                                        \
                auto kern = k_nn_step_sens_pair<WT_>;                                                                  \
                int rc_ = set_lds_limit(h, kern, lds_p);                                                                  \
                if (rc_ != AC_OK) return rc_;                                                                          \
                hipLaunchKernelGGL(kern, grid_p, kBlock, lds_p, st, h->dp, h->plan_sens, h->d_blob, X, U, dt, dt_per_unit, n, \
                                   blk, Xn, A, Bm, c, n_main);                                                          \
                launched = true;                                                                                       \
            }
            AC_PAIR_CASE(2) AC_PAIR_CASE(4) AC_PAIR_CASE(8)
#undef AC_PAIR_CASE
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
            if (n_main == 0) note_launch(h, "k_nn_step_sens_pair", grid_p, kBlock, lds_p);
            AC_HIP(hipGetLastError());
        }
        return AC_OK;
    }
    // analytic: 4 N units per wave, N = directions per lane of the model (AnalyticSensN), 4 waves per workgroup
    const int upb = 16 * (h->dp.p.model_kind == AC_MODEL_POLY ? AnalyticSensN<AC_MODEL_POLY>::value : AnalyticSensN<AC_MODEL_DEFAULT>::value);
    static_assert(AnalyticSensN<AC_MODEL_DEFAULT>::value == AnalyticSensN<AC_MODEL_LINEAR>::value, "grid size below");
    const int grid_an = (int)((n + upb - 1) / upb);
    // (sub-stepped updates: a kernel of its own, so that the composition code does not set the registers of the common one)
    if (h->dp.p.substeps > 1) {
        switch (h->dp.p.model_kind) {
            case AC_MODEL_LINEAR: hipLaunchKernelGGL((k_step_sens<AC_MODEL_LINEAR, true>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            case AC_MODEL_POLY: hipLaunchKernelGGL((k_step_sens<AC_MODEL_POLY, true>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            case AC_MODEL_QUAD: hipLaunchKernelGGL((k_step_sens<AC_MODEL_QUAD, true>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            default: hipLaunchKernelGGL((k_step_sens<AC_MODEL_DEFAULT, true>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
        }
    } else {
        switch (h->dp.p.model_kind) {
            case AC_MODEL_LINEAR: hipLaunchKernelGGL((k_step_sens<AC_MODEL_LINEAR, false>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            case AC_MODEL_POLY: hipLaunchKernelGGL((k_step_sens<AC_MODEL_POLY, false>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            case AC_MODEL_QUAD: hipLaunchKernelGGL((k_step_sens<AC_MODEL_QUAD, false>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
            default: hipLaunchKernelGGL((k_step_sens<AC_MODEL_DEFAULT, false>), grid_an, kBlock, 0, st, h->dp, X, U, dt, dt_per_unit, n, blk, Xn, A, Bm, c); break;
        }
    }
    note_launch(h, "k_step_sens", grid_an, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_step_sens_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n,
                     float* Xn, float* A, float* Bm, float* c, void* stream) {
    return sens_impl(h, X, U, dt, dt_per_unit, n, n, Xn, A, Bm, c, stream);
}

int ac_shoot_sens_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B, long H,
                      float* Xn, float* A, float* Bm, float* c, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return sens_impl(h, X, U, dt, dt_per_unit, B * H, B, Xn, A, Bm, c, stream);
}

// ---- x_dot with df/dx, df/du; envelope rows; quaternion rows (control/base.py:282-304, control/aircraft.py:44-59) ----
static int deriv_sens_impl(ac_handle* h, const float* X, const float* U, long n, long blk, float* Xdot, float* Fx,
                           float* Fu, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !X || !U || !Xdot || !Fx || !Fu || n < 0 || blk <= 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN && !h->use_mfma && h->has_vplan) {
        const long cus_t = h->num_cus > 0 ? h->num_cus : 256;
        const int grid = (int)std::min<long>((n + 63) / 64, cus_t);  // persistent workgroups + work queue (GroupQueue, ac_mlp_valu.hpp)
        const int lds = h->vplan.image_floats * 4 + 8 * 48 * (h->vwidth + 4) * 4;
        bool launched = false;
#define AC_TILED_DS(W_)                                                                                            \
        if (h->vwidth == W_) {                                                                                     \
            auto kern = k_nn_deriv_sens_tiled8<W_>;                                                                \
            int rc_ = set_lds_limit(h, kern, lds);                                                                 \
            if (rc_ != AC_OK) return rc_;                                                                          \
            hipLaunchKernelGGL(kern, grid, kBlock8, lds, st, h->dp, h->vplan, h->d_vblob, X, U, n, blk, Xdot, Fx, Fu, h->d_queue); \
            launched = true;                                                                                       \
        }
        AC_TILED_DS(32) AC_TILED_DS(64)
#undef AC_TILED_DS
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
        note_launch(h, "k_nn_deriv_sens_tiled8", grid, kBlock8, lds);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    if (h->dp.p.model_kind == AC_MODEL_NN) {
        const int grid = (int)((n + 63) / 64);
        bool launched = false;
        AC_NN_CASE_SENS(2, true, (k_nn_deriv_sens<2, true>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        AC_NN_CASE_SENS(4, true, (k_nn_deriv_sens<4, true>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        AC_NN_CASE_SENS(8, true, (k_nn_deriv_sens<8, true>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        AC_NN_CASE_SENS(2, false, (k_nn_deriv_sens<2, false>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        AC_NN_CASE_SENS(4, false, (k_nn_deriv_sens<4, false>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        AC_NN_CASE_SENS(8, false, (k_nn_deriv_sens<8, false>), grid, kBlock, X, U, n, blk, Xdot, Fx, Fu)
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
        note_launch(h, "k_nn_deriv_sens", grid, kBlock, (h->use_mfma ? h->plan_sens : h->plan).lds_total);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    const int upb = 16 * (h->dp.p.model_kind == AC_MODEL_POLY ? AnalyticSensN<AC_MODEL_POLY>::value : AnalyticSensN<AC_MODEL_DEFAULT>::value);
    const int grid = (int)((n + upb - 1) / upb);
    AC_LAUNCH_ANALYTIC(k_deriv_sens, grid, kBlock, X, U, n, blk, Xdot, Fx, Fu);
    note_launch(h, "k_deriv_sens", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_state_derivative_sens_f32(ac_handle* h, const float* X, const float* U, long n, float* Xdot, float* Fx, float* Fu,
                                 void* stream) {
    return deriv_sens_impl(h, X, U, n, n > 0 ? n : 1, Xdot, Fx, Fu, stream);
}

int ac_shoot_derivative_sens_f32(ac_handle* h, const float* X, const float* U, long B, long H, float* Xdot, float* Fx,
                                 float* Fu, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return deriv_sens_impl(h, X, U, B * H, B > 0 ? B : 1, Xdot, Fx, Fu, stream);
}

static int envelope_impl(ac_handle* h, const float* X, long n, long blk, float* rows, float* Jx, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !X || !rows || n < 0 || blk <= 0) return AC_ERR_BAD_ARG;
    if (h->dp.p.model_kind == AC_MODEL_QUAD) return fail(AC_ERR_UNSUPPORTED, "envelope rows are the fixed-wing plugin's (control/aircraft.py)");
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_envelope<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, X, n, blk, rows, Jx);
    note_launch(h, "k_envelope", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_envelope_f32(ac_handle* h, const float* X, long n, float* rows, float* Jx, void* stream) {
    return envelope_impl(h, X, n, n > 0 ? n : 1, rows, Jx, stream);
}

int ac_shoot_envelope_f32(ac_handle* h, const float* X, long B, long H, float* rows, float* Jx, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return envelope_impl(h, X, B * H, B > 0 ? B : 1, rows, Jx, stream);
}

static EnvelopePenalty to_dev_penalty(const ac_envelope_penalty* p) {
    EnvelopePenalty d;
    static_assert(sizeof(EnvelopePenalty) == sizeof(ac_envelope_penalty), "ac_envelope_penalty layout");
    memcpy(&d, p, sizeof(d));
    return d;
}

int ac_envelope_al_cost_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* lam, long Bl, const float* X, long B,
                            long H, float* cost, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !pen || !X || !cost || B < 0 || H < 0 || (lam && (!(pen->weight > 0.f) || Bl <= 0 || B % Bl != 0))) return AC_ERR_BAD_ARG;
    if (h->dp.p.model_kind == AC_MODEL_QUAD) return fail(AC_ERR_UNSUPPORTED, "envelope rows are the fixed-wing plugin's (control/aircraft.py)");
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_envelope_cost<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, to_dev_penalty(pen), lam, lam ? Bl : 1, X, B, H, cost);
    note_launch(h, "k_envelope_cost", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_envelope_cost_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* cost,
                         void* stream) {
    return ac_envelope_al_cost_f32(h, pen, nullptr, 1, X, B, H, cost, stream);
}

int ac_envelope_al_model_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* lam, const float* X, long B, long H,
                             float* node_glin, float* Hz, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !pen || !X || (!node_glin && !Hz) || B < 0 || H < 0 || (lam && !(pen->weight > 0.f))) return AC_ERR_BAD_ARG;
    if (h->dp.p.model_kind == AC_MODEL_QUAD) return fail(AC_ERR_UNSUPPORTED, "envelope rows are the fixed-wing plugin's (control/aircraft.py)");
    const long n = (H + 1) * B;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_envelope_model<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, to_dev_penalty(pen), lam, X, B, H, node_glin, Hz);
    note_launch(h, "k_envelope_model", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_envelope_model_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* node_glin,
                          float* Hz, void* stream) {
    return ac_envelope_al_model_f32(h, pen, nullptr, X, B, H, node_glin, Hz, stream);
}

int ac_envelope_al_update_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* lam,
                              float* viol_max, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !pen || !X || !lam || B < 0 || H < 0 || !(pen->weight > 0.f)) return AC_ERR_BAD_ARG;
    if (h->dp.p.model_kind == AC_MODEL_QUAD) return fail(AC_ERR_UNSUPPORTED, "envelope rows are the fixed-wing plugin's (control/aircraft.py)");
    const long n = (H + 1) * B;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_envelope_multipliers<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, to_dev_penalty(pen), X, B, H, lam, viol_max);
    note_launch(h, "k_envelope_multipliers", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_quat_rows_f32(ac_handle* h, int mode, const float* X, const float* Xdot, const float* Fx, const float* Fu, long B,
                     long H, float* row, float* Jx, float* Ju, void* stream) {
    AC_ENTER(h);
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    const long n = B * H;
    if (h && n == 0) return AC_OK;
    if (!h || !X || !row || !Jx || !Ju || (mode != 0 && mode != 1)) return AC_ERR_BAD_ARG;
    if (mode == 1 && (!Xdot || !Fx || !Fu)) return AC_ERR_BAD_ARG;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_quat_rows<0>, grid, kBlock, 0, (hipStream_t)stream, X, Xdot, Fx, Fu, n, B, mode, row, Jx, Ju);
    note_launch(h, "k_quat_rows", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

// ---- second-order step sensitivities ---------------------------------------------------------------
// persistent grid of k_nn_stage_tensors_rev: one workgroup per CU at most (its weight plan fills the LDS)
static int rev_grid(const ac_handle* h, long n) {
    const long tasks = (n + 63) / 64, cus = h->num_cus > 0 ? h->num_cus : 256;
    return (int)(tasks < cus ? tasks : cus);
}
// One RK4 sub-step's second-order block with the handle's CURRENT parameters (the caller sets substeps = 1).
static int hess_single(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, const float* Lam,
                       long n, long blk, float* Hout, hipStream_t st) {
    int grid = 0;
    AC_HIP(hipMemsetAsync(Hout, 0, (size_t)n * 441 * sizeof(float), st));
    if (h->dp.p.model_kind == AC_MODEL_NN) {
        // stage tensors (y, J, T at the four RK4 stage points) into the handle's workspace, then the same second-order
        // kernel with the tensor provider.  The workspace is sized by ac_reserve_hess_workspace (a host-side call that may
        // allocate); a compute call never allocates, frees or synchronises.
        if ((size_t)n * kStageFloats > h->hess_ws_floats)
            return fail(AC_ERR_WORKSPACE, "second-order workspace too small: call ac_reserve_hess_workspace(h, n) first");
        const int grid_t = (int)((n + 63) / 64);
        bool launched = false;
#ifndef AC_NO_HESS_REV
        if (h->has_rev && h->wt == 8) {
            // width 128: forward tangents + reverse sweep (12 slab-layer products per hidden layer instead of 29), persistent grid
            const int grid_r = rev_grid(h, n);
            const size_t need_r = (size_t)grid_r * (kBlock / 64) * (size_t)rev_scratch_f32x4(h->wt, h->rev_layers - 2) * 4;
            if (need_r > h->rev_scratch_floats)
                return fail(AC_ERR_WORKSPACE, "second-order workspace too small: call ac_reserve_hess_workspace(h, n) first");
#define AC_REV_LAUNCH(KERN_)                                                                                        \
            {                                                                                                       \
                auto kern = KERN_;                                                                                  \
                int rc_ = set_lds_limit(h, kern, h->plan_rev.lds_total);                                            \
                if (rc_ != AC_OK) return rc_;                                                                       \
                hipLaunchKernelGGL(kern, grid_r, kBlock, h->plan_rev.lds_total, st, h->dp, h->plan_rev, h->d_blob, X, U, dt,    \
                                   dt_per_unit, n, blk, h->rev_layers, h->d_rev_scratch, h->d_hess_ws);            \
            }
            if (!h->use_mfma) AC_REV_LAUNCH((k_nn_stage_tensors_rev3<8, false>))  // the cross-lane validation form of the product
#ifdef AC_HESS_REV6  // (A/B flavour: the six-slab reverse sweep, tools/archive/variant_lib.sh)
            else AC_REV_LAUNCH(k_nn_stage_tensors_rev<8>)
#else
            else AC_REV_LAUNCH((k_nn_stage_tensors_rev3<8, true>))
#endif
#undef AC_REV_LAUNCH
            note_launch(h, "k_nn_stage_tensors_rev", grid_r, kBlock, h->plan_rev.lds_total);
            launched = true;
        } else {
#endif
        AC_NN_CASE(2, true, (k_nn_stage_tensors<2, true, 0>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
        AC_NN_CASE(4, true, (k_nn_stage_tensors<4, true, 0>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
        AC_NN_CASE(8, true, (k_nn_stage_tensors<8, true, 0>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
        AC_NN_CASE(2, false, (k_nn_stage_tensors<2, false, 0>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
        AC_NN_CASE(4, false, (k_nn_stage_tensors<4, false, 0>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
        // width 128 on the matrix cores: the cross pairs between inputs {0, 1} and {3, 4} come from a second launch
        AC_NN_CASE(8, true, (k_nn_stage_tensors<8, true, 1>), grid_t, kBlock, X, U, dt, dt_per_unit, n, blk, h->d_hess_ws)
#ifndef AC_NO_HESS_REV
        }
#endif
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "second-order blocks at width > 64 need the MFMA path (use_mfma = 1)");
        AC_HIP(hipGetLastError());
        launch_hess<AC_MODEL_NN>(h, st, X, U, dt, dt_per_unit, Lam, n, blk, Hout, &grid);
        note_launch(h, "k_step_hess", grid, kBlock, 0);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    switch (h->dp.p.model_kind) {
        case AC_MODEL_LINEAR: launch_hess<AC_MODEL_LINEAR>(h, st, X, U, dt, dt_per_unit, Lam, n, blk, Hout, &grid); break;
        case AC_MODEL_POLY: launch_hess<AC_MODEL_POLY>(h, st, X, U, dt, dt_per_unit, Lam, n, blk, Hout, &grid); break;
        case AC_MODEL_QUAD: launch_hess<AC_MODEL_QUAD>(h, st, X, U, dt, dt_per_unit, Lam, n, blk, Hout, &grid); break;
        default: launch_hess<AC_MODEL_DEFAULT>(h, st, X, U, dt, dt_per_unit, Lam, n, blk, Hout, &grid); break;
    }
    note_launch(h, "k_step_hess", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

// floats per unit of the sub-step composition workspace (ac_hess.hpp, "composition across RK4 sub-steps")
static size_t hess_compose_floats(int ns) { return ns > 1 ? (size_t)ns * (13 + 169 + 91 + 13 + 273) + 13 + 13 + 441 + 1 + 13 : 0; }

static int sens_impl(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n, long blk,
                     float* Xn, float* A, float* Bm, float* c, void* stream);

static int hess_impl(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, const float* Lam,
                     long n, long blk, float* Hout, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !X || !U || !Lam || !Hout || n < 0 || blk <= 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->dp.p.model_kind == AC_MODEL_NN && (size_t)n * kStageFloats > h->hess_ws_floats)
        return fail(AC_ERR_WORKSPACE, "second-order workspace too small: call ac_reserve_hess_workspace(h, n) first");
    const int ns = h->dp.p.substeps;
    if (ns == 1) return hess_single(h, X, U, dt, dt_per_unit, Lam, n, blk, Hout, st);

    // ---- physical_integration_substeps > 1: compose the per-sub-step blocks (ac_hess.hpp) -------------------------------
    if ((size_t)n * hess_compose_floats(ns) > h->hess_ws2_floats)
        return fail(AC_ERR_WORKSPACE, "sub-step composition workspace too small: call ac_reserve_hess_workspace(h, n) first");
    const size_t N = (size_t)n;
    float* w = h->d_hess_ws2;
    float* XS = w;                       w += N * 13 * ns;    // x_0 .. x_{ns-1}   (x_0 is a copy of the caller's X)
    float* JA = w;                       w += N * 169 * ns;
    float* JB = w;                       w += N * 91 * ns;
    float* JC = w;                       w += N * 13 * ns;
    float* XZ = w;                       w += N * 273 * ns;   // XZ[s] = d x_s / dz for s = 1 .. ns-1 (slot 0 unused)
    float* MU0 = w;                      w += N * 13;
    float* MU1 = w;                      w += N * 13;
    float* HS = w;                       w += N * 441;
    float* DT = w;                       w += N;
    float* XN = w;
    const ac_params saved = h->dp.p;
    const float inv_ns = 1.0f / (float)ns;
    const float hdt = dt * inv_ns;
    const float* hdtp = nullptr;
    const int g1 = (int)((n + kBlock - 1) / kBlock);
    if (dt_per_unit) {
        hipLaunchKernelGGL(k_scale_rows<0>, g1, kBlock, 0, st, dt_per_unit, inv_ns, n, DT);
        hdtp = DT;
    }
    rc = AC_OK;
    h->dp.p.substeps = 1;
    // forward: the sub-steps with their first-order blocks, and the chain of state sensitivities
    for (int s = 0; s < ns && rc == AC_OK; ++s) {
        const float* xs = s == 0 ? X : XS + N * 13 * s;
        float* xn = s + 1 < ns ? XS + N * 13 * (s + 1) : XN;
        h->dp.p.normalise = (s == ns - 1) ? saved.normalise : 0;  // q <- q/|q| once, after the last sub-step
        rc = sens_impl(h, xs, U, hdt, hdtp, n, blk, xn, JA + N * 169 * s, JB + N * 91 * s, JC + N * 13 * s, stream);
        if (rc == AC_OK && s + 1 < ns)
            hipLaunchKernelGGL(k_hess_chain<0>, dim3(g1, 21), kBlock, 0, st, JA + N * 169 * s, JB + N * 91 * s, JC + N * 13 * s,
                               s == 0 ? (const float*)nullptr : (const float*)(XZ + N * 273 * s), inv_ns, n, blk,
                               XZ + N * 273 * (s + 1));
    }
    // backward: adjoint of the later sub-steps, the block of each sub-step, its congruence into the caller's variables
    if (rc == AC_OK) rc = hipMemsetAsync(Hout, 0, N * 441 * sizeof(float), st) == hipSuccess ? AC_OK : AC_ERR_HIP;
    const float* mu = Lam;
    for (int s = ns - 1; s >= 0 && rc == AC_OK; --s) {
        const float* xs = s == 0 ? X : XS + N * 13 * s;
        h->dp.p.normalise = (s == ns - 1) ? saved.normalise : 0;
        rc = hess_single(h, xs, U, hdt, hdtp, mu, n, blk, HS, st);
        if (rc != AC_OK) break;
        hipLaunchKernelGGL(k_hess_accum<0>, dim3(g1, 21), kBlock, 0, st, HS,
                           s == 0 ? (const float*)nullptr : (const float*)(XZ + N * 273 * s), inv_ns, n, blk, Hout);
        if (s > 0) {
            float* mo = (mu == MU0) ? MU1 : MU0;
            hipLaunchKernelGGL(k_hess_adjoint<0>, g1, kBlock, 0, st, JA + N * 169 * s, mu, n, blk, mo);
            mu = mo;
        }
    }
    h->dp.p = saved;
    if (rc != AC_OK) return rc;
    note_launch(h, "k_step_hess (composed over sub-steps)", g1, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_reserve_hess_workspace(ac_handle* h, long n) {
    AC_ENTER(h);
    if (!h || n < 0) return AC_ERR_BAD_ARG;
    const size_t need = (size_t)n * kStageFloats;
    if (need > h->hess_ws_floats) {
        if (h->d_hess_ws) (void)hipFree(h->d_hess_ws);
        h->d_hess_ws = nullptr; h->hess_ws_floats = 0;
        AC_HIP(hipMalloc((void**)&h->d_hess_ws, need * sizeof(float)));
        h->hess_ws_floats = need;
    }
    if (h->has_rev && n > 0) {  // the reverse-sweep kernel's layer states: one slot per wave of a persistent grid
        const size_t need_r = (size_t)rev_grid(h, n) * (kBlock / 64) * (size_t)rev_scratch_f32x4(h->wt, h->rev_layers - 2) * 4;
        if (need_r > h->rev_scratch_floats) {
            if (h->d_rev_scratch) (void)hipFree(h->d_rev_scratch);
            h->d_rev_scratch = nullptr; h->rev_scratch_floats = 0;
            AC_HIP(hipMalloc((void**)&h->d_rev_scratch, need_r * sizeof(float)));
            h->rev_scratch_floats = need_r;
        }
    }
    const size_t need2 = (size_t)n * hess_compose_floats(h->dp.p.substeps);  // sized for the handle's CURRENT sub-step count
    if (need2 > h->hess_ws2_floats) {
        if (h->d_hess_ws2) (void)hipFree(h->d_hess_ws2);
        h->d_hess_ws2 = nullptr; h->hess_ws2_floats = 0;
        AC_HIP(hipMalloc((void**)&h->d_hess_ws2, need2 * sizeof(float)));
        h->hess_ws2_floats = need2;
    }
    return AC_OK;
}

int ac_step_hess_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit,
                     const float* lambda, long n, float* Hout, void* stream) {
    return hess_impl(h, X, U, dt, dt_per_unit, lambda, n, n > 0 ? n : 1, Hout, stream);
}

int ac_shoot_hess_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit,
                      const float* lambda, long B, long H, float* Hout, void* stream) {
    if (B < 0 || H < 0) return AC_ERR_BAD_ARG;
    return hess_impl(h, X, U, dt, dt_per_unit, lambda, B * H, B > 0 ? B : 1, Hout, stream);
}

int ac_traj_cost_f32(ac_handle* h, const float* X, long B, long H, const float* goal3, float w_track, float w_goal,
                     float* cost, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !X || !goal3 || !cost || B < 0 || H < 0) return AC_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_traj_cost, grid, kBlock, 0, st, X, B, H, goal3[0], goal3[1], goal3[2], w_track, w_goal, cost);
    note_launch(h, "k_traj_cost", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_best_records_f32(ac_handle* h, const float* cost, const float* X, const float* U, long B, long H, int K,
                        float* rec, void* stream) {
    AC_ENTER(h);
    if (!h || K < 0 || B < 0 || H < 0) return AC_ERR_BAD_ARG;
    if (K == 0) return AC_OK;
    if (!cost || !X || !rec || (H > 0 && !U)) return AC_ERR_BAD_ARG;
    if (K > kMaxBestK) return fail(AC_ERR_BAD_ARG, "ac_best_records_f32: K > 8");
    if (K > B) return fail(AC_ERR_BAD_ARG, "ac_best_records_f32: K exceeds the number of instances");
    hipLaunchKernelGGL(k_best_records, K, kSelBlock, 0, (hipStream_t)stream, cost, X, U, B, H, K, rec);
    note_launch(h, "k_best_records", K, kSelBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_merge_records_f32(ac_handle* h, const float* rec_in, long n, long R, float* rec_out, void* stream) {
    AC_ENTER(h);
    if (!h || n < 0 || R < 1) return AC_ERR_BAD_ARG;
    if (n == 0) return AC_OK;
    if (!rec_in || !rec_out || rec_in == rec_out) return AC_ERR_BAD_ARG;
    if (n > 1024) return fail(AC_ERR_BAD_ARG, "ac_merge_records_f32: more than 1024 rows");
    hipLaunchKernelGGL(k_merge_records, (int)n, 256, 0, (hipStream_t)stream, rec_in, n, R, rec_out);
    note_launch(h, "k_merge_records", (int)n, 256, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_ilqr_accept_f32(ac_handle* h, const float* Jc, const float* J0, const float* Xc, const float* Uc, int n_alpha,
                       long B, long H, float* X, float* U, float* Jout, unsigned char* improved, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !Jc || !J0 || !Xc || !X || !Jout || (H > 0 && (!Uc || !U)) || n_alpha < 1 || n_alpha > 8 || B < 0 || H < 0)
        return AC_ERR_BAD_ARG;
    // Jout is written by the row-0 workgroups while the others still read J0 and Jc to decide whether to copy: an aliased
    // Jout tears the iterate
    if (Jout == J0 || (Jout >= Jc && Jout < Jc + (long)n_alpha * B) || (Jc >= Jout && Jc < Jout + B))
        return fail(AC_ERR_BAD_ARG, "ac_ilqr_accept_f32: Jout must not alias J0 or Jc");
    const long nrows = (H + 1) * 13 + H * 7;
    dim3 grid((unsigned)((B + 255) / 256), (unsigned)((nrows + kAcceptRows - 1) / kAcceptRows));
    hipLaunchKernelGGL(k_ilqr_accept, grid, 256, 0, (hipStream_t)stream, Jc, J0, Xc, Uc, n_alpha, B, H, X, U, Jout, improved);
    note_launch(h, "k_ilqr_accept", (int)grid.x, 256, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

static IlqrCost to_dev_cost(const ac_ilqr_cost* c) {
    IlqrCost d;
    static_assert(sizeof(IlqrCost) == sizeof(ac_ilqr_cost), "ac_ilqr_cost layout");
    memcpy(&d, c, sizeof(d));
    return d;
}

int ac_ilqr_backward_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* X, const float* U, const float* A,
                         const float* Bm, long B, long H, float* K, float* kff, float* dV, void* stream) {
    return ac_ilqr_backward_node_f32(h, cost, nullptr, nullptr, nullptr, X, U, A, Bm, B, H, K, kff, dV, stream);
}

int ac_ilqr_backward_node_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                              const float* node_glin, const float* X, const float* U, const float* A, const float* Bm,
                              long B, long H, float* K, float* kff, float* dV, void* stream) {
    return ac_ilqr_backward_newton_f32(h, cost, node_q, node_xref, node_glin, nullptr, X, U, A, Bm, B, H, K, kff, dV,
                                       stream);
}

int ac_ilqr_costate_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                        const float* node_glin, const float* X, const float* A, long B, long H, float* Lam,
                        void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !cost || !X || !A || !Lam || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    if ((node_q || node_xref || node_glin) && !(node_q && node_xref && node_glin)) return AC_ERR_BAD_ARG;
    const NodeCost nc{node_q, node_xref, node_glin, B};
    const int grid = (int)((B + kBlock - 1) / kBlock);
    if (node_q) hipLaunchKernelGGL(k_ilqr_costate<true>, grid, kBlock, 0, (hipStream_t)stream, to_dev_cost(cost), nc, X, A, B, H, Lam);
    else hipLaunchKernelGGL(k_ilqr_costate<false>, grid, kBlock, 0, (hipStream_t)stream, to_dev_cost(cost), nc, X, A, B, H, Lam);
    note_launch(h, "k_ilqr_costate", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_ilqr_backward_newton_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                                const float* node_glin, const float* Hz, const float* X, const float* U, const float* A,
                                const float* Bm, long B, long H, float* K, float* kff, float* dV, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !cost || !X || !U || !A || !Bm || !K || !kff || !dV || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    if ((node_q || node_xref || node_glin) && !(node_q && node_xref && node_glin)) return AC_ERR_BAD_ARG;
    const NodeCost nc{node_q, node_xref, node_glin, B};
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)B;  // one wave per instance
#define AC_BACKWARD(NODE_, NEWTON_) \
    hipLaunchKernelGGL((k_ilqr_backward<NODE_, NEWTON_>), grid, 64, 0, st, to_dev_cost(cost), nc, X, U, A, Bm, Hz, B, H, K, kff, dV)
    if (node_q && Hz) AC_BACKWARD(true, true);
    else if (node_q) AC_BACKWARD(true, false);
    else if (Hz) AC_BACKWARD(false, true);
    else AC_BACKWARD(false, false);
#undef AC_BACKWARD
    note_launch(h, "k_ilqr_backward", grid, 64, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

// ---- the goal-acquisition loss of the reference's MPC driver (main/control/control.py:44-68; ac_goal.hpp) ----------------
static GoalLoss to_dev_goal(const ac_goal_loss* g) {
    GoalLoss d;
    static_assert(sizeof(GoalLoss) == sizeof(ac_goal_loss), "ac_goal_loss layout");
    memcpy(&d, g, sizeof(d));
    return d;
}

int ac_goal_cost_f32(ac_handle* h, const ac_goal_loss* loss, const float* goal, const float* lam, long Bn, const float* X,
                     const float* U, long B, long H, float* cost_inout, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !loss || !goal || !X || !U || !cost_inout || B < 0 || H < 1 || Bn < 1 || B % Bn != 0) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_goal_cost<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, to_dev_goal(loss), goal, lam, Bn, X, U, B, H,
                       cost_inout);
    note_launch(h, "k_goal_cost", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_goal_model_f32(ac_handle* h, const ac_goal_loss* loss, const float* goal, const float* lam, const float* X,
                      const float* U, long B, long H, float* node_q, float* node_xref, float* node_glin, float* node_uglin,
                      float* Hz, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !loss || !goal || !X || !U || !node_q || !node_xref || !node_glin || !node_uglin || B < 0 || H < 1)
        return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    const long n = (H + 1) * B;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_goal_model<0>, grid, kBlock, 0, (hipStream_t)stream, h->dp, to_dev_goal(loss), goal, lam, X, U, B, H,
                       node_q, node_xref, node_glin, node_uglin, Hz);
    note_launch(h, "k_goal_model", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_goal_multiplier_f32(ac_handle* h, const ac_goal_loss* loss, const float* X, long B, long H, float* lam, float* viol,
                           void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !loss || !X || !lam || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_goal_multiplier<0>, grid, kBlock, 0, (hipStream_t)stream, to_dev_goal(loss), X, B, H, lam, viol);
    note_launch(h, "k_goal_multiplier", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_ilqr_backward_goal_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                              const float* node_glin, const float* node_uglin, const float* Hz, const float* X,
                              const float* U, const float* A, const float* Bm, long B, long H, float* K, float* kff,
                              float* dV, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !cost || !node_q || !node_xref || !node_glin || !Hz || !X || !U || !A || !Bm || !K || !kff || !dV || B < 0 || H < 1)
        return AC_ERR_BAD_ARG;
    NodeCost nc{node_q, node_xref, node_glin, B};
    nc.uglin = node_uglin;
    const int grid = (int)B;  // one wave per instance
    hipLaunchKernelGGL((k_ilqr_backward<true, true>), grid, 64, 0, (hipStream_t)stream, to_dev_cost(cost), nc, X, U, A, Bm, Hz,
                       B, H, K, kff, dV);
    note_launch(h, "k_ilqr_backward", grid, 64, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_ilqr_cost_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* X, const float* U, long B, long H,
                     float* out, void* stream) {
    return ac_ilqr_cost_node_f32(h, cost, nullptr, nullptr, nullptr, B, X, U, B, H, out, stream);
}

int ac_ilqr_cost_node_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                          const float* node_glin, long Bn, const float* X, const float* U, long B, long H, float* out,
                          void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !cost || !X || !U || !out || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    if ((node_q || node_xref || node_glin) && !(node_q && node_xref && node_glin && Bn > 0)) return AC_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    const NodeCost nc{node_q, node_xref, node_glin, Bn > 0 ? Bn : 1};
    if (node_q) hipLaunchKernelGGL(k_ilqr_cost<true>, grid, kBlock, 0, st, to_dev_cost(cost), nc, X, U, B, H, out);
    else hipLaunchKernelGGL(k_ilqr_cost<false>, grid, kBlock, 0, st, to_dev_cost(cost), nc, X, U, B, H, out);
    note_launch(h, "k_ilqr_cost", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

// ---- track + progress terms -------------------------------------------------------------------------
int ac_set_track(ac_handle* h, int n_segments, const float* coef, float length) {
    if (!h || !coef || n_segments < 1 || !(length > 0.f)) return AC_ERR_BAD_ARG;
    AC_ENTER(h);
    if (h->d_track) { (void)hipFree(h->d_track); h->d_track = nullptr; }
    memset(&h->track, 0, sizeof(h->track));  // no dangling device pointer if anything below fails
    // [nseg][3][4] cubics, then one flag per knot: is the knot's float64 value (numpy linspace(0, 1, nseg + 1): i * step, the
    // last one exactly 1) representable in fp32?  Only such a knot can be hit EXACTLY by an fp32 progress value — where the
    // reference's closed segment intervals count the point twice (track_eval, ac_track.hpp).
    std::vector<float> img((size_t)n_segments * 12 + (size_t)n_segments + 1);
    memcpy(img.data(), coef, (size_t)n_segments * 12 * sizeof(float));
    const double step = 1.0 / (double)n_segments;
    for (int i = 0; i <= n_segments; ++i) {
        const double si = (i == n_segments) ? 1.0 : (double)i * step;
        img[(size_t)n_segments * 12 + (size_t)i] = ((double)(float)si == si) ? 1.f : 0.f;
    }
    const size_t bytes = img.size() * sizeof(float);
    {
        hipError_t e = hipMalloc((void**)&h->d_track, bytes);
        if (e != hipSuccess) { h->d_track = nullptr; return hip_fail(e, "hipMalloc(track)"); }
        e = hipMemcpy(h->d_track, img.data(), bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(h->d_track); h->d_track = nullptr; return hip_fail(e, "hipMemcpy(track)"); }
    }
    h->track.coef = h->d_track;
    h->track.knot_exact = h->d_track + (size_t)n_segments * 12;
    h->track.nseg = n_segments;
    h->track.inv_length = 1.f / length;
    const float* last = coef + (size_t)(n_segments - 1) * 12;
    for (int a = 0; a < 3; ++a) h->track.end_pos[a] = last[a * 4] + last[a * 4 + 1] + last[a * 4 + 2] + last[a * 4 + 3];
    return AC_OK;
}

int ac_track_eval_f32(ac_handle* h, const float* s, long n, float* pos, float* tangent, void* stream) {
    AC_ENTER(h);
    if (h && n == 0) return AC_OK;
    if (!h || !s || !pos || !tangent || n < 0) return AC_ERR_BAD_ARG;
    if (!h->d_track) return AC_ERR_NO_MODEL;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_track_eval, grid, kBlock, 0, (hipStream_t)stream, h->track, s, n, pos, tangent);
    note_launch(h, "k_track_eval", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

static MhttWeights to_dev_weights(const ac_mhtt_weights* w) {
    MhttWeights d{};
    static_assert(sizeof(MhttWeights) == sizeof(ac_mhtt_weights), "ac_mhtt_weights layout");
    if (w) memcpy(&d, w, sizeof(d));
    return d;
}

int ac_track_progress_f32(ac_handle* h, const ac_mhtt_weights* weights, const float* X, const float* s0, float dt,
                          long B, long H, int mode, float* S, float* s_dot, float* track_err, float* node_q,
                          float* node_xref, float* node_glin, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !X || !s0 || !S || B < 0 || H < 1 || (mode != 0 && mode != 1)) return AC_ERR_BAD_ARG;
    const bool any = node_q || node_xref || node_glin;
    if (any && !(node_q && node_xref && node_glin && weights)) return AC_ERR_BAD_ARG;
    if (!h->d_track) return AC_ERR_NO_MODEL;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_track_progress, grid, kBlock, 0, (hipStream_t)stream, h->track, to_dev_weights(weights), X, s0,
                       dt, B, H, mode, S, s_dot, track_err, node_q, node_xref, node_glin);
    note_launch(h, "k_track_progress", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_mhtt_loss_f32(ac_handle* h, const ac_mhtt_weights* weights, const float* X, const float* U, const float* S,
                     long B, long H, float* J, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !weights || !X || !U || !S || !J || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    if (!h->d_track) return AC_ERR_NO_MODEL;
    const int grid = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_mhtt_loss, grid, kBlock, 0, (hipStream_t)stream, h->track, to_dev_weights(weights), X, U, S, B,
                       H, J);
    note_launch(h, "k_mhtt_loss", grid, kBlock, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_rollout_policy_f32(ac_handle* h, const ac_ilqr_cost* limits, const float* X0, const float* Xnom, const float* U,
                          const float* K, const float* kff, const float* alphas, int n_alpha, float dt, long B, long H,
                          float* Xout, float* Uout, void* stream) {
    AC_ENTER(h);
    if (h && B == 0) return AC_OK;
    if (!h || !limits || !X0 || !Xnom || !U || !K || !kff || !alphas || !Xout || !Uout || B < 0 || H < 1) return AC_ERR_BAD_ARG;
    if (n_alpha < 1 || n_alpha > 8) return AC_ERR_BAD_ARG;
    int rc = model_ready(h);
    if (rc != AC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    Policy pol;
    pol.Xnom = Xnom; pol.U = U; pol.K = K; pol.kff = kff; pol.B = B;
    pol.alphas.n = n_alpha;
    for (int i = 0; i < 8; ++i) pol.alphas.a[i] = i < n_alpha ? alphas[i] : 0.f;
    memcpy(pol.u_min, limits->u_min, sizeof(pol.u_min));
    memcpy(pol.u_max, limits->u_max, sizeof(pol.u_max));
    pol.dt_row = limits->dt_row;
    if (pol.dt_row >= 7) return AC_ERR_BAD_ARG;
    if (pol.dt_row > 0) {
        // the time row must be one the force model ignores, and its box must keep dt positive
        const bool quad = h->dp.p.model_kind == AC_MODEL_QUAD;
        if (quad ? pol.dt_row < 4 : (pol.dt_row < 3 || pol.dt_row > 5)) return fail(AC_ERR_BAD_ARG, "dt_row must be a control row without effect (aircraft: 3-5, quadrotor: 4-6)");
        if (!(limits->u_min[pol.dt_row] > 0.f)) return fail(AC_ERR_BAD_ARG, "the time row's lower bound (dt_bounds[0]) must be > 0");
    }
    const long Bout = B * n_alpha;
    if (h->dp.p.model_kind == AC_MODEL_NN) {
        if (!h->use_mfma) {
            if (!h->has_vplan) return fail(AC_ERR_UNSUPPORTED, "policy rollout through the MLP with the MFMA path off needs hidden width 32 or 64");
            const long roll_units = h->vwidth == 64 ? kRolloutUnits<64> : kRolloutUnits<32>;
            const int grid = (int)((Bout + 4 * roll_units - 1) / (4 * roll_units));
            const int lds = h->vplan.image_floats * 4 + 4 * (int)roll_units * (h->vwidth + 4) * 4;
            bool launched = false;
#define AC_TILED_POL8(W_)                                                                                          \
            if (h->vwidth == W_) {                                                                                 \
                auto kern = k_nn_rollout_policy_tiled8<W_>;                                                        \
                int rc_ = set_lds_limit(h, kern, lds);                                                             \
                if (rc_ != AC_OK) return rc_;                                                                      \
                hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->vplan, h->d_vblob, pol, X0, dt, Bout, H, Xout, Uout); \
                launched = true;                                                                                   \
            }
            AC_TILED_POL8(32) AC_TILED_POL8(64)
#undef AC_TILED_POL8
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no tiled vector-ALU kernel instance for this hidden width");
            note_launch(h, "k_nn_rollout_policy_tiled8", grid, kBlock, lds);
            AC_HIP(hipGetLastError());
            return AC_OK;
        }
        const int grid = (int)((Bout + 15) / 16);
        bool launched = false;
        const int nh = h->plan.n_layers - 2;
        if (nh >= 1 && nh <= 3) {
            const int ldsr = 2 * h->wt * 1024;
#define AC_POLREG_CASE(WT_, NH_)                                                                                \
            if (h->wt == WT_ && nh == NH_) {                                                                    \
                hipLaunchKernelGGL((k_nn_rollout_policy_reg<WT_, NH_>), grid, kBlock, ldsr, st, h->dp, h->plan, h->d_blob, pol, X0, dt, Bout, H, Xout, Uout); \
                launched = true;                                                                                \
            }
            AC_POLREG_CASE(2, 1) AC_POLREG_CASE(2, 2) AC_POLREG_CASE(2, 3) AC_POLREG_CASE(4, 1) AC_POLREG_CASE(4, 2)
            AC_POLREG_CASE(4, 3) AC_POLREG_CASE(8, 1) AC_POLREG_CASE(8, 2) AC_POLREG_CASE(8, 3)
#undef AC_POLREG_CASE
            if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
            note_launch(h, "k_nn_rollout_policy_reg", grid, kBlock, ldsr);
            AC_HIP(hipGetLastError());
            return AC_OK;
        }
        const int lds = h->plan.lds_total + h->wt * 1024;
#define AC_POL_CASE(WT_)                                                                                        \
        if (h->wt == WT_) {                                                                                     \
            auto kern = k_nn_rollout_policy_coop<WT_, true>;                                                    \
            int rc_ = set_lds_limit(h, kern, lds);                                                                 \
            if (rc_ != AC_OK) return rc_;                                                                       \
            hipLaunchKernelGGL(kern, grid, kBlock, lds, st, h->dp, h->plan, h->d_blob, pol, X0, dt, Bout, H, Xout, Uout); \
            launched = true;                                                                                    \
        }
        AC_POL_CASE(2) AC_POL_CASE(4) AC_POL_CASE(8)
#undef AC_POL_CASE
        if (!launched) return fail(AC_ERR_UNSUPPORTED, "no kernel instance for this MLP width / flavour");
        note_launch(h, "k_nn_rollout_policy_coop", grid, kBlock, lds);
        AC_HIP(hipGetLastError());
        return AC_OK;
    }
    const int grid = (int)((Bout + 63) / 64);
    switch (h->dp.p.model_kind) {
        case AC_MODEL_LINEAR: hipLaunchKernelGGL(k_rollout_policy<AC_MODEL_LINEAR>, grid, 64, 0, st, h->dp, pol, X0, dt, Bout, H, Xout, Uout); break;
        case AC_MODEL_POLY: hipLaunchKernelGGL(k_rollout_policy<AC_MODEL_POLY>, grid, 64, 0, st, h->dp, pol, X0, dt, Bout, H, Xout, Uout); break;
        case AC_MODEL_QUAD: hipLaunchKernelGGL(k_rollout_policy<AC_MODEL_QUAD>, grid, 64, 0, st, h->dp, pol, X0, dt, Bout, H, Xout, Uout); break;
        default: hipLaunchKernelGGL(k_rollout_policy<AC_MODEL_DEFAULT>, grid, 64, 0, st, h->dp, pol, X0, dt, Bout, H, Xout, Uout); break;
    }
    note_launch(h, "k_rollout_policy", grid, 64, 0);
    AC_HIP(hipGetLastError());
    return AC_OK;
}

int ac_hess_workspace(const ac_handle* h, float** ptr, size_t* floats) {
    if (!h) return AC_ERR_BAD_ARG;
    if (ptr) *ptr = h->d_hess_ws;
    if (floats) *floats = h->hess_ws_floats;
    return AC_OK;
}

int ac_last_launch(const ac_handle* h, char* name, size_t len, int* grid, int* block, int* lds_bytes) {
    if (!h) return AC_ERR_BAD_ARG;
    if (name && len) snprintf(name, len, "%s", h->last_name);
    if (grid) *grid = h->last_grid;
    if (block) *block = h->last_block;
    if (lds_bytes) *lds_bytes = h->last_lds;
    return AC_OK;
}

}  // extern "C"

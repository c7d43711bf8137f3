/* abi_demo.c — libaircraft_hip.so from plain C: no Python, no torch, device memory from the HIP runtime.
 *
 *   gcc -O2 examples/abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Laircraft_amd -laircraft_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/aircraft_amd -Wl,-rpath,/opt/rocm/lib -o abi_demo
 *   ./abi_demo            (prints one JSON line; tests/test_gpu_abi.py compares it with the Python host path)
 *
 * A glider with the analytic DefaultModel (dynamics/coefficient_models.py:41-78): n units, one RK4 step of 0.01 s with
 * its Jacobians dF/dx, dF/du, dF/ddt, then a 20-node rollout of unit 0..n-1 under a constant elevator.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aircraft_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_AC(x) do { int r_ = (x); if (r_ != AC_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, r_, ac_last_error()); return 3; } } while (0)

static double frob(const float* a, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += (double)a[i] * a[i];
    return sqrt(s);
}

int main(void) {
    enum { N = 1000, H = 20 };
    /* airframe: data/glider/problem_definition.json:12-24, inertia about the (unshifted) reference point */
    ac_params p;
    memset(&p, 0, sizeof p);
    p.mass = 3.3f; p.S = 0.238f; p.b = 1.75f; p.c = 0.1375f;
    const double Ixx = 0.155, Iyy = 0.16, Izz = 0.3, Ixz = 0.01;
    const double I[9] = {Ixx, 0, Ixz, 0, Iyy, 0, Ixz, 0, Izz};
    const double det = Ixx * Izz - Ixz * Ixz;
    const double Ii[9] = {Izz / det, 0, -Ixz / det, 0, 1.0 / Iyy, 0, -Ixz / det, 0, Ixx / det};
    for (int i = 0; i < 9; ++i) { p.inertia[i] = (float)I[i]; p.inertia_inv[i] = (float)Ii[i]; }
    p.rudder_moment_arm = 0.5f; p.epsilon = 1e-6f; p.gravity[2] = 9.81f;
    p.substeps = 1; p.normalise = 1; p.model_kind = AC_MODEL_DEFAULT;

    ac_handle* h = NULL;
    CHECK_AC(ac_create(&p, &h));

    /* deterministic inputs: level flight at 40..60 m/s with small attitude and rate variations */
    float* X = (float*)malloc(sizeof(float) * 13 * N);
    float* U = (float*)calloc(7 * N, sizeof(float));
    for (int i = 0; i < N; ++i) {
        const float t = (float)i / N;
        const float phi = 0.2f * sinf(7.f * t), th = 0.05f * cosf(5.f * t);
        float* x = X + i;
        x[0 * N] = 10.f * t; x[1 * N] = -5.f * t; x[2 * N] = -200.f;
        x[3 * N] = 40.f + 20.f * t; x[4 * N] = 1.f - 2.f * t; x[5 * N] = 0.5f;
        /* small-angle quaternion (xyzw), normalised */
        float q[4] = {0.5f * phi, 0.5f * th, 0.1f * t, 1.f};
        const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) x[(6 + k) * N] = q[k] / nq;
        x[10 * N] = 0.1f * phi; x[11 * N] = 0.05f; x[12 * N] = -0.02f * t;
        U[0 * N + i] = 2.f * sinf(3.f * t); U[1 * N + i] = -1.f + t; U[2 * N + i] = 0.5f * t;
    }
    float *dX, *dU, *dXn, *dA, *dB, *dc, *dUh, *dTraj;
    CHECK_HIP(hipMalloc((void**)&dX, sizeof(float) * 13 * N));
    CHECK_HIP(hipMalloc((void**)&dU, sizeof(float) * 7 * N));
    CHECK_HIP(hipMalloc((void**)&dXn, sizeof(float) * 13 * N));
    CHECK_HIP(hipMalloc((void**)&dA, sizeof(float) * 169 * N));
    CHECK_HIP(hipMalloc((void**)&dB, sizeof(float) * 91 * N));
    CHECK_HIP(hipMalloc((void**)&dc, sizeof(float) * 13 * N));
    CHECK_HIP(hipMalloc((void**)&dUh, sizeof(float) * H * 7 * N));
    CHECK_HIP(hipMalloc((void**)&dTraj, sizeof(float) * (H + 1) * 13 * N));
    CHECK_HIP(hipMemcpy(dX, X, sizeof(float) * 13 * N, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dU, U, sizeof(float) * 7 * N, hipMemcpyHostToDevice));
    for (int k = 0; k < H; ++k) CHECK_HIP(hipMemcpy(dUh + (size_t)k * 7 * N, dU, sizeof(float) * 7 * N, hipMemcpyDeviceToDevice));

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_AC(ac_step_sens_f32(h, dX, dU, 0.01f, NULL, N, dXn, dA, dB, dc, st));
    CHECK_AC(ac_rollout_f32(h, dX, dUh, 0.01f, N, H, dTraj, st));
    CHECK_HIP(hipStreamSynchronize(st));

    float* Xn = (float*)malloc(sizeof(float) * 13 * N);
    float* A = (float*)malloc(sizeof(float) * 169 * N);
    float* B = (float*)malloc(sizeof(float) * 91 * N);
    float* c = (float*)malloc(sizeof(float) * 13 * N);
    float* last = (float*)malloc(sizeof(float) * 13 * N);
    CHECK_HIP(hipMemcpy(Xn, dXn, sizeof(float) * 13 * N, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(A, dA, sizeof(float) * 169 * N, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(B, dB, sizeof(float) * 91 * N, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(c, dc, sizeof(float) * 13 * N, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(last, dTraj + (size_t)H * 13 * N, sizeof(float) * 13 * N, hipMemcpyDeviceToHost));

    char arch[64] = "";
    ac_device_arch(arch, sizeof arch);
    printf("{\"version\": \"%s\", \"arch\": \"%s\", \"n\": %d, \"x_next_unit0\": [", ac_version(), arch, N);
    for (int r = 0; r < 13; ++r) printf("%s%.9g", r ? ", " : "", Xn[(size_t)r * N]);
    printf("], \"frob_xn\": %.9g, \"frob_A\": %.9g, \"frob_B\": %.9g, \"frob_c\": %.9g, \"frob_rollout_end\": %.9g}\n",
           frob(Xn, 13 * N), frob(A, 169 * N), frob(B, 91 * N), frob(c, 13 * N), frob(last, 13 * N));

    CHECK_AC(ac_destroy(h));
    hipFree(dX); hipFree(dU); hipFree(dXn); hipFree(dA); hipFree(dB); hipFree(dc); hipFree(dUh); hipFree(dTraj);
    free(X); free(U); free(Xn); free(A); free(B); free(c); free(last);
    return 0;
}

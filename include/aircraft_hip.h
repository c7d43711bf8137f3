/*
 * aircraft_hip.h — C ABI of libaircraft_hip.so: the MI355X (gfx950) implementation of the
 * AIrcraft MPC-rollout hot path (6-DoF RK4 step through an aerodynamic-coefficient model,
 * plus first-order step sensitivities).
 *
 * Boundary: these entry points replace the CasADi `ca.Function` objects that the reference's
 * control layer consumes — `system.state_update` / `system.state_derivative`
 * (reference: src/aircraft/control/base.py:187-190) — and the `ca.jacobian` of them
 * (control/aircraft.py:85-95, control/base.py:279-280, 314-315).  A reference-side binding is a
 * ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - state  x = [p_ned(3), v_ned(3), q_frd_ned(4, xyzw), omega_frd(3)]   (dynamics/base.py:84-106)
 *   - control u = [aileron, elevator, rudder (deg), thrust(3), flaps]       (dynamics/aircraft.py:143-166)
 *   - batched arrays are component-major float32 DEVICE buffers: X is [13][n], U is [7][n] — the
 *     reference's own column-mapped call convention (`(13,N)` matrices, main/control/control.py:63),
 *     which is also the coalesced layout on the GPU.  A "unit" is one (x_k, u_k) pair: one MPC
 *     instance at one shooting node.
 *   - the caller owns every buffer; the library never allocates outputs.  The handle owns device
 *     copies of the coefficient-model data.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     does no allocation and no synchronisation, and is therefore hipGraph-capturable.
 *   - return value: AC_OK (0) or a negative ac_status; nothing is thrown across the ABI.  NaNs in the
 *     inputs propagate to the outputs unchanged (the reference's callers test np.isnan,
 *     main/dynamics/dynamics.py:108).
 *   - thread-compatible per handle: no global mutable state.  A handle does own mutable DEVICE state that its
 *     launches share, so launches of ONE handle must be ordered on ONE stream (or otherwise serialised) where they use it:
 *       * the ticket counter of the persistent vector-ALU kernels of an MLP set with use_mfma = 0 (ac_step_sens_f32,
 *         ac_shoot_sens_f32, ac_*_derivative_sens_f32 on hidden widths <= 64): 0 between launches, drawn from by every
 *         wave of a launch and reset by the wave that draws the last ticket.  Two launches of one handle running
 *         concurrently on two streams would skip or duplicate unit groups; use one handle per stream;
 *       * the second-order workspace (ac_reserve_hess_workspace; see ac_step_hess_f32).
 *     ac_destroy / ac_set_* free or replace device memory: never while a stream of the process is capturing.
 */
#ifndef AIRCRAFT_HIP_H
#define AIRCRAFT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_NUM_STATES 13   /* SixDOF.num_states,   dynamics/base.py:101 */
#define AC_NUM_CONTROLS 7  /* Aircraft.num_controls, dynamics/aircraft.py:162 */
#define AC_AERO_ROWS 22
#define AC_MAX_LAYERS 8
#define AC_MAX_WIDTH 128   /* widest MLP layer the register-resident MFMA engine supports */

typedef enum ac_status {
    AC_OK = 0,
    AC_ERR_BAD_ARG = -1,       /* NULL pointer, negative size, unknown enum */
    AC_ERR_HIP = -2,           /* a HIP runtime call failed; see ac_last_error() */
    AC_ERR_UNSUPPORTED = -3,   /* e.g. MLP wider than AC_MAX_WIDTH */
    AC_ERR_NO_MODEL = -4,      /* model_kind needs data that was never set */
    AC_ERR_NO_DEVICE = -5,     /* no gfx950 device visible */
    AC_ERR_WORKSPACE = -6      /* a handle-owned workspace is too small: call the matching ac_reserve_* first */
} ac_status;

/* Registry keys of COEFF_MODEL_REGISTRY, dynamics/coefficient_models.py:32-37 */
typedef enum ac_model_kind { AC_MODEL_DEFAULT = 0, AC_MODEL_LINEAR = 1, AC_MODEL_NN = 2, AC_MODEL_POLY = 3,
                             /* not a coefficient model: the Quadrotor plugin (dynamics/quadrotor.py:8-54) — body force
                              * (0, 0, sum T) and rotor moments straight from control rows 0-3 (the four thrusts);
                              * control rows 4-6 are ignored and their Jacobian columns are zero. */
                             AC_MODEL_QUAD = 4 } ac_model_kind;

/* Constants of one airframe + integration options.
 * Mirrors AircraftOpts / SixDOFOpts / AircraftConfiguration
 * (dynamics/aircraft.py:22-38, dynamics/base.py:9-14, utils.py:201-215).
 * inertia / inertia_inv are computed by the host in float64 from Ixx..Ixz, mass and com
 * (dynamics/aircraft.py:168-187, dynamics/base.py:139-144) and rounded once. */
typedef struct ac_params {
    float mass, S, b, c;
    float inertia[9];
    float inertia_inv[9];
    float com[3];
    float rudder_moment_arm;
    float epsilon;
    float gravity[3];
    int substeps;       /* physical_integration_substeps (>=1) */
    int normalise;      /* SixDOF.normalise: q <- q/|q| once after the last sub-step */
    int stall_scaling;  /* AircraftOpts.stall_scaling */
    int model_kind;     /* ac_model_kind */
} ac_params;

typedef struct ac_handle ac_handle;

/* Lifetime.  ac_create binds the handle to the CURRENT HIP device. */
int ac_create(const ac_params* params, ac_handle** out);
int ac_destroy(ac_handle* h);
/* Replace the scalar parameters (e.g. after the driver overrides aircraft.com, control.py:172). */
int ac_set_params(ac_handle* h, const ac_params* params);

/* Coefficient-model data (HOST pointers; copied).
 * linear: W[6][6] row-major, columns [qbar, alpha, beta, aileron, elevator, 1]   (coefficient_models.py:80-89)
 * poly:   coef[6][34], intercept[6]; sklearn PolynomialFeatures(3) order over (alpha, beta, aileron, elevator)
 *                                                                                 (coefficient_models.py:106-133)
 * mlp:    n_layers Linear layers; W[l] is [widths[l+1]][widths[l]] row-major (torch layout); act[l] = 0 identity,
 *         1 tanh after layer l; widths[0] must be 5, widths[n_layers] must be 6; input/output scalers as in
 *         ScaledModel (surrogates/models.py:101-155).  use_mfma = 1: the v_mfma_f32_16x16x4_f32 engines.  use_mfma = 0
 *         ("MFMA off", BASELINE configs[1]): the register-tiled v_pk_fma_f32 engine on the vector ALUs for hidden widths
 *         <= 64 and at least two layers after the fold (step, derivative, getters, rollout, policy rollout, step +
 *         sensitivities, df/dx); wider nets, single-layer nets and the second-order path of this flavour use the
 *         cross-lane validation form or are refused (see the entry points).
 *         An activation-free layer that is not the last is folded into its successor on the host, in float64
 *         (W2 (W1 x + b1) + b2 = (W2 W1) x + W2 b1 + b2): the reference checkpoint's Linear-Linear-Tanh-Linear net
 *         (surrogates/models.py:114-123) runs as 5-32-6.  Same function, fewer layers; results differ from a
 *         layer-by-layer fp32 evaluation only by rounding. */
int ac_set_linear(ac_handle* h, const float* W);
int ac_set_poly(ac_handle* h, const float* coef, const float* intercept);
int ac_set_mlp(ac_handle* h, int n_layers, const int* widths, const int* act, const float* const* W,
               const float* const* b, const float* in_mean, const float* in_std, const float* out_mean,
               const float* out_std, int use_mfma);

/* x_dot = f(x,u)                                   — SixDOF.state_derivative, dynamics/base.py:385-406 */
int ac_state_derivative_f32(ac_handle* h, const float* X, const float* U, long n, float* Xdot, void* stream);

/* x+ = F(x,u,dt): `substeps` RK4 steps of dt/substeps — SixDOF.state_update, dynamics/base.py:450-480.
 * dt_per_unit: NULL -> every unit uses dt; else [n] per-unit step (dt_k = 1/progress_k^2, control/base.py:276). */
int ac_step_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n,
                float* Xn, void* stream);

/* Rollout X[k+1] = F(X[k], U[k], dt), k = 0..H-1    — Controller.initialise, main/control/control.py:72-93;
 * MHTT.initialise, control/moving_horizon.py:203-213.
 * X0 [13][B]; U [H][7][B]; Xout [H+1][13][B] with Xout[0] = X0. */
int ac_rollout_f32(ac_handle* h, const float* X0, const float* U, float dt, long B, long H, float* Xout,
                   void* stream);

/* Step + first-order sensitivities (the multiple-shooting defect Jacobian blocks):
 *   Xn [13][n], A = dF/dx [13][13][n], Bm = dF/du [13][7][n], c = dF/d(dt) [13][n] (c may be NULL)
 *                                                   — ca.jacobian(state_update, .), control/aircraft.py:85-95 */
int ac_step_sens_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long n,
                     float* Xn, float* A, float* Bm, float* c, void* stream);

/* Multiple-shooting forms: every (instance b, node k) pair of a trajectory is an independent unit
 * (x_{k+1} - F(x_k, u_k, dt_k) = 0, control/base.py:275-286, built for k = 0..N-1 by setup(), :423-443).
 * Buffers are rollout-shaped and used IN PLACE (no transpose): X [>=H][13][B] (nodes 0..H-1 are read),
 * U [H][7][B], dt_per_unit NULL or [H][B]; outputs Xn [H][13][B] (= F(x_k,u_k): subtract from X[k+1] for the
 * defect), A [H][13][13][B], Bm [H][13][7][B], c [H][13][B] (c may be NULL). */
int ac_shoot_step_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                      long H, float* Xn, void* stream);
int ac_shoot_sens_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                      long H, float* Xn, float* A, float* Bm, float* c, void* stream);

/* x_dot = f(x, u) with its Jacobians Fx = df/dx [13][13][n], Fu = df/du [13][7][n] — ca.jacobian(state_derivative, .):
 * the implicit defect row  x_{k+1} - x_k - dt_k f(x_{k+1}, u_k)  (control/base.py:282-284), the Baumgarte row (:288-304)
 * and the LQR wrapper (dynamics/base.py:51-52) differentiate f, not the step.  df/dp = 0 and df/d(thrust) = 0 exactly.
 * ac_shoot_derivative_sens_f32 reads H nodes of rollout-shaped X [>=H][13][B], U [H][7][B] in place (pass X + 13 B to
 * evaluate at the NEXT nodes x_1..x_H with the controls u_0..u_{H-1}) and writes Xdot [H][13][B], Fx [H][13][13][B],
 * Fu [H][13][7][B]. */
int ac_state_derivative_sens_f32(ac_handle* h, const float* X, const float* U, long n, float* Xdot, float* Fx, float* Fu,
                                 void* stream);
/* x_dot alone on H nodes of rollout-shaped X [>=H][13][B], U [H][7][B] in place -> Xdot [H][13][B]: the residual of the
 * implicit defect row (control/base.py:282-284) needs f(x_{k+1}, u_k) but no Jacobian. */
int ac_shoot_derivative_f32(ac_handle* h, const float* X, const float* U, long B, long H, float* Xdot, void* stream);
int ac_shoot_derivative_sens_f32(ac_handle* h, const float* X, const float* U, long B, long H, float* Xdot, float* Fx,
                                 float* Fu, void* stream);

/* Envelope rows of AircraftControl.state_constraint (control/aircraft.py:44-59) and their state Jacobian:
 *   rows[0] = v_rel . v_rel   (bounded 20^2 .. 100^2)        rows[1] = beta   (|.| <= 10 deg)
 *   rows[2] = alpha           (|.| <= 20 deg)                rows[3] = z = x[2]   (< 0)
 * rows [4][n]; Jx = d rows / dx [4][13][n] (may be NULL).  The rows do not depend on the control.  Shooting form:
 * X [>=H][13][B] in place -> rows [H][4][B], Jx [H][4][13][B].  Fixed-wing plugin only (AC_ERR_UNSUPPORTED for the quadrotor). */
int ac_envelope_f32(ac_handle* h, const float* X, long n, float* rows, float* Jx, void* stream);
int ac_shoot_envelope_f32(ac_handle* h, const float* X, long B, long H, float* rows, float* Jx, void* stream);

/* The envelope as a SOFT constraint of the batched solver sweep (build-side, SURVEY §8 f1: the state_constraint role of
 * control/aircraft.py:44-59 — IPOPT enforces the rows, the iLQR sweep penalises their violation):
 *   penalty(x) = weight * sum_r max(0, g_r(x) - hi_r)^2 + max(0, lo_r - g_r(x))^2,   g = (|v_rel|^2, beta, alpha, z)
 * ac_envelope_cost_f32:   cost[b] += sum_{k=0..H} penalty(x_k)            X [H+1][13][B], cost [B]
 * ac_envelope_model_f32:  its quadratic model around X for the backward pass: node_glin [H+1][13][B] += 2 w sum_r viol_r grad g_r,
 *                         Hz [H][21][21][B] (x,x block) += 2 w sum_{r violated} grad g_r grad g_r'  (either of node_glin, Hz may be NULL). */
typedef struct ac_envelope_penalty { float lo[4], hi[4], weight; } ac_envelope_penalty;
int ac_envelope_cost_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* cost,
                         void* stream);
int ac_envelope_model_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* node_glin,
                          float* Hz, void* stream);
/* The envelope as a HARD constraint of the sweep, augmented-Lagrangian form: multipliers lam [H+1][8][B] (device; rows 0-3
 * the upper, 4-7 the lower bounds of the four envelope rows; all >= 0; start at zero) per node, row and instance —
 *   L_A = w sum_r max(0, g_r - hi_r + lam_hi / 2w)^2 - (lam_hi / 2w)^2 + max(0, lo_r - g_r + lam_lo / 2w)^2 - (lam_lo / 2w)^2
 * i.e. the penalty above on bounds shifted inwards by lam / 2w.  ac_envelope_al_cost_f32 / _model_f32 are the two calls above
 * with that shift (lam NULL = the plain penalty; a cost batch wider than Bl — the line-search candidates — reads the
 * multipliers of instance b % Bl); ac_envelope_al_update_f32 is the first-order multiplier update at the iterate X,
 *   lam_hi <- max(0, lam_hi + 2 w (g - hi)),   lam_lo <- max(0, lam_lo + 2 w (lo - g)),
 * and (viol_max [B] non-NULL, zeroed by the caller) reports each instance's largest bound excess over all nodes and rows,
 * relative to the row's range where it has one.  The weight must be > 0. */
int ac_envelope_al_cost_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* lam, long Bl, const float* X, long B,
                            long H, float* cost, void* stream);
int ac_envelope_al_model_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* lam, const float* X, long B, long H,
                             float* node_glin, float* Hz, void* stream);
int ac_envelope_al_update_f32(ac_handle* h, const ac_envelope_penalty* pen, const float* X, long B, long H, float* lam,
                              float* viol_max, void* stream);

/* Quaternion rows of ControlProblem.state_constraint on H nodes of X [>=H][13][B] (control/base.py:285-304):
 *   mode 0 ('constraint'):  row = q . q - 1                                                        (:285-286)
 *   mode 1 ('baumgarte'):   row = 2 a phi_dot + b^2 phi,  phi = q . q - 1,  phi_dot = 2 q . q_dot,  a = b = 2   (:288-304)
 * row [H][B], Jx = d row / dx [H][13][B], Ju = d row / du [H][7][B].  Mode 1 needs Xdot, Fx, Fu of the same (x, u) from
 * ac_shoot_derivative_sens_f32 (NULL allowed in mode 0). */
int ac_quat_rows_f32(ac_handle* h, int mode, const float* X, const float* Xdot, const float* Fx, const float* Fu, long B,
                     long H, float* row, float* Jx, float* Ju, void* stream);

/* Second-order step sensitivities: Hout [21][21][n] = sum_i lambda_i d2F_i / dz dz over z = (x[13], u[7], dt) — the
 * block the defect rows x_{k+1} - F(x_k, u_k, dt_k) (control/base.py:279-280) contribute to IPOPT's `nlp_hess_l`
 * (the reference's largest time sink, todo.md:102).  lambda [13][n] (device) are the multipliers of the 13 rows of F.
 * Exact second-order forward mode of the same fp32 arithmetic as ac_step_f32.  Rows/columns of p (0-2) and of controls
 * without effect are zero.  All force models.  The MLP surrogate evaluates its second-derivative tensor with the MFMA
 * engine (hidden width 128 with one to three hidden 128 x 128 products: forward tangents + a reverse sweep through the
 * transposed blocks, csrc/ac_hess_rev.hpp, which also keeps per-wave layer states in a handle-owned scratch of
 * CUs x 4 x (1 + 6 hidden products) x 8 KiB; otherwise one slab per derivative, csrc/ac_hess_nn.hpp)
 * into a handle-owned workspace of n*504 floats: size it once with ac_reserve_hess_workspace(h, n_max) — a
 * host-side call that may allocate — BEFORE the first compute call (and before capturing a hipGraph); the compute calls
 * themselves never allocate, free or synchronise and return AC_ERR_WORKSPACE when the workspace is too small.  The
 * workspace is one buffer per handle: second-order calls of one handle must be ordered on one stream (or use one handle
 * per stream).  physical_integration_substeps > 1 (the reference's default is 10, dynamics/base.py:12, 463-474) composes the
 * per-sub-step blocks: d2/dz2 = sum_s T_s' H_s T_s with T_s = d(x_{s-1}, u, dt/ns)/dz carried by the first-order chain and the
 * multipliers pulled back through the later sub-steps (mu_{s-1} = A_s' mu_s); it needs (559 ns + 481) n floats more of the
 * same reserved workspace, sized for the handle's sub-step count at the time of ac_reserve_hess_workspace.
 * AC_ERR_UNSUPPORTED for an MLP wider than 64 with use_mfma = 0 unless it has one to three hidden 128 x 128 products (the
 * reverse-sweep kernel has a cross-lane flavour, the slab-per-derivative kernel at that width has not).  Hout is zero-filled by the call (hipMemsetAsync on `stream`) before the active block is
 * written.  ac_shoot_hess_f32 reads rollout-shaped X [H(+1)][13][B], U [H][7][B], lambda [H][13][B] in place and writes
 * Hout [H][21][21][B]. */
int ac_step_hess_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit,
                     const float* lambda, long n, float* Hout, void* stream);
/* Size the second-order workspaces for n units: the MLP path's stage tensors (and the reverse sweep's scratch) and, when the
 * handle integrates with more than one RK4 sub-step, the composition buffers (hipFree + hipMalloc when one must grow: implicit device synchronisation,
 * not capturable); a no-op when both are already large enough.  Call again after changing `substeps`. */
int ac_reserve_hess_workspace(ac_handle* h, long n);
int ac_shoot_hess_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit,
                      const float* lambda, long B, long H, float* Hout, void* stream);

/* The defect rows of multiple shooting, written by the step / derivative kernels themselves
 *                                                  — ControlProblem.state_constraint, control/base.py:275-286.
 * X [H+1][13][B] (every node), U [H][7][B]; dt / dt_per_unit [H][B] as in ac_shoot_step_f32.  R [H][13][B]:
 *   ac_shoot_defect_f32             R_k = x_{k+1} - F(x_k, u_k, dt_k)                   ('explicit', :279-280)
 *   ac_shoot_implicit_defect_f32    R_k = x_{k+1} - x_k - dt_k f(x_{k+1}, u_k)          ('implicit', :282-284)
 *   ac_shoot_implicit_rows_f32      the same R with its Jacobian blocks: Jnext [H][13][13][B] = d R_k / d x_{k+1} =
 *                                   I - dt_k Fx, Ju [H][13][7][B] = d R_k / d u_k = -dt_k Fu, Jdt [H][13][B] =
 *                                   d R_k / d dt_k = -f  (d R_k / d x_k = -I is not stored). */
int ac_shoot_defect_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B, long H,
                        float* R, void* stream);
int ac_shoot_implicit_defect_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                                 long H, float* R, void* stream);
int ac_shoot_implicit_rows_f32(ac_handle* h, const float* X, const float* U, float dt, const float* dt_per_unit, long B,
                               long H, float* R, float* Jnext, float* Ju, float* Jdt, void* stream);

/* Getters, out [22][n]: v_frd_rel(3), airspeed, alpha, beta, qbar, coefficients(6), forces_frd(3), moments_frd(3),
 * phi, theta, psi (Euler angles of q)        — dynamics/base.py:147-278, dynamics/aircraft.py:255-330, base.py:179-195 */
int ac_aero_f32(ac_handle* h, const float* X, const float* U, long n, float* out, void* stream);

/* Trajectory cost + best-K selection used by the sharded random-restart driver (build-side).
 * cost[b] = sum_k |p_k - goal|^2 * w_track  +  w_goal * |p_H - goal|^2 ; X is a rollout [H+1][13][B] (device),
 * goal3 is a HOST pointer to 3 floats, cost [B] (device). */
int ac_traj_cost_f32(ac_handle* h, const float* X, long B, long H, const float* goal3, float w_track, float w_goal,
                     float* cost, void* stream);

/* Second half of K6: the per-rank best-K select and the record pack in ONE launch.  cost [B] (device; NaN counts as
 * +inf, equal costs are ordered by instance index), X [H+1][13][B], U [H][7][B] rollout-shaped, 1 <= K <= min(8, B);
 * rec [K][1 + (H+1)*13 + H*7] = rows [cost, X(H+1,13) node-major, U(H,7) node-major] in ascending cost — the block each
 * rank hands to the path's one all-gather (8 056 B per row at H = 100). */
int ac_best_records_f32(ac_handle* h, const float* cost, const float* X, const float* U, long B, long H, int K,
                        float* rec, void* stream);
/* The gathered rows of all ranks sorted by cost: rec_in [n][R] -> rec_out [n][R] (n <= 1024 = K * world, column 0 is
 * the key; NaN -> +inf; stable).  rec_out must not alias rec_in. */
int ac_merge_records_f32(ac_handle* h, const float* rec_in, long n, long R, float* rec_out, void* stream);

/* ---- batched iLQR / Gauss-Newton sweep on (F, A, B)  (build-side) ---------------------------
 * Plays the `loss` and control-limit roles of ControlProblem (control/base.py:323-337, control/aircraft.py:29-41,
 * main/control/control.py:35-70) for B independent instances; the reference itself hands the NLP to IPOPT.
 *   J = sum_k 1/2 (x_k - x_ref)' diag(q) (x_k - x_ref) + 1/2 u_k' diag(r) u_k + 1/2 (x_N - x_goal)' diag(qf) (x_N - x_goal) */
typedef struct ac_ilqr_cost {
    float q[13], qf[13], r[7];
    float x_ref[13], x_goal[13];
    float u_min[7], u_max[7];  /* control box (aileron/elevator/rudder limits, control/aircraft.py:29-41) */
    float reg;                 /* Levenberg term on Quu */
    float u_lin[7];            /* + sum_k u_lin . u_k: linear control cost (w_time on the time row: the reference's time loss,
                                  main/control/control.py:44, 66-67) */
    int dt_row;                /* <= 0: fixed time.  r > 0: time is a decision variable per node (control/base.py:276, 339-385:
                                  dt_k = 1/progress_k^2 or progress_k^2, bounded by dt_bounds): control row dt_row — one the force
                                  model ignores (aircraft 3-5, quadrotor 4-6) — carries dt_k; u_min/u_max[dt_row] are dt_bounds
                                  (lower > 0).  The policy rollout integrates node k with the clipped u_k[dt_row]; the caller
                                  linearises with dt_per_unit = that row and puts c = dF/d(dt) into column dt_row of Bm. */
} ac_ilqr_cost;

/* Backward (Riccati) pass along X [H+1][13][B], U [H][7][B] with A [H][13][13][B], Bm [H][13][7][B] from
 * ac_shoot_sens_f32.  Outputs K [H][7][13][B], kff [H][7][B], dV [2][B] (expected-improvement terms). */
int ac_ilqr_backward_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* X, const float* U, const float* A,
                         const float* Bm, long B, long H, float* K, float* kff, float* dV, void* stream);
/* cost[b] of a trajectory batch X [H+1][13][B], U [H][7][B] */
int ac_ilqr_cost_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* X, const float* U, long B, long H,
                     float* out, void* stream);
/* Closed-loop rollout with a parallel line search: output instance o = a*B + b uses step alphas[a] (HOST array,
 * n_alpha <= 8):  u = clip(U_k[b] + alpha kff_k[b] + K_k[b] (x - Xnom_k[b])),  x+ = F(x, u, dt).
 * X0 [13][B]; Xout [H+1][13][n_alpha*B]; Uout [H][7][n_alpha*B].  MLP with use_mfma = 0: hidden widths 32 and 64
 * (AC_ERR_UNSUPPORTED otherwise). */
int ac_rollout_policy_f32(ac_handle* h, const ac_ilqr_cost* limits, const float* X0, const float* Xnom,
                          const float* U, const float* K, const float* kff, const float* alphas, int n_alpha, float dt,
                          long B, long H, float* Xout, float* Uout, void* stream);

/* Line-search acceptance, one launch: candidate a of instance b is column a*B + b of Xc [H+1][13][n_alpha*B],
 * Uc [H][7][n_alpha*B] (the layout ac_rollout_policy_f32 writes) with cost Jc [n_alpha*B]; J0 [B] is the cost of the
 * current iterate X [H+1][13][B], U [H][7][B].  Per instance: best = min_a Jc (non-finite costs never win, ties -> the
 * lowest a); if best < J0 the candidate's columns replace the iterate's IN PLACE.  Jout [B] = the accepted cost,
 * improved [B] = 0/1 bytes (may be NULL).  Jout must not alias J0 or Jc (AC_ERR_BAD_ARG): the workgroups that copy the
 * rows re-read both while Jout is being written. */
int ac_ilqr_accept_f32(ac_handle* h, const float* Jc, const float* J0, const float* Xc, const float* Uc, int n_alpha,
                       long B, long H, float* X, float* U, float* Jout, unsigned char* improved, void* stream);

/* Per-node, per-instance state cost for the two calls above (device arrays [H+1][13][Bn], node H = terminal):
 *   l_k(x) = 1/2 sum_j node_q[k][j] (x_j - node_xref[k][j])^2 + node_glin[k] . x
 * replacing the q, qf, x_ref, x_goal fields of the cost struct; r, limits and reg still come from it.  A batch wider than Bn
 * (the line-search candidates) reads instance b % Bn.  Produced by ac_track_progress_f32 for the MHTT loss. */
int ac_ilqr_backward_node_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                              const float* node_glin, const float* X, const float* U, const float* A, const float* Bm,
                              long B, long H, float* K, float* kff, float* dV, void* stream);
int ac_ilqr_cost_node_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                          const float* node_glin, long Bn, const float* X, const float* U, long B, long H, float* out,
                          void* stream);

/* The goal-acquisition loss of the reference's MPC driver — Controller.loss, main/control/control.py:44-68:
 *   J = w_goal |p_xy(N) - goal|^2 + w_rate sum_k sum_i l0(u_{k+1,i} - u_{k,i}; eps_rate) + w_height (z_N - z_0)^2
 *       - (w_speed / N) sum_{k<N} v_rel(x_k).v_rel(x_k) + w_vx v_x(N) + w_vyz (v_y(N)^2 + v_z(N)^2),   l0(d) = 1 - exp(-d^2/eps),
 *   subject to v_x(N) < vx_max (augmented Lagrangian, weight w_al, one multiplier per instance; w_al = 0: not enforced).
 * Reference values: w_goal 1000, w_rate 100, eps_rate 1e-2, w_height 1, w_speed 1/100, w_vx = vel_param * 1000, w_vyz 1000,
 * vx_max -2; the time term 10000 T is the linear control cost of the time row (ac_ilqr_cost::u_lin).
 *   ac_goal_cost_f32        cost_inout [B] += J of every column of X [H+1][13][B], U [H][7][B] (EXACT value); column o
 *                           belongs to instance o % Bn: goal [2][Bn], lam [Bn] (NULL = zeros)
 *   ac_goal_model_f32       the convex quadratic model around the iterate for the backward pass: WRITES node_q / node_xref /
 *                           node_glin [H+1][13][B] and node_uglin [H][7][B] (control gradient of the rate term, neighbours
 *                           held at the iterate), ADDS the rate term's Gauss-Newton curvature to the (u, u) diagonal of
 *                           Hz [H][21][21][B] (zero it or fill it with ac_shoot_hess_f32 / ac_envelope_al_model_f32 first)
 *   ac_goal_multiplier_f32  lam <- max(0, lam + 2 w_al (v_x(N) - vx_max)); viol [B] (may be NULL) = the excess before it
 *   ac_ilqr_backward_goal_f32  ac_ilqr_backward_newton_f32 with node_uglin added to Q_u */
typedef struct ac_goal_loss {
    float w_goal, w_rate, eps_rate, w_height, w_speed, w_vx, w_vyz;
    float vx_max, w_al;
    int time_row;   /* control row carrying dt_k, excluded from the rate term; <= 0: none */
} ac_goal_loss;
int ac_goal_cost_f32(ac_handle* h, const ac_goal_loss* loss, const float* goal, const float* lam, long Bn, const float* X,
                     const float* U, long B, long H, float* cost_inout, void* stream);
int ac_goal_model_f32(ac_handle* h, const ac_goal_loss* loss, const float* goal, const float* lam, const float* X,
                      const float* U, long B, long H, float* node_q, float* node_xref, float* node_glin, float* node_uglin,
                      float* Hz, void* stream);
int ac_goal_multiplier_f32(ac_handle* h, const ac_goal_loss* loss, const float* X, long B, long H, float* lam, float* viol,
                           void* stream);
int ac_ilqr_backward_goal_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                              const float* node_glin, const float* node_uglin, const float* Hz, const float* X,
                              const float* U, const float* A, const float* Bm, long B, long H, float* K, float* kff,
                              float* dV, void* stream);

/* Exact-Hessian (Newton / SQP) variant of the sweep, for the force models ac_shoot_hess_f32 supports:
 *   ac_ilqr_costate_f32        Lam [H][13][B]: multipliers of the defect rows at the current iterate,
 *                              Lam[H-1] = grad l_N(x_N), Lam[k-1] = grad l_k(x_k) + A_k' Lam[k]
 *   ac_shoot_hess_f32          Hz [H][21][21][B] = sum_i Lam[k][i] d2F_i/dz dz      (the nlp_hess_l blocks)
 *   ac_ilqr_backward_newton_f32  the backward pass with Hz's (x,x), (u,x), (u,u) blocks added to Qxx, Qux, Quu
 * node_q/node_xref/node_glin and Hz may be NULL (then this is ac_ilqr_backward_f32). */
int ac_ilqr_costate_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                        const float* node_glin, const float* X, const float* A, long B, long H, float* Lam,
                        void* stream);
int ac_ilqr_backward_newton_f32(ac_handle* h, const ac_ilqr_cost* cost, const float* node_q, const float* node_xref,
                                const float* node_glin, const float* Hz, const float* X, const float* U, const float* A,
                                const float* Bm, long B, long H, float* K, float* kff, float* dV, void* stream);

/* ---- track + progress terms of the moving-horizon track tracker   ---------------------------
 * Track: piecewise cubic Hermite curve over s in [0,1] through the sampled Dubins path
 * (control/initialisation.py:782-851).  `coef` is a HOST array [n_segments][3][4]: per segment and axis the cubic
 * c0 + c1 t + c2 t^2 + c3 t^3 in the local parameter t = s*n_segments - segment (the host layer computes the
 * Hermite slopes in float64, aircraft_amd/control/track.py); `length` is DubinsInitialiser.length()
 * (initialisation.py:738-758).  The handle keeps a device copy. */
int ac_set_track(ac_handle* h, int n_segments, const float* coef, float length);
/* pos [3][n], tangent [3][n] = track.eval(s), track.eval_tangent(s) for s [n] (device) */
int ac_track_eval_f32(ac_handle* h, const float* s, long n, float* pos, float* tangent, void* stream);

typedef struct ac_mhtt_weights { /* control/moving_horizon.py:47-55 */
    float w_tracking, w_progress, w_progress_rate, w_backward, w_terminal_align, w_low_velocity, w_control;
} ac_mhtt_weights;

/* Progress along the track for all nodes of all instances: X [H+1][13][B], s0 [B] -> S [H+1][B],
 * s_dot [H][B] (may be NULL), track_err [H][B] = |p_k - track(s_k)|^2 (may be NULL).
 *   mode 0: s_{k+1} = clip(s_k + s_dot dt, 0, 1)                    the initial guess, moving_horizon.py:216-233
 *   mode 1: s_{k+1} = clip(s_k + s_dot dt + 0.05 delta_s, 0, 1)     the constraint row at its bound, :161-168
 * If node_q/node_xref/node_glin [H+1][13][B] are non-NULL (all three), the diagonal-quadratic model of the MHTT
 * loss around this progress sequence is written for ac_ilqr_*_node_f32 (weights may be NULL otherwise). */
int ac_track_progress_f32(ac_handle* h, const ac_mhtt_weights* weights, const float* X, const float* s0, float dt,
                          long B, long H, int mode, float* S, float* s_dot, float* track_err, float* node_q,
                          float* node_xref, float* node_glin, void* stream);
/* J [B] = MHTT.loss (moving_horizon.py:44-105) of X [H+1][13][B], U [H][7][B] with progress S [H+1][B] */
int ac_mhtt_loss_f32(ac_handle* h, const ac_mhtt_weights* weights, const float* X, const float* U, const float* S,
                     long B, long H, float* J, void* stream);

/* Diagnostics */
const char* ac_last_error(void);     /* thread-local text of the last failing HIP call */
const char* ac_version(void);
int ac_device_arch(char* buf, size_t len);  /* gcnArchName of the current device */
/* Device pointer and size (floats) of the handle's second-order workspace (tests poison it with NaNs to prove that every
 * element a second-order call reads was written by that call). */
int ac_hess_workspace(const ac_handle* h, float** ptr, size_t* floats);
/* Name + launch geometry of the kernel the last call on this handle dispatched (for profiling). */
int ac_last_launch(const ac_handle* h, char* name, size_t len, int* grid, int* block, int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif

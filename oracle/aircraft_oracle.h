/*
 * aircraft_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * The oracle is a float64 restatement of the reference's MPC-rollout hot path
 * (wgrosche/AIrcraft: src/aircraft/dynamics/base.py, dynamics/aircraft.py,
 * dynamics/coefficient_models.py, surrogates/models.py).  It exists so that the
 * HIP kernels in aircraft_amd/csrc can be checked for parity.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (aircraft_amd) never links, imports or falls back to it.
 *
 * Parity pin: tests/test_oracle.py replays tests/golden/simulation_h5.npz
 * (the reference's own stored rollout, data/trajectories/simulation.h5) through
 * oracle_rollout_f64 and requires <=1e-12 max rel. error on all 40 stored
 * states, and checks the MLP against golden vectors produced by importing the
 * reference's ScaledModel (tests/golden/make_fixtures.py).
 *
 * All arrays are host memory, float64.  Batched arrays are component-major
 * ("SoA"): X is [13][n], U is [7][n] — the reference's column-mapped calling
 * convention (main/control/control.py:63, control/aircraft.py:92-94).
 */
#ifndef AIRCRAFT_ORACLE_H
#define AIRCRAFT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_MODEL_DEFAULT = 0, ORACLE_MODEL_LINEAR = 1, ORACLE_MODEL_NN = 2, ORACLE_MODEL_POLY = 3,
       ORACLE_MODEL_QUAD = 4 /* the Quadrotor plugin, dynamics/quadrotor.py:8-54: thrusts in control rows 0-3 */ };
enum { ORACLE_MAX_LAYERS = 8 };

typedef struct oracle_params {
    /* airframe: dynamics/aircraft.py:123-141, data/glider/problem_definition.json:12-24 */
    double mass, S, b, c;
    double Ixx, Iyy, Izz, Ixz;
    double com[3];            /* aero_centre_offset / driver override (main/control/control.py:172) */
    double rudder_moment_arm; /* utils.py:215 */
    double epsilon;           /* dynamics/base.py:11 */
    double gravity[3];        /* dynamics/base.py:13 */
    int substeps;             /* physical_integration_substeps, dynamics/base.py:12 */
    int normalise;            /* SixDOF.normalise, control/base.py:182-185 */
    int stall_scaling;        /* AircraftOpts.stall_scaling, dynamics/aircraft.py:30 */
    int model_kind;           /* ORACLE_MODEL_* ; registry keys coefficient_models.py:32-37 */
    /* LinearModel: W[6][6] row-major, columns = [qbar, alpha, beta, aileron, elevator, 1] */
    double linear_W[36];
    /* PolynomialModel: 6 cubic fits over (alpha, beta, aileron, elevator), 34 terms each */
    double poly_coef[6 * 34];
    double poly_intercept[6];
    /* NeuralModel: generic MLP. weights[l] is [widths[l+1]][widths[l]] row-major (torch layout) */
    int mlp_n_layers;
    int mlp_widths[ORACLE_MAX_LAYERS + 1];
    int mlp_act[ORACLE_MAX_LAYERS]; /* 0 = identity, 1 = tanh after layer l */
    const double* mlp_W[ORACLE_MAX_LAYERS];
    const double* mlp_b[ORACLE_MAX_LAYERS];
    double mlp_in_mean[5], mlp_in_std[5], mlp_out_mean[6], mlp_out_std[6];
} oracle_params;

/* x_dot = f(x,u)            — SixDOF.state_derivative, dynamics/base.py:385-406 */
int oracle_state_derivative_f64(const oracle_params* p, const double* X, const double* U, long n, double* Xdot);

/* x+ = F(x,u,dt)            — SixDOF.state_update, dynamics/base.py:450-480.
 * dt: pointer to n per-unit values, or to a single value when dt_is_scalar != 0. */
int oracle_step_f64(const oracle_params* p, const double* X, const double* U, const double* dt, int dt_is_scalar,
                    long n, double* Xn);

/* X[k+1] = F(X[k], U[k], dt), k = 0..H-1 — Controller.initialise, main/control/control.py:72-93.
 * X0 [13][B]; U [H][7][B]; Xout [H+1][13][B] (Xout[0] = X0). */
int oracle_rollout_f64(const oracle_params* p, const double* X0, const double* U, double dt, long B, long H,
                       double* Xout);

/* x+, A = dF/dx [13][13][n], Bm = dF/du [13][7][n], c = dF/ddt [13][n]
 *   — ca.jacobian(state_update, .): control/aircraft.py:85-95, control/base.py:279-280.
 * Exact forward-mode AD of the same arithmetic; d sign/dx = 0.  Any of A/Bm/c may be NULL. */
int oracle_step_sens_f64(const oracle_params* p, const double* X, const double* U, const double* dt,
                         int dt_is_scalar, long n, double* Xn, double* A, double* Bm, double* c);

/* Getters: out [22][n] = v_frd_rel(3), airspeed, alpha, beta, qbar, coefficients(6), forces_frd(3),
 * moments_frd(3), phi, theta, psi   — dynamics/base.py:147-278, aircraft.py:255-330, base.py:179-195 */
int oracle_aero_f64(const oracle_params* p, const double* X, const double* U, long n, double* out);
/* x_dot with Fx = df/dx [13][13][n], Fu = df/du [13][7][n] (either may be NULL) — ca.jacobian(state_derivative, .):
 * control/base.py:282-304 (implicit and Baumgarte rows), dynamics/base.py:51-52 (LQR). */
int oracle_state_derivative_sens_f64(const oracle_params* p, const double* X, const double* U, long n, double* Xdot,
                                     double* Fx, double* Fu);
/* envelope rows [4][n] = (|v_rel|^2, beta, alpha, z) of control/aircraft.py:44-59 and Jx [4][13][n] (may be NULL) */
int oracle_envelope_f64(const oracle_params* p, const double* X, long n, double* rows, double* Jx);

/* Coefficient MLP alone: inputs [n][5] row-major -> outputs [n][6], jac [n][6][5] (may be NULL)
 *   — ScaledModel.forward, surrogates/models.py:143-155 (no rudder term). */
int oracle_mlp_f64(const oracle_params* p, const double* inputs, long n, double* outputs, double* jac);

/* Number of OpenMP threads the batched entry points will use (1 if built without OpenMP). */
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif

"""NumPy float64 restatement of the receding-horizon closed loop (TEST INFRASTRUCTURE, NOT PRODUCT).

Follows the loop of the reference's moving-horizon driver, /root/reference/main/mhe/mhtt.py:79-124:

    while progress < 1:
        sol = mhtt.solve()                                       # :80   (IPOPT there; a fixed number of iLQR iterations here)
        state   = sol.value(mhtt.state)[:, :-overlap]            # :86   nodes 0 .. N - overlap   (N + 1 - overlap columns)
        control = sol.value(mhtt.control)[:, :-overlap]          # :88   nodes 0 .. N - overlap - 1
        full_state = hcat(full_state, state[:, 1:])              # :91-97 executed history: nodes 1 .. N - overlap
        initial_state = state[:, -1]                             # :108  node N - overlap becomes the next x0
        guess = mhtt.initialise(initial_state, progress)         # :110  rollout from x0 (control/moving_horizon.py:203-213:
                                                                 #       zero controls; 'shift' re-uses the solved tail)
        mhtt.set_initial_from_array(guess); update_parameters    # :111-112

for B instances at once.  The solver inside is the NumPy iLQR sweep of ilqr_oracle.py (the reference hands the NLP to
IPOPT, which is out of scope); the dynamics are the C++ float64 oracle.  Plain Python loops: small cases only.
Checks aircraft_amd/control/moving_horizon.py::RecedingHorizon."""
import numpy as np

import ilqr_oracle as io


def linearise(orc, X, U, dt):
    """A (H,13,13,B), Bm (H,13,7,B) of every node (the defect-row Jacobian blocks, control/base.py:279-280)."""
    H, _, B = U.shape
    A = np.empty((H, 13, 13, B)); Bm = np.empty((H, 13, 7, B))
    for k in range(H):
        _, A[k], Bm[k], _ = orc.step_sens(X[k], U[k], dt)
    return A, Bm


def iterate(orc, cost, x0, X, U, alphas, dt):
    """One iLQR iteration in place on (X, U); returns (cost (B,), improved (B,), chosen alpha index (B,)).
    Same acceptance rule as aircraft_amd/control/ilqr.py::ILQR.iterate: the best line-search candidate replaces the
    iterate only where it lowers the cost; non-finite candidates never win."""
    H, _, B = U.shape
    A, Bm = linearise(orc, X, U, dt)
    K, kff, _ = io.backward(cost, X, U, A, Bm)
    Xc, Uc = io.forward(orc, cost, x0, X, U, K, kff, alphas, dt)
    with np.errstate(all="ignore"):
        Jc = io.cost(cost, Xc, Uc).reshape(len(alphas), B)
    Jc = np.where(np.isfinite(Jc), Jc, np.inf)
    J0 = io.cost(cost, X, U)
    idx = Jc.argmin(axis=0)
    best = Jc[idx, np.arange(B)]
    improved = best < J0
    col = idx * B + np.arange(B)
    X[:] = np.where(improved[None, None, :], Xc[:, :, col], X)
    U[:] = np.where(improved[None, None, :], Uc[:, :, col], U)
    return np.where(improved, best, J0), improved, idx


def receding_horizon(orc, cost, x0, U0, overlap, iterations, cycles, alphas, dt, warm_start="shift"):
    """Returns (history (cycles*keep + 1, 13, B) of executed states, final x0, final U, per-cycle alpha choices)."""
    H, _, B = U0.shape
    keep = H - overlap
    x0 = np.array(x0, dtype=np.float64); U = np.array(U0, dtype=np.float64)
    X = orc.rollout(x0, U, dt)
    hist = [x0.copy()[None]]
    choices = []
    for _ in range(cycles):
        ch = []
        for _ in range(iterations):
            _, improved, idx = iterate(orc, cost, x0, X, U, alphas, dt)
            ch.append(np.where(improved, idx, -1))
        choices.append(ch)
        hist.append(X[1 : keep + 1].copy())          # mhtt.py:91-97: nodes 1 .. N - overlap were executed
        x0 = X[keep].copy()                          # mhtt.py:108
        if warm_start == "shift":
            tail = U[keep:].copy()
            U[:overlap] = tail
            U[overlap:] = tail[-1:]
        else:                                        # control/moving_horizon.py:204: zero controls
            U[:] = 0.0
        X = orc.rollout(x0, U, dt)                   # mhtt.py:110: the guess is the rollout from the new x0
    return np.concatenate(hist), x0, U, choices

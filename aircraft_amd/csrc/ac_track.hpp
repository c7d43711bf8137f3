// ac_track.hpp — track evaluation and the progress terms of the reference's moving-horizon track tracker
// (SURVEY.md §8 f3): control/initialisation.py:782-851 (piecewise cubic Hermite track over s in [0,1]) and
// control/moving_horizon.py:144-183 / 203-239 / 44-105 (progress recursion, its initial guess, the MHTT loss).
//
// The host computes the Hermite slopes in float64 and hands over one cubic per segment and axis,
//   pos_a(t) = c[a][0] + c[a][1] t + c[a][2] t^2 + c[a][3] t^3,   t = s (n-1) - seg,   d pos/ds = pos'(t) (n-1),
// so a lookup is one index computation and 12 loads.  The reference sums its segments over CLOSED intervals
// (initialisation.py:818-819: s >= s0 and s <= s1), so a progress value exactly on an interior knot is counted by both
// neighbours — position and tangent come out as the SUM of the two one-sided values (twice the knot).  Reproduced here for
// the only progress values that can do that in fp32: those equal to a knot whose float64 value is fp32-representable
// (flags from ac_set_track).  Outside [0,1] the position is the end knot and the tangent is zero (:821-823).
#pragma once
#include "ac_math.hpp"

namespace ac {

struct TrackDev {
    const float* __restrict__ coef;  // [nseg][3][4]
    const float* __restrict__ knot_exact;  // [nseg + 1]: 1 where the knot's float64 value is an fp32 number
    int nseg;
    float inv_length;   // 1 / track.length()
    float end_pos[3];   // track.eval(1.0), the terminal-alignment target (moving_horizon.py:91)
};

struct MhttWeights {  // moving_horizon.py:47-55
    float w_tracking, w_progress, w_progress_rate, w_backward, w_terminal_align, w_low_velocity, w_control;
};

AC_DI void track_eval(const TrackDev& T, float s, float pos[3], float tan[3]) {
    const bool below = s < 0.f, above = s > 1.f;
    const float sc = fminf(fmaxf(s, 0.f), 1.f) * (float)T.nseg;
    int seg = (int)sc;
    seg = seg > T.nseg - 1 ? T.nseg - 1 : seg;
    const float t = sc - (float)seg;
    const float* c = T.coef + (long)seg * 12;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float c0 = c[a * 4], c1 = c[a * 4 + 1], c2 = c[a * 4 + 2], c3 = c[a * 4 + 3];
        pos[a] = fmaf(fmaf(fmaf(c3, t, c2), t, c1), t, c0);
        const float d = fmaf(fmaf(3.f * c3, t, 2.f * c2), t, c1) * (float)T.nseg;
        tan[a] = (below || above) ? 0.f : d;
    }
    // exactly on an interior knot: the closed interval of the segment to the left holds s too (its t = 1)
    if (t == 0.f && seg > 0 && !below && T.knot_exact[seg] != 0.f && s == (float)seg / (float)T.nseg) {
        const float* cl = c - 12;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float c1 = cl[a * 4 + 1], c2 = cl[a * 4 + 2], c3 = cl[a * 4 + 3];
            pos[a] += ((c3 + c2) + c1) + cl[a * 4];  // the same Horner evaluation at t = 1
            tan[a] += ((3.f * c3 + 2.f * c2) + c1) * (float)T.nseg;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_track_eval(const TrackDev T, const float* __restrict__ s, long n,
                                                       float* __restrict__ pos, float* __restrict__ tan) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float p[3], t[3];
    track_eval(T, s[i], p, t);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pos[(long)a * n + i] = p[a];
        tan[(long)a * n + i] = t[a];
    }
}

// One node's track terms (moving_horizon.py:147-166).  safe_norm = the `norm > 1e-3 ? norm : 1` guard of the
// constraint rows; the initial guess (moving_horizon.py:226-227) divides by the plain norm.
struct TrackTerms {
    float s_dot, delta_s, err2;
    float that[3], ref[3];
};

AC_DI TrackTerms track_terms(const TrackDev& T, float s, const float p[3], const float v[3], bool safe_norm) {
    TrackTerms o;
    float tan[3];
    track_eval(T, s, o.ref, tan);
    const float nrm = sqrtf(tan[0] * tan[0] + tan[1] * tan[1] + tan[2] * tan[2]);
    const float div = safe_norm ? (nrm > 1e-3f ? nrm : 1.f) : nrm;
#pragma unroll
    for (int a = 0; a < 3; ++a) o.that[a] = tan[a] / div;
    const float e0 = p[0] - o.ref[0], e1 = p[1] - o.ref[1], e2 = p[2] - o.ref[2];
    o.s_dot = (v[0] * o.that[0] + v[1] * o.that[1] + v[2] * o.that[2]) * T.inv_length;
    o.delta_s = (e0 * o.that[0] + e1 * o.that[1] + e2 * o.that[2]) * T.inv_length;
    o.err2 = e0 * e0 + e1 * e1 + e2 * e2;
    return o;
}

// Progress along the track for every node of every instance; one lane per instance, sequential in k.
//   mode 0  initial guess      s_{k+1} = clip(s_k + s_dot_k dt, 0, 1)                       moving_horizon.py:216-233
//   mode 1  tight constraint   s_{k+1} = clip(s_k + s_dot_k dt + 0.05 delta_s_k, 0, 1)      moving_horizon.py:161-168
// (the NLP has s_{k+1} <= prediction and rewards progress, so the bound is active at its optimum).
// Optional outputs for the batched solver: the diagonal-quadratic model of the MHTT loss around this progress
// sequence, node_q / node_xref / node_glin [H+1][13][B] (ac_ilqr.hpp NodeCost; see DESIGN.md "MHTT on iLQR").
__global__ __launch_bounds__(kBlock) void k_track_progress(const TrackDev T, const MhttWeights W,
                                                           const float* __restrict__ X, const float* __restrict__ s0,
                                                           float dt, long B, long H, int mode, float* __restrict__ S,
                                                           float* __restrict__ sdot, float* __restrict__ err2,
                                                           float* __restrict__ node_q, float* __restrict__ node_xref,
                                                           float* __restrict__ node_glin) {
    const long b = (long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    float s = s0[b];
    S[b] = s;
    const bool model = node_q != nullptr;
    // the recursion in s is the only dependence between nodes: the next node's position / velocity are loaded while this
    // node's track terms are computed
    float pn[3] = {X[b], X[B + b], X[2 * B + b]}, vn[3] = {X[3 * B + b], X[4 * B + b], X[5 * B + b]};
    for (long k = 0; k < H; ++k) {
        const float p[3] = {pn[0], pn[1], pn[2]};
        const float v[3] = {vn[0], vn[1], vn[2]};
        {
            const float* xn = X + (k + 1) * 13 * B + b;  // node k+1 <= H always exists
            pn[0] = xn[0]; pn[1] = xn[B]; pn[2] = xn[2 * B];
            vn[0] = xn[3 * B]; vn[1] = xn[4 * B]; vn[2] = xn[5 * B];
        }
        const TrackTerms t = track_terms(T, s, p, v, mode != 0);
        if (sdot) sdot[k * B + b] = t.s_dot;
        if (err2) err2[k * B + b] = t.err2;
        const float pred = s + t.s_dot * dt + (mode != 0 ? 0.05f * t.delta_s : 0.f);
        const float sn = fminf(fmaxf(pred, 0.f), 1.f);
        if (model) {
            // nodes after k whose progress still responds to x_k: all of them until the clip at 1 binds
            const float tail = (pred < 1.f) ? (float)(H - k) : 0.f;
            const float back = t.s_dot < 0.f ? 2.f * W.w_backward * t.s_dot : 0.f;
            const float speed = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float slow = (k > 0 && speed < 0.1f) ? -2.f * W.w_low_velocity * (0.1f - speed) / fmaxf(speed, 1e-6f) : 0.f;
            float* q = node_q + k * 13 * B + b;
            float* xr = node_xref + k * 13 * B + b;
            float* gl = node_glin + k * 13 * B + b;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                q[a * B] = 2.f * W.w_tracking;
                xr[a * B] = t.ref[a];
                gl[a * B] = -W.w_progress * tail * 0.05f * t.that[a] * T.inv_length * (mode != 0 ? 1.f : 0.f);
                q[(3 + a) * B] = 0.f;
                xr[(3 + a) * B] = 0.f;
                gl[(3 + a) * B] = (-W.w_progress_rate - W.w_progress * tail * dt + back) * t.that[a] * T.inv_length + slow * v[a];
            }
#pragma unroll
            for (int r = 6; r < 13; ++r) { q[r * B] = 0.f; xr[r * B] = 0.f; gl[r * B] = 0.f; }
        }
        s = sn;
        S[(k + 1) * B + b] = s;
    }
    if (model) {
        // terminal alignment w |p_H - track(1)|: gradient w d/|d|, curvature bounded by (w/|d|) I
        const float* xk = X + H * 13 * B + b;
        float d[3] = {xk[0] - T.end_pos[0], xk[B] - T.end_pos[1], xk[2 * B] - T.end_pos[2]};
        const float dist = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-3f);
        const float v[3] = {xk[3 * B], xk[4 * B], xk[5 * B]};
        const float speed = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float slow = speed < 0.1f ? -2.f * W.w_low_velocity * (0.1f - speed) / fmaxf(speed, 1e-6f) : 0.f;
        float* q = node_q + H * 13 * B + b;
        float* xr = node_xref + H * 13 * B + b;
        float* gl = node_glin + H * 13 * B + b;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            q[a * B] = W.w_terminal_align / dist;
            xr[a * B] = T.end_pos[a];
            gl[a * B] = 0.f;
            q[(3 + a) * B] = 0.f; xr[(3 + a) * B] = 0.f; gl[(3 + a) * B] = slow * v[a];
        }
#pragma unroll
        for (int r = 6; r < 13; ++r) { q[r * B] = 0.f; xr[r * B] = 0.f; gl[r * B] = 0.f; }
    }
}

// The MHTT objective (moving_horizon.py:44-105) of every instance for a given progress sequence S [H+1][B].
// Terms attached to node i >= 1 use the tracking error / progress rate computed at node i-1 (:174-175), the
// node's own speed (:80-81) and control (:84); U holds u_0..u_{H-1} (node H's control is a free variable that
// only the effort term touches, so it is zero at any optimum and contributes nothing).
__global__ __launch_bounds__(kBlock) void k_mhtt_loss(const TrackDev T, const MhttWeights W, const float* __restrict__ X,
                                                      const float* __restrict__ U, const float* __restrict__ S,
                                                      long B, long H, float* __restrict__ J) {
    const long b = (long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    float tracking = 0.f, progress = 0.f, rate = 0.f, backward = 0.f, slow = 0.f, effort = 0.f;
    for (long k = 0; k < H; ++k) {
        const float* xk = X + k * 13 * B + b;
        const float p[3] = {xk[0], xk[B], xk[2 * B]};
        const float v[3] = {xk[3 * B], xk[4 * B], xk[5 * B]};
        const TrackTerms t = track_terms(T, S[k * B + b], p, v, true);
        tracking += t.err2;
        rate += t.s_dot;
        const float neg = fmaxf(0.f, -t.s_dot);
        backward = fmaf(neg, neg, backward);
        progress += S[(k + 1) * B + b];
        const float* xn = X + (k + 1) * 13 * B + b;
        const float vn = sqrtf(xn[3 * B] * xn[3 * B] + xn[4 * B] * xn[4 * B] + xn[5 * B] * xn[5 * B]);
        const float lv = fmaxf(0.1f - vn, 0.f);
        slow = fmaf(lv, lv, slow);
        if (k >= 1) {
#pragma unroll
            for (int r = 0; r < 7; ++r) { const float u = U[(k * 7 + r) * B + b]; effort = fmaf(u, u, effort); }
        }
    }
    const float* xN = X + H * 13 * B + b;
    const float d0 = xN[0] - T.end_pos[0], d1 = xN[B] - T.end_pos[1], d2 = xN[2 * B] - T.end_pos[2];
    const float term = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    J[b] = W.w_tracking * tracking - W.w_progress * progress - W.w_progress_rate * rate + W.w_backward * backward +
           W.w_low_velocity * slow + W.w_terminal_align * term + W.w_control * effort;
}

}  // namespace ac

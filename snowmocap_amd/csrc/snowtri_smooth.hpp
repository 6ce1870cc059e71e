// snowtri_smooth.hpp -- row N1: Human_Triangulation_Smooth / SecondOrderDynamic over a whole track.
//
// Reference (triangulation.py:4-22,164-186): per (person, joint, axis) lane, frame 0 passes through and seeds
// xp = y = x0, yd = 0; every later frame
//     xd = (x - xp)/T;  xp = x;  y += T*yd;  yd += T*(x + k3*xd - y - k1*yd)/k2
// with k1 = z/(pi f), k2 = 1/(2 pi f)^2, k3 = r z/(2 pi f).
//
// The recurrence couples frames, so it cannot ride the frame sharding of the triangulation path; but it is
// LINEAR in the state s = (y, yd):  s_t = A s_{t-1} + (0, c_t),  c_t = (T/k2)(x_t + k3 xd_t),
//     A = [[1, T], [-T/k2, 1 - T^2/k2 - T k1/k2]].
// It is therefore evaluated as a chunked scan over frames (chunk length L):
//   k_smooth_local   lane x chunk: zero-state response z_t of the chunk, written to y; chunk-end state E[c]
//   k_smooth_carry   lane: S_{c+1} = A^L S_c + E[c]  (S_0 = (x0, 0));  A^L precomputed on the host
//   k_smooth_fix     lane x chunk: y_t += (A^{t-t0+1} S_c).y
// Lanes are the fast axis of x[T][n] / y[T][n]: every load and store is coalesced.
#pragma once
#include "snowtri_math.hpp"

namespace snowtri {

struct SmoothCoef {
    double a00, a01, a10, a11;  // A
    double cx, cxd;             // c_t = cx * x_t + cxd * (x_t - x_{t-1})      (cxd = (T/k2) k3 / T)
    double p00, p01, p10, p11;  // A^L
};

constexpr int kSmoothBlock = 256;
constexpr int kSmoothUnroll = 8;   // frames whose loads are in flight per lane

// Where a lane's coefficients come from: one set for the whole track (N1), or a table indexed by the lane's
// control point (N2: per-bone f, z, r -- blender.py:171; lanes = [person][ncoef points][comps]).
struct UniformCoef {
    SmoothCoef k;
    __device__ __forceinline__ const SmoothCoef &at(int64_t) const { return k; }
};
struct TableCoef {
    const SmoothCoef *tab;
    int comps, ncoef;
    __device__ __forceinline__ SmoothCoef at(int64_t lane) const { return tab[(lane / comps) % ncoef]; }
};

// What a lane's input is: the track itself (N1), or the track with invalid points replaced by the lane's
// previous input (N2, blender.py:157-160: `flt.update(dt, x if score else flt.xp)`).
struct NoHold {
    __device__ __forceinline__ double entering(const double *x, int64_t t0, int64_t, int64_t n, int64_t lane) const {
        return x[(t0 > 0 ? t0 - 1 : 0) * n + lane];
    }
    __device__ __forceinline__ int64_t groups(int64_t) const { return 0; }
    __device__ __forceinline__ int64_t group(int64_t) const { return 0; }
    __device__ __forceinline__ bool ok(int64_t) const { return true; }
};
struct HoldInput {
    const uint8_t *valid;  // [T][n / comps]
    const double *start;   // [nchunks][n]: the held input entering each chunk (k_hold_carry)
    int comps;
    __device__ __forceinline__ double entering(const double *, int64_t, int64_t c, int64_t n, int64_t lane) const {
        return start[c * n + lane];
    }
    __device__ __forceinline__ int64_t groups(int64_t n) const { return n / comps; }
    __device__ __forceinline__ int64_t group(int64_t lane) const { return lane / comps; }
    __device__ __forceinline__ bool ok(int64_t vidx) const { return valid[vidx] != 0; }
};

// frames tb..T-1 are the filtered ones (tb = 1: frame 0 is the seed and passes through; tb = 0: a later
// shard of a frame-sharded track, whose first frame is filtered with xd = 0 -- the caller corrects for the
// true previous input afterwards); chunk c covers frames [tb + c L, min(T, tb + (c+1) L)).
// S (optional): state entering each chunk, [nchunks][n][2]; absent = zero state.
// y (optional): the response is written; E (optional): the chunk-end state is written.
template <typename KS, typename HS>
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_local(int64_t T, int64_t n, int L, int tb, KS ks, HS hs,
                                                               const double *__restrict__ x,
                                                               const double *__restrict__ S,
                                                               double *__restrict__ y, double *__restrict__ E) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (lane >= n) return;
    const SmoothCoef k = ks.at(lane);
    const int64_t t0 = tb + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    if (y && c == 0 && tb == 1) y[lane] = x[lane];  // frame 0 passes through (:180-181)
    double sy = S ? S[(c * n + lane) * 2] : 0.0, syd = S ? S[(c * n + lane) * 2 + 1] : 0.0;
    double xp = hs.entering(x, t0, c, n, lane);
    const int64_t nv = hs.groups(n), g = hs.group(lane);
    auto step = [&](int64_t t, double xraw, bool okt) {
        const double xt = okt ? xraw : xp;   // an invalid point repeats the previous input (N2 hold)
        const double ct = fma(k.cxd, xt - xp, k.cx * xt);
        xp = xt;
        const double ny = fma(k.a01, syd, k.a00 * sy);
        const double nyd = fma(k.a11, syd, fma(k.a10, sy, ct));
        sy = ny;
        syd = nyd;
        if (y) y[t * n + lane] = sy;
    };
    // the loads do not depend on the recurrence: issue kSmoothUnroll frames of them ahead of the dependent
    // FMA chain, otherwise every frame pays a full memory latency (the loop is not unrolled by itself)
    int64_t t = t0;
    for (; t + kSmoothUnroll <= t1; t += kSmoothUnroll) {
        double xv[kSmoothUnroll];
        bool ov[kSmoothUnroll];
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) {
            xv[u] = x[(t + u) * n + lane];
            ov[u] = hs.ok((t + u) * nv + g);
        }
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) step(t + u, xv[u], ov[u]);
    }
    for (; t < t1; t++) step(t, x[t * n + lane], hs.ok(t * nv + g));
    if (E) {
        E[(c * n + lane) * 2] = sy;
        E[(c * n + lane) * 2 + 1] = syd;
    }
}

// start: state entering the first chunk -- [2n] array, or nullptr = zero.  E: chunk zero-state end states,
// or nullptr = none (pure homogeneous propagation).  end_out (optional, [2n]): state after the last FULL-LENGTH
// step count, i.e. exact only when every chunk is full; callers that need the shard's end state use the
// per-lane sequential tail below instead.
template <typename KS>
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_carry(int64_t n, int64_t nchunks, KS ks,
                                                               const double *__restrict__ start,
                                                               const double *__restrict__ E,
                                                               double *__restrict__ S) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    const SmoothCoef k = ks.at(lane);
    double sy = start ? start[2 * lane] : 0.0, syd = start ? start[2 * lane + 1] : 0.0;
    auto step = [&](int64_t c, double ey, double eyd) {
        S[(c * n + lane) * 2] = sy;
        S[(c * n + lane) * 2 + 1] = syd;
        const double ny = fma(k.p01, syd, fma(k.p00, sy, ey));
        const double nyd = fma(k.p11, syd, fma(k.p10, sy, eyd));
        sy = ny;
        syd = nyd;
    };
    // few lanes, many chunks: keep kSmoothUnroll chunk-end states in flight ahead of the dependent chain
    int64_t c = 0;
    for (; c + kSmoothUnroll <= nchunks; c += kSmoothUnroll) {
        double ey[kSmoothUnroll], eyd[kSmoothUnroll];
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) {
            ey[u] = E ? E[((c + u) * n + lane) * 2] : 0.0;
            eyd[u] = E ? E[((c + u) * n + lane) * 2 + 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) step(c + u, ey[u], eyd[u]);
    }
    for (; c < nchunks; c++) step(c, E ? E[(c * n + lane) * 2] : 0.0, E ? E[(c * n + lane) * 2 + 1] : 0.0);
}

// seed state of a track: (x0, 0) per lane (:11-13)
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_seed(int64_t n, const double *__restrict__ x0,
                                                              double *__restrict__ st) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    st[2 * lane] = x0[lane];
    st[2 * lane + 1] = 0.0;
}

// y_t += (A^{t-t0+1} S_c).y ; the last chunk (optionally) also reports the propagated state (end_out, [2n])
template <typename KS>
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_fix(int64_t T, int64_t n, int L, int tb, KS ks,
                                                             const double *__restrict__ S, double *__restrict__ y,
                                                             double *__restrict__ end_out) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (lane >= n) return;
    const SmoothCoef k = ks.at(lane);
    const int64_t t0 = tb + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    double vy = S[(c * n + lane) * 2], vyd = S[(c * n + lane) * 2 + 1];
    int64_t t = t0;
    for (; t + kSmoothUnroll <= t1; t += kSmoothUnroll) {
        double yv[kSmoothUnroll];
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) yv[u] = y[(t + u) * n + lane];
#pragma unroll
        for (int u = 0; u < kSmoothUnroll; u++) {
            const double ny = fma(k.a01, vyd, k.a00 * vy);
            const double nyd = fma(k.a11, vyd, k.a10 * vy);
            vy = ny;
            vyd = nyd;
            y[(t + u) * n + lane] = yv[u] + vy;
        }
    }
    for (; t < t1; t++) {
        const double ny = fma(k.a01, vyd, k.a00 * vy);
        const double nyd = fma(k.a11, vyd, k.a10 * vy);
        vy = ny;
        vyd = nyd;
        y[t * n + lane] += vy;
    }
    if (end_out && t1 == T) {
        end_out[2 * lane] = vy;
        end_out[2 * lane + 1] = vyd;
    }
}

// zero-state end state of the whole shard = zero-start carry over the chunks, advanced through the last chunk:
// the last chunk's own zero-state end E[last] plus the homogeneous propagation of its start state.
template <typename KS>
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_shard_end(int64_t T, int64_t n, int L, int tb,
                                                                   int64_t nchunks, KS ks,
                                                                   const double *__restrict__ S,
                                                                   const double *__restrict__ E,
                                                                   double *__restrict__ end_out) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    const SmoothCoef k = ks.at(lane);
    const int64_t c = nchunks - 1;
    const int64_t t0 = tb + c * L;
    double vy = S[(c * n + lane) * 2], vyd = S[(c * n + lane) * 2 + 1];
    for (int64_t t = t0; t < T; t++) {
        const double ny = fma(k.a01, vyd, k.a00 * vy);
        const double nyd = fma(k.a11, vyd, k.a10 * vy);
        vy = ny;
        vyd = nyd;
    }
    end_out[2 * lane] = vy + E[(c * n + lane) * 2];
    end_out[2 * lane + 1] = vyd + E[(c * n + lane) * 2 + 1];
}

// Sharded track (snowtri_smooth_shard_local / _fix): the true state entering shard `rank` from the gathered carries of the
// shards before it -- one lane per thread, the shards walked in frame order.  gathered[q] = 4n + 1 doubles of shard q:
// [end state (y, yd) per lane | first input row | last input row | length T_q].  Per preceding shard
//     start_q = S_q + A^-1 (0, cxd (x_first_q - x_last_{q-1}))     (its first frame was filtered with xd = 0)
//     S_{q+1} = A^m start_q + E_q,   m = T_q  (T_q - 1 for the shard that starts the track: its frame 0 passes through)
// A^m by squaring (<= 63 steps, the same for every lane).  Empty shards (T_q = 0) are skipped; no shard before `rank`
// holds a frame -> rank starts the track and start = (x_first, 0).  (Host twin: snowmocap_amd/sharded.py::combine_carries.)
template <typename KS>
__global__ __launch_bounds__(kSmoothBlock) void k_smooth_combine(int world, int rank, int64_t n, const double *__restrict__ gathered,
                                                                 KS ks, double *__restrict__ start_state) {
    const int64_t i = (int64_t)blockIdx.x * kSmoothBlock + threadIdx.x;
    if (i >= n) return;
    const SmoothCoef k = ks.at(i);
    const int64_t stride = 4 * n + 1;
    const double det = k.a00 * k.a11 - k.a01 * k.a10;
    const double iv0 = -k.a01 / det, iv1 = k.a00 / det;   // second column of A^-1
    bool have = false;
    double s0 = 0.0, s1 = 0.0, x_last_prev = 0.0, o0 = 0.0, o1 = 0.0;
    for (int q = 0; q <= rank && q < world; q++) {
        const double *g = gathered + (int64_t)q * stride;
        const int64_t T = (int64_t)g[4 * n];
        if (T <= 0) continue;
        const double x_first = g[2 * n + i], x_last = g[3 * n + i];
        double b0, b1;
        int64_t m;
        if (!have) {
            s0 = x_first;
            s1 = 0.0;
            have = true;
            b0 = s0;
            b1 = s1;
            m = T - 1;
        } else {
            const double delta = k.cxd * (x_first - x_last_prev);
            b0 = s0 + delta * iv0;
            b1 = s1 + delta * iv1;
            m = T;
        }
        if (q == rank) {
            o0 = b0;
            o1 = b1;
            break;
        }
        if (m > 0) {
            double p00 = 1.0, p01 = 0.0, p10 = 0.0, p11 = 1.0;              // A^m
            double q00 = k.a00, q01 = k.a01, q10 = k.a10, q11 = k.a11;      // A^(2^j)
            for (int64_t e = m; e > 0; e >>= 1) {
                if (e & 1) {
                    const double t00 = p00 * q00 + p01 * q10, t01 = p00 * q01 + p01 * q11;
                    const double t10 = p10 * q00 + p11 * q10, t11 = p10 * q01 + p11 * q11;
                    p00 = t00; p01 = t01; p10 = t10; p11 = t11;
                }
                const double u00 = q00 * q00 + q01 * q10, u01 = q00 * q01 + q01 * q11;
                const double u10 = q10 * q00 + q11 * q10, u11 = q10 * q01 + q11 * q11;
                q00 = u00; q01 = u01; q10 = u10; q11 = u11;
            }
            s0 = p00 * b0 + p01 * b1 + g[2 * i];
            s1 = p10 * b0 + p11 * b1 + g[2 * i + 1];
        }
        x_last_prev = x_last;
    }
    start_state[2 * i] = o0;
    start_state[2 * i + 1] = o1;
}

}  // namespace snowtri

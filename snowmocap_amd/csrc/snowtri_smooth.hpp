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

}  // namespace snowtri

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
    double p00, p01, p10, p11;  // A^L   (L = 256 frames: a chunk of the three-pass scan = a workgroup of the one-pass scan)
    double r00, r01, r10, r11;  // A^32  (the frames one wave of the one-pass scan holds in registers)
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
    static constexpr bool kHold = false;
    __device__ __forceinline__ double entering(const double *x, int64_t t0, int64_t, int64_t n, int64_t lane) const {
        return x[(t0 > 0 ? t0 - 1 : 0) * n + lane];
    }
    __device__ __forceinline__ int64_t groups(int64_t) const { return 0; }
    __device__ __forceinline__ int64_t group(int64_t) const { return 0; }
    __device__ __forceinline__ bool ok(int64_t) const { return true; }
};
struct HoldInput {
    static constexpr bool kHold = true;
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

// ---- the whole track in ONE pass over HBM (round 6): a chained scan with decoupled look-back ------------------------------------
// The three-pass form above reads x twice and writes y once: 24 bytes moved per 16 algorithmic, and it is HBM-bound (0.27 of 8 TB/s,
// round-5 review).  Here a workgroup of kScanWaves waves owns kScanSuper = 256 consecutive frames of 64 lanes, each wave 32 of them
// IN REGISTERS (32 loads per lane in flight), and the track is read once and written once:
//   A  every wave: zero-state response of its 32 frames -> end state e_w (LDS);
//   B  in the workgroup: z_w = sum_{j<w} R^(w-1-j) e_j (R = A^32): the zero-state state entering wave w; the last wave has the
//      workgroup's aggregate G = R z + e and PUBLISHES it (flag 1);
//   C  the last wave looks back over the workgroups before it in its lane column: S_in = G_(b-1) + P G_(b-2) + P^2 G_(b-3) + ...
//      (P = A^256) until it meets one that has published its INCLUSIVE state (flag 2), then publishes its own P S_in + G.  The
//      workgroups of a column finish their local phase at about the same time, the first publishes at once, and the distance a
//      workgroup walks grows like the square root of its position among those in flight (~10 steps of 57): microseconds;
//   D  every wave: its true entering state R^w S_in + z_w, then the exact recurrence over its registers, y stored.
// (The look-back multiplies by growing powers of P: for a STABLE filter -- spectral radius of A below 1, the reference's profiles:
// 1.5 ... 3 Hz at 30 fps -- they decay; a filter beyond ~4.7 Hz at z = 0.75 diverges in the reference too, and here its powers
// overflow after a few workgroups.)
// Workgroups take their (time, column) position from a ticket: a workgroup only ever waits for SMALLER tickets, which have started
// and cannot be descheduled, so the spin cannot deadlock whatever order the dispatcher picks.  SKIP4: the track is [T][m][4] joint
// records (x, y, z, score) and every fourth lane is copied, not filtered (triangulation.py:169-184 filters the points only).
constexpr int kScanL = 32, kScanWaves = 8, kScanSuper = kScanL * kScanWaves, kScanThreads = 64 * kScanWaves;
// A wave publishes (y, yd) per lane, then the flag.  Everything that crosses workgroups -- they may sit on different XCDs, whose L2s
// are not coherent with one another -- goes through agent-scope atomics (sc1: written through to / read from the memory side),
// and the flag is stored once the state's stores have been acknowledged (s_waitcnt vmcnt(0)).  A release FENCE instead
// (__threadfence: write back the L2, 4 MB of it dirty with the track's own stores) and an acquire load per poll (invalidate)
// cost ~2 us per link of a chain of hundreds: the first version of this kernel ran 2.7 x SLOWER than the three passes it replaces.
__device__ __forceinline__ void scan_publish(double *dst, double vy, double vyd, unsigned int *flag, unsigned int value, int l) {
    __hip_atomic_store(dst, vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 64, vyd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (l == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static_assert(kScanSuper == 256, "a workgroup of the one-pass scan covers one chunk of the hold kernels (kSmoothChunk)");
// The same kernel serves a frame SHARD (round 6: the sharded protocol used to run the three passes from a zero state and then a
// fourth and fifth over y to add the entering state's response -- 40 bytes per lane-frame): tb = 0 filters every frame (the first
// one with xd = 0: the caller's combine corrects for the true previous input), `start` ([2n], optional) is the state entering the
// first filtered frame (absent: (seed_row, 0), or zero without a seed row), `end_out` ([2n], optional) receives the state behind
// the LAST frame, and y == nullptr stores nothing (a zero-length buffer descriptor drops every store): the shard's zero-state end
// state costs ONE read of x ("reduce"), and after the exchange the shard is filtered in one more pass ("scan"): 24 bytes.
template <typename KS, typename HS, bool SKIP4>
__global__ __launch_bounds__(kScanThreads) void k_smooth_scan(int64_t T, int64_t n, KS ks, HS hs, const double *__restrict__ x,
                                                              const double *__restrict__ seed_row, double *__restrict__ y,
                                                              unsigned int *ticket, unsigned int *flags, double *agg, double *incl,
                                                              int tb = 1, const double *__restrict__ start = nullptr,
                                                              double *__restrict__ end_out = nullptr) {
    __shared__ unsigned int s_bid;
    __shared__ double s_e[kScanWaves][2][64];
    __shared__ double s_in[2][64];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const unsigned int bid = s_bid;
    const unsigned int ncols = (unsigned int)((n + 63) >> 6);
    const unsigned int sc = bid / ncols, col = bid - sc * ncols;
    const int64_t lane = (int64_t)col * 64 + l;
    const bool live = lane < n;
    const int64_t lc = live ? lane : n - 1;   // (idle lanes of the last column read the last lane's data and store nothing)
    const SmoothCoef k = ks.at(lc);
    const int64_t t0 = tb + (int64_t)sc * kScanSuper + (int64_t)w * kScanL;   // frames tb .. T-1 are the filtered ones
    const int nt = (int)(T - t0 < 0 ? 0 : (T - t0 > kScanL ? kScanL : T - t0));
    const int64_t nv = hs.groups(n), g = hs.group(lc);
    // ---- A: this wave's frames into registers, zero-state end state
    double xv[kScanL];
    unsigned int okm = 0xffffffffu;   // bit u: frame t0 + u carries a valid point (HoldInput; one register instead of 32 flags)
    double x_enter = nt > 0 ? hs.entering(x, t0, (int64_t)sc, n, lc) : 0.0;   // the input in front of frame t0 (HoldInput: of the WORKGROUP's first frame)
    // Range-checked buffer accesses over the rows [t0, T) of x and y (byte offset = row * 8 n + 8 lane, all of it in the
    // per-lane offset): a frame behind the track's end reads zeros and its store is dropped, so the 32 loads and the 32
    // stores of a wave are branch-free (with `if (u < nt)` around each the compiler built a branch per frame and kept
    // 170 VGPRs: one workgroup per CU).  A partial wave's end state is then garbage -- and unused: only later frames
    // would read it.  Idle lanes of the last column get an offset no descriptor covers.
    typedef unsigned scan_u2 __attribute__((ext_vector_type(2)));
    const int64_t rows_left = T - t0 > 0 ? T - t0 : 0;
    const unsigned long long span = (unsigned long long)rows_left * (unsigned long long)n * 8ull;
    // (a wave's offsets stay below 32 x 8 n < 2^30: the host keeps n at 2^22 lanes at most; an idle lane starts at 2^31 and
    // never wraps into the range -- it did, from 0xfffffff0, and wrote its garbage over lane 0's results: found by the n < 64 tests)
    const unsigned int span32 = span > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned int)span;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x + t0 * n), 0, (int)span32, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y ? y + t0 * n : const_cast<double *>(x), 0, y ? (int)span32 : 0, 0x00020000);
    const unsigned int row_bytes = (unsigned int)n * 8u;
    const unsigned int off0 = live ? (unsigned int)lane * 8u : 0x80000000u;
    unsigned int off = off0;   // (a running offset: one register, not 32 kept for the stores)
#pragma unroll
    for (int u = 0; u < kScanL; u++, off += row_bytes) {
        const scan_u2 wv = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)off, 0, 0);
        xv[u] = __hiloint2double((int)wv.y, (int)wv.x);
        if constexpr (HS::kHold) {
            const int64_t tt = t0 + u < T ? t0 + u : T - 1;
            if (!hs.ok(tt * nv + g)) okm &= ~(1u << u);
        }
    }
    if constexpr (HS::kHold) {
        // an invalid point repeats the previous input (blender.py:157-160): the held input entering wave w is the last valid
        // input of the nearest wave before it that has one, else the one entering the workgroup (start[sc], k_hold_carry)
        double hv = 0.0;
        bool hf = false;
#pragma unroll
        for (int u = 0; u < kScanL; u++)
            if (u < nt && ((okm >> u) & 1u)) {   // (nt: a frame behind the track's end is not an input)
                hv = xv[u];
                hf = true;
            }
        s_e[w][0][l] = hv;
        s_e[w][1][l] = hf ? 1.0 : 0.0;
        __syncthreads();
        for (int j = 0; j < w; j++)
            if (s_e[j][1][l] != 0.0) x_enter = s_e[j][0][l];
        __syncthreads();   // (s_e is rewritten below)
    }
    double sy = 0.0, syd = 0.0, xp = x_enter;
    auto step = [&](double xraw, bool okt) {
        const double xt = okt ? xraw : xp;
        const double ct = fma(k.cxd, xt - xp, k.cx * xt);
        xp = xt;
        const double ny = fma(k.a01, syd, k.a00 * sy);
        const double nyd = fma(k.a11, syd, fma(k.a10, sy, ct));
        sy = ny;
        syd = nyd;
    };
#pragma unroll
    for (int u = 0; u < kScanL; u++) step(xv[u], !HS::kHold || ((okm >> u) & 1u));
    // (the inputs are laundered: otherwise the compiler keeps the 32 forcing terms c_t of this pass for the second one BESIDE
    // the inputs the copied lanes need -- 128 registers of frame data, one workgroup per CU; recomputing c_t costs three
    // instructions per frame of a kernel that waits for HBM)
#pragma unroll
    for (int u = 0; u < kScanL; u++) asm volatile("" : "+v"(xv[u]));
    s_e[w][0][l] = sy;
    s_e[w][1][l] = syd;
    __syncthreads();
    // ---- B: zero-state state entering this wave; the last wave: the workgroup's aggregate
    double zy = 0.0, zyd = 0.0;
    for (int j = 0; j < w; j++) {
        const double ey = s_e[j][0][l], eyd = s_e[j][1][l];
        const double ny = fma(k.r01, zyd, fma(k.r00, zy, ey));
        const double nyd = fma(k.r11, zyd, fma(k.r10, zy, eyd));
        zy = ny;
        zyd = nyd;
    }
    if (w == kScanWaves - 1) {
        const double gy = fma(k.r01, zyd, fma(k.r00, zy, sy)), gyd = fma(k.r11, zyd, fma(k.r10, zy, syd));   // G = R z + e
        const size_t slot = ((size_t)sc * ncols + col) * 128 + l;
        double iny, inyd;   // S_in: the true state entering this workgroup
        if (sc == 0) {
            iny = start ? start[2 * lc] : (seed_row ? seed_row[lc] : 0.0);   // (x_0, 0): frame 0 seeds the filters (:11-13); a shard: its entering state
            inyd = start ? start[2 * lc + 1] : 0.0;
        } else {
            scan_publish(agg + slot, gy, gyd, flags + (size_t)sc * ncols + col, 1u, l);
            // ---- C: look back
            double ay = 0.0, ayd = 0.0, m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;   // acc, M = P^(steps so far)
            for (int64_t b = (int64_t)sc - 1;; b--) {
                unsigned int f;
                while ((f = __hip_atomic_load(&flags[(size_t)b * ncols + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u)
                    __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");   // (the state is read behind the flag, by loads of the same coherence level: no cache to invalidate)
                const double *src = (f == 2u ? incl : agg) + ((size_t)b * ncols + col) * 128 + l;
                const double vy = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double vyd = __hip_atomic_load(src + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ay = fma(m01, vyd, fma(m00, vy, ay));
                ayd = fma(m11, vyd, fma(m10, vy, ayd));
                if (f == 2u) break;
                const double t00 = fma(m00, k.p00, m01 * k.p10), t01 = fma(m00, k.p01, m01 * k.p11);   // M <- M P
                const double t10 = fma(m10, k.p00, m11 * k.p10), t11 = fma(m10, k.p01, m11 * k.p11);
                m00 = t00; m01 = t01; m10 = t10; m11 = t11;
            }
            iny = ay;
            inyd = ayd;
        }
        // the state BEHIND this workgroup: P S_in + G
        scan_publish(incl + slot, fma(k.p01, inyd, fma(k.p00, iny, gy)), fma(k.p11, inyd, fma(k.p10, iny, gyd)), flags + (size_t)sc * ncols + col, 2u, l);
        s_in[0][l] = iny;
        s_in[1][l] = inyd;
    }
    __syncthreads();
    // ---- D: the true state entering this wave, then the exact recurrence over the registers
    sy = s_in[0][l];
    syd = s_in[1][l];
    for (int j = 0; j < w; j++) {
        const double ny = fma(k.r01, syd, k.r00 * sy), nyd = fma(k.r11, syd, k.r10 * sy);
        sy = ny;
        syd = nyd;
    }
    sy += zy;
    syd += zyd;
    xp = x_enter;
    const bool copy = SKIP4 && (lane & 3) == 3;
    if (sc == 0 && w == 0 && live && tb == 1 && y) y[lane] = x[lane];   // frame 0 passes through (:180-181)
    const int last_u = (t0 + nt == T) ? nt - 1 : -1;   // this wave holds the last frame of the track / shard at position last_u
    off = off0;
    asm volatile("" : "+v"(off));   // (not the offsets of the loads again: they would live across the whole kernel)
#pragma unroll
    for (int u = 0; u < kScanL; u++, off += row_bytes) {
        step(xv[u], !HS::kHold || ((okm >> u) & 1u));
        const double out = copy ? xv[u] : sy;
        scan_u2 ow;
        ow.x = (unsigned int)__double2loint(out);
        ow.y = (unsigned int)__double2hiint(out);
        __builtin_amdgcn_raw_buffer_store_b64(ow, ry, (int)off, 0, 0);
        if (end_out && u == last_u && live) {   // (wave-uniform; the steps behind it run on zeros and are never read)
            end_out[2 * lane] = sy;
            end_out[2 * lane + 1] = syd;
        }
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // a frame's store leaves with its step: the results are not collected in registers first
    }
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

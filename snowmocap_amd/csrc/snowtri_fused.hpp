// snowtri_fused.hpp -- the batched hot path: A1..A4 of SURVEY.md §8a in ONE launch.
//
//   general_frame     one workgroup resolves one frame with the reference's full algorithm
//                     (any person count, any thresholds); candidates spill to a per-workgroup
//                     scratch slab in HBM (L2-resident for small rigs)
//   k_frame_general   persistent workgroups over frames, each calling general_frame
//   k_fused_single    single-detection-per-camera fast path: one lane per (frame, joint) keeps all
//                     C(C,2) pair solves in registers and fuses them on the spot -- no candidate
//                     ever leaves the register file.  It is SPECULATIVE: it assumes every candidate
//                     is kept and that all of them fall into one cluster (what the shipped config,
//                     condense_distance_tol = 10 m, always produces), verifies that per frame, and
//                     hands any frame that violates it to general_frame inside the same launch.
#pragma once
#include "snowtri_item.hpp"

namespace snowtri {

constexpr uint32_t kFlagSingular = 1u, kFlagOverflow = 2u, kFlagFast = 4u;
constexpr uint32_t kSlow = 0x100u;  // LDS-only marker: frame needs general_frame

// Per-workgroup scratch (bytes) general_frame needs for Kc candidate slots of J joints.
__host__ __device__ constexpr size_t general_scratch_bytes(int64_t Kc, int J) {
    return (((size_t)Kc * J * 32) + 255) & ~(size_t)255;
}

template <typename TIn, typename Writer>
__device__ __forceinline__ void general_frame(int64_t f, int Pmax, int J, int Kc, const Rig &rig,
                                              const TIn *__restrict__ kpts,
                                              const int32_t *__restrict__ n_persons, const Params &prm,
                                              int Pout, const Writer &wr, int32_t *__restrict__ out_count,
                                              uint32_t *__restrict__ out_flags, double *scratch, char *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *cxyz = scratch;
    double *cks = scratch + (size_t)Kc * J * 3;
    int32_t *keep_lds = reinterpret_cast<int32_t *>(lds) + Kc;  // = condense_frame's cluster_of
    const int pp = Pmax * Pmax;
    const int32_t *np_f = n_persons ? n_persons + f * rig.C : nullptr;
    if (tid == 0 && out_flags) out_flags[f] = 0u;  // this routine owns the frame's flag word
    // A1 + A3 per (slot, joint): triangulation.py:56-78
    bool sing = false;
    for (int i = tid; i < Kc * J; i += kBlock) {
        const int k = i / J, j = i - k * J;
        const int q = k / pp, r = k - q * pp, pm = r / Pmax, ps = r - pm * Pmax;
        const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
        const int nm = np_f ? np_f[mc] : Pmax, ns = np_f ? np_f[sc] : Pmax;
        if (pm >= nm || ps >= ns) continue;
        const TIn *km = kpts + ((((f * rig.C + mc) * Pmax + pm) * (int64_t)J) + j) * 3;
        const TIn *ks = kpts + ((((f * rig.C + sc) * Pmax + ps) * (int64_t)J) + j) * 3;
        const TIn um = km[0], vm = km[1], sm = km[2];
        const TIn us = ks[0], vs = ks[1], ss = ks[2];
        const Vec3 hm = ray_from_pixel(rig.M + 9 * mc, (double)um, (double)vm);
        const Vec3 hs = ray_from_pixel(rig.M + 9 * sc, (double)us, (double)vs);
        const Vec3 tm = {rig.t[3 * mc], rig.t[3 * mc + 1], rig.t[3 * mc + 2]};
        const Vec3 ts = {rig.t[3 * sc], rig.t[3 * sc + 1], rig.t[3 * sc + 2]};
        const SkewOut o = skew_ray_solve(hm, hs, tm, ts);
        sing |= o.singular;
        cxyz[3 * i] = o.W.x;
        cxyz[3 * i + 1] = o.W.y;
        cxyz[3 * i + 2] = o.W.z;
        cks[i] = pair_score(sm, ss, o.dist, prm);
    }
    __syncthreads();
    if (sing && out_flags) atomicOr(&out_flags[f], kFlagSingular);
    // A3 candidate means (triangulation.py:79-81): one wave per slot, flags parked in LDS
    for (int k = wave; k < Kc; k += kBlock / 64) {
        const int q = k / pp, r = k - q * pp, pm = r / Pmax, ps = r - pm * Pmax;
        const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
        const int nm = np_f ? np_f[mc] : Pmax, ns = np_f ? np_f[sc] : Pmax;
        const bool valid = pm < nm && ps < ns;
        double s = 0.0;
        if (valid)
            for (int j = lane; j < J; j += 64) s += cks[(size_t)k * J + j];
        s = wave_sum(s);
        const double mean = s / (double)J;
        if (lane == 0) keep_lds[k] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;
    }
    __syncthreads();
    condense_frame(f, Kc, J, cxyz, cks, nullptr, true, prm, Pout, wr, out_count, out_flags, lds);
}

// Persistent workgroups over frames [0, F); dynamic LDS = condense_lds_bytes(Kc);
// scratch = gridDim.x slabs of general_scratch_bytes(Kc, J).
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock) void k_frame_general(int64_t F, int Pmax, int J, int Kc, Rig rig,
                                                          const TIn *__restrict__ kpts,
                                                          const int32_t *__restrict__ n_persons, Params prm,
                                                          int Pout, TOut *__restrict__ out4,
                                                          TOut *__restrict__ out_ps,
                                                          int32_t *__restrict__ out_count,
                                                          uint32_t *__restrict__ out_flags, char *scratch,
                                                          size_t scratch_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *slab = reinterpret_cast<double *>(scratch + (size_t)blockIdx.x * scratch_per_block);
    const PackedWriter<TOut> wr{out4, out_ps};
    for (int64_t f = blockIdx.x; f < F; f += gridDim.x)
        general_frame<TIn>(f, Pmax, J, Kc, rig, kpts, n_persons, prm, Pout, wr, out_count, out_flags, slab, smem);
}

// ------------------------------------------------------------------------------------------------
// Fast path.  Layout assumptions: Pmax == 1 (kpts[F][C][1][J][3]).  Host-checked preconditions:
// 3 <= C, average_score_threshold <= 0, C(C,2) >= condense_person_num_tol.  Per-frame conditions
// verified in the kernel (violations -> general_frame):
//   every camera reports exactly one detection;  no pair is singular;
//   no joint score is negative (so every candidate mean is >= 0 >= average_score_threshold
//   and every candidate is kept; NaN means are kept by the reference as well);
//   every candidate's centre joint lies within condense_distance_tol of candidate 0's
//   (seed 0 then absorbs all: one cluster);  the fused mean score is not below condense_score_tol.
// Minimum waves per SIMD the fast kernel is compiled for (caps its VGPR allocation: 4 -> 128).
// Depth of the register prefetch ring of k_fused_single (2 or 3 keypoint buffers per lane).
constexpr int kFastWaves = 2;
// Shape of k_fused_single per instantiation.  The pairwise method runs here for up to four cameras only (six pairs: every
// constant and every pair's intermediate in registers); five and more take the lean kernels on cluster_item or the streaming
// route (fused_dispatch).  The DLT item of five and more cameras reads a camera's world->pixel matrix where its observation
// is added and keeps TWO keypoint buffers in flight instead of three, as do float64 keypoints (buffers of twice the size):
// with three those shapes spilled 7-260 VGPRs (round-4 review).
#ifndef SNOWTRI_DLT_RING1_FROM
#define SNOWTRI_DLT_RING1_FROM 6
#endif
#ifndef SNOWTRI_DLT_WIDE_WAVES
#define SNOWTRI_DLT_WIDE_WAVES 2
#endif
// (DLT up to four cameras, round 6: the matrices read per camera as well, three waves per SIMD and two buffers -- 139 VGPRs; with the
// 48 doubles of P hoisted it held 226 and two waves: 29.0 against 28.4 us per 10 000 frames of 4 x 1, the item is VALU-bound either way)
template <int C, int METHOD, typename TIn>
struct FusedShape {
    static constexpr int kRing = C >= SNOWTRI_DLT_RING1_FROM ? 1 : ((C >= 5 || sizeof(TIn) == 8) ? 2 : (METHOD == 1 ? 2 : 3));
    static constexpr int kWaves = C >= 5 ? SNOWTRI_DLT_WIDE_WAVES : (METHOD == 1 ? 3 : kFastWaves);
};

template <typename T>
struct Vec4T {
    T x, y, z, w;
};

// LDS bytes the fast kernel needs for T frames per tile.
constexpr size_t kFusedConstBytes = 8 * (12 * 8 + 3 * 28);  // M, t of <= 8 cameras + d of <= 28 pairs

__host__ __device__ constexpr size_t fused_single_lds_bytes(int T, int kn, int NP) {
    const size_t a = (((size_t)T * kn * 8 + (size_t)T * 4 + 16) + 15) & ~(size_t)15;
    const size_t b = (condense_lds_bytes(NP) + 15) & ~(size_t)15;
    return (a > b ? a : b) + kFusedConstBytes;
}

// ---- item cursor / fetch helpers of k_fused_single ------------------------------------------------
__device__ __forceinline__ void advance_item(int &fl, int &j, int dfl, int dj, int J) {
    fl += dfl;  // next item of this lane is kBlock items further: (fl, j) += (kBlock / J, kBlock % J)
    j += dj;
    if (j >= J) {
        j -= J;
        fl++;
    }
}

// Loads are issued UNCONDITIONALLY (item index clamped into the tile): the compiler can then count
// exactly how many vector-memory operations are younger than the buffer it is about to read and emits
// s_waitcnt vmcnt(N>0); with conditional fetches it falls back to vmcnt(0) and the ring is useless.
template <int C, typename TIn>
__device__ __forceinline__ void fetch_item(Kp3<TIn> (&dst)[C], const Kp3<TIn> *__restrict__ tile_in, int fl,
                                           int j, int J, unsigned last_off) {
    unsigned off = (unsigned)(fl * C * J + j);
    off = off < last_off ? off : last_off;
#pragma unroll
    for (int c = 0; c < C; c++) dst[c] = tile_in[off + (unsigned)(c * J)];
}

// first two items of this lane in tile `tile` -> bufA, bufB (a ring of two: the first item -> bufA)
template <int C, int RING, typename TIn>
__device__ __forceinline__ void prefetch_tile_head(Kp3<TIn> (&bufA)[C], Kp3<TIn> (&bufB)[RING >= 2 ? C : 1],
                                                   const Kp3<TIn> *__restrict__ kp3, int64_t tile, int T,
                                                   int64_t F, int J, int tid, int dfl, int dj) {
    const int64_t f0 = tile * T;
    const int nf = (int)((F - f0) < T ? (F - f0) : T);
    const unsigned last_off = (unsigned)((nf - 1) * C * J + J - 1);
    const Kp3<TIn> *tile_in = kp3 + f0 * C * (int64_t)J;
    int fl = tid / J, j = tid - fl * J;
    fetch_item<C>(bufA, tile_in, fl, j, J, last_off);
    if constexpr (RING >= 3) {
        advance_item(fl, j, dfl, dj, J);
        fetch_item<C>(bufB, tile_in, fl, j, J, last_off);
    }
}

// ------------------------------------------------------------------------------------------------
// One (frame, joint) of the single-detection fast path: C rays, all C(C,2) pair solves, fusion.
// Returns true if the item needs the IEEE-exact general routine (a pair whose dist^2 is not a
// comfortably normal positive number: exact intersection, singular pair, or NaN).
//
// Fusion is regrouped per RAY instead of per pair: with Wm + Ws = (t_m + t_s) + hm S0 - hs S1,
//   sum_q s_q (Wm+Ws)_q = sum_c ( alpha_c h_c + beta_c t_c ),
//   alpha_c = sum_{q: c=m} s_q S0_q - sum_{q: c=s} s_q S1_q,   beta_c = sum_{q contains c} s_q,
// so a pair costs 2 FMA + 2 adds here instead of 9 FMA, and sum_q s_q = (sum_c beta_c) / 2.
template <int C, typename TIn>
__device__ __forceinline__ bool pairwise_item(const Rig &rig, const double *__restrict__ Mlds,
                                              const Kp3<TIn> (&cur)[C], const Params &prm, double &ox,
                                              double &oy, double &oz, double &os) {
    // Products that are meant to be fused are explicit fma() calls; implicit contraction is off: the item is inlined
    // once per slot of the prefetch ring and the compiler is free to fuse `beta += x * y` differently in each copy,
    // which made a frame's last bit depend on where in the launch it sat (found by the 125 000-frame shard test).
#pragma clang fp contract(off)
    // Rig constants are wave-uniform and all come from LDS (broadcast reads into transient VGPRs):
    // ray matrices M[C][9], camera centres t[C][3], per-pair d = t_s - t_m.  Holding the ~70 doubles
    // in scalar registers instead overflows the SGPR file: loop-invariant scalars then live in
    // VGPR-lane spills and the remainder is re-fetched with serialised scalar loads every item.
    // Issue every LDS read of M and d up front into registers (sched_barrier keeps them there): one wait
    // per item instead of ~25 scattered ones (+2.5 % measured); t is read where it is used, at the end.
    double Mp[9 * C], pc[3 * (C * (C - 1) / 2)];
#pragma unroll
    for (int i = 0; i < 9 * C; i++) Mp[i] = Mlds[i];
#pragma unroll
    for (int i = 0; i < 3 * (C * (C - 1) / 2); i++) pc[i] = Mlds[12 * C + i];
    __builtin_amdgcn_sched_barrier(0);
    const double *tp = Mlds + 9 * C;
    Vec3 h[C];
    double a[C], alpha[C], beta[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        // A1, camera.py:241-243 with M = R inv(K)
        const double u = (double)cur[c].u, v = (double)cur[c].v;
        h[c].x = fma(Mp[9 * c + 0], u, fma(Mp[9 * c + 1], v, Mp[9 * c + 2]));
        h[c].y = fma(Mp[9 * c + 3], u, fma(Mp[9 * c + 4], v, Mp[9 * c + 5]));
        h[c].z = fma(Mp[9 * c + 6], u, fma(Mp[9 * c + 7], v, Mp[9 * c + 8]));
        a[c] = dot3(h[c], h[c]);
    }
    bool bad = false;
    int q = 0;
    // one reciprocal for all C(C,2) determinants (Montgomery's trick): 1/det_q from 1/prod(det) and prefix
    // products -- 3 multiplies per pair instead of a v_rcp_f64 (16 cycles) + two Newton steps.  A singular
    // pair poisons the product; every pair then reports dist^2 = NaN and the item takes the exact path.
    constexpr int NPc = C * (C - 1) / 2;
    double bq[NPc], detq[NPc], pre[NPc], invq[NPc];
#pragma unroll
    for (int mc = 0; mc < C - 1; mc++)
#pragma unroll
        for (int sc = mc + 1; sc < C; sc++, q++) {
            bq[q] = dot3(h[mc], h[sc]);
            detq[q] = a[mc] * a[sc] - bq[q] * bq[q];   // separately rounded products (contraction off): singular as the reference sees it (a c == b b) <=> 0 exactly, which poisons the product below; the fused form is the rounding error of b b there
            pre[q] = q == 0 ? detq[0] : pre[q - 1] * detq[q];
        }
    {
        double run = rcp_nr2(pre[NPc - 1]);
#pragma unroll
        for (int k = NPc - 1; k > 0; k--) {
            invq[k] = run * pre[k - 1];
            run *= detq[k];
        }
        invq[0] = run;
    }
    q = 0;
#pragma unroll
    for (int mc = 0; mc < C - 1; mc++) {
#pragma unroll
        for (int sc = mc + 1; sc < C; sc++, q++) {
            // A2 (triangulation.py:24-31): per-ray norms hoisted, d = ts - tm precomputed
            const Vec3 &hm = h[mc], &hs = h[sc];
            const Vec3 d = {pc[3 * q], pc[3 * q + 1], pc[3 * q + 2]};
            const double b = bq[q], inv = invq[q];
            const double e = dot3(hm, d), g = dot3(hs, d);
            const double S0 = fma(a[sc], e, -(b * g)) * inv;
            const double S1 = fma(a[mc], g, -(b * e)) * inv;
            // Wm - Ws = hm S0 + hs S1 - d
            const Vec3 df = {fma(hs.x, S1, fma(hm.x, S0, -d.x)), fma(hs.y, S1, fma(hm.y, S0, -d.y)),
                             fma(hs.z, S1, fma(hm.z, S0, -d.z))};
            const double d2 = dot3(df, df);
            const double idist = rsq_nr1(d2);
            // :72-74  score = ((sm+ss)/2) / (dist*1000), zeroed by the three gates.  The gate is applied to
            // the (element-typed) score sum: items whose 1/dist is not finite take the exact path anyway.
            const bool keep = !below_kthr(cur[mc].s, prm) && !below_kthr(cur[sc].s, prm) && !(d2 > prm.dthr2);
            const double sq = gated_sum(cur[mc].s, cur[sc].s, keep) * (idist * 0.0005);  // halving folded in
            bad |= !(d2 > 1e-280);  // exact intersection / singular / NaN / rsq out of range -> exact path
            // first touch of every accumulator happens in the pairs with mc == 0 (compile-time known):
            // no zero-initialisation, no wasted FMA
            if (mc == 0) {
                alpha[sc] = -sq * S1;
                beta[sc] = sq;
                if (sc == 1) {
                    alpha[0] = sq * S0;
                    beta[0] = sq;
                } else {
                    alpha[0] = fma(sq, S0, alpha[0]);
                    beta[0] += sq;
                }
            } else {
                alpha[mc] = fma(sq, S0, alpha[mc]);
                alpha[sc] = fma(-sq, S1, alpha[sc]);
                beta[mc] += sq;
                beta[sc] += sq;
            }
        }
    }
    double sx = alpha[0] * h[0].x, sy = alpha[0] * h[0].y, sz = alpha[0] * h[0].z, sb = beta[0];
    sx = fma(beta[0], tp[0], sx);
    sy = fma(beta[0], tp[1], sy);
    sz = fma(beta[0], tp[2], sz);
#pragma unroll
    for (int c = 1; c < C; c++) {
        sx = fma(alpha[c], h[c].x, fma(beta[c], tp[3 * c + 0], sx));
        sy = fma(alpha[c], h[c].y, fma(beta[c], tp[3 * c + 1], sy));
        sz = fma(alpha[c], h[c].z, fma(beta[c], tp[3 * c + 2], sz));
        sb += beta[c];
    }
    // sb = 2 sum_q s_q (:141).  sum == 0 -> the joint stays (0,0,0)/0 (:142-143): then sx = sy = sz = 0 too,
    // so zeroing the reciprocal is enough.
    double r = rcp_nr1(sb);  // 1 / (2 sum s): the 1/2 of the midpoint folded in
    r = (sb == 0.0) ? 0.0 : r;
    ox = sx * r;  // :144-147 as (sum s (Wm+Ws)) / (2 sum s)
    oy = sy * r;
    oz = sz * r;
    os = sb * (0.5 / (double)(C * (C - 1) / 2));  // :148
    return bad;
}

// ------------------------------------------------------------------------------------------------
// method = SNOWTRI_DLT (row N3; what north_star describes, NOT what the reference computes):
// N-view DLT for one (frame, joint) in one lane.  Rows u*P[2]-P[0], v*P[2]-P[1] of every camera whose
// confidence is not below keypoint_score_threshold are accumulated straight into the 10 unique
// entries of A^T A; its smallest eigenvector comes from shifted inverse iteration on a register-resident Cholesky
// factor (dlt_inverse_iteration), and from a register-resident cyclic Jacobi for the lanes that do not settle there
// (dlt_min_eigenvector: gross outliers; ghost clusters of nearly parallel rays, whose two smallest eigenvalues coincide).
// One observation (u, v) of a camera with world->pixel matrix P (12 doubles): adds the two rows
// u P[2] - P[0], v P[2] - P[1] (scaled by w in {0, 1}) to the upper triangle of A^T A.
// MASKED = false: the caller knows w == 1 in every lane of the wave (1.0 * x is exact: the same bits without the eight products).
// FIRST: A holds nothing yet -- the entries are SET (r1 r1' + r2 r2' without the ten zeros and their additions).
template <bool MASKED = true, bool FIRST = false, typename PPtr>
__device__ __forceinline__ void dlt_add_observation(double (&A)[4][4], PPtr P, double u, double v, double w) {
#pragma clang fp contract(off)   // results must not depend on which inlined copy computes them (see pairwise_item)
    double r1[4], r2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        r1[k] = fma(u, P[8 + k], -P[k]);
        r2[k] = fma(v, P[8 + k], -P[4 + k]);
        if constexpr (MASKED) {
            r1[k] = w * r1[k];
            r2[k] = w * r2[k];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = i; k < 4; k++) A[i][k] = FIRST ? fma(r1[i], r1[k], r2[i] * r2[k]) : fma(r1[i], r1[k], fma(r2[i], r2[k], A[i][k]));
}

constexpr int kJacobiSweeps = 6;
// Eigenvector of the smallest eigenvalue of the symmetric 4x4 whose upper triangle is in A: cyclic Jacobi,
// register resident, at most kJacobiSweeps sweeps (converged to 2e-14 m after 5 on the bench rig; a lane whose off-diagonal
// part is gone stops rotating, a wave leaves when all its lanes have).
// Rotation: t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), c = 1/sqrt(t^2 + 1), s = t c, with
// rcp/rsq + Newton instead of IEEE divide/sqrt.  Those helpers return NaN on denormal inputs, and off-diagonal
// entries decay THROUGH the denormal range on their way to zero: an entry below 1e-280 is treated as already
// zero (its rotation angle is below 1e-280 / gap, i.e. nothing), and |theta| is clamped so theta^2 stays finite.
__device__ __forceinline__ void dlt_min_eigenvector(double (&A)[4][4], double (&e)[4]) {
#pragma clang fp contract(off)   // results must not depend on which inlined copy computes them (see pairwise_item)
#pragma unroll
    for (int i = 1; i < 4; i++)
#pragma unroll
        for (int k = 0; k < i; k++) A[i][k] = A[k][i];
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) V[i][k] = (i == k) ? 1.0 : 0.0;
    bool done = false;   // the lane's off-diagonal part is gone: it rotates no more (its answer must not depend on how long its wave goes on)
#pragma unroll 1
    for (int sweep = 0; sweep < kJacobiSweeps; sweep++) {
        // off-diagonal against diagonal, squared: below 1e-36 the rotations left move the eigenvector by less than 1e-18 x the
        // spectrum's spread over the gap -- nothing in 53 bits; a wave leaves when all its lanes are there (usually after 4 of the 6 sweeps)
        const double off2 = fma(A[0][1], A[0][1], fma(A[0][2], A[0][2], fma(A[0][3], A[0][3], fma(A[1][2], A[1][2], fma(A[1][3], A[1][3], A[2][3] * A[2][3])))));
        const double dia2 = fma(A[0][0], A[0][0], fma(A[1][1], A[1][1], fma(A[2][2], A[2][2], A[3][3] * A[3][3])));
        done = done || !(off2 > 1e-36 * dia2);
        if (__all(done)) break;
#pragma unroll
        for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                const double apq = A[p][q];
                const bool rot = !done && fabs(apq) > 1e-280;
                const double theta = (A[q][q] - A[p][p]) * (0.5 * rcp_nr2(rot ? apq : 1.0));
                const double at = fmin(fabs(theta), 1e150);
                double t = rcp_nr2(at + sqrt_nr(fma(at, at, 1.0)));
                t = copysign(t, theta);
                t = rot ? t : 0.0;
                const double cth = rsq_nr2(fma(t, t, 1.0));
                const double sth = t * cth;
                A[p][p] = fma(-t, apq, A[p][p]);
                A[q][q] = fma(t, apq, A[q][q]);
                A[p][q] = A[q][p] = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r != p && r != q) {
                        const double arp = A[r][p], arq = A[r][q];
                        A[r][p] = A[p][r] = fma(cth, arp, -(sth * arq));
                        A[r][q] = A[q][r] = fma(sth, arp, cth * arq);
                    }
                    const double vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = fma(cth, vrp, -(sth * vrq));
                    V[r][q] = fma(sth, vrp, cth * vrq);
                }
            }
        }
    }
    double best = A[0][0];
    e[0] = V[0][0]; e[1] = V[1][0]; e[2] = V[2][0]; e[3] = V[3][0];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const bool lt = A[k][k] < best;
        best = lt ? A[k][k] : best;
        e[0] = lt ? V[0][k] : e[0];
        e[1] = lt ? V[1][k] : e[1];
        e[2] = lt ? V[2][k] : e[2];
        e[3] = lt ? V[3][k] : e[3];
    }
}

constexpr int kDltInvit = 8;
// The same eigenvector by shifted inverse iteration: G = A^T A + mu I = L D L^T (mu = 64 eps trace keeps
// every pivot positive when the data are exact and A^T A is singular), x <- normalise(G^-1 x) from e_4.
// The wanted eigenvalue is the squared reprojection residual (tiny), the next one is ~1e4..1e5 times larger at
// one pixel of noise, so four steps reach 2e-14 m (same as the SVD oracle) -- ~15x fewer instructions than the
// Jacobi sweeps.  Convergence is linear, so a lane is done when (step length)^2 / (previous step length), the
// estimate of the error left, drops below 1e-14.  The first FOUR steps run without a test (round 6: a step with its
// normalisation, step length, test, selects and wave vote was 70 instructions for 20 of solve; the first step from e_4 is
// half a solve, the second needs no normalisation, the step lengths of steps 3 and 4 decide) -- 4 x 70 -> 140; only a wave
// with a lane that has not settled by then goes on, step by step, up to kDltInvit; lanes that have not by then
// (gross outliers: eigenvalue ratio above ~0.02) report false and the caller re-solves them with Jacobi.
// `live` = false lanes (fewer than two cameras) never block.
__device__ __forceinline__ bool dlt_inverse_iteration(const double (&A)[4][4], bool live, double (&e)[4]) {
#pragma clang fp contract(off)   // results must not depend on which inlined copy computes them (see pairwise_item)
    const double mu = (A[0][0] + A[1][1] + A[2][2] + A[3][3]) * (64.0 * 2.220446049250313e-16);
    // G = L D L^T with a UNIT lower triangle (round 6; before: Cholesky): four reciprocals (v_rcp_f64 + two Newton steps, 5 instructions)
    // where the square-root form took four reciprocal square roots (9 each), and a solve is 12 fma + 4 products instead of 12 + 8
    // A pivot that is not positive (A^T A singular to working precision: ghost clusters of nearly parallel rays) made the square-root form
    // NaN and sent the lane to Jacobi; here it would iterate on an indefinite matrix: `posdef` keeps such a lane unsettled.
    double L[4][4], W[4][4], inv[4];  // L strictly-lower entries; W[i][j] = L[i][j] d_j (the entry before its division); inv[j] = 1 / d_j
    bool posdef = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double d = A[j][j] + mu;
#pragma unroll
        for (int k = 0; k < j; k++) d = fma(-W[j][k], L[j][k], d);
        posdef = posdef && d > 0.0;
        inv[j] = rcp_nr2(d);
#pragma unroll
        for (int i = j + 1; i < 4; i++) {
            double v = A[j][i];
#pragma unroll
            for (int k = 0; k < j; k++) v = fma(-L[i][k], W[j][k], v);
            W[i][j] = v;
            L[i][j] = v * inv[j];
        }
    }
    auto solve = [&](const double (&x)[4], double (&z)[4]) {   // z = G^-1 x
        double y[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {  // L y = x
            double v = x[i];
#pragma unroll
            for (int k = 0; k < i; k++) v = fma(-L[i][k], y[k], v);
            y[i] = v;
        }
#pragma unroll
        for (int i = 3; i >= 0; i--) {  // L^T z = D^-1 y
            double v = y[i] * inv[i];
#pragma unroll
            for (int k = i + 1; k < 4; k++) v = fma(-L[k][i], z[k], v);
            z[i] = v;
        }
    };
    auto unit = [](const double (&z)[4], double (&n)[4]) {
        const double rn = rsq_nr2(fma(z[3], z[3], fma(z[2], z[2], fma(z[1], z[1], z[0] * z[0]))));
#pragma unroll
        for (int i = 0; i < 4; i++) n[i] = z[i] * rn;
    };
    auto step_length = [](const double (&a)[4], const double (&b)[4]) {
        return fmax(fmax(fabs(a[0] - b[0]), fabs(a[1] - b[1])), fmax(fabs(a[2] - b[2]), fabs(a[3] - b[3])));
    };
    double x[4], z[4], n[4];
    // step 1 from e_4: L y = e_4 has y = e_4, D^-1 y = (0, 0, 0, 1 / d_3), so only the back substitution is left
    z[3] = inv[3];
    z[2] = -(L[3][2] * z[3]);
    z[1] = fma(-L[3][1], z[3], -(L[2][1] * z[2]));
    z[0] = fma(-L[3][0], z[3], fma(-L[2][0], z[2], -(L[1][0] * z[1])));
    solve(z, x);          // step 2 (the length of its input does not matter)
    unit(x, n);
    solve(n, z);          // step 3
    unit(z, x);
    const double d3 = step_length(x, n);
    solve(x, z);          // step 4
    unit(z, n);
    double prev = step_length(n, x);
    // linear convergence with ratio r = step / previous step: the error left after a step is ~ step * r
    bool conv = !live || (posdef && prev * prev < 1e-14 * d3);
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = n[i];
    if (!__all(conv || !posdef)) {
#pragma unroll 1
        for (int it = 4; it < kDltInvit; it++) {
            solve(x, z);
            unit(z, n);
            const double diff = step_length(n, x);
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = conv ? x[i] : n[i];   // a settled lane keeps its answer: it must not depend on how long its wave iterates
            conv = conv || (posdef && diff * diff < 1e-14 * prev);
            prev = diff;
            if (__all(conv || !posdef)) break;   // (a lane without a factor never settles: Jacobi takes it)
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) e[i] = x[i];
    return conv;
}

// smallest eigenvector: inverse iteration, Jacobi for the lanes (whole wave executes it) that did not settle
__device__ __forceinline__ void dlt_solve(double (&A)[4][4], bool live, double (&e)[4]) {
    const bool conv = dlt_inverse_iteration(A, live, e);
    if (__any(!conv)) {
        double ej[4];
        dlt_min_eigenvector(A, ej);
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = conv ? e[i] : ej[i];
    }
}

// X = e / e_3 and the mean confidence: reciprocals by v_rcp_f64 + two Newton steps (~1 ulp; the IEEE divisions were 2 x ~14
// instructions per joint), 1 / 0 kept as the division has it
__device__ __forceinline__ double dlt_recip(double x) { return x == 0.0 ? copysign(__builtin_inf(), x) : rcp_nr2(x); }

// npmask: bit c set = camera c lists a detection in this frame (all ones without an n_persons array)
template <int C, typename TIn>
__device__ __forceinline__ void dlt_item(const double *__restrict__ Plds, const Kp3<TIn> (&cur)[C], uint32_t npmask,
                                         const Params &prm, double &ox, double &oy, double &oz, double &os) {
#pragma clang fp contract(off)   // (same reason as in pairwise_item)
    // world->pixel matrices P[C][12] come from LDS (broadcast reads), like the ray matrices of the pairwise item, a camera's
    // matrix where its observation is added: 96 doubles in scalar registers overflow the SGPR file and come back as v_readlane
    // traffic, in vector registers they cost a wave per SIMD
    double A[4][4];   // (upper triangle; the first camera's observation sets it)
    double ssum = 0.0;
    int cnt = 0;
    bool use[C];
    bool every = true;
#pragma unroll
    for (int c = 0; c < C; c++) {
        use[c] = !((double)cur[c].s < prm.kthr) && ((npmask >> c) & 1u);
        every = every && use[c];
    }
    auto accumulate = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
        for (int c = 0; c < C; c++) {
            __builtin_amdgcn_sched_barrier(0);   // a camera's twelve LDS reads stay with its observation (register budget)
            if (c == 0)
                dlt_add_observation<MASKED, true>(A, Plds + 12 * c, (double)cur[c].u, (double)cur[c].v, use[c] ? 1.0 : 0.0);
            else
                dlt_add_observation<MASKED, false>(A, Plds + 12 * c, (double)cur[c].u, (double)cur[c].v, use[c] ? 1.0 : 0.0);
        }
    };
    // every camera of every lane counts (the usual wave): the rows are added as they are, no product with the mask, and the
    // confidences are summed as they are (+ 0.0 for a gated one is exact: the same sum either way)
    if (__all(every)) {
        accumulate(std::false_type{});
#pragma unroll
        for (int c = 0; c < C; c++) ssum += (double)cur[c].s;
        cnt = C;
        asm volatile("" : "+v"(cnt));   // (1 / cnt below by the SAME instructions as in the other branch: a constant here would be folded, and differently rounded)
    } else {
        accumulate(std::true_type{});
#pragma unroll
        for (int c = 0; c < C; c++) {
            ssum += use[c] ? (double)cur[c].s : 0.0;
            cnt += use[c] ? 1 : 0;
        }
    }
    const bool ok = cnt >= 2;
    double e[4];
    dlt_solve(A, ok, e);
    const double r = dlt_recip(e[3]);
    ox = ok ? e[0] * r : 0.0;
    oy = ok ? e[1] * r : 0.0;
    oz = ok ? e[2] * r : 0.0;
    os = ok ? ssum * dlt_recip((double)cnt) : 0.0;
}

template <int C, int METHOD, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, (FusedShape<C, METHOD, TIn>::kWaves)) void k_fused_single(int64_t F, int J, int T, Rig rig,
                                                         const TIn *__restrict__ kpts,
                                                         const int32_t *__restrict__ n_persons, Params prm,
                                                         int Pout, TOut *__restrict__ out4,
                                                         TOut *__restrict__ out_ps,
                                                         int32_t *__restrict__ out_count,
                                                         uint32_t *__restrict__ out_flags, char *scratch,
                                                         size_t scratch_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = C * (C - 1) / 2;
    const int kn = prm.kn, ci = prm.center;
    double *stash = reinterpret_cast<double *>(smem);                        // [T][kn] fused joint scores
    uint32_t *fflag = reinterpret_cast<uint32_t *>(stash + (size_t)T * kn);  // [T]
    // rig constants M[C][9], t[C][3], d[NP][3] live at the very end of the allocation
    // (general_frame reuses the front)
    double *Mlds = reinterpret_cast<double *>(smem + fused_single_lds_bytes(T, kn, NP) - kFusedConstBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const PackedWriter<TOut> wr{out4, out_ps};
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    if constexpr (METHOD == 1) {
        if (tid < 12 * C) Mlds[tid] = rig.P[tid];  // DLT: the 12 C doubles of M and t hold P instead
    } else {
        if (tid < 9 * C) Mlds[tid] = rig.M[tid];  // visible after the first __syncthreads() below
        if (tid < 3 * C) Mlds[9 * C + tid] = rig.t[tid];
        if (tid < 3 * NP) Mlds[12 * C + tid] = rig.pairc[6 * (tid / 3) + tid % 3];
    }
    const int dfl = kBlock / J, dj = kBlock - dfl * J;
    const int64_t ntiles = (F + T - 1) / T;
    using Shape = FusedShape<C, METHOD, TIn>;
    constexpr int kRing = Shape::kRing;

    Kp3<TIn> bufA[C], bufB[kRing >= 2 ? C : 1], bufC[kRing >= 3 ? C : 1];
    if (blockIdx.x < ntiles) prefetch_tile_head<C, kRing>(bufA, bufB, kp3, (int64_t)blockIdx.x, T, F, J, tid, dfl, dj);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t f0 = tile * T;
        const int nf = (int)((F - f0) < T ? (F - f0) : T);
        const int nitems = nf * J;
        // (DLT: bits 16.. = the cameras that list a detection in the frame -- read once per frame here, not once per item)
        for (int i = tid; i < nf; i += kBlock) {
            uint32_t m = 0u;
            if constexpr (METHOD == 1) {
                m = 0xffff0000u;
                if (n_persons) {
                    m = 0u;
                    for (int c = 0; c < C; c++) m |= n_persons[(f0 + i) * C + c] > 0 ? (0x10000u << c) : 0u;
                }
            }
            fflag[i] = m;
        }
        // centre joints for the single-cluster check of the epilogue: fetched now, used after the item loop
        Kp3<TIn> ck[4];
        bool have_centres = false;
        if (METHOD == 0 && tid < nf * (NP - 1)) {
            const int w = tid / (NP - 1), qq = 1 + (tid - w * (NP - 1));
            const Kp3<TIn> *p = kp3 + ((f0 + w) * C) * (int64_t)J + ci;
            ck[0] = p[(size_t)rig.pairs[0] * J];
            ck[1] = p[(size_t)rig.pairs[1] * J];
            ck[2] = p[(size_t)rig.pairs[2 * qq] * J];
            ck[3] = p[(size_t)rig.pairs[2 * qq + 1] * J];
            have_centres = true;
        }
        __syncthreads();

        // ---- main loop: one lane per (frame, joint).  A ring of three register buffers keeps the
        //      keypoints of the next TWO items in flight while the current one is solved: with only
        //      two waves per SIMD the kernel is otherwise bound by bytes-in-flight / HBM latency.
        //      The first two items of a tile were fetched before the previous tile's epilogue.
        {
            auto solve_store = [&](const Kp3<TIn>(&buf)[C], int fl_, int j_) {
                double ox, oy, oz, os;
                bool bad;
                if constexpr (METHOD == 0) {
                    static_assert(C <= 4, "the pairwise item of k_fused_single keeps everything in registers: up to six pairs");
                    bad = pairwise_item<C>(rig, Mlds, buf, prm, ox, oy, oz, os);
                } else {
                    bad = false;
                    if constexpr (C >= 5) asm volatile("" ::: "memory");   // (P is re-read from LDS by every item: hoisted out of the loop it takes 12 C doubles)
                    dlt_item<C>(Mlds, buf, fflag[fl_] >> 16, prm, ox, oy, oz, os);
                }
                if (j_ < kn) {
                    Vec4T<TOut> *tile_out = reinterpret_cast<Vec4T<TOut> *>(out4) + f0 * Pout * (int64_t)kn;
                    const unsigned o = (unsigned)(fl_ * Pout * kn + j_);
                    tile_out[o] = Vec4T<TOut>{(TOut)ox, (TOut)oy, (TOut)oz, (TOut)os};
                    stash[fl_ * kn + j_] = os;
                }
                if (bad) atomicOr(&fflag[fl_], kSlow);
            };
            const Kp3<TIn> *tile_in = kp3 + f0 * C * (int64_t)J;  // wave-uniform base, 32-bit lane offsets
            const unsigned last_off = (unsigned)((nf - 1) * C * J + J - 1);
            // item k of this lane is tid + k*kBlock -> (fl, j); two cursors walk the ring
            const int n_my = tid < nitems ? (nitems - tid + kBlock - 1) / kBlock : 0;
            int fs = tid / J, js = tid - fs * J;  // item being solved
            int ff = fs, jf = js;                 // item being fetched: kRing - 1 ahead
            advance_item(ff, jf, dfl, dj, J);
            if constexpr (kRing >= 3) {
                advance_item(ff, jf, dfl, dj, J);
                for (int k = 0; k < n_my; k += 3) {
                    fetch_item<C>(bufC, tile_in, ff, jf, J, last_off);
                    solve_store(bufA, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    advance_item(ff, jf, dfl, dj, J);
                    fetch_item<C>(bufA, tile_in, ff, jf, J, last_off);
                    if (k + 1 < n_my) solve_store(bufB, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    advance_item(ff, jf, dfl, dj, J);
                    fetch_item<C>(bufB, tile_in, ff, jf, J, last_off);
                    if (k + 2 < n_my) solve_store(bufC, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    advance_item(ff, jf, dfl, dj, J);
                }
            } else if constexpr (kRing == 1) {   // one buffer: the other waves of the SIMD cover the fetch
                for (int k = 0; k < n_my; k++) {
                    solve_store(bufA, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    fetch_item<C>(bufA, tile_in, fs, js, J, last_off);
                }
            } else {   // two buffers: the next item's keypoints fly under the current item
                for (int k = 0; k < n_my; k += 2) {
                    fetch_item<C>(bufB, tile_in, ff, jf, J, last_off);
                    solve_store(bufA, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    advance_item(ff, jf, dfl, dj, J);
                    fetch_item<C>(bufA, tile_in, ff, jf, J, last_off);
                    if (k + 1 < n_my) solve_store(bufB, fs, js);
                    advance_item(fs, js, dfl, dj, J);
                    advance_item(ff, jf, dfl, dj, J);
                }
            }
            // unused person slots are zero-filled here, NOT inside the item loop: a store loop with a
            // run-time trip count there makes the compiler's vmcnt bookkeeping give up and wait for
            // every outstanding load (vmcnt(0)) at each item, which defeats the prefetch ring
            if (Pout > 1 && !prm.no_zero_fill) {
                Vec4T<TOut> *tile_out = reinterpret_cast<Vec4T<TOut> *>(out4) + f0 * Pout * (int64_t)kn;
                const int per = (Pout - 1) * kn;
                for (int i = tid; i < nf * per; i += kBlock) {
                    const int w = i / per, r = i - w * per;
                    tile_out[(unsigned)(w * Pout * kn + kn + r)] = Vec4T<TOut>{(TOut)0, (TOut)0, (TOut)0, (TOut)0};
                }
            }
            // next tile of this workgroup: start its first two fetches now, they fly during the epilogue
            const int64_t nt = tile + gridDim.x;
            if (nt < ntiles) prefetch_tile_head<C, kRing>(bufA, bufB, kp3, nt, T, F, J, tid, dfl, dj);  // wave-uniform
        }

        // ---- single-cluster check (:116-130): candidate q >= 1 must have its centre joint within
        //      condense_distance_tol of candidate 0's.  One lane per (frame, q); ~1 % of the work.
        for (int i = tid; METHOD == 0 && i < nf * (NP - 1); i += kBlock) {
            const int w = i / (NP - 1), qq = 1 + (i - w * (NP - 1));
            const Kp3<TIn> *p = kp3 + ((f0 + w) * C) * (int64_t)J + ci;
            const bool pre = i == tid && have_centres;   // first pass: keypoints fetched before the item loop
            auto centre_of = [&](int qk, int slot) {
                const int mc = rig.pairs[2 * qk], sc = rig.pairs[2 * qk + 1];
                const Kp3<TIn> km = pre ? ck[2 * slot] : p[(size_t)mc * J], ks = pre ? ck[2 * slot + 1] : p[(size_t)sc * J];
                const double *pcq = rig.pairc + 6 * qk;
                const PairSolve o = pair_solve_fast<true>(make_ray(rig.M + 9 * mc, km.u, km.v),
                                                          make_ray(rig.M + 9 * sc, ks.u, ks.v),
                                                          Vec3{pcq[0], pcq[1], pcq[2]}, Vec3{pcq[3], pcq[4], pcq[5]});
                return o.sw;  // = 2 W
            };
            const Vec3 w0 = centre_of(0, 0);
            const Vec3 wq = centre_of(qq, 1);
            const double dx = 0.5 * (w0.x - wq.x), dy = 0.5 * (w0.y - wq.y), dz = 0.5 * (w0.z - wq.z);
            const double cd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));  // :124
            if (cd > prm.ctol) atomicOr(&fflag[w], kSlow);                // :125
        }
        __syncthreads();

        // ---- per-frame epilogue: mean fused score (:150), filters, count.  Eight lanes per frame
        //      (32 frames per pass of the workgroup) sum the stash and combine with three shuffles.
        {
            constexpr int G = 8;
            const int sub = tid & (G - 1);
            for (int base = 0; base < nf; base += kBlock / G) {
                const int w = base + tid / G;
                const bool live = w < nf;
                const int64_t f = f0 + (live ? w : 0);
                double sum = 0.0;
                if (live) {
                    // four independent partial sums: the LDS reads of one round trip are issued together
                    // (a single running sum pays the LDS latency once per element)
                    const double *row = stash + w * kn;
                    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int b = sub;
                    for (; b + 3 * G < kn; b += 4 * G) {
                        const double v0 = row[b], v1 = row[b + G], v2 = row[b + 2 * G], v3 = row[b + 3 * G];
                        sum += v0;
                        s1 += v1;
                        s2 += v2;
                        s3 += v3;
                    }
                    for (; b < kn; b += G) sum += row[b];
                    sum = (sum + s1) + (s2 + s3);
                }
                int not_one = 0;
                if (METHOD == 0 && n_persons && live && sub < C) not_one = n_persons[f * C + sub] != 1;
#pragma unroll
                for (int off = G / 2; off > 0; off >>= 1) {
                    sum += __shfl_xor(sum, off, 64);
                    not_one |= __shfl_xor(not_one, off, 64);
                }
                if (live && sub == 0) {
                    const double avg = sum / (double)kn;
                    bool slow = (fflag[w] & kSlow) != 0u || (avg < prm.score_tol) || not_one != 0;  // :151-152
                    if (METHOD != 0) slow = false;
                    if (slow) {
                        fflag[w] |= kSlow;
                    } else {
                        out_count[f] = 1;
                        if (out_ps) {
                            out_ps[f * Pout] = (TOut)avg;
                            for (int slot = 1; slot < Pout; slot++) out_ps[f * Pout + slot] = (TOut)0;
                        }
                        if (out_flags) out_flags[f] = kFlagFast;
                    }
                }
            }
        }
        __syncthreads();
        // ---- rare: frames the speculation could not resolve -> the reference's full algorithm
        unsigned long long slow_mask = __ballot(lane < nf && (fflag[lane] & kSlow) != 0u);  // T <= 64
        __syncthreads();
        if constexpr (METHOD != 0) slow_mask = 0ull;   // (DLT never speculates: the general routine is not compiled into those kernels)
        if (slow_mask) {
            double *slab = reinterpret_cast<double *>(scratch + (size_t)blockIdx.x * scratch_per_block);
            while (slow_mask) {
                const int w = __ffsll((long long)slow_mask) - 1;
                slow_mask &= slow_mask - 1ull;
                general_frame<TIn>(f0 + w, 1, J, NP, rig, kpts, n_persons, prm, Pout, wr, out_count, out_flags,
                                   slab, smem);
            }
        }
    }
}

}  // namespace snowtri

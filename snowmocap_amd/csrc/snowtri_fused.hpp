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
#include "snowtri_kernels.hpp"

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
    if (sing && out_flags) atomicOr(&out_flags[f], kFlagSingular);
    __syncthreads();
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
template <typename T>
struct Vec4T {
    T x, y, z, w;
};

template <int C, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock) void k_fused_single(int64_t F, int J, int T, Rig rig,
                                                         const TIn *__restrict__ kpts,
                                                         const int32_t *__restrict__ n_persons, Params prm,
                                                         int Pout, TOut *__restrict__ out4,
                                                         TOut *__restrict__ out_ps,
                                                         int32_t *__restrict__ out_count,
                                                         uint32_t *__restrict__ out_flags,
                                                         unsigned long long *counters, char *scratch,
                                                         size_t scratch_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = C * (C - 1) / 2;
    const int kn = prm.kn, ci = prm.center;
    double *stash = reinterpret_cast<double *>(smem);           // [T][kn] fused joint scores
    uint32_t *fflag = reinterpret_cast<uint32_t *>(stash + (size_t)T * kn);  // [T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PackedWriter<TOut> wr{out4, out_ps};

    // Rig constants (M[C][9], t[C][3]) are wave-uniform: they are fetched with scalar loads INSIDE
    // the item loop (pointer laundered so the loads cannot be hoisted) -- hoisting them keeps
    // 48 doubles live across the whole loop and costs ~100 VGPRs.
    const double inv_np = 1.0 / (double)NP;
    const int dfl = kBlock / J, dj = kBlock - dfl * J;
    const int64_t ntiles = (F + T - 1) / T;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t f0 = tile * T;
        const int nf = (int)((F - f0) < T ? (F - f0) : T);
        const int nitems = nf * J;
        for (int i = tid; i < nf; i += kBlock) fflag[i] = 0;
        __syncthreads();
        int fl = tid / J, j = tid - fl * J;
        for (int it = tid; it < nitems; it += kBlock) {
            const int64_t f = f0 + fl;
            const TIn *kp = kpts + ((f * C) * (int64_t)J + j) * 3;
            const double *Mp = rig.M, *tp = rig.t;
            asm volatile("" : "+s"(Mp), "+s"(tp));
            Vec3 tc[C];
#pragma unroll
            for (int c = 0; c < C; c++) tc[c] = {tp[3 * c], tp[3 * c + 1], tp[3 * c + 2]};
            Vec3 h[C];
            double a[C];
            TIn s[C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                const TIn u = kp[(size_t)c * J * 3], v = kp[(size_t)c * J * 3 + 1];
                s[c] = kp[(size_t)c * J * 3 + 2];
                h[c] = ray_from_pixel(Mp + 9 * c, (double)u, (double)v);
                a[c] = dot3(h[c], h[c]);
            }
            const bool want_centre = __ballot(j == ci) != 0ull;  // wave-uniform
            double accS = 0.0, accX = 0.0, accY = 0.0, accZ = 0.0;
            Vec3 W0 = {0.0, 0.0, 0.0};
            bool bad = false, sing = false;
            int q = 0;
#pragma unroll
            for (int mc = 0; mc < C - 1; mc++) {
#pragma unroll
                for (int sc = mc + 1; sc < C; sc++, q++) {
                    // A2 with the per-ray norms hoisted (triangulation.py:24-31)
                    const Vec3 &hm = h[mc], &hs = h[sc];
                    const double b = dot3(hm, hs);
                    const double det = fma(a[mc], a[sc], -(b * b));
                    const Vec3 d = {tc[sc].x - tc[mc].x, tc[sc].y - tc[mc].y, tc[sc].z - tc[mc].z};
                    const double e = dot3(hm, d), g = dot3(hs, d);
                    const double inv = 1.0 / det;
                    const double S0 = fma(a[sc], e, -(b * g)) * inv;
                    const double S1 = fma(a[mc], g, -(b * e)) * inv;
                    const Vec3 Wm = {fma(hm.x, S0, tc[mc].x), fma(hm.y, S0, tc[mc].y), fma(hm.z, S0, tc[mc].z)};
                    const Vec3 Ws = {fma(-hs.x, S1, tc[sc].x), fma(-hs.y, S1, tc[sc].y), fma(-hs.z, S1, tc[sc].z)};
                    const Vec3 df = {Wm.x - Ws.x, Wm.y - Ws.y, Wm.z - Ws.z};
                    const double dist = sqrt(dot3(df, df));
                    const Vec3 W = {0.5 * (Wm.x + Ws.x), 0.5 * (Wm.y + Ws.y), 0.5 * (Wm.z + Ws.z)};
                    const double sq = pair_score(s[mc], s[sc], dist, prm);  // :72-74
                    sing |= (det == 0.0);
                    bad |= (sq < 0.0);
                    accS += sq;  // fusion, :141-147, as (sum s W) / (sum s)
                    accX = fma(sq, W.x, accX);
                    accY = fma(sq, W.y, accY);
                    accZ = fma(sq, W.z, accZ);
                    if (q == 0) {
                        W0 = W;
                    } else if (want_centre) {
                        const double dx = W0.x - W.x, dy = W0.y - W.y, dz = W0.z - W.z;
                        const double cd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));  // :124
                        bad |= (j == ci) && (cd > prm.ctol);                          // :125
                    }
                }
            }
            double ox = 0.0, oy = 0.0, oz = 0.0, os = 0.0;
            if (!(accS == 0.0)) {  // :142-143
                const double r = 1.0 / accS;
                ox = accX * r;
                oy = accY * r;
                oz = accZ * r;
                os = accS * inv_np;  // :148
            }
            if (j < kn) {
                Vec4T<TOut> o4 = {(TOut)ox, (TOut)oy, (TOut)oz, (TOut)os};
                *reinterpret_cast<Vec4T<TOut> *>(out4 + ((f * Pout) * (int64_t)kn + j) * 4) = o4;
                for (int slot = 1; slot < Pout; slot++) {
                    Vec4T<TOut> z4 = {(TOut)0, (TOut)0, (TOut)0, (TOut)0};
                    *reinterpret_cast<Vec4T<TOut> *>(out4 + ((f * Pout + slot) * (int64_t)kn + j) * 4) = z4;
                }
                stash[fl * kn + j] = os;
            }
            if (bad | sing) atomicOr(&fflag[fl], kSlow | (sing ? kFlagSingular : 0u));
            fl += dfl;
            j += dj;
            if (j >= J) {
                j -= J;
                fl++;
            }
        }
        __syncthreads();
        // per-frame epilogue: mean fused score (:150), filters, count; one wave per frame
        for (int w = wave; w < nf; w += kBlock / 64) {
            const int64_t f = f0 + w;
            double sum = 0.0;
            for (int b = lane; b < kn; b += 64) sum += stash[w * kn + b];
            sum = wave_sum(sum);
            const double avg = sum / (double)kn;
            bool slow = (fflag[w] & kSlow) != 0u || (avg < prm.score_tol);  // :151-152
            if (n_persons) {
                const bool one = lane < C ? (n_persons[f * C + lane] == 1) : true;
                slow |= (__ballot(!one) != 0ull);
            }
            if (lane == 0) {
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
        __syncthreads();
        // rare: frames the speculation could not resolve -> the reference's full algorithm
        unsigned long long slow_mask = 0ull;
        for (int w = 0; w < nf; w++)
            if (fflag[w] & kSlow) slow_mask |= 1ull << w;
        __syncthreads();
        if (slow_mask) {
            double *slab = reinterpret_cast<double *>(scratch + (size_t)blockIdx.x * scratch_per_block);
            if (tid == 0) atomicAdd(&counters[1], (unsigned long long)__popcll(slow_mask));
            while (slow_mask) {
                const int w = __ffsll((long long)slow_mask) - 1;
                slow_mask &= slow_mask - 1ull;
                general_frame<TIn>(f0 + w, 1, J, NP, rig, kpts, n_persons, prm, Pout, wr, out_count, out_flags,
                                   slab, smem);
            }
        }
    }
}

// LDS bytes the fast kernel needs for T frames per tile.
__host__ __device__ constexpr size_t fused_single_lds_bytes(int T, int kn, int NP) {
    const size_t a = (size_t)T * kn * 8 + (size_t)T * 4 + 16;
    const size_t b = condense_lds_bytes(NP);
    return a > b ? a : b;
}

}  // namespace snowtri

// snowtri_lean.hpp -- k_fused_lean: the production shape of the fast path (A1..A4 of SURVEY.md §8a in one launch).
//
// Same algorithm, speculation and fall-back as k_fused_single (snowtri_fused.hpp) for the case the bench and the
// shipped configuration run: pairwise method, FLOAT32 outputs, one detection per camera, one output slot, and
// keypoint_num == J known at compile time (JC = 133, the Wholebody skeleton of main.py).  float64 outputs and
// every other shape stay on k_fused_single, bit for bit as before.
//
// What is different, and why (the kernel is bound by VALU issue):
//   * WAVE-AUTONOMOUS tiles.  A wave owns whole frames (<= kLeanTw of them per tile): its item loop, the
//     single-cluster check and the per-frame mean score never leave the wave, so there is no __syncthreads()
//     per tile and the prologue / epilogue of one wave overlap the item loops of the others on the same SIMD.
//     The only workgroup barrier is at the very end, before the (rare) frames the speculation could not
//     resolve are re-done by general_frame.
//   * Tiles are cut so that every wave of the launch gets the same number of frames +-1 (a 10 000-frame
//     launch no longer leaves a fifth of the chip idle behind the 25-frame tiles of 400 workgroups).
//   * A shorter item (lean_item): 1/dist is the raw v_rsq_f64 (measured 2^-24.2 relative, below the float32 rounding
//     of the stored score); the 1/2000 of triangulation.py:72 is applied once to the score sum; the keypoint
//     gate (:73) is evaluated once per camera into a lane mask; exact intersection / singular pair / NaN are
//     detected on the score sum; the per-pair offsets d = t_s - t_m live in scalar registers (one SGPR operand
//     per v_fma_f64, no LDS read, no VGPR); item -> (frame, joint) uses the compile-time J; loads and stores
//     go through BUFFER instructions whose descriptor covers exactly the wave's tile: the byte offset of item
//     lane + 64 k inside a tile is the same for every tile, so it comes from a small LDS table (no division, no
//     64-bit address arithmetic, no clamp), lanes past the end of the tile read zeros and their stores are
//     dropped by the range check of the hardware.
#pragma once
#include "snowtri_fused.hpp"

namespace snowtri {

// Six cameras and more (15 / 21 / 28 pairs): the unrolled lean_item with its three keypoint buffers, 9 C matrix entries and
// the pair offsets in scalar registers needs far more than 256 VGPRs (round-4 review: k_fused_lean<8> spilled 431 of them,
// the cooperative kernel 355).  Those rigs run the SAME kernels on cluster_item (snowtri_item.hpp: constants read from LDS
// where they are used, pairs in groups of four) with kLeanRolledRing keypoint buffers at kLeanRolledWaves waves per SIMD;
// such an item is 800-1 500 VALU instructions for 72-96 bytes of keypoints: compute-bound, the SIMD's other waves cover a
// fetch.  Per-frame check, mean and fall-back are unchanged.
#ifndef SNOWTRI_LEAN_ROLLED_WAVES
#define SNOWTRI_LEAN_ROLLED_WAVES 2
#endif
#ifndef SNOWTRI_LEAN_ROLLED_RING
#define SNOWTRI_LEAN_ROLLED_RING 2
#endif
#ifndef SNOWTRI_LEAN_ROLLED_GROUP
#define SNOWTRI_LEAN_ROLLED_GROUP 4
#endif
constexpr int kLeanRolledWaves = SNOWTRI_LEAN_ROLLED_WAVES, kLeanRolledRing = SNOWTRI_LEAN_ROLLED_RING;   // (A/B builds override them)
constexpr int kLeanRolledGroup = SNOWTRI_LEAN_ROLLED_GROUP;   // pairs between two scheduling barriers of the item
// float64 outputs (the reference's own output type) exist on the rolled item only -- cluster_item<..., double> refines 1/dist by a
// Newton step, lean_item carries the raw v_rsq_f64 of the float32 contract -- so five cameras roll too when they are asked for.
template <int C, typename TOut = float>
struct LeanShape {
    static constexpr bool kRolled = C >= 6 || (C == 5 && sizeof(TOut) == 8);
    static constexpr int kWaves = kRolled ? kLeanRolledWaves : kFastWaves;
    static constexpr int kRing = kRolled ? kLeanRolledRing : 3;
};

constexpr int kLeanTw = 12;  // frames per wave tile (12 x 133 = 24.94 passes of 64 lanes; 10 would make the single-cluster check one pass of 60 lanes instead of two, measured: the same 348 instructions per item)
constexpr int kLeanWaves = kBlock / 64;  // waves per workgroup
constexpr int kLeanSlowShift = 4;        // slow-frame bit index = (tile ordinal of the workgroup << 4) | frame in tile

__host__ __device__ constexpr int lean_items_pad(int JC) { return (kLeanTw * JC + 63) / 64 * 64; }   // item slots of a wave tile, whole passes
__host__ __device__ constexpr int lean_table_entries(int JC) { return lean_items_pad(JC) + 256; }      // + the prefetch distance past the last pass
__host__ __device__ constexpr int lean_const_doubles(int C) { return (12 * C + 4 * (C * (C - 1) / 2) + 1) & ~1; }  // M[C][9], t[C][3], d[NP][3], pairs[NP][2] (int32)

__host__ __device__ constexpr size_t lean_lds_bytes(int C, int JC, int slow_words, int score_bytes = 4) {
    const size_t stash = (size_t)kLeanWaves * lean_items_pad(JC) * score_bytes;   // fused joint scores as stored (float32 / float64), per wave
    const size_t table = (size_t)4 * lean_table_entries(JC);                 // item -> byte offset of its camera-0 keypoint in the tile
    return (((size_t)8 * lean_const_doubles(C) + table + stash + (size_t)4 * slow_words) + 15) & ~(size_t)15;
}

// Raw buffer descriptor (stride 0, range-checked on the byte offset) over [base, base + bytes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lean_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// cache-policy bits of the buffer instructions (gfx942 / gfx950: 1 = sc0, 2 = nt, 16 = sc1)
constexpr int kLeanStoreAux = 2;   // non-temporal output stores: the 21 MB a 10 000-frame launch writes leave the L2 while the launch
                                   // runs instead of as one write-back at its end (measured: 32.2 -> 31.3 us per launch)
constexpr int kLeanLoadAux = 0;

typedef unsigned lean_u3 __attribute__((ext_vector_type(3)));
typedef unsigned lean_u4 __attribute__((ext_vector_type(4)));
typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));

// one keypoint record at byte offset voff + soff of the buffer (zeros when out of range)
template <typename TIn>
__device__ __forceinline__ Kp3<TIn> lean_load_kp3(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    Kp3<TIn> k;
    if constexpr (sizeof(TIn) == 4) {
        const lean_u3 w = __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, kLeanLoadAux);
        k.u = __uint_as_float(w.x);
        k.v = __uint_as_float(w.y);
        k.s = __uint_as_float(w.z);
    } else {
        const lean_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, kLeanLoadAux);
        const lean_u2 z = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff + 16, (int)soff, kLeanLoadAux);
        k.u = __hiloint2double((int)w.y, (int)w.x);
        k.v = __hiloint2double((int)w.w, (int)w.z);
        k.s = __hiloint2double((int)z.y, (int)z.x);
    }
    return k;
}

__device__ __forceinline__ double uniform_f64(double x) {  // wave-uniform value -> scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double uniform_lane_f64(double x, int lane) {  // the value lane `lane` (wave-uniform) holds
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// x if the lane's bit of `mask` is set, else 0: one v_cndmask_b32 on a scalar lane mask (the compiler turns
// a bool that crosses basic blocks into a VGPR 0/1 and three more VALU instructions)
__device__ __forceinline__ float select_by_mask(float x, unsigned long long mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}

// One (frame, joint): C rays, all C(C,2) pair solves, score-weighted fusion (see pairwise_item for the
// algebra of the fusion regrouped per ray; here the determinants cancel out of it: no reciprocal per pair).  Returns
// true if the item needs the IEEE-exact routine.
template <int C, typename TIn>
__device__ __forceinline__ bool lean_item(const double *__restrict__ Mlds,
                                          const double (&dS)[3 * (C * (C - 1) / 2)],
                                          const Kp3<TIn> (&cur)[C], float kthr_f32, double kthr, double dthr2,
                                          float &ox, float &oy, float &oz, double &os) {
    // Every product-sum below that is meant to be fused is an explicit fma().  Implicit contraction is switched off:
    // the item is inlined three times (the prefetch ring) and the compiler may fuse `beta += x * y` differently in each
    // copy -- a frame's result would then depend (in the last bit) on which ring slot its position in the launch maps to.
#pragma clang fp contract(off)
    constexpr int NPc = C * (C - 1) / 2;
    double Mp[9 * C];
#pragma unroll
    for (int i = 0; i < 9 * C; i++) Mp[i] = Mlds[i];
    __builtin_amdgcn_sched_barrier(0);  // one burst of LDS reads, one wait (+2.5 % measured in round 1)
    const double *tp = Mlds + 9 * C;
    Vec3 h[C];
    double a[C], alpha[C], beta[C];
    unsigned long long okm[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        // A1, camera.py:241-243 with M = R inv(K)
        const double u = (double)cur[c].u, v = (double)cur[c].v;
        h[c].x = fma(Mp[9 * c + 0], u, fma(Mp[9 * c + 1], v, Mp[9 * c + 2]));
        h[c].y = fma(Mp[9 * c + 3], u, fma(Mp[9 * c + 4], v, Mp[9 * c + 5]));
        h[c].z = fma(Mp[9 * c + 6], u, fma(Mp[9 * c + 7], v, Mp[9 * c + 8]));
        a[c] = dot3(h[c], h[c]);
        // triangulation.py:73, once per camera: lanes whose confidence is NOT below the threshold
        if constexpr (sizeof(TIn) == 4)
            okm[c] = __ballot(!((float)cur[c].s < kthr_f32));
        else
            okm[c] = __ballot(!((double)cur[c].s < kthr));
    }
    // A2 + :72-74 per camera pair WITHOUT a reciprocal of the determinant.  For rays t_m + S0 h_m and t_s - S1 h_s:
    //   S0 = N0 / det, S1 = N1 / det,  N0 = a_s e - b g,  N1 = a_m g - b e,  det = a_m a_s - b^2 = |h_m x h_s|^2,
    //   dist = |d . (h_m x h_s)| / |h_m x h_s| = |n| / sqrt(det)        (distance of two skew lines; d = t_s - t_m)
    // so with rho = rsq(n^2 det):  1 / dist = rho det, and the pair's weight 2000 x score = ssum / dist = (ssum rho) det:
    //   sq S0 = (ssum rho) N0,   sq S1 = (ssum rho) N1,   sq = (ssum rho) det
    // -- the determinant cancels out of the fused point's numerators: one VALU instruction per pair less and no shared
    // reciprocal (no chain over the pairs, no range of a product to watch).  Conditioning: n cancels to dist |h_m x h_s|
    // from terms of size |d| |h_m| |h_s|, 1e-16 |d| / dist relative -- the same cancellation as ||Wm - Ws|| from
    // 5 m coordinates in the reference.  The gate dist > dthr (:74) is taken on n^2 > dthr^2 det (both sides x det > 0).
    // A singular pair (a c == b b in separately rounded products: det = 0 exactly) or an exact intersection (n = 0) give
    // rho = inf, a negative (rounding of nearly parallel rays) or NaN determinant rho = NaN: they reach the score sum,
    // whatever the gates select (0 x inf = NaN).
    int q = 0;
#pragma unroll
    for (int mc = 0; mc < C - 1; mc++) {
#pragma unroll
        for (int sc = mc + 1; sc < C; sc++, q++) {
            const Vec3 &hm = h[mc], &hs = h[sc];
            const double dx = dS[3 * q], dy = dS[3 * q + 1], dz = dS[3 * q + 2];
            const double b = dot3(hm, hs);
            const double det = a[mc] * a[sc] - b * b;   // separately rounded products (contraction off): singular as the reference sees it <=> det == 0 exactly (cluster_item)
            const double e = fma(hm.z, dz, fma(hm.y, dy, hm.x * dx));
            const double g = fma(hs.z, dz, fma(hs.y, dy, hs.x * dx));
            const double N0 = fma(a[sc], e, -(b * g));
            const double N1 = fma(a[mc], g, -(b * e));
            // n = h_m . (h_s x d)
            const double cx = fma(hs.y, dz, -(hs.z * dy)), cy = fma(hs.z, dx, -(hs.x * dz)), cz = fma(hs.x, dy, -(hs.y * dx));
            const double n = fma(hm.z, cz, fma(hm.y, cy, hm.x * cx));
            const double n2 = n * n;
            const double rho = __builtin_amdgcn_rsq(n2 * det);
            // :72-74  score = ((sm+ss)/2) / (dist*1000), zeroed by the three gates; w det = 2000 x that score
            const unsigned long long keep = okm[mc] & okm[sc] & __ballot(!(n2 > dthr2 * det));
            double w;
            if constexpr (sizeof(TIn) == 4) {
                w = (double)select_by_mask((float)cur[mc].s + (float)cur[sc].s, keep) * rho;  // float32 sum as NumPy
            } else {
                const double ssum = (double)cur[mc].s + (double)cur[sc].s;
                w = __hiloint2double(__float_as_int(select_by_mask(__int_as_float(__double2hiint(ssum)), keep)),
                                     __float_as_int(select_by_mask(__int_as_float(__double2loint(ssum)), keep))) * rho;
            }
            // alpha_m += sq S0, alpha_s -= sq S1, beta_m, beta_s += sq  with  sq S0 = w N0, sq S1 = w N1, sq = w det
            if (mc == 0) {
                alpha[sc] = -w * N1;
                beta[sc] = w * det;
                if (sc == 1) {
                    alpha[0] = w * N0;
                    beta[0] = beta[1];
                } else {
                    alpha[0] = fma(w, N0, alpha[0]);
                    beta[0] = fma(w, det, beta[0]);
                }
            } else {
                alpha[mc] = fma(w, N0, alpha[mc]);
                alpha[sc] = fma(-w, N1, alpha[sc]);
                beta[mc] = fma(w, det, beta[mc]);
                beta[sc] = fma(w, det, beta[sc]);
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
    // sb = 2 x 2000 x sum_q s_q (:141).  sum == 0 -> the joint stays (0,0,0)/0 (:142-143): sx = sy = sz = 0 then,
    // so any finite reciprocal will do: 1 / max(sb, 1e-300) saves the compare-and-select.
    const double r = rcp_nr1(fmax(sb, 1e-300));
    ox = (float)(sx * r);  // :144-147 as (sum s (Wm+Ws)) / (2 sum s)
    oy = (float)(sy * r);
    oz = (float)(sz * r);
    os = sb * (0.00025 / (double)NPc);  // :148
    // dist == 0 or a singular pair (rho = inf), a negative determinant or NaN input (rho = NaN) leave sum s inf or NaN:
    // the IEEE-exact routine decides those frames
    return !(sb < 1e300);
}

// The item of a rig: lean_item up to five cameras, cluster_item (float32 outputs: raw v_rsq_f64, the same contract) beyond.
template <int C, typename TIn, int ND>
__device__ __forceinline__ bool lean_solve(const double *__restrict__ Mlds, const double (&dS)[ND], const Kp3<TIn> (&cur)[C], float kthr_f32,
                                           double kthr, double dthr2, float &ox, float &oy, float &oz, double &os) {
    if constexpr (LeanShape<C>::kRolled) {
        double x, y, z;
        asm volatile("" ::: "memory");   // the rig constants are re-read from LDS by every item (hoisted out of the item loop they take hundreds of registers)
        const bool bad = cluster_item<C, TIn, float, kLeanRolledGroup>(Mlds, cur, kthr_f32, kthr, dthr2, x, y, z, os);
        ox = (float)x;
        oy = (float)y;
        oz = (float)z;
        return bad;
    } else {
        return lean_item<C>(Mlds, dS, cur, kthr_f32, kthr, dthr2, ox, oy, oz, os);
    }
}

// One item solved, its joint record stored (non-temporal, range-checked by the descriptor) and its score stashed for the frame's
// mean: float32 records through lean_solve, float64 records (two 16-byte stores) through cluster_item's Newton-refined branch.
template <int C, typename TIn, typename TOut, int ND>
__device__ __forceinline__ bool lean_solve_store(const double *__restrict__ Mlds, const double (&dS)[ND], const Kp3<TIn> (&cur)[C], float kthr_f32,
                                                 double kthr, double dthr2, __amdgpu_buffer_rsrc_t rout, unsigned out_off, TOut *stash_slot) {
    if constexpr (sizeof(TOut) == 4) {
        float ox, oy, oz;
        double os;
        const bool bad = lean_solve<C>(Mlds, dS, cur, kthr_f32, kthr, dthr2, ox, oy, oz, os);
        const float osf = (float)os;
        lean_u4 rec;
        rec.x = __float_as_uint(ox);
        rec.y = __float_as_uint(oy);
        rec.z = __float_as_uint(oz);
        rec.w = __float_as_uint(osf);
        __builtin_amdgcn_raw_buffer_store_b128(rec, rout, (int)out_off, 0, kLeanStoreAux);
        *stash_slot = osf;
        return bad;
    } else {
        double x, y, z, os;
        asm volatile("" ::: "memory");   // (as lean_solve: the rig constants are re-read from LDS by every item)
        const bool bad = cluster_item<C, TIn, double, kLeanRolledGroup>(Mlds, cur, kthr_f32, kthr, dthr2, x, y, z, os);
        lean_u4 lo, hi;
        lo.x = (unsigned)__double2loint(x);
        lo.y = (unsigned)__double2hiint(x);
        lo.z = (unsigned)__double2loint(y);
        lo.w = (unsigned)__double2hiint(y);
        hi.x = (unsigned)__double2loint(z);
        hi.y = (unsigned)__double2hiint(z);
        hi.z = (unsigned)__double2loint(os);
        hi.w = (unsigned)__double2hiint(os);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rout, (int)out_off, 0, kLeanStoreAux);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rout, (int)out_off + 16, 0, kLeanStoreAux);
        *stash_slot = os;
        return bad;
    }
}

// ---- per-frame steps shared by k_fused_lean and k_fused_lean_coop (ONE definition: both kernels give the same bits) ----
// Single-cluster check (:116-130) for lane = (frame wl of the pass, pair qq), lane = wl NP + qq: the lane solves its pair
// at the centre joint (midpoint only) and takes candidate 0's point from the frame's first lane of the same pass.
// Returns true if candidate qq lies farther than condense_distance_tol from candidate 0 (or cannot be decided here).
template <int C, typename TIn>
__device__ __forceinline__ bool lean_centre_far(const double *__restrict__ Mlds, int mc, int sc, int qq, int lane, const Kp3<TIn> &km,
                                                const Kp3<TIn> &ks, double ctol2_lo) {
    constexpr int NP = C * (C - 1) / 2;
    (void)NP;
    const double *tm = Mlds + 9 * C + 3 * mc, *ts = Mlds + 9 * C + 3 * sc, *dq = Mlds + 12 * C + 3 * qq;
    // only the midpoint is needed here: A2 without the distance (sw = Wm + Ws = 2 W)
    const RayRec a = make_ray(Mlds + 9 * mc, km.u, km.v), b = make_ray(Mlds + 9 * sc, ks.u, ks.v);
    const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
    const double e = fma(a.z, dq[2], fma(a.y, dq[1], a.x * dq[0]));
    const double g = fma(b.z, dq[2], fma(b.y, dq[1], b.x * dq[0]));
    const double inv = rcp_nr2(fma(a.a, b.a, -(bq * bq)));
    const double S0 = fma(b.a, e, -(bq * g)) * inv, S1 = fma(a.a, g, -(bq * e)) * inv;
    const double swx = fma(-b.x, S1, fma(a.x, S0, tm[0] + ts[0])), swy = fma(-b.y, S1, fma(a.y, S0, tm[1] + ts[1])),
                 swz = fma(-b.z, S1, fma(a.z, S0, tm[2] + ts[2]));
    const int src = lane - qq;  // the frame's pair 0, same pass
    const double w0x = __shfl(swx, src, 64), w0y = __shfl(swy, src, 64), w0z = __shfl(swz, src, 64);
    const double ex = 0.5 * (w0x - swx), ey = 0.5 * (w0y - swy), ez = 0.5 * (w0z - swz);
    // :124-125 `norm > tol` on the squares, with a 1e-12 guard band: whatever comes near the tolerance (or is
    // NaN) is left to the exact routine, which takes the square root as the reference does
    const double c2 = fma(ez, ez, fma(ey, ey, ex * ex));
    return qq > 0 && !(c2 < ctol2_lo);
}

// Sum of a frame's JC fused joint scores (float32 as stored) by four lanes (sub = 0..3 takes joints sub, sub + 4, ...),
// in double: the partial of lane `sub`; the caller adds the four partials with two xor-shuffles.
template <int JC, typename TS>
__device__ __forceinline__ double lean_row_partial(const TS *__restrict__ row, int sub) {
    constexpr int G = 4;
    double sum = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = sub;
#pragma unroll 2
    for (; b + 3 * G < JC; b += 4 * G) {
        const double v0 = (double)row[b], v1 = (double)row[b + G], v2 = (double)row[b + 2 * G], v3 = (double)row[b + 3 * G];
        sum += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    for (; b < JC; b += G) sum += (double)row[b];
    return (sum + s1) + (s2 + s3);
}

// frames [f0, f0 + nf) of tile `t` when F frames are cut into ntiles tiles of base or base + 1 frames
__device__ __forceinline__ void lean_tile_range(int64_t t, int base, int64_t rem, int64_t &f0, int &nf) {
    f0 = t * base + (t < rem ? t : rem);
    nf = base + (t < rem ? 1 : 0);
}

// Grid: any number of workgroups; wave gw = 4 blockIdx.x + wave takes tiles gw, gw + 4 gridDim.x, ...
// Dynamic LDS: lean_lds_bytes(C, JC, slow_words); slow_words >= ceil(tiles of one WORKGROUP * 16 / 32).
template <int C, typename TIn, int JC, typename TOut = float>
__global__ __launch_bounds__(kBlock, (LeanShape<C, TOut>::kWaves)) void k_fused_lean(
    int64_t F, int64_t ntiles, int tile_base, int64_t tile_rem, int slow_words, Rig rig, const TIn *__restrict__ kpts,
    const int32_t *__restrict__ n_persons, Params prm, TOut *__restrict__ out4, TOut *__restrict__ out_ps,
    int32_t *__restrict__ out_count, uint32_t *__restrict__ out_flags, char *scratch, size_t scratch_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = C * (C - 1) / 2;
    constexpr int kItemsPad = lean_items_pad(JC);
    constexpr int kTable = lean_table_entries(JC);
    constexpr int kConstDoubles = lean_const_doubles(C);
    constexpr unsigned kRec = (unsigned)sizeof(Kp3<TIn>);       // bytes of one keypoint record
    constexpr unsigned kCamStride = (unsigned)JC * kRec;        // camera c of a frame is c * kCamStride past camera 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile bookkeeping and loop control stay on the SALU
    // LDS: [rig constants | item -> input offset table | per-wave stash of fused joint scores | slow-frame bit words];
    // the constants sit at offset 0 so that every ds_read of them is base + immediate
    double *Mlds = reinterpret_cast<double *>(smem);
    uint32_t *table = reinterpret_cast<uint32_t *>(Mlds + kConstDoubles);
    TOut *stash = reinterpret_cast<TOut *>(table + kTable) + wave * kItemsPad;
    uint32_t *slowbits = reinterpret_cast<uint32_t *>(reinterpret_cast<TOut *>(table + kTable) + kLeanWaves * kItemsPad);
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const int64_t wstride = (int64_t)gridDim.x * kLeanWaves;
    constexpr bool kRolled = LeanShape<C, TOut>::kRolled;
    constexpr unsigned kOutRec = 4u * (unsigned)sizeof(TOut);   // bytes of one joint record [x, y, z, score]
    static_assert(sizeof(TOut) == 4 || kRolled, "float64 outputs: the rolled item only");
    constexpr int kRing = (kRolled && sizeof(TIn) == 8 && C >= 8) ? 1 : LeanShape<C, TOut>::kRing;   // (float64 keypoints: buffers of twice the size)
    Kp3<TIn> bufA[C], bufB[kRing >= 2 ? C : 1], bufC[kRing >= 3 ? C : 1];

    // Item i = lane + 64 k of a tile is joint i % JC of the tile's frame i / JC; its camera-c record sits at byte
    // (i + (i / JC) (C-1) JC + c JC) kRec of the tile.  The part that does not depend on c is the same for every tile:
    // table[i].  Cameras come in pairs: an even camera goes into the scalar offset, the odd one adds an immediate.
    auto fetch = [&](Kp3<TIn>(&dst)[C], __amdgpu_buffer_rsrc_t rin, unsigned voff) {
#pragma unroll
        for (int c = 0; c < C; c++)
            dst[c] = lean_load_kp3<TIn>(rin, voff + (unsigned)(c & 1) * kCamStride, (unsigned)(c & ~1) * kCamStride);
    };
    // offsets of the lane's first two items, the same in every tile
    const unsigned i0 = (unsigned)lane, i1 = (unsigned)lane + 64u;
    const unsigned voff0 = (i0 + (i0 / (unsigned)JC) * (unsigned)((C - 1) * JC)) * kRec;
    const unsigned voff1 = (i1 + (i1 / (unsigned)JC) * (unsigned)((C - 1) * JC)) * kRec;

    // the rig constants are requested BEFORE the first keypoints (see k_fused_lean_coop: the vector-memory counter returns
    // in order, a constant requested behind the keypoints would wait for their trip to HBM)
    const double cM = rig.M[tid < 9 * C ? tid : 0], cT = rig.t[tid < 3 * C ? tid : 0];
    const double cD = rig.pairc[tid < 3 * NP ? 6 * (tid / 3) + tid % 3 : 0];
    const int32_t cP = rig.pairs[tid < 2 * NP ? tid : 0];
    double dS[kRolled ? 1 : 3 * NP];  // per-pair d = t_s - t_m, wave-uniform -> scalar registers (cluster_item reads them from LDS)
    if constexpr (!kRolled) {
#pragma unroll
        for (int i = 0; i < 3 * NP; i++) dS[i] = rig.pairc[6 * (i / 3) + i % 3];
    }
    int64_t tile = (int64_t)blockIdx.x * kLeanWaves + wave;
    int64_t f0 = 0;
    int nf = 0;
    if (tile < ntiles) {
        lean_tile_range(tile, tile_base, tile_rem, f0, nf);
        const __amdgpu_buffer_rsrc_t rin = lean_rsrc(kp3 + f0 * (int64_t)(C * JC), (unsigned)(nf * C * JC) * kRec);
        fetch(bufA, rin, voff0);
        if constexpr (kRing >= 3) fetch(bufB, rin, voff1);
    }
    // (the first keypoints are in flight while the constants are set up)
    for (unsigned i = (unsigned)tid; i < (unsigned)kTable; i += kBlock)
        table[i] = (i + (i / (unsigned)JC) * (unsigned)((C - 1) * JC)) * kRec;
    for (int i = tid; i < slow_words; i += kBlock) slowbits[i] = 0u;
    if (tid < 9 * C) Mlds[tid] = cM;
    if (tid < 3 * C) Mlds[9 * C + tid] = cT;
    if (tid < 3 * NP) Mlds[12 * C + tid] = cD;
    int32_t *pairs_lds = reinterpret_cast<int32_t *>(Mlds + 12 * C + 3 * NP);
    if (tid < 2 * NP) pairs_lds[tid] = cP;
    if constexpr (!kRolled) {
#pragma unroll
        for (int i = 0; i < 3 * NP; i++) dS[i] = uniform_f64(dS[i]);
    }
    const float kthr_f32 = prm.kthr_f32;
    const double kthr = prm.kthr, dthr2 = prm.dthr2;
    const double ctol2_lo = prm.ctol < 0.0 ? -1.0 : prm.ctol * prm.ctol * (1.0 - 1e-12);   // single-cluster check, see there
    __syncthreads();  // constants, table and the cleared slow-frame bits are visible to every wave

    for (int ord = 0; tile < ntiles; tile += wstride, ord++) {
        SNOWTRI_DEV_CHECK(f0 >= 0 && nf >= 1 && nf <= kLeanTw && f0 + nf <= F, 1);                 // the tile lies inside the batch
        SNOWTRI_DEV_CHECK((ord * kLeanWaves + wave + 1) << kLeanSlowShift <= slow_words * 32, 2);  // its slow-frame bits exist
        const __amdgpu_buffer_rsrc_t rin = lean_rsrc(kp3 + f0 * (int64_t)(C * JC), (unsigned)(nf * C * JC) * kRec);
        const __amdgpu_buffer_rsrc_t rout =
            lean_rsrc(reinterpret_cast<char *>(out4) + f0 * (int64_t)JC * kOutRec, (unsigned)(nf * JC) * kOutRec);
        const unsigned last = (unsigned)(nf * JC - 1);
        const int npass = (nf * JC + 63) >> 6;
        const unsigned slow_base = (unsigned)((ord * kLeanWaves + wave) << kLeanSlowShift);
        // centre-joint keypoints for the single-cluster check after the item loop (lane = frame x pair)
        constexpr int kCheckFrames = 64 / NP;
        // (not for the rolled item: its registers are spoken for, and an item of 15+ pair solves hides the check's loads anyway)
        constexpr int kCheckPre = kRolled ? 0 : 2;
        Kp3<TIn> ckm[kCheckPre ? kCheckPre : 1], cks[kCheckPre ? kCheckPre : 1];
#pragma unroll
        for (int pass = 0; pass < kCheckPre; pass++) {
            const int wl = lane / NP, qq = lane - wl * NP, w = pass * kCheckFrames + wl;
            const bool live = wl < kCheckFrames && w < nf;
            const Kp3<TIn> *p = kp3 + (f0 + (live ? w : 0)) * (int64_t)(C * JC) + prm.center;
            ckm[pass] = p[pairs_lds[2 * qq] * JC];
            cks[pass] = p[pairs_lds[2 * qq + 1] * JC];
        }

        // item `is` + 64 k of the lane: output record at byte 16 (is + 64 k), stash slot is + 64 k.  Lanes past the
        // tile's last item work on zeros; their store is out of the descriptor's range and their stash slot is padding.
        auto solve_store = [&](const Kp3<TIn>(&buf)[C], unsigned out_off, TOut *stash_slot) {
            const bool bad = lean_solve_store<C, TIn, TOut>(Mlds, dS, buf, kthr_f32, kthr, dthr2, rout, out_off, stash_slot);
            if (__ballot(bad)) {  // rare, wave-uniform branch
                unsigned o = out_off;
                asm volatile("" : "+v"(o));  // (keeps the bit arithmetic below inside the branch)
                const unsigned i = o / kOutRec;
                const unsigned bit = slow_base + i / (unsigned)JC;
                if (bad && i <= last) atomicOr(&slowbits[bit >> 5], 1u << (bit & 31u));
            }
        };
        // ---- item loop: a ring of three register buffers keeps the keypoints of the next two items in flight
        unsigned out_off = (unsigned)lane * kOutRec;       // record of the item being solved; the one being fetched is two passes ahead
        TOut *sp = stash + lane;                           // its stash slot
        const uint32_t *tp = table + lane;                 // its table entry
        if constexpr (kRing >= 3) {
            unsigned t0 = tp[128], t1 = tp[192], t2 = tp[256];  // (read one iteration ahead of their use)
            for (int k = 0; k < npass; k += 3) {
                fetch(bufC, rin, t0);
                solve_store(bufA, out_off, sp);
                fetch(bufA, rin, t1);
                if (k + 1 < npass) solve_store(bufB, out_off + 64u * kOutRec, sp + 64);
                fetch(bufB, rin, t2);
                if (k + 2 < npass) solve_store(bufC, out_off + 128u * kOutRec, sp + 128);
                tp += 192;
                t0 = tp[128];
                t1 = tp[192];
                t2 = tp[256];
                out_off += 192u * kOutRec;
                sp += 192;
            }
        } else if constexpr (kRing == 2) {   // the next item's keypoints fly under the current item
            for (int k = 0; k < npass; k += 2) {
                fetch(bufB, rin, tp[64]);
                solve_store(bufA, out_off, sp);
                fetch(bufA, rin, tp[128]);
                if (k + 1 < npass) solve_store(bufB, out_off + 64u * kOutRec, sp + 64);
                tp += 128;
                out_off += 128u * kOutRec;
                sp += 128;
            }
        } else {                             // one buffer: the SIMD's other waves cover the fetch
            for (int k = 0; k < npass; k++) {
                solve_store(bufA, out_off, sp);
                fetch(bufA, rin, tp[64]);
                tp += 64;
                out_off += 64u * kOutRec;
                sp += 64;
            }
        }
        // this wave's next tile: its first two fetches fly during the epilogue
        const int64_t f0_cur = f0;
        const int nf_cur = nf;
        const int64_t nt = tile + wstride;
        if (nt < ntiles) {
            lean_tile_range(nt, tile_base, tile_rem, f0, nf);
            const __amdgpu_buffer_rsrc_t rnext = lean_rsrc(kp3 + f0 * (int64_t)(C * JC), (unsigned)(nf * C * JC) * kRec);
            fetch(bufA, rnext, voff0);
            if constexpr (kRing >= 3) fetch(bufB, rnext, voff1);
        }

        // ---- single-cluster check (:116-130): every candidate's centre joint within condense_distance_tol of
        //      candidate 0's.  One lane per (frame, pair), whole frames per 64-lane pass: the lane solves its pair at
        //      the centre joint and takes candidate 0's point from the frame's first lane.  The keypoints of the
        //      first two passes were fetched before the item loop; rig constants come from LDS.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int pass = 0; pass * kCheckFrames < nf_cur; pass++) {
            const int wl = lane / NP, qq = lane - wl * NP, w = pass * kCheckFrames + wl;
            const bool live = wl < kCheckFrames && w < nf_cur;
            const int mc = pairs_lds[2 * qq], sc = pairs_lds[2 * qq + 1];
            Kp3<TIn> km, ks;
            if (pass < kCheckPre) {
                km = ckm[pass];
                ks = cks[pass];
            } else {
                const Kp3<TIn> *p = kp3 + (f0_cur + (live ? w : 0)) * (int64_t)(C * JC) + prm.center;
                km = p[mc * JC];
                ks = p[sc * JC];
            }
            const bool far = lean_centre_far<C, TIn>(Mlds, mc, sc, qq, lane, km, ks, ctol2_lo);
            if (live && far) {
                const unsigned bit = slow_base + (unsigned)w;
                atomicOr(&slowbits[bit >> 5], 1u << (bit & 31u));
            }
        }
        // ---- per-frame mean fused score (:150), filters, count: four lanes per frame
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        {
            constexpr int G = 4;
            const int w = lane / G, sub = lane & (G - 1);
            const bool live = w < nf_cur;
            const int64_t f = f0_cur + (live ? w : 0);
            double sum = 0.0;
            if (live) sum = lean_row_partial<JC, TOut>(stash + w * JC, sub);
            int not_one = 0;
            if (n_persons && live)
                for (int c = sub; c < C; c += G) not_one |= n_persons[f * C + c] != 1;
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) {
                sum += __shfl_xor(sum, off, 64);
                not_one |= __shfl_xor(not_one, off, 64);
            }
            if (live && sub == 0) {
                const double avg = sum / (double)JC;
                const unsigned bit = slow_base + (unsigned)w;
                // :151-152; the mean is taken over the joint scores as stored (float32): a mean within 1e-6 of the
                // tolerance is decided by the exact routine on the float64 scores
                const bool slow = ((slowbits[bit >> 5] >> (bit & 31u)) & 1u) != 0u || !(avg >= prm.score_tol) ||
                                  fabs(avg - prm.score_tol) < 1e-6 * fabs(prm.score_tol) || not_one != 0;
                if (slow) {
                    atomicOr(&slowbits[bit >> 5], 1u << (bit & 31u));
                } else {
                    out_count[f] = 1;
                    if (out_ps) out_ps[f] = (TOut)avg;
                    if (out_flags) out_flags[f] = kFlagFast;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next tile's stash writes stay behind these reads
    }

    // ---- rare: frames the speculation could not resolve -> the reference's full algorithm, by the whole workgroup
    __syncthreads();
    unsigned any = 0u;
    for (int i = lane; i < slow_words; i += 64) any |= slowbits[i];
    if (__ballot(any != 0u) == 0ull) return;  // every wave reads the same words: uniform exit
    {
        const PackedWriter<TOut> wr{out4, out_ps};
        double *slab = reinterpret_cast<double *>(scratch + (size_t)blockIdx.x * scratch_per_block);
        for (int wd = 0; wd < slow_words; wd++) {
            uint32_t m = slowbits[wd];
            __syncthreads();  // general_frame reuses the front of the LDS, not the bit words, but keep passes apart
            while (m) {
                const int bpos = __ffs((int)m) - 1;
                m &= m - 1u;
                const unsigned bit = (unsigned)(wd * 32 + bpos);
                const unsigned slot = bit >> kLeanSlowShift;              // = ord * kLeanWaves + wave
                const int64_t t = (int64_t)blockIdx.x * kLeanWaves + (slot % kLeanWaves) + (int64_t)(slot / kLeanWaves) * wstride;
                int64_t tf0;
                int tnf;
                lean_tile_range(t, tile_base, tile_rem, tf0, tnf);
                general_frame<TIn>(tf0 + (bit & ((1u << kLeanSlowShift) - 1u)), 1, JC, NP, rig, kpts, n_persons, prm, 1, wr,
                                   out_count, out_flags, slab, smem);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_fused_lean_coop
// The same path for SMALL launches (one tile per wave in k_fused_lean: up to kLeanTw frames x the resident waves; the
// bench's 10 000-frame step).  There the whole-frame tiles of k_fused_lean cost: a wave of 5 frames runs 11 passes for 10.4
// passes of items, both waves of a SIMD run the per-tile epilogue (a pass-equivalent of instructions each), i.e.
// 2 x (11 + 1.1) pass-equivalents per SIMD where the items alone are 20.3.  Here a WORKGROUP owns a tile of frames
// (10 000 frames over 512 workgroups: 19-20 frames = 40-42 passes) and
//   * its four waves split the tile's PASSES evenly (10 or 11 each; the workgroups alternate which waves take the extra
//     pass, neighbouring workgroups rotate them by two; rotating by dispatch round instead, b / num_cus, measured the same): frames straddle waves, the fused joint
//     scores meet in a workgroup-wide LDS stash;
//   * after ONE barrier the epilogue of the whole tile is dealt to the waves by passes: the single-cluster check 10
//     frames per pass, the mean scores 16 frames per pass, each pass on another wave -- a fifth of the epilogue
//     instructions per SIMD; a second barrier, then one wave writes count / person score / flags of the frames that
//     passed, and the frames that did not are re-done by the workgroup as in k_fused_lean.
// Items, check and mean are the SAME functions as in k_fused_lean: a frame's bits do not depend on which kernel ran it.
// Grid = number of tiles (tile t = frames lean_tile_range(t, tile_base, tile_rem), nf <= nf_max <= kCoopMaxFrames).
// Dynamic LDS: lean_coop_lds_bytes(C, JC, nf_max).
constexpr int kCoopMaxFrames = 32;   // frames per workgroup tile (one bit word of slow frames)
__host__ __device__ constexpr int lean_coop_items_pad(int JC, int nf_max) { return (nf_max * JC + 63) / 64 * 64; }
__host__ __device__ constexpr size_t lean_coop_lds_bytes(int C, int JC, int nf_max, int score_bytes = 4) {
    // [rig constants | item -> input offset table (+ 256 entries of prefetch distance) | stash | mean per frame | slow bits]
    return (((size_t)8 * lean_const_doubles(C) + (size_t)4 * (lean_coop_items_pad(JC, nf_max) + 256) + (size_t)score_bytes * lean_coop_items_pad(JC, nf_max) +
             (size_t)8 * kCoopMaxFrames + 16) + 15) & ~(size_t)15;
}

template <int C, typename TIn, int JC, typename TOut = float>
__global__ __launch_bounds__(kBlock, (LeanShape<C, TOut>::kWaves)) void k_fused_lean_coop(
    int64_t F, int tile_base, int64_t tile_rem, int nf_max, Rig rig, const TIn *__restrict__ kpts, const int32_t *__restrict__ n_persons,
    Params prm, TOut *__restrict__ out4, TOut *__restrict__ out_ps, int32_t *__restrict__ out_count, uint32_t *__restrict__ out_flags,
    char *scratch, size_t scratch_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = C * (C - 1) / 2;
    constexpr int kConstDoubles = lean_const_doubles(C);
    constexpr unsigned kRec = (unsigned)sizeof(Kp3<TIn>);
    constexpr unsigned kCamStride = (unsigned)JC * kRec;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int items_pad = lean_coop_items_pad(JC, nf_max), ntable = items_pad + 256;
    double *Mlds = reinterpret_cast<double *>(smem);
    uint32_t *table = reinterpret_cast<uint32_t *>(Mlds + kConstDoubles);
    TOut *stash = reinterpret_cast<TOut *>(table + ntable);
    double *favg = reinterpret_cast<double *>(stash + items_pad);          // [kCoopMaxFrames] mean fused score of a frame
    uint32_t *slowbits = reinterpret_cast<uint32_t *>(favg + kCoopMaxFrames);   // one word: bit = frame of the tile
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    constexpr bool kRolled = LeanShape<C, TOut>::kRolled;
    constexpr unsigned kOutRec = 4u * (unsigned)sizeof(TOut);   // bytes of one joint record [x, y, z, score]
    static_assert(sizeof(TOut) == 4 || kRolled, "float64 outputs: the rolled item only");
    constexpr int kRing = (kRolled && sizeof(TIn) == 8 && C >= 8) ? 1 : LeanShape<C, TOut>::kRing;   // (float64 keypoints: buffers of twice the size)
    Kp3<TIn> bufA[C], bufB[kRing >= 2 ? C : 1], bufC[kRing >= 3 ? C : 1];

#ifdef SNOWTRI_LEAN_TRACE   // dev build: wall-clock stamps (100 MHz) of every wave at the phase boundaries, in the workgroup's scratch slab
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(scratch + (size_t)blockIdx.x * scratch_per_block) + 8 * (threadIdx.x >> 6);
#define SNOWTRI_STAMP(i) do { if ((threadIdx.x & 63) == 0) trace[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SNOWTRI_STAMP(i) ((void)0)
#endif
    SNOWTRI_STAMP(0);
#ifdef SNOWTRI_LEAN_TRACE
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    int64_t f0;
    int nf;
    lean_tile_range((int64_t)blockIdx.x, tile_base, tile_rem, f0, nf);
    SNOWTRI_DEV_CHECK(f0 >= 0 && nf >= 1 && nf <= nf_max && nf_max <= kCoopMaxFrames && f0 + nf <= F, 3);
    // passes of the tile, dealt to the waves: position `pos` in the rotated wave order takes passes [p0, p0 + np)
    const int npass_tile = (nf * JC + 63) >> 6;
    const int pos = (wave + 2 * (int)(blockIdx.x & 1u)) & (kLeanWaves - 1);
    const int pq = npass_tile / kLeanWaves, pr = npass_tile - pq * kLeanWaves;
    const int p0 = pos * pq + (pos < pr ? pos : pr), npass = pq + (pos < pr ? 1 : 0);
    const unsigned i0 = (unsigned)p0 * 64u;   // first item of the wave
    const __amdgpu_buffer_rsrc_t rin = lean_rsrc(kp3 + f0 * (int64_t)(C * JC), (unsigned)(nf * C * JC) * kRec);
    const __amdgpu_buffer_rsrc_t rout = lean_rsrc(reinterpret_cast<char *>(out4) + f0 * (int64_t)JC * kOutRec, (unsigned)(nf * JC) * kOutRec);
    const unsigned last = (unsigned)(nf * JC - 1);
    auto fetch = [&](Kp3<TIn>(&dst)[C], unsigned voff) {
#pragma unroll
        for (int c = 0; c < C; c++)
            dst[c] = lean_load_kp3<TIn>(rin, voff + (unsigned)(c & 1) * kCamStride, (unsigned)(c & ~1) * kCamStride);
    };
    auto item_offset = [&](unsigned i) { return (i + (i / (unsigned)JC) * (unsigned)((C - 1) * JC)) * kRec; };
    // The rig constants are requested BEFORE the first keypoints: the vector-memory counter returns in order, so waiting for
    // a constant that was requested behind the keypoints would wait for the keypoints' trip to HBM (cold TLB: ~3 us) too
    // -- measured with wall-clock stamps per wave (-DSNOWTRI_LEAN_TRACE): 4.5 us from entry to the first item that way.
    // Unconditional loads of clamped indices: a load inside a branch is waited for where the branch joins.
    const double cM = rig.M[tid < 9 * C ? tid : 0], cT = rig.t[tid < 3 * C ? tid : 0];
    const double cD = rig.pairc[tid < 3 * NP ? 6 * (tid / 3) + tid % 3 : 0];
    const int32_t cP = rig.pairs[tid < 2 * NP ? tid : 0];
    double dS[kRolled ? 1 : 3 * NP];
    if constexpr (!kRolled) {
#pragma unroll
        for (int i = 0; i < 3 * NP; i++) dS[i] = rig.pairc[6 * (i / 3) + i % 3];
    }
    // the wave's first two items (offsets computed: the table is not there yet); a pass the wave does not own reads nothing
    constexpr unsigned kNoItem = 0x40000000u;   // beyond every descriptor (and no wrap-around with the camera offsets): the load returns zeros without touching memory
    fetch(bufA, npass > 0 ? item_offset(i0 + (unsigned)lane) : kNoItem);
    if constexpr (kRing >= 3) fetch(bufB, npass > 1 ? item_offset(i0 + 64u + (unsigned)lane) : kNoItem);
    // centre-joint keypoints of the check pass this wave will run after the barrier (pass `pos`: frames 10 pos ...)
    constexpr int kCheckFrames = 64 / NP;
    const int ncheck = (nf + kCheckFrames - 1) / kCheckFrames;
    Kp3<TIn> ckm, cks;
    int32_t *pairs_lds = reinterpret_cast<int32_t *>(Mlds + 12 * C + 3 * NP);
    // (the first keypoints are in flight while the constants are set up)
    for (unsigned i = (unsigned)tid; i < (unsigned)ntable; i += kBlock) table[i] = item_offset(i);
    if (tid == 0) slowbits[0] = 0u;
    if (tid < 9 * C) Mlds[tid] = cM;
    if (tid < 3 * C) Mlds[9 * C + tid] = cT;
    if (tid < 3 * NP) Mlds[12 * C + tid] = cD;
    if (tid < 2 * NP) pairs_lds[tid] = cP;
    if constexpr (!kRolled) {
#pragma unroll
        for (int i = 0; i < 3 * NP; i++) dS[i] = uniform_f64(dS[i]);
    }
    const float kthr_f32 = prm.kthr_f32;
    const double kthr = prm.kthr, dthr2 = prm.dthr2;
    const double ctol2_lo = prm.ctol < 0.0 ? -1.0 : prm.ctol * prm.ctol * (1.0 - 1e-12);
    SNOWTRI_STAMP(1);
    __syncthreads();
    SNOWTRI_STAMP(2);
    {
        const int wl = lane / NP, qq = lane - wl * NP, w = pos * kCheckFrames + wl;
        const bool live = pos < ncheck && wl < kCheckFrames && w < nf;
        const Kp3<TIn> *p = kp3 + (f0 + (live ? w : 0)) * (int64_t)(C * JC) + prm.center;
        ckm = p[pairs_lds[2 * qq] * JC];
        cks = p[pairs_lds[2 * qq + 1] * JC];
    }

    auto solve_store = [&](const Kp3<TIn>(&buf)[C], unsigned out_off, TOut *stash_slot) {
        const bool bad = lean_solve_store<C, TIn, TOut>(Mlds, dS, buf, kthr_f32, kthr, dthr2, rout, out_off, stash_slot);
        if (__ballot(bad)) {  // rare, wave-uniform branch
            unsigned o = out_off;
            asm volatile("" : "+v"(o));
            const unsigned i = o / kOutRec;
            if (bad && i <= last) atomicOr(&slowbits[0], 1u << (i / (unsigned)JC));
        }
    };
    // ---- item loop over the wave's passes: the ring of three register buffers as in k_fused_lean; the fetch two passes
    //      ahead stops at the wave's last pass (the passes behind it belong to the next wave)
    {
        unsigned out_off = (i0 + (unsigned)lane) * kOutRec;
        TOut *sp = stash + i0 + lane;
        const uint32_t *tp = table + i0 + lane;
        if constexpr (kRing >= 3) {
            for (int k = 0; k < npass; k += 3) {
                fetch(bufC, k + 2 < npass ? tp[128] : kNoItem);
                solve_store(bufA, out_off, sp);
                fetch(bufA, k + 3 < npass ? tp[192] : kNoItem);
                if (k + 1 < npass) solve_store(bufB, out_off + 64u * kOutRec, sp + 64);
                fetch(bufB, k + 4 < npass ? tp[256] : kNoItem);
                if (k + 2 < npass) solve_store(bufC, out_off + 128u * kOutRec, sp + 128);
                tp += 192;
                out_off += 192u * kOutRec;
                sp += 192;
            }
        } else if constexpr (kRing == 2) {
            for (int k = 0; k < npass; k += 2) {
                fetch(bufB, k + 1 < npass ? tp[64] : kNoItem);
                solve_store(bufA, out_off, sp);
                fetch(bufA, k + 2 < npass ? tp[128] : kNoItem);
                if (k + 1 < npass) solve_store(bufB, out_off + 64u * kOutRec, sp + 64);
                tp += 128;
                out_off += 128u * kOutRec;
                sp += 128;
            }
        } else {
            for (int k = 0; k < npass; k++) {
                solve_store(bufA, out_off, sp);
                fetch(bufA, k + 1 < npass ? tp[64] : kNoItem);
                tp += 64;
                out_off += 64u * kOutRec;
                sp += 64;
            }
        }
    }
    SNOWTRI_STAMP(3);
    __syncthreads();   // the tile's fused joint scores are in the stash, the `bad` bits in slowbits
    SNOWTRI_STAMP(4);

    // ---- epilogue of the tile, dealt to the waves by passes: check pass c on the wave at position c (mod 4), mean pass m
    //      on the wave at position ncheck + m (mod 4)
    for (int pass = pos; pass < ncheck; pass += kLeanWaves) {
        const int wl = lane / NP, qq = lane - wl * NP, w = pass * kCheckFrames + wl;
        const bool live = wl < kCheckFrames && w < nf;
        const int mc = pairs_lds[2 * qq], sc = pairs_lds[2 * qq + 1];
        Kp3<TIn> km = ckm, ks = cks;
        if (pass != pos) {
            const Kp3<TIn> *p = kp3 + (f0 + (live ? w : 0)) * (int64_t)(C * JC) + prm.center;
            km = p[mc * JC];
            ks = p[sc * JC];
        }
        const bool far = lean_centre_far<C, TIn>(Mlds, mc, sc, qq, lane, km, ks, ctol2_lo);
        if (live && far) atomicOr(&slowbits[0], 1u << w);
    }
    {
        constexpr int G = 4;
        const int nmean = (nf + 64 / G - 1) / (64 / G);
        for (int pass = (pos - ncheck) & (kLeanWaves - 1); pass < nmean; pass += kLeanWaves) {
            const int w = pass * (64 / G) + lane / G, sub = lane & (G - 1);
            const bool live = w < nf;
            const int64_t f = f0 + (live ? w : 0);
            double sum = 0.0;
            if (live) sum = lean_row_partial<JC, TOut>(stash + w * JC, sub);
            int not_one = 0;
            if (n_persons && live)
                for (int c = sub; c < C; c += G) not_one |= n_persons[f * C + c] != 1;
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) {
                sum += __shfl_xor(sum, off, 64);
                not_one |= __shfl_xor(not_one, off, 64);
            }
            if (live && sub == 0) {
                const double avg = sum / (double)JC;
                favg[w] = avg;
                // :151-152, as in k_fused_lean
                if (!(avg >= prm.score_tol) || fabs(avg - prm.score_tol) < 1e-6 * fabs(prm.score_tol) || not_one != 0)
                    atomicOr(&slowbits[0], 1u << w);
            }
        }
    }
    SNOWTRI_STAMP(5);
    __syncthreads();   // slowbits and favg are final
    const uint32_t slow = slowbits[0];
    if (wave == 0 && lane < nf && !((slow >> lane) & 1u)) {
        const int64_t f = f0 + lane;
        out_count[f] = 1;
        if (out_ps) out_ps[f] = (TOut)favg[lane];
        if (out_flags) out_flags[f] = kFlagFast;
    }
    SNOWTRI_STAMP(6);
    if (slow == 0u) return;   // (uniform: every thread reads the same word)
    // ---- rare: frames the speculation could not resolve -> the reference's full algorithm, by the whole workgroup
    {
        const PackedWriter<TOut> wr{out4, out_ps};
        double *slab = reinterpret_cast<double *>(scratch + (size_t)blockIdx.x * scratch_per_block);
        uint32_t m = slow;
        __syncthreads();  // general_frame reuses the front of the LDS (the finalising reads of favg are done)
        while (m) {
            const int bpos = __ffs((int)m) - 1;
            m &= m - 1u;
            general_frame<TIn>(f0 + bpos, 1, JC, NP, rig, kpts, n_persons, prm, 1, wr, out_count, out_flags, slab, smem);
        }
    }
}

}  // namespace snowtri

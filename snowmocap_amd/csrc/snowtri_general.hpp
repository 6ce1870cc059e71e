// snowtri_general.hpp -- k_frame_recompute: the reference's full algorithm (any person count, any
// thresholds) for one frame per workgroup WITHOUT materialising candidates in HBM.
//
// The reference separates A3 and A4 by a candidate list of Kc x J x 32 B per frame (1.9 MB for
// 8 cameras x 4 persons, 32 MB for 16 x 8): spilling it makes the multi-person configs HBM-bound
// at ~7x their arithmetic cost.  Here every pair solve is recomputed instead (SURVEY.md §8a note):
//
//   phase 1  per candidate: mean of the J joint scores -> keep flag        (triangulation.py:70-81)
//            rays of a joint chunk are built once into LDS; one lane per candidate walks the chunk
//   phase 2  kept list (ordered), centre joints, greedy clustering by one wave, its state in the idle
//            LDS chunk when it fits (every step is a dependent load)            (triangulation.py:107-130)
//   phase 3  per surviving cluster and joint: sum s, sum s*(Wm+Ws) over the members, recomputing
//            their solves in a second sweep over the joint chunks (rays back in LDS); thread = (joint of
//            the chunk, member group), group partials added with shuffles        (triangulation.py:136-152)
// Only O(Kc) bookkeeping (57 B per candidate slot) and the fused joint scores of up to 64 persons
// live in a per-workgroup scratch slab.
#pragma once
#include "snowtri_cluster.hpp"

namespace snowtri {

constexpr int kRecomputeMaxKn = 256;          // keypoint_num bound: one thread per joint in the DLT phase, chunk joints x groups <= 256
constexpr int kRayChunkBytes = 40 * 1024;     // LDS budget for one chunk of rays + scores (the power-of-two joint count rarely needs more); the tables follow at this fixed offset
constexpr int kPairTabMaxPairs = 120;         // camera pairs whose constants (d, t_m + t_s, camera indices) also live in LDS: <= 16 cameras

constexpr int kRecomputeSlotTile = 64;        // fused persons whose joint scores are parked per sweep (phase 3)

__host__ __device__ constexpr size_t recompute_scratch_bytes(int64_t Kc, int R, int kn) {
    return (((size_t)Kc * 64) + (size_t)R * 8 + 1024 + (size_t)kRecomputeSlotTile * kn * 8 + 255) & ~(size_t)255;
}

// ---- LDS map of k_frame_recompute -------------------------------------------------------------------------------
//   [0, 256)            reduction slots + flags
//   [256, ...)          camera-pair table (56 B per pair) and the ray matrices M[C][9]
//   [arena_off, total)  the ARENA, reused phase by phase:
//       phase 1   candidate score sums (when they fit, see p1_sums_in_lds) + one joint chunk of ray records (below)
//       phase 2   centres and cluster ids of the kept candidates
//       hand-over descriptor staging
//       phase 3   (frames that are not handed over) ray rows of a joint chunk in the row-major layout of
//                 recompute_chunk_joints (<= kRayChunkBytes) + the clusters' member words behind it
//
// Phase-3 chunk (row-major): rays[R][Jc] with a row stride of 32 Jc + 16 bytes and scores[R][Jc | 1].  Joints per
// chunk: what fits kRayChunkBytes, rounded DOWN to a power of two -- phase 3 maps thread = (joint of the chunk, member
// group of G lanes) with G a power of two and joints x G <= 256.
__host__ __device__ inline int recompute_chunk_joints(int R, int J, int score_bytes) {
    const int per_row = kRayChunkBytes / R - 16 - score_bytes;
    int jc = per_row / (32 + score_bytes);
    if (jc < 1) return 0;
    int p2 = 1;
    while (2 * p2 <= jc && 2 * p2 <= 256) p2 *= 2;
    return p2 > J ? J : p2;
}
__host__ __device__ inline size_t recompute_ray_stride(int Jc) { return (size_t)32 * Jc + 16; }
__host__ __device__ inline int recompute_score_stride(int Jc) { return Jc | 1; }

// bytes of the LDS pair table (6 doubles + 2 int32 per pair); the kernel needs npairs <= kPairTabMaxPairs
__host__ __device__ inline size_t recompute_pairtab_bytes(int npairs) { return (size_t)npairs * 56; }
__host__ __device__ inline size_t recompute_arena_offset(int C, int npairs) {
    return ((size_t)256 + recompute_pairtab_bytes(npairs) + (size_t)72 * C + 15) & ~(size_t)15;
}
// LDS of k_frame_recompute; the launcher asks for at least kRecomputeLdsBytes (three workgroups per CU share 160 KB).
// (52 KB, not 53: the occupancy API still answers 3 at 53 KB, but the hardware then admits only two -- measured as
// wave lifetimes of half the kernel's duration; the LDS is handed out in granules that 54 272 B does not fill evenly.)
constexpr int kRecomputeLdsBytes = 52 * 1024;
__host__ __device__ inline size_t recompute_lds_bytes(int C, int npairs) {
    return recompute_arena_offset(C, npairs) + (size_t)kRayChunkBytes + 16;
}

// Phase-1 chunk (joint-major): one 40-byte record {x, y, z, |h|^2, score} per (joint of the chunk, ray row),
//   address = jj * (40 R + 8) + 40 r.
// A phase-1 lane reads the records of ITS rows for the same joint jj as every other lane of its wave: 8-byte slot
// (5 r + field) mod 32 -- rows that differ mod 32 never share a bank, equal rows broadcast (conflict-free ds_read_b64
// for every rig of <= 32 rows, and for the row sets one wave touches on larger rigs); the rows of one camera are 40 B
// apart, so the GS second rays of a lane's group sit at compile-time offsets from one address register.  The 8 bytes
// of padding per joint spread the fill's writes (lanes = consecutive joints of one row) over the banks.
constexpr int kP1Rec = 40;
__host__ __device__ inline int p1_joint_stride(int R) { return kP1Rec * R + 8; }
// item = (camera pair, person of its first camera, group of GS persons of its second); JS = how many ways a chunk's
// joints are split over the waves when one pass of the workgroup has spare waves (wave-uniform: whole waves take a
// joint sub-range, so the lanes of a wave still read the same joint)
__host__ __device__ inline int p1_joint_split(int nitems) {
    const int iw = (nitems + 63) >> 6;
    return iw <= 1 ? 4 : (iw == 2 ? 2 : 1);
}
__host__ __device__ inline int p1_group_size(int Pmax) { return (Pmax & 3) == 0 ? 4 : ((Pmax & 1) == 0 ? 2 : 1); }
// the candidate score sums live in LDS while phase 1 runs when JS x Kc doubles take <= 8 KB
__host__ __device__ inline bool p1_sums_in_lds(int npairs, int Pmax) {
    const int gs = p1_group_size(Pmax);
    const long long kc = (long long)npairs * Pmax * Pmax;
    return kc * p1_joint_split((int)(kc / gs > 0x7fffffff ? 0x7fffffff : kc / gs)) <= 1024;
}
__host__ __device__ inline int p1_sum_bytes(int npairs, int Pmax) {
    if (!p1_sums_in_lds(npairs, Pmax)) return 0;
    const int gs = p1_group_size(Pmax);
    const int kc = npairs * Pmax * Pmax;
    return kc * p1_joint_split(kc / gs) * 8;
}
// joints per phase-1 chunk: as many as the arena holds, then evened out over the chunks (133 joints: 34+33+33+33
// instead of 4 x 32 + 5)
__host__ __device__ inline int p1_chunk_joints(int R, int J, int arena_bytes, int sum_bytes) {
    const int cap = (arena_bytes - sum_bytes) / p1_joint_stride(R);
    if (cap < 1) return 0;
    const int jmax = cap > 64 ? 64 : cap;
    const int nch = (J + jmax - 1) / jmax;
    return (J + nch - 1) / nch;
}

// host + device: the LDS a launch asks for, and whether both chunk layouts hold at least one joint
__host__ __device__ inline size_t recompute_launch_lds(int C, int npairs) {
    const size_t need = recompute_lds_bytes(C, npairs);
    return need > (size_t)kRecomputeLdsBytes ? need : (size_t)kRecomputeLdsBytes;
}
__host__ __device__ inline bool recompute_shape_ok(int C, int Pmax, int J, int npairs, int score_bytes) {
    const int R = C * Pmax;
    if (R > 1024 || npairs > kPairTabMaxPairs) return false;
    const int arena = (int)(recompute_launch_lds(C, npairs) - recompute_arena_offset(C, npairs));
    return recompute_chunk_joints(R, J, score_bytes) >= 1 && p1_chunk_joints(R, J, arena, p1_sum_bytes(npairs, Pmax)) >= 1;
}

// ---- phase-1 records in LDS ------------------------------------------------------------------------------------
template <typename TIn>
__device__ __forceinline__ void p1_store_record(char *rec, const RayRec &h, TIn s) {
    double *p = reinterpret_cast<double *>(rec);
    p[0] = h.x;
    p[1] = h.y;
    p[2] = h.z;
    p[3] = h.a;
    if constexpr (sizeof(TIn) == 4)
        *reinterpret_cast<unsigned long long *>(rec + 32) = (unsigned long long)__float_as_uint((float)s);   // one 8-byte slot
    else
        p[4] = (double)s;
}
// Reads of the solve loops: single ds_read_b64 each (2 LDS cycles per wave-instruction, 32-lane groups over 32 8-byte
// bank pairs).  Volatile keeps the compiler from pairing neighbouring fields into ds_read2_b64 -- which the LDS serves as
// two passes of 4 x 16 lanes, at HALF the bytes per clock -- and from narrowing the score slot to a 4-byte read (its
// 4-byte banks alias rows 16 apart).
// (lds_cv_f64 / lds_cv_u64: snowtri_math.hpp -- explicit LDS address space, a volatile generic access is a flat load)
template <typename TIn>
__device__ __forceinline__ TIn p1_load_score(const char *rec) {
    if constexpr (sizeof(TIn) == 4)
        return (TIn)__uint_as_float((uint32_t)*(lds_cv_u64)(rec + 32));
    else
        return (TIn) * (lds_cv_f64)(rec + 32);
}
__device__ __forceinline__ RayRec p1_load_ray(const char *rec) {
    lds_cv_f64 p = (lds_cv_f64)rec;
    RayRec r;
    r.x = p[0];
    r.y = p[1];
    r.z = p[2];
    r.a = p[3];
    return r;
}

// ---- the accurate phase-1 sum of ONE candidate over the joints [0, nj) of a chunk: sum of 2000 x the pair score
// (triangulation.py:70-78 without the 1/2000 of :72, applied by the caller).  1/dist by v_rsq_f64 + one Newton step
// (2e-14), exact intersection -> inf, four joints share one reciprocal unless their determinants' product leaves the
// normal range (singular pairs are detected there).  Used for the candidates the fast sums of p1_item_sums cannot
// decide, and for every candidate when negative confidences may pass the keypoint gate.
//   ra, rb: records of the candidate's two ray rows for the chunk's first joint; consecutive joints are jstr bytes apart
template <typename TIn>
__device__ __forceinline__ double candidate_chunk_sum_exact(const char *__restrict__ ra, const char *__restrict__ rb, int jstr, int nj,
                                                            const Vec3 &d, const Params &prm, bool &sing) {
    double acc = 0.0;
    // one solve, reciprocal of the determinant supplied (A2 + the score of :72-74)
    auto finish = [&](const RayRec &a, const RayRec &b, double bq, double e, double g, double inv, TIn sm, TIn ss) {
        const double S0 = fma(b.a, e, -(bq * g)) * inv;
        const double S1 = fma(a.a, g, -(bq * e)) * inv;
        const Vec3 df = {fma(b.x, S1, fma(a.x, S0, -d.x)), fma(b.y, S1, fma(a.y, S0, -d.y)),
                         fma(b.z, S1, fma(a.z, S0, -d.z))};
        const double d2 = dot3(df, df);
        const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(d2 > prm.dthr2);  // :73-74
        double idist = rsq_nr1(d2);
        idist = (d2 == 0.0) ? __builtin_inf() : idist;
        // the gate ASSIGNS 0 (:73-74): select after the product, 0 * inf (exact intersection) would be NaN
        acc += kp_ ? sum_score(sm, ss) * idist : 0.0;                                     // :72
    };
    int jj = 0;
    // four joints at a time share ONE reciprocal (Montgomery): v_rcp_f64 issues at quarter rate.
    for (; jj + 4 <= nj; jj += 4) {
        RayRec a[4], b[4];
        TIn sm[4], ss[4];
        double bq[4], e[4], g[4], det[4], inv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = p1_load_ray(ra + (jj + u) * jstr);
            b[u] = p1_load_ray(rb + (jj + u) * jstr);
            sm[u] = p1_load_score<TIn>(ra + (jj + u) * jstr);
            ss[u] = p1_load_score<TIn>(rb + (jj + u) * jstr);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bq[u] = fma(a[u].z, b[u].z, fma(a[u].y, b[u].y, a[u].x * b[u].x));
            e[u] = fma(a[u].z, d.z, fma(a[u].y, d.y, a[u].x * d.x));
            g[u] = fma(b[u].z, d.z, fma(b[u].y, d.y, b[u].x * d.x));
            det[u] = fma(a[u].a, b[u].a, -(bq[u] * bq[u]));
            // singular as the reference's LU sees equal rays and as the oracle defines it: a c == b b in separately rounded
            // products (skew_ray_solve).  The fused determinant above is the rounding error of b b there -- not 0.
            sing |= a[u].a * b[u].a == bq[u] * bq[u];
        }
        const double p01 = det[0] * det[1], p012 = p01 * det[2], p0123 = p012 * det[3];
        // a product outside the normal range (a singular or wildly conditioned pair in the group) falls back to
        // four separate reciprocals
        if (fabs(p0123) > 1e-250 && fabs(p0123) < 1e250) {
            double run = rcp_nr2(p0123);
            inv[3] = run * p012;
            run *= det[3];
            inv[2] = run * p01;
            run *= det[2];
            inv[1] = run * det[0];
            inv[0] = run * det[1];
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) inv[u] = rcp_nr2(det[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) finish(a[u], b[u], bq[u], e[u], g[u], inv[u], sm[u], ss[u]);
    }
    for (; jj < nj; jj++) {
        const RayRec a = p1_load_ray(ra + jj * jstr), b = p1_load_ray(rb + jj * jstr);
        const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
        const double e = fma(a.z, d.z, fma(a.y, d.y, a.x * d.x));
        const double g = fma(b.z, d.z, fma(b.y, d.y, b.x * d.x));
        const double det = fma(a.a, b.a, -(bq * bq));
        sing |= a.a * b.a == bq * bq;
        finish(a, b, bq, e, g, rcp_nr2(det), p1_load_score<TIn>(ra + jj * jstr), p1_load_score<TIn>(rb + jj * jstr));
    }
    return acc;
}

// ---- the fast phase-1 arithmetic: GS candidates that share their FIRST ray -- one person of camera m against GS
// consecutive persons of camera s -- over the joints [0, nj) of a chunk; acc[u] += 2000 x score of candidate u (:72-74).
//
// What phase 1 needs of a pair solve is only the DISTANCE of the two rays (the 3D point is needed for the centre joint
// and for the candidates that survive: phase 2 / phase 3).  For rays t_m + S0 a and t_s - S1 b the closest approach the
// reference computes with its 2x2 solve (triangulation.py:24-31) is the distance of two skew lines:
//     dist = |d . (a x b)| / |a x b|,      d = t_s - t_m,    |a x b|^2 = |a|^2 |b|^2 - (a.b)^2 = det(H^T H).
// With c = d x a per FIRST ray (shared by the GS candidates): d . (a x b) = c . b, so a solve is two dot products, the
// determinant, and ONE transcendental:  1/dist = det * rsq((c.b)^2 det)  -- 20 VALU instead of 41, no reciprocal
// of the determinant, nothing shared between joints (any joint count, no tail).  Conditioning: c.b cancels to
// dist |a x b| from terms of size |d| |a| |b|, i.e. a relative error of 1e-16 |d| / dist ~ 5e-13 at 1 mm -- the reference's own
// ||Wm - Ws|| cancels from 5 m coordinates to the same distance (1e-12).
// The distance gate dist > dthr (:74) is taken on accurately rounded products, (c.b)^2 > dthr^2 det (both sides x det > 0).
// A determinant that is 0, negative (rounding of nearly parallel rays) or NaN makes 1/dist itself NaN (0 x inf, rsq of
// a negative number), and a NaN factor reaches the sum whatever the gates select (0 x NaN); an exact intersection gives
// inf -- and a sum that is not finite is re-done with the accurate arithmetic, which also flags singular pairs.  The raw
// v_rsq_f64 (measured 2^-24.2 relative) only scales a score: the caller re-does candidates whose mean lies within 1e-6 of
// average_score_threshold (:79-81).
//   pa: record of the first ray's row, pb: record of the FIRST of the GS second rows (rows are kP1Rec bytes apart)
template <int GS, typename TIn>
__device__ __forceinline__ void p1_item_sums(const char *__restrict__ pa, const char *__restrict__ pb, int jstr, int nj,
                                             const Vec3 &d, const Params &prm, double (&acc)[GS]) {
#pragma clang fp contract(off)   // (the determinant must not be fused: see p1_tile_sums)
    for (int t = 0; t < nj; t++, pa += jstr, pb += jstr) {
        const RayRec a = p1_load_ray(pa);
        const TIn sm = p1_load_score<TIn>(pa);
        RayRec b[GS];
        TIn ss[GS];
#pragma unroll
        for (int u = 0; u < GS; u++) {
            b[u] = p1_load_ray(pb + kP1Rec * u);
            ss[u] = p1_load_score<TIn>(pb + kP1Rec * u);
        }
        const double cx = fma(d.y, a.z, -(d.z * a.y)), cy = fma(d.z, a.x, -(d.x * a.z)), cz = fma(d.x, a.y, -(d.y * a.x));
        const bool okm = !below_kthr(sm, prm);
#pragma unroll
        for (int u = 0; u < GS; u++) {
            const double bq = fma(a.z, b[u].z, fma(a.y, b[u].y, a.x * b[u].x));
            const double det = a.a * b[u].a - bq * bq;   // separately rounded: singular (a c == b b) <=> det == 0 -> NaN -> exact pass
            const double dn = fma(cz, b[u].z, fma(cy, b[u].y, cx * b[u].x));
            const double dn2 = dn * dn;
            const bool kp_ = okm && !below_kthr(ss[u], prm) && !(dn2 > det * prm.dthr2);   // :73-74
            acc[u] = fma(gated_sum_sel(sm, ss[u], kp_), det * __builtin_amdgcn_rsq(dn2 * det), acc[u]);
        }
    }
}

// Dynamic LDS = recompute_lds_bytes(R, J, kn, sizeof(TIn)); scratch = gridDim.x slabs of
// recompute_scratch_bytes(Kc, R, kn).  R = C * Pmax ray rows.  Requires keypoint_num <= kRecomputeMaxKn.
constexpr int kRecomputeWaves = 3;
constexpr int kRecomputeUnroll = 4;
// Frames are handed out through an atomic counter (next_frame, zeroed by the host before the launch): the
// time of a frame depends on how many candidates survive, so a static frame->workgroup map leaves CUs idle.
//
// METHOD = 1 (SNOWTRI_DLT with several detections per camera -- row N3; not reference behaviour): phases 1-2
// are the reference's association unchanged; phase 3 instead solves, per surviving cluster and joint, the
// N-view DLT over the DISTINCT (camera, person) observations its member candidates are made of, keeping those
// whose confidence is not below keypoint_score_threshold (>= 2 needed, else the joint stays (0,0,0)/0);
// joint score = their mean confidence.  One lane per (cluster, joint); no cross-lane reduction.
template <int METHOD, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, kRecomputeWaves) void k_frame_recompute(int64_t F, int Pmax, int J, int Kc, Rig rig,
                                                            const TIn *__restrict__ kpts,
                                                            const int32_t *__restrict__ n_persons, Params prm,
                                                            int Pout, TOut *__restrict__ out4,
                                                            TOut *__restrict__ out_ps,
                                                            int32_t *__restrict__ out_count,
                                                            uint32_t *__restrict__ out_flags, char *scratch,
                                                            size_t scratch_per_block,
                                                            unsigned long long *next_frame, int lds_total,
                                                            ClusterDesc *__restrict__ desc, uint32_t *__restrict__ hand_words,
                                                            unsigned long long *hand_counters, uint32_t desc_cap,
                                                            uint32_t word_cap, const uint32_t *__restrict__ frame_list,
                                                            const unsigned long long *__restrict__ frame_list_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the launch behind the streaming route finds its list empty on the reference's workloads: leave before the tables
    if (frame_list && (unsigned long long)blockIdx.x >= *frame_list_count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax;
    const int kn = prm.kn, ci = prm.center;
    // ---- LDS map (see the top of this file)
    double *red = reinterpret_cast<double *>(smem);             // [4] + misc
    int32_t *misc = reinterpret_cast<int32_t *>(red + kBlock / 64);
    // camera-pair constants in LDS: d = t_s - t_m, t_m + t_s (6 doubles) and the two camera indices -- the solve loops
    // index them per lane (host-checked: npairs <= kPairTabMaxPairs, i.e. <= 16 cameras; larger rigs take k_frame_general)
    double *pairc = reinterpret_cast<double *>(smem + 256);
    int32_t *pairs = reinterpret_cast<int32_t *>(pairc + 6 * rig.npairs);
    double *Ml = reinterpret_cast<double *>(pairs + 2 * rig.npairs);    // [C][9] ray matrices (per-lane camera in the fill)
    for (int i = tid; i < 6 * rig.npairs; i += kBlock) pairc[i] = rig.pairc[i];
    for (int i = tid; i < 2 * rig.npairs; i += kBlock) pairs[i] = rig.pairs[i];
    for (int i = tid; i < 9 * C; i += kBlock) Ml[i] = rig.M[i];
    const int arena_off = (int)recompute_arena_offset(C, rig.npairs);
    char *arena = smem + arena_off;
    const int arena_bytes = lds_total - arena_off;
    // phase-3 chunk geometry (row-major rays + scores, <= kRayChunkBytes); what the arena has left behind it holds the
    // member words of phase 3
    const int Jc = recompute_chunk_joints(R, J, (int)sizeof(TIn));
    const int rstride = (int)recompute_ray_stride(Jc);          // bytes between the ray rows (32-bit LDS offsets)
    const int sstride = recompute_score_stride(Jc);             // elements between the score rows
    char *rays = arena;                                         // [R] rows of Jc RayRec (+16 B pad)
    TIn *rsc = reinterpret_cast<TIn *>(arena + (size_t)R * rstride);                     // [R][sstride]
    uint32_t *lmem_lds = reinterpret_cast<uint32_t *>(arena + kRayChunkBytes);
    const int lmem_cap = arena_bytes > kRayChunkBytes ? (arena_bytes - kRayChunkBytes) / 4 : 0;
    // phase-1 geometry (joint-major records)
    const int gs_full = p1_group_size(Pmax);                    // candidates per item when every camera lists Pmax persons
    const int sum_bytes = p1_sum_bytes(rig.npairs, Pmax);
    double *lsum = reinterpret_cast<double *>(arena);           // [JS][Kc] raw score sums of phase 1, if sum_bytes != 0
    char *p1rec = arena + sum_bytes;
    const int jstr = p1_joint_stride(R);
    const int Jc1 = p1_chunk_joints(R, J, arena_bytes, sum_bytes);
    const unsigned long long magic_pmax = (((unsigned long long)1 << 40) + (unsigned)Pmax - 1) / (unsigned)Pmax;
    const bool exact_only = prm.kthr < 0.0;   // negative scores may pass the keypoint gate: no relative error bound on a sum

    // per-workgroup bookkeeping slab (global, reused frame after frame)
    char *slab = scratch + (size_t)blockIdx.x * scratch_per_block;
    double *sum = reinterpret_cast<double *>(slab);                 // [Kc] candidate score sums
    double *centre = sum + Kc;                                      // [Kc][3] centre joints of kept candidates
    int32_t *kidx = reinterpret_cast<int32_t *>(centre + 3 * (size_t)Kc);  // [Kc] kept slots in list order
    int32_t *cluster_of = kidx + Kc;                                // [Kc]
    int32_t *csize = cluster_of + Kc;                               // [Kc]
    int32_t *cseed = csize + Kc;                                    // [Kc]
    int32_t *members = cseed + Kc;                                  // [Kc] kept indices grouped by cluster
    int32_t *cstart = members + Kc;                                 // [Kc + 1]
    uint8_t *keep = reinterpret_cast<uint8_t *>(cstart + Kc + 1);   // [Kc]
    // [Kc] this frame's candidates: ray rows rm | rs << 10 and camera pair q << 20 (host-checked: R <= 1024,
    // npairs <= kPairTabMaxPairs), or kNoCand where a camera lists fewer persons than the slot's
    uint32_t *cw = reinterpret_cast<uint32_t *>(slab + (((size_t)Kc * 57 + 4 + 3) & ~(size_t)3));
    constexpr uint32_t kNoCand = 0xffffffffu;
    int32_t *rowlist = reinterpret_cast<int32_t *>(slab + (((size_t)Kc * 64 + 1024) & ~(size_t)7));  // [R] (DLT)
    int32_t *rowflag = rowlist + R;                                                                   // [R] (DLT)
    double *osbuf = reinterpret_cast<double *>(rowflag + R);   // [kRecomputeSlotTile][kn] fused joint scores

    const PackedWriter<TOut> wr{out4, out_ps};
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);

    for (;;) {
        if (tid == 0) {
            const unsigned long long nf = atomicAdd(next_frame, 1ull);
            misc[2] = (int32_t)(nf & 0xffffffffu);
            misc[3] = (int32_t)(nf >> 32);
        }
        __syncthreads();
        int64_t f = (int64_t)(((unsigned long long)(uint32_t)misc[3] << 32) | (uint32_t)misc[2]);
        // (frame_list: only the listed frames -- the ones k_associate left behind, snowtri_assoc.hpp)
        if (frame_list) {
            if ((unsigned long long)f >= *frame_list_count) break;
            f = (int64_t)frame_list[f];
        }
        if (f >= F) break;
        SNOWTRI_DEV_CHECK(f >= 0 && (size_t)blockIdx.x * scratch_per_block + scratch_per_block <= (size_t)gridDim.x * scratch_per_block, 40);   // frame and slab of this workgroup
        const int32_t *np_f = n_persons ? n_persons + f * C : nullptr;
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * J;
        bool ragged = false;
        for (int k = tid; k < Kc; k += kBlock) {
            // candidate order of triangulation.py:56-65: camera pair, person of the first camera, person of the second
            const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
            const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
            const int nm = np_f ? np_f[mc] : Pmax, ns = np_f ? np_f[sc] : Pmax;
            cw[k] = (pm < nm && ps < ns) ? ((uint32_t)(mc * Pmax + pm) | ((uint32_t)(sc * Pmax + ps) << 10) | ((uint32_t)q << 20))
                                         : kNoCand;
            ragged |= !(pm < nm && ps < ns);
        }
        if (tid == 0 && out_flags) out_flags[f] = 0u;
        bool sing = false;
        // every camera lists Pmax persons: phase 1 runs on groups of GS candidates that share their first ray; a ragged
        // frame (or an odd Pmax) keeps one candidate per lane.  (The barrier also orders cw[] before its readers.)
        const int GS = __syncthreads_or(ragged ? 1 : 0) ? 1 : gs_full;
        const int NG = Pmax / GS, per_q = Pmax * NG, nitems = rig.npairs * per_q;
        const int JS = sum_bytes ? p1_joint_split(Kc / gs_full) : 1;   // (the sums' LDS is sized for the full-frame split)
        const int wpg = (kBlock / 64) / JS;                            // waves that share a joint sub-range
        const int wv = tid >> 6, jsub = wv / wpg, iw = wv - jsub * wpg;
        // raw sums (2000 x the score of :72), [JS][Kc]: in LDS if they fit, else in the slab (two explicit address
        // spaces: a generic pointer costs flat accesses and 64-bit address arithmetic)
        const bool acc_lds = sum_bytes != 0;
        auto acc_get = [&](int idx) -> double { return acc_lds ? lsum[idx] : sum[idx]; };
        auto acc_put = [&](int idx, double v) {
            if (acc_lds)
                lsum[idx] = v;
            else
                sum[idx] = v;
        };
        for (int k = tid; k < (acc_lds ? JS * Kc : Kc); k += kBlock) acc_put(k, 0.0);

        // ---------------- phase 1: candidate score sums, joint chunk by joint chunk -------------
        for (int j0 = 0; j0 < J; j0 += Jc1) {
            const int nj = (J - j0) < Jc1 ? (J - j0) : Jc1;
            const unsigned long long magic_nj = (((unsigned long long)1 << 40) + (unsigned)nj - 1) / (unsigned)nj;
            __syncthreads();
            // fill: one record per (ray row, joint of the chunk); lanes = consecutive joints of a row (coalesced reads)
            for (int i = tid; i < R * nj; i += kBlock) {
                const int r = (int)(((unsigned long long)(unsigned)i * magic_nj) >> 40), jj = i - r * nj;
                const int c = (int)(((unsigned long long)(unsigned)r * magic_pmax) >> 40), p = r - c * Pmax;
                const int nc = np_f ? np_f[c] : Pmax;
                if (p < nc) {
                    const Kp3<TIn> kp = kpf[(size_t)r * J + j0 + jj];
                    p1_store_record<TIn>(p1rec + jj * jstr + kP1Rec * r, make_ray(Ml + 9 * c, kp.u, kp.v), kp.s);
                }
            }
            __syncthreads();
            if (exact_only) {
                for (int k = tid; k < Kc; k += kBlock) {
                    const uint32_t w = cw[k];
                    if (w == kNoCand) continue;
                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    acc_put(k, acc_get(k) + candidate_chunk_sum_exact<TIn>(p1rec + kP1Rec * rm, p1rec + kP1Rec * rs, jstr, nj, d, prm, sing));
                }
            } else {
                // the waves of joint sub-range jsub walk joints [jlo, jhi) of the chunk; item = base + lane of the round
                const int jlo = jsub * nj / JS, jhi = (jsub + 1) * nj / JS;
                const unsigned long long magic_pq = (((unsigned long long)1 << 40) + (unsigned)per_q - 1) / (unsigned)per_q;
                const unsigned long long magic_ng = (((unsigned long long)1 << 40) + (unsigned)NG - 1) / (unsigned)NG;
                for (int base = iw * 64; base < nitems; base += wpg * 64) {
                    const int item = base + lane;
                    const bool live = item < nitems;
                    const int it = live ? item : 0;
                    const int q = (int)(((unsigned long long)(unsigned)it * magic_pq) >> 40), r2 = it - q * per_q;
                    const int pm = (int)(((unsigned long long)(unsigned)r2 * magic_ng) >> 40), ps0 = (r2 - pm * NG) * GS;
                    const int k0 = q * pp + pm * Pmax + ps0;
                    const int rm = pairs[2 * q] * Pmax + pm, rs0 = pairs[2 * q + 1] * Pmax + ps0;
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    const char *pa = p1rec + jlo * jstr + kP1Rec * rm, *pb = p1rec + jlo * jstr + kP1Rec * rs0;
                    const int dst = jsub * Kc + k0;
                    if (GS == 4) {
                        double old[4], acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int u = 0; u < 4; u++) old[u] = live ? acc_get(dst + u) : 0.0;   // (in flight during the solves)
                        p1_item_sums<4, TIn>(pa, pb, jstr, live ? jhi - jlo : 0, d, prm, acc);
                        if (live) {
#pragma unroll
                            for (int u = 0; u < 4; u++) acc_put(dst + u, old[u] + acc[u]);
                        }
                    } else if (GS == 2) {
                        double old[2], acc[2] = {0.0, 0.0};
#pragma unroll
                        for (int u = 0; u < 2; u++) old[u] = live ? acc_get(dst + u) : 0.0;
                        p1_item_sums<2, TIn>(pa, pb, jstr, live ? jhi - jlo : 0, d, prm, acc);
                        if (live) {
#pragma unroll
                            for (int u = 0; u < 2; u++) acc_put(dst + u, old[u] + acc[u]);
                        }
                    } else {
                        const bool cand = live && cw[k0] != kNoCand;   // a ragged frame's empty slots stay at sum 0
                        double acc[1] = {0.0};
                        const double old = cand ? acc_get(dst) : 0.0;
                        p1_item_sums<1, TIn>(pa, pb, jstr, cand ? jhi - jlo : 0, d, prm, acc);
                        if (cand) acc_put(dst, old + acc[0]);
                    }
                }
            }
        }
        // the 1 / (2 * 1000) of :72 and the JS partial sums of a candidate; then the candidates whose fast sum cannot
        // decide :80-81 (see p1_item_sums): not finite, or within 1e-6 relative of average_score_threshold (the fast sum
        // is within 2.4e-7 of the accurate one).  Rare: a second sweep over the joint chunks re-does just those with the
        // accurate arithmetic; exactly singular pairs are flagged there.  (Thread tid owns slots tid, tid + 256, ... in
        // every loop below.)
        __syncthreads();
        int redo_any = 0;
        for (int k = tid; k < Kc; k += kBlock) {
            double v = acc_get(k);
            for (int s_ = 1; s_ < JS; s_++) v += acc_get(s_ * Kc + k);
            const double s_ = v * 0.0005, mean = s_ / (double)J;
            const bool redo = !exact_only && (!(fabs(mean) < 1e300) || (s_ != 0.0 && fabs(mean - prm.avg_thr) <= 1e-6 * fabs(mean)));
            sum[k] = redo ? 0.0 : s_;
            keep[k] = redo ? 2 : 0;      // (invalid slots hold sum 0: never re-done)
            redo_any |= redo ? 1 : 0;
        }
        redo_any = __syncthreads_or(redo_any);   // (also: the sums in the arena are consumed)
        if (redo_any) {
            for (int j0 = 0; j0 < J; j0 += Jc1) {
                const int nj = (J - j0) < Jc1 ? (J - j0) : Jc1;
                const unsigned long long magic_nj = (((unsigned long long)1 << 40) + (unsigned)nj - 1) / (unsigned)nj;
                __syncthreads();
                for (int i = tid; i < R * nj; i += kBlock) {
                    const int r = (int)(((unsigned long long)(unsigned)i * magic_nj) >> 40), jj = i - r * nj;
                    const int c = (int)(((unsigned long long)(unsigned)r * magic_pmax) >> 40), p = r - c * Pmax;
                    const int nc = np_f ? np_f[c] : Pmax;
                    if (p < nc) {
                        const Kp3<TIn> kp = kpf[(size_t)r * J + j0 + jj];
                        p1_store_record<TIn>(p1rec + jj * jstr + kP1Rec * r, make_ray(Ml + 9 * c, kp.u, kp.v), kp.s);
                    }
                }
                __syncthreads();
                for (int k = tid; k < Kc; k += kBlock) {
                    if (keep[k] != 2) continue;
                    const uint32_t w = cw[k];
                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    sum[k] += candidate_chunk_sum_exact<TIn>(p1rec + kP1Rec * rm, p1rec + kP1Rec * rs, jstr, nj, d, prm, sing);
                }
            }
            for (int k = tid; k < Kc; k += kBlock)
                if (keep[k] == 2) sum[k] *= 0.0005;
        }
        __syncthreads();
        if (sing && out_flags) atomicOr(&out_flags[f], 1u /*SNOWTRI_FLAG_SINGULAR*/);
        for (int k = tid; k < Kc; k += kBlock) {
            const bool valid = cw[k] != kNoCand;
            const double mean = sum[k] / (double)J;                                               // :79
            keep[k] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;                                   // :80-81
        }
        __syncthreads();

        // ---------------- phase 2: kept list, centre joints, greedy clustering -------------------
        if (tid < 64) {
            int n = 0;
            for (int base = 0; base < Kc; base += 64) {
                const int k = base + lane;
                const bool kp_ = k < Kc && keep[k] != 0;
                const unsigned long long m = __ballot(kp_);
                if (kp_) kidx[n + __popcll(m & ((1ull << lane) - 1ull))] = k;
                n += __popcll(m);
            }
            if (lane == 0) misc[0] = n;
        }
        __syncthreads();
        const int n = misc[0];
        // The clustering below is one wave walking the kept list: every step is a dependent load, so it runs
        // at memory LATENCY.  While it runs the ray chunk is idle: if the centres (24 B) and cluster ids (4 B) of
        // the n kept candidates fit there they live in LDS (~8x lower latency than the L2-resident slab).
        const bool in_lds = (size_t)n * 28 + 16 <= (size_t)arena_bytes;
        auto phase2 = [&](int32_t *cof, double *cen) {
            for (int i = tid; i < n; i += kBlock) {
                const uint32_t w = cw[kidx[i]];
                const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
                const Kp3<TIn> km = kpf[(size_t)rm * J + ci], ks = kpf[(size_t)rs * J + ci];
                const RayRec a = make_ray(Ml + 9 * mc, km.u, km.v), b = make_ray(Ml + 9 * sc, ks.u, ks.v);
                const double *pc = pairc + 6 * q;
                const PairSolve o = pair_solve_fast<true>(a, b, Vec3{pc[0], pc[1], pc[2]}, Vec3{pc[3], pc[4], pc[5]});
                cen[3 * i] = 0.5 * o.sw.x;
                cen[3 * i + 1] = 0.5 * o.sw.y;
                cen[3 * i + 2] = 0.5 * o.sw.z;
                cof[i] = -1;
            }
            __syncthreads();
            if (tid < 64) {
                // triangulation.py:107-130 -- seeds in list order, the last candidate never seeds,
                // distance to the SEED's centre, `dist > tol` skips (NaN absorbs)
                int ncl = 0;
                for (int next = 0;;) {
                    // first candidate >= next that is not absorbed yet (64 at a time)
                    int mc = -1;
                    for (int base = next & ~63; base < n - 1 && mc < 0; base += 64) {
                        const int i = base + lane;
                        const unsigned long long m = __ballot(i >= next && i < n - 1 && cof[i] == -1);
                        if (m) mc = base + __ffsll((long long)m) - 1;
                    }
                    if (mc < 0) break;
                    const double mx = cen[3 * mc], my = cen[3 * mc + 1], mz = cen[3 * mc + 2];
                    int cnt = 0;
                    for (int base = mc + 1; base < n; base += 64) {
                        const int sc = base + lane;
                        bool ab = false;
                        if (sc < n && cof[sc] == -1) {
                            const double dx = mx - cen[3 * sc], dy = my - cen[3 * sc + 1], dz = mz - cen[3 * sc + 2];
                            const double dist = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
                            if (!(dist > prm.ctol)) {
                                cof[sc] = ncl;
                                ab = true;
                            }
                        }
                        cnt += __popcll(__ballot(ab));
                    }
                    if (lane == 0) {
                        cof[mc] = ncl;
                        csize[ncl] = cnt + 1;
                        cseed[ncl] = mc;
                    }
                    ncl++;
                    next = mc + 1;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                // members grouped by cluster, list order inside each cluster
                int off = 0;
                for (int c = 0; c < ncl; c++) {
                    if (lane == 0) cstart[c] = off;
                    for (int base = cseed[c]; base < n; base += 64) {
                        const int i = base + lane;
                        const bool in = i < n && cof[i] == c;
                        const unsigned long long m = __ballot(in);
                        if (in) members[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
                        off += __popcll(m);
                    }
                }
                if (lane == 0) {
                    cstart[ncl] = off;
                    misc[1] = ncl;
                }
            }
        };
        if (in_lds)
            phase2(reinterpret_cast<int32_t *>(arena + (((size_t)n * 24 + 15) & ~(size_t)15)), reinterpret_cast<double *>(arena));
        else
            phase2(cluster_of, centre);
        __syncthreads();
        const int ncl = misc[1];

        // ---------------- phase 3: fusion per surviving cluster ------------------------------
        int nout = 0;
        if constexpr (METHOD == 1) {
            for (int cid = 0; cid < ncl; cid++) {
                const int size = csize[cid];
                if ((double)size < prm.num_tol) continue;                                         // :132-134
                const int m0 = cstart[cid];
                double ox = 0.0, oy = 0.0, oz = 0.0, os = 0.0;
                // distinct observation rows of this cluster, in row order
                for (int r = tid; r < R; r += kBlock) rowflag[r] = 0;
                __syncthreads();
                for (int mi = tid; mi < size; mi += kBlock) {
                    const int k = kidx[members[m0 + mi]];
                    const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
                    rowflag[pairs[2 * q] * Pmax + pm] = 1;
                    rowflag[pairs[2 * q + 1] * Pmax + ps] = 1;
                }
                __syncthreads();
                if (tid < 64) {
                    int nr = 0;
                    for (int base = 0; base < R; base += 64) {
                        const int r = base + lane;
                        const bool in = r < R && rowflag[r] != 0;
                        const unsigned long long m = __ballot(in);
                        if (in) rowlist[nr + __popcll(m & ((1ull << lane) - 1ull))] = r;
                        nr += __popcll(m);
                    }
                    if (lane == 0) misc[4] = nr;
                }
                __syncthreads();
                const int nrows = misc[4];
                if (tid < kn) {
                    double A[4][4];
    #pragma unroll
                    for (int i = 0; i < 4; i++)
    #pragma unroll
                        for (int k2 = 0; k2 < 4; k2++) A[i][k2] = 0.0;
                    double ssum = 0.0;
                    int cnt = 0;
                    for (int i = 0; i < nrows; i++) {
                        const int r = rowlist[i];
                        const Kp3<TIn> kp = kpf[(size_t)r * J + tid];
                        const bool use = !below_kthr(kp.s, prm);
                        dlt_add_observation(A, rig.P + 12 * (r / Pmax), (double)kp.u, (double)kp.v, use ? 1.0 : 0.0);
                        ssum += use ? (double)kp.s : 0.0;
                        cnt += use ? 1 : 0;
                    }
                    double e[4];
                    dlt_solve(A, cnt >= 2, e);
                    if (cnt >= 2) {
                        const double r = dlt_recip(e[3]);
                        ox = e[0] * r;
                        oy = e[1] * r;
                        oz = e[2] * r;
                        os = ssum * dlt_recip((double)cnt);
                    }
                }

                const double avg = block_sum(tid < kn ? os : 0.0, red) / (double)kn;               // :150
                if (!(avg < prm.score_tol)) {                                                      // :151-152
                    if (nout < Pout) {
                        if (tid < kn) wr.joint(f, Pout, kn, nout, tid, ox, oy, oz, os);
                        if (tid == 0) wr.person(f, Pout, nout, avg);
                    }
                    nout++;
                }
                __syncthreads();
            }
        } else {
            // Pairwise fusion (:136-152), second sweep over the joint chunks with the rays back in LDS:
            // thread (jj, g) walks members g, g + G, ... of a cluster for joint jj of the chunk; G is a power of
            // two, so the G partial sums of a joint sit in adjacent lanes and are added with shuffles -- no LDS
            // traffic and NO barrier per cluster.  Joints go straight to their output slot, assigned in cluster
            // order to the clusters that pass the size filter; their scores are parked in `osbuf` (up to
            // kRecomputeSlotTile persons per sweep) for the mean-score filter, applied at the end by compacting.
            // One word per member (its candidate word), cluster-member order; in LDS when the list fits what the
            // workgroup's allocation has left (read once per member solve: from the L2-resident slab every one of
            // them costs a memory round trip), else in the slab.
            // ---- hand-over (snowtri_cluster.hpp): when the mean-score filter (:150-152) of every cluster that passes the size
            // filter can be decided from the candidate means of phase 1, the frame's output persons become descriptors
            // for the streaming kernels and phase 3 is skipped: a cluster that is the complete graph over one detection
            // per camera becomes a complete-graph descriptor (persons packed in one word), any other a member-list descriptor (its member
            // words are appended to a global list).  The host passes `desc` only when the shape allows it (float32 outputs,
            // <= 8 cameras, <= 16 persons per camera, keypoint_num == J so that a person's mean score (:150) is the mean of
            // its members' candidate means (:79), keypoint threshold >= 0, the staging fits the idle ray chunk).
            bool handed = false;
            if constexpr (sizeof(TOut) == 4) {
                if (desc != nullptr) {
                    const size_t pq = ((size_t)Pout * 4 + 15) & ~(size_t)15;
                    uint32_t *st_a = reinterpret_cast<uint32_t *>(arena);            // [Pout] persons word | first member (cstart)
                    int32_t *st_size = reinterpret_cast<int32_t *>(arena + pq);      // [Pout] 0: complete graph, else members
                    uint32_t *st_idx = reinterpret_cast<uint32_t *>(arena + 2 * pq);  // [Pout] index inside its descriptor list
                    uint32_t *st_word = reinterpret_cast<uint32_t *>(arena + 3 * pq); // [Pout] offset of its member words
                    double *st_avg = reinterpret_cast<double *>(arena + 4 * pq);     // [Pout]
                    // the clusters' sizes / starts, member words and candidate score sums, fetched by the whole workgroup at
                    // once: the decisions below are then one wave walking LDS (from the L2-resident slab every step is a
                    // chain of dependent loads: 15 us per frame)
                    const int nmem = cstart[ncl];
                    double *l_sum = st_avg + Pout;                                   // [nmem]
                    uint32_t *l_word = reinterpret_cast<uint32_t *>(l_sum + nmem);   // [nmem]
                    int32_t *l_size = reinterpret_cast<int32_t *>(l_word + nmem);    // [ncl]
                    int32_t *l_start = l_size + ncl;                                 // [ncl]
                    const bool fits = 4 * pq + (size_t)Pout * 8 + (size_t)nmem * 12 + (size_t)ncl * 8 <= (size_t)arena_bytes;
                    if (fits) {
                        for (int pos = tid; pos < nmem; pos += kBlock) {
                            const int k = kidx[members[pos]];
                            l_word[pos] = cw[k];
                            l_sum[pos] = sum[k];
                        }
                        for (int c = tid; c < ncl; c += kBlock) {
                            l_size[c] = csize[c];
                            l_start[c] = cstart[c];
                        }
                    }
                    __syncthreads();
                    if (tid < 64) {
                        const int NPq = rig.npairs;
                        int ok = fits ? 1 : 0, no = 0;            // wave-uniform
                        uint32_t ncomp = 0, ngen = 0, nwords = 0;
                        for (int cid = 0; ok && cid < ncl; cid++) {
                            const int size = l_size[cid];
                            if ((double)size < prm.num_tol) continue;                                  // :132-134
                            const int m0 = l_start[cid];
                            double ssum = 0.0;
                            for (int base = 0; base < size; base += 64) ssum += base + lane < size ? l_sum[m0 + base + lane] : 0.0;
                            ssum = wave_sum(ssum);
                            const double avg = ssum / ((double)size * (double)J);                         // :150 from :79
                            // (a sum that is not finite, or a mean within 1e-6 of the tolerance -- the fast sums of phase 1
                            // are within 6e-8 -- is left to phase 3; a sum of exactly 0 is exact)
                            if (!(fabs(avg) < 1e300) || (ssum != 0.0 && fabs(avg - prm.score_tol) <= 1e-6 * fabs(avg))) {
                                ok = 0;
                                break;
                            }
                            if (avg < prm.score_tol) continue;                                         // :151-152
                            if (no < Pout) {
                                bool complete = false;
                                uint32_t nib = 0u;
                                if (size == NPq) {
                                    // members are in candidate order (camera pair major): in a complete graph member i IS pair
                                    // i; the row of camera 0 is the first row of pair (0,1) = member 0, the row of camera c >= 1
                                    // the second row of pair (0,c) = member c - 1, and every other member must repeat them.
                                    // (Shuffles stay outside selects: a lane masked off during ds_bpermute is read as 0.)
                                    const bool act = lane < size;
                                    const uint32_t w = act ? l_word[m0 + lane] : 0u;
                                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                                    const int mc = act ? pairs[2 * lane] : 0, sc = act ? pairs[2 * lane + 1] : 1;
                                    const int row0 = __shfl(rm, 0, 64);
                                    const int via_m = __shfl(rs, mc > 0 ? mc - 1 : 0, 64), via_s = __shfl(rs, sc - 1, 64);
                                    const bool good = !act || (q == lane && rm == (mc == 0 ? row0 : via_m) && rs == via_s);
                                    complete = __ballot(!good) == 0ull;
                                    const int via = __shfl(rs, lane > 0 ? lane - 1 : 0, 64);
                                    const int rowc = lane == 0 ? row0 : via;           // lane c < C: row of camera c
                                    nib = (complete && lane < C) ? (uint32_t)(rowc - lane * Pmax) << (4 * lane) : 0u;
#pragma unroll
                                    for (int off = 4; off > 0; off >>= 1) nib |= (uint32_t)__shfl_xor((int)nib, off, 64);
                                }
                                if (lane == 0) {
                                    st_a[no] = complete ? nib : (uint32_t)m0;
                                    st_size[no] = complete ? 0 : size;
                                    st_idx[no] = complete ? ncomp : ngen;
                                    st_word[no] = nwords;
                                    st_avg[no] = avg;
                                }
                                if (complete) {
                                    ncomp++;
                                } else {
                                    ngen++;
                                    nwords += (uint32_t)size;
                                }
                            }
                            no++;
                        }
                        if (lane == 0) {
                            unsigned long long bc = 0ull, bg = 0ull, bw = 0ull;
                            if (ok) {
                                if (ncomp) bc = atomicAdd(hand_counters + kHandComplete, (unsigned long long)ncomp);
                                if (ngen || nwords) {
                                    const unsigned long long gm = atomicAdd(hand_counters + kHandMembers, ((unsigned long long)ngen << 32) | (unsigned long long)nwords);
                                    bg = hand_member_descs(gm);
                                    bw = hand_member_words(gm);
                                }
                                if (bc + ncomp > (unsigned long long)desc_cap || bg + ngen > (unsigned long long)desc_cap ||
                                    bw + nwords > (unsigned long long)word_cap) {
                                    // (cannot happen: the host sizes the lists for Pout persons and Kc members of every frame)
                                    // -- void the reserved descriptors that exist and keep the frame here
                                    for (unsigned long long i = bc; i < bc + ncomp && i < (unsigned long long)desc_cap; i++)
                                        desc[i] = ClusterDesc{0u, 0u, 0xffffffffu, 0u};
                                    for (unsigned long long i = bg; i < bg + ngen && i < (unsigned long long)desc_cap; i++)
                                        desc[(unsigned long long)desc_cap + i] = ClusterDesc{0u, 0u, 0xffffffffu, 0u};
                                    ok = 0;
                                }
                            }
                            misc[7] = ok;
                            misc[8] = no;
                            misc[9] = (int32_t)(uint32_t)bc;
                            misc[10] = (int32_t)(uint32_t)bg;
                            misc[11] = (int32_t)(uint32_t)bw;
                        }
                    }
                    __syncthreads();
                    if (misc[7]) {
                        handed = true;
                        nout = misc[8];
                        const uint32_t bc = (uint32_t)misc[9], bg = (uint32_t)misc[10], bw = (uint32_t)misc[11];
                        const int nsl = nout < Pout ? nout : Pout;
                        for (int sl = tid; sl < nsl; sl += kBlock) {
                            if (st_size[sl] == 0)
                                desc[bc + st_idx[sl]] = ClusterDesc{(uint32_t)f, st_a[sl], (uint32_t)sl, 0u};
                            else
                                desc[desc_cap + bg + st_idx[sl]] = ClusterDesc{(uint32_t)f, bw + st_word[sl], (uint32_t)sl, (uint32_t)st_size[sl]};
                            wr.person(f, Pout, sl, st_avg[sl]);
                        }
                        for (int sl = 0; sl < nsl; sl++) {
                            const int size = st_size[sl], m0 = (int)st_a[sl];
                            for (int i = tid; i < size; i += kBlock) hand_words[bw + st_word[sl] + (uint32_t)i] = l_word[m0 + i];
                        }
                    }
                    __syncthreads();   // the staging area is the ray chunk of the next frame (or of phase 3 below)
                }
            }
            if (!handed) {
            uint32_t *cmem = reinterpret_cast<uint32_t *>(centre);    // [n]
            int32_t *cslot = reinterpret_cast<int32_t *>(cmem + Kc);  // [ncl] preliminary output slot or -1
            int32_t *cid_of_slot = cslot + Kc;                        // [nsurv] its inverse
            const int nmem = cstart[ncl];   // members of all clusters (<= n: the last candidate never seeds, :107)
            const bool mem_in_lds = nmem <= lmem_cap;
            for (int pos = tid; pos < nmem; pos += kBlock) {
                const uint32_t w = cw[kidx[members[pos]]];
                if (mem_in_lds)
                    lmem_lds[pos] = w;
                else
                    cmem[pos] = w;
            }
            // (two explicit address spaces: a generic pointer costs a flat load and 64-bit address arithmetic per member)
            auto member_word = [&](int pos) -> uint32_t { return mem_in_lds ? lmem_lds[pos] : cmem[pos]; };
            if (tid < 64) {
                int ns = 0;
                for (int base = 0; base < ncl; base += 64) {
                    const int c = base + lane;
                    const bool in = c < ncl && !((double)csize[c] < prm.num_tol);                  // :132-134
                    const unsigned long long m = __ballot(in);
                    const int sl = ns + __popcll(m & ((1ull << lane) - 1ull));
                    if (c < ncl) cslot[c] = in ? sl : -1;
                    if (in) cid_of_slot[sl] = c;
                    ns += __popcll(m);
                }
                if (lane == 0) misc[5] = ns;
            }
            __syncthreads();
            // pass 0 writes to the preliminary slots; if the mean-score filter then drops a cluster, the later
            // ones belong further down (and one that did not fit Pout_max may now fit): pass 1 repeats the
            // sweep with the final slots.  The filter never fires with the shipped thresholds.
            for (int pass = 0; pass < 2; pass++) {
            const int nsurv = misc[5];
            for (int sbase = 0; sbase < nsurv; sbase += kRecomputeSlotTile) {
                for (int j0 = 0; j0 < kn; j0 += Jc) {
                    const int nj = (kn - j0) < Jc ? (kn - j0) : Jc;
                    for (int i = tid; i < R * nj; i += kBlock) {
                        const int r = i / nj, jj = i - r * nj;
                        const int c = r / Pmax, p = r - c * Pmax;
                        const int nc = np_f ? np_f[c] : Pmax;
                        if (p < nc) {
                            const Kp3<TIn> kp = kpf[(size_t)r * J + j0 + jj];
                            *reinterpret_cast<RayRec *>(rays + r * rstride + 32 * jj) = make_ray(Ml + 9 * c, kp.u, kp.v);
                            rsc[r * sstride + jj] = kp.s;
                        }
                    }
                    __syncthreads();
                    // thread = (joint jj of the chunk, cluster of the sweep, member group g): the L = 256 / nj lanes of a
                    // joint are shared by up to L clusters at once, G = L / clusters (power of two) lanes each; a lane
                    // walks members g, g + G, ... of ITS cluster, the G partial sums sit in adjacent lanes and are added
                    // with shuffles.  (One cluster after the other, 64 lanes each, left every lane with 3-4 solves
                    // between a 5-step reduction and the next cluster's set-up: 3x the time of the solves themselves.)
                    int L = 1;                                    // largest power of two with nj * L <= 256, within one wave
                    while (2 * L * nj <= kBlock && L < 64) L *= 2;
                    const int nsw = (nsurv - sbase) < kRecomputeSlotTile ? (nsurv - sbase) : kRecomputeSlotTile;  // clusters of this sweep
                    int CB = 1;                                   // clusters side by side: power of two <= L covering nsw if it can
                    while (CB < nsw && CB < L) CB *= 2;
                    const int G = L / CB;
                    const int jj = tid / L, li = tid & (L - 1), ci = li / G, g = li & (G - 1);
                    const int jc = jj < nj ? jj : 0;
                    for (int cb = 0; cb < nsw; cb += CB) {
                        const bool active = jj < nj && cb + ci < nsw;
                        const int slot = sbase + cb + ci;
                        const int cid = active ? cid_of_slot[slot] : 0;
                        const int size = active ? csize[cid] : 0, m0 = active ? cstart[cid] : 0;
                        double aS = 0.0, aX = 0.0, aY = 0.0, aZ = 0.0;
                        if constexpr (sizeof(TOut) == 4) {
                            // float32 outputs, TWO members per iteration: the member loop is a chain of dependent LDS reads
                            // (member word -> pair constants and two ray rows at per-lane addresses) and of a dependent
                            // solve; two independent chains in flight hide half of that latency, and the pair shares one
                            // reciprocal.  1/dist is the raw v_rsq_f64 (measured 2^-24.2 relative, below the float32
                            // rounding of the stored score: same contract as k_fused_lean); sq = 2000 x the score of :72.
                            struct Half {
                                RayRec a, b;
                                Vec3 d, tsum;
                                TIn sm, ss;
                                double bq, e, g, det;
                            };
                            auto front = [&](uint32_t w) {
                                Half h;
                                const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                                const double *pc = pairc + 6 * q;
                                h.d = {pc[0], pc[1], pc[2]};
                                h.tsum = {pc[3], pc[4], pc[5]};
                                h.a = *reinterpret_cast<const RayRec *>(rays + rm * rstride + 32 * jc);
                                h.b = *reinterpret_cast<const RayRec *>(rays + rs * rstride + 32 * jc);
                                h.sm = rsc[rm * sstride + jc];
                                h.ss = rsc[rs * sstride + jc];
                                h.bq = fma(h.a.z, h.b.z, fma(h.a.y, h.b.y, h.a.x * h.b.x));
                                h.e = fma(h.a.z, h.d.z, fma(h.a.y, h.d.y, h.a.x * h.d.x));
                                h.g = fma(h.b.z, h.d.z, fma(h.b.y, h.d.y, h.b.x * h.d.x));
                                h.det = fma(h.a.a, h.b.a, -(h.bq * h.bq));
                                return h;
                            };
                            auto back = [&](const Half &h, double inv, bool use) {
                                const double S0 = fma(h.b.a, h.e, -(h.bq * h.g)) * inv;
                                const double S1 = fma(h.a.a, h.g, -(h.bq * h.e)) * inv;
                                const double fx = fma(h.b.x, S1, fma(h.a.x, S0, -h.d.x)), fy = fma(h.b.y, S1, fma(h.a.y, S0, -h.d.y)),
                                             fz = fma(h.b.z, S1, fma(h.a.z, S0, -h.d.z));
                                const double d2 = fma(fz, fz, fma(fy, fy, fx * fx));
                                const bool kp_ = use && !below_kthr(h.sm, prm) && !below_kthr(h.ss, prm) && !(d2 > prm.dthr2);
                                // the gate ASSIGNS 0 (:73-74): select after the product (0 * inf at an exact intersection)
                                const double sq = kp_ ? sum_score(h.sm, h.ss) * __builtin_amdgcn_rsq(d2) : 0.0;
                                aS += sq;                                                          // :141
                                aX = fma(sq, fma(-h.b.x, S1, fma(h.a.x, S0, h.tsum.x)), aX);       // :144-147
                                aY = fma(sq, fma(-h.b.y, S1, fma(h.a.y, S0, h.tsum.y)), aY);
                                aZ = fma(sq, fma(-h.b.z, S1, fma(h.a.z, S0, h.tsum.z)), aZ);
                            };
                            uint32_t w0 = g < size ? member_word(m0 + g) : 0u;
                            uint32_t w1 = g + G < size ? member_word(m0 + g + G) : w0;
                            for (int mi = g; mi < size; mi += 2 * G) {
                                const bool two = mi + G < size;
                                const uint32_t n0 = (mi + 2 * G < size) ? member_word(m0 + mi + 2 * G) : 0u;   // in flight during the solves
                                const uint32_t n1 = (mi + 3 * G < size) ? member_word(m0 + mi + 3 * G) : n0;
                                const Half h0 = front(w0), h1 = front(w1);   // (w1 repeats w0 when the cluster has no second member left)
                                // one reciprocal for the two members -- unless their determinants' product is not a normal number
                                // (a NaN ray, a singular or wildly conditioned pair): the shared reciprocal would hand the one
                                // member's NaN to the OTHER, healthy member (found by the round-5 sweep: a NaN pixel whose
                                // confidence is gated poisoned its partner's score); then each member takes its own
                                const double prod = h0.det * h1.det;
                                double i0, i1;
                                if (fabs(prod) > 1e-250 && fabs(prod) < 1e250) {
                                    const double run = rcp_nr2(prod);
                                    i0 = run * h1.det;
                                    i1 = run * h0.det;
                                } else {
                                    i0 = rcp_nr2(h0.det);
                                    i1 = rcp_nr2(h1.det);
                                }
                                back(h0, i0, true);
                                back(h1, i1, two);
                                w0 = n0;
                                w1 = n1;
                            }
                        }
                        uint32_t mw = (sizeof(TOut) != 4 && g < size) ? member_word(m0 + g) : 0u;
                        for (int mi = g; sizeof(TOut) != 4 && mi < size; mi += G) {
                            const uint32_t mw_next = (mi + G < size) ? member_word(m0 + mi + G) : 0u;   // in flight during this solve
                            const int rm = (int)(mw & 1023u), rs = (int)((mw >> 10) & 1023u), q = (int)(mw >> 20);
                            mw = mw_next;
                            const double *pc = pairc + 6 * q;
                            const Vec3 d = {pc[0], pc[1], pc[2]}, tsum = {pc[3], pc[4], pc[5]};
                            const RayRec a = *reinterpret_cast<const RayRec *>(rays + rm * rstride + 32 * jc);
                            const RayRec b = *reinterpret_cast<const RayRec *>(rays + rs * rstride + 32 * jc);
                            const TIn sm = rsc[rm * sstride + jc], ss = rsc[rs * sstride + jc];
                            double sq;
                            Vec3 sw;
                            if constexpr (sizeof(TOut) == 4) {
                                // float32 outputs: 1/dist is the raw v_rsq_f64 (measured 2^-24.2 relative, below the float32 rounding
                                // of the stored score: same contract as k_fused_lean); sq = 2000 x the score of :72
                                const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
                                const double e = fma(a.z, d.z, fma(a.y, d.y, a.x * d.x));
                                const double g_ = fma(b.z, d.z, fma(b.y, d.y, b.x * d.x));
                                const double inv = rcp_nr2(fma(a.a, b.a, -(bq * bq)));
                                const double S0 = fma(b.a, e, -(bq * g_)) * inv;
                                const double S1 = fma(a.a, g_, -(bq * e)) * inv;
                                const double fx = fma(b.x, S1, fma(a.x, S0, -d.x)), fy = fma(b.y, S1, fma(a.y, S0, -d.y)),
                                             fz = fma(b.z, S1, fma(a.z, S0, -d.z));
                                const double d2 = fma(fz, fz, fma(fy, fy, fx * fx));
                                const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(d2 > prm.dthr2);
                                // the gate ASSIGNS 0 (:73-74): select after the product (0 * inf at an exact intersection)
                                sq = kp_ ? sum_score(sm, ss) * __builtin_amdgcn_rsq(d2) : 0.0;
                                sw = {fma(-b.x, S1, fma(a.x, S0, tsum.x)), fma(-b.y, S1, fma(a.y, S0, tsum.y)),
                                      fma(-b.z, S1, fma(a.z, S0, tsum.z))};
                            } else {
                                const PairSolve o = pair_solve_fast<true>(a, b, d, tsum);
                                const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(o.d2 > prm.dthr2);
                                sq = kp_ ? sum_score(sm, ss) * (0.5 * o.score_base) : 0.0;             // gate assigns 0
                                sw = o.sw;
                            }
                            aS += sq;                                                              // :141
                            aX = fma(sq, sw.x, aX);                                                // :144-147
                            aY = fma(sq, sw.y, aY);
                            aZ = fma(sq, sw.z, aZ);
                        }
                        for (int off = G >> 1; off > 0; off >>= 1) {   // the G lanes of a (joint, cluster) are adjacent
                            aS += __shfl_xor(aS, off, 64);
                            aX += __shfl_xor(aX, off, 64);
                            aY += __shfl_xor(aY, off, 64);
                            aZ += __shfl_xor(aZ, off, 64);
                        }
                        if (active && g == 0) {
                            double ox = 0.0, oy = 0.0, oz = 0.0, os = 0.0;
                            if (!(aS == 0.0)) {                                                    // :142-143
                                if constexpr (sizeof(TOut) == 4) {
                                    const double r = 0.5 * rcp_nr2(aS);   // (an IEEE divide is ~30 instructions for the whole wave)
                                    ox = aX * r;
                                    oy = aY * r;
                                    oz = aZ * r;
                                    os = aS * (0.0005 * rcp_nr2((double)size));   // :148; this branch sums 2000 x the score
                                } else {
                                    const double r = 0.5 / aS;
                                    ox = aX * r;
                                    oy = aY * r;
                                    oz = aZ * r;
                                    os = aS / (double)size;                                        // :148
                                }
                            }
                            if (slot < Pout) wr.joint(f, Pout, kn, slot, j0 + jj, ox, oy, oz, os);
                            osbuf[(size_t)(slot - sbase) * kn + j0 + jj] = os;
                        }
                    }
                    __syncthreads();
                }
                // mean fused score of every person of this sweep (:150): one wave per person, fixed order
                for (int cid = 0; cid < ncl; cid++) {
                    const int slot = cslot[cid];
                    if (slot < sbase || slot >= sbase + kRecomputeSlotTile || ((slot - sbase) & (kBlock / 64 - 1)) != (tid >> 6))
                        continue;
                    double v = 0.0;
                    for (int b = lane; b < kn; b += 64) v += osbuf[(size_t)(slot - sbase) * kn + b];
                    v = wave_sum(v);
                    if (lane == 0) sum[cid] = v;
                }
                __syncthreads();
            }
            // mean-score filter (:150-152) and final slots
            if (tid < 64) {
                int nf_ = 0, moved = 0;
                for (int base = 0; base < ncl; base += 64) {
                    const int c = base + lane;
                    const int slot = c < ncl ? cslot[c] : -1;
                    const double avg = slot >= 0 ? sum[c] / (double)kn : 0.0;
                    const bool in = slot >= 0 && !(avg < prm.score_tol);
                    const unsigned long long m = __ballot(in);
                    const int fin = nf_ + __popcll(m & ((1ull << lane) - 1ull));
                    if (in && fin < Pout) wr.person(f, Pout, fin, avg);
                    moved |= __ballot(in && fin != slot) != 0ull;
                    moved |= __ballot(slot >= 0 && !in) != 0ull;
                    if (c < ncl) cslot[c] = in ? fin : -1;           // the slots of pass 1, if there is one
                    if (in) cid_of_slot[fin] = c;
                    nf_ += __popcll(m);
                }
                if (lane == 0) {
                    misc[5] = nf_;
                    misc[6] = moved;
                }
            }
            __syncthreads();
            nout = misc[5];
            if (!misc[6]) break;
            __syncthreads();   // misc[5] (= survivors of pass 1) is read at the top of the loop
            }
            }   // !handed
        }
        // unused slots: one flat sweep of 16-byte stores (unless the call opted out: SNOWTRI_CALL_NO_ZERO_FILL)
        if (!prm.no_zero_fill) {
            for (int i = tid; i < (Pout - nout) * kn; i += kBlock) wr.zero_joint(f, Pout, kn, nout + i / kn, i % kn);
            for (int slot = nout + tid; slot < Pout; slot += kBlock) wr.person(f, Pout, slot, 0.0);
        }
        if (tid == 0) {
            out_count[f] = nout;
            if (out_flags && nout > Pout) atomicOr(&out_flags[f], 2u /*SNOWTRI_FLAG_OVERFLOW*/);
        }
        __syncthreads();
    }
}

}  // namespace snowtri

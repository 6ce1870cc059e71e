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

// LDS layout of one joint chunk (Jc joints): rays[R][Jc] with a row stride of 32 Jc + 16 bytes and scores[R][Jc | 1]:
// a lane walks the joints of ITS candidate's two rows with compile-time offsets (no address arithmetic in the
// solve loop), and the odd strides (in 16-byte / 4-byte units) spread the rows of neighbouring lanes over the banks.
// Joints per chunk: what fits the budget, rounded DOWN to a power of two -- phase 3 maps thread = (joint of the chunk,
// member group of G lanes) with G a power of two and joints x G <= 256, so only a power-of-two chunk fills the workgroup
// (42 joints x 4 lanes = 168 of 256 threads at 8 cameras x 4 persons; 32 x 8 = 256).
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
// LDS of k_frame_recompute: [ray chunk: kRayChunkBytes][reduction slots + flags: 256 B][pair table: 56 B per pair]
// [member words of phase 3: the rest].
// The launcher asks for at least 52 KB (three workgroups per CU share 160 KB).
__host__ __device__ inline size_t recompute_lds_bytes(int R, int J, int kn, int score_bytes, int npairs) {
    (void)J; (void)kn; (void)score_bytes;
    (void)R;
    return (size_t)kRayChunkBytes + 256 + recompute_pairtab_bytes(npairs) + 16;
}

// ---- phase-1 work of ONE candidate on one joint chunk: sum over the chunk's joints of 2000 x the pair score
// (triangulation.py:70-78 without the 1/2000 of :72, applied by the caller).
//   EXACT = true   the arithmetic the kernel has always used: 1/dist by v_rsq_f64 + one Newton step (2e-14), exact
//                  intersection -> inf, four joints share one reciprocal unless their determinants' product leaves the
//                  normal range (singular pairs are detected there).
//   EXACT = false  the same solves with the raw v_rsq_f64 (measured 5.2e-8 = 2^-24.2 relative, tests pin <= 2^-23) and no range check -- 36 instead of 57 VALU
//                  instructions per solve.  Its sum decides only whether the candidate is KEPT (:79-81; the scores that
//                  are output come from phase 3): the caller re-does a candidate with EXACT = true when its mean is not
//                  finite (singular pair, exact intersection, NaN input) or lies within 1e-6 relative of
//                  average_score_threshold, so the decision is always taken on the accurate sum.  The gates (:73-74)
//                  do not involve 1/dist and d2 is computed identically in both variants: a sum of exactly 0 is exact.
template <bool EXACT, typename TIn>
__device__ __forceinline__ double candidate_chunk_sum(const RayRec *__restrict__ ra, const RayRec *__restrict__ rb,
                                                      const TIn *__restrict__ sa, const TIn *__restrict__ sb, int nj,
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
        if constexpr (EXACT) {
            double idist = rsq_nr1(d2);
            idist = (d2 == 0.0) ? __builtin_inf() : idist;
            // the gate ASSIGNS 0 (:73-74): select after the product, 0 * inf (exact intersection) would be NaN
            acc += kp_ ? sum_score(sm, ss) * idist : 0.0;                                     // :72
        } else {
            // 0 * inf / 0 * NaN leave NaN in the sum: the caller then re-does the candidate exactly
            acc = fma(gated_sum(sm, ss, kp_), __builtin_amdgcn_rsq(d2), acc);
        }
    };
    int jj = 0;
    // four joints at a time share ONE reciprocal (Montgomery): v_rcp_f64 issues at quarter rate.
    for (; jj + 4 <= nj; jj += 4) {
        RayRec a[4], b[4];
        TIn sm[4], ss[4];
        double bq[4], e[4], g[4], det[4], inv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = ra[jj + u];
            b[u] = rb[jj + u];
            sm[u] = sa[jj + u];
            ss[u] = sb[jj + u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bq[u] = fma(a[u].z, b[u].z, fma(a[u].y, b[u].y, a[u].x * b[u].x));
            e[u] = fma(a[u].z, d.z, fma(a[u].y, d.y, a[u].x * d.x));
            g[u] = fma(b[u].z, d.z, fma(b[u].y, d.y, b[u].x * d.x));
            det[u] = fma(a[u].a, b[u].a, -(bq[u] * bq[u]));
        }
        const double p01 = det[0] * det[1], p012 = p01 * det[2], p0123 = p012 * det[3];
        // EXACT: a product outside the normal range (a singular or wildly conditioned pair in the group) falls back to
        // four separate reciprocals; the fast variant lets it poison the sum instead
        if (!EXACT || (fabs(p0123) > 1e-250 && fabs(p0123) < 1e250)) {
            double run = rcp_nr2(p0123);
            inv[3] = run * p012;
            run *= det[3];
            inv[2] = run * p01;
            run *= det[2];
            inv[1] = run * det[0];
            inv[0] = run * det[1];
        } else {   // includes every exactly singular pair (product 0): flagged here, off the common path
#pragma unroll
            for (int u = 0; u < 4; u++) {
                inv[u] = rcp_nr2(det[u]);
                sing |= det[u] == 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) finish(a[u], b[u], bq[u], e[u], g[u], inv[u], sm[u], ss[u]);
    }
    for (; jj < nj; jj++) {
        const RayRec a = ra[jj], b = rb[jj];
        const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
        const double e = fma(a.z, d.z, fma(a.y, d.y, a.x * d.x));
        const double g = fma(b.z, d.z, fma(b.y, d.y, b.x * d.x));
        const double det = fma(a.a, b.a, -(bq * bq));
        if (EXACT) sing |= det == 0.0;
        finish(a, b, bq, e, g, rcp_nr2(det), sa[jj], sb[jj]);
    }
    return acc;
}

// ---- the fast phase-1 arithmetic for GS candidates that share their FIRST ray: one person of camera m against GS
// persons of camera s.  The first ray, its score and e = h_m . d are read / computed once per joint instead of once per
// candidate (LDS reads per solve 6 -> 3.75 at GS = 4), and the GS solves of a joint share one reciprocal, so any joint
// count works without a tail.  The lane walks joints jsub, jsub + JS, ...; acc[u] += 2000 x score of candidate u.
// Same solves, gates and raw 1/sqrt as candidate_chunk_sum<false>: the caller's re-do rule applies unchanged.
template <int GS, typename TIn>
__device__ __forceinline__ void rowgroup_chunk_sums(const RayRec *__restrict__ ra, const TIn *__restrict__ sa,
                                                    const RayRec *const (&rb)[GS], const TIn *const (&sb)[GS], int jsub, int JS,
                                                    int nj, const Vec3 &d, const Params &prm, double (&acc)[GS]) {
    for (int jj = jsub; jj < nj; jj += JS) {
        const RayRec a = ra[jj];
        const TIn sm = sa[jj];
        const double e = fma(a.z, d.z, fma(a.y, d.y, a.x * d.x));
        const bool okm = !below_kthr(sm, prm);
        RayRec b[GS];
        TIn ss[GS];
        double bq[GS], g[GS], det[GS], inv[GS];
#pragma unroll
        for (int u = 0; u < GS; u++) {
            b[u] = rb[u][jj];
            ss[u] = sb[u][jj];
        }
#pragma unroll
        for (int u = 0; u < GS; u++) {
            bq[u] = fma(a.z, b[u].z, fma(a.y, b[u].y, a.x * b[u].x));
            g[u] = fma(b[u].z, d.z, fma(b[u].y, d.y, b[u].x * d.x));
            det[u] = fma(a.a, b[u].a, -(bq[u] * bq[u]));
        }
        // one reciprocal for the GS determinants; a singular pair poisons the group's sums (NaN): re-done exactly
        if constexpr (GS == 4) {
            const double p01 = det[0] * det[1], p012 = p01 * det[2];
            double run = rcp_nr2(p012 * det[3]);
            inv[3] = run * p012;
            run *= det[3];
            inv[2] = run * p01;
            run *= det[2];
            inv[1] = run * det[0];
            inv[0] = run * det[1];
        } else {
            static_assert(GS == 2, "groups of two or four persons");
            const double run = rcp_nr2(det[0] * det[1]);
            inv[0] = run * det[1];
            inv[1] = run * det[0];
        }
#pragma unroll
        for (int u = 0; u < GS; u++) {
            const double S0 = fma(b[u].a, e, -(bq[u] * g[u])) * inv[u];
            const double S1 = fma(a.a, g[u], -(bq[u] * e)) * inv[u];
            const double fx = fma(b[u].x, S1, fma(a.x, S0, -d.x)), fy = fma(b[u].y, S1, fma(a.y, S0, -d.y)),
                         fz = fma(b[u].z, S1, fma(a.z, S0, -d.z));
            const double d2 = fma(fz, fz, fma(fy, fy, fx * fx));
            const bool kp_ = okm && !below_kthr(ss[u], prm) && !(d2 > prm.dthr2);   // :73-74
            acc[u] = fma(gated_sum(sm, ss[u], kp_), __builtin_amdgcn_rsq(d2), acc[u]);
        }
    }
}

// Dynamic LDS = recompute_lds_bytes(R, J, kn, sizeof(TIn)); scratch = gridDim.x slabs of
// recompute_scratch_bytes(Kc, R, kn).  R = C * Pmax ray rows.  Requires keypoint_num <= kRecomputeMaxKn.
#ifndef SNOWTRI_RECOMPUTE_WAVES
#define SNOWTRI_RECOMPUTE_WAVES 3
#endif
#ifndef SNOWTRI_RECOMPUTE_UNROLL
#define SNOWTRI_RECOMPUTE_UNROLL 4
#endif
constexpr int kRecomputeUnroll = SNOWTRI_RECOMPUTE_UNROLL;
// Frames are handed out through an atomic counter (next_frame, zeroed by the host before the launch): the
// time of a frame depends on how many candidates survive, so a static frame->workgroup map leaves CUs idle.
//
// METHOD = 1 (SNOWTRI_DLT with several detections per camera -- row N3; not reference behaviour): phases 1-2
// are the reference's association unchanged; phase 3 instead solves, per surviving cluster and joint, the
// N-view DLT over the DISTINCT (camera, person) observations its member candidates are made of, keeping those
// whose confidence is not below keypoint_score_threshold (>= 2 needed, else the joint stays (0,0,0)/0);
// joint score = their mean confidence.  One lane per (cluster, joint); no cross-lane reduction.
template <int METHOD, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, SNOWTRI_RECOMPUTE_WAVES) void k_frame_recompute(int64_t F, int Pmax, int J, int Kc, Rig rig,
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
                                                            uint32_t word_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax;
    const int kn = prm.kn, ci = prm.center;
    const int Jc = recompute_chunk_joints(R, J, (int)sizeof(TIn));
    const int rstride = (int)recompute_ray_stride(Jc);          // bytes between the ray rows (32-bit LDS offsets)
    const int sstride = recompute_score_stride(Jc);             // elements between the score rows
    char *rays = smem;                                          // [R] rows of Jc RayRec (+16 B pad)
    TIn *rsc = reinterpret_cast<TIn *>(smem + (size_t)R * rstride);                      // [R][sstride]
    double *red = reinterpret_cast<double *>(smem + kRayChunkBytes);  // [4] + misc, at a fixed offset behind the chunk
    int32_t *misc = reinterpret_cast<int32_t *>(red + kBlock / 64);
    // camera-pair constants in LDS: d = t_s - t_m, t_m + t_s (6 doubles) and the two camera indices -- the solve loops
    // index them per lane (host-checked: npairs <= kPairTabMaxPairs, i.e. <= 16 cameras; larger rigs take k_frame_general)
    double *pairc = reinterpret_cast<double *>(reinterpret_cast<char *>(red) + 256);
    int32_t *pairs = reinterpret_cast<int32_t *>(pairc + 6 * rig.npairs);
    for (int i = tid; i < 6 * rig.npairs; i += kBlock) pairc[i] = rig.pairc[i];
    for (int i = tid; i < 2 * rig.npairs; i += kBlock) pairs[i] = rig.pairs[i];
    // what is left of the workgroup's LDS allocation holds the member words of phase 3
    uint32_t *lmem_lds = reinterpret_cast<uint32_t *>(pairs + 2 * rig.npairs);
    const int lmem_cap = (int)((lds_total - (int)(reinterpret_cast<char *>(lmem_lds) - smem)) / 4);
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
        const int64_t f = (int64_t)(((unsigned long long)(uint32_t)misc[3] << 32) | (uint32_t)misc[2]);
        if (f >= F) break;
        const int32_t *np_f = n_persons ? n_persons + f * C : nullptr;
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * J;
        bool ragged = false;
        for (int k = tid; k < Kc; k += kBlock) {
            sum[k] = 0.0;
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
        // every camera lists Pmax persons (and Pmax is even): phase 1 runs on groups of candidates that share their
        // first ray (rowgroup_chunk_sums); a ragged frame keeps one lane per candidate
        const int GS = (Pmax & 3) == 0 ? 4 : ((Pmax & 1) == 0 ? 2 : 0);
        const bool grouped = !exact_only && GS != 0 && !__syncthreads_or(ragged ? 1 : 0);

        // ---------------- phase 1: candidate score sums, joint chunk by joint chunk -------------
        for (int j0 = 0; j0 < J; j0 += Jc) {
            const int nj = (J - j0) < Jc ? (J - j0) : Jc;
            __syncthreads();
            for (int i = tid; i < R * nj; i += kBlock) {
                const int r = i / nj, jj = i - r * nj;
                const int c = r / Pmax, p = r - c * Pmax;
                const int nc = np_f ? np_f[c] : Pmax;
                if (p < nc) {
                    const Kp3<TIn> kp = kpf[(size_t)r * J + j0 + jj];
                    *reinterpret_cast<RayRec *>(rays + r * rstride + 32 * jj) = make_ray(rig.M + 9 * c, kp.u, kp.v);
                    rsc[r * sstride + jj] = kp.s;
                }
            }
            __syncthreads();
            if (grouped) {
                // item = (camera pair q, person pm of its first camera, group of GS persons of its second); JS lanes
                // share an item (interleaved joints) when there are fewer items than threads
                const int NG = Pmax / GS, per_q = Pmax * NG, nitems = rig.npairs * per_q;
                int JS = 1;
                while (JS < 8 && 2 * JS * nitems <= kBlock) JS *= 2;
                const int total = nitems * JS;
                for (int base = 0; base < total; base += kBlock) {   // same trip count for every lane: shuffles below
                    const int idx = base + tid;
                    const bool live = idx < total;
                    const int item = live ? idx / JS : 0, jsub = live ? idx - item * JS : 0;
                    const int q = item / per_q, r2 = item - q * per_q, pm = r2 / NG, ps0 = (r2 - pm * NG) * GS;
                    const int k0 = q * pp + pm * Pmax + ps0;
                    const int rm = pairs[2 * q] * Pmax + pm, rs0 = pairs[2 * q + 1] * Pmax + ps0;
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    const RayRec *ra = reinterpret_cast<const RayRec *>(rays + rm * rstride);
                    const TIn *sa = rsc + rm * sstride;
                    const int njl = live ? nj : 0;
                    if (GS == 4) {
                        const RayRec *const rb[4] = {reinterpret_cast<const RayRec *>(rays + (rs0 + 0) * rstride),
                                                     reinterpret_cast<const RayRec *>(rays + (rs0 + 1) * rstride),
                                                     reinterpret_cast<const RayRec *>(rays + (rs0 + 2) * rstride),
                                                     reinterpret_cast<const RayRec *>(rays + (rs0 + 3) * rstride)};
                        const TIn *const sb[4] = {rsc + (rs0 + 0) * sstride, rsc + (rs0 + 1) * sstride, rsc + (rs0 + 2) * sstride,
                                                  rsc + (rs0 + 3) * sstride};
                        double acc[4] = {0.0, 0.0, 0.0, 0.0};
                        rowgroup_chunk_sums<4>(ra, sa, rb, sb, jsub, JS, njl, d, prm, acc);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            double v = acc[u];
                            for (int off = JS >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                            if (live && jsub == 0) sum[k0 + u] += v * 0.0005;   // the 1 / (2 * 1000) of :72
                        }
                    } else {
                        const RayRec *const rb[2] = {reinterpret_cast<const RayRec *>(rays + (rs0 + 0) * rstride),
                                                     reinterpret_cast<const RayRec *>(rays + (rs0 + 1) * rstride)};
                        const TIn *const sb[2] = {rsc + (rs0 + 0) * sstride, rsc + (rs0 + 1) * sstride};
                        double acc[2] = {0.0, 0.0};
                        rowgroup_chunk_sums<2>(ra, sa, rb, sb, jsub, JS, njl, d, prm, acc);
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            double v = acc[u];
                            for (int off = JS >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                            if (live && jsub == 0) sum[k0 + u] += v * 0.0005;
                        }
                    }
                }
            } else {
                for (int k = tid; k < Kc; k += kBlock) {
                    const uint32_t w = cw[k];
                    if (w == kNoCand) continue;
                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    const RayRec *ra = reinterpret_cast<const RayRec *>(rays + rm * rstride);
                    const RayRec *rb = reinterpret_cast<const RayRec *>(rays + rs * rstride);
                    const TIn *sa = rsc + rm * sstride, *sb = rsc + rs * sstride;
                    const double acc = exact_only ? candidate_chunk_sum<true>(ra, rb, sa, sb, nj, d, prm, sing)
                                                  : candidate_chunk_sum<false>(ra, rb, sa, sb, nj, d, prm, sing);
                    sum[k] += acc * 0.0005;   // the 1 / (2 * 1000) of :72
                }
            }
        }
        // candidates whose fast sum cannot decide :80-81 (see candidate_chunk_sum): not finite, or within 1e-6 relative
        // of average_score_threshold (the fast sum is within 2.4e-7 of the accurate one).  Rare: a second sweep over the
        // joint chunks re-does just those with the accurate arithmetic; exactly singular pairs are flagged there.
        int redo_any = 0;
        if (!exact_only) {
            __syncthreads();
            for (int k = tid; k < Kc; k += kBlock) {
                const double s_ = sum[k], mean = s_ / (double)J;
                const bool redo = !(fabs(mean) < 1e300) || (s_ != 0.0 && fabs(mean - prm.avg_thr) <= 1e-6 * fabs(mean));
                keep[k] = redo ? 2 : 0;      // (invalid slots hold sum 0: never re-done)
                redo_any |= redo ? 1 : 0;
            }
            redo_any = __syncthreads_or(redo_any);
        }
        if (redo_any) {
            for (int k = tid; k < Kc; k += kBlock)
                if (keep[k] == 2) sum[k] = 0.0;
            for (int j0 = 0; j0 < J; j0 += Jc) {
                const int nj = (J - j0) < Jc ? (J - j0) : Jc;
                __syncthreads();
                for (int i = tid; i < R * nj; i += kBlock) {
                    const int r = i / nj, jj = i - r * nj;
                    const int c = r / Pmax, p = r - c * Pmax;
                    const int nc = np_f ? np_f[c] : Pmax;
                    if (p < nc) {
                        const Kp3<TIn> kp = kpf[(size_t)r * J + j0 + jj];
                        *reinterpret_cast<RayRec *>(rays + r * rstride + 32 * jj) = make_ray(rig.M + 9 * c, kp.u, kp.v);
                        rsc[r * sstride + jj] = kp.s;
                    }
                }
                __syncthreads();
                for (int k = tid; k < Kc; k += kBlock) {
                    if (keep[k] != 2) continue;
                    const uint32_t w = cw[k];
                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                    const double *pc = pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]};
                    sum[k] += candidate_chunk_sum<true>(reinterpret_cast<const RayRec *>(rays + rm * rstride),
                                                        reinterpret_cast<const RayRec *>(rays + rs * rstride), rsc + rm * sstride,
                                                        rsc + rs * sstride, nj, d, prm, sing) * 0.0005;
                }
            }
        }
        __syncthreads();
        if (sing && out_flags) atomicOr(&out_flags[f], 1u /*SNOWTRI_FLAG_SINGULAR*/);
        for (int k = tid; k < Kc; k += kBlock) {
            const bool valid = cw[k] != kNoCand;
            const double mean = sum[k] / (double)J;                                               // :79
            keep[k] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;                                   // :80-81
        }
        __syncthreads();

#ifdef SNOWTRI_REC_STOP_AFTER_P1  // dev experiment (timing only, outputs are wrong): phase 1 alone
        if (tid == 0) out_count[f] = 0;
        __syncthreads();
        continue;
#endif
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
        const size_t chunk_bytes = (size_t)kRayChunkBytes;
        const bool in_lds = (size_t)n * 28 + 16 <= chunk_bytes;
        auto phase2 = [&](int32_t *cof, double *cen) {
            for (int i = tid; i < n; i += kBlock) {
                const uint32_t w = cw[kidx[i]];
                const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
                const Kp3<TIn> km = kpf[(size_t)rm * J + ci], ks = kpf[(size_t)rs * J + ci];
                const RayRec a = make_ray(rig.M + 9 * mc, km.u, km.v), b = make_ray(rig.M + 9 * sc, ks.u, ks.v);
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
            phase2(reinterpret_cast<int32_t *>(smem + (((size_t)n * 24 + 15) & ~(size_t)15)), reinterpret_cast<double *>(smem));
        else
            phase2(cluster_of, centre);
        __syncthreads();
        const int ncl = misc[1];

#ifdef SNOWTRI_REC_STOP_AFTER_P2  // dev experiment (timing only, outputs are wrong): phases 1 + 2
        if (tid == 0) out_count[f] = 0;
        __syncthreads();
        continue;
#endif
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
                        const double r = 1.0 / e[3];
                        ox = e[0] * r;
                        oy = e[1] * r;
                        oz = e[2] * r;
                        os = ssum / (double)cnt;
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
                    uint32_t *st_a = reinterpret_cast<uint32_t *>(smem);            // [Pout] persons word | first member (cstart)
                    int32_t *st_size = reinterpret_cast<int32_t *>(smem + pq);      // [Pout] 0: complete graph, else members
                    uint32_t *st_idx = reinterpret_cast<uint32_t *>(smem + 2 * pq);  // [Pout] index inside its descriptor list
                    uint32_t *st_word = reinterpret_cast<uint32_t *>(smem + 3 * pq); // [Pout] offset of its member words
                    double *st_avg = reinterpret_cast<double *>(smem + 4 * pq);     // [Pout]
                    // the clusters' sizes / starts, member words and candidate score sums, fetched by the whole workgroup at
                    // once: the decisions below are then one wave walking LDS (from the L2-resident slab every step is a
                    // chain of dependent loads: 15 us per frame)
                    const int nmem = cstart[ncl];
                    double *l_sum = st_avg + Pout;                                   // [nmem]
                    uint32_t *l_word = reinterpret_cast<uint32_t *>(l_sum + nmem);   // [nmem]
                    int32_t *l_size = reinterpret_cast<int32_t *>(l_word + nmem);    // [ncl]
                    int32_t *l_start = l_size + ncl;                                 // [ncl]
                    const bool fits = 4 * pq + (size_t)Pout * 8 + (size_t)nmem * 12 + (size_t)ncl * 8 <= (size_t)kRayChunkBytes;
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
                            const double avg = wave_sum(ssum) / ((double)size * (double)J);               // :150 from :79
                            // (a sum that is not finite, or a mean within 1e-6 of the tolerance -- the fast sums of phase 1
                            // are within 6e-8 -- is left to phase 3)
                            if (!(fabs(avg) < 1e300) || fabs(avg - prm.score_tol) <= 1e-6 * fabs(avg)) {
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
                                if (ncomp) bc = atomicAdd(hand_counters, (unsigned long long)ncomp);
                                if (ngen) bg = atomicAdd(hand_counters + 1, (unsigned long long)ngen);
                                if (nwords) bw = atomicAdd(hand_counters + 2, (unsigned long long)nwords);
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
                            *reinterpret_cast<RayRec *>(rays + r * rstride + 32 * jj) = make_ray(rig.M + 9 * c, kp.u, kp.v);
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
#ifdef SNOWTRI_P3_NOSOLVE   // dev experiment (timing only, outputs are wrong): phase 3 without its member solves
                        const int size = 0, m0 = 0;
#else
                        const int size = active ? csize[cid] : 0, m0 = active ? cstart[cid] : 0;
#endif
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
                                const double run = rcp_nr2(h0.det * h1.det);
                                back(h0, run * h1.det, true);
                                back(h1, run * h0.det, two);
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
        // unused slots: one flat sweep of 16-byte stores
        for (int i = tid; i < (Pout - nout) * kn; i += kBlock) wr.zero_joint(f, Pout, kn, nout + i / kn, i % kn);
        for (int slot = nout + tid; slot < Pout; slot += kBlock) wr.person(f, Pout, slot, 0.0);
        if (tid == 0) {
            out_count[f] = nout;
            if (out_flags && nout > Pout) atomicOr(&out_flags[f], 2u /*SNOWTRI_FLAG_OVERFLOW*/);
        }
        __syncthreads();
    }
}

}  // namespace snowtri

// snowtri_cluster.hpp -- k_cluster_fuse: phase 3 of the multi-person path (triangulation.py:136-152) for the clusters
// k_frame_recompute could hand over, as a separate streaming kernel.
//
// After its association (phases 1-2) k_frame_recompute knows the clusters of a frame.  In a clean recording every
// person is seen by every camera and its cluster is the COMPLETE graph over one detection per camera: C(C,2) member
// candidates, one per camera pair, every camera with the same person in all of its pairs.  Such a cluster is exactly
// the input of the single-person item of k_fused_lean -- C rays, all pair solves, fusion regrouped per ray -- so
// instead of walking its members one by one from LDS-staged rays (64 VALU per member solve, a second sweep over the
// frame's ray chunks, barriers) the frame emits one 16-byte descriptor per output person and skips phase 3; this kernel
// then runs (descriptor, joint) items at the fast kernel's cost per pair.  Clusters of any other shape (ghost candidates
// that form a cluster of their own, a person one camera missed, two persons merged) take cluster_member_passes in the
// same launch, which walks the member list of the cluster per (descriptor, joint) lane -- the arithmetic of phase 3, without its ray
// staging.  Frames whose filter decisions are not safe on the fast arithmetic keep the in-kernel phase 3
// (snowtri_general.hpp).
//
// The descriptor carries everything that depends on the association: frame, output slot, the person index of every
// camera.  out_count / out_ps / flags / the zero-fill of unused slots are written by k_frame_recompute (the person's
// mean score is the mean of its members' candidate means, which phase 1 already has).  float32 outputs: 1/dist is the raw
// v_rsq_f64 as in k_fused_lean (same numerics contract); float64 outputs: Newton-refined, and the person's mean
// score is taken from the fused joints afterwards (k_person_scores), as it is for keypoint_num < J.
#pragma once
#include <type_traits>
#include <utility>

#include "snowtri_lean.hpp"   // (snowtri_item.hpp: cluster_item, through snowtri_fused.hpp)

namespace snowtri {

struct ClusterDesc {
    uint32_t frame;    // frame index inside the launch
    uint32_t persons;  // complete graph: 4 bits per camera, the person of camera c that belongs to this cluster;
                       // any other cluster: index of its first member word in the launch's member list
    uint32_t slot;     // output slot (< Pout); 0xffffffff = voided entry
    uint32_t size;     // 0 = complete graph, else the number of members
};
// The two list counters of a launch (zeroed by the host), on different 128-byte lines: every frame reserves its descriptors
// with ONE atomic per list (member descriptors and member words share a word).  (Measured on 8 x 4, wall-clock stamps of
// k_associate: a wave waits ~17 us for its reservation whether the counters share a line or not and whether it issues two
// atomics or three -- the round trip of a device-scope atomic under 4 096 waves, not the line.)
//   cnt[kHandComplete]: complete-graph descriptors;  cnt[kHandMembers]: member-list descriptors << 32 | member words
constexpr int kHandComplete = 0, kHandMembers = 16;
__host__ __device__ inline uint32_t hand_member_descs(unsigned long long v) { return (uint32_t)(v >> 32); }
__host__ __device__ inline uint32_t hand_member_words(unsigned long long v) { return (uint32_t)(v & 0xffffffffull); }
constexpr int kClusterMaxCams = 8;      // 4-bit person fields in one word, register-resident rays
constexpr int kClusterMaxPersons = 16;

__host__ __device__ constexpr int cluster_const_doubles(int C) { return 12 * C + 3 * (C * (C - 1) / 2); }  // M[C][9], t[C][3], d[NP][3]
__host__ __device__ constexpr size_t cluster_lds_bytes(int C) { return (size_t)8 * cluster_const_doubles(C) + 16; }

// One joint record [x, y, z, score] of an output person, rounded to the output type: one 16-byte store (two for doubles).
template <typename TOut>
__device__ __forceinline__ void cluster_store(TOut *__restrict__ out4, uint64_t rec, double x, double y, double z, double s) {
    if constexpr (sizeof(TOut) == 4) {
        reinterpret_cast<float4 *>(out4)[rec] = make_float4((float)x, (float)y, (float)z, (float)s);
    } else {
        double2 *o = reinterpret_cast<double2 *>(out4) + 2 * rec;
        o[0] = make_double2(x, y);
        o[1] = make_double2(z, s);
    }
}

// One member of a cluster at one joint, with the select semantics of phase 3 of k_frame_recompute: sq = 2000 x the pair
// score of :72 for float32 outputs, the score itself for float64 (the gates ASSIGN 0, :73-74: select after the product --
// 0 * inf at an exact intersection), sw = Wm + Ws.
// float32 outputs: raw v_rsq_f64 for 1/dist; float64 outputs: pair_solve_fast (Newton-refined, d2 == 0 -> inf).
// Returns true for an exactly singular pair (det == 0: np.linalg.inv raises, triangulation.py:26 -> SNOWTRI_FLAG_SINGULAR).
template <typename TIn, typename TOut>
__device__ __forceinline__ bool cluster_member_solve(const RayRec &a, const RayRec &b, const double *__restrict__ c6, TIn sm, TIn ss,
                                                     const Params &prm, double &sq, Vec3 &sw) {
    if constexpr (sizeof(TOut) == 4) {
        const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
        const double e = fma(a.z, c6[2], fma(a.y, c6[1], a.x * c6[0]));
        const double g = fma(b.z, c6[2], fma(b.y, c6[1], b.x * c6[0]));
        const double det = fma(a.a, b.a, -(bq * bq));
        const double inv = rcp_nr2(det);
        const double S0 = fma(b.a, e, -(bq * g)) * inv;
        const double S1 = fma(a.a, g, -(bq * e)) * inv;
        const double fx = fma(b.x, S1, fma(a.x, S0, -c6[0])), fy = fma(b.y, S1, fma(a.y, S0, -c6[1])),
                     fz = fma(b.z, S1, fma(a.z, S0, -c6[2]));
        const double d2 = fma(fz, fz, fma(fy, fy, fx * fx));
        const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(d2 > prm.dthr2);
        sq = kp_ ? sum_score(sm, ss) * __builtin_amdgcn_rsq(d2) : 0.0;
        sw = {fma(-b.x, S1, fma(a.x, S0, c6[3])), fma(-b.y, S1, fma(a.y, S0, c6[4])), fma(-b.z, S1, fma(a.z, S0, c6[5]))};
        return a.a * b.a == bq * bq;   // (see skew_ray_solve, snowtri_math.hpp)
    } else {
        const PairSolve o = pair_solve_fast<true>(a, b, Vec3{c6[0], c6[1], c6[2]}, Vec3{c6[3], c6[4], c6[5]});
        const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(o.d2 > prm.dthr2);
        sq = kp_ ? sum_score(sm, ss) * (0.5 * o.score_base) : 0.0;       // the score of :72 itself (k_frame_recompute's float64 branch)
        sw = o.sw;
        return o.singular;
    }
}

// sums over the members -> the joint (:141-148); aS = the sum of the members' scores (float32 outputs: 2000 x that)
template <typename TOut>
__device__ __forceinline__ void cluster_finish(double aS, double aX, double aY, double aZ, int size, double &x, double &y, double &z, double &s) {
    x = y = z = s = 0.0;
    if (!(aS == 0.0)) {                                                        // :142-143
        if constexpr (sizeof(TOut) == 4) {
            const double r = 0.5 * rcp_nr2(aS);
            x = aX * r;
            y = aY * r;
            z = aZ * r;
            s = aS * (0.0005 * rcp_nr2((double)size));                         // :148
        } else {
            const double r = 0.5 / aS;
            x = aX * r;
            y = aY * r;
            z = aZ * r;
            s = aS / (double)size;                                             // :148
        }
    }
}

// The same joint member by member, in the order and with the select semantics of phase 3 of k_frame_recompute: for the rare
// joints the fast items cannot finish (0 x inf at an exact intersection whose confidence is gated, inf / NaN sums, a
// singular pair).  plo / phi: the person of camera c in 4 bits (cameras 0-7 / 8-15).
template <typename TIn, typename TOut>
__device__ __noinline__ void cluster_joint_sequential(const Rig &rig, const Kp3<TIn> *__restrict__ kp3, int64_t frame0, uint32_t plo,
                                                      uint32_t phi, int Pmax, int J, int j, const Params &prm, double &ox, double &oy,
                                                      double &oz, double &os, uint32_t *__restrict__ out_flags) {
    const int C = rig.C, NP = rig.npairs;
    auto person = [&](int c) { return (int)(((c < 8 ? plo : phi) >> (4 * (c & 7))) & 15u); };
    double aS = 0.0, aX = 0.0, aY = 0.0, aZ = 0.0;
    bool sing = false;
    for (int q = 0; q < NP; q++) {
        const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
        const int64_t rm = (frame0 * C + mc) * Pmax + person(mc);
        const int64_t rs = (frame0 * C + sc) * Pmax + person(sc);
        const Kp3<TIn> km = kp3[rm * J + j], ks = kp3[rs * J + j];
        const RayRec a = make_ray(rig.M + 9 * mc, km.u, km.v), b = make_ray(rig.M + 9 * sc, ks.u, ks.v);
        double sq;
        Vec3 sw;
        sing |= cluster_member_solve<TIn, TOut>(a, b, rig.pairc + 6 * q, km.s, ks.s, prm, sq, sw);
        aS += sq;                                                              // :141
        aX = fma(sq, sw.x, aX);                                                // :144-147
        aY = fma(sq, sw.y, aY);
        aZ = fma(sq, sw.z, aZ);
    }
    if (sing && out_flags) atomicOr(&out_flags[frame0], 1u /*SNOWTRI_FLAG_SINGULAR*/);   // (idempotent beside the candidate pass's flag)
    cluster_finish<TOut>(aS, aX, aY, aZ, NP, ox, oy, oz, os);
}

// Clusters of any shape: lane = (descriptor, joint), one loop over the cluster's member words (rm | rs << 10 | q << 20,
// the candidate words of k_frame_recompute); arithmetic, order and select semantics of its phase 3 for float32 outputs.
// Called by every wave of k_cluster_fuse after its own passes (these clusters are few: ghost candidates, partly seen
// persons; a pass is a chain of dependent loads that hides behind the other waves' complete-graph items).
//   Ml [C][9] ray matrices, pc [NP][6] pair constants (d, t_m + t_s), pairs [NP][2] camera indices: in LDS.
template <typename TIn, typename TOut>
__device__ __forceinline__ void cluster_member_passes(const ClusterDesc *__restrict__ desc, uint32_t ndesc,
                                                      const uint32_t *__restrict__ words, const double *__restrict__ Ml,
                                                      const double *__restrict__ pc, const int32_t *__restrict__ pairs, int R,
                                                      const Kp3<TIn> *__restrict__ kp3, const Params &prm, int J, int kn,
                                                      unsigned long long kmagic, int Pout, TOut *__restrict__ out4, uint32_t p0, uint32_t W,
                                                      uint32_t *__restrict__ out_flags) {
    const int lane = threadIdx.x & 63;
    const uint32_t total = ndesc * (uint32_t)kn;
    const uint32_t npass = (total + 63u) >> 6;
    for (uint32_t p = p0; p < npass; p += W) {
        const uint32_t i = (p << 6) + (uint32_t)lane;
        bool valid = i < total;
        const uint32_t ic = valid ? i : 0u;
        const uint32_t di = (uint32_t)(((unsigned long long)ic * kmagic) >> 40);
        const uint32_t j = ic - di * (uint32_t)kn;
        uint4 d = make_uint4(0u, 0u, 0u, 0u);
        if (valid) d = *reinterpret_cast<const uint4 *>(desc + di);
        valid = valid && d.z < (uint32_t)Pout;
        const int size = valid ? (int)d.w : 0;
        const uint64_t row0 = (uint64_t)d.x * (uint32_t)R;
        double aS = 0.0, aX = 0.0, aY = 0.0, aZ = 0.0;
        bool sing = false;
        for (int m = 0; __ballot(m < size) != 0ull; m++) {
            if (m < size) {
                const uint32_t w = words[d.y + (uint32_t)m];
                const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                const Kp3<TIn> km = kp3[(row0 + (uint32_t)rm) * (uint32_t)J + j], ks = kp3[(row0 + (uint32_t)rs) * (uint32_t)J + j];
                const RayRec a = make_ray(Ml + 9 * pairs[2 * q], km.u, km.v), b = make_ray(Ml + 9 * pairs[2 * q + 1], ks.u, ks.v);
                double sq;
                Vec3 sw;
                sing |= cluster_member_solve<TIn, TOut>(a, b, pc + 6 * q, km.s, ks.s, prm, sq, sw);
                aS += sq;                                                              // :141
                aX = fma(sq, sw.x, aX);                                                // :144-147
                aY = fma(sq, sw.y, aY);
                aZ = fma(sq, sw.z, aZ);
            }
        }
        if (__ballot(sing)) {   // rare, wave-uniform branch: an exactly singular pair (the reference raises, :26)
            if (sing && valid && out_flags) atomicOr(&out_flags[d.x], 1u /*SNOWTRI_FLAG_SINGULAR*/);
        }
        if (valid) {
            double x, y, z, sc;
            cluster_finish<TOut>(aS, aX, aY, aZ, size, x, y, z, sc);
            cluster_store<TOut>(out4, ((uint64_t)d.x * (uint32_t)Pout + d.z) * (uint64_t)(uint32_t)kn + j, x, y, z, sc);
        }
    }
}

// Grid: any number of workgroups.  Wave gw takes 64-item passes gw, gw + W, ... of the ndesc x kn items
// (item = descriptor * kn + joint; joints >= keypoint_num are never fused, :136-149).  Three waves per SIMD hide the
// keypoint fetch of a pass behind the items of the other two (a register prefetch of the next pass cost the third wave:
// 182 -> 150 registers without it, 8 x 4: 344 -> 295 us).  The clusters of any other shape: k_cluster_members.
// cnt[kHandComplete]: descriptors in desc[0, cap).  Dynamic LDS: cluster_lds_bytes(C).
//   kmagic = ceil(2^40 / kn): item / kn = (item * kmagic) >> 40 for item < 2^31, kn <= 256.
constexpr int kClusterWaves = 3;
template <int C, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, kClusterWaves) void k_cluster_fuse(const ClusterDesc *__restrict__ desc,
                                                          const unsigned long long *__restrict__ cnt, uint32_t desc_cap,
                                                          Rig rig, const TIn *__restrict__ kpts, Params prm, int Pmax, int J, int kn,
                                                          unsigned long long kmagic, int Pout, TOut *__restrict__ out4,
                                                          uint32_t *__restrict__ out_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = C * (C - 1) / 2;
    double *K = reinterpret_cast<double *>(smem);   // [M | t | d] for cluster_item
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 9 * C; i += kBlock) K[i] = rig.M[i];
    if (tid < 3 * C) K[9 * C + tid] = rig.t[tid];
    if (tid < 3 * NP) K[12 * C + tid] = rig.pairc[6 * (tid / 3) + tid % 3];
    const unsigned long long nd64 = cnt[kHandComplete];
    const uint32_t ndesc = nd64 < (unsigned long long)desc_cap ? (uint32_t)nd64 : desc_cap;
    const uint32_t total = ndesc * (uint32_t)kn;
    const uint32_t npass = (total + 63u) >> 6;
    const uint32_t W = gridDim.x * (uint32_t)(kBlock / 64);
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const float kthr_f32 = prm.kthr_f32;
    const double kthr = prm.kthr, dthr2 = prm.dthr2;
    __syncthreads();

    for (uint32_t p = blockIdx.x * (uint32_t)(kBlock / 64) + (uint32_t)wave; p < npass; p += W) {
        const uint32_t i = (p << 6) + (uint32_t)lane;
        bool valid = i < total;
        const uint32_t ic = valid ? i : 0u;
        const uint32_t di = (uint32_t)(((unsigned long long)ic * kmagic) >> 40);
        const uint32_t j = ic - di * (uint32_t)kn;
        uint4 d = make_uint4(0u, 0u, 0u, 0u);
        if (valid) {
            d = *reinterpret_cast<const uint4 *>(desc + di);
            SNOWTRI_DEV_CHECK(di < ndesc && j < (uint32_t)kn && (d.z < (uint32_t)Pout || d.z == 0xffffffffu), 30);   // descriptor and joint inside their ranges
            valid = d.z < (uint32_t)Pout;   // (a voided entry, see the hand-over in k_frame_recompute)
        }
        const uint32_t frame = d.x, persons = d.y, slot = d.z;
        Kp3<TIn> cur[C];
        if (valid) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                SNOWTRI_DEV_CHECK(((persons >> (4 * c)) & 15u) < (uint32_t)Pmax, 31);   // person index of camera c
                const uint32_t row = (frame * (uint32_t)C + (uint32_t)c) * (uint32_t)Pmax + ((persons >> (4 * c)) & 15u);
                cur[c] = kp3[(uint64_t)row * (uint32_t)J + j];
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) cur[c] = Kp3<TIn>{(TIn)0, (TIn)0, (TIn)0};
        }
        double ox, oy, oz, os;
        asm volatile("" ::: "memory");   // the rig constants are re-read from LDS in every pass (hoisted out of the loop they would take 360 registers)
        const bool bad = cluster_item<C, TIn, TOut>(K, cur, kthr_f32, kthr, dthr2, ox, oy, oz, os);
        if (__ballot(bad && valid)) {   // rare, wave-uniform branch
            if (bad && valid) cluster_joint_sequential<TIn, TOut>(rig, kp3, (int64_t)frame, persons, 0u, Pmax, J, (int)j, prm, ox, oy, oz, os, out_flags);
        }
        if (valid) cluster_store<TOut>(out4, ((uint64_t)frame * (uint32_t)Pout + slot) * (uint64_t)(uint32_t)kn + j, ox, oy, oz, os);
    }
}


// ------------------------------------------------------------------------------------------- k_cluster_fuse_wide
// Complete-graph clusters of rigs with more than 8 cameras (<= 16): the C rays of an item do not fit a lane's registers
// (16 rays x 4 doubles), so an item = (descriptor, joint) is shared by FOUR adjacent lanes and its rays live in LDS.
//   wave pass = 16 items; lane = 4 item + g.
//   build   lane g fetches the keypoints of cameras g, g + 4, ... of its item (12 B each, prefetched one pass ahead) and
//           writes ray, |h|^2 and score to the wave's own LDS tile [camera][field][item]: 8-byte slots, camera stride 704 B.
//   solve   lane g owns first cameras m = g, g + 4, ... and walks the second camera CYCLICALLY, s = (m + delta) mod C for
//           delta = 1 .. C/2 (delta = C/2 of an even C only for m < C/2): every unordered pair once, C(C-1)/8 per lane.
//           At any step the four lanes of an item read cameras that differ mod 4 (C a multiple of 4) and the 8 items of a
//           32-lane group read 64 contiguous bytes each: with the 704-byte camera stride (= -64 mod 256) the four
//           cameras fall into the four quarters of the 64 banks -- conflict-free ds_read_b64.  A pair that wraps (s < m)
//           is solved with the roles of its rays exchanged: d = t_s - t_m comes from an ORDERED table [m][s], the solve is
//           symmetric under that exchange (same distance, same midpoint; last-bit rounding only, float32 outputs).
//           Fusion accumulates sum s (d + a S0 - b S1) per pair and adds 2 t_m sum s once per first camera
//           (t_m + t_s = 2 t_m + d), so the camera centres are read once per run, not per pair.
//   reduce  the four lanes' partial sums meet through two quad shuffles; lane g = 0 stores the joint.
// Arithmetic per pair as cluster_item (raw v_rsq_f64 for 1/dist with float32 outputs, one Newton step on it with float64
// outputs; gates select the score sum before the product);
// one reciprocal per pair (rcp + one Newton step).  Joints whose sum is not finite take cluster_joint_sequential.
// Member-list descriptors of the same launch are left to k_cluster_members.
// Dynamic LDS: cluster_wide_lds_bytes(C).
constexpr int kWideRayStride = 704;    // 5 fields x 16 items x 8 B = 640, padded to -64 mod 256
constexpr int kWideFieldStride = 128;
__host__ __device__ constexpr size_t cluster_wide_lds_bytes(int C) {
    return (size_t)72 * C + (size_t)24 * C + (size_t)24 * C * C + (size_t)(kBlock / 64) * C * kWideRayStride + 16;
}

constexpr int kWideWaves = 3;
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, kWideWaves) void k_cluster_fuse_wide(const ClusterDesc *__restrict__ desc,
                                                                                 const unsigned long long *__restrict__ cnt,
                                                                                 uint32_t desc_cap, Rig rig, const TIn *__restrict__ kpts,
                                                                                 Params prm, int Pmax, int J, int kn, unsigned long long kmagic,
                                                                                 int Pout, TOut *__restrict__ out4, uint32_t *__restrict__ out_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kMaxOwn = 4;                        // first cameras per lane: C <= 16
    const int C = rig.C, NP = rig.npairs;
    const int tid = threadIdx.x, lane = tid & 63, g = lane & 3, it = lane >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Ml = reinterpret_cast<double *>(smem);                        // [C][9]
    double *tl = Ml + 9 * C;                                              // [C][3]
    double *dl = tl + 3 * C;                                              // [C][C][3]: t_s - t_m for the ORDERED pair (m, s)
    char *tile = reinterpret_cast<char *>(dl + 3 * C * C) + (size_t)wave * C * kWideRayStride;   // this wave's rays
    for (int i = tid; i < 9 * C; i += kBlock) Ml[i] = rig.M[i];
    for (int i = tid; i < 3 * C; i += kBlock) tl[i] = rig.t[i];
    for (int i = tid; i < 3 * C * C; i += kBlock) {
        const int m = i / (3 * C), r = i - m * 3 * C, s = r / 3, k = r - 3 * s;
        // the host's d of the pair in its candidate order, negated for the reverse direction (bit-identical to the
        // pair table the other kernels use)
        double v = 0.0;
        if (m != s) {
            const int lo = m < s ? m : s, hi = m < s ? s : m;
            const int q = lo * C - lo * (lo + 1) / 2 + (hi - lo - 1);   // index of (lo, hi) in triangulation.py:56-57 order
            v = rig.pairc[6 * q + k];
            if (m > s) v = -v;
        }
        dl[i] = v;
    }
    const unsigned long long nd64 = cnt[kHandComplete];
    const uint32_t ndesc = nd64 < (unsigned long long)desc_cap ? (uint32_t)nd64 : desc_cap;
    const uint32_t total = ndesc * (uint32_t)kn;
    const uint32_t npass = (total + 15u) >> 4;
    const uint32_t W = gridDim.x * (uint32_t)(kBlock / 64);
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const float kthr_f32 = prm.kthr_f32;
    const double kthr = prm.kthr, dthr2 = prm.dthr2;
    const int half = C >> 1;
    const bool even = (C & 1) == 0;
    __syncthreads();

    struct Item {
        uint32_t frame, plo, phi, slot, j;
        bool valid;
    };
    auto locate = [&](uint32_t pass) {
        Item t;
        const uint32_t i = (pass << 4) + (uint32_t)it;
        t.valid = pass < npass && i < total;
        const uint32_t ic = t.valid ? i : 0u;
        const uint32_t di = (uint32_t)(((unsigned long long)ic * kmagic) >> 40);
        t.j = ic - di * (uint32_t)kn;
        t.frame = t.plo = t.phi = t.slot = 0u;
        if (t.valid) {
            const uint4 d = *reinterpret_cast<const uint4 *>(desc + di);
            t.frame = d.x;
            t.plo = d.y;
            t.slot = d.z;
            t.phi = d.w;
            t.valid = d.z < (uint32_t)Pout;   // (a voided entry)
        }
        return t;
    };
    auto fetch = [&](Kp3<TIn>(&dst)[kMaxOwn], const Item &t) {
#pragma unroll
        for (int i = 0; i < kMaxOwn; i++) {
            const int c = g + 4 * i;
            dst[i] = Kp3<TIn>{(TIn)0, (TIn)0, (TIn)0};
            if (t.valid && c < C) {
                const uint32_t p = ((c < 8 ? t.plo : t.phi) >> (4 * (c & 7))) & 15u;
                const uint32_t row = (t.frame * (uint32_t)C + (uint32_t)c) * (uint32_t)Pmax + p;
                dst[i] = kp3[(uint64_t)row * (uint32_t)J + t.j];
            }
        }
    };
    auto score_of = [&](const char *rec) -> TIn {
        if constexpr (sizeof(TIn) == 4)
            return (TIn)__uint_as_float((uint32_t)*(lds_cv_u64)(rec + 4 * kWideFieldStride));
        else
            return (TIn) * (lds_cv_f64)(rec + 4 * kWideFieldStride);
    };

    uint32_t p = blockIdx.x * (uint32_t)(kBlock / 64) + (uint32_t)wave;
    if (p >= npass) return;
    Kp3<TIn> cur[kMaxOwn], nxt[kMaxOwn];
    Item it0 = locate(p);
    fetch(cur, it0);
    Item it1 = locate(p + W);
    for (; p < npass; p += W) {
        fetch(nxt, it1);
        const Item it2 = locate(p + 2 * W);
        // ---- build: this lane's cameras of its item -> the wave's tile
#pragma unroll
        for (int i = 0; i < kMaxOwn; i++) {
            const int c = g + 4 * i;
            if (c < C) {
                const RayRec h = make_ray(Ml + 9 * c, cur[i].u, cur[i].v);
                char *rec = tile + c * kWideRayStride + it * 8;
                *reinterpret_cast<double *>(rec) = h.x;
                *reinterpret_cast<double *>(rec + kWideFieldStride) = h.y;
                *reinterpret_cast<double *>(rec + 2 * kWideFieldStride) = h.z;
                *reinterpret_cast<double *>(rec + 3 * kWideFieldStride) = h.a;
                if constexpr (sizeof(TIn) == 4)
                    *reinterpret_cast<unsigned long long *>(rec + 4 * kWideFieldStride) = (unsigned long long)__float_as_uint((float)cur[i].s);
                else
                    *reinterpret_cast<double *>(rec + 4 * kWideFieldStride) = (double)cur[i].s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- solve: first cameras m = g, g + 4, ...; second camera cyclic
        double aS = 0.0, aX = 0.0, aY = 0.0, aZ = 0.0;
        bool sing = false;   // a pair that is singular as the reference sees it (a c == b b, skew_ray_solve): the sequential routine flags the joint
        for (int m = g; m < C; m += 4) {
            const char *ra = tile + m * kWideRayStride + it * 8;
            lds_cv_f64 pa = (lds_cv_f64)ra;
            const double ax = pa[0], ay = pa[kWideFieldStride / 8], az = pa[2 * kWideFieldStride / 8], aa = pa[3 * kWideFieldStride / 8];
            const TIn sm = score_of(ra);
            bool okm;
            if constexpr (sizeof(TIn) == 4)
                okm = !((float)sm < kthr_f32);
            else
                okm = !((double)sm < kthr);
            double rS = 0.0, rX = 0.0, rY = 0.0, rZ = 0.0;   // this first camera's run
            const int dmax = (even && m >= half) ? half - 1 : half;
            int s = m;
            for (int dd = 1; dd <= dmax; dd++) {
                s = s + 1 == C ? 0 : s + 1;
                const char *rb = tile + s * kWideRayStride + it * 8;
                lds_cv_f64 pb = (lds_cv_f64)rb;
                const double bx = pb[0], by = pb[kWideFieldStride / 8], bz = pb[2 * kWideFieldStride / 8], bb = pb[3 * kWideFieldStride / 8];
                const TIn ss = score_of(rb);
                const double *dq = dl + 3 * (m * C + s);
                const double dx = dq[0], dy = dq[1], dz = dq[2];
                // A2 (triangulation.py:24-31)
                const double bq = fma(az, bz, fma(ay, by, ax * bx));
                const double det = fma(aa, bb, -(bq * bq));
                // (the fused determinant of equal rays is the rounding error of b b -- a finite, wrong member that `aS` below
                // does not give away: S0 = S1 = inf -> d2 = inf -> 1/dist = 0.  Tested explicitly instead.)
                sing |= aa * bb == bq * bq;
                const double e = fma(az, dz, fma(ay, dy, ax * dx));
                const double gg = fma(bz, dz, fma(by, dy, bx * dx));
                const double inv = rcp_nr1(det);   // (2^-46: 1e-13 m on the point)
                const double S0 = fma(bb, e, -(bq * gg)) * inv;
                const double S1 = fma(aa, gg, -(bq * e)) * inv;
                const double fx = fma(bx, S1, fma(ax, S0, -dx)), fy = fma(by, S1, fma(ay, S0, -dy)), fz = fma(bz, S1, fma(az, S0, -dz));
                const double d2 = fma(fz, fz, fma(fy, fy, fx * fx));
                // :72-74, sq = 2000 x the pair score; the gates select the score sum before the product
                bool keep;
                if constexpr (sizeof(TIn) == 4)
                    keep = okm && !((float)ss < kthr_f32) && !(d2 > dthr2);
                else
                    keep = okm && !((double)ss < kthr) && !(d2 > dthr2);
                double idist;
                if constexpr (sizeof(TOut) == 4)
                    idist = __builtin_amdgcn_rsq(d2);
                else
                    idist = rsq_nr1(d2);   // (d2 == 0: NaN instead of inf; the sum is not finite either way and the joint is re-done)
                const double sq = gated_sum_sel(sm, ss, keep) * idist;
                rS += sq;
                rX = fma(sq, fma(-bx, S1, fma(ax, S0, dx)), rX);   // d + a S0 - b S1
                rY = fma(sq, fma(-by, S1, fma(ay, S0, dy)), rY);
                rZ = fma(sq, fma(-bz, S1, fma(az, S0, dz)), rZ);
            }
            const double *tm = tl + 3 * m;
            aS += rS;
            aX += fma(2.0 * rS, tm[0], rX);    // sum s (t_m + t_s + a S0 - b S1) with t_m + t_s = 2 t_m + d
            aY += fma(2.0 * rS, tm[1], rY);
            aZ += fma(2.0 * rS, tm[2], rZ);
        }
        // ---- reduce over the four lanes of the item
        aS += __shfl_xor(aS, 1, 64);
        aX += __shfl_xor(aX, 1, 64);
        aY += __shfl_xor(aY, 1, 64);
        aZ += __shfl_xor(aZ, 1, 64);
        aS += __shfl_xor(aS, 2, 64);
        aX += __shfl_xor(aX, 2, 64);
        aY += __shfl_xor(aY, 2, 64);
        aZ += __shfl_xor(aZ, 2, 64);
        // aS = 2000 x sum_q s_q (:141); sum == 0 -> (0,0,0)/0 (:142-143): aX = aY = aZ = 0 then
        const double r = 0.5 * rcp_nr1(fmax(aS, 1e-300));
        double ox = aX * r, oy = aY * r, oz = aZ * r;   // :144-147
        double os = aS * (0.0005 / (double)NP);         // :148
        const unsigned long long sing_lanes = __ballot(sing);
        const bool bad = (!(aS < 1e300) || ((sing_lanes >> (lane & ~3)) & 15ull) != 0ull) && it0.valid;   // (the item's four lanes sit side by side)
        if (__ballot(bad)) {   // rare, wave-uniform branch
            if (bad && g == 0)
                cluster_joint_sequential<TIn, TOut>(rig, kp3, (int64_t)it0.frame, it0.plo, it0.phi, Pmax, J, (int)it0.j, prm, ox, oy, oz, os, out_flags);
        }
        if (it0.valid && g == 0)
            cluster_store<TOut>(out4, ((uint64_t)it0.frame * (uint32_t)Pout + it0.slot) * (uint64_t)(uint32_t)kn + it0.j, ox, oy, oz, os);
        __builtin_amdgcn_wave_barrier();   // the tile is rewritten by the next pass
#pragma unroll
        for (int i = 0; i < kMaxOwn; i++) cur[i] = nxt[i];
        it0 = it1;
        it1 = it2;
    }
}

// ------------------------------------------------------------------------------------------------ k_cluster_dlt
// method = SNOWTRI_DLT behind the streaming association (row N3; k_frame_recompute<1> keeps the frames the association leaves
// behind and the batches with an active condense_score_tol): lane = (descriptor, joint) over BOTH lists of a launch,
// first the complete-graph descriptors, then the member lists.
//   complete graph   one observation per camera: the person its 4-bit field names (cameras 0-7 in `persons`, 8-15 in `size`).
//                    CT = the camera count at compile time (2..8): the CT keypoints requested together, dlt_item<CT> as in
//                    k_fused_single<CT,1>; CT = 0: any count up to 16, four keypoints in flight at a time.
//   member list      the DISTINCT (camera, person) rows its member words name: one pass over the words marks them in a 256-bit
//                    set (<= 16 cameras x 16 persons), a second walks the set in row order -- one observation per row, not two
//                    per member (a 7-camera cluster has 21 members and 7 rows).
// Per lane the N-view DLT of k_fused_single<C,1> / k_frame_recompute<1> over them (dlt_add_observation, dlt_solve): the rows of
// the observations whose confidence is not below keypoint_score_threshold, two needed, joint score = their mean confidence.
// The persons' mean scores: k_person_scores.  P [C][12] in LDS (broadcast reads).  Dynamic LDS: cluster_dlt_lds_bytes(C).
constexpr int kClusterDltWaves = 3;
__host__ __device__ constexpr size_t cluster_dlt_lds_bytes(int C) { return (size_t)96 * C + 16; }
template <int CT, typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock, kClusterDltWaves) void k_cluster_dlt(const ClusterDesc *__restrict__ desc, const uint32_t *__restrict__ words,
                                                              const unsigned long long *__restrict__ cnt, uint32_t desc_cap, Rig rig,
                                                              const TIn *__restrict__ kpts, Params prm, int Pmax, int J, int kn,
                                                              unsigned long long kmagic, int Pout, TOut *__restrict__ out4) {
#pragma clang fp contract(off)   // (as in dlt_item)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Pl = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = CT > 0 ? CT : rig.C, R = C * Pmax;
    for (int i = tid; i < 12 * C; i += kBlock) Pl[i] = rig.P[i];
    const unsigned long long nc64 = cnt[kHandComplete];
    const uint32_t ng_raw = hand_member_descs(cnt[kHandMembers]);
    const uint32_t nc = nc64 < (unsigned long long)desc_cap ? (uint32_t)nc64 : desc_cap, ng = ng_raw < desc_cap ? ng_raw : desc_cap;
    const uint32_t total_c = nc * (uint32_t)kn, total_g = ng * (uint32_t)kn;
    const uint32_t npass_c = (total_c + 63u) >> 6, npass_g = (total_g + 63u) >> 6;
    const uint32_t W = gridDim.x * (uint32_t)(kBlock / 64);
    const unsigned long long magic_pmax = (((unsigned long long)1 << 40) + (unsigned)Pmax - 1) / (unsigned)Pmax;
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    __syncthreads();
    for (uint32_t p = blockIdx.x * (uint32_t)(kBlock / 64) + (uint32_t)wave; p < npass_c + npass_g; p += W) {
        const bool gen = p >= npass_c;   // (wave-uniform)
        const uint32_t i = ((gen ? p - npass_c : p) << 6) + (uint32_t)lane;
        bool valid = i < (gen ? total_g : total_c);
        const uint32_t ic = valid ? i : 0u;
        const uint32_t di = (uint32_t)(((unsigned long long)ic * kmagic) >> 40);
        const uint32_t j = ic - di * (uint32_t)kn;
        uint4 d = make_uint4(0u, 0u, 0u, 0u);
        if (valid) {
            d = *reinterpret_cast<const uint4 *>(desc + (gen ? desc_cap : 0u) + di);
            valid = d.z < (uint32_t)Pout;   // (a voided entry)
        }
        const uint64_t row0 = (uint64_t)d.x * (uint32_t)R;
        double ox, oy, oz, os;
        if constexpr (CT > 0) {
            if (!gen) {
                Kp3<TIn> cur[CT];
#pragma unroll
                for (int c = 0; c < CT; c++) {
                    const uint32_t person = (d.y >> (4 * c)) & 15u;
                    SNOWTRI_DEV_CHECK(!valid || person < (uint32_t)Pmax, 32);   // person index of camera c
                    cur[c] = kp3[(row0 + (uint32_t)(c * Pmax) + (valid ? person : 0u)) * (uint32_t)J + j];
                }
                asm volatile("" ::: "memory");   // (P is read from LDS by every item)
                dlt_item<CT, TIn>(Pl, cur, valid ? 0xffffu : 0u, prm, ox, oy, oz, os);
            }
        }
        if (CT == 0 || gen) {
            double A[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) A[a][b] = 0.0;
            double ssum = 0.0;
            int nuse = 0;
            auto observe = [&](int c, const Kp3<TIn> &k, bool have) {
                const bool use = have && !below_kthr(k.s, prm);
                dlt_add_observation(A, Pl + 12 * c, (double)k.u, (double)k.v, use ? 1.0 : 0.0);
                ssum += use ? (double)k.s : 0.0;
                nuse += use ? 1 : 0;
            };
            if (!gen) {
                for (int c0 = 0; c0 < C; c0 += 4) {   // four keypoints in flight
                    Kp3<TIn> k[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int c = c0 + u < C ? c0 + u : C - 1;
                        const uint32_t person = ((c < 8 ? d.y : d.w) >> (4 * (c & 7))) & 15u;
                        SNOWTRI_DEV_CHECK(!valid || person < (uint32_t)Pmax, 32);   // person index of camera c
                        k[u] = kp3[(row0 + (uint32_t)(c * Pmax) + (valid ? person : 0u)) * (uint32_t)J + j];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (c0 + u < C) observe(c0 + u, k[u], valid);
                }
            } else {
                const int size = valid ? (int)d.w : 0;
                unsigned long long seen[4] = {0ull, 0ull, 0ull, 0ull};   // rows of the frame (<= 16 cameras x 16 persons)
                for (int m0 = 0; __ballot(m0 < size) != 0ull; m0 += 8) {   // eight member words requested together (one after the other
                    uint32_t w[8];                                         // they were a chain of 10-21 round trips per item)
#pragma unroll
                    for (int u = 0; u < 8; u++) w[u] = m0 + u < size ? words[d.y + (uint32_t)(m0 + u)] : 0u;   // (rows 0 behind the list's end: in range for the shifts below, never marked)
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool live = m0 + u < size;
                        const int rm = (int)(w[u] & 1023u), rs = (int)((w[u] >> 10) & 1023u);
                        SNOWTRI_DEV_CHECK(!live || (rm < R && rs < R && R <= 256), 33);   // rows of the member inside the frame
                        if (R <= 64) {   // (wave-uniform; the usual rig: one word of the set -- the four-word update is 40 instructions per member)
                            seen[0] |= live ? ((1ull << rm) | (1ull << rs)) : 0ull;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                seen[q] |= live ? (((rm >> 6) == q ? 1ull << (rm & 63) : 0ull) | ((rs >> 6) == q ? 1ull << (rs & 63) : 0ull)) : 0ull;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (64 * q >= R) break;   // (wave-uniform)
                    unsigned long long left = seen[q];
                    while (__ballot(left != 0ull) != 0ull) {   // four rows per trip: their keypoints requested together
                        bool h[4];
                        int r[4];
                        Kp3<TIn> k[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            h[u] = left != 0ull;
                            r[u] = 64 * q + (h[u] ? __ffsll((long long)left) - 1 : 0);
                            left &= left - 1ull;
                            k[u] = kp3[(row0 + (uint32_t)r[u]) * (uint32_t)J + j];
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) observe((int)(((unsigned long long)(unsigned)r[u] * magic_pmax) >> 40), k[u], h[u]);
                    }
                }
            }
            const bool ok = valid && nuse >= 2;
            double e[4];
            dlt_solve(A, ok, e);   // (every lane of the wave: the solver votes)
            const double r = dlt_recip(e[3]);
            ox = ok ? e[0] * r : 0.0;
            oy = ok ? e[1] * r : 0.0;
            oz = ok ? e[2] * r : 0.0;
            os = ok ? ssum * dlt_recip((double)nuse) : 0.0;
        }
        if (valid) cluster_store<TOut>(out4, ((uint64_t)d.x * (uint32_t)Pout + d.z) * (uint64_t)(uint32_t)kn + j, ox, oy, oz, os);
    }
}

}  // namespace snowtri

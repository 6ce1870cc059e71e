// snowtri_assoc.hpp -- the multi-person association as STREAMING kernels (float32 outputs, pairwise method):
//
//   k_candidate_sums   A3  per frame and candidate: sum over the joints of the pair score       (triangulation.py:56-81)
//   k_associate        A4  per frame: kept list, centre joints, greedy clustering, filters; every output person becomes
//                          a descriptor for k_cluster_fuse                                       (triangulation.py:95-135,150-152)
//   k_cluster_fuse*    A4  per (output person, joint): score-weighted fusion                     (triangulation.py:136-149)
//
// k_frame_recompute (snowtri_general.hpp) runs the same three phases for one frame per workgroup: its candidate pass
// shares the CU with a one-wave clustering and with chains of dependent loads on its bookkeeping slab, and every phase
// pays the registers of the largest one.  Here every phase is a kernel of its own shape: the candidate pass keeps all
// waves of a workgroup in the solve loop (keypoints of the next joint chunk in flight in registers), the association
// is one WAVE per frame (thousands of frames in flight hide its dependent loads), the fusion streams (person, joint)
// items.  Between them: one double per candidate slot (Kc x 8 B per frame, against 12 C P J B of keypoints) and one
// 16-byte descriptor per output person.  Frames whose filter decisions are not safe on the fast arithmetic are listed
// and re-done by k_frame_recompute.
#pragma once
#include "snowtri_general.hpp"

namespace snowtri {

// ---------------------------------------------------------------------------------------------- k_candidate_sums
// LDS: [0, 64) flags and tickets | [64, 128) per frame parity and PAIR of cameras (16 bits each): the persons whose records are
// not finite | n_persons of the frame, per frame parity (2 x 4 B x C) | per camera pair d = t_s - t_m and the two camera
// indices (32 B) | ray matrices | arena = TWO buffers of one joint chunk of ray records each (layout and bank map:
// snowtri_general.hpp, p1_joint_stride): the waves solve the chunk in one buffer while the records of the next chunk are
// written to the other; at the end of a frame the buffer of its last chunk holds the partial sums of the joint sub-ranges.
// (Every byte in front of the arena counts: 8 cameras x 4 persons hold chunks of 20 joints in 52 KB with 1 728 bytes to spare.)
constexpr int kSumsHeadBytes = 64 + 2 * 8 * 4;
// workgroup shapes (threads, waves per SIMD the registers must allow, records of the coming chunk a thread holds in
// registers): 256 threads x 3 workgroups per CU for the small rigs, 512 x 2 or 1024 x 1 for the large ones
#ifndef SNOWTRI_SUMS_WAVES256
#define SNOWTRI_SUMS_WAVES256 3
#endif
#ifndef SNOWTRI_SUMS_GA
#define SNOWTRI_SUMS_GA 2
#endif
constexpr int kSumsWaves256 = SNOWTRI_SUMS_WAVES256;   // (4 waves per SIMD = 128 registers: the 2 x 4 tile spills, 8 x 4 measured 1027 us against 729)
// (Round 5 also carried a 64-thread shape -- ONE wave per workgroup, no barrier, one chunk buffer -- and a kernel with one lane per
// ray (snowtri_sums_rays.hpp of round 5); both measured no faster per call and were removed in round 6: EXPERIMENTS.md keeps their numbers.)
template <int THREADS>
struct SumsShape {
    static constexpr int kWavesPerSimd = THREADS == 256 ? kSumsWaves256 : 4;
    static constexpr int kPrefetch = THREADS == 256 ? 3 : 2;
};
constexpr int kSumsGA = SNOWTRI_SUMS_GA;   // persons of the FIRST camera per tile (when the person count is even)

__host__ __device__ inline size_t sums_arena_offset(int C, int npairs) {
    return ((size_t)kSumsHeadBytes + (size_t)8 * (C + (C & 1)) + (size_t)32 * npairs + (size_t)72 * C + 15) & ~(size_t)15;   // head | n_persons [2][C] | pairs | ray matrices
}
// bytes of ONE of the two chunk buffers
__host__ __device__ inline int sums_buffer_bytes(int C, int npairs, int lds_total, int) {
    return ((lds_total - (int)sums_arena_offset(C, npairs)) / 2) & ~15;
}
// item = tile of GA persons of a pair's first camera x GS persons of its second; JS = how many ways a chunk's joints are
// split over the workgroup's NW waves when one pass over the items leaves waves idle (whole waves take a joint sub-range:
// the lanes of a wave still read the same joint)
__host__ __device__ inline int sums_joint_split(int nitems, int NW) {
    const int iw = (nitems + 63) >> 6;
    int js = 1;
    while (2 * js * iw <= NW) js *= 2;
    return js;
}
// joints per chunk: what one buffer holds and the threads can prefetch, evened out over the chunks, and a multiple of the
// joint split of a frame whose cameras all list Pmax persons (the waves of a chunk meet at a barrier: 19 joints dealt to
// four waves would cost every chunk the time of 5)
__host__ __device__ inline int sums_chunk_joints(int C, int Pmax, int J, int npairs, int threads, int prefetch, int lds_total) {
    const int R = C * Pmax;
    int cap = sums_buffer_bytes(C, npairs, lds_total, threads) / p1_joint_stride(R);
    const int pf = prefetch * threads / R;   // every record of a chunk prefetched
    if (cap > pf) cap = pf;
    if (cap > 64) cap = 64;
    if (cap < 1) return 0;
    const int nch = (J + cap - 1) / cap;
    int jc = (J + nch - 1) / nch;   // 133 joints, room for 40 -> 34 + 33 + 33 + 33, not 3 x 40 + 13
    const int gs = p1_group_size(Pmax), ga = gs >= 2 ? kSumsGA : 1;
    const long long nitems = (long long)npairs * (Pmax / ga) * (Pmax / gs);
    if (nitems * 2 <= threads) {
        const int js = sums_joint_split((int)nitems, threads / 64);
        if (jc > js) {
            const int up = (jc + js - 1) / js * js;
            jc = up <= cap ? up : jc / js * js;
        }
    }
    return jc;
}

// The fast phase-1 arithmetic of p1_item_sums (snowtri_general.hpp) on a GA x GS TILE of candidates: GA consecutive persons
// of camera m against GS consecutive persons of camera s, acc[i * GS + u] += 2000 x score of candidate (pm0 + i, ps0 + u).
// A 2 x 4 tile reads 6 records per 8 solves where the 1 x 4 item read 5 per 4 (3.75 instead of 6.25 ds_read_b64 per solve).
//
// The gates of :73-74 WITHOUT compares.  The loop is bound by VALU issue, and measured in isolation (registers only,
// scripts/ubench/sums_mix.hip) the three gates of p1_item_sums -- two keypoint-threshold compares and `dn2 > det * dthr2`
// into scalar masks, s_or, v_cndmask -- cost 42 of its 110 SIMD cycles per candidate against 68 for the arithmetic.  Here:
//   * the keypoint gate is applied when the RECORD is written: a score below the threshold is stored as -infinity (GatedScore),
//     so that the sum of a pair's scores is negative -- or NaN -- exactly when one of them is gated (scores that pass are
//     >= keypoint_score_threshold >= 0, host-checked);
//   * the distance gate is the SIGN of r = fma(det, dthr2, -dn2) (r < 0 <=> dn2 > det * dthr2, rounded once instead of twice),
//     OR-ed into the sign of the score sum (v_and_or_b32);
//   * one v_max with 0 then zeroes the weight of a gated candidate: 5 full-rate instructions (v_add, v_fma, v_and_or, v_max,
//     v_cvt) in place of 9 with three of them on the scalar path: 110 -> 87 cycles per candidate in the same harness.
// The weight of a kept candidate is the float (double: the double) sum of the two scores as before and the geometry is
// untouched: a kept candidate adds the bits it added before.  What the sign trick cannot carry is a NaN (v_max returns
// its other operand, the sign of a NaN r is arbitrary): records that are not finite are caught where they are written
// (k_candidate_sums, `commit`; the arithmetic: gated_weight, snowtri_math.hpp) and send the frame to the exact pass.  r = -0 needs det = dn2 = 0 and dthr2 < 0: the host
// keeps batches with a negative distance_threshold off this kernel.
template <typename TIn>
struct GatedScore;
template <>
struct GatedScore<float> {
    static constexpr float value = -__builtin_huge_valf();
};
template <>
struct GatedScore<double> {
    static constexpr double value = -__builtin_huge_val();
};
template <int GA, int GS, typename TIn>
__device__ __forceinline__ void p1_tile_sums(const char *__restrict__ pa, const char *__restrict__ pb, int jstr, int nj, const Vec3 &d,
                                             const Params &prm, double (&acc)[GA * GS]) {
#pragma clang fp contract(off)   // (every fused product-sum below is an explicit fma(); the determinant must NOT be fused)
    for (int t = 0; t < nj; t++, pa += jstr, pb += jstr) {
        RayRec b[GS];
        TIn ss[GS];
#pragma unroll
        for (int u = 0; u < GS; u++) {
            b[u] = p1_load_ray(pb + kP1Rec * u);
            ss[u] = p1_load_score<TIn>(pb + kP1Rec * u);
        }
#pragma unroll
        for (int i = 0; i < GA; i++) {
            const RayRec a = p1_load_ray(pa + kP1Rec * i);
            const TIn sm = p1_load_score<TIn>(pa + kP1Rec * i);
            const double cx = fma(d.y, a.z, -(d.z * a.y)), cy = fma(d.z, a.x, -(d.x * a.z)), cz = fma(d.x, a.y, -(d.y * a.x));
#pragma unroll
            for (int u = 0; u < GS; u++) {
                const double bq = fma(a.z, b[u].z, fma(a.y, b[u].y, a.x * b[u].x));
                // separately rounded products: a pair that is singular as the reference sees it (a c == b b, skew_ray_solve)
                // has det == 0 exactly -> det * rsq(0) = NaN reaches the sum whatever the gates select -> the frame goes to
                // the exact pass, which flags it.  (The fused determinant is the rounding error of b b there.)
                const double det = a.a * b[u].a - bq * bq;
                const double dn = fma(cz, b[u].z, fma(cy, b[u].y, cx * b[u].x));
                const double dn2 = dn * dn;
                const double r = fma(det, prm.dthr2, -dn2);                                         // :73-74: r < 0 <=> dist > distance_threshold
                acc[i * GS + u] = fma(gated_weight(sm, ss[u], r), det * __builtin_amdgcn_rsq(dn2 * det), acc[i * GS + u]);
            }
        }
    }
}

#ifdef SNOWTRI_SUMS_TRACE   // dev build: wall-clock stamps (100 MHz) of every wave: kernel entry / exit and the phase boundaries of its second frame (scripts/dbg_sums_trace.py)
__device__ unsigned long long g_sums_trace[4096 * 4 * 16];
#define SUMS_STAMP_ALWAYS(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && threadIdx.x < 256) g_sums_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SUMS_STAMP(i) do { if (it == 1) SUMS_STAMP_ALWAYS(i); } while (0)
#if SNOWTRI_SUMS_TRACE == 2
#define SUMS_STAMP_C2(i) ((void)0)
#else
#define SUMS_STAMP_C2(i) do { if (c == 2) SUMS_STAMP(i); } while (0)   // the phases of the frame's third chunk
#endif
#else
#define SUMS_STAMP_ALWAYS(i) ((void)0)
#define SUMS_STAMP(i) ((void)0)
#define SUMS_STAMP_C2(i) ((void)0)
#endif
// csum[f][k] = sum over the joints of the score of candidate slot k (J x the mean of :79; 0 for a slot whose cameras
// list fewer persons), with the fast arithmetic of p1_tile_sums; a frame with a candidate whose fast sum cannot decide
// :80-81 -- not finite, or within 1e-6 relative of average_score_threshold -- is appended to exact_list and re-done by
// k_candidate_sums_exact; so is a frame with a record that is not finite in a listed row (`commit`).  out_flags[f] = 0.
// Host-checked: keypoint_score_threshold >= 0, distance_threshold >= 0, C <= 16, <= 16 persons per camera.  Dynamic LDS = lds_total.
//
// Frames: the first one by workgroup index, the following ones through a ticket counter (next_frame, zeroed by the host).
// Every frame costs the same instructions, but the SIMDs serve their waves oldest first: with a static deal the oldest
// workgroup of a CU ran through its frames at nearly full speed and left (8 x 4: after 455 of 850 us, wall-clock stamps
// per wave, -DSNOWTRI_SUMS_TRACE), the youngest finished alone on its CU with nothing to hide its latencies behind.
//
// Per frame, a software pipeline over the joint chunks with ONE barrier per chunk:
//     request the keypoints of chunk c + 1 -- in the frame's last chunk: of the NEXT frame's first chunk, with its n_persons
//     (the ticket is drawn at the frame's start)  ->  solve chunk c (buffer c & 1) with the requests in flight  ->  write
//     the records of chunk c + 1 (other buffer)  ->  barrier.
// A wave that is done with its solves fills LDS for the next chunk instead of waiting for the others, a frame starts on
// records that are already in LDS (its top is the ticket and one barrier), and what used to be two barriers and an exposed
// fill per chunk plus an HBM round trip per frame overlaps the solves.  The requests are made and used inside ONE pass of the
// chunk loop: round 4 requested chunk c + 2 behind chunk c + 1's fill and carried the registers around the loop, and the
// compiler put a copy of one loaded register -- with s_waitcnt vmcnt(0) in front of it -- right behind the loads: every
// request waited for its own data (1.0-1.4 us of the 9 us of a chunk of 8 x 4, wall-clock stamps of the chunk's phases).
// The items (tiles) are dealt to the lanes once per frame.  If one pass of the workgroup covers them (SINGLE: items x JS
// <= threads -- 8 cameras x 4 persons: 56 tiles, four joint sub-ranges, 224 of 256 lanes; 16 x 8: 960 tiles on 15 of 16
// waves) a lane keeps the sums of its tile in registers over all joint chunks; larger rigs walk the items in rounds and
// add a chunk's sums to csum (loaded at the start of the round, stored at its end: in flight during the solves).
template <typename TIn, int THREADS>
__global__ __launch_bounds__(THREADS, SumsShape<THREADS>::kWavesPerSimd) void k_candidate_sums(
    int64_t F, int Pmax, int J, int Jrow, int Kc, Rig rig, const TIn *__restrict__ kpts, const int32_t *__restrict__ n_persons, Params prm,
    double *__restrict__ csum, uint32_t *__restrict__ out_flags, uint32_t *__restrict__ exact_list, unsigned long long *exact_count,
    unsigned long long *next_frame, int lds_total) {
    // J: the joints summed (the first J of a row); Jrow: the joints a keypoint row holds.  The second launch of a batch with
    // keypoint_num < J and an active condense_score_tol sums the first keypoint_num joints only (:150 needs that mean before the
    // slots are assigned); it is given no exact list and no flags (a sum that is not finite sends the frame to k_frame_recompute
    // through k_associate, and average_score_threshold is not its business).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int B = THREADS, NW = THREADS / 64, NPF = SumsShape<THREADS>::kPrefetch;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    SUMS_STAMP_ALWAYS(12);
#ifdef SNOWTRI_SUMS_TRACE
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && threadIdx.x < 256) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sums_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 14] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax;
    // head: per frame parity p = it & 1: [4p] a candidate needs the exact sum, [4p + 1] ragged frame, [4p + 2] ticket of the
    // frame after it (a slot is rewritten two frames later, behind that frame's barriers)
    int32_t *head = reinterpret_cast<int32_t *>(smem);
    uint32_t *badrows = reinterpret_cast<uint32_t *>(smem + 64);           // [2][8]: bit p + 16 (c & 1) of [frame parity][c >> 1]: a record of person p of camera c is not finite
    const int Cp = C + (C & 1);
    int32_t *np_both = reinterpret_cast<int32_t *>(smem + kSumsHeadBytes); // [2][Cp]: persons listed by the cameras, by frame parity
    double *paird = reinterpret_cast<double *>(np_both + 2 * Cp);          // [npairs][3]
    int32_t *pairs = reinterpret_cast<int32_t *>(paird + 3 * rig.npairs);  // [npairs][2]
    double *Ml = reinterpret_cast<double *>(pairs + 2 * rig.npairs);
    if (tid < kSumsHeadBytes / 4) head[tid] = 0;
    for (int i = tid; i < 3 * rig.npairs; i += B) paird[i] = rig.pairc[6 * (i / 3) + i % 3];
    for (int i = tid; i < 2 * rig.npairs; i += B) pairs[i] = rig.pairs[i];
    for (int i = tid; i < 9 * C; i += B) Ml[i] = rig.M[i];
    char *const rec0 = smem + sums_arena_offset(C, rig.npairs);
    const int half = sums_buffer_bytes(C, rig.npairs, lds_total, B);   // bytes of one chunk buffer
    const int bufstep = half;
    const int jstr = p1_joint_stride(R), Jc = sums_chunk_joints(C, Pmax, J, rig.npairs, B, NPF, lds_total);
    const int nch = (J + Jc - 1) / Jc;
    const unsigned long long magic_pmax = (((unsigned long long)1 << 40) + (unsigned)Pmax - 1) / (unsigned)Pmax;
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);

    // ---- the records of a joint chunk: keypoints fetched into registers (a chunk ahead), then ray + |h|^2 + score into
    // LDS.  Lanes = consecutive joints of a row: coalesced 12-byte reads.  Rows a camera does not list are filled with
    // whatever the buffer holds: no valid candidate reads them.
    // The keypoints of a chunk are requested in FRONT of the solves of the chunk before it and turned into records BEHIND them,
    // in the same pass of the chunk loop: no register carries a load across a loop edge.  (Carried around the loops -- requested
    // behind chunk c, used behind chunk c + 1 -- the compiler placed its copies between the six loop variants right behind the
    // loads, with s_waitcnt vmcnt(0) in front of them: every request waited for its own data, ~1 us per chunk of 8 x 4.)
    Kp3<TIn> pre[NPF];
    int npv = Pmax;     // threads < C: n_persons of the frame whose first chunk is in `pre`
    // Slot n of a thread holds record i = tid + n B of a chunk of Jc joints: row r = i / Jc, joint jj = i % Jc of the chunk --
    // the same for every chunk of every frame, worked out once (a last chunk of fewer joints leaves the slots of its missing
    // joints idle; 133 joints are 7 x 19).  A chunk's fetch is then an add and a load per slot.
    int pre_map[NPF];   // row r | camera << 8 | jj << 12; -1: no record
    {
        const unsigned long long magic_jc = (((unsigned long long)1 << 40) + (unsigned)Jc - 1) / (unsigned)Jc;
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const int i = tid + n * B, ic = i < R * Jc ? i : 0;
            const int r = (int)(((unsigned long long)(unsigned)ic * magic_jc) >> 40), jj = ic - r * Jc;
            const int c = (int)(((unsigned long long)(unsigned)r * magic_pmax) >> 40);
            pre_map[n] = i < R * Jc ? (r | (c << 8) | (jj << 12)) : -1;
        }
    }
    auto fetch = [&](const Kp3<TIn> *kpf, int j0, int nj) {
        // (range-checked on the byte offset: an idle slot reads zeros from past the frame's end)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<Kp3<TIn> *>(kpf), 0, R * Jrow * (int)sizeof(Kp3<TIn>), 0x00020000);
        // every offset first, then the loads back to back: whatever the offsets need from scratch (the 1 024-thread shape spills)
        // is reloaded in FRONT of the requests -- a reload between or behind them waits for them (vmcnt counts in order).
        // Three loads of one component each, at offsets the compiler cannot relate (or it merges them again): one load of three
        // registers ties them to a register triple, and the compiler, wanting one of the three elsewhere during the solves,
        // copied it right behind the load -- s_waitcnt vmcnt(0) in front of the copy
        int o0[NPF], o1[NPF], o2[NPF];
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const int r = pre_map[n] & 255, jj = (pre_map[n] >> 12) & 63;
            const bool live = ((unsigned)pre_map[n] >> 12) < (unsigned)nj;
            SNOWTRI_DEV_CHECK(!live || (r < R && j0 + jj < J), 10);   // keypoint (row, joint) inside the frame
            o0[n] = live ? (r * Jrow + jj + j0) * (int)sizeof(Kp3<TIn>) : 0x7ffffff0;
            o1[n] = o0[n];
            o2[n] = o0[n];
            asm volatile("" : "+v"(o0[n]), "+v"(o1[n]), "+v"(o2[n]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            if constexpr (sizeof(TIn) == 4) {
                pre[n].u = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o0[n], 0, 0));
                pre[n].v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o1[n] + 4, 0, 0));
                pre[n].s = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o2[n] + 8, 0, 0));
            } else {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const u2v a = __builtin_amdgcn_raw_buffer_load_b64(rs, o0[n], 0, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs, o1[n] + 8, 0, 0),
                          d = __builtin_amdgcn_raw_buffer_load_b64(rs, o2[n] + 16, 0, 0);
                pre[n].u = __hiloint2double((int)a.y, (int)a.x);
                pre[n].v = __hiloint2double((int)b.y, (int)b.x);
                pre[n].s = __hiloint2double((int)d.y, (int)d.x);
            }
        }
    };
    auto fetch_frame = [&](int64_t fr) {   // first chunk and person counts of frame fr
        fetch(kp3 + fr * (int64_t)R * Jrow, 0, J < Jc ? J : Jc);
        if (n_persons && tid < C) npv = n_persons[fr * C + tid];
    };
    // (the score of a record is stored GATED, see p1_tile_sums; a record that is not finite -- a NaN or infinite pixel, a NaN
    // score -- sets the bit of its person in `bad`: the frame's end looks at the bits of the persons its cameras list)
    auto commit = [&](char *buf, uint32_t *bad, int nj) {
#ifdef SNOWTRI_K1_NOFILL   // TIMING-ONLY build (wrong sums): no ray records written
        if (F >= 0) return;
#endif
        // the ray matrices of ALL the thread's records first, then the rays, then the stores under their predicates: record by
        // record inside `if (live)` the compiler read a matrix in three dependent LDS round trips per record -- nine in a row,
        // ~1 us per chunk of 8 x 4 (wall-clock stamps) -- with nothing of the other records to put between them
        double Mr[NPF][9];
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const double *M = Ml + 9 * (pre_map[n] >= 0 ? (pre_map[n] >> 8) & 15 : 0);
#pragma unroll
            for (int k = 0; k < 9; k++) Mr[n][k] = M[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const int r = pre_map[n] & 255, c = (pre_map[n] >> 8) & 15, off = ((pre_map[n] >> 12) & 63) * jstr + kP1Rec * r;
            const RayRec h = make_ray(Mr[n], pre[n].u, pre[n].v);
            const TIn sc = pre[n].s;
            const bool notfinite = !(fma((double)sc, 0.0, h.a) < 1e300);   // NaN or infinite |h|^2, NaN or infinite score
            if (((unsigned)pre_map[n] >> 12) < (unsigned)nj) {
                SNOWTRI_DEV_CHECK(off + kP1Rec <= half && c < C, 11);   // record inside the buffer
                if (notfinite) atomicOr(&bad[c >> 1], 1u << (r - c * Pmax + 16 * (c & 1)));
                p1_store_record<TIn>(buf + off, h, below_kthr(sc, prm) ? GatedScore<TIn>::value : sc);
            }
        }
    };

    int64_t f = blockIdx.x;
    int par = 0;   // buffer of the frame's first chunk
    __syncthreads();   // constants and the cleared head
    if (f < F) {   // the first chunk of the workgroup's first frame (every later one is written during the frame before it)
        fetch_frame(f);
        commit(rec0, badrows, J < Jc ? J : Jc);
        if (tid < C) np_both[tid] = npv;
    }
    for (int it = 0; f < F; it++) {
        int32_t *hd = head + 4 * (it & 1);
        const int32_t *np_l = np_both + Cp * (it & 1);   // [C] persons listed by the cameras in this frame
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * Jrow;
        double *cs_f = csum + f * (int64_t)Kc;
        SUMS_STAMP(0);
        // (the ticket is needed in the frame's LAST chunk: with two and more chunks it is handed to the workgroup behind the
        // solves of the first one, so that the barrier below does not wait for the atomic's round trip)
        unsigned long long ticket = 0;
        if (tid == 0) {
            ticket = atomicAdd(next_frame, 1ull);
            if (nch == 1) hd[2] = (int32_t)ticket;
        }
        if (tid < C && np_l[tid] != Pmax) hd[1] = 1;   // (np_l[tid] was written by this thread)
        uint32_t *bad = badrows + 8 * (it & 1);
        __syncthreads();   // first chunk (written during the frame before), np_l, ragged flag and ticket are there
        SUMS_STAMP(1);
        if (tid == 0) head[4 * ((it & 1) ^ 1)] = head[4 * ((it & 1) ^ 1) + 1] = 0;   // the flags of the frame before: read for the last time in front of this barrier, set again behind this frame's last one
        int64_t fnext = nch == 1 ? (int64_t)gridDim.x + (int64_t)(uint32_t)hd[2] : 0;
        const int GS = hd[1] ? 1 : p1_group_size(Pmax);   // a ragged frame keeps one candidate per lane
        const int GA = GS >= 2 ? kSumsGA : 1;      // (GS >= 2: Pmax is even)
        const int NG = Pmax / GS, per_q = (Pmax / GA) * NG, nitems = rig.npairs * per_q;
        int JS = sums_joint_split(nitems, NW);
        const bool single = nitems * JS <= B && (JS == 1 || (size_t)JS * Kc * 8 <= (size_t)half);
        if (!single) JS = 1;
        const int wpg = NW / JS;               // waves that share a joint sub-range
        const int jsub = wv / wpg, iw = wv - jsub * wpg;
        const unsigned long long magic_pq = (((unsigned long long)1 << 40) + (unsigned)per_q - 1) / (unsigned)per_q;
        const unsigned long long magic_ng = (((unsigned long long)1 << 40) + (unsigned)NG - 1) / (unsigned)NG;
        bool redo = false;

        // one loop nest per (tile, SINGLE): the variants share no registers with loads in flight
        auto frame_body = [&](auto gs_c, auto single_c) {
            constexpr int GSC = decltype(gs_c)::value, GAC = GSC >= 2 ? kSumsGA : 1, NT = GAC * GSC;
            constexpr bool SINGLE = decltype(single_c)::value;
            // item of (round base, lane) -> first candidate slot, record offsets of its rows, pair offset; live?
            struct Item {
                int k0, oa, ob, dq;   // dq: the pair's d = t_s - t_m in `paird` (read where a chunk's solves start: six registers less over the frame)
                bool live, cand;
            };
            auto locate = [&](int base) {
                Item t;
                const int item = base + lane;
                const bool live = item < nitems;
                const int ii = live ? item : 0;
                const int q = (int)(((unsigned long long)(unsigned)ii * magic_pq) >> 40), r2 = ii - q * per_q;
                const int pmg = (int)(((unsigned long long)(unsigned)r2 * magic_ng) >> 40), pm = pmg * GAC, ps0 = (r2 - pmg * NG) * GSC;
                const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
                t.dq = 3 * q;
                t.k0 = q * pp + pm * Pmax + ps0;   // candidate (i, u) of the tile: slot k0 + i Pmax + u
                t.oa = kP1Rec * (mc * Pmax + pm);
                t.ob = kP1Rec * (sc * Pmax + ps0);
                t.live = t.cand = live;
                SNOWTRI_DEV_CHECK(!live || (t.k0 >= 0 && t.k0 + (GAC - 1) * Pmax + GSC <= Kc && t.oa + GAC * kP1Rec <= jstr && t.ob + GSC * kP1Rec <= jstr), 12);   // slots and rows of the item
                if constexpr (GSC == 1) t.cand = live && pm < np_l[mc] && ps0 < np_l[sc];   // empty slots stay at 0
                return t;
            };
            auto slot = [&](const Item &t, int u) { return t.k0 + (u / GSC) * Pmax + u % GSC; };
            Item mine{};
            double tot[NT];
            if constexpr (SINGLE) {
                mine = locate(iw * 64);
#pragma unroll
                for (int u = 0; u < NT; u++) tot[u] = 0.0;
            }
            for (int c = 0, j0 = 0; c < nch; c++, j0 += Jc) {
                const int nj = (J - j0) < Jc ? (J - j0) : Jc;
                const char *cur = rec0 + ((par ^ c) & 1) * bufstep;
                // the waves of joint sub-range jsub walk joints [jlo, jhi) of the chunk (the sub-ranges rotate from chunk to
                // chunk: where the joints of a chunk do not divide by JS every wave gets the long sub-range in turn)
                const int jrot = (jsub + c) & (JS - 1);
                const int jlo = jrot * nj / JS, jhi = (jrot + 1) * nj / JS;
                // requested here, in flight during the solves: the chunk after this one -- behind the frame's last chunk the
                // first chunk of the NEXT frame (its ticket was drawn at the frame's start) with its n_persons
                // (one request site: a frame without a successor requests a chunk of no joints)
                const bool last = c + 1 >= nch, succ = fnext < F;
                const int j0_next = last ? 0 : j0 + Jc;
                const int nj_next = last ? (succ ? (J < Jc ? J : Jc) : 0) : ((J - j0 - Jc) < Jc ? (J - j0 - Jc) : Jc);
                const char *pa = cur + jlo * jstr + mine.oa, *pb = cur + jlo * jstr + mine.ob;
                const int njs = mine.cand ? jhi - jlo : 0;
                if (last && succ && n_persons && tid < C) npv = n_persons[fnext * C + tid];   // (in front of the keypoint requests, as everything that may touch scratch)
                fetch(last && succ ? kp3 + fnext * (int64_t)R * Jrow : kpf, j0_next, nj_next);
                __builtin_amdgcn_sched_barrier(0);   // (nothing of the solves' set-up behind the requests: a reload from scratch there waits for them)
                SUMS_STAMP_C2(2);
                if constexpr (SINGLE) {
                    const Vec3 d = {paird[mine.dq], paird[mine.dq + 1], paird[mine.dq + 2]};
                    p1_tile_sums<GAC, GSC, TIn>(pa, pb, jstr, njs, d, prm, tot);
                } else {
                    for (int base = iw * 64; base < nitems; base += wpg * 64) {
                        const Item t = locate(base);
                        double old[NT], acc[NT];
#pragma unroll
                        for (int u = 0; u < NT; u++) {
                            acc[u] = 0.0;
                            old[u] = (c > 0 && t.cand) ? cs_f[slot(t, u)] : 0.0;   // (in flight during the solves)
                        }
                        const Vec3 d = {paird[t.dq], paird[t.dq + 1], paird[t.dq + 2]};
                        p1_tile_sums<GAC, GSC, TIn>(cur + jlo * jstr + t.oa, cur + jlo * jstr + t.ob, jstr, t.cand ? jhi - jlo : 0, d, prm, acc);
                        // (the first chunk defines every slot of the frame, the empty ones as 0)
                        if (t.cand || (c == 0 && t.live)) {
#pragma unroll
                            for (int u = 0; u < NT; u++) cs_f[slot(t, u)] = t.cand ? old[u] + acc[u] : 0.0;
                        }
                    }
                }
                SUMS_STAMP_C2(3);
                // (the buffer chunk c - 1 was solved in; the next frame's first chunk lands where this frame's last chunk is not)
                commit(rec0 + ((par ^ (c + 1)) & 1) * bufstep, last ? badrows + 8 * ((it & 1) ^ 1) : bad, nj_next);
                if (last && succ && tid < C) np_both[Cp * ((it & 1) ^ 1) + tid] = npv;
                SUMS_STAMP_C2(4);
                if (nch > 1 && c == 0 && tid == 0) hd[2] = (int32_t)ticket;
                __syncthreads();   // chunk c is solved (its buffer is free), chunk c + 1 is in LDS, csum is up to date
                if (nch > 1 && c == 0) fnext = (int64_t)gridDim.x + (int64_t)(uint32_t)hd[2];
                SUMS_STAMP_C2(6);
#if defined(SNOWTRI_SUMS_TRACE) && SNOWTRI_SUMS_TRACE == 2   // (instead of the stamps of chunk 2: the end of every chunk, slots 2 .. 9)
                if (c < 8) SUMS_STAMP(2 + c);
#endif
            }
            SUMS_STAMP(10);
            // the 1 / (2 * 1000) of :72, and whether a mean can decide :80-81
            auto finish = [&](int k, double v) {
                const double s_ = v * 0.0005, mean = s_ / (double)J;
                cs_f[k] = s_;
                redo |= exact_list != nullptr && (!(fabs(mean) < 1e300) || (s_ != 0.0 && fabs(mean - prm.avg_thr) <= 1e-6 * fabs(mean)));
            };
            if constexpr (SINGLE) {
                if (JS == 1) {   // a lane owns its candidates' whole sums
                    if (mine.live) {
#pragma unroll
                        for (int u = 0; u < NT; u++) finish(slot(mine, u), mine.cand ? tot[u] : 0.0);
                    }
                } else {         // the partial sums of the joint sub-ranges meet in the buffer of the last chunk
                    double *lsum = reinterpret_cast<double *>(rec0 + ((par ^ (nch - 1)) & 1) * bufstep);
                    if (mine.live) {
#pragma unroll
                        for (int u = 0; u < NT; u++) lsum[jsub * Kc + slot(mine, u)] = mine.cand ? tot[u] : 0.0;
                    }
                    __syncthreads();
                    for (int k = tid; k < Kc; k += B) {
                        double v = lsum[k];
                        for (int s_ = 1; s_ < JS; s_++) v += lsum[s_ * Kc + k];
                        finish(k, v);
                    }
                }
            } else {
                for (int k = tid; k < Kc; k += B) finish(k, cs_f[k]);
            }
        };
        if (single) {
            if (GS == 4)
                frame_body(std::integral_constant<int, 4>{}, std::true_type{});
            else if (GS == 2)
                frame_body(std::integral_constant<int, 2>{}, std::true_type{});
            else
                frame_body(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            if (GS == 4)
                frame_body(std::integral_constant<int, 4>{}, std::false_type{});
            else if (GS == 2)
                frame_body(std::integral_constant<int, 2>{}, std::false_type{});
            else
                frame_body(std::integral_constant<int, 1>{}, std::false_type{});
        }
        if (redo) hd[0] = 1;
        // a listed row with a record that is not finite (every commit of the frame is behind a barrier by now): the exact pass
        // takes the frame; a launch without an exact list makes the frame's sums NaN -- which is what they would be -- so that
        // k_associate leaves it to k_frame_recompute
        if (tid < C) {
            const uint32_t listed = np_l[tid] >= 16 ? 0xffffu : ((1u << np_l[tid]) - 1u);
            if ((bad[tid >> 1] >> (16 * (tid & 1))) & listed) hd[0] = exact_list ? 1 : 2;
        }
        __syncthreads();
        if (tid < 8) bad[tid] = 0u;   // (read above by two threads each; set again during the frame after the next)
        if (hd[0] == 2)
            for (int k = tid; k < Kc; k += B) cs_f[k] = __longlong_as_double(0x7ff8000000000000ll);
        if (tid == 0) {
            if (out_flags) out_flags[f] = 0u;
            if (hd[0] && exact_list) exact_list[atomicAdd(exact_count, 1ull)] = (uint32_t)f;
        }   // (hd[0] and hd[1] are cleared behind the next frame's first barrier: every thread reads hd[0] here)
        SUMS_STAMP(11);
        par ^= nch & 1;   // the next frame's first chunk goes where this frame's last chunk was not (its partial sums may still be read)
        f = fnext;
    }
    SUMS_STAMP_ALWAYS(13);
}

// The frames k_candidate_sums listed: every candidate sum again with the accurate arithmetic (1/dist by v_rsq_f64 + one
// Newton step, exact intersection -> inf, the gate assigns 0 AFTER the product as :72-74 do; singular pairs flagged).
// One workgroup per listed frame, one wave per candidate slot, lanes = joints; keypoints straight from global memory.
template <typename TIn>
__global__ __launch_bounds__(kBlock) void k_candidate_sums_exact(int Pmax, int J, int Kc, Rig rig, const TIn *__restrict__ kpts,
                                                                 const int32_t *__restrict__ n_persons, Params prm,
                                                                 double *__restrict__ csum, uint32_t *__restrict__ out_flags,
                                                                 const uint32_t *__restrict__ exact_list,
                                                                 const unsigned long long *__restrict__ exact_count) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax;
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const unsigned long long n = *exact_count;
    for (unsigned long long e = blockIdx.x; e < n; e += gridDim.x) {
        const int64_t f = (int64_t)exact_list[e];
        const int32_t *np_f = n_persons ? n_persons + f * C : nullptr;
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * J;
        bool sing = false;
        for (int k = wv; k < Kc; k += kBlock / 64) {
            const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
            const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
            double acc = 0.0;
            if (!np_f || (pm < np_f[mc] && ps < np_f[sc])) {
                const double *pc = rig.pairc + 6 * q;
                const Vec3 d = {pc[0], pc[1], pc[2]}, ts = {pc[3], pc[4], pc[5]};
                for (int j = lane; j < J; j += 64) {
                    const Kp3<TIn> km = kpf[(size_t)(mc * Pmax + pm) * J + j], ks = kpf[(size_t)(sc * Pmax + ps) * J + j];
                    const RayRec a = make_ray(rig.M + 9 * mc, km.u, km.v), b = make_ray(rig.M + 9 * sc, ks.u, ks.v);
                    const PairSolve o = pair_solve_fast<false>(a, b, d, ts);
                    sing |= o.singular;
                    const bool kp_ = !below_kthr(km.s, prm) && !below_kthr(ks.s, prm) && !(o.d2 > prm.dthr2);   // :73-74
                    // the gate ASSIGNS 0: select after the product, 0 * inf (exact intersection) would be NaN
                    acc += kp_ ? sum_score(km.s, ks.s) * (0.5 * o.score_base) : 0.0;                             // :72
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) csum[f * (int64_t)Kc + k] = acc;
        }
        if (sing && out_flags) atomicOr(&out_flags[f], 1u /*SNOWTRI_FLAG_SINGULAR*/);
    }
}

// ---------------------------------------------------------------------------------------------------- k_associate
// One WAVE per frame (64-thread workgroups dealt round-robin): phase 2 and the filters of k_frame_recompute on the
// candidate sums of k_candidate_sums.  Every step is a handful of dependent loads: they are hidden by the thousands of
// frames in flight, and everything a frame needs more than once is fetched ONCE, by all lanes in parallel, into LDS.
// Per frame:
//   kept list in candidate order (:79-81; eight 64-candidate loads of csum in flight at a time) -> per kept candidate, in
//   parallel: its ray rows + camera pair (one word), its score sum, its centre joint (one fast solve) -> greedy clustering
//   (:107-130) -> members grouped by cluster -> per cluster the size filter (:132-134) and the mean-score filter
//   (:150-152; a person's mean score is the mean of its members' candidate means because keypoint_num == J) ->
//   out_count, out_pscore, zero-filled unused slots, and one descriptor per output person: a cluster that is the
//   complete graph over one detection per camera carries the person index of every camera (4 bits x 16), any other its
//   member words (rm | rs << 10 | q << 20, as in k_frame_recompute).
// A frame whose decisions are not safe here -- a mean that is not finite or within 1e-6 of condense_score_tol, more kept
// candidates or clusters than the wave's LDS holds -- is appended to slow_list: the host runs the listed frames through a
// second launch with LDS for every candidate slot (one wave per CU), and what that leaves behind through k_frame_recompute.
// LDS: [0, 64) scalars | staging of the output persons (~24 B each) | rows of the cameras (complete-graph test, 4 B x C) |
// camera indices of the pairs (8 B each) | arena: kept index, cluster id, word (4 B each), score sum (8 B), centre (24 B)
// per kept candidate, then 12 B per cluster.
__host__ __device__ inline size_t associate_arena_offset(int C, int npairs, int Pout) {
    const size_t pq = ((size_t)Pout * 4 + 15) & ~(size_t)15;   // four 4-byte arrays padded to 16 bytes + one of doubles
    return ((size_t)64 + 4 * pq + (size_t)8 * Pout + (size_t)4 * C + (size_t)8 * npairs + 15) & ~(size_t)15;
}
// bytes per kept candidate: kept index, member index, word (4 B each), score sum (8 B) -- and its centre (24 B) unless the
// centres live in registers (RC > 0 rounds of 64 candidates: lane l holds the centres of candidates l, 64 + l, ...)
constexpr int kAssocKeptBytes = 44, kAssocKeptBytesRegs = 20, kAssocClusterRoom = 12 * 48 + 16;
struct AssocShape {
    int rc;       // 0: centres in LDS; 4 / 16: in registers, at most 64 rc kept candidates
    size_t lds;
};
__host__ inline AssocShape associate_shape(int C, int npairs, int Pout, int64_t Kc) {
    const int64_t Pmax = (int64_t)(0.5 + __builtin_sqrt((double)Kc / (double)(npairs > 0 ? npairs : 1)));
    const int64_t true_pairs = (int64_t)npairs * Pmax;   // the candidates of Pmax persons every camera sees
    // The kernel is latency-bound at one wave per frame: the frames in flight per CU are what its LDS allows, so the first
    // launch is sized for what the reference's own workloads keep (25 % of the slots at 8 x 4, 13 % at 16 x 8: the true
    // pairs plus a few ghosts) and a frame that keeps more goes through the second launch.  With the centres in registers a
    // kept candidate costs 20 bytes: 8 x 4 holds 4.6 KB per frame, 16 x 8 22 KB (7 frames per CU instead of 3 at 48 KB).
    AssocShape a;
    if (true_pairs * 11 / 8 <= 256 && (Kc / 4 < 64 ? 64 : Kc / 4) <= 256) {
        int64_t slots = Kc / 4 < 64 ? 64 : Kc / 4;
        if (slots < true_pairs * 11 / 8) slots = true_pairs * 11 / 8;
        if (slots > Kc) slots = Kc;
        a.rc = 4;
        a.lds = associate_arena_offset(C, npairs, Pout) + (size_t)kAssocKeptBytesRegs * (size_t)slots + kAssocClusterRoom + 64;
    } else if (true_pairs + 64 <= 1024) {
        a.rc = 16;
        a.lds = associate_arena_offset(C, npairs, Pout) + (size_t)kAssocKeptBytesRegs * 1024 + kAssocClusterRoom + 64;
    } else {
        int64_t slots = Kc / 4 < 64 ? 64 : Kc / 4;
        if (slots < true_pairs * 11 / 8) slots = true_pairs * 11 / 8;
        if (slots > Kc) slots = Kc;
        a.rc = 0;
        a.lds = associate_arena_offset(C, npairs, Pout) + (size_t)(kAssocKeptBytes + 6) * (size_t)slots + 64;
        if (a.lds > 48 * 1024) a.lds = 48 * 1024;
    }
    if (a.lds < 4096) a.lds = 4096;
    return a;
}

// LDS that holds EVERY candidate slot of a frame (second launch, for the frames whose kept list did not fit the first)
__host__ inline size_t associate_lds_bytes_full(int C, int npairs, int Pout, int64_t Kc) {
    const size_t want = associate_arena_offset(C, npairs, Pout) + (size_t)(kAssocKeptBytes + 6) * (size_t)Kc + 64;
    const size_t cap = 160 * 1024;
    return want > cap ? cap : (want < 4096 ? 4096 : want);
}

#ifdef SNOWTRI_ASSOC_TRACE   // dev build: wall-clock stamps (100 MHz) of every workgroup's second frame at the phase boundaries (scripts/dbg_assoc_trace.py)
__device__ unsigned long long g_assoc_trace[4096 * 16];
#define ASSOC_STAMP(i) do { if (threadIdx.x == 0 && it == 1 && blockIdx.x < 4096) g_assoc_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ASSOC_STAMP(i) ((void)0)
#endif
// waves per SIMD of the small-rig variants: the kernel is latency-bound, so frames in flight count for more than registers
// (96 VGPRs, 28 of them spilled: 8 x 4 1.18 -> 1.14 ms per 10 000 frames against 4 waves without spills; 6 waves: the same)
constexpr int kAssocWaves = 5;
template <typename TIn, typename TOut, int RC>
__global__ __launch_bounds__(64, RC > 4 ? 2 : kAssocWaves) void k_associate(int64_t F, int Pmax, int J, int Kc, Rig rig, const TIn *__restrict__ kpts,
                                                  const int32_t *__restrict__ n_persons, Params prm, int Pout,
                                                  const double *__restrict__ csum, TOut *__restrict__ out4,
                                                  TOut *__restrict__ out_ps, int32_t *__restrict__ out_count,
                                                  uint32_t *__restrict__ out_flags, ClusterDesc *__restrict__ desc,
                                                  uint32_t *__restrict__ hand_words, unsigned long long *hand_counters,
                                                  uint32_t desc_cap, uint32_t word_cap, uint32_t *__restrict__ slow_list,
                                                  unsigned long long *slow_count, int lds_total, int allow_complete,
                                                  const uint32_t *__restrict__ frame_list, const unsigned long long *frame_count,
                                                  int write_person_scores, const double *__restrict__ csum_kn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax, NPq = rig.npairs;
    const int ci = prm.center;
    const size_t pq = ((size_t)Pout * 4 + 15) & ~(size_t)15;
    uint32_t *st_a = reinterpret_cast<uint32_t *>(smem + 64);             // [Pout] persons (low word) | first member
    int32_t *st_size = reinterpret_cast<int32_t *>(smem + 64 + pq);       // [Pout] 0: complete graph, else members
    uint32_t *st_idx = reinterpret_cast<uint32_t *>(smem + 64 + 2 * pq);  // [Pout] index inside its descriptor list
    uint32_t *st_word = reinterpret_cast<uint32_t *>(smem + 64 + 3 * pq); // [Pout] persons (high word) | offset of its member words
    double *st_avg = reinterpret_cast<double *>(smem + 64 + 4 * pq);      // [Pout]
    int32_t *rowc = reinterpret_cast<int32_t *>(st_avg + Pout);           // [C]
    int32_t *pairs = rowc + C;                                            // [npairs][2]
    const int arena_off = (int)associate_arena_offset(C, NPq, Pout);
    char *arena = smem + arena_off;
    const int arena_bytes = lds_total - arena_off;
    constexpr int kKept = RC > 0 ? kAssocKeptBytesRegs : kAssocKeptBytes;
    int n_cap = arena_bytes > 64 ? (arena_bytes - 64 - (RC > 0 ? kAssocClusterRoom : 0)) / kKept : 0;
    if (RC > 0 && n_cap > 64 * RC) n_cap = 64 * RC;
    if (n_cap < 0) n_cap = 0;
    const unsigned long long magic_pmax = (((unsigned long long)1 << 40) + (unsigned)Pmax - 1) / (unsigned)Pmax;
    const unsigned long long magic_pp = (((unsigned long long)1 << 40) + (unsigned)pp - 1) / (unsigned)pp;
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const PackedWriter<TOut> wr{out4, out_ps};
    // keypoint_num < J: the mean of :150 runs over the first keypoint_num joints.  With condense_score_tol <= 0 the filter of
    // :151-152 never drops a person of non-negative scores and the candidate sums over all J joints will do; otherwise csum_kn
    // holds the sums over the first keypoint_num joints (a second launch of k_candidate_sums) and the filter is decided on them.
    // There and with float64 outputs the persons' mean scores are written by k_person_scores from the fused joints
    // (write_person_scores == 0); the unused slots' zeros are written here either way
    const int kn = prm.kn;
    for (int i = lane; i < 2 * NPq; i += 64) pairs[i] = rig.pairs[i];

    // frame_list: the frames a first launch with less LDS left behind (frame_count of them, known on the device only)
    const int64_t nframes = frame_list ? (int64_t)*frame_count : F;
    // (frames dealt round-robin: a ticket counter as in k_candidate_sums measured WORSE here -- 8 x 4: 97 -> 128 us per launch --
    // the ticket's round trip sits in front of the frame's first loads on the in-order memory counter)
    int it = 0;
    for (int64_t fi = blockIdx.x; fi < nframes; fi += gridDim.x, it++) {
        const int64_t f = frame_list ? (int64_t)frame_list[fi] : fi;
        ASSOC_STAMP(0);
        SNOWTRI_DEV_CHECK(f >= 0 && f < F, 21);   // (a listed frame index belongs to the segment)
        const int32_t *np_f = n_persons ? n_persons + f * C : nullptr;
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * J;
        const double *cs_f = csum + f * (int64_t)Kc;
        const double *cs_a = csum_kn ? csum_kn + f * (int64_t)Kc : cs_f;   // what a person's mean score is made of
        // candidate slot k -> ray rows, camera pair; false if a camera lists fewer persons than the slot's
        auto slot_rows = [&](int k, int &rm, int &rs, int &q) -> bool {
            q = (int)(((unsigned long long)(unsigned)k * magic_pp) >> 40);
            const int rr = k - q * pp, pm = (int)(((unsigned long long)(unsigned)rr * magic_pmax) >> 40), ps = rr - pm * Pmax;
            const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
            rm = mc * Pmax + pm;
            rs = sc * Pmax + ps;
            return !np_f || (pm < np_f[mc] && ps < np_f[sc]);
        };
        bool slow = false, zero_filled = false;   // wave-uniform
        __syncthreads();     // (the previous frame's LDS is dead; the pair table is there)
        bool ragged = false;
        if (np_f) {
            bool rg = false;
            for (int c = lane; c < C; c += 64) rg |= np_f[c] != Pmax;
            ragged = __ballot(rg) != 0ull;
        }
        ASSOC_STAMP(1);
        // ---- kept list in candidate order (:79-81)
        int32_t *kidx = reinterpret_cast<int32_t *>(arena);
        int n = 0;
        for (int base0 = 0; base0 < Kc && !slow; base0 += 512) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = base0 + 64 * u + lane;
                v[u] = k < Kc ? cs_f[k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = base0 + 64 * u + lane;
                bool kp_ = k < Kc;
                if (kp_ && ragged) {
                    int rm, rs, q;
                    kp_ = slot_rows(k, rm, rs, q);
                }
                kp_ = kp_ && !(v[u] / (double)J < prm.avg_thr);
                const unsigned long long m = __ballot(kp_);
                const int cnt = __popcll(m);
                if (n + cnt > n_cap) slow = true;
                SNOWTRI_DEV_CHECK(!(kp_ && !slow) || n + __popcll(m & ((1ull << lane) - 1ull)) < n_cap, 20);   // kept index inside the arena
                if (kp_ && !slow) kidx[n + __popcll(m & ((1ull << lane) - 1ull))] = k;
                n += cnt;
            }
        }
        int nout = 0;
        if (!slow) {
            int32_t *members = kidx + n;   // [n] kept indices grouped by cluster, list order inside a cluster (written while clustering)
            uint32_t *lword = reinterpret_cast<uint32_t *>(members + n);
            double *lsum = reinterpret_cast<double *>(arena + (((size_t)n * 12 + 15) & ~(size_t)15));
            double *cen = lsum + n;                                // [n][3] centre joints (RC == 0; in registers otherwise)
            int32_t *csize = reinterpret_cast<int32_t *>(RC > 0 ? lsum + n : cen + 3 * (size_t)n);
            const int ncl_rem = arena_bytes - (int)(reinterpret_cast<char *>(csize) - arena);
            const int ncl_cap = ncl_rem >= 16 ? (ncl_rem - 4) / 12 : 0;   // csize, cseed [ncl_cap], cstart [ncl_cap + 1]
            ASSOC_STAMP(2);
            __syncthreads();
            // ---- per kept candidate: word, score sum, centre joint.  Two candidates per lane and step: their keypoint loads
            // are in flight together.
            double cx[RC > 0 ? RC : 1], cy[RC > 0 ? RC : 1], cz[RC > 0 ? RC : 1];   // RC > 0: centre of candidate 64 r + lane
            auto centre_pair = [&](int ia, int ib, Vec3 &ca, Vec3 &cb) {
                const bool va = ia < n, vb = ib < n;
                const int ka = va ? kidx[ia] : 0, kb = vb ? kidx[ib] : 0;
                int rma, rsa, qa, rmb, rsb, qb;
                slot_rows(ka, rma, rsa, qa);
                slot_rows(kb, rmb, rsb, qb);
                const Kp3<TIn> kma = kpf[(size_t)rma * J + ci], ksa = kpf[(size_t)rsa * J + ci];
                const Kp3<TIn> kmb = kpf[(size_t)rmb * J + ci], ksb = kpf[(size_t)rsb * J + ci];
                const double sa = cs_a[ka], sb = cs_a[kb];
                auto centre = [&](int i, int rm, int rs, int q, const Kp3<TIn> &km, const Kp3<TIn> &ks, double s_) {
                    const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
                    const RayRec a = make_ray(rig.M + 9 * mc, km.u, km.v), b = make_ray(rig.M + 9 * sc, ks.u, ks.v);
                    const double *pc = rig.pairc + 6 * q;
                    const PairSolve o = pair_solve_fast<true>(a, b, Vec3{pc[0], pc[1], pc[2]}, Vec3{pc[3], pc[4], pc[5]});
                    lword[i] = (uint32_t)rm | ((uint32_t)rs << 10) | ((uint32_t)q << 20);
                    lsum[i] = s_;
                    return Vec3{0.5 * o.sw.x, 0.5 * o.sw.y, 0.5 * o.sw.z};
                };
                ca = cb = Vec3{0.0, 0.0, 0.0};
                if (va) ca = centre(ia, rma, rsa, qa, kma, ksa, sa);
                if (vb) cb = centre(ib, rmb, rsb, qb, kmb, ksb, sb);
            };
            if constexpr (RC > 0) {
#pragma unroll
                for (int r = 0; r < RC; r += 2) {
                    cx[r] = cy[r] = cz[r] = 0.0;
                    if (r + 1 < RC) cx[r + 1] = cy[r + 1] = cz[r + 1] = 0.0;
                    if (64 * r < n) {   // wave-uniform
                        Vec3 ca, cb;
                        centre_pair(64 * r + lane, 64 * (r + 1) + lane, ca, cb);
                        cx[r] = ca.x;
                        cy[r] = ca.y;
                        cz[r] = ca.z;
                        if (r + 1 < RC) {
                            cx[r + 1] = cb.x;
                            cy[r + 1] = cb.y;
                            cz[r + 1] = cb.z;
                        }
                    }
                }
            } else {
                for (int i0 = 0; i0 < n; i0 += 128) {
                    Vec3 ca, cb;
                    centre_pair(i0 + lane, i0 + 64 + lane, ca, cb);
                    if (i0 + lane < n) {
                        cen[3 * (i0 + lane)] = ca.x;
                        cen[3 * (i0 + lane) + 1] = ca.y;
                        cen[3 * (i0 + lane) + 2] = ca.z;
                    }
                    if (i0 + 64 + lane < n) {
                        cen[3 * (i0 + 64 + lane)] = cb.x;
                        cen[3 * (i0 + 64 + lane) + 1] = cb.y;
                        cen[3 * (i0 + 64 + lane) + 2] = cb.z;
                    }
                }
            }
            ASSOC_STAMP(3);
            __syncthreads();
            // ---- triangulation.py:107-130 -- seeds in list order, the last candidate never seeds, distance to the SEED's
            // centre, `dist > tol` skips (NaN absorbs).  Which kept candidates are still free is a bit set in REGISTERS: lane c
            // holds the 64 bits of candidates 64 c .. 64 c + 63 (n <= 4096).  Finding the next seed is one ballot + one
            // readlane; a seed's pass over the later candidates skips every 64-block without a free one (after the first
            // persons' seeds most blocks are empty) and writes the absorbed indices straight into `members` -- the clusters come
            // out grouped, in list order, without a second sweep per cluster (16 x 8, 983 kept candidates in 29 clusters:
            // clustering + grouping took 110 of a frame's 172 us, wall-clock stamps).  The comparison is on squared distances;
            // only a value within 1e-14 of the squared tolerance takes the square root the reference takes.
            int ncl = 0;
            int32_t *cseed = csize + ncl_cap;
            int32_t *cstart = cseed + ncl_cap;                    // [ncl + 1]
            const int nblk = (n + 63) >> 6;
            if (ncl_cap < 1 || nblk > 64 || (RC > 0 && nblk > RC)) slow = true;
            unsigned long long free_bits = 0ull;                  // lane c: candidates 64 c + b, bit b
            if (lane < nblk) free_bits = (lane == nblk - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1ull) : ~0ull;
            const double ctol2 = prm.ctol < 0.0 ? -1.0 : prm.ctol * prm.ctol;   // (dist >= 0 > tol: nothing is absorbed but NaN)
            const double ctol2_lo = ctol2 * (1.0 - 1e-14), ctol2_hi = ctol2 * (1.0 + 1e-14);
            auto bits_of = [&](int blk) {                         // lane blk's word, wave-uniform
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(free_bits & 0xffffffffull), blk);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(free_bits >> 32), blk);
                return ((unsigned long long)hi << 32) | lo;
            };
            int off = 0;
            for (int next = 0; !slow;) {
                // the first free candidate at or behind `next` that may seed (index < n - 1)
                unsigned long long mine = free_bits;
                const int blk_l = lane;
                if (blk_l == (next >> 6)) mine &= ~((1ull << (next & 63)) - 1ull);
                if (blk_l < (next >> 6)) mine = 0ull;
                if (blk_l == ((n - 1) >> 6)) mine &= ~(1ull << ((n - 1) & 63));
                const unsigned long long have = __ballot(mine != 0ull);
                if (!have) break;
                const int sblk = __ffsll((long long)have) - 1;
                const unsigned slo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine & 0xffffffffull), sblk);
                const unsigned shi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), sblk);
                const unsigned long long sw = ((unsigned long long)shi << 32) | slo;
                const int mc = (sblk << 6) + __ffsll((long long)sw) - 1;
                if (ncl >= ncl_cap) {
                    slow = true;
                    break;
                }
                SNOWTRI_DEV_CHECK(mc >= 0 && mc < n - 1, 22);   // the seed is a kept candidate, never the last one (:107)
                double mx, my, mz;
                if constexpr (RC > 0) {
                    mx = my = mz = 0.0;
#pragma unroll
                    for (int r = 0; r < RC; r++)
                        if (r == sblk) {   // wave-uniform
                            mx = uniform_lane_f64(cx[r], mc & 63);
                            my = uniform_lane_f64(cy[r], mc & 63);
                            mz = uniform_lane_f64(cz[r], mc & 63);
                        }
                } else {
                    mx = cen[3 * mc];
                    my = cen[3 * mc + 1];
                    mz = cen[3 * mc + 2];
                }
                if (lane == 0) members[off] = mc;
                int cnt = 1;
                // one 64-block of candidates against the seed; (px, py, pz) = the lane's candidate of that block
                auto absorb_block = [&](int blk, double px, double py, double pz) {
                    unsigned long long fb = bits_of(blk);
                    if (blk == sblk) fb &= ~((2ull << (mc & 63)) - 1ull);   // later candidates only (:116)
                    if (fb == 0ull) return;
                    const int sc = (blk << 6) + lane;
                    bool ab = false;
                    if ((fb >> lane) & 1ull) {
                        const double dx = mx - px, dy = my - py, dz = mz - pz;
                        const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                        bool far = d2 > ctol2_hi;
                        if (!far && !(d2 < ctol2_lo)) far = sqrt(d2) > prm.ctol;   // (on the tolerance, or NaN: as :124-125)
                        ab = !far;
                    }
                    const unsigned long long m = __ballot(ab);
                    SNOWTRI_DEV_CHECK(!ab || off + cnt + __popcll(m & ((1ull << lane) - 1ull)) < n, 26);   // member inside the list
                    if (ab) members[off + cnt + __popcll(m & ((1ull << lane) - 1ull))] = sc;
                    cnt += __popcll(m);
                    if (lane == blk) free_bits &= ~m;
                };
                if constexpr (RC > 0) {
#pragma unroll
                    for (int r = 0; r < RC; r++)
                        if (r >= sblk && r < nblk) absorb_block(r, cx[r], cy[r], cz[r]);   // wave-uniform
                } else {
                    for (int blk = sblk; blk < nblk; blk++) {
                        const int sc = (blk << 6) + lane, scc = sc < n ? sc : 0;
                        absorb_block(blk, cen[3 * scc], cen[3 * scc + 1], cen[3 * scc + 2]);
                    }
                }
                if (lane == sblk) free_bits &= ~(1ull << (mc & 63));
                if (lane == 0) {
                    csize[ncl] = cnt;
                    cseed[ncl] = mc;
                    cstart[ncl] = off;
                }
                off += cnt;
                ncl++;
                next = mc + 1;
            }
            if (!slow) {
                if (lane == 0) cstart[ncl] = off;
                __syncthreads();
                ASSOC_STAMP(4);
                // ---- filters and descriptors, cluster by cluster (LDS only)
                uint32_t ncomp = 0, ngen = 0, nwords = 0;
                for (int cid = 0; cid < ncl; cid++) {
                    const int size = csize[cid];
                    if ((double)size < prm.num_tol) continue;                                  // :132-134
                    const int m0 = cstart[cid];
                    double ssum = 0.0;
                    for (int base = 0; base < size; base += 64) ssum += base + lane < size ? lsum[members[m0 + base + lane]] : 0.0;
                    ssum = wave_sum(ssum);
                    const double avg = ssum / ((double)size * (double)(csum_kn ? kn : J));     // :150 from :79
                    // (a sum that is not finite, or a mean within 1e-6 of the tolerance -- the fast sums are within 6e-8 --
                    // is left to k_frame_recompute; a sum of exactly 0 is exact)
                    if (!(fabs(avg) < 1e300) || (ssum != 0.0 && fabs(avg - prm.score_tol) <= 1e-6 * fabs(avg))) {
                        slow = true;
                        break;
                    }
                    if (avg < prm.score_tol) continue;                                         // :151-152
                    if (nout < Pout) {
                        bool complete = false;
                        uint32_t nib_lo = 0u, nib_hi = 0u;
                        if (size == NPq && allow_complete) {
                            // members are in candidate order (camera pair major): in a complete graph member i IS pair i; the
                            // row of camera 0 is the first row of pair (0,1) = member 0, the row of camera c >= 1 the second
                            // row of pair (0,c) = member c - 1, and every member must repeat the rows of its two cameras.
                            for (int c = lane; c < C; c += 64) {
                                const uint32_t w = lword[members[m0 + (c == 0 ? 0 : c - 1)]];
                                rowc[c] = c == 0 ? (int)(w & 1023u) : (int)((w >> 10) & 1023u);
                            }
                            __syncthreads();
                            bool bad = false;
                            for (int base = 0; base < size; base += 64) {
                                const int i = base + lane;
                                if (i < size) {
                                    const uint32_t w = lword[members[m0 + i]];
                                    const int rm = (int)(w & 1023u), rs = (int)((w >> 10) & 1023u), q = (int)(w >> 20);
                                    bad |= !(q == i && rm == rowc[pairs[2 * i]] && rs == rowc[pairs[2 * i + 1]]);
                                }
                            }
                            complete = __ballot(bad) == 0ull;
                            if (complete) {
                                uint32_t lo = 0u, hi = 0u;
                                if (lane < C) {
                                    const uint32_t p = (uint32_t)(rowc[lane] - lane * Pmax);
                                    if (lane < 8)
                                        lo = p << (4 * lane);
                                    else
                                        hi = p << (4 * (lane - 8));
                                }
#pragma unroll
                                for (int o = 8; o > 0; o >>= 1) {
                                    lo |= (uint32_t)__shfl_xor((int)lo, o, 64);
                                    hi |= (uint32_t)__shfl_xor((int)hi, o, 64);
                                }
                                nib_lo = (uint32_t)__shfl((int)lo, 0, 64);
                                nib_hi = (uint32_t)__shfl((int)hi, 0, 64);
                            }
                            __syncthreads();
                        }
                        if (lane == 0) {
                            st_a[nout] = complete ? nib_lo : (uint32_t)m0;
                            st_size[nout] = complete ? 0 : size;
                            st_idx[nout] = complete ? ncomp : ngen;
                            st_word[nout] = complete ? nib_hi : nwords;
                            st_avg[nout] = avg;
                        }
                        if (complete) {
                            ncomp++;
                        } else {
                            ngen++;
                            nwords += (uint32_t)size;
                        }
                    }
                    nout++;
                }
                ASSOC_STAMP(5);
                if (!slow) {
                    // lanes 0 and 1 reserve room in the lists at once: one atomic per counter (member descriptors and member
                    // words share a word), the counters on different memory channels
                    const unsigned long long want = lane == 0 ? (unsigned long long)ncomp
                                                              : (lane == 1 ? ((unsigned long long)ngen << 32) | (unsigned long long)nwords : 0ull);
                    unsigned long long got = 0ull;
                    if (want) got = atomicAdd(hand_counters + (lane == 0 ? kHandComplete : kHandMembers), want);
                    // the round trip of that device-scope atomic is ~16 us under thousands of waves (wall-clock stamps): the
                    // zero-fill of the unused slots, which needs nothing of it, goes out in its shadow
                    if (!prm.no_zero_fill) {
                        for (int i = lane; i < (Pout - nout) * kn; i += 64) wr.zero_joint(f, Pout, kn, nout + i / kn, i % kn);
                        for (int slot = nout + lane; slot < Pout; slot += 64) wr.person(f, Pout, slot, 0.0);
                    }
                    zero_filled = true;
                    const unsigned long long bc = (unsigned long long)__shfl((long long)got, 0, 64), gm = (unsigned long long)__shfl((long long)got, 1, 64),
                                             bg = hand_member_descs(gm), bw = hand_member_words(gm);
                    // (cannot happen: the host sizes the lists for Pout persons and Kc members of every frame)
                    if (bc + ncomp > (unsigned long long)desc_cap || bg + ngen > (unsigned long long)desc_cap ||
                        bw + nwords > (unsigned long long)word_cap) {
                        for (unsigned long long i = bc + lane; i < bc + ncomp && i < (unsigned long long)desc_cap; i += 64)
                            desc[i] = ClusterDesc{0u, 0u, 0xffffffffu, 0u};
                        for (unsigned long long i = bg + lane; i < bg + ngen && i < (unsigned long long)desc_cap; i += 64)
                            desc[(unsigned long long)desc_cap + i] = ClusterDesc{0u, 0u, 0xffffffffu, 0u};
                        slow = true;
                    }
                    ASSOC_STAMP(6);
                    __syncthreads();
                    if (!slow) {
                        const int nsl = nout < Pout ? nout : Pout;
                        for (int sl = lane; sl < nsl; sl += 64) {
                            SNOWTRI_DEV_CHECK((st_size[sl] == 0 ? (uint32_t)bc : (uint32_t)bg) + st_idx[sl] < desc_cap, 23);   // descriptor inside its list
                            SNOWTRI_DEV_CHECK(st_size[sl] == 0 || (unsigned long long)bw + st_word[sl] + (unsigned)st_size[sl] <= (unsigned long long)word_cap, 24);
                            if (st_size[sl] == 0)
                                desc[(uint32_t)bc + st_idx[sl]] = ClusterDesc{(uint32_t)f, st_a[sl], (uint32_t)sl, st_word[sl]};
                            else
                                desc[desc_cap + (uint32_t)bg + st_idx[sl]] =
                                    ClusterDesc{(uint32_t)f, (uint32_t)bw + st_word[sl], (uint32_t)sl, (uint32_t)st_size[sl]};
                            if (write_person_scores) wr.person(f, Pout, sl, st_avg[sl]);
                        }
                        for (int sl = 0; sl < nsl; sl++) {
                            const int size = st_size[sl], m0 = (int)st_a[sl];
                            for (int i = lane; i < size; i += 64) hand_words[(uint32_t)bw + st_word[sl] + (uint32_t)i] = lword[members[m0 + i]];
                        }
                    }
                }
            }
        }
        if (slow) {
            if (lane == 0) {
                const unsigned long long at = atomicAdd(slow_count, 1ull);
                SNOWTRI_DEV_CHECK(at < (unsigned long long)F, 25);   // the list holds one entry per frame of the segment
                slow_list[at] = (uint32_t)f;
                out_count[f] = -1;   // (k_person_scores skips the frame; whoever re-does it writes the count)
            }
            continue;
        }
        ASSOC_STAMP(7);
        // unused slots: one flat sweep of 16-byte stores (already out for a frame that reserved list room)
        if (!zero_filled && !prm.no_zero_fill) {
            for (int i = lane; i < (Pout - nout) * kn; i += 64) wr.zero_joint(f, Pout, kn, nout + i / kn, i % kn);
            for (int slot = nout + lane; slot < Pout; slot += 64) wr.person(f, Pout, slot, 0.0);
        }
        if (lane == 0) {
            out_count[f] = nout;
            if (out_flags && nout > Pout) atomicOr(&out_flags[f], 2u /*SNOWTRI_FLAG_OVERFLOW*/);
        }
        ASSOC_STAMP(8);
    }
}

// ------------------------------------------------------------------------------------------- k_cluster_members
// The member-list descriptors alone (cluster_member_passes of snowtri_cluster.hpp) for rigs k_cluster_fuse has no
// instantiation for: any camera count up to kPairTabMaxPairs pairs.  Dynamic LDS: cluster_members_lds_bytes(C, npairs).
__host__ __device__ inline size_t cluster_members_lds_bytes(int C, int npairs) { return (size_t)72 * C + (size_t)56 * npairs + 16; }

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock) void k_cluster_members(const ClusterDesc *__restrict__ desc, const uint32_t *__restrict__ words,
                                                            const unsigned long long *__restrict__ cnt, uint32_t desc_cap, Rig rig,
                                                            const TIn *__restrict__ kpts, Params prm, int Pmax, int J, int kn,
                                                            unsigned long long kmagic, int Pout, TOut *__restrict__ out4,
                                                            uint32_t *__restrict__ out_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = rig.C, NP = rig.npairs, tid = threadIdx.x;
    double *Ml = reinterpret_cast<double *>(smem);
    double *pc = Ml + 9 * C;
    int32_t *pairs_l = reinterpret_cast<int32_t *>(pc + 6 * NP);
    for (int i = tid; i < 9 * C; i += kBlock) Ml[i] = rig.M[i];
    for (int i = tid; i < 6 * NP; i += kBlock) pc[i] = rig.pairc[i];
    for (int i = tid; i < 2 * NP; i += kBlock) pairs_l[i] = rig.pairs[i];
    const unsigned long long ng64 = hand_member_descs(cnt[kHandMembers]);
    const uint32_t ngen = ng64 < (unsigned long long)desc_cap ? (uint32_t)ng64 : desc_cap;
    __syncthreads();
    const uint32_t W = gridDim.x * (uint32_t)(kBlock / 64);
    cluster_member_passes<TIn, TOut>(desc + desc_cap, ngen, words, Ml, pc, pairs_l, C * Pmax, reinterpret_cast<const Kp3<TIn> *>(kpts), prm, J, kn,
                                     kmagic, Pout, out4, blockIdx.x * (uint32_t)(kBlock / 64) + (uint32_t)(tid >> 6), W, out_flags);
}

// ---------------------------------------------------------------------------------------------------- k_person_scores
// out_ps[f][slot] = mean over the keypoint_num fused joint scores of the person (:150), for every used slot of every frame
// the streaming route finished (out_count >= 0; the frames it left behind get theirs from k_frame_recompute).  Used when
// the association kernel cannot derive the mean from the candidate sums: keypoint_num < J (the sums run over all J joints,
// :79) and float64 outputs (the sums carry the raw v_rsq_f64).  One wave per (frame, slot), lanes = joints, fixed order.
template <typename TOut>
__global__ __launch_bounds__(kBlock) void k_person_scores(int64_t F, int Pout, int kn, const TOut *__restrict__ out4, TOut *__restrict__ out_ps,
                                                          const int32_t *__restrict__ out_count) {
    const int lane = threadIdx.x & 63;
    const int64_t W = (int64_t)gridDim.x * (kBlock / 64), total = F * Pout;
    for (int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < total; i += W) {
        // slot-major items (slot = i / F): with frame-major items and a wave count that is a multiple of Pout a wave met the
        // same slot in every round -- the waves of the unused slots idle, those of slot 0 do all the work
        const int slot = (int)(i / F);
        const int64_t f = i - (int64_t)slot * F;
        const int n = out_count[f];
        if (slot >= n) continue;   // (n < 0: a frame left to k_frame_recompute)
        const int64_t rec = f * Pout + slot;
        const TOut *row = out4 + (size_t)rec * kn * 4;
        double v = 0.0;   // lane-strided partial sums in the order of the one-load loop; four loads in flight (+ 0.0 is exact)
        for (int b0 = lane; b0 < kn; b0 += 256) {
            TOut a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = b0 + 64 * u < kn ? row[4 * (b0 + 64 * u) + 3] : (TOut)0;
#pragma unroll
            for (int u = 0; u < 4; u++) v += (double)a[u];
        }
        v = wave_sum(v);
        if (lane == 0) out_ps[rec] = (TOut)(v / (double)kn);
    }
}

// ---------------------------------------------------------------------------------------------------- k_singular_scan
// The streaming route WITHOUT its candidate pass (one detection per camera, `sumless` in launch_frame_recompute) flags a
// singular pair where its joint is FUSED -- which leaves out what the reference solves and this route never touches: the joints
// behind keypoint_num, and the candidates outside every kept cluster (the lone last candidate of :107, clusters dropped by
// condense_person_num_tol).  The reference's np.linalg.inv raises for those too (triangulation.py:26 runs for every listed
// pair and every joint), and with the candidate pass the same frames ARE flagged: the flag must not depend on the route
// (round-5 advice).  This scan is the candidate pass reduced to that one question -- lane = (frame, joint), every pair of
// listed cameras, rays and products exactly as skew_ray_solve forms them: a c == b b in separately rounded products.
template <typename TIn>
__global__ __launch_bounds__(kBlock) void k_singular_scan(int64_t F, int J, Rig rig, const TIn *__restrict__ kpts,
                                                          const int32_t *__restrict__ n_persons, uint32_t *__restrict__ out_flags) {
    const int C = rig.C;
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    const int64_t total = F * (int64_t)J;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t f = i / J;
        const int j = (int)(i - f * J);
        bool sing = false;
        for (int m = 0; m + 1 < C; m++) {
            if (n_persons && n_persons[f * C + m] < 1) continue;
            const Kp3<TIn> km = kp3[(f * C + m) * J + j];
            const RayRec a = make_ray(rig.M + 9 * m, km.u, km.v);
            for (int s = m + 1; s < C; s++) {
                if (n_persons && n_persons[f * C + s] < 1) continue;
                const Kp3<TIn> ks = kp3[(f * C + s) * J + j];
                const RayRec b = make_ray(rig.M + 9 * s, ks.u, ks.v);
                const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
                sing |= a.a * b.a == bq * bq;
            }
        }
        if (sing) atomicOr(&out_flags[f], 1u /*SNOWTRI_FLAG_SINGULAR*/);
    }
}

}  // namespace snowtri

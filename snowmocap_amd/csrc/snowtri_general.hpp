// snowtri_general.hpp -- k_frame_recompute: the reference's full algorithm (any person count, any
// thresholds) for one frame per workgroup WITHOUT materialising candidates in HBM.
//
// The reference separates A3 and A4 by a candidate list of Kc x J x 32 B per frame (1.9 MB for
// 8 cameras x 4 persons, 32 MB for 16 x 8): spilling it makes the multi-person configs HBM-bound
// at ~7x their arithmetic cost.  Here every pair solve is recomputed instead (SURVEY.md §8a note):
//
//   phase 1  per candidate: mean of the J joint scores -> keep flag        (triangulation.py:70-81)
//            rays of a joint chunk are built once into LDS; one lane per candidate walks the chunk
//   phase 2  kept list (ordered), centre joints, greedy clustering (one wave) (triangulation.py:107-130)
//   phase 3  per surviving cluster and joint: sum s, sum s*(Wm+Ws) over the members, recomputing
//            their solves; lanes = joints (coalesced pixel reads), members split over the 4 waves
//                                                                          (triangulation.py:136-152)
// Only O(Kc) bookkeeping (57 B per candidate slot) lives in a per-workgroup scratch slab.
#pragma once
#include "snowtri_fused.hpp"

namespace snowtri {

constexpr int kRecomputeMaxKn = 256;          // joints handled per lane in phase 3: lane + 64 p, p < 4
constexpr int kRayChunkBytes = 32 * 1024;     // LDS budget for one chunk of rays

__host__ __device__ constexpr size_t recompute_scratch_bytes(int64_t Kc, int R) {
    return (((size_t)Kc * 64) + (size_t)R * 8 + 1024 + 255) & ~(size_t)255;
}

__host__ __device__ inline int recompute_chunk_joints(int R, int J, int score_bytes) {
    int jc = kRayChunkBytes / (R * (32 + score_bytes));
    return jc < 1 ? 0 : (jc > J ? J : jc);
}

__host__ __device__ inline size_t recompute_lds_bytes(int R, int J, int kn, int score_bytes) {
    const int jc = recompute_chunk_joints(R, J, score_bytes);
    return (size_t)jc * R * (32 + score_bytes) + (size_t)(kBlock / 64) * kn * 32 + 256;
}

// Dynamic LDS = recompute_lds_bytes(R, J, kn, sizeof(TIn)); scratch = gridDim.x slabs of
// recompute_scratch_bytes(Kc).  R = C * Pmax ray rows.  Requires keypoint_num <= kRecomputeMaxKn.
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
                                                            unsigned long long *next_frame) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = rig.C, R = C * Pmax, pp = Pmax * Pmax;
    const int kn = prm.kn, ci = prm.center;
    const int Jc = recompute_chunk_joints(R, J, (int)sizeof(TIn));
    RayRec *rays = reinterpret_cast<RayRec *>(smem);                                   // [Jc][R]
    TIn *rsc = reinterpret_cast<TIn *>(rays + (size_t)Jc * R);                          // [Jc][R]
    double *partial = reinterpret_cast<double *>(smem + (size_t)Jc * R * (32 + sizeof(TIn)));  // [4][kn][4]
    partial = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(partial) + 15) & ~(uintptr_t)15);
    double *red = partial + (size_t)(kBlock / 64) * kn * 4;                              // [4] + misc
    int32_t *misc = reinterpret_cast<int32_t *>(red + kBlock / 64);

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
    int32_t *rowlist = reinterpret_cast<int32_t *>(slab + (((size_t)Kc * 64 + 1024) & ~(size_t)7));  // [R] (DLT)
    int32_t *rowflag = rowlist + R;                                                                   // [R] (DLT)

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
        for (int k = tid; k < Kc; k += kBlock) sum[k] = 0.0;
        if (tid == 0 && out_flags) out_flags[f] = 0u;
        bool sing = false;

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
                    rays[jj * R + r] = make_ray(rig.M + 9 * c, kp.u, kp.v);
                    rsc[jj * R + r] = kp.s;
                }
            }
            __syncthreads();
            for (int k = tid; k < Kc; k += kBlock) {
                const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
                const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
                const int nm = np_f ? np_f[mc] : Pmax, ns = np_f ? np_f[sc] : Pmax;
                if (pm >= nm || ps >= ns) continue;
                const int rm = mc * Pmax + pm, rs = sc * Pmax + ps;
                const double *pc = rig.pairc + 6 * q;
                const Vec3 d = {pc[0], pc[1], pc[2]}, tsum = {0.0, 0.0, 0.0};
                double acc = 0.0;
#pragma unroll(kRecomputeUnroll)
                for (int jj = 0; jj < nj; jj++) {
                    const RayRec a = rays[jj * R + rm], b = rays[jj * R + rs];
                    const TIn sm = rsc[jj * R + rm], ss = rsc[jj * R + rs];
                    const PairSolve o = pair_solve_fast<false>(a, b, d, tsum);
                    const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(o.d2 > prm.dthr2);  // :73-74
                    acc += gated_sum(sm, ss, kp_) * (0.5 * o.score_base);                              // :72
                    sing |= o.singular;
                }
                sum[k] += acc;
            }
        }
        __syncthreads();
        if (sing && out_flags) atomicOr(&out_flags[f], 1u /*SNOWTRI_FLAG_SINGULAR*/);
        for (int k = tid; k < Kc; k += kBlock) {
            const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
            const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
            const int nm = np_f ? np_f[mc] : Pmax, ns = np_f ? np_f[sc] : Pmax;
            const bool valid = pm < nm && ps < ns;
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
        for (int i = tid; i < n; i += kBlock) {
            const int k = kidx[i];
            const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
            const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
            const Kp3<TIn> km = kpf[(size_t)(mc * Pmax + pm) * J + ci], ks = kpf[(size_t)(sc * Pmax + ps) * J + ci];
            const RayRec a = make_ray(rig.M + 9 * mc, km.u, km.v), b = make_ray(rig.M + 9 * sc, ks.u, ks.v);
            const double *pc = rig.pairc + 6 * q;
            const PairSolve o = pair_solve_fast<true>(a, b, Vec3{pc[0], pc[1], pc[2]}, Vec3{pc[3], pc[4], pc[5]});
            centre[3 * i] = 0.5 * o.sw.x;
            centre[3 * i + 1] = 0.5 * o.sw.y;
            centre[3 * i + 2] = 0.5 * o.sw.z;
            cluster_of[i] = -1;
        }
        __syncthreads();
        if (tid < 64) {
            // triangulation.py:107-130 -- seeds in list order, the last candidate never seeds,
            // distance to the SEED's centre, `dist > tol` skips (NaN absorbs)
            int ncl = 0;
            for (int mc = 0; mc < n - 1; mc++) {
                if (cluster_of[mc] != -1) continue;
                const double mx = centre[3 * mc], my = centre[3 * mc + 1], mz = centre[3 * mc + 2];
                int cnt = 0;
                for (int base = mc + 1; base < n; base += 64) {
                    const int sc = base + lane;
                    bool ab = false;
                    if (sc < n && cluster_of[sc] == -1) {
                        const double dx = mx - centre[3 * sc], dy = my - centre[3 * sc + 1], dz = mz - centre[3 * sc + 2];
                        const double dist = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
                        if (!(dist > prm.ctol)) {
                            cluster_of[sc] = ncl;
                            ab = true;
                        }
                    }
                    cnt += __popcll(__ballot(ab));
                }
                if (lane == 0) {
                    cluster_of[mc] = ncl;
                    csize[ncl] = cnt + 1;
                    cseed[ncl] = mc;
                }
                ncl++;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            // members grouped by cluster, list order inside each cluster
            int off = 0;
            for (int c = 0; c < ncl; c++) {
                if (lane == 0) cstart[c] = off;
                for (int base = cseed[c]; base < n; base += 64) {
                    const int i = base + lane;
                    const bool in = i < n && cluster_of[i] == c;
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
        __syncthreads();
        const int ncl = misc[1];

        // ---------------- phase 3: fusion per surviving cluster ------------------------------
        int nout = 0;
        for (int cid = 0; cid < ncl; cid++) {
            const int size = csize[cid];
            if ((double)size < prm.num_tol) continue;                                             // :132-134
            const int m0 = cstart[cid];
            double ox = 0.0, oy = 0.0, oz = 0.0, os = 0.0;
            if constexpr (METHOD == 1) {
                // distinct observation rows of this cluster, in row order
                for (int r = tid; r < R; r += kBlock) rowflag[r] = 0;
                __syncthreads();
                for (int mi = tid; mi < size; mi += kBlock) {
                    const int k = kidx[members[m0 + mi]];
                    const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
                    rowflag[rig.pairs[2 * q] * Pmax + pm] = 1;
                    rowflag[rig.pairs[2 * q + 1] * Pmax + ps] = 1;
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
            } else {
                double aS[4] = {0, 0, 0, 0}, aX[4] = {0, 0, 0, 0}, aY[4] = {0, 0, 0, 0}, aZ[4] = {0, 0, 0, 0};
                for (int mi = wave; mi < size; mi += kBlock / 64) {
                    const int k = kidx[members[m0 + mi]];
                    const int q = k / pp, rr = k - q * pp, pm = rr / Pmax, ps = rr - pm * Pmax;
                    const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
                    const Kp3<TIn> *rowm = kpf + (size_t)(mc * Pmax + pm) * J, *rows = kpf + (size_t)(sc * Pmax + ps) * J;
                    const double *pc = rig.pairc + 6 * q;
                    const Vec3 d = {pc[0], pc[1], pc[2]}, tsum = {pc[3], pc[4], pc[5]};
                    const double *Mm = rig.M + 9 * mc, *Ms = rig.M + 9 * sc;
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int b = lane + 64 * p;
                        if (b < kn) {
                            const Kp3<TIn> km = rowm[b], ks = rows[b];
                            const PairSolve o = pair_solve_fast<true>(make_ray(Mm, km.u, km.v), make_ray(Ms, ks.u, ks.v), d, tsum);
                            const bool kp_ = !below_kthr(km.s, prm) && !below_kthr(ks.s, prm) && !(o.d2 > prm.dthr2);
                            const double s = gated_sum(km.s, ks.s, kp_) * (0.5 * o.score_base);
                            aS[p] += s;                                                                // :141
                            aX[p] = fma(s, o.sw.x, aX[p]);                                             // :144-147
                            aY[p] = fma(s, o.sw.y, aY[p]);
                            aZ[p] = fma(s, o.sw.z, aZ[p]);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int b = lane + 64 * p;
                    if (b < kn) {
                        double *dst = partial + ((size_t)wave * kn + b) * 4;
                        dst[0] = aS[p];
                        dst[1] = aX[p];
                        dst[2] = aY[p];
                        dst[3] = aZ[p];
                    }
                }
                __syncthreads();
                if (tid < kn) {
                    double S = 0.0, X = 0.0, Y = 0.0, Z = 0.0;
#pragma unroll
                    for (int w = 0; w < kBlock / 64; w++) {
                        const double *src = partial + ((size_t)w * kn + tid) * 4;
                        S += src[0];
                        X += src[1];
                        Y += src[2];
                        Z += src[3];
                    }
                    if (!(S == 0.0)) {                                                                 // :142-143
                        const double r = 0.5 / S;
                        ox = X * r;
                        oy = Y * r;
                        oz = Z * r;
                        os = S / (double)size;                                                         // :148
                    }
                }
            }
            const double avg = block_sum(tid < kn ? os : 0.0, red) / (double)kn;                   // :150
            if (!(avg < prm.score_tol)) {                                                          // :151-152
                if (nout < Pout) {
                    if (tid < kn) wr.joint(f, Pout, kn, nout, tid, ox, oy, oz, os);
                    if (tid == 0) wr.person(f, Pout, nout, avg);
                }
                nout++;
            }
            __syncthreads();
        }
        for (int slot = nout; slot < Pout; slot++) {
            for (int b = tid; b < kn; b += kBlock) wr.joint(f, Pout, kn, slot, b, 0.0, 0.0, 0.0, 0.0);
            if (tid == 0) wr.person(f, Pout, slot, 0.0);
        }
        if (tid == 0) {
            out_count[f] = nout;
            if (out_flags && nout > Pout) atomicOr(&out_flags[f], 2u /*SNOWTRI_FLAG_OVERFLOW*/);
        }
        __syncthreads();
    }
}

}  // namespace snowtri

// snowtri_sums_rays.hpp -- k_candidate_sums_rays: the candidate pass (A3, triangulation.py:56-81) for rigs of EXACTLY 32 rays per
// frame (C cameras x P persons = 32: the 8 x 4 of BASELINE configs[2], 16 x 2, 4 x 8), with every lane of the workgroup at work.
//
// k_candidate_sums (snowtri_assoc.hpp) deals TILES of a camera pair -- 2 persons of its first camera x 4 of its second -- to the
// lanes: 8 x 4 has 28 pairs x 2 = 56 tiles, and 56 tiles x 4 joint sub-ranges fill 224 of 256 lanes; 448 candidates = 7 x 64 --
// no pair-wise grouping fills a wave (round-4 review: "56 of 64 lanes").  Here a lane owns a RAY (camera a, person p): lane = joint
// sub-range x 32 + ray, 8 sub-ranges x 32 rays = 256 lanes, and every unordered camera pair is solved exactly once by walking
// the second camera CYCLICALLY (as k_cluster_fuse_wide does): the lane's ray against all P persons of cameras
// a + 1 .. a + (C - 1) / 2 (mod C), and -- even C -- half of the persons of camera a + C / 2: the candidates of that pair whose
// two person indices have equal parity belong to the lane of the pair's first camera, the others to the lane of its second.
// Every lane: P (C - 1) / 2 candidates, 14 at 8 x 4, per joint of its sub-range; per frame 133 / 8 joints.
//
// Arithmetic per candidate: the distance-only solve of p1_tile_sums with the cross product d x a taken on the lane's OWN ray,
// also where that ray is the pair's second one (n = d . (a x b) changes sign only; its rounding differs in the last bit from
// the first-ray form, which the users of these sums -- the filter of :80-81 with its 1e-6 guard band, a float32 mean score --
// do not see).  A frame in which a camera lists fewer than P persons takes a plain loop over the candidate slots instead.
// Same outputs, hand-shake (ticket counter, exact list, flags) and LDS record layout as k_candidate_sums.
//
// MEASURED (round 5, EXPERIMENTS.md) and NOT the default: 8 x 4 x 10 000 frames 831 us against the tile kernel's 710 us.  The
// kernel issues 14 % fewer VALU instructions per frame (74 472 against 86 671 per wave) but keeps the SIMDs busy 54 % of the
// time against 77 %: a lane solves two joints between two barriers, each behind 75 LDS reads, and at three waves per SIMD
// (168 registers: 28 of sums, 24 of pair offsets, one set of partner records) nothing hides them; a second register set
// for the next tile's records spills inside the loop.  SNOWTRI_SUMS_RAYS=1 selects it (tests/test_gpu_handover.py).
//
// Fill: thread t owns joint t % 16 of rows t / 16 and t / 16 + 16 of every chunk of 16 joints: no index arithmetic per
// record, and the 16 lanes of an LDS write group stay inside one row (consecutive joints = consecutive banks): the write
// groups of k_candidate_sums straddled a row every 20 joints, the source of its bank-conflict cycles (0.30 of the active LDS
// cycles at 8 x 4, round-4 review).
#pragma once
#include "snowtri_assoc.hpp"

namespace snowtri {

constexpr int kRaysThreads = 256;
constexpr int kRaysRows = 32;                     // rays per frame = C * P
constexpr int kRaysJS = kRaysThreads / kRaysRows;  // joint sub-ranges
constexpr int kRaysJPS = 2;                       // joints per sub-range and chunk
constexpr int kRaysJc = kRaysJS * kRaysJPS;       // joints per chunk (16)
constexpr int kRaysWaves = 3;                     // waves per SIMD the registers must allow

__host__ __device__ constexpr size_t rays_arena_offset(int C, int npairs) {
    return ((size_t)kSumsHeadBytes + (size_t)4 * (C + (C & 1)) + (size_t)8 * npairs + (size_t)72 * C + 15) & ~(size_t)15;
}
__host__ __device__ constexpr int rays_buffer_bytes() { return kRaysJc * (kP1Rec * kRaysRows + 8); }
__host__ __device__ constexpr size_t rays_lds_bytes(int C, int npairs) { return rays_arena_offset(C, npairs) + 2 * (size_t)rays_buffer_bytes(); }

template <typename TIn, int C, int P>
__global__ __launch_bounds__(kRaysThreads, kRaysWaves) void k_candidate_sums_rays(
    int64_t F, int J, int Jrow, Rig rig, const TIn *__restrict__ kpts, const int32_t *__restrict__ n_persons, Params prm,
    double *__restrict__ csum, uint32_t *__restrict__ out_flags, uint32_t *__restrict__ exact_list, unsigned long long *exact_count,
    unsigned long long *next_frame) {
    static_assert(C * P == kRaysRows, "one lane per ray and joint sub-range: C * P == 32");
    static_assert(C % 2 == 1 || P % 2 == 0, "the half partner of an even rig splits its persons by parity");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int B = kRaysThreads, R = kRaysRows, JS = kRaysJS, Jc = kRaysJc, NP = C * (C - 1) / 2, pp = P * P, Kc = NP * pp;
    constexpr int H = (C - 1) / 2;                  // partner cameras a lane takes with all their persons
    constexpr bool kHalf = C % 2 == 0;              // + half the persons of camera a + C / 2
    constexpr int NT = H + (kHalf ? 1 : 0);         // tiles of a lane
    constexpr int PH = P / 2;
    constexpr int NACC = P * H + (kHalf ? PH : 0);  // candidates of a lane
    constexpr int jstr = kP1Rec * R + 8;
    constexpr int half = rays_buffer_bytes();
    constexpr int NPF = R * Jc / B;                 // records a thread fetches per chunk (2)
    static_assert(NPF * B == R * Jc && B % Jc == 0, "fixed (row, joint) per thread");
    static_assert(Kc <= 2 * B && 4 * Kc * 8 <= half, "ragged frames: two slots per thread; the waves' partial sums fit a chunk buffer");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    (void)lane;

    int32_t *head = reinterpret_cast<int32_t *>(smem);
    int32_t *np_l = reinterpret_cast<int32_t *>(smem + kSumsHeadBytes);           // [C]
    int32_t *pairs = np_l + C + (C & 1);                                          // [NP][2]
    double *Ml = reinterpret_cast<double *>(pairs + 2 * NP);                      // [C][9]
    char *const rec0 = smem + rays_arena_offset(C, NP);
    if (tid < kSumsHeadBytes / 4) head[tid] = 0;
    for (int i = tid; i < 2 * NP; i += B) pairs[i] = rig.pairs[i];
    for (int i = tid; i < 9 * C; i += B) Ml[i] = rig.M[i];
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);

    // ---- the lane's ray, joint sub-range and tiles: fixed for the whole launch
    const int ray = tid & (R - 1), jsub = tid / R;
    const int cam = ray / P, per = ray - cam * P;
    int t_ob[NT];      // byte offset of the tile's first partner record inside a joint's records
    int t_k0[NT];      // candidate slot of the tile's partner person 0 ...
    int t_ks[NT];      // ... and the slot step per partner person
    Vec3 t_d[NT];      // the pair's d = t_s - t_m (the sign does not matter: n enters squared)
#pragma unroll
    for (int ti = 0; ti < NT; ti++) {
        const int sc = (cam + ti + 1) % C;     // ti < H: every person; ti == H: the half partner, delta = C / 2
        const int lo = cam < sc ? cam : sc, hi = cam < sc ? sc : cam;
        const int q = lo * C - lo * (lo + 1) / 2 + (hi - lo - 1);   // pair (lo, hi) in the order of triangulation.py:56-57
        SNOWTRI_DEV_CHECK(q >= 0 && q < NP && rig.pairs[2 * q] == lo && rig.pairs[2 * q + 1] == hi, 40);
        int first = 0;                         // first partner person of the tile
        if (kHalf && ti == H) first = cam < sc ? (per & 1) : 1 - (per & 1);   // equal parity -> first camera's lane, else the second's
        t_ob[ti] = kP1Rec * (sc * P + first);
        // own ray first in the pair: slot q pp + per P + partner; own ray second: slot q pp + partner P + per
        t_ks[ti] = cam < sc ? 1 : P;
        t_k0[ti] = q * pp + (cam < sc ? per * P : per) + first * t_ks[ti];
        t_d[ti] = Vec3{rig.pairc[6 * q], rig.pairc[6 * q + 1], rig.pairc[6 * q + 2]};
    }
    const int oa = kP1Rec * ray;

    // ---- fill: thread t holds joint t % Jc of rows t / Jc + (B / Jc) n of a chunk
    const int fj = tid % Jc, frow0 = tid / Jc;
    Kp3<TIn> pre[NPF];
    int pre_nj = 0;    // joints of the chunk in `pre`
    int npv = P;       // threads < C: n_persons of the coming frame
    auto fetch = [&](const Kp3<TIn> *kpf, int j0, int nj) {
        const int j = j0 + (fj < nj ? fj : nj - 1);   // (every thread loads: a load inside a branch is waited for where the branch joins)
        SNOWTRI_DEV_CHECK(j >= 0 && j < J, 41);
#pragma unroll
        for (int n = 0; n < NPF; n++) pre[n] = kpf[(size_t)(frow0 + (B / Jc) * n) * Jrow + j];
        pre_nj = nj;
    };
    auto fetch_frame = [&](int64_t fr) {
        fetch(kp3 + fr * (int64_t)R * Jrow, 0, J < Jc ? J : Jc);
        if (n_persons && tid < C) npv = n_persons[fr * C + tid];
    };
    auto commit = [&](char *buf) {
        if (fj < pre_nj) {
#pragma unroll
            for (int n = 0; n < NPF; n++) {
                const int row = frow0 + (B / Jc) * n;
                p1_store_record<TIn>(buf + fj * jstr + kP1Rec * row, make_ray(Ml + 9 * (row / P), pre[n].u, pre[n].v), pre[n].s);
            }
        }
    };

    const int nch = (J + Jc - 1) / Jc;
    int64_t f = blockIdx.x;
    if (f < F) fetch_frame(f);
    int par = 0;
    __syncthreads();   // constants and the cleared head
    for (int it = 0; f < F; it++) {
        int32_t *hd = head + 4 * (it & 1);
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * Jrow;
        double *cs_f = csum + f * (int64_t)Kc;
        if (tid == 0) hd[2] = (int32_t)atomicAdd(next_frame, 1ull);
        if (tid < C) {
            np_l[tid] = npv;
            if (npv != P) hd[1] = 1;
        }
        commit(rec0 + par * half);
        if (nch >= 2) fetch(kpf, Jc, (J - Jc) < Jc ? (J - Jc) : Jc);
        __syncthreads();   // first chunk, np_l, ragged flag and ticket are there
        const int64_t fnext = (int64_t)gridDim.x + (int64_t)(uint32_t)hd[2];
        if (nch == 1 && fnext < F) fetch_frame(fnext);
        const bool ragged = hd[1] != 0;   // workgroup-uniform
        bool redo = false;
        auto finish = [&](int k, double v) {   // the 1 / (2 * 1000) of :72, and whether a mean can decide :80-81
            const double s_ = v * 0.0005, mean = s_ / (double)J;
            cs_f[k] = s_;
            redo |= exact_list != nullptr && (!(fabs(mean) < 1e300) || (s_ != 0.0 && fabs(mean - prm.avg_thr) <= 1e-6 * fabs(mean)));
        };
        auto advance = [&](int c, int j0) {    // behind the solves of chunk c: the next chunk's records, the one after's keypoints
            if (c + 1 < nch) {
                commit(rec0 + ((par ^ (c + 1)) & 1) * half);
                if (c + 2 < nch)
                    fetch(kpf, j0 + 2 * Jc, (J - j0 - 2 * Jc) < Jc ? (J - j0 - 2 * Jc) : Jc);
                else if (fnext < F)
                    fetch_frame(fnext);
            }
        };

        if (!ragged) {
            double tot[NACC];
#pragma unroll
            for (int u = 0; u < NACC; u++) tot[u] = 0.0;
            for (int c = 0, j0 = 0; c < nch; c++, j0 += Jc) {
                const int nj = (J - j0) < Jc ? (J - j0) : Jc;
                const char *cur = rec0 + ((par ^ c) & 1) * half;
                // the sub-ranges rotate from chunk to chunk: the short last chunk lands on other lanes than the one before
                const int jrot = (jsub + c) & (JS - 1);
                const int jlo = jrot * kRaysJPS;
                const int cnt = nj - jlo < kRaysJPS ? (nj - jlo < 0 ? 0 : nj - jlo) : kRaysJPS;
                const char *pj = cur + jlo * jstr;
                for (int t = 0; t < cnt; t++, pj += jstr) {
                    const RayRec a = p1_load_ray(pj + oa);
                    const TIn sm = p1_load_score<TIn>(pj + oa);
                    const bool okm = !below_kthr(sm, prm);
#pragma unroll
                    for (int ti = 0; ti < NT; ti++) {
                        const Vec3 d = t_d[ti];
                        const double cx = fma(d.y, a.z, -(d.z * a.y)), cy = fma(d.z, a.x, -(d.x * a.z)), cz = fma(d.x, a.y, -(d.y * a.x));
                        RayRec b[P];
                        TIn ss[P];
#pragma unroll
                        for (int u = 0; u < P; u++)
                            if (u < ((kHalf && ti == H) ? PH : P)) {   // (compile-time after unrolling)
                                const char *pb = pj + t_ob[ti] + kP1Rec * ((kHalf && ti == H) ? 2 * u : u);
                                b[u] = p1_load_ray(pb);
                                ss[u] = p1_load_score<TIn>(pb);
                            }
#pragma unroll
                        for (int u = 0; u < P; u++) {
                            if (u < ((kHalf && ti == H) ? PH : P)) {
                                const double bq = fma(a.z, b[u].z, fma(a.y, b[u].y, a.x * b[u].x));
                                const double det = fma(a.a, b[u].a, -(bq * bq));
                                const double dn = fma(cz, b[u].z, fma(cy, b[u].y, cx * b[u].x));
                                const double dn2 = dn * dn;
                                const bool kp_ = okm && !below_kthr(ss[u], prm) && !(dn2 > det * prm.dthr2);   // :73-74
                                // (float32 confidences add in float32, first camera's + second camera's: the sum commutes)
                                const int ai = (ti < H ? ti * P : H * P) + u;
                                tot[ai] = fma(gated_sum_sel(sm, ss[u], kp_), det * __builtin_amdgcn_rsq(dn2 * det), tot[ai]);
                            }
                        }
                    }
                }
                advance(c, j0);
                __syncthreads();   // chunk c is solved (its buffer is free), chunk c + 1 is in LDS
            }
            // the two joint sub-ranges of a wave meet in its lower half, the four waves in the buffer of the last chunk
            double *lsum = reinterpret_cast<double *>(rec0 + ((par ^ (nch - 1)) & 1) * half);
#pragma unroll
            for (int u = 0; u < NACC; u++) tot[u] += __shfl_xor(tot[u], 32, 64);
            if ((tid & 32) == 0) {
#pragma unroll
                for (int ti = 0; ti < NT; ti++) {
                    const int nu = (kHalf && ti == H) ? PH : P;
#pragma unroll
                    for (int u = 0; u < P; u++)
                        if (u < nu) {
                            const int k = t_k0[ti] + t_ks[ti] * ((kHalf && ti == H) ? 2 * u : u);
                            SNOWTRI_DEV_CHECK(k >= 0 && k < Kc, 42);
                            lsum[wv * Kc + k] = tot[(ti < H ? ti * P : H * P) + u];
                        }
                }
            }
            __syncthreads();
            for (int k = tid; k < Kc; k += B) finish(k, (lsum[k] + lsum[Kc + k]) + (lsum[2 * Kc + k] + lsum[3 * Kc + k]));
        } else {
            // ---- a camera lists fewer than P persons: one candidate slot per thread and round, all joints of a chunk, the
            //      first-ray form of the solve (p1_tile_sums with a 1 x 1 tile); empty slots stay at 0
            double acc[2] = {0.0, 0.0};
            int oa2[2], ob2[2];
            Vec3 d2[2];
            bool live[2];
#pragma unroll
            for (int rd = 0; rd < 2; rd++) {
                const int k = tid + rd * B, kc = k < Kc ? k : 0;
                const int q = kc / pp, rr = kc - q * pp, pm = rr / P, ps = rr - pm * P;
                const int mc = pairs[2 * q], sc = pairs[2 * q + 1];
                live[rd] = k < Kc && pm < np_l[mc] && ps < np_l[sc];
                oa2[rd] = kP1Rec * (mc * P + pm);
                ob2[rd] = kP1Rec * (sc * P + ps);
                d2[rd] = Vec3{rig.pairc[6 * q], rig.pairc[6 * q + 1], rig.pairc[6 * q + 2]};
            }
            for (int c = 0, j0 = 0; c < nch; c++, j0 += Jc) {
                const int nj = (J - j0) < Jc ? (J - j0) : Jc;
                const char *cur = rec0 + ((par ^ c) & 1) * half;
#pragma unroll
                for (int rd = 0; rd < 2; rd++) {
                    const char *pj = cur;
                    for (int t = 0; t < (live[rd] ? nj : 0); t++, pj += jstr) {
                        const RayRec a = p1_load_ray(pj + oa2[rd]), b = p1_load_ray(pj + ob2[rd]);
                        const TIn sm = p1_load_score<TIn>(pj + oa2[rd]), ss = p1_load_score<TIn>(pj + ob2[rd]);
                        const Vec3 d = d2[rd];
                        const double cx = fma(d.y, a.z, -(d.z * a.y)), cy = fma(d.z, a.x, -(d.x * a.z)), cz = fma(d.x, a.y, -(d.y * a.x));
                        const double bq = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
                        const double det = fma(a.a, b.a, -(bq * bq));
                        const double dn = fma(cz, b.z, fma(cy, b.y, cx * b.x));
                        const double dn2 = dn * dn;
                        const bool kp_ = !below_kthr(sm, prm) && !below_kthr(ss, prm) && !(dn2 > det * prm.dthr2);   // :73-74
                        acc[rd] = fma(gated_sum_sel(sm, ss, kp_), det * __builtin_amdgcn_rsq(dn2 * det), acc[rd]);
                    }
                }
                advance(c, j0);
                __syncthreads();
            }
#pragma unroll
            for (int rd = 0; rd < 2; rd++)
                if (tid + rd * B < Kc) finish(tid + rd * B, live[rd] ? acc[rd] : 0.0);
        }
        if (redo) hd[0] = 1;
        __syncthreads();
        if (tid == 0) {
            if (out_flags) out_flags[f] = 0u;
            if (hd[0] && exact_list) exact_list[atomicAdd(exact_count, 1ull)] = (uint32_t)f;
            hd[0] = hd[1] = 0;   // (for the frame after the next one)
        }
        par ^= nch & 1;   // the next frame's first chunk goes where this frame's last chunk was not (its partial sums may still be read)
        f = fnext;
    }
}

}  // namespace snowtri

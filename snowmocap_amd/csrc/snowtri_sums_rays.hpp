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
// MEASURED (round 5, EXPERIMENTS.md) and NOT the default.  First version: 8 x 4 x 10 000 frames 831 us against the tile kernel's
// 710 us, VALU busy 0.54 against 0.77 -- blamed on its 75 LDS reads per joint.  The real causes were the tile kernel's too (found
// there with wall-clock stamps): every keypoint request waited for its own data, the record fill read its matrices in dependent
// LDS round trips, the pair gates went through scalar masks.  With the same fixes (requests inside one pass of the chunk loop,
// gated_weight, bad-record masks): 635-655 us against the reworked tile kernel's 674-680 us on one stream -- 10 % fewer VALU
// instructions per wave (71 566 against 79 269), conflict ratio 0.05, VALU busy 0.72 against 0.75 (nine chunks of 16 joints per
// frame against seven of 20: more barriers per solve) -- and END TO END NOTHING: 1.071-1.085 against 1.086-1.091 ms per call on
// one stream, 1.040-1.050 against 1.036-1.043 ms in the default two-segment mode.  SNOWTRI_SUMS_RAYS=1 selects it
// (tests/test_gpu_handover.py, tests/test_gpu_debug_bounds.py).
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
    return ((size_t)kSumsHeadBytes + (size_t)8 * (C + (C & 1)) + (size_t)8 * npairs + (size_t)72 * C + 15) & ~(size_t)15;   // head | n_persons [2][C] | pairs | ray matrices
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
    uint32_t *badrows = reinterpret_cast<uint32_t *>(smem + 64);                  // [2][8]: records that are not finite, as in k_candidate_sums
    constexpr int Cp = C + (C & 1);
    int32_t *np_both = reinterpret_cast<int32_t *>(smem + kSumsHeadBytes);        // [2][Cp]: persons listed by the cameras, by frame parity
    int32_t *pairs = np_both + 2 * Cp;                                            // [NP][2]
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
    // (requested in front of a chunk's solves, written behind them, inside one pass of the chunk loop; one-component loads at
    // offsets the compiler cannot relate; the ray matrices of both records before the rays: k_candidate_sums has the reasons)
    Kp3<TIn> pre[NPF];
    int npv = P;       // threads < C: n_persons of the frame whose first chunk is in `pre`
    auto fetch = [&](const Kp3<TIn> *kpf, int j0, int nj) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<Kp3<TIn> *>(kpf), 0, R * Jrow * (int)sizeof(Kp3<TIn>), 0x00020000);
        SNOWTRI_DEV_CHECK(fj >= nj || (j0 + fj >= 0 && j0 + fj < J), 41);
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            int o0 = fj < nj ? ((frow0 + (B / Jc) * n) * Jrow + j0 + fj) * (int)sizeof(Kp3<TIn>) : 0x7ffffff0, o1 = o0, o2 = o0;   // (past the range: zeros)
            asm volatile("" : "+v"(o1));
            asm volatile("" : "+v"(o2));
            if constexpr (sizeof(TIn) == 4) {
                pre[n].u = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o0, 0, 0));
                pre[n].v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o1 + 4, 0, 0));
                pre[n].s = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o2 + 8, 0, 0));
            } else {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const u2v a = __builtin_amdgcn_raw_buffer_load_b64(rs, o0, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs, o1 + 8, 0, 0),
                          d = __builtin_amdgcn_raw_buffer_load_b64(rs, o2 + 16, 0, 0);
                pre[n].u = __hiloint2double((int)a.y, (int)a.x);
                pre[n].v = __hiloint2double((int)b.y, (int)b.x);
                pre[n].s = __hiloint2double((int)d.y, (int)d.x);
            }
        }
    };
    auto commit = [&](char *buf, uint32_t *bad, int nj) {
        double Mr[NPF][9];
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const double *M = Ml + 9 * ((frow0 + (B / Jc) * n) / P);
#pragma unroll
            for (int k = 0; k < 9; k++) Mr[n][k] = M[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NPF; n++) {
            const int row = frow0 + (B / Jc) * n, c = row / P;
            const RayRec h = make_ray(Mr[n], pre[n].u, pre[n].v);
            const TIn sc = pre[n].s;
            const bool notfinite = !(fma((double)sc, 0.0, h.a) < 1e300);   // NaN or infinite |h|^2, NaN or infinite score
            if (fj < nj) {
                if (notfinite) atomicOr(&bad[c >> 1], 1u << (row - c * P + 16 * (c & 1)));
                p1_store_record<TIn>(buf + fj * jstr + kP1Rec * row, h, below_kthr(sc, prm) ? GatedScore<TIn>::value : sc);
            }
        }
    };

    const int nch = (J + Jc - 1) / Jc;
    int64_t f = blockIdx.x;
    int par = 0;
    __syncthreads();   // constants and the cleared head
    if (f < F) {       // the first chunk of the workgroup's first frame (every later one is written during the frame before it)
        fetch(kp3 + f * (int64_t)R * Jrow, 0, J < Jc ? J : Jc);
        if (n_persons && tid < C) npv = n_persons[f * C + tid];
        commit(rec0, badrows, J < Jc ? J : Jc);
        if (tid < C) np_both[tid] = npv;
    }
    for (int it = 0; f < F; it++) {
        int32_t *hd = head + 4 * (it & 1);
        const int32_t *np_l = np_both + Cp * (it & 1);
        uint32_t *bad = badrows + 8 * (it & 1);
        const Kp3<TIn> *kpf = kp3 + f * (int64_t)R * Jrow;
        double *cs_f = csum + f * (int64_t)Kc;
        unsigned long long ticket = 0;   // (handed to the workgroup behind the first chunk's solves: the barrier below does not wait for the atomic)
        if (tid == 0) {
            ticket = atomicAdd(next_frame, 1ull);
            if (nch == 1) hd[2] = (int32_t)ticket;
        }
        if (tid < C && np_l[tid] != P) hd[1] = 1;   // (np_l[tid] was written by this thread)
        __syncthreads();   // first chunk (written during the frame before), np_l, ragged flag and ticket are there
        if (tid == 0) head[4 * ((it & 1) ^ 1)] = head[4 * ((it & 1) ^ 1) + 1] = 0;   // the flags of the frame before
        int64_t fnext = nch == 1 ? (int64_t)gridDim.x + (int64_t)(uint32_t)hd[2] : 0;
        const bool ragged = hd[1] != 0;   // workgroup-uniform
        bool redo = false;
        auto finish = [&](int k, double v) {   // the 1 / (2 * 1000) of :72, and whether a mean can decide :80-81
            const double s_ = v * 0.0005, mean = s_ / (double)J;
            cs_f[k] = s_;
            redo |= exact_list != nullptr && (!(fabs(mean) < 1e300) || (s_ != 0.0 && fabs(mean - prm.avg_thr) <= 1e-6 * fabs(mean)));
        };
        // in front of the solves of chunk c: the keypoints of chunk c + 1 -- in the frame's last chunk those of the NEXT frame's
        // first chunk, with its n_persons; behind them: its records, into the buffer chunk c - 1 was solved in
        int nj_next = 0;
        bool last = false, succ = false;
        auto request = [&](int c, int j0) {
            last = c + 1 >= nch;
            succ = fnext < F;
            nj_next = last ? (succ ? (J < Jc ? J : Jc) : 0) : ((J - j0 - Jc) < Jc ? (J - j0 - Jc) : Jc);
            fetch(last && succ ? kp3 + fnext * (int64_t)R * Jrow : kpf, last ? 0 : j0 + Jc, nj_next);
            if (last && succ && n_persons && tid < C) npv = n_persons[fnext * C + tid];
            __builtin_amdgcn_sched_barrier(0);
        };
        auto written = [&](int c) {
            commit(rec0 + ((par ^ (c + 1)) & 1) * half, last ? badrows + 8 * ((it & 1) ^ 1) : bad, nj_next);
            if (last && succ && tid < C) np_both[Cp * ((it & 1) ^ 1) + tid] = npv;
            if (nch > 1 && c == 0 && tid == 0) hd[2] = (int32_t)ticket;
        };
        auto handed = [&](int c) {   // behind the barrier of chunk c
            if (nch > 1 && c == 0) fnext = (int64_t)gridDim.x + (int64_t)(uint32_t)hd[2];
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
                request(c, j0);
                for (int t = 0; t < cnt; t++, pj += jstr) {
                    const RayRec a = p1_load_ray(pj + oa);
                    const TIn sm = p1_load_score<TIn>(pj + oa);
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
                                // :73-74 without compares (gated_weight; float32 confidences add in float32, the sum commutes)
                                const int ai = (ti < H ? ti * P : H * P) + u;
                                tot[ai] = fma(gated_weight(sm, ss[u], fma(det, prm.dthr2, -dn2)), det * __builtin_amdgcn_rsq(dn2 * det), tot[ai]);
                            }
                        }
                    }
                }
                written(c);
                __syncthreads();   // chunk c is solved (its buffer is free), chunk c + 1 is in LDS
                handed(c);
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
                request(c, j0);
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
                written(c);
                __syncthreads();
                handed(c);
            }
#pragma unroll
            for (int rd = 0; rd < 2; rd++)
                if (tid + rd * B < Kc) finish(tid + rd * B, live[rd] ? acc[rd] : 0.0);
        }
        if (redo) hd[0] = 1;
        if (tid < C) {   // a listed row with a record that is not finite: the exact pass, or NaN sums where the launch has no exact list
            const uint32_t listed = np_l[tid] >= 16 ? 0xffffu : ((1u << np_l[tid]) - 1u);
            if ((bad[tid >> 1] >> (16 * (tid & 1))) & listed) hd[0] = exact_list ? 1 : 2;
        }
        __syncthreads();
        if (tid < 8) bad[tid] = 0u;
        if (hd[0] == 2)
            for (int k = tid; k < Kc; k += B) cs_f[k] = __longlong_as_double(0x7ff8000000000000ll);
        if (tid == 0) {
            if (out_flags) out_flags[f] = 0u;
            if (hd[0] && exact_list) exact_list[atomicAdd(exact_count, 1ull)] = (uint32_t)f;
        }   // (hd[0] and hd[1] are cleared behind the next frame's first barrier)
        par ^= nch & 1;   // the next frame's first chunk goes where this frame's last chunk was not (its partial sums may still be read)
        f = fnext;
    }
}

}  // namespace snowtri

// snowtri_kernels.hpp -- materialised-candidate kernels (API-faithful path and general fallback).
//
//   k_rays          A1  camera.py:234-253
//   k_skew          A2  triangulation.py:24-31
//   k_triangulate   A1+A3 per (frame, candidate slot, joint)   triangulation.py:50-78
//   k_cand_mean     A3  np.mean(p_score) + average_score_threshold filter   triangulation.py:79-81
//   k_condense      A4  greedy clustering + score-weighted fusion   triangulation.py:95-162
//
// The fused single-cluster fast path lives in snowtri_fused.hpp.
#pragma once
#include "snowtri_math.hpp"

namespace snowtri {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_rays(int64_t n, const double *__restrict__ M,
                                                 const double *__restrict__ uv, double *__restrict__ rays) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        Vec3 h = ray_from_pixel(M, uv[2 * i], uv[2 * i + 1]);
        rays[3 * i] = h.x;
        rays[3 * i + 1] = h.y;
        rays[3 * i + 2] = h.z;
    }
}

__global__ __launch_bounds__(kBlock) void k_skew(int64_t n, const double *__restrict__ hm,
                                                 const double *__restrict__ hs, const double *__restrict__ tm,
                                                 const double *__restrict__ ts, double *__restrict__ dist,
                                                 double *__restrict__ W, unsigned long long *n_singular) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        Vec3 a = {hm[3 * i], hm[3 * i + 1], hm[3 * i + 2]}, b = {hs[3 * i], hs[3 * i + 1], hs[3 * i + 2]};
        Vec3 c = {tm[3 * i], tm[3 * i + 1], tm[3 * i + 2]}, d = {ts[3 * i], ts[3 * i + 1], ts[3 * i + 2]};
        SkewOut o = skew_ray_solve(a, b, c, d);
        dist[i] = o.dist;
        W[3 * i] = o.W.x;
        W[3 * i + 1] = o.W.y;
        W[3 * i + 2] = o.W.z;
        if (o.singular) atomicAdd(n_singular, 1ull);
    }
}

template <typename T>
struct Kp3 {  // one detected keypoint as stored in kpts: (u, v, score)
    T u, v, s;
};

// Accuracy probe of the fast reciprocal / rsqrt helpers (tests/test_gpu_parity.py).
__global__ __launch_bounds__(kBlock) void k_fastmath_probe(int64_t n, const double *__restrict__ x,
                                                           double *__restrict__ r2, double *__restrict__ r1,
                                                           double *__restrict__ q1) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        r2[i] = rcp_nr2(x[i]);
        r1[i] = rcp_nr1(x[i]);
        q1[i] = rsq_nr1(x[i]);
    }
}

// Raw hardware approximations (no Newton step): what k_fused_lean / the float32 branch of k_frame_recompute use for
// 1/dist.  Probed so that the accuracy the float32 score contract relies on is a measured number.
__global__ __launch_bounds__(kBlock) void k_fastmath_probe_raw(int64_t n, const double *__restrict__ x,
                                                               double *__restrict__ rcp_raw, double *__restrict__ rsq_raw) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        rcp_raw[i] = __builtin_amdgcn_rcp(x[i]);
        rsq_raw[i] = __builtin_amdgcn_rsq(x[i]);
    }
}

// HBM-counter calibration (measurement aid): streams with the access shapes of the fused kernels and a KNOWN byte
// count -- 12-byte records read per lane (global_load_dwordx3, consecutive lanes = consecutive records) and 16-byte
// records written per lane (global_store_dwordx4) -- so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled on a
// kernel other than the one being measured (scripts/profile.sh).
struct Rec12 {
    float a, b, c;
};
__global__ __launch_bounds__(kBlock) void k_calib_read12(int64_t nrec, const Rec12 *__restrict__ src, float *__restrict__ sink) {
    float acc = 0.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nrec; i += (int64_t)gridDim.x * blockDim.x) {
        const Rec12 r = src[i];
        acc += r.a + r.b + r.c;
    }
    if (acc == 123.456f) sink[0] = acc;   // keeps the loads alive; practically never true
}
__global__ __launch_bounds__(kBlock) void k_calib_write16(int64_t nrec, float4 *__restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nrec; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}

// Rig constants resident in HBM (<= 1.5 KB, L2/scalar-cache resident): M[C][9], t[C][3],
// pair table [npairs][2] in the reference's loop order mc < sc (triangulation.py:56-58).
struct Rig {
    const double *M;
    const double *t;
    const int32_t *pairs;
    const double *pairc;  // [npairs][6]: d = t_sc - t_mc, tsum = t_mc + t_sc (host-precomputed)
    const double *P;      // [C][12]: world->pixel matrices K [R^T | -R^T t] (DLT method)
    int32_t C, npairs;
};

template <typename TIn>
__global__ __launch_bounds__(kBlock) void k_triangulate(int64_t F, int Pmax, int J, int Kc, Rig rig,
                                                        const TIn *__restrict__ kpts,
                                                        const int32_t *__restrict__ n_persons, Params prm,
                                                        double *__restrict__ cand_xyz,
                                                        double *__restrict__ cand_kscore,
                                                        unsigned long long *n_singular, double *__restrict__ mirror_xyz,
                                                        double *__restrict__ mirror_kscore) {
    // mirror_*: the small per-frame host calls -- a second copy of every output straight into page-locked HOST memory mapped
    // into the device (no D2H copy behind the kernels: an SDMA copy of 30 KB costs ~10 us of latency, these stores ~2);
    // slots that are not real pairs are then written as zeros here (no memset in front of the kernel)
    const int64_t total = F * (int64_t)Kc * J;
    const int pp = Pmax * Pmax;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % J);
        const int64_t fk = i / J;
        const int k = (int)(fk % Kc);
        const int64_t f = fk / Kc;
        const int q = k / pp, r = k - q * pp, pm = r / Pmax, ps = r - pm * Pmax;
        const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
        const int nm = n_persons ? n_persons[f * rig.C + mc] : Pmax;
        const int ns = n_persons ? n_persons[f * rig.C + sc] : Pmax;
        if (pm >= nm || ps >= ns) {  // not a real pair: slot stays untouched (zeroed when mirrored), keep = 0
            if (mirror_xyz) {
                cand_xyz[3 * i] = cand_xyz[3 * i + 1] = cand_xyz[3 * i + 2] = cand_kscore[i] = 0.0;
                mirror_xyz[3 * i] = mirror_xyz[3 * i + 1] = mirror_xyz[3 * i + 2] = mirror_kscore[i] = 0.0;
            }
            continue;
        }
        const TIn *km = kpts + ((((f * rig.C + mc) * Pmax + pm) * (int64_t)J) + j) * 3;
        const TIn *ks = kpts + ((((f * rig.C + sc) * Pmax + ps) * (int64_t)J) + j) * 3;
        const TIn um = km[0], vm = km[1], sm = km[2];
        const TIn us = ks[0], vs = ks[1], ss = ks[2];
        const Vec3 hm = ray_from_pixel(rig.M + 9 * mc, (double)um, (double)vm);
        const Vec3 hs = ray_from_pixel(rig.M + 9 * sc, (double)us, (double)vs);
        const Vec3 tm = {rig.t[3 * mc], rig.t[3 * mc + 1], rig.t[3 * mc + 2]};
        const Vec3 ts = {rig.t[3 * sc], rig.t[3 * sc + 1], rig.t[3 * sc + 2]};
        const SkewOut o = skew_ray_solve(hm, hs, tm, ts);
        if (o.singular) atomicAdd(n_singular, 1ull);
        const double s = pair_score(sm, ss, o.dist, prm);
        cand_xyz[3 * i] = o.W.x;
        cand_xyz[3 * i + 1] = o.W.y;
        cand_xyz[3 * i + 2] = o.W.z;
        cand_kscore[i] = s;
        if (mirror_xyz) {
            mirror_xyz[3 * i] = o.W.x;
            mirror_xyz[3 * i + 1] = o.W.y;
            mirror_xyz[3 * i + 2] = o.W.z;
            mirror_kscore[i] = s;
        }
    }
}

// One wave per (frame, slot): mean of the J joint scores and the keep decision.
__global__ __launch_bounds__(kBlock) void k_cand_mean(int64_t F, int Pmax, int J, int Kc, Rig rig,
                                                      const int32_t *__restrict__ n_persons, Params prm,
                                                      const double *__restrict__ cand_kscore,
                                                      double *__restrict__ cand_pscore,
                                                      uint8_t *__restrict__ cand_keep, double *__restrict__ mirror_pscore,
                                                      uint8_t *__restrict__ mirror_keep, unsigned long long *n_singular,
                                                      unsigned long long *mirror_singular) {
    // mirror_*: see k_triangulate; the singular-pair counter of that kernel is handed over here and left at 0 for the next call
    if (mirror_singular && blockIdx.x == 0 && threadIdx.x == 0) {
        *mirror_singular = *n_singular;
        *n_singular = 0ull;
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int pp = Pmax * Pmax;
    for (int64_t fk = wave; fk < F * (int64_t)Kc; fk += nwaves) {
        const int k = (int)(fk % Kc);
        const int64_t f = fk / Kc;
        const int q = k / pp, r = k - q * pp, pm = r / Pmax, ps = r - pm * Pmax;
        const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
        const int nm = n_persons ? n_persons[f * rig.C + mc] : Pmax;
        const int ns = n_persons ? n_persons[f * rig.C + sc] : Pmax;
        const bool valid = pm < nm && ps < ns;
        double s = 0.0;
        if (valid)
            for (int j = lane; j < J; j += 64) s += cand_kscore[fk * J + j];
        s = wave_sum(s);
        const double mean = s / (double)J;  // triangulation.py:79
        if (lane == 0) {
            cand_pscore[fk] = valid ? mean : 0.0;
            cand_keep[fk] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;  // :80-81 (NaN mean is kept)
            if (mirror_pscore) {
                mirror_pscore[fk] = valid ? mean : 0.0;
                mirror_keep[fk] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;
            }
        }
    }
}

// Completion word of the small host calls.  Their outputs land in mapped page-locked memory, so the host does not need the
// runtime to tell it that the kernel is over (hipStreamSynchronize sleeps on an interrupt: ~10 us to wake up): every wave
// makes its stores visible at system scope, the last workgroup to arrive writes `seq` into a mapped word behind them, and the
// host spins on that word.  count: device counter, 0 between launches.  flag == nullptr: nothing to do.
struct HostDone {
    unsigned int *count;
    unsigned long long *flag;   // mapped host memory
    unsigned long long seq;
};
__device__ __forceinline__ void host_done_signal(const HostDone &d) {
    if (!d.flag) return;
    __threadfence_system();     // this wave's stores (device scratch and host mirror) are out
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(d.count, 1u) == gridDim.x - 1u) {
            *d.count = 0u;
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(d.flag) = d.seq;
        }
    }
}

// Do two HIP streams run side by side?  The runtime multiplexes a process's streams over a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default; a new stream joins the least-used one) and packets of one queue run in order, whichever
// stream they came from: an internal stream that lands on the caller's queue turns the two-segment split of a multi-person
// call into a serial run (measured: 8 x 4 with float64 outputs 1.19 -> 1.36 ms per 10 000 frames in a process that had created
// 40 streams before).  The probe: k_probe_wait on stream A spins (bounded) on a mapped word that k_probe_set on stream B
// writes.  B behind A on one queue never gets to run in time.  w[0]: flag, w[1]: verdict.
__global__ void k_probe_wait(volatile unsigned int *w, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();   // 100 MHz
    unsigned int seen = 0;
    do {
        seen = w[0];
        if (!seen) __builtin_amdgcn_s_sleep(64);
    } while (!seen && wall_clock64() - t0 < ticks);
    w[1] = seen ? 1u : 0u;
    __threadfence_system();
}
__global__ void k_probe_set(volatile unsigned int *w) {
    w[0] = 1u;
    __threadfence_system();
}

// k_triangulate + k_cand_mean in ONE launch for the per-frame calls (a handful of candidate slots: the second launch was ~6 of
// the call's 30 us): one workgroup per (frame, slot); the joints' scores meet in LDS and wave 0 takes their mean in the order
// k_cand_mean takes it (lane-strided partial sums, then the wave tree), so a slot's bits do not depend on which route ran.
// Dynamic LDS: 8 J bytes.  mirror_* as in k_triangulate (always given: this kernel serves the host calls only).
template <typename TIn>
__global__ __launch_bounds__(kBlock) void k_triangulate_slots(int64_t F, int Pmax, int J, int Kc, Rig rig, const TIn *__restrict__ kpts,
                                                              const int32_t *__restrict__ n_persons, Params prm,
                                                              double *__restrict__ cand_xyz, double *__restrict__ cand_kscore,
                                                              double *__restrict__ cand_pscore, uint8_t *__restrict__ cand_keep,
                                                              unsigned long long *n_singular, double *__restrict__ mirror_xyz,
                                                              double *__restrict__ mirror_kscore, double *__restrict__ mirror_pscore,
                                                              uint8_t *__restrict__ mirror_keep, unsigned long long *mirror_singular,
                                                              unsigned int *slots_done, HostDone done) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sc_l = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int pp = Pmax * Pmax;
    const int64_t fk = blockIdx.x;
    const int k = (int)(fk % Kc);
    const int64_t f = fk / Kc;
    const int q = k / pp, r = k - q * pp, pm = r / Pmax, ps = r - pm * Pmax;
    const int mc = rig.pairs[2 * q], sc = rig.pairs[2 * q + 1];
    const int nm = n_persons ? n_persons[f * rig.C + mc] : Pmax;
    const int ns = n_persons ? n_persons[f * rig.C + sc] : Pmax;
    const bool valid = pm < nm && ps < ns;
    const Vec3 tm = {rig.t[3 * mc], rig.t[3 * mc + 1], rig.t[3 * mc + 2]};
    const Vec3 ts = {rig.t[3 * sc], rig.t[3 * sc + 1], rig.t[3 * sc + 2]};
    for (int j = tid; j < J; j += kBlock) {
        const int64_t i = fk * J + j;
        double x = 0.0, y = 0.0, z = 0.0, s = 0.0;
        if (valid) {
            const TIn *km = kpts + ((((f * rig.C + mc) * Pmax + pm) * (int64_t)J) + j) * 3;
            const TIn *ks = kpts + ((((f * rig.C + sc) * Pmax + ps) * (int64_t)J) + j) * 3;
            const TIn um = km[0], vm = km[1], sm = km[2];
            const TIn us = ks[0], vs = ks[1], ss = ks[2];
            const Vec3 hm = ray_from_pixel(rig.M + 9 * mc, (double)um, (double)vm);
            const Vec3 hs = ray_from_pixel(rig.M + 9 * sc, (double)us, (double)vs);
            const SkewOut o = skew_ray_solve(hm, hs, tm, ts);
            if (o.singular) atomicAdd(n_singular, 1ull);
            s = pair_score(sm, ss, o.dist, prm);
            x = o.W.x;
            y = o.W.y;
            z = o.W.z;
        }
        cand_xyz[3 * i] = mirror_xyz[3 * i] = x;
        cand_xyz[3 * i + 1] = mirror_xyz[3 * i + 1] = y;
        cand_xyz[3 * i + 2] = mirror_xyz[3 * i + 2] = z;
        cand_kscore[i] = mirror_kscore[i] = s;
        sc_l[j] = s;
    }
    __threadfence();   // every thread's n_singular atomics are visible device-wide before this workgroup counts itself done below
    __syncthreads();
    if (tid < 64) {
        double s = 0.0;
        if (valid)
            for (int j = lane; j < J; j += 64) s += sc_l[j];
        s = wave_sum(s);
        const double mean = s / (double)J;  // triangulation.py:79
        if (lane == 0) {
            cand_pscore[fk] = mirror_pscore[fk] = valid ? mean : 0.0;
            cand_keep[fk] = mirror_keep[fk] = (valid && !(mean < prm.avg_thr)) ? 1 : 0;  // :80-81 (NaN mean is kept)
            // the last workgroup to finish hands the singular-pair count over and leaves both counters at 0 for the next call
            __threadfence();
            if (atomicAdd(slots_done, 1u) == gridDim.x - 1u) {
                *mirror_singular = atomicAdd(n_singular, 0ull);
                *n_singular = 0ull;
                *slots_done = 0u;
            }
        }
    }
    host_done_signal(done);
}

// ---------------------------------------------------------------------------------------------
// Output writers for k_condense: the API-faithful split fp64 arrays, or the packed
// (x, y, z, score) layout of the batched hot path.
struct SplitWriter {
    double *xyz, *kscore, *pscore;
    __device__ __forceinline__ void joint(int64_t f, int Pout, int kn, int slot, int b, double x, double y,
                                          double z, double s) const {
        const int64_t o = ((f * Pout + slot) * (int64_t)kn + b);
        xyz[3 * o] = x;
        xyz[3 * o + 1] = y;
        xyz[3 * o + 2] = z;
        kscore[o] = s;
    }
    __device__ __forceinline__ void person(int64_t f, int Pout, int slot, double s) const {
        pscore[f * Pout + slot] = s;
    }
};

template <typename TOut>
struct PackedWriter {
    TOut *xyzs, *pscore;
    __device__ __forceinline__ void joint(int64_t f, int Pout, int kn, int slot, int b, double x, double y,
                                          double z, double s) const {
        TOut *o = xyzs + ((f * Pout + slot) * (int64_t)kn + b) * 4;
        o[0] = (TOut)x;
        o[1] = (TOut)y;
        o[2] = (TOut)z;
        o[3] = (TOut)s;
    }
    // an unused slot's joint: one 16-byte store (xyzs is 16-byte aligned: checked at the ABI)
    __device__ __forceinline__ void zero_joint(int64_t f, int Pout, int kn, int slot, int b) const {
        TOut *o = xyzs + ((f * Pout + slot) * (int64_t)kn + b) * 4;
        if constexpr (sizeof(TOut) == 4) {
            *reinterpret_cast<float4 *>(o) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            *reinterpret_cast<double2 *>(o) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2 *>(o + 2) = make_double2(0.0, 0.0);
        }
    }
    __device__ __forceinline__ void person(int64_t f, int Pout, int slot, double s) const {
        if (pscore) pscore[f * Pout + slot] = (TOut)s;
    }
};

__device__ __forceinline__ double block_sum(double v, double *red /*[kBlock/64]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) t += red[w];
    return t;
}

constexpr int kCondenseMaxJointsPerThread = 4;  // keypoint_num <= 4 * kBlock

// LDS footprint of condense_frame for N candidate slots.
__host__ __device__ constexpr size_t condense_lds_bytes(int N) { return (size_t)16 * N + 64; }

// A4 for ONE frame by one 256-thread workgroup.
//   cx[N][J][3], cs[N][J]   candidate slots of this frame (global memory)
//   keep                    global uint8[N] keep flags, or nullptr:
//                             keep_in_lds == false -> every slot is a candidate
//                             keep_in_lds == true  -> flags were left in the LDS `cluster_of` array
//   f                       absolute frame index used for the outputs
// Ends with a __syncthreads(), so the LDS can be reused right after it returns.
template <typename Writer>
__device__ __forceinline__ void condense_frame(int64_t f, int N, int J, const double *__restrict__ cx,
                                               const double *__restrict__ cs,
                                               const uint8_t *__restrict__ keep, bool keep_in_lds,
                                               const Params &prm, int Pout, const Writer &wr,
                                               int32_t *__restrict__ out_count,
                                               uint32_t *__restrict__ out_flags, char *lds) {
    int32_t *idx = reinterpret_cast<int32_t *>(lds);  // kept slots, reference list order
    int32_t *cluster_of = idx + N;                    // cluster id per kept candidate, -1 = free
    int32_t *csize = cluster_of + N;                  // members per cluster
    int32_t *cseed = csize + N;                       // seed (kept index) per cluster
    double *red = reinterpret_cast<double *>(cseed + N);
    int32_t *misc = reinterpret_cast<int32_t *>(red + kBlock / 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int ci = prm.center, kn = prm.kn;

    if (tid < 64) {
        // ordered compaction of the kept slots (= the reference's candidate list)
        int n = 0;
        for (int base = 0; base < N; base += 64) {
            const int k = base + lane;
            bool kp = k < N;
            if (kp && keep) kp = keep[k] != 0;
            if (kp && !keep && keep_in_lds) kp = cluster_of[k] != 0;
            const unsigned long long m = __ballot(kp);
            if (kp) idx[n + __popcll(m & ((1ull << lane) - 1ull))] = k;
            n += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int i = lane; i < n; i += 64) cluster_of[i] = -1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // greedy clustering, triangulation.py:107-130: seeds in list order, the last candidate
        // never seeds, distance to the SEED's centre joint, `dist > tol` skips (NaN absorbs).
        int ncl = 0;
        for (int mc = 0; mc < n - 1; mc++) {
            if (cluster_of[mc] != -1) continue;
            const double *pm = cx + ((int64_t)idx[mc] * J + ci) * 3;
            const double mx = pm[0], my = pm[1], mz = pm[2];
            int cnt = 0;
            for (int base = mc + 1; base < n; base += 64) {
                const int sc = base + lane;
                bool ab = false;
                if (sc < n && cluster_of[sc] == -1) {
                    const double *ps = cx + ((int64_t)idx[sc] * J + ci) * 3;
                    const double dx = mx - ps[0], dy = my - ps[1], dz = mz - ps[2];
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
        if (lane == 0) {
            misc[0] = n;
            misc[1] = ncl;
        }
    }
    __syncthreads();
    const int n = misc[0], ncl = misc[1];
    int nout = 0;
    for (int cid = 0; cid < ncl; cid++) {
        const int size = csize[cid];
        if ((double)size < prm.num_tol) continue;  // :132-134 (members stay absorbed)
        const int seed = cseed[cid];
        double jx[kCondenseMaxJointsPerThread], jy[kCondenseMaxJointsPerThread],
            jz[kCondenseMaxJointsPerThread], js[kCondenseMaxJointsPerThread];
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < kCondenseMaxJointsPerThread; r++) {
            const int b = tid + r * kBlock;
            jx[r] = jy[r] = jz[r] = js[r] = 0.0;
            if (b < kn) {
                double sum = 0.0;  // :141
                for (int i = seed; i < n; i++)
                    if (cluster_of[i] == cid) sum += cs[(int64_t)idx[i] * J + b];
                if (!(sum == 0.0)) {  // :142-143
                    double x = 0.0, y = 0.0, z = 0.0;
                    for (int i = seed; i < n; i++)
                        if (cluster_of[i] == cid) {
                            const int64_t o = (int64_t)idx[i] * J + b;
                            const double w = cs[o] / sum;  // :144
                            x = fma(cx[3 * o], w, x);      // :145-147
                            y = fma(cx[3 * o + 1], w, y);
                            z = fma(cx[3 * o + 2], w, z);
                        }
                    jx[r] = x;
                    jy[r] = y;
                    jz[r] = z;
                    js[r] = sum / (double)size;  // :148
                }
                part += js[r];
            }
        }
        const double avg = block_sum(part, red) / (double)kn;  // :150
        if (avg < prm.score_tol) continue;                     // :151-152
        if (nout < Pout) {
#pragma unroll
            for (int r = 0; r < kCondenseMaxJointsPerThread; r++) {
                const int b = tid + r * kBlock;
                if (b < kn) wr.joint(f, Pout, kn, nout, b, jx[r], jy[r], jz[r], js[r]);
            }
            if (tid == 0) wr.person(f, Pout, nout, avg);
        }
        nout++;
    }
    for (int slot = nout; slot < Pout && !prm.no_zero_fill; slot++) {  // deterministic padding (unless the call opted out)
        for (int b = tid; b < kn; b += kBlock) wr.joint(f, Pout, kn, slot, b, 0.0, 0.0, 0.0, 0.0);
        if (tid == 0) wr.person(f, Pout, slot, 0.0);
    }
    if (tid == 0) {
        out_count[f] = nout;
        if (out_flags && nout > Pout) atomicOr(&out_flags[f], 2u /*SNOWTRI_FLAG_OVERFLOW*/);
    }
    __syncthreads();
}

// A4 over materialised candidates: one workgroup per frame (grid-stride).
// Dynamic LDS: condense_lds_bytes(N).
template <typename Writer>
__global__ __launch_bounds__(kBlock) void k_condense(int64_t F, int N, int J,
                                                     const double *__restrict__ cand_xyz,
                                                     const double *__restrict__ cand_kscore,
                                                     const uint8_t *__restrict__ cand_keep, Params prm,
                                                     int Pout, Writer wr, int32_t *__restrict__ out_count,
                                                     uint32_t *__restrict__ out_flags, HostDone done) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int64_t f = blockIdx.x; f < F; f += gridDim.x)
        condense_frame(f, N, J, cand_xyz + f * (int64_t)N * J * 3, cand_kscore + f * (int64_t)N * J,
                       cand_keep ? cand_keep + f * (int64_t)N : nullptr, false, prm, Pout, wr, out_count,
                       out_flags, smem);
    host_done_signal(done);
}

}  // namespace snowtri

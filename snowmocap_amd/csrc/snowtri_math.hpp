// snowtri_math.hpp -- device-side small linear algebra shared by every kernel.
//
// All arithmetic is fp64 (the reference is NumPy float64; an fp32 restatement misses the
// 1e-4 m budget and gives meaningless scores -- SURVEY.md F5).  hipcc contracts a*b+c into
// v_fma_f64, which only tightens rounding relative to the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snowtri {

// Device-side bounds checks under a debug macro (the reference has none; `make debug` builds
// snowmocap_amd/libsnowtri_dbg.so with -DSNOWTRI_DEBUG_BOUNDS).  A violated check counts itself in g_dev_fault[0] and
// the first one leaves (code << 32 | source line) in g_dev_fault[1]; snowtri_debug_faults() reads and clears them.
// The production build compiles the checks out.
// Build variants.  The production library is compiled with NONE of these (snowtri_build_info() lists the ones a binary
// carries, tests/test_abi_and_host.py asserts the shipped one reports none):
//   SNOWTRI_DEBUG_BOUNDS     device-side index checks (libsnowtri_dbg.so, tests only)
//   SNOWTRI_TEST_KNOBS       the route-forcing environment knobs of include/snowtri.h (libsnowtri_dbg.so, tests only: the
//                            production library reads no environment)
//   SNOWTRI_DEV_MIN          4-camera float32 instantiations only (fast A/B builds, the ASan build)
//   SNOWTRI_DEV_EXPERIMENTS  gate of everything that is a measurement aid: the wall-clock stamps of SNOWTRI_LEAN_TRACE /
//                            _SUMS_TRACE / _ASSOC_TRACE.  Round 3's timing-only switches that produced WRONG outputs
//                            (..._NOSOLVE, _NOFILL, _REPEAT, _NOLOOP, _NOEPI, _STOP_AFTER_*, _MEMTEST, _COMPUTETEST) are gone.
#if (defined(SNOWTRI_LEAN_TRACE) || defined(SNOWTRI_SUMS_TRACE) || defined(SNOWTRI_ASSOC_TRACE)) && !defined(SNOWTRI_DEV_EXPERIMENTS)
#error "trace stamps are development experiments: add -DSNOWTRI_DEV_EXPERIMENTS"
#endif
#ifdef SNOWTRI_DEBUG_BOUNDS
__device__ unsigned long long g_dev_fault[2];
#define SNOWTRI_DEV_CHECK(cond, code)                                                                             \
    do {                                                                                                          \
        if (!(cond)) {                                                                                            \
            if (atomicAdd(&::snowtri::g_dev_fault[0], 1ull) == 0ull)                                              \
                ::snowtri::g_dev_fault[1] = ((unsigned long long)(code) << 32) | (unsigned long long)__LINE__;    \
        }                                                                                                         \
    } while (0)
#else
#define SNOWTRI_DEV_CHECK(cond, code) ((void)0)
#endif

// Device copy of snowtri_params (include/snowtri.h), already validated on the host:
// center is wrapped into [0, J), 0 <= kn <= J.
struct Params {
    double kthr, avg_thr, dthr;       // Human_Triangulation        (triangulation.py:50)
    double ctol, num_tol, score_tol;  // Human_Triangulation_Condense (triangulation.py:95-100)
    int32_t center, kn;
    // derived on the host for the throughput kernels
    double dthr2;    // dist > dthr  <=>  dist^2 > dthr2   (dthr < 0: -1, so every finite dist^2 exceeds it)
    float kthr_f32;  // smallest float >= kthr: for a float s,  s < kthr  <=>  s < kthr_f32  (exactly)
    int32_t no_zero_fill;  // SNOWTRI_CALL_NO_ZERO_FILL: the slots behind out_count[f] are left as the caller's buffer holds them
};

// `s < kthr` of triangulation.py:73 on the stored element type, without widening float scores.
__device__ __forceinline__ bool below_kthr(float s, const Params &p) { return s < p.kthr_f32; }
__device__ __forceinline__ bool below_kthr(double s, const Params &p) { return s < p.kthr; }

struct Vec3 {
    double x, y, z;
};

__device__ __forceinline__ double dot3(const Vec3 &a, const Vec3 &b) {
    return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
}

// A1, camera.py:241-243: f = R . (inv(K) . [u, v, 1]) with M = R . inv(K) folded on the host.
__device__ __forceinline__ Vec3 ray_from_pixel(const double *__restrict__ M, double u, double v) {
    Vec3 h;
    h.x = fma(M[0], u, fma(M[1], v, M[2]));
    h.y = fma(M[3], u, fma(M[4], v, M[5]));
    h.z = fma(M[6], u, fma(M[7], v, M[8]));
    return h;
}

// (sm + ss) / 2 of triangulation.py:72.  When the caller handed float32 confidence arrays the
// reference evaluates this in float32 (NumPy scalar arithmetic) -- reproduced by the overload.
__device__ __forceinline__ double half_score(float sm, float ss) { return (double)((sm + ss) * 0.5f); }
__device__ __forceinline__ double half_score(double sm, double ss) { return (sm + ss) * 0.5; }

// sm + ss with the reference's evaluation type (float32 scalars add in float32), widened to double
__device__ __forceinline__ double sum_score(float sm, float ss) { return (double)(sm + ss); }
__device__ __forceinline__ double sum_score(double sm, double ss) { return sm + ss; }

// sm + ss (reference evaluation type) if the pair passes its gates, else 0 -- selected before widening
__device__ __forceinline__ double gated_sum(float sm, float ss, bool keep) { return (double)(keep ? sm + ss : 0.0f); }
__device__ __forceinline__ double gated_sum(double sm, double ss, bool keep) { return keep ? sm + ss : 0.0; }

// the same with the select pinned BEFORE the widening (one v_cndmask on the float instead of two on the double)
__device__ __forceinline__ double gated_sum_sel(float sm, float ss, bool keep) {
    float t = keep ? sm + ss : 0.0f;
    asm volatile("" : "+v"(t));
    return (double)t;
}
__device__ __forceinline__ double gated_sum_sel(double sm, double ss, bool keep) { return keep ? sm + ss : 0.0; }

// The pair gates of triangulation.py:73-74 WITHOUT compares: the weight (sm + ss) of a pair whose confidences were gated when
// they were read (a confidence below the threshold replaced by -infinity: the sum is negative or NaN exactly when one of them
// is gated; confidences that pass are >= keypoint_score_threshold >= 0) and whose distance gate is the SIGN of r =
// fma(det, dthr2, -n2) (r < 0 <=> dist > distance_threshold): the sign of r is OR-ed into the sign of the sum, one v_max with 0
// zeroes a gated weight.  Four full-rate instructions where two compares into scalar masks, s_and and v_cndmask cost three
// times their issue slots (scripts/ubench/sums_mix.hip).  A kept pair weighs the float32 (float64) sum NumPy takes, a gated
// one exactly +0.  A NaN confidence does not survive the v_max: callers catch it where they read it.
__device__ __forceinline__ double gated_weight(float sm, float ss, double r) {
    const uint32_t t = ((uint32_t)__double2hiint(r) & 0x80000000u) | __float_as_uint(sm + ss);
    float v;
    asm("v_max_f32 %0, %1, 0" : "=v"(v) : "v"(t));   // (fmaxf would canonicalize t first: one more instruction)
    return (double)v;
}
__device__ __forceinline__ double gated_weight(double sm, double ss, double r) {
    const double sd = sm + ss;
    const uint32_t hi = ((uint32_t)__double2hiint(r) & 0x80000000u) | (uint32_t)__double2hiint(sd);
    const double t = __hiloint2double((int)hi, __double2loint(sd));
    double v;
    asm("v_max_f64 %0, %1, 0" : "=v"(v) : "v"(t));
    return v;
}
struct SkewOut {
    Vec3 W;       // midpoint (Wm + Ws) / 2
    double dist;  // ||Wm - Ws||
    bool singular;
};

// A2, Skew_Ray_Solver (triangulation.py:24-31) in closed form:
//   a = hm.hm, b = hm.hs, c = hs.hs, d = ts - tm, e = hm.d, f = hs.d, det = a c - b^2
//   S0 = (c e - b f) / det,  S1 = (a f - b e) / det
//   Wm = tm + hm S0,  Ws = ts - hs S1
__device__ __forceinline__ SkewOut skew_ray_solve(const Vec3 &hm, const Vec3 &hs, const Vec3 &tm,
                                                  const Vec3 &ts) {
    const double a = dot3(hm, hm), b = dot3(hm, hs), c = dot3(hs, hs);
    const double det = fma(a, c, -(b * b));
    const Vec3 d = {ts.x - tm.x, ts.y - tm.y, ts.z - tm.z};
    const double e = dot3(hm, d), f = dot3(hs, d);
    const double inv = 1.0 / det;
    const double S0 = fma(c, e, -(b * f)) * inv;
    const double S1 = fma(a, f, -(b * e)) * inv;
    const Vec3 Wm = {fma(hm.x, S0, tm.x), fma(hm.y, S0, tm.y), fma(hm.z, S0, tm.z)};
    const Vec3 Ws = {fma(-hs.x, S1, ts.x), fma(-hs.y, S1, ts.y), fma(-hs.z, S1, ts.z)};
    const Vec3 df = {Wm.x - Ws.x, Wm.y - Ws.y, Wm.z - Ws.z};
    SkewOut o;
    o.dist = sqrt(dot3(df, df));
    o.W = {0.5 * (Wm.x + Ws.x), 0.5 * (Wm.y + Ws.y), 0.5 * (Wm.z + Ws.z)};
    // exactly singular as the reference's LU sees it for equal rays (H^T H = [[a, a], [a, a]]: u22 = a - a) and as the
    // oracle defines it (a c - b b == 0 in separately rounded products).  The FUSED determinant above is the rounding
    // error of b b when a c == b b mathematically, i.e. usually not zero: it serves the accuracy of S0 / S1, not this test.
    o.singular = (a * c == b * b);
    return o;
}

// triangulation.py:72-74: score = half / (dist * 1000), zeroed by the three gates (strict < and >,
// so NaN operands leave the score untouched, exactly like the NumPy comparisons).
template <typename TS>
__device__ __forceinline__ double pair_score(TS sm, TS ss, double dist, const Params &p) {
    double s = half_score(sm, ss) / (dist * 1000.0);
    if ((double)sm < p.kthr || (double)ss < p.kthr || dist > p.dthr) s = 0.0;
    return s;
}

// ---- fast reciprocal / reciprocal square root for the fused fast path ------------------------
// v_rcp_f64 / v_rsq_f64 deliver 2^-23 relative accuracy per the ISA (measured on MI355X: 5.2e-8 = 2^-24.2, tests/test_gpu_lean.py); each Newton step squares the error.
//   rcp_nr2 : two steps  -> ~1 ulp            (used for 1/det: feeds the 3D point)
//   rcp_nr1 : one step   -> ~2^-46 (1.4e-14)  (used for 1/sum(score): 1e-13 m on a 5 m coordinate)
//   rsq_nr1 : one step   -> ~2e-14            (used for 1/dist: only scales the score)
__device__ __forceinline__ double rcp_nr2(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rcp_nr1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
// two steps -> ~1 ulp; sqrt(x) = x * rsq(x) with one more correction (x > 0, normal range)
__device__ __forceinline__ double rsq_nr2(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(y * 0.5, fma(-(x * y), y, 1.0), y);
    y = fma(y * 0.5, fma(-(x * y), y, 1.0), y);
    return y;
}
__device__ __forceinline__ double sqrt_nr(double x) {
    const double y = rsq_nr2(x);
    const double g = x * y;                      // ~sqrt(x)
    return fma(fma(-g, g, x), y * 0.5, g);       // g + (x - g^2) / (2 sqrt(x))
}
__device__ __forceinline__ double rsq_nr1(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);  // 1 - x y^2
    return fma(y * 0.5, e, y);               // y (1 + e/2)
}

// volatile LDS reads in an explicit address space: single ds_read_b64 each (the compiler pairs neighbouring plain reads
// into ds_read2_b64, which the LDS serves at half the bytes per clock; a volatile GENERIC access would be a flat load)
typedef __attribute__((address_space(3))) const volatile double *lds_cv_f64;
typedef __attribute__((address_space(3))) const volatile unsigned long long *lds_cv_u64;

struct alignas(16) RayRec {  // one world ray in LDS: direction (un-normalised) and its squared norm
    double x, y, z, a;
};

// fast-math pair solve (k_frame_recompute phases 1/3, centre check of k_fused_single)
struct PairSolve {
    double score_base;  // idist * 0.001 (multiply by (sm+ss)/2)
    double d2;          // dist^2 (gate: dist > dthr  <=>  d2 > dthr2)
    Vec3 sw;            // Wm + Ws
    bool singular;
};

template <bool kNeedW>
__device__ __forceinline__ PairSolve pair_solve_fast(const RayRec &rm, const RayRec &rs, const Vec3 &d,
                                                     const Vec3 &tsum) {
    const double b = fma(rm.z, rs.z, fma(rm.y, rs.y, rm.x * rs.x));
    const double det = fma(rm.a, rs.a, -(b * b));
    const double e = fma(rm.z, d.z, fma(rm.y, d.y, rm.x * d.x));
    const double g = fma(rs.z, d.z, fma(rs.y, d.y, rs.x * d.x));
    const double inv = rcp_nr2(det);
    const double S0 = fma(rs.a, e, -(b * g)) * inv;
    const double S1 = fma(rm.a, g, -(b * e)) * inv;
    const Vec3 df = {fma(rs.x, S1, fma(rm.x, S0, -d.x)), fma(rs.y, S1, fma(rm.y, S0, -d.y)),
                     fma(rs.z, S1, fma(rm.z, S0, -d.z))};
    const double d2 = dot3(df, df);
    double idist = rsq_nr1(d2);
    idist = (d2 == 0.0) ? __builtin_inf() : idist;
    PairSolve o;
    o.d2 = d2;
    o.score_base = idist * 0.001;
    o.singular = (rm.a * rs.a == b * b);   // (see skew_ray_solve)
    if (kNeedW)
        o.sw = {fma(-rs.x, S1, fma(rm.x, S0, tsum.x)), fma(-rs.y, S1, fma(rm.y, S0, tsum.y)),
                fma(-rs.z, S1, fma(rm.z, S0, tsum.z))};
    return o;
}

template <typename TIn>
__device__ __forceinline__ RayRec make_ray(const double *__restrict__ M, TIn u, TIn v) {
    const Vec3 h = ray_from_pixel(M, (double)u, (double)v);
    return RayRec{h.x, h.y, h.z, dot3(h, h)};
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace snowtri

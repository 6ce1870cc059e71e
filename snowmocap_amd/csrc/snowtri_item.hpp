// snowtri_item.hpp -- the complete-graph item with its pair constants in LDS: one (frame or cluster, joint) = C rays, all
// C(C,2) pair solves of triangulation.py:24-31,70-75 and the score-weighted fusion of :136-149 regrouped per ray.
//
// Two kernels run it: k_cluster_fuse (the complete-graph clusters of the multi-person route, snowtri_cluster.hpp) and
// k_fused_single for rigs of five and more cameras with one detection each (snowtri_fused.hpp).  The fully unrolled items
// of the small rigs (lean_item, pairwise_item) keep every ray matrix, pair offset and -- pairwise_item -- every pair's
// determinant in registers: C(C,2) = 28 pairs at eight cameras do not fit 256 VGPRs (k_fused_single<8> spilled 1 075 of
// them, round-4 review).  Here a pair's offset d = t_s - t_m is read from LDS when the pair is solved, the ray matrices
// when a ray is built, and the pairs run in groups of four behind scheduling barriers: 136 VGPRs at eight cameras, no
// scratch, three waves per SIMD.
#pragma once
#include <type_traits>
#include <utility>

#include "snowtri_kernels.hpp"

namespace snowtri {

// camera pair q = (m, s) in the candidate order of triangulation.py:56-57, as compile-time tables (indexed by the
// unrolled pair counter: a loop that searches for q would leave the ray arrays dynamically indexed, i.e. in scratch)
template <int C>
struct ClusterPairs {
    static constexpr int NP = C * (C - 1) / 2;
    struct Tab {
        int m[NP > 0 ? NP : 1], s[NP > 0 ? NP : 1];
    };
    static constexpr Tab make() {
        Tab t{};
        int k = 0;
        for (int m = 0; m < C - 1; m++)
            for (int s = m + 1; s < C; s++, k++) {
                t.m[k] = m;
                t.s[k] = s;
            }
        return t;
    }
    static constexpr Tab tab = make();
};
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <typename F, int... I>
__device__ __forceinline__ void cluster_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void cluster_static_for(F &&f) {
    cluster_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One (cluster, joint): the item of k_fused_lean with the pair offsets read from LDS (28 pairs x 3 doubles do not fit
// the scalar registers), pairs in groups of four (register budget).  K = [M | t | d] in LDS at offset 0.
// Returns true if the joint needs the sequential routine (exact intersection, singular pair, NaN).
// TOut = float: 1/dist is the raw v_rsq_f64 (2^-24.2 relative, below the rounding of the stored score -- the contract of
// k_fused_lean); TOut = double: one Newton step on it (2e-14), as the float64 outputs of every other kernel.  The
// results come back in double; the caller rounds them to TOut when it stores.
// GROUP: pairs between two scheduling barriers -- 4 keeps the item at 136 VGPRs (three waves per SIMD: k_cluster_fuse); the lean
// kernels of 6-8 cameras run two waves per SIMD and give the scheduler larger groups to overlap the pairs' dependency chains.
template <int C, typename TIn, typename TOut, int GROUP = 4>
__device__ __forceinline__ bool cluster_item(const double *__restrict__ K, const Kp3<TIn> (&cur)[C], float kthr_f32, double kthr,
                                             double dthr2, double &ox, double &oy, double &oz, double &os) {
#pragma clang fp contract(off)
    constexpr int NP = C * (C - 1) / 2;
    constexpr int kGroup = GROUP;
    Vec3 h[C];
    double a[C], alpha[C], beta[C];
    TIn sg[C];          // the camera's confidence, -infinity where it is below the threshold (:73, once per camera)
    bool nanscore = false;
    cluster_static_for<C>([&](auto CC) {
        constexpr int c = CC;
        const double *M = K + 9 * c;
        const double u = (double)cur[c].u, v = (double)cur[c].v;
        h[c].x = fma(M[0], u, fma(M[1], v, M[2]));   // A1, camera.py:241-243 with M = R inv(K)
        h[c].y = fma(M[3], u, fma(M[4], v, M[5]));
        h[c].z = fma(M[6], u, fma(M[7], v, M[8]));
        a[c] = dot3(h[c], h[c]);
        if constexpr (sizeof(TIn) == 4)
            sg[c] = (float)cur[c].s < kthr_f32 ? -__builtin_huge_valf() : (float)cur[c].s;
        else
            sg[c] = (double)cur[c].s < kthr ? -__builtin_huge_val() : (double)cur[c].s;
        nanscore |= cur[c].s != cur[c].s;
        alpha[c] = 0.0;
        beta[c] = 0.0;
    });
    // per pair, without a reciprocal of the determinant (see lean_item): with n = h_m . (h_s x d) the distance of the two
    // rays is |n| / sqrt(det), so 1 / dist = det rsq(n^2 det) and the pair's weight times S0, S1, 1 is w N0, w N1, w det with
    // w = ssum rsq(n^2 det)
    cluster_static_for<(NP + kGroup - 1) / kGroup>([&](auto GG) {
        constexpr int q0 = kGroup * GG;
        constexpr int n = NP - q0 < kGroup ? NP - q0 : kGroup;
        __builtin_amdgcn_sched_barrier(0);   // a group's LDS reads and temporaries stay inside the group (register budget)
        cluster_static_for<n>([&](auto UU) {
            constexpr int u = UU, q = q0 + u, mc = ClusterPairs<C>::tab.m[q], sc = ClusterPairs<C>::tab.s[q];
            const Vec3 &hm = h[mc], &hs = h[sc];
            const double *dq = K + 12 * C + 3 * q;
            const double dx = dq[0], dy = dq[1], dz = dq[2];
            // A2 (triangulation.py:24-31)
            const double b = dot3(hm, hs);
            // det with SEPARATELY rounded products (contraction is off): a pair that is singular as the reference sees it
            // (a c == b b, skew_ray_solve) has det == 0 exactly -> rho = inf -> the sum is not finite -> the sequential
            // routine takes the joint and flags it.  The fused form is the rounding error of b b there: usually not 0, and
            // positive often enough (18 % of equal rays) for a finite, wrong joint to pass (round-5 advice).
            const double det = a[mc] * a[sc] - b * b;
            const double e = fma(hm.z, dz, fma(hm.y, dy, hm.x * dx));
            const double g = fma(hs.z, dz, fma(hs.y, dy, hs.x * dx));
            const double N0 = fma(a[sc], e, -(b * g));
            const double N1 = fma(a[mc], g, -(b * e));
            const double cx = fma(hs.y, dz, -(hs.z * dy)), cy = fma(hs.z, dx, -(hs.x * dz)), cz = fma(hs.x, dy, -(hs.y * dx));
            const double nn = fma(hm.z, cz, fma(hm.y, cy, hm.x * cx));
            const double n2 = nn * nn;
            double rho;
            if constexpr (sizeof(TOut) == 4)
                rho = __builtin_amdgcn_rsq(n2 * det);
            else
                rho = rsq_nr1(n2 * det);   // (n2 det == 0: inf -> NaN here; either way the sum is not finite and the joint is re-done)
            // :72-74, w det = 2000 x the pair score.  The gates without compares (gated_weight, snowtri_math.hpp): a gated
            // confidence is -infinity, the distance gate the sign of fma(det, dthr2, -n2), one v_max with 0 -- the sum of a
            // kept pair is the float32 (float64) sum NumPy takes, a gated pair weighs exactly +0 as the select it replaces
            const double w = gated_weight(sg[mc], sg[sc], fma(det, dthr2, -n2)) * rho;
            alpha[mc] = fma(w, N0, alpha[mc]);
            alpha[sc] = fma(-w, N1, alpha[sc]);
            beta[mc] = fma(w, det, beta[mc]);
            beta[sc] = fma(w, det, beta[sc]);
        });
    });
    __builtin_amdgcn_sched_barrier(0);
    const double *tp = K + 9 * C;
    double sx = 0.0, sy = 0.0, sz = 0.0, sb = 0.0;
    cluster_static_for<C>([&](auto CC) {
        constexpr int c = CC;
        sx = fma(alpha[c], h[c].x, fma(beta[c], tp[3 * c + 0], sx));
        sy = fma(alpha[c], h[c].y, fma(beta[c], tp[3 * c + 1], sy));
        sz = fma(alpha[c], h[c].z, fma(beta[c], tp[3 * c + 2], sz));
        sb += beta[c];
    });
    // sb = 2 x 2000 x sum_q s_q (:141); sum == 0 -> (0,0,0)/0 (:142-143): sx = sy = sz = 0 then
    const double r = rcp_nr1(fmax(sb, 1e-300));
    ox = sx * r;   // :144-147 as (sum s (Wm+Ws)) / (2 sum s)
    oy = sy * r;
    oz = sz * r;
    os = sb * (0.00025 / (double)NP);   // :148
    return !(sb < 1e300) || nanscore;   // (a NaN confidence does not survive the v_max: the sequential routine takes the joint)
}

}  // namespace snowtri

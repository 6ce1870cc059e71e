// Dev aid: what does the instruction mix of k_candidate_sums' solve loop cost on gfx950, instruction class by class?
// The 2 x 4 tile of p1_tile_sums (snowtri_assoc.hpp) with its operands in REGISTERS (no LDS, no memory in the loop), at the
// kernel's occupancy (256 threads, 3 workgroups per CU), in variants that drop or replace one class of instructions each:
// the differences are the marginal cost of that class in SIMD cycles per candidate.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/sums_mix scripts/ubench/sums_mix.hip && /tmp/sums_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Ray { double x, y, z, a; };
// DIST: 0 no distance gate | 1 dn2 > det * dthr2 (mul + cmp, the kernel's) | 2 !(w >= 1 / dthr) on the finished score (cmp)
// SCORE: 0 weight 1.0 | 1 float sum + v_cndmask + cvt (the kernel's) | 2 double sum + two v_cndmask | 3 double sum, gate by multiplying w with a 0/1 mask
// KTHR: the two keypoint-threshold compares; RSQ: 1 v_rsq_f64 | 0 a v_mul_f64 in its place | 2 v_rsq_f32 on a converted operand + cvt back (timing only)
template <int DIST, int SCORE, bool KTHR, int RSQ, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(double *out, int iters, Ray ra, Ray rb, double dthr2, float kthr, unsigned long long *clk) {
    constexpr int GA = 2, GS = 4;
    Ray a[GA], b[GS];
    float sm[GA], ss[GS];
    double smd[GA], ssd[GS];
    const double e = threadIdx.x * 1e-9;
    for (int i = 0; i < GA; i++) { a[i] = {ra.x + e, ra.y + i, ra.z, ra.a}; sm[i] = 4.f + i; smd[i] = sm[i]; }
    for (int u = 0; u < GS; u++) { b[u] = {rb.x - e, rb.y + u, rb.z, rb.a}; ss[u] = 5.f + u; ssd[u] = ss[u]; }
    double acc[GA * GS];
    for (int q = 0; q < GA * GS; q++) acc[q] = 0.0;
    const double dx = 0.3 + e, dy = -0.2, dz = 0.1, inv_dthr = 20.0;
    unsigned long long t0 = 0, w0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_readcyclecounter(); w0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < GS; u++) asm volatile("" : "+v"(b[u].x), "+v"(b[u].y), "+v"(b[u].z), "+v"(b[u].a), "+v"(ss[u]), "+v"(ssd[u]));
#pragma unroll
        for (int i = 0; i < GA; i++) {
            asm volatile("" : "+v"(a[i].x), "+v"(a[i].y), "+v"(a[i].z), "+v"(a[i].a), "+v"(sm[i]), "+v"(smd[i]));
            const double cx = fma(dy, a[i].z, -(dz * a[i].y)), cy = fma(dz, a[i].x, -(dx * a[i].z)), cz = fma(dx, a[i].y, -(dy * a[i].x));
            const bool okm = KTHR ? !(sm[i] < kthr) : true;
#pragma unroll
            for (int u = 0; u < GS; u++) {
                const double bq = fma(a[i].z, b[u].z, fma(a[i].y, b[u].y, a[i].x * b[u].x));
                const double det = fma(a[i].a, b[u].a, -(bq * bq));
                const double dn = fma(cz, b[u].z, fma(cy, b[u].y, cx * b[u].x));
                const double dn2 = dn * dn;
                double w;
                if (RSQ == 1) w = det * __builtin_amdgcn_rsq(dn2 * det);
                else if (RSQ == 2) w = det * (double)__builtin_amdgcn_rsqf((float)(dn2 * det));
                else w = det * ((dn2 * det) * dn2);
                bool kp = okm && (KTHR ? !(ss[u] < kthr) : true);
                if (DIST == 1) kp = kp && !(dn2 > det * dthr2);
                if (DIST == 2) kp = kp && !(w < inv_dthr);
                if (SCORE == 0) acc[i * GS + u] += kp ? w : 0.0;
                if (SCORE == 1) acc[i * GS + u] = fma((double)(kp ? sm[i] + ss[u] : 0.f), w, acc[i * GS + u]);
                if (SCORE == 2) acc[i * GS + u] = fma(kp ? smd[i] + ssd[u] : 0.0, w, acc[i * GS + u]);
                if (SCORE == 3) acc[i * GS + u] = fma(smd[i] + ssd[u], kp ? w : 0.0, acc[i * GS + u]);
                if (SCORE == 4) {   // scores gated when the record is written (a large negative sentinel): no threshold compare here
                    const float sel = fmaxf(sm[i] + ss[u], 0.f);
                    acc[i * GS + u] = fma((double)(kp ? sel : 0.f), w, acc[i * GS + u]);
                }
                if (SCORE == 5) {   // the same with double scores: select on the HIGH word only, then max with 0
                    const double sd = smd[i] + ssd[u];
                    const int hi = kp ? __double2hiint(sd) : (int)0xfff00000;
                    acc[i * GS + u] = fma(fmax(__hiloint2double(hi, __double2loint(sd)), 0.0), w, acc[i * GS + u]);
                }
                if (SCORE == 6) {   // float scores, the gate as a second sentinel: one v_max after the select
                    const float sel = kp ? sm[i] + ss[u] : -1e30f;
                    acc[i * GS + u] = fma((double)fmaxf(sel, 0.f), w, acc[i * GS + u]);
                }
            }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - w0; }
    double s = 0;
    for (int q = 0; q < GA * GS; q++) s += acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// plain chains of one instruction (ILP 8) at the same occupancy: the reference points
template <int MODE>
__global__ __launch_bounds__(256, 3) void chain(double *out, int iters, double p, double q, float kthr) {
    double x[8];
    float f[8];
    for (int i = 0; i < 8; i++) { x[i] = p + i + threadIdx.x * 1e-9; f[i] = (float)x[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) x[i] = fma(x[i], q, p);
                if (MODE == 1) x[i] = __builtin_amdgcn_rsq(x[i]);
                if (MODE == 2) { f[i] += kthr; x[i] += (double)f[i]; }                       // v_add_f32 + v_cvt_f64_f32 + v_add_f64
                if (MODE == 3) { f[i] += kthr; x[i] += q; }                                   // v_add_f32 + v_add_f64
                if (MODE == 4) x[i] = (x[i] > p) ? x[i] * q : x[i];                           // v_cmp_gt_f64 + v_mul_f64 + 2 v_cndmask
                if (MODE == 5) { f[i] = (x[i] > p) ? f[i] + kthr : 0.f; x[i] += q; }          // v_cmp_gt_f64 + v_add_f32 + v_cndmask + v_add_f64
                if (MODE == 6) { x[i] = x[i] * q; }
                if (MODE == 7) { x[i] = x[i] + q; }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_mhz = 0;
template <int DIST, int SCORE, bool KTHR, int RSQ, int WAVES = 3>
int run(const char *name) {
    const int blocks = 256 * WAVES, iters = 4000;
    double *d;
    unsigned long long *clk, h[2];
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CHECK(hipMalloc(&clk, 16));
    const Ray ra{0.1, 0.2, 0.97, 1.0}, rb{-0.2, 0.1, 0.96, 1.0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) k<DIST, SCORE, KTHR, RSQ, WAVES><<<blocks, 256>>>(d, iters, ra, rb, 0.0025, 3.5f, clk);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<DIST, SCORE, KTHR, RSQ, WAVES><<<blocks, 256>>>(d, iters, ra, rb, 0.0025, 3.5f, clk);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;   // s_memtime ticks per microsecond of s_memrealtime (100 MHz)
    g_mhz = mhz;
    // per SIMD: WAVES waves x iters tiles x 8 candidates; time in SIMD cycles at 2.4 GHz nominal and at the counter's rate
    const double ns_per_cand = ms * 1e6 / ((double)WAVES * iters * 8);
    printf("%-64s %8.3f ms  %6.2f ns per candidate and SIMD = %6.2f cycles @2.4 GHz  (counter %.0f MHz)\n", name, ms, ns_per_cand, ns_per_cand * 2.4, mhz);
    hipFree(d); hipFree(clk);
    return 0;
}
template <int MODE>
int run_chain(const char *name, int instr) {
    const int blocks = 256 * 3, iters = 4000;
    double *d;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) chain<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 0.9999999, 1e-3f);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    chain<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 0.9999999, 1e-3f);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (3.0 * iters * 64);
    printf("%-64s %8.3f ms  %6.2f ns per group and SIMD = %6.2f cycles @2.4 GHz (%d instructions: %.2f each)\n", name, ms, ns, ns * 2.4, instr, ns * 2.4 / instr);
    hipFree(d);
    return 0;
}

int main() {
    run_chain<0>("chain v_fma_f64", 1);
    run_chain<6>("chain v_mul_f64", 1);
    run_chain<7>("chain v_add_f64", 1);
    run_chain<1>("chain v_rsq_f64", 1);
    run_chain<3>("chain v_add_f32 + v_add_f64", 2);
    run_chain<2>("chain v_add_f32 + v_cvt_f64_f32 + v_add_f64", 3);
    run_chain<4>("chain v_cmp_gt_f64 + v_mul_f64 + 2 v_cndmask_b32", 4);
    run_chain<5>("chain v_cmp_gt_f64 + v_add_f32 + v_cndmask_b32 + v_add_f64", 4);
    run<1, 1, true, 1>("tile as in the kernel (dist gate mul+cmp, f32 score, kthr, rsq)");
    run<1, 1, true, 1, 2>("  the same at two waves per SIMD");
    run<1, 1, true, 1, 4>("  the same at four waves per SIMD (register cap 128)");
    run<0, 1, true, 1>("  without the distance gate");
    run<2, 1, true, 1>("  distance gate on the finished score (one cmp)");
    run<1, 1, false, 1>("  without the keypoint-threshold compares");
    run<1, 0, true, 1>("  weight 1 (no score sum, select on w)");
    run<1, 2, true, 1>("  double scores: v_add_f64 + 2 v_cndmask");
    run<1, 3, true, 1>("  double scores: v_add_f64, select on w");
    run<2, 3, true, 1>("  double scores, select on w, gate on the finished score");
    run<1, 1, true, 0>("  v_mul_f64 x2 in place of v_rsq_f64");
    run<1, 1, true, 2>("  v_rsq_f32 between two conversions in place of v_rsq_f64");
    run<1, 4, false, 1>("  pre-gated f32 scores (add, max, select, cvt), gate mul+cmp");
    run<2, 4, false, 1>("  pre-gated f32 scores, gate on the finished score");
    run<2, 6, false, 1>("  pre-gated f32 scores, select then max, gate on the finished score");
    run<2, 5, false, 1>("  pre-gated f64 scores (add, select hi, max), gate on the finished score");
    run<1, 5, false, 1>("  pre-gated f64 scores (add, select hi, max), gate mul+cmp");
    run<0, 4, false, 1>("  pre-gated f32 scores, no distance gate");
    run<0, 0, false, 1>("  arithmetic only (no gate, no score)");
    run<0, 0, false, 0>("  arithmetic only, no rsq");
    return 0;
}

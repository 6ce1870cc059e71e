// Dev aid: fp64 VALU issue-rate microbenchmark on gfx950 (fma / mul / add / rcp / cvt / v_mov mixes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP, int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MODE == 0) x[i] = fma(x[i], b, a);                  // v_fma_f64
                if (MODE == 1) x[i] = x[i] * b;                          // v_mul_f64
                if (MODE == 2) x[i] = x[i] + b;                          // v_add_f64
                if (MODE == 3) x[i] = __builtin_amdgcn_rcp(x[i]);        // v_rcp_f64
                if (MODE == 4) { float f = (float)x[i]; x[i] = (double)(f + 1.0f); }  // cvt + f32 add + cvt
                if (MODE == 5) x[i] = __builtin_amdgcn_rsq(x[i]);        // v_rsq_f64
                if (MODE == 6) x[i] = fma(x[i], x[(i + 1) % ILP], x[(i + 2) % ILP]);  // three VGPR-pair operands
                if (MODE == 7) {  // the solve's mix: 40 fma/mul/add per 2 transcendental ops (counted as 42 instructions)
#pragma unroll
                    for (int q = 0; q < 20; q++) x[i] = fma(x[i], b, a);
                    x[i] = __builtin_amdgcn_rsq(x[i]);
#pragma unroll
                    for (int q = 0; q < 20; q++) x[i] = fma(x[i], b, a);
                    x[i] = __builtin_amdgcn_rcp(x[i]);
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, int MODE>
int run(const char *name, int blocks_per_cu, int instr_per_elem) {
    int ncu = 256;
    int blocks = ncu * blocks_per_cu;
    double *d;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP, MODE><<<blocks, 256>>>(d, 10, 1.0000001, 0.9999999);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<ILP, MODE><<<blocks, 256>>>(d, iters, 1.0000001, 0.9999999);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * 16 * ILP * instr_per_elem;   // wave-instructions
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("%-28s ILP=%d waves/SIMD=%d : %.3f ms  -> %.2f Gwave-instr/s per SIMD = %.2f cycles/instr @2.4GHz (%.1f TFLOP/s if fma)\n",
           name, ILP, blocks_per_cu, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 * 2 / (ms * 1e-3) / 1e12);
    hipFree(d);
    return 0;
}

int main() {
    for (int occ : {1, 2, 4, 8}) {
        if (occ == 1) { run<1, 0>("fma dep-chain", 1, 1); run<4, 0>("fma", 1, 1); run<8, 0>("fma", 1, 1); }
        if (occ == 2) { run<1, 0>("fma dep-chain", 2, 1); run<4, 0>("fma", 2, 1); run<8, 0>("fma", 2, 1); }
        if (occ == 4) { run<4, 0>("fma", 4, 1); }
        if (occ == 8) { run<4, 0>("fma", 8, 1); }
    }
    run<8, 1>("mul", 2, 1);
    run<8, 2>("add", 2, 1);
    run<8, 3>("rcp (trans)", 2, 1);
    run<8, 3>("rcp (trans)", 8, 1);
    run<8, 4>("cvt+addf32+cvt (3 instr)", 2, 3);
    run<8, 5>("rsq (trans)", 2, 1);
    run<8, 6>("fma 3 vgpr operands", 2, 1);
    run<2, 0>("fma ILP2", 1, 1);
    run<2, 0>("fma ILP2", 2, 1);
    run<1, 0>("fma dep-chain", 3, 1);
    run<2, 0>("fma ILP2", 3, 1);
    run<3, 0>("fma ILP3", 1, 1);
    run<6, 7>("mix 40 fma + rsq + rcp", 1, 42);
    run<6, 7>("mix 40 fma + rsq + rcp", 2, 42);
    run<6, 7>("mix 40 fma + rsq + rcp", 3, 42);
    run<1, 1>("mul dep-chain", 1, 1);
    run<1, 2>("add dep-chain", 1, 1);
    return 0;
}

// Dev aid (GPU box): a shared library with ONE C entry that keeps the fp64 VALUs busy for a given time, so that a
// sampler in the calling process (scripts/power_trace.py) can read the clock and the socket power the chip settles at
// under a pure fp64 issue load -- the comparison point for the clock the triangulation kernels hold.
//   mode 0: v_fma_f64 only (8 independent chains per lane, two workgroups of 256 per CU... `wg_per_cu` of them)
//   mode 1: the pair solve's mix, 40 fma per rsq + rcp
//   mode 2: v_fma_f64 beside a streaming read of `stream_bytes` per launch (fp64 issue + HBM traffic together)
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libfp64_burn.so scripts/ubench/fp64_burn.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>

namespace {

template <int MODE>
__global__ __launch_bounds__(256) void k_burn(double *out, int iters, double a, double b, const float4 *src, int64_t n4) {
    constexpr int ILP = 8;
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = a + threadIdx.x * 1e-9 + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 2 && p < n4) {
            const float4 v = src[p];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            p += stride;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 20; q++) x[i] = fma(x[i], b, a);
                    x[i] = __builtin_amdgcn_rsq(x[i]);
#pragma unroll
                    for (int q = 0; q < 20; q++) x[i] = fma(x[i], b, a);
                    x[i] = __builtin_amdgcn_rcp(x[i]);
                } else {
                    x[i] = fma(x[i], b, a);
                }
            }
        }
    }
    double s = (double)(acc.x + acc.y + acc.z + acc.w);
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

// Runs launches back to back for `seconds`; returns the VALU wave-instructions issued per second per SIMD (1024 SIMDs),
// i.e. the effective issue clock / 4 for full-rate instructions, or a negative HIP error code.
extern "C" double fp64_burn(int mode, double seconds, int wg_per_cu, int64_t stream_bytes) {
    const int blocks = 256 * (wg_per_cu > 0 ? wg_per_cu : 2);
    double *d = nullptr;
    float4 *src = nullptr;
    if (hipMalloc(&d, sizeof(double) * blocks * 256) != hipSuccess) return -1.0;
    int64_t n4 = 0;
    if (mode == 2 && stream_bytes > 0) {
        n4 = stream_bytes / 16;
        if (hipMalloc(&src, (size_t)n4 * 16) != hipSuccess) return -2.0;
        (void)hipMemset(src, 0, (size_t)n4 * 16);
    }
    // mode 2: every iteration reads 16 bytes per lane; iterations so that one launch walks the whole buffer once
    int iters = 400;
    if (mode == 2 && n4 > 0) iters = (int)((n4 + (int64_t)blocks * 256 - 1) / ((int64_t)blocks * 256));
    const int per_iter = mode == 1 ? 16 * 8 * 42 : 16 * 8;
    double launches = 0.0;
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double elapsed = 0.0;
    while (elapsed < seconds) {
        for (int r = 0; r < 8; r++) {
            if (mode == 1)
                hipLaunchKernelGGL(k_burn<1>, dim3(blocks), dim3(256), 0, nullptr, d, iters, 1.0000001, 0.9999999, src, n4);
            else if (mode == 2)
                hipLaunchKernelGGL(k_burn<2>, dim3(blocks), dim3(256), 0, nullptr, d, iters, 1.0000001, 0.9999999, src, n4);
            else
                hipLaunchKernelGGL(k_burn<0>, dim3(blocks), dim3(256), 0, nullptr, d, iters, 1.0000001, 0.9999999, src, n4);
        }
        launches += 8.0;
        if (hipDeviceSynchronize() != hipSuccess) return -3.0;
        elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    (void)hipFree(d);
    if (src) (void)hipFree(src);
    const double wave_instr = launches * (double)blocks * 4.0 * (double)iters * (double)per_iter;
    return wave_instr / elapsed / 1024.0;
}

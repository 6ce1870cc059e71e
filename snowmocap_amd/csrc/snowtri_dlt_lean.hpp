// snowtri_dlt_lean.hpp -- k_dlt_coop: method = SNOWTRI_DLT (row N3) with ONE detection per camera on the frame of
// k_fused_lean_coop (snowtri_lean.hpp) instead of k_fused_single's.
//
// k_fused_single<C,1> ran the DLT item at VALU busy 0.46-0.48 (profiles/r06/dlt_counters.txt): a tile of frames per workgroup
// with three barriers, an epilogue on a few lanes, 64-bit address arithmetic and an item -> (frame, joint) walk per item.
// The item (dlt_item, snowtri_fused.hpp) is unchanged; around it, as in k_fused_lean_coop:
//   * a workgroup owns a tile of <= kCoopMaxFrames frames, its four waves split the tile's 64-item PASSES evenly;
//   * keypoints through BUFFER loads whose descriptor covers the tile (item -> byte offset from an LDS table: no division, no
//     64-bit arithmetic, lanes past the tile's end read zeros and their stores are dropped), two buffers per lane;
//   * joint records by non-temporal buffer stores, the joint scores into a workgroup-wide LDS stash;
//   * ONE barrier, then the frames' mean scores dealt to the waves by passes of 16 frames; count = 1, flags = FASTPATH
//     (DLT never speculates: there is no check and no fall-back).
// Shapes: keypoint_num == J == JC (133) known at compile time, one output slot; everything else stays on k_fused_single<C,1>.
// The cameras that list a detection (n_persons > 0) come from one LDS word per frame; without an n_persons array every camera counts.
#pragma once
#include "snowtri_lean.hpp"

namespace snowtri {

constexpr int kDltCoopWaves = 3;   // waves per SIMD the kernel is compiled for (dlt_item with P read per camera: ~140 VGPRs)
__host__ __device__ constexpr size_t dlt_coop_lds_bytes(int C, int JC, int nf_max, int score_bytes = 4) {
    // [P[C][12] | item -> input offset table (+ 128 entries of prefetch distance) | stash | mean per frame | detection mask per frame]
    return (((size_t)96 * C + (size_t)4 * (lean_coop_items_pad(JC, nf_max) + 128) + (size_t)score_bytes * lean_coop_items_pad(JC, nf_max) +
             (size_t)12 * kCoopMaxFrames + 16) + 15) & ~(size_t)15;
}

template <int C, typename TIn, int JC, typename TOut>
__global__ __launch_bounds__(kBlock, kDltCoopWaves) void k_dlt_coop(int64_t F, int tile_base, int64_t tile_rem, int nf_max, Rig rig,
                                                         const TIn *__restrict__ kpts, const int32_t *__restrict__ n_persons, Params prm,
                                                         TOut *__restrict__ out4, TOut *__restrict__ out_ps, int32_t *__restrict__ out_count,
                                                         uint32_t *__restrict__ out_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned kRec = (unsigned)sizeof(Kp3<TIn>);
    constexpr unsigned kCamStride = (unsigned)JC * kRec;
    constexpr unsigned kOutRec = 4u * (unsigned)sizeof(TOut);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int items_pad = lean_coop_items_pad(JC, nf_max), ntable = items_pad + 128;
    double *Pl = reinterpret_cast<double *>(smem);
    uint32_t *table = reinterpret_cast<uint32_t *>(Pl + 12 * C);
    TOut *stash = reinterpret_cast<TOut *>(table + ntable);
    double *favg = reinterpret_cast<double *>(stash + items_pad);               // [kCoopMaxFrames]
    uint32_t *fmask = reinterpret_cast<uint32_t *>(favg + kCoopMaxFrames);      // [kCoopMaxFrames] bit c: camera c lists a detection
    const Kp3<TIn> *kp3 = reinterpret_cast<const Kp3<TIn> *>(kpts);
    Kp3<TIn> bufA[C], bufB[C];

    int64_t f0;
    int nf;
    lean_tile_range((int64_t)blockIdx.x, tile_base, tile_rem, f0, nf);
    SNOWTRI_DEV_CHECK(f0 >= 0 && nf >= 1 && nf <= nf_max && nf_max <= kCoopMaxFrames && f0 + nf <= F, 34);
    // passes of the tile, dealt to the waves as in k_fused_lean_coop
    const int npass_tile = (nf * JC + 63) >> 6;
    const int pos = (wave + 2 * (int)(blockIdx.x & 1u)) & (kLeanWaves - 1);
    const int pq = npass_tile / kLeanWaves, pr = npass_tile - pq * kLeanWaves;
    const int p0 = pos * pq + (pos < pr ? pos : pr), npass = pq + (pos < pr ? 1 : 0);
    const unsigned i0 = (unsigned)p0 * 64u;
    const __amdgpu_buffer_rsrc_t rin = lean_rsrc(kp3 + f0 * (int64_t)(C * JC), (unsigned)(nf * C * JC) * kRec);
    const __amdgpu_buffer_rsrc_t rout = lean_rsrc(reinterpret_cast<char *>(out4) + f0 * (int64_t)JC * kOutRec, (unsigned)(nf * JC) * kOutRec);
    auto fetch = [&](Kp3<TIn>(&dst)[C], unsigned voff) {
#pragma unroll
        for (int c = 0; c < C; c++)
            dst[c] = lean_load_kp3<TIn>(rin, voff + (unsigned)(c & 1) * kCamStride, (unsigned)(c & ~1) * kCamStride);
    };
    auto item_offset = [&](unsigned i) { return (i + (i / (unsigned)JC) * (unsigned)((C - 1) * JC)) * kRec; };
    // the constants are requested BEFORE the first keypoints (the vector-memory counter returns in order)
    const double cP = rig.P[tid < 12 * C ? tid : 0];
    uint32_t cmask = 0xffffu;
    if (n_persons && tid < nf) {
        cmask = 0u;
        for (int c = 0; c < C; c++) cmask |= n_persons[(f0 + tid) * C + c] > 0 ? (1u << c) : 0u;
    }
    constexpr unsigned kNoItem = 0x40000000u;   // beyond every descriptor: the load returns zeros without touching memory
    fetch(bufA, npass > 0 ? item_offset(i0 + (unsigned)lane) : kNoItem);
    for (unsigned i = (unsigned)tid; i < (unsigned)ntable; i += kBlock) table[i] = item_offset(i);
    if (tid < 12 * C) Pl[tid] = cP;
    if (tid < kCoopMaxFrames) fmask[tid] = cmask;
    const bool masked = n_persons != nullptr;   // (uniform)
    __syncthreads();

    auto solve_store = [&](const Kp3<TIn>(&buf)[C], unsigned out_off, TOut *stash_slot) {
        uint32_t m = 0xffffu;
        if (masked) {
            unsigned o = out_off;
            const unsigned fl = (o / kOutRec) / (unsigned)JC;
            m = fmask[fl < (unsigned)kCoopMaxFrames ? fl : 0u];
        }
        double x, y, z, os;
        asm volatile("" ::: "memory");   // (P is read from LDS by every item)
        dlt_item<C, TIn>(Pl, buf, m, prm, x, y, z, os);
        if constexpr (sizeof(TOut) == 4) {
            const float osf = (float)os;
            lean_u4 rec;
            rec.x = __float_as_uint((float)x);
            rec.y = __float_as_uint((float)y);
            rec.z = __float_as_uint((float)z);
            rec.w = __float_as_uint(osf);
            __builtin_amdgcn_raw_buffer_store_b128(rec, rout, (int)out_off, 0, kLeanStoreAux);
            *stash_slot = osf;
        } else {
            lean_u4 lo, hi;
            lo.x = (unsigned)__double2loint(x);
            lo.y = (unsigned)__double2hiint(x);
            lo.z = (unsigned)__double2loint(y);
            lo.w = (unsigned)__double2hiint(y);
            hi.x = (unsigned)__double2loint(z);
            hi.y = (unsigned)__double2hiint(z);
            hi.z = (unsigned)__double2loint(os);
            hi.w = (unsigned)__double2hiint(os);
            __builtin_amdgcn_raw_buffer_store_b128(lo, rout, (int)out_off, 0, kLeanStoreAux);
            __builtin_amdgcn_raw_buffer_store_b128(hi, rout, (int)out_off + 16, 0, kLeanStoreAux);
            *stash_slot = os;
        }
    };
    // ---- item loop over the wave's passes: the next item's keypoints fly under the current item
    {
        unsigned out_off = (i0 + (unsigned)lane) * kOutRec;
        TOut *sp = stash + i0 + lane;
        const uint32_t *tp = table + i0 + lane;
        for (int k = 0; k < npass; k += 2) {
            fetch(bufB, k + 1 < npass ? tp[64] : kNoItem);
            solve_store(bufA, out_off, sp);
            fetch(bufA, k + 2 < npass ? tp[128] : kNoItem);
            if (k + 1 < npass) solve_store(bufB, out_off + 64u * kOutRec, sp + 64);
            tp += 128;
            out_off += 128u * kOutRec;
            sp += 128;
        }
    }
    __syncthreads();   // the tile's joint scores are in the stash
    // ---- the frames' mean scores (the person score of the DLT definition: mean of the keypoint_num joint scores), 16 frames per pass
    {
        constexpr int G = 4;
        const int nmean = (nf + 64 / G - 1) / (64 / G);
        for (int pass = pos; pass < nmean; pass += kLeanWaves) {
            const int w = pass * (64 / G) + lane / G, sub = lane & (G - 1);
            const bool live = w < nf;
            double sum = 0.0;
            if (live) sum = lean_row_partial<JC, TOut>(stash + w * JC, sub);
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (live && sub == 0) {
                const int64_t f = f0 + w;
                out_count[f] = 1;
                if (out_ps) out_ps[f] = (TOut)(sum / (double)JC);
                if (out_flags) out_flags[f] = kFlagFast;
            }
        }
    }
}

}  // namespace snowtri

// snowtri.hip -- C ABI of libsnowtri.so (declared in include/snowtri.h) over the HIP kernels.
//
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC snowtri.hip
// No torch types, no exceptions across the boundary; every entry point returns a snowtri_status.
#include "../../include/snowtri.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "snowtri_fused.hpp"
#include "snowtri_lean.hpp"
#include "snowtri_cluster.hpp"
#include "snowtri_general.hpp"
#include "snowtri_assoc.hpp"
#include "snowtri_dlt_lean.hpp"
#include "snowtri_smooth.hpp"
#include "snowtri_blender.hpp"
#include "snowtri_undistort.hpp"
#include "snowtri_kernels.hpp"

using namespace snowtri;

namespace {

thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            char _buf[512];                                                                      \
            snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                     __FILE__, __LINE__);                                                        \
            g_last_error = _buf;                                                                 \
            return SNOWTRI_ERR_HIP;                                                              \
        }                                                                                        \
    } while (0)

// Every entry point runs on its context's device and leaves the caller's current device as it found it
// (one process may drive several GPUs, or torch may have selected another device for the thread).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err != hipSuccess) prev = -1;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ENTER_DEVICE(device)              \
    DeviceGuard _device_guard(device);    \
    HIP_TRY(_device_guard.err)

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SNOWTRI_OK;
        if (p) {
            HIP_TRY(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        size_t want = std::max(bytes, (size_t)1 << 20);
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return SNOWTRI_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Page-locked host staging for the small per-frame calls: one H2D and one D2H per call, both truly asynchronous
// (a pageable copy is staged and synchronised by the runtime, ~10-20 us each).
struct PinnedScratch {
    void *p = nullptr;
    void *dev = nullptr;   // the same memory as the device sees it (mapped: the small calls' kernels read / write it in place)
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SNOWTRI_OK;
        if (p) {
            HIP_TRY(hipHostFree(p));
            p = dev = nullptr;
            cap = 0;
        }
        size_t want = std::max(bytes, (size_t)64 << 10);
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&dev, p, 0));
        std::memset(p, 0, want);   // (completion words are compared with a sequence number: never start from stale memory)
        cap = want;
        return SNOWTRI_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        cap = 0;
    }
};
constexpr size_t kPinnedMaxBytes = (size_t)4 << 20;   // larger host batches keep the direct copies
// The per-frame calls (main.py hands over 6 KB of keypoints and takes back 30 KB of candidates): below this size the kernels
// read their inputs from, and mirror their outputs into, the mapped page-locked buffers directly -- no copy engine in the chain
// (H2D + D2H were ~20 of the 39 us of a snowtri_triangulate call).
constexpr size_t kZeroCopyMaxBytes = (size_t)256 << 10;

int inv3(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c01 = m[5] * m[6] - m[3] * m[8];
    const double c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0 || !std::isfinite(det)) return 1;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 0;
}

constexpr int kTimingRing = 1024;
// ctx->d_counters: [0] singular pairs, [1] slow frames, [2] frame queue of k_frame_recompute, [6..9] slow / exact / slow
// (second pass) frame counts and the frame tickets of k_candidate_sums, [16 .. 32] the two hand-over list counters
// (snowtri_cluster.hpp: kHandComplete, kHandMembers -- 128 bytes apart)
constexpr int kHandCountersAt = 16, kCounterWords = 48, kTriangulateSingularAt = 40, kHostDoneCountAt = 42;   // [40]: singular pairs of the zero-copy snowtri_triangulate (self-resetting)

// Everything a fused call writes besides its outputs: the slabs of the fall-back routines, the hand-over lists and
// candidate sums of the multi-person path, the frame queue / list counters, the flags of a call that did not ask for
// them.  A context owns kMaxSets of them: set 0 is the caller's stream; the others carry an internal stream each, so that
// (a) the segments of ONE multi-person call alternate between two sets -- the latency-bound kernels of one segment
// (k_associate, the member lists) run beside the VALU-bound ones of the other -- and (b) in overlap mode
// (snowtri_ctx_set_overlap) consecutive calls run on different sets, i.e. the tail of one launch overlaps the ramp-up of
// the next without the caller managing streams.  Work on one set is ordered by its stream, so a set's scratch is never
// shared by two launches in flight.
struct StreamSet {
    Scratch work, desc, sums, misc;
    unsigned long long *d_counters = nullptr;   // kCounterWords
    hipStream_t stream = nullptr;               // internal stream (set 0 never has one: it is ordered by the caller's stream)
    hipEvent_t done = nullptr;                  // recorded behind the last launch the set received
    bool pending = false;                       // `done` has been recorded and not yet joined
};
constexpr int kMaxSets = 5;   // set 0 = the caller's stream; sets 1..4 carry an internal stream each
constexpr int kMaxOverlap = kMaxSets - 1;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the FUNCTION (per device, process-wide) and the call SETS
// it: two contexts with different rigs in one process must not lower each other's limit.  One monotone
// cache per (device, kernel) for the whole process.
std::mutex g_lds_mutex;
std::map<std::pair<int, const void *>, int> g_lds_raised;
int raise_dynamic_lds(int device, const void *kern, int lds) {
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    int &have = g_lds_raised[{device, kern}];
    if (have >= lds) return 0;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    have = lds;
    return 0;
}

int grid_for(int64_t work_items, int per_block, int cap_blocks) {
    int64_t b = (work_items + per_block - 1) / per_block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, cap_blocks));
}

}  // namespace

struct snowtri_ctx {
    int device = 0;
    int32_t C = 0, npairs = 0;
    int num_cus = 256;
    std::vector<double> hM, ht, hK;
    double *dLens = nullptr;  // [C][kLensStride], set by snowtri_ctx_set_distortion
    void *dBlenderTab = nullptr;          // 24 SmoothCoef of the last (fzr, dt) given to snowtri_blender_smooth
    std::vector<double> blender_key;      // that (fzr[72], dt)
    std::vector<int32_t> hpairs;
    double *dM = nullptr, *dt = nullptr, *dpairc = nullptr, *dP = nullptr;
    int32_t *dpairs = nullptr;
    StreamSet sets[kMaxSets];
    StreamSet *cur = &sets[0];        // the set the launch in progress writes to (set 0 between fused calls)
    StreamSet *last_set = &sets[0];   // the set of the last fused call's last segment (diagnostics read its counters)
    // set 0 under the names every other entry point uses
    unsigned long long *&d_counters = sets[0].d_counters;  // [0] singular pairs, [1] slow frames, [2..] spare
    Scratch &work = sets[0].work, &misc = sets[0].misc;
    Scratch &desc = sets[0].desc;             // cluster descriptors handed from k_frame_recompute / k_associate to k_cluster_fuse
    Scratch &sums = sets[0].sums;             // candidate sums of k_candidate_sums [frames][Kc] + the frames k_associate left behind
    Scratch in, out, aux;
    PinnedScratch pin_in, pin_out;            // host staging of the per-frame calls
    PinnedScratch pin_done;                   // the completion word of the small host calls (HostDone, snowtri_kernels.hpp)
    unsigned long long done_seq = 0;
    Scratch cand;                             // candidates of the last SNOWTRI_HOST snowtri_triangulate call (device-resident hand-over)
    int64_t cand_token = 0, cand_F = 0, cand_Kc = 0, cand_J = 0;   // token 0: none
    int overlap = 1;                  // snowtri_ctx_set_overlap: device calls rotate over this many sets (1 = the caller's stream)
    int64_t call_index = 0;
    hipEvent_t ev_fork = nullptr;     // the caller's stream at the time of a call / a split: the internal streams wait for it
    int split_segments = 2;           // one multi-person call is cut into >= this many segments on two sets (1: no split)
    bool split_forced = false;        // SNOWTRI_SPLIT_SEGMENTS given: also batches that would not fill the chip twice
    // Internal streams are only worth having on hardware queues of their own (k_probe_wait, snowtri_kernels.hpp):
    PinnedScratch pin_probe;
    int64_t probes = 0, probe_replaced = 0;   // snowtri_ctx_stream_probes
    int probe_verdict = -1;                   // of the stream kept last: 1 side by side, 0 none of the candidates was, -1 no probe yet
    hipStream_t split_beside[4];              // the caller streams sets[1].stream has been probed against (a ring of the last four)
    int n_split_beside = 0;
    bool split_knows(hipStream_t caller) const {
        for (int i = 0; i < std::min(n_split_beside, 4); ++i)
            if (split_beside[i] == caller) return true;
        return false;
    }
    // The probe SYNCHRONISES both streams (<= 300 us).  Policy: when a context CREATES its internal stream -- its
    // first split call -- the caller's stream is waited for once (documented in include/snowtri.h); a caller stream the
    // context meets LATER is probed only while it is idle (`only_if_idle`: a stream with work in flight is never waited for
    // -- the internal stream is kept, and the question is asked again at a later call); and a context probes at most
    // kMaxProbes times in its life: a caller that rotates many streams, or a box whose queues never come out side by side,
    // keeps what it has instead of paying a synchronisation per call.
    static constexpr int64_t kMaxProbes = 16;
    int probe_side_by_side(hipStream_t a, hipStream_t b, bool only_if_idle = false) {   // 1 / 0, -1: not probed
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;   // never put the probe into a graph under capture
        if (hipStreamIsCapturing(a, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;
        if (probes >= kMaxProbes) return -1;
        if (only_if_idle && hipStreamQuery(a) != hipSuccess) {
            (void)hipGetLastError();
            return -2;   // busy: ask again later
        }
        if (pin_probe.ensure(64)) return -1;
        volatile unsigned int *w = (volatile unsigned int *)pin_probe.p;
        w[0] = 0u;
        w[1] = 2u;
        ++probes;
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, (volatile unsigned int *)pin_probe.dev, 30000ull);   // <= 300 us
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, (volatile unsigned int *)pin_probe.dev);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        return w[1] == 1u ? 1 : 0;
    }
    // a new non-blocking stream that runs beside `caller` (when `with_caller`) and beside the internal streams of the other
    // sets; up to 6 candidates (each rejected one stays alive until the end, so that the next lands on another queue)
    int fresh_stream(hipStream_t *out, int k, bool with_caller, hipStream_t caller) {
        hipStream_t rejected[6];
        int nrej = 0, rc = 0;
        for (int attempt = 0; attempt < 6 && !rc; ++attempt) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
                rc = 1;
                break;
            }
            int ok = with_caller ? probe_side_by_side(caller, s) : 1;
            for (int j = 0; j < kMaxSets && ok == 1; ++j)
                if (j != k && sets[j].stream) ok = probe_side_by_side(sets[j].stream, s);
            if (ok < 0) {   // the probe itself failed (a caller stream that cannot be synchronised now, e.g. under capture):
                (void)hipGetLastError();   // keep the stream unprobed rather than fail the call
                probe_verdict = -1;
                *out = s;
                break;
            }
            probe_verdict = ok == 1 ? 1 : 0;
            if (ok == 1 || attempt == 5) {
                *out = s;
                break;
            }
            rejected[nrej++] = s;
            ++probe_replaced;
        }
        for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
        return rc;
    }
    int ensure_set(int k, bool with_caller = false, hipStream_t caller = nullptr) {   // counters, stream and event of set k
        StreamSet &S = sets[k];
        if (!S.d_counters) {
            if (hipMalloc(&S.d_counters, sizeof(unsigned long long) * kCounterWords) != hipSuccess) return 1;
            if (hipMemset(S.d_counters, 0, sizeof(unsigned long long) * kCounterWords) != hipSuccess) return 1;
        }
        if (S.stream && with_caller && !split_knows(caller)) {
            // the split of a call that arrives on another stream than the last one did: still side by side?
            const int ok = probe_side_by_side(caller, S.stream, true);
            if (ok == -2) with_caller = false;     // the caller's stream is busy: keep the stream, probe at a later call
            if (ok < 0) (void)hipGetLastError();   // unprobed: keep the stream
            if (ok == 0) {
                if (hipStreamSynchronize(S.stream) != hipSuccess) return 1;
                hipStream_t old = S.stream;
                S.stream = nullptr;
                const int rc = fresh_stream(&S.stream, k, true, caller);
                (void)hipStreamDestroy(old);
                ++probe_replaced;
                n_split_beside = 0;
                if (rc) return 1;
            } else if (ok != -2)
                probe_verdict = ok == 1 ? 1 : -1;
        } else if (!S.stream && fresh_stream(&S.stream, k, with_caller, caller))
            return 1;
        if (with_caller && !split_knows(caller)) split_beside[n_split_beside++ & 3] = caller;
        if (!S.done && hipEventCreateWithFlags(&S.done, hipEventDisableTiming) != hipSuccess) return 1;
        if (!ev_fork && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return 1;
        return 0;
    }
    // measurement
    bool timing = false;
    bool timing_attach = false;   // snowtri_set_timing(ctx, 2): single-kernel launches carry their ring events themselves
    bool ev_attached = false;     // the last launcher did (fused_dispatch then records no end event of its own)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    hipStream_t ev_stream = nullptr;
    std::vector<hipEvent_t> ev_ring;  // 2 events per recorded fused call (begin, end), kTimingRing calls deep
    int64_t ev_count = 0;             // fused calls recorded since the last snowtri_timing_collect
    int64_t last_slow_frames = 0;
    bool last_handover = false;  // the last fused call went through k_frame_recompute with the cluster hand-over armed
    // test knobs (environment, read at creation; snowtri_ctx_overrides names the ones that are set)
    int general_mode = 0;        // SNOWTRI_GENERAL_MODE: 0 auto, 1 force the spill kernel, 2 force the recompute kernel
    int lean_mode = 1;           // SNOWTRI_LEAN_MODE: 0 keeps float32-output batches (and DLT batches) on k_fused_single
    int lean_coop = 1;           // SNOWTRI_LEAN_COOP: 0 keeps small launches on k_fused_lean
    int sumless_mode = 1;        // SNOWTRI_SUMLESS_MODE: 0 keeps the candidate pass for single-detection batches on the streaming route
    int handover_mode = 1;       // SNOWTRI_HANDOVER_MODE: 1 streaming association (k_candidate_sums / k_associate / k_cluster_fuse), 2 hand-over
                                 // from inside k_frame_recompute, 0 the whole multi-person path inside k_frame_recompute
    int handover_seg_frames = 0; // SNOWTRI_HANDOVER_SEG_FRAMES: short segments of the streaming route
    int sums_threads = 0, sums_lds_kb = 0;   // SNOWTRI_SUMS_THREADS / _LDS_KB: workgroup shape of k_candidate_sums (0: automatic)
    int lean_tiles_per_wave = 0; // SNOWTRI_LEAN_TILES_PER_WAVE
    int debug = 0;               // SNOWTRI_DEBUG: launch shapes on stderr
    std::string overrides;
    static constexpr int assoc_wg_per_cu = 4 * kAssocWaves, lean_wg_per_cu = 2, lean_scratch_mb = 256;   // (settled by measurement: EXPERIMENTS.md)
    struct OccCache {
        size_t lds = 0;
        int per_cu = -1;
    } recompute_occ[8];   // resident workgroups per CU of the k_frame_recompute instantiations (queried once per LDS size)
    int raise_lds(const void *kern, int lds) { return raise_dynamic_lds(device, kern, lds); }   // once per (device, kernel, size), process-wide
    bool last_stream = false;        // the last fused call went through the streaming association (snowtri_last_stream_counts)
    // names of the kernels the last fused call launched (snowtri_last_kernel_names): a pointer to a string that lives as
    // long as the library (one per template instantiation) or to `names_buf`, rebuilt only when the route changes
    const char *last_kernels = "";
    std::string names_buf;
    long long names_key = -1;
    Rig rig() const { return Rig{dM, dt, dpairs, dpairc, dP, C, npairs}; }
};

extern "C" {

int snowtri_version(void) { return SNOWTRI_VERSION; }

const char *snowtri_status_string(int s) {
    switch (s) {
        case SNOWTRI_OK: return "ok";
        case SNOWTRI_ERR_BAD_ARG: return "bad argument";
        case SNOWTRI_ERR_BAD_INDEX: return "center_point_index / keypoint_num out of range";
        case SNOWTRI_ERR_HIP: return "HIP runtime error";
        case SNOWTRI_ERR_SINGULAR: return "singular ray pair (parallel rays)";
        case SNOWTRI_ERR_OVERFLOW: return "more output persons than Pout_max";
        case SNOWTRI_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

const char *snowtri_last_error(void) { return g_last_error.c_str(); }

const char *snowtri_build_info(void) {
    // what this binary was compiled as: the production library reports no variant at all
    static const std::string info = [] {
        std::string v;
        auto add = [&](const char *name) { v += (v.empty() ? "" : ",") + std::string(name); };
        (void)add;
#ifdef SNOWTRI_DEBUG_BOUNDS
        add("SNOWTRI_DEBUG_BOUNDS");
#endif
#ifdef SNOWTRI_TEST_KNOBS
        add("SNOWTRI_TEST_KNOBS");
#endif
#ifdef SNOWTRI_DEV_MIN
        add("SNOWTRI_DEV_MIN");
#endif
#ifdef SNOWTRI_DEV_EXPERIMENTS
        add("SNOWTRI_DEV_EXPERIMENTS");
#endif
#ifdef SNOWTRI_LEAN_TRACE
        add("SNOWTRI_LEAN_TRACE");
#endif
#ifdef SNOWTRI_SUMS_TRACE
        add("SNOWTRI_SUMS_TRACE");
#endif
#ifdef SNOWTRI_ASSOC_TRACE
        add("SNOWTRI_ASSOC_TRACE");
#endif
        return "version=" + std::to_string(SNOWTRI_VERSION) + ";arch=gfx950;variants=" + v;
    }();
    return info.c_str();
}

const char *snowtri_ctx_overrides(const snowtri_ctx *ctx) { return ctx ? ctx->overrides.c_str() : ""; }

int snowtri_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int snowtri_ctx_create(int32_t C, const double *K, const double *R, const double *t, int device,
                       snowtri_ctx **out) {
    if (!out || C < 0 || (C > 0 && (!K || !R || !t))) return SNOWTRI_ERR_BAD_ARG;
    *out = nullptr;
    if (snowtri_device_count() <= 0) return SNOWTRI_ERR_NO_DEVICE;
    snowtri_ctx *ctx = new (std::nothrow) snowtri_ctx();
    if (!ctx) return SNOWTRI_ERR_BAD_ARG;
    ctx->device = device;
    ctx->C = C;
    // Test knobs (include/snowtri.h lists them) exist ONLY in a -DSNOWTRI_TEST_KNOBS build (libsnowtri_dbg.so, which the tests that
    // force a route load): the production library reads no environment.  There they are read HERE, once -- nothing on the launch
    // path calls getenv -- and every one that is set is named by snowtri_ctx_overrides().
#ifdef SNOWTRI_TEST_KNOBS
    auto knob = [&](const char *name, int *field, int lo) {
        if (const char *e = getenv(name)) {
            *field = std::max(lo, atoi(e));
            ctx->overrides += (ctx->overrides.empty() ? "" : ",") + std::string(name) + "=" + std::to_string(*field);
        }
    };
    knob("SNOWTRI_GENERAL_MODE", &ctx->general_mode, 0);
    knob("SNOWTRI_LEAN_MODE", &ctx->lean_mode, 0);
    knob("SNOWTRI_LEAN_COOP", &ctx->lean_coop, 0);
    knob("SNOWTRI_SUMLESS_MODE", &ctx->sumless_mode, 0);
    knob("SNOWTRI_HANDOVER_MODE", &ctx->handover_mode, 0);
    knob("SNOWTRI_HANDOVER_SEG_FRAMES", &ctx->handover_seg_frames, 1);
    knob("SNOWTRI_SPLIT_SEGMENTS", &ctx->split_segments, 1);
    ctx->split_forced = getenv("SNOWTRI_SPLIT_SEGMENTS") != nullptr && ctx->split_segments >= 2;
    knob("SNOWTRI_SUMS_THREADS", &ctx->sums_threads, 0);
    knob("SNOWTRI_SUMS_LDS_KB", &ctx->sums_lds_kb, 0);
    knob("SNOWTRI_LEAN_TILES_PER_WAVE", &ctx->lean_tiles_per_wave, 1);
    knob("SNOWTRI_DEBUG", &ctx->debug, 0);
#endif
    ctx->hM.resize((size_t)C * 9);
    ctx->ht.assign(t, t + (size_t)C * 3);
    ctx->hK.assign(K, K + (size_t)C * 9);
    for (int c = 0; c < C; c++) {
        double Ki[9];
        if (inv3(K + 9 * c, Ki)) {  // np.linalg.inv(K) raises on a singular K (camera.py:242)
            delete ctx;
            return SNOWTRI_ERR_SINGULAR;
        }
        const double *Rc = R + 9 * c;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                ctx->hM[9 * c + 3 * i + j] = Rc[3 * i] * Ki[j] + Rc[3 * i + 1] * Ki[3 + j] + Rc[3 * i + 2] * Ki[6 + j];
    }
    for (int mc = 0; mc < C - 1; mc++)  // triangulation.py:56-58 loop order
        for (int sc = mc + 1; sc < C; sc++) {
            ctx->hpairs.push_back(mc);
            ctx->hpairs.push_back(sc);
        }
    ctx->npairs = (int32_t)(ctx->hpairs.size() / 2);
    std::vector<double> hpairc((size_t)ctx->npairs * 6);
    for (int q = 0; q < ctx->npairs; q++) {
        const double *tm = &ctx->ht[3 * ctx->hpairs[2 * q]], *ts = &ctx->ht[3 * ctx->hpairs[2 * q + 1]];
        for (int i = 0; i < 3; i++) {
            hpairc[6 * q + i] = ts[i] - tm[i];
            hpairc[6 * q + 3 + i] = tm[i] + ts[i];
        }
    }
    auto fail = [&](int rc) {
        snowtri_ctx_destroy(ctx);
        return rc;
    };
#define CTX_TRY(expr)                                         \
    do {                                                      \
        if ((expr) != hipSuccess) {                           \
            g_last_error = std::string(#expr) + " failed";    \
            return fail(SNOWTRI_ERR_HIP);                     \
        }                                                     \
    } while (0)
    DeviceGuard _device_guard(device);
    CTX_TRY(_device_guard.err);
    hipDeviceProp_t prop;
    CTX_TRY(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CTX_TRY(hipMalloc(&ctx->dM, sizeof(double) * std::max<size_t>(9, ctx->hM.size())));
    CTX_TRY(hipMalloc(&ctx->dt, sizeof(double) * std::max<size_t>(3, ctx->ht.size())));
    CTX_TRY(hipMalloc(&ctx->dpairs, sizeof(int32_t) * std::max<size_t>(2, ctx->hpairs.size())));
    CTX_TRY(hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * kCounterWords));
    CTX_TRY(hipMalloc(&ctx->dpairc, sizeof(double) * std::max<size_t>(6, hpairc.size())));
    {   // world->pixel matrices P_c = K_c [R_c^T | -R_c^T t_c] for the DLT method
        std::vector<double> hP((size_t)std::max(1, C) * 12, 0.0);
        for (int c = 0; c < C; c++) {
            const double *Kc = K + 9 * c, *Rc = R + 9 * c, *tc = t + 3 * c;
            double Rt[12];  // [R^T | -R^T t]
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) Rt[4 * i + j] = Rc[3 * j + i];
                Rt[4 * i + 3] = -(Rc[0 + i] * tc[0] + Rc[3 + i] * tc[1] + Rc[6 + i] * tc[2]);
            }
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++)
                    hP[12 * c + 4 * i + j] = Kc[3 * i] * Rt[j] + Kc[3 * i + 1] * Rt[4 + j] + Kc[3 * i + 2] * Rt[8 + j];
        }
        CTX_TRY(hipMalloc(&ctx->dP, sizeof(double) * hP.size()));
        CTX_TRY(hipMemcpy(ctx->dP, hP.data(), sizeof(double) * hP.size(), hipMemcpyHostToDevice));
    }
    if (ctx->npairs > 0)
        CTX_TRY(hipMemcpy(ctx->dpairc, hpairc.data(), sizeof(double) * hpairc.size(), hipMemcpyHostToDevice));
    if (C > 0) {
        CTX_TRY(hipMemcpy(ctx->dM, ctx->hM.data(), sizeof(double) * ctx->hM.size(), hipMemcpyHostToDevice));
        CTX_TRY(hipMemcpy(ctx->dt, ctx->ht.data(), sizeof(double) * ctx->ht.size(), hipMemcpyHostToDevice));
    }
    if (ctx->npairs > 0)
        CTX_TRY(hipMemcpy(ctx->dpairs, ctx->hpairs.data(), sizeof(int32_t) * ctx->hpairs.size(), hipMemcpyHostToDevice));
    CTX_TRY(hipMemset(ctx->d_counters, 0, sizeof(unsigned long long) * kCounterWords));
    for (auto &e : ctx->ev) CTX_TRY(hipEventCreate(&e));
#undef CTX_TRY
    *out = ctx;
    return SNOWTRI_OK;
}

int snowtri_ctx_destroy(snowtri_ctx *ctx) {
    if (!ctx) return SNOWTRI_OK;
    DeviceGuard _device_guard(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->dM) (void)hipFree(ctx->dM);
    if (ctx->dt) (void)hipFree(ctx->dt);
    if (ctx->dpairs) (void)hipFree(ctx->dpairs);
    if (ctx->dpairc) (void)hipFree(ctx->dpairc);
    if (ctx->dP) (void)hipFree(ctx->dP);
    if (ctx->dLens) (void)hipFree(ctx->dLens);
    if (ctx->dBlenderTab) (void)hipFree(ctx->dBlenderTab);
    for (auto &S : ctx->sets) {
        if (S.d_counters) (void)hipFree(S.d_counters);
        S.work.release();
        S.misc.release();
        S.desc.release();
        S.sums.release();
        if (S.stream) (void)hipStreamDestroy(S.stream);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    ctx->in.release();
    ctx->out.release();
    ctx->aux.release();
    ctx->cand.release();
    ctx->pin_done.release();
    ctx->pin_probe.release();
    ctx->pin_in.release();
    ctx->pin_out.release();
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_ring)
        if (e) (void)hipEventDestroy(e);
    delete ctx;
    return SNOWTRI_OK;
}

int snowtri_ctx_num_cameras(const snowtri_ctx *ctx) { return ctx ? ctx->C : -1; }

int snowtri_ctx_ray_matrices(const snowtri_ctx *ctx, double *M_out) {
    if (!ctx || !M_out) return SNOWTRI_ERR_BAD_ARG;
    std::memcpy(M_out, ctx->hM.data(), sizeof(double) * ctx->hM.size());
    return SNOWTRI_OK;
}

int snowtri_ctx_synchronize(snowtri_ctx *ctx) {
    if (!ctx) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    HIP_TRY(hipDeviceSynchronize());
    return SNOWTRI_OK;
}

int snowtri_ctx_set_overlap(snowtri_ctx *ctx, int n_streams) {
    if (!ctx || n_streams < 1 || n_streams > kMaxOverlap) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    for (auto &S : ctx->sets)   // calls still in flight on the internal streams finish under the old mode
        if (S.pending) {
            HIP_TRY(hipStreamSynchronize(S.stream));
            S.pending = false;
        }
    ctx->overlap = n_streams;
    ctx->call_index = 0;
    return SNOWTRI_OK;
}

int snowtri_ctx_set_split(snowtri_ctx *ctx, int segments) {
    if (!ctx || segments < 0 || segments > 64) return SNOWTRI_ERR_BAD_ARG;
    ctx->split_segments = segments == 0 ? 2 : segments;
    ctx->split_forced = segments >= 2;
    return SNOWTRI_OK;
}

int snowtri_ctx_join(snowtri_ctx *ctx, void *stream) {
    if (!ctx) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    for (auto &S : ctx->sets)
        if (S.pending) {
            HIP_TRY(hipEventRecord(S.done, S.stream));
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, S.done, 0));
            S.pending = false;
        }
    return SNOWTRI_OK;
}

int snowtri_set_timing(snowtri_ctx *ctx, int enabled) {
    if (!ctx) return SNOWTRI_ERR_BAD_ARG;
    ctx->timing = enabled != 0;
    ctx->timing_attach = enabled == 2;
    ctx->ev_valid = false;
    ctx->ev_count = 0;
    if (ctx->timing && ctx->ev_ring.empty()) {
        ENTER_DEVICE(ctx->device);
        ctx->ev_ring.resize(2 * kTimingRing, nullptr);
        for (auto &e : ctx->ev_ring) HIP_TRY(hipEventCreate(&e));
    }
    return SNOWTRI_OK;
}

int snowtri_timing_collect(snowtri_ctx *ctx, float *kernel_ms, int32_t cap) {
    if (!ctx || !kernel_ms || cap < 0) return -1;
    const int64_t n = std::min<int64_t>(std::min<int64_t>(ctx->ev_count, kTimingRing), cap);
    const int64_t first = ctx->ev_count - n;
    for (int64_t i = 0; i < n; i++) {
        const int64_t slot = (first + i) % kTimingRing;
        if (hipEventSynchronize(ctx->ev_ring[2 * slot + 1]) != hipSuccess) return -1;
        if (hipEventElapsedTime(&kernel_ms[i], ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1]) != hipSuccess) return -1;
    }
    ctx->ev_count = 0;
    return (int)n;
}

int snowtri_last_kernel_ms(snowtri_ctx *ctx, float kernel_ms[2]) {
    if (!ctx || !kernel_ms) return SNOWTRI_ERR_BAD_ARG;
    kernel_ms[0] = kernel_ms[1] = 0.f;
    if (!ctx->ev_valid) return SNOWTRI_ERR_BAD_ARG;
    HIP_TRY(hipEventSynchronize(ctx->ev[3]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, ctx->ev[2], ctx->ev[3]));
    kernel_ms[0] = a;
    kernel_ms[1] = b;
    return SNOWTRI_OK;
}

int64_t snowtri_last_slow_frames(snowtri_ctx *ctx) { return ctx ? ctx->last_slow_frames : -1; }

int64_t snowtri_last_handover_persons(snowtri_ctx *ctx, int64_t *n_other) {
    if (n_other) *n_other = -1;
    if (!ctx || !ctx->last_handover) return -1;
    DeviceGuard guard(ctx->device);
    unsigned long long n[kHandMembers + 1] = {0};   // [kHandComplete] complete-graph clusters, [kHandMembers] >> 32 clusters of any other shape
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(n, ctx->last_set->d_counters + kHandCountersAt, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (n_other) *n_other = (int64_t)hand_member_descs(n[kHandMembers]);
    return (int64_t)n[kHandComplete];
}

int snowtri_last_stream_counts(snowtri_ctx *ctx, int64_t counts[3]) {
    if (!ctx || !counts) return SNOWTRI_ERR_BAD_ARG;
    counts[0] = counts[1] = counts[2] = -1;
    if (!ctx->last_stream) return SNOWTRI_OK;
    ENTER_DEVICE(ctx->device);
    unsigned long long n[3] = {0, 0, 0};   // d_counters[6..8]: slow frames, frames with an exact candidate sum, slow frames of the second pass
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(n, ctx->last_set->d_counters + 6, sizeof(n), hipMemcpyDeviceToHost));
    counts[0] = (int64_t)n[0];
    counts[1] = (int64_t)n[1];
    counts[2] = (int64_t)n[2];
    return SNOWTRI_OK;
}

int snowtri_ctx_stream_probes(const snowtri_ctx *ctx, int64_t out[3]) {
    if (!ctx || !out) return SNOWTRI_ERR_BAD_ARG;
    out[0] = ctx->probes;
    out[1] = ctx->probe_replaced;
    out[2] = ctx->probe_verdict;
    return SNOWTRI_OK;
}

const char *snowtri_last_kernel_names(const snowtri_ctx *ctx) { return ctx ? ctx->last_kernels : ""; }

#ifdef SNOWTRI_DEBUG_BOUNDS
namespace {
__global__ void k_debug_selftest(int n) { SNOWTRI_DEV_CHECK((int)threadIdx.x < n, 99); }   // lanes n .. 63 violate it
}
#endif
#ifdef SNOWTRI_LEAN_TRACE
extern "C" int snowtri_debug_read_scratch(snowtri_ctx *ctx, void *dst, size_t bytes) {   // dev build only: the stamps of k_fused_lean_coop
    if (!ctx || bytes > ctx->work.cap) return SNOWTRI_ERR_BAD_ARG;
    return hipMemcpy(dst, ctx->work.p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? SNOWTRI_OK : SNOWTRI_ERR_HIP;
}
#endif
#ifdef SNOWTRI_SUMS_TRACE
extern "C" int snowtri_debug_read_sums_trace(void *dst) {   // dev build only: the stamps of k_candidate_sums
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(snowtri::g_sums_trace), sizeof(unsigned long long) * 4096 * 4 * 16) == hipSuccess ? 0 : 3;
}
#endif
#ifdef SNOWTRI_ASSOC_TRACE
extern "C" int snowtri_debug_read_assoc_trace(void *dst) {   // dev build only: the stamps of k_associate
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(snowtri::g_assoc_trace), sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : 3;
}
#endif
int snowtri_debug_selftest(snowtri_ctx *ctx) {
#ifdef SNOWTRI_DEBUG_BOUNDS
    if (!ctx) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipLaunchKernelGGL(k_debug_selftest, dim3(1), dim3(64), 0, nullptr, 61);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SNOWTRI_OK;
#else
    (void)ctx;
    return -1;
#endif
}

int64_t snowtri_debug_faults(snowtri_ctx *ctx, uint64_t *first) {
    if (first) *first = 0;
#ifdef SNOWTRI_DEBUG_BOUNDS
    if (!ctx) return -2;
    DeviceGuard guard(ctx->device);
    if (guard.err != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -2;
    unsigned long long v[2] = {0ull, 0ull}, zero[2] = {0ull, 0ull};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(snowtri::g_dev_fault), sizeof(v)) != hipSuccess) return -2;
    if (hipMemcpyToSymbol(HIP_SYMBOL(snowtri::g_dev_fault), zero, sizeof(zero)) != hipSuccess) return -2;
    if (first) *first = v[1];
    return (int64_t)v[0];
#else
    (void)ctx;
    return -1;   // this build has no device-side checks
#endif
}

int64_t snowtri_num_candidate_slots(int32_t C, int32_t Pmax) {
    if (C < 0 || Pmax < 0) return -1;
    return (int64_t)C * (C - 1) / 2 * Pmax * Pmax;
}

// Test hook: the fast-math helpers of the throughput kernels on caller-supplied values.
int snowtri_fastmath_probe(snowtri_ctx *ctx, int64_t n, const double *x, double *rcp_nr2_out,
                           double *rcp_nr1_out, double *rsq_nr1_out) {
    if (!ctx || n < 1 || !x || !rcp_nr2_out || !rcp_nr1_out || !rsq_nr1_out) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    const size_t b = sizeof(double) * n;
    int rc = ctx->in.ensure(b);
    if (rc) return rc;
    rc = ctx->out.ensure(3 * b);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ctx->in.p, x, b, hipMemcpyHostToDevice));
    double *o = (double *)ctx->out.p;
    hipLaunchKernelGGL(k_fastmath_probe, dim3(grid_for(n, kBlock, ctx->num_cus * 8)), dim3(kBlock), 0, 0, n,
                       (const double *)ctx->in.p, o, o + n, o + 2 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rcp_nr2_out, o, b, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rcp_nr1_out, o + n, b, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rsq_nr1_out, o + 2 * n, b, hipMemcpyDeviceToHost));
    return SNOWTRI_OK;
}

int snowtri_fastmath_probe_raw(snowtri_ctx *ctx, int64_t n, const double *x, double *rcp_raw_out, double *rsq_raw_out) {
    if (!ctx || n < 1 || !x || !rcp_raw_out || !rsq_raw_out) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    const size_t b = sizeof(double) * n;
    int rc = ctx->in.ensure(b);
    if (rc) return rc;
    rc = ctx->out.ensure(2 * b);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ctx->in.p, x, b, hipMemcpyHostToDevice));
    double *o = (double *)ctx->out.p;
    hipLaunchKernelGGL(k_fastmath_probe_raw, dim3(grid_for(n, kBlock, ctx->num_cus * 8)), dim3(kBlock), 0, 0, n,
                       (const double *)ctx->in.p, o, o + n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rcp_raw_out, o, b, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rsq_raw_out, o + n, b, hipMemcpyDeviceToHost));
    return SNOWTRI_OK;
}

int snowtri_calib_stream(snowtri_ctx *ctx, const void *src, int64_t read_bytes, void *dst, int64_t write_bytes, void *stream) {
    if (!ctx || read_bytes < 0 || write_bytes < 0 || (read_bytes > 0 && !src) || (write_bytes > 0 && !dst)) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (read_bytes >= 12) {
        int rc = ctx->misc.ensure(64);
        if (rc) return rc;
        hipLaunchKernelGGL(k_calib_read12, dim3(ctx->num_cus * 8), dim3(kBlock), 0, st, read_bytes / 12, (const Rec12 *)src,
                           (float *)ctx->misc.p);
    }
    if (write_bytes >= 16)
        hipLaunchKernelGGL(k_calib_write16, dim3(ctx->num_cus * 8), dim3(kBlock), 0, st, write_bytes / 16, (float4 *)dst);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

// ------------------------------------------------------------------------------------------ A1
int snowtri_rays_from_pixels(snowtri_ctx *ctx, int32_t cam, int64_t n, const double *uv, double *rays) {
    if (!ctx || n < 0 || (n > 0 && (!uv || !rays))) return SNOWTRI_ERR_BAD_ARG;
    if (cam < 0 || cam >= ctx->C) return SNOWTRI_ERR_BAD_INDEX;   // cameras[camera_index] raises IndexError (camera.py:236)
    if (n == 0) return SNOWTRI_OK;
    ENTER_DEVICE(ctx->device);
    int rc = ctx->in.ensure(sizeof(double) * 2 * n);
    if (rc) return rc;
    rc = ctx->out.ensure(sizeof(double) * 3 * n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ctx->in.p, uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rays, dim3(grid_for(n, kBlock, ctx->num_cus * 8)), dim3(kBlock), 0, 0, n,
                       ctx->dM + 9 * cam, (const double *)ctx->in.p, (double *)ctx->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rays, ctx->out.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    return SNOWTRI_OK;
}

// ------------------------------------------------------------------------------------------ A2
int snowtri_skew_ray_batch(snowtri_ctx *ctx, int64_t n, const double *hm, const double *hs,
                           const double *tm, const double *ts, double *dist, double *W,
                           int64_t *n_singular) {
    if (!ctx || n < 0 || (n > 0 && (!hm || !hs || !tm || !ts || !dist || !W))) return SNOWTRI_ERR_BAD_ARG;
    if (n_singular) *n_singular = 0;
    if (n == 0) return SNOWTRI_OK;
    ENTER_DEVICE(ctx->device);
    const size_t v = sizeof(double) * 3 * n;
    int rc = ctx->in.ensure(4 * v);
    if (rc) return rc;
    rc = ctx->out.ensure(v + sizeof(double) * n);
    if (rc) return rc;
    char *din = (char *)ctx->in.p;
    HIP_TRY(hipMemcpy(din, hm, v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(din + v, hs, v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(din + 2 * v, tm, v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(din + 3 * v, ts, v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(ctx->d_counters, 0, sizeof(unsigned long long)));
    double *dW = (double *)ctx->out.p, *ddist = dW + 3 * n;
    hipLaunchKernelGGL(k_skew, dim3(grid_for(n, kBlock, ctx->num_cus * 8)), dim3(kBlock), 0, 0, n,
                       (const double *)din, (const double *)(din + v), (const double *)(din + 2 * v),
                       (const double *)(din + 3 * v), ddist, dW, ctx->d_counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(W, dW, v, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dist, ddist, sizeof(double) * n, hipMemcpyDeviceToHost));
    unsigned long long ns = 0;
    HIP_TRY(hipMemcpy(&ns, ctx->d_counters, sizeof(ns), hipMemcpyDeviceToHost));
    if (n_singular) *n_singular = (int64_t)ns;
    return ns ? SNOWTRI_ERR_SINGULAR : SNOWTRI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- helpers
namespace {

int validate_params(const snowtri_params *p, int J, Params *out, bool need_condense) {
    if (!p) return SNOWTRI_ERR_BAD_ARG;
    Params q;
    q.kthr = p->keypoint_score_threshold;
    q.avg_thr = p->average_score_threshold;
    q.dthr = p->distance_threshold;
    q.ctol = p->condense_distance_tol;
    q.num_tol = p->condense_person_num_tol;
    q.score_tol = p->condense_score_tol;
    q.center = p->center_point_index;
    q.kn = p->keypoint_num;
    if (need_condense) {
        if (q.center < 0) q.center += J;  // Python negative indexing (triangulation.py:112)
        if (q.center < 0 || q.center >= J || q.kn < 0 || q.kn > J) return SNOWTRI_ERR_BAD_INDEX;
        if (q.kn > kCondenseMaxJointsPerThread * kBlock) return SNOWTRI_ERR_BAD_ARG;
    }
    q.dthr2 = (q.dthr < 0.0) ? -1.0 : q.dthr * q.dthr;
    {
        float kf = (float)q.kthr;  // round to nearest, then step up if that landed below kthr
        if ((double)kf < q.kthr) kf = std::nextafter(kf, std::numeric_limits<float>::infinity());
        q.kthr_f32 = kf;            // NaN stays NaN: every comparison false, as in NumPy
    }
    q.no_zero_fill = 0;             // (snowtri_triangulate_condense_ex sets it from its call flags)
    *out = q;
    return SNOWTRI_OK;
}

size_t dtype_size(int dt) { return dt == SNOWTRI_F32 ? 4 : 8; }

// The completion word of a small host call: arm it before the launch, spin on it afterwards (the kernels' outputs are already
// in mapped host memory then).  hipStreamSynchronize stays as the fall-back after 50 ms of spinning.
int host_done_arm(snowtri_ctx *ctx, HostDone *hd) {
    int rc = ctx->pin_done.ensure(64);
    if (rc) return rc;
    hd->count = (unsigned int *)(ctx->d_counters + kHostDoneCountAt);
    hd->flag = (unsigned long long *)ctx->pin_done.dev;
    hd->seq = ++ctx->done_seq;
    return SNOWTRI_OK;
}
static inline void cpu_relax() {   // spin-wait hint of the host CPU
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
int host_done_wait(snowtri_ctx *ctx, hipStream_t st, const HostDone &hd) {
    const volatile unsigned long long *w = (const volatile unsigned long long *)ctx->pin_done.p;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *w != hd.seq; spins++) {
        cpu_relax();
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            HIP_TRY(hipStreamSynchronize(st));   // (a failed launch never writes the word: the runtime reports it here)
            if (*w != hd.seq) {
                g_last_error = "a per-frame kernel finished without its completion word";
                return SNOWTRI_ERR_HIP;
            }
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SNOWTRI_OK;
}

template <typename TIn>
void launch_triangulate(snowtri_ctx *ctx, hipStream_t st, int64_t F, int Pmax, int J, int Kc,
                        const void *kpts, const int32_t *n_persons, const Params &prm, double *cxyz,
                        double *cks, double *cps, uint8_t *ckeep, unsigned long long *n_singular, char *mirror = nullptr,
                        HostDone done = HostDone{nullptr, nullptr, 0ull}, bool *one_launch = nullptr) {
    // mirror: host-mapped block [xyz | kscore | pscore | keep | (16-byte aligned) singular count] the kernels fill as well
    const int64_t total = F * (int64_t)Kc * J;
    double *mx = (double *)mirror, *mk = mirror ? mx + 3 * total : nullptr, *mp = mirror ? mk + total : nullptr;
    uint8_t *mkeep = mirror ? (uint8_t *)(mp + F * (int64_t)Kc) : nullptr;
    const size_t ns_off = ((sizeof(double) * ((size_t)total * 4 + (size_t)F * Kc) + (size_t)F * Kc) + 15) & ~(size_t)15;
    if (mirror && F * (int64_t)Kc <= 4096 && J <= 4096) {   // the per-frame call: one launch, a workgroup per candidate slot
        hipLaunchKernelGGL((k_triangulate_slots<TIn>), dim3((unsigned)(F * Kc)), dim3(kBlock), sizeof(double) * J, st, F, Pmax, J, Kc, ctx->rig(),
                           (const TIn *)kpts, n_persons, prm, cxyz, cks, cps, ckeep, n_singular, mx, mk, mp, mkeep,
                           (unsigned long long *)(mirror + ns_off), (unsigned int *)(n_singular + 1), done);
        if (one_launch) *one_launch = true;
        return;
    }
    hipLaunchKernelGGL((k_triangulate<TIn>), dim3(grid_for(total, kBlock, ctx->num_cus * 16)), dim3(kBlock), 0,
                       st, F, Pmax, J, Kc, ctx->rig(), (const TIn *)kpts, n_persons, prm, cxyz, cks,
                       n_singular, mx, mk);
    hipLaunchKernelGGL(k_cand_mean, dim3(grid_for(F * (int64_t)Kc, kBlock / 64, ctx->num_cus * 16)),
                       dim3(kBlock), 0, st, F, Pmax, J, Kc, ctx->rig(), n_persons, prm, (const double *)cks,
                       cps, ckeep, mp, mkeep, n_singular, mirror ? (unsigned long long *)(mirror + ns_off) : nullptr);
}

constexpr int kCondenseMaxN = 9000;  // condense_lds_bytes(N) <= 160 KiB

template <typename Writer>
int launch_condense(snowtri_ctx *ctx, hipStream_t st, int64_t nframes, int N, int J, const double *cxyz,
                    const double *cks, const uint8_t *ckeep, const Params &prm, int Pout, Writer wr,
                    int32_t *out_count, uint32_t *out_flags, HostDone done = HostDone{nullptr, nullptr, 0ull}) {
    if (N > kCondenseMaxN) return SNOWTRI_ERR_BAD_ARG;
    const size_t lds = condense_lds_bytes(N);
    auto kern = k_condense<Writer>;
    if (lds > 48 * 1024)
        {
            if (ctx->raise_lds((const void *)kern, (int)lds)) {
                g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return SNOWTRI_ERR_HIP;
            }
        }
    hipLaunchKernelGGL(kern, dim3(grid_for(nframes, 1, ctx->num_cus * 8)), dim3(kBlock), lds, st, nframes, N, J,
                       cxyz, cks, ckeep, prm, Pout, wr, out_count, out_flags, done);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

}  // namespace

extern "C" {

// --------------------------------------------------------------------------------------- A1+A3
int snowtri_triangulate(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J, const void *kpts,
                        int in_dtype, const int32_t *n_persons, const snowtri_params *params,
                        double *cand_xyz, double *cand_kscore, double *cand_pscore, uint8_t *cand_keep,
                        int memspace, void *stream) {
    if (!ctx || ctx->C < 1 || F < 0 || Pmax < 1 || J < 1 || !params) return SNOWTRI_ERR_BAD_ARG;
    if (in_dtype != SNOWTRI_F32 && in_dtype != SNOWTRI_F64) return SNOWTRI_ERR_BAD_ARG;
    if (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE) return SNOWTRI_ERR_BAD_ARG;
    const int64_t Kc = snowtri_num_candidate_slots(ctx->C, Pmax);
    if (F == 0 || Kc == 0) return SNOWTRI_OK;
    if (!kpts || !cand_xyz || !cand_kscore || !cand_pscore || !cand_keep) return SNOWTRI_ERR_BAD_ARG;
    Params prm;
    int rc = validate_params(params, J, &prm, false);
    if (rc) return rc;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t in_bytes = (size_t)F * ctx->C * Pmax * J * 3 * dtype_size(in_dtype);
    const size_t np_bytes = n_persons ? sizeof(int32_t) * F * ctx->C : 0;
    const size_t nx = (size_t)F * Kc * J;
    const void *d_kpts = kpts;
    const int32_t *d_np = n_persons;
    double *d_xyz = cand_xyz, *d_ks = cand_kscore, *d_ps = cand_pscore;
    uint8_t *d_keep = cand_keep;
    unsigned long long *d_nsing = ctx->d_counters;
    const size_t in_np_off = (in_bytes + 15) & ~(size_t)15;
    const size_t out_bytes = sizeof(double) * (nx * 4 + (size_t)F * Kc) + (size_t)F * Kc;
    const size_t out_ns_off = (out_bytes + 15) & ~(size_t)15;          // the singular-pair counter rides behind the outputs
    const bool host = memspace == SNOWTRI_HOST;
    const bool pinned = host && in_np_off + np_bytes <= kPinnedMaxBytes && out_ns_off + 8 <= kPinnedMaxBytes;
    // the per-frame call: kernels read the staged inputs and mirror their outputs in mapped page-locked memory, no copy engine
    const bool zero_copy = pinned && in_np_off + np_bytes + out_ns_off + 8 <= kZeroCopyMaxBytes;
    char *mirror = nullptr;
    if (host) {
        rc = ctx->in.ensure(in_np_off + np_bytes + 16);
        if (rc) return rc;
        rc = ctx->cand.ensure(out_ns_off + 64);   // (a scratch of its own: the candidates stay resident for snowtri_condense_resident)
        if (rc) return rc;
        ctx->cand_token = 0;
        d_kpts = ctx->in.p;
        if (n_persons) d_np = (int32_t *)((char *)ctx->in.p + in_np_off);
        d_xyz = (double *)ctx->cand.p;
        d_ks = d_xyz + nx * 3;
        d_ps = d_ks + nx;
        d_keep = (uint8_t *)(d_ps + (size_t)F * Kc);
        d_nsing = (unsigned long long *)((char *)ctx->cand.p + out_ns_off);
        if (pinned) {   // one staged upload
            rc = ctx->pin_in.ensure(in_np_off + np_bytes);
            if (rc) return rc;
            rc = ctx->pin_out.ensure(out_ns_off + 8);
            if (rc) return rc;
            std::memcpy(ctx->pin_in.p, kpts, in_bytes);
            if (n_persons) std::memcpy((char *)ctx->pin_in.p + in_np_off, n_persons, np_bytes);
            if (zero_copy) {
                d_kpts = ctx->pin_in.dev;
                if (n_persons) d_np = (int32_t *)((char *)ctx->pin_in.dev + in_np_off);
                mirror = (char *)ctx->pin_out.dev;
                d_nsing = ctx->d_counters + kTriangulateSingularAt;   // (left at 0 by every call: k_cand_mean hands it over)
            } else {
                HIP_TRY(hipMemcpyAsync(ctx->in.p, ctx->pin_in.p, in_np_off + np_bytes, hipMemcpyHostToDevice, st));
            }
        } else {
            HIP_TRY(hipMemcpyAsync(ctx->in.p, kpts, in_bytes, hipMemcpyHostToDevice, st));
            if (n_persons) HIP_TRY(hipMemcpyAsync((void *)d_np, n_persons, np_bytes, hipMemcpyHostToDevice, st));
        }
        if (!zero_copy) HIP_TRY(hipMemsetAsync(d_xyz, 0, out_ns_off + 8, st));  // invalid slots read back as zeros; counter = 0
    } else {
        HIP_TRY(hipMemsetAsync(d_nsing, 0, sizeof(unsigned long long), st));
    }
    HostDone done{nullptr, nullptr, 0ull};
    bool polled = false;
    if (zero_copy) {
        rc = host_done_arm(ctx, &done);
        if (rc) return rc;
    }
    if (in_dtype == SNOWTRI_F32)
        launch_triangulate<float>(ctx, st, F, Pmax, J, (int)Kc, d_kpts, d_np, prm, d_xyz, d_ks, d_ps, d_keep, d_nsing, mirror, done, &polled);
    else
        launch_triangulate<double>(ctx, st, F, Pmax, J, (int)Kc, d_kpts, d_np, prm, d_xyz, d_ks, d_ps, d_keep, d_nsing, mirror, done, &polled);
    HIP_TRY(hipGetLastError());
    if (host) {
        unsigned long long ns = 0;
        if (pinned) {   // one staged download (zero_copy: the kernels have written the staging buffer themselves)
            if (!zero_copy) HIP_TRY(hipMemcpyAsync(ctx->pin_out.p, ctx->cand.p, out_ns_off + 8, hipMemcpyDeviceToHost, st));
            if (polled) {   // the one-launch kernel signals a mapped word: no sleep on the runtime's interrupt
                rc = host_done_wait(ctx, st, done);
                if (rc) return rc;
            } else {
                HIP_TRY(hipStreamSynchronize(st));
            }
            const char *o = (const char *)ctx->pin_out.p;
            std::memcpy(cand_xyz, o, sizeof(double) * nx * 3);
            std::memcpy(cand_kscore, o + sizeof(double) * nx * 3, sizeof(double) * nx);
            std::memcpy(cand_pscore, o + sizeof(double) * nx * 4, sizeof(double) * F * Kc);
            std::memcpy(cand_keep, o + sizeof(double) * (nx * 4 + (size_t)F * Kc), (size_t)F * Kc);
            std::memcpy(&ns, o + out_ns_off, sizeof(ns));
        } else {
            HIP_TRY(hipMemcpyAsync(cand_xyz, d_xyz, sizeof(double) * nx * 3, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(cand_kscore, d_ks, sizeof(double) * nx, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(cand_pscore, d_ps, sizeof(double) * F * Kc, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(cand_keep, d_keep, (size_t)F * Kc, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&ns, d_nsing, sizeof(ns), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        // the candidates stay on the device under a fresh token (snowtri_condense_resident)
        static std::atomic<int64_t> next_token{1};
        ctx->cand_token = next_token.fetch_add(1);
        ctx->cand_F = F;
        ctx->cand_Kc = Kc;
        ctx->cand_J = J;
        if (ns) return SNOWTRI_ERR_SINGULAR;
    }
    return SNOWTRI_OK;
}

int64_t snowtri_candidates_token(const snowtri_ctx *ctx) { return ctx ? ctx->cand_token : 0; }

// ------------------------------------------------------------------------------------------ A4
int snowtri_condense(snowtri_ctx *ctx, int64_t F, int32_t N, int32_t J, const double *cand_xyz,
                     const double *cand_kscore, const uint8_t *cand_keep, const snowtri_params *params,
                     int32_t Pout_max, double *out_xyz, double *out_kscore, double *out_pscore,
                     int32_t *out_count, uint32_t *out_flags, int memspace, void *stream) {
    if (!ctx || F < 0 || N < 0 || J < 1 || Pout_max < 1 || !params) return SNOWTRI_ERR_BAD_ARG;
    if (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE) return SNOWTRI_ERR_BAD_ARG;
    if (F == 0) return SNOWTRI_OK;
    if (!out_xyz || !out_kscore || !out_pscore || !out_count) return SNOWTRI_ERR_BAD_ARG;
    if (N > 0 && (!cand_xyz || !cand_kscore)) return SNOWTRI_ERR_BAD_ARG;
    Params prm;
    int rc = validate_params(params, J, &prm, true);
    if (rc) return rc;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int kn = prm.kn;
    const size_t nx = (size_t)F * N * J;
    const size_t no = (size_t)F * Pout_max * kn;
    const double *d_xyz = cand_xyz, *d_ks = cand_kscore;
    const uint8_t *d_keep = cand_keep;
    double *o_xyz = out_xyz, *o_ks = out_kscore, *o_ps = out_pscore;
    int32_t *o_cnt = out_count;
    uint32_t *o_fl = out_flags;
    const bool host = memspace == SNOWTRI_HOST;
    const size_t in_total = sizeof(double) * nx * 4 + (size_t)F * N;
    const size_t out_total = sizeof(double) * (no * 4 + (size_t)F * Pout_max) + 8 * (size_t)F;
    const bool pinned = host && in_total <= kPinnedMaxBytes && out_total <= kPinnedMaxBytes;
    if (host) {
        rc = ctx->in.ensure(in_total + 64);
        if (rc) return rc;
        rc = ctx->out.ensure(out_total + 64);
        if (rc) return rc;
        double *p = (double *)ctx->in.p;
        d_xyz = p;
        d_ks = p + nx * 3;
        if (cand_keep && N) d_keep = (uint8_t *)(p + nx * 4);
        if (pinned) {   // one staged upload
            rc = ctx->pin_in.ensure(in_total + 16);
            if (rc) return rc;
            rc = ctx->pin_out.ensure(out_total + 16);
            if (rc) return rc;
            char *h = (char *)ctx->pin_in.p;
            if (nx) {
                std::memcpy(h, cand_xyz, sizeof(double) * nx * 3);
                std::memcpy(h + sizeof(double) * nx * 3, cand_kscore, sizeof(double) * nx);
            }
            if (cand_keep && N) std::memcpy(h + sizeof(double) * nx * 4, cand_keep, (size_t)F * N);
            const size_t up = sizeof(double) * nx * 4 + ((cand_keep && N) ? (size_t)F * N : 0);
            if (up) HIP_TRY(hipMemcpyAsync(p, h, up, hipMemcpyHostToDevice, st));
        } else {
            if (nx) {
                HIP_TRY(hipMemcpyAsync(p, cand_xyz, sizeof(double) * nx * 3, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(p + nx * 3, cand_kscore, sizeof(double) * nx, hipMemcpyHostToDevice, st));
            }
            if (cand_keep && N) HIP_TRY(hipMemcpyAsync((void *)d_keep, cand_keep, (size_t)F * N, hipMemcpyHostToDevice, st));
        }
        o_xyz = (double *)ctx->out.p;
        o_ks = o_xyz + no * 3;
        o_ps = o_ks + no;
        o_cnt = (int32_t *)(o_ps + (size_t)F * Pout_max);
        o_fl = (uint32_t *)(o_cnt + F);
    }
    if (o_fl) HIP_TRY(hipMemsetAsync(o_fl, 0, sizeof(uint32_t) * F, st));
    rc = launch_condense(ctx, st, F, N, J, d_xyz, d_ks, d_keep, prm, Pout_max, SplitWriter{o_xyz, o_ks, o_ps},
                         o_cnt, o_fl);
    if (rc) return rc;
    if (host) {
        if (pinned) {   // one staged download
            HIP_TRY(hipMemcpyAsync(ctx->pin_out.p, ctx->out.p, out_total, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const char *o = (const char *)ctx->pin_out.p;
            if (no) {
                std::memcpy(out_xyz, o, sizeof(double) * no * 3);
                std::memcpy(out_kscore, o + sizeof(double) * no * 3, sizeof(double) * no);
            }
            std::memcpy(out_pscore, o + sizeof(double) * no * 4, sizeof(double) * F * Pout_max);
            std::memcpy(out_count, o + sizeof(double) * (no * 4 + (size_t)F * Pout_max), sizeof(int32_t) * F);
            if (out_flags) std::memcpy(out_flags, o + sizeof(double) * (no * 4 + (size_t)F * Pout_max) + sizeof(int32_t) * F, sizeof(uint32_t) * F);
        } else {
            if (no) {
                HIP_TRY(hipMemcpyAsync(out_xyz, o_xyz, sizeof(double) * no * 3, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(out_kscore, o_ks, sizeof(double) * no, hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipMemcpyAsync(out_pscore, o_ps, sizeof(double) * F * Pout_max, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_count, o_cnt, sizeof(int32_t) * F, hipMemcpyDeviceToHost, st));
            if (out_flags) HIP_TRY(hipMemcpyAsync(out_flags, o_fl, sizeof(uint32_t) * F, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        for (int64_t f = 0; f < F; f++)
            if (out_count[f] > Pout_max) return SNOWTRI_ERR_OVERFLOW;
    }
    return SNOWTRI_OK;
}

// A4 on the candidates the last SNOWTRI_HOST snowtri_triangulate call left on the device: what Human_Triangulation_Condense
// does when it is handed the unmodified result of Human_Triangulation (main.py:62-71) -- no second trip over PCIe for the
// candidates (25 KB per frame at 4 x 1 x 133: the upload was a quarter of the per-frame latency).
int snowtri_condense_resident(snowtri_ctx *ctx, int64_t token, const snowtri_params *params, int32_t Pout_max, double *out_xyz,
                              double *out_kscore, double *out_pscore, int32_t *out_count, uint32_t *out_flags) {
    if (!ctx || !params || Pout_max < 1 || !out_xyz || !out_kscore || !out_pscore || !out_count) return SNOWTRI_ERR_BAD_ARG;
    if (token == 0 || token != ctx->cand_token) return SNOWTRI_ERR_BAD_ARG;   // (another triangulate call replaced them)
    const int64_t F = ctx->cand_F, N = ctx->cand_Kc;
    const int J = (int)ctx->cand_J;
    Params prm;
    int rc = validate_params(params, J, &prm, true);
    if (rc) return rc;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = nullptr;
    const int kn = prm.kn;
    const size_t nx = (size_t)F * N * J, no = (size_t)F * Pout_max * kn;
    const double *d_xyz = (const double *)ctx->cand.p, *d_ks = d_xyz + nx * 3;
    const uint8_t *d_keep = (const uint8_t *)(d_ks + nx + (size_t)F * N);
    const size_t out_total = sizeof(double) * (no * 4 + (size_t)F * Pout_max) + 4 * (size_t)F;   // xyz | kscore | pscore | count
    const bool zero_copy = out_total <= kZeroCopyMaxBytes;   // the kernel writes the page-locked staging buffer itself
    char *base;
    if (zero_copy) {
        rc = ctx->pin_out.ensure(out_total + 16);
        if (rc) return rc;
        base = (char *)ctx->pin_out.dev;
    } else {
        rc = ctx->out.ensure(out_total + 64);
        if (rc) return rc;
        base = (char *)ctx->out.p;
    }
    double *o_xyz = (double *)base, *o_ks = o_xyz + no * 3, *o_ps = o_ks + no;
    int32_t *o_cnt = (int32_t *)(o_ps + (size_t)F * Pout_max);
    // (the only flag of this entry is the overflow bit: derived from the counts below, no atomics on mapped memory)
    HostDone done{nullptr, nullptr, 0ull};
    if (zero_copy) {
        rc = host_done_arm(ctx, &done);
        if (rc) return rc;
    }
    rc = launch_condense(ctx, st, F, (int)N, J, d_xyz, d_ks, d_keep, prm, Pout_max, SplitWriter{o_xyz, o_ks, o_ps}, o_cnt, (uint32_t *)nullptr, done);
    if (rc) return rc;
    if (zero_copy) {
        rc = host_done_wait(ctx, st, done);
        if (rc) return rc;
    } else {
        HIP_TRY(hipStreamSynchronize(st));
    }
    auto fetch = [&](void *dst, size_t off, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        if (zero_copy) {
            std::memcpy(dst, (const char *)ctx->pin_out.p + off, bytes);
            return hipSuccess;
        }
        return hipMemcpy(dst, base + off, bytes, hipMemcpyDeviceToHost);
    };
    HIP_TRY(fetch(out_xyz, 0, sizeof(double) * no * 3));
    HIP_TRY(fetch(out_kscore, sizeof(double) * no * 3, sizeof(double) * no));
    HIP_TRY(fetch(out_pscore, sizeof(double) * no * 4, sizeof(double) * F * Pout_max));
    HIP_TRY(fetch(out_count, sizeof(double) * (no * 4 + (size_t)F * Pout_max), sizeof(int32_t) * F));
    int status = SNOWTRI_OK;
    for (int64_t f = 0; f < F; f++) {
        const bool over = out_count[f] > Pout_max;
        if (out_flags) out_flags[f] = over ? SNOWTRI_FLAG_OVERFLOW : 0u;
        if (over) status = SNOWTRI_ERR_OVERFLOW;
    }
    return status;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- N1
namespace {

constexpr int kSmoothChunk = 256;

SmoothCoef smooth_coef(double f, double z, double r, double dt) {
    const double pi = 3.141592653589793;
    const double k1 = z / (pi * f), k2 = 1.0 / ((2 * pi * f) * (2 * pi * f)), k3 = r * z / (2 * pi * f);
    SmoothCoef k;
    k.a00 = 1.0;
    k.a01 = dt;
    k.a10 = -dt / k2;
    k.a11 = 1.0 - dt * dt / k2 - dt * k1 / k2;
    k.cx = dt / k2;
    k.cxd = k3 / k2;  // (T/k2) * k3 * (x - xp)/T
    double p[4] = {1, 0, 0, 1};  // A^L by repeated multiplication (fp64)
    for (int i = 0; i < kSmoothChunk; i++) {
        const double q0 = k.a00 * p[0] + k.a01 * p[2], q1 = k.a00 * p[1] + k.a01 * p[3];
        const double q2 = k.a10 * p[0] + k.a11 * p[2], q3 = k.a10 * p[1] + k.a11 * p[3];
        p[0] = q0; p[1] = q1; p[2] = q2; p[3] = q3;
    }
    k.p00 = p[0]; k.p01 = p[1]; k.p10 = p[2]; k.p11 = p[3];
    double q[4] = {1, 0, 0, 1};  // A^32: the frames one wave of the one-pass scan holds (k_smooth_scan)
    for (int i = 0; i < kScanL; i++) {
        const double q0 = k.a00 * q[0] + k.a01 * q[2], q1 = k.a00 * q[1] + k.a01 * q[3];
        const double q2 = k.a10 * q[0] + k.a11 * q[2], q3 = k.a10 * q[1] + k.a11 * q[3];
        q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
    }
    k.r00 = q[0]; k.r01 = q[1]; k.r10 = q[2]; k.r11 = q[3];
    return k;
}

dim3 smooth_grid(int64_t n, int64_t nchunks) {
    return dim3((unsigned)((n + kSmoothBlock - 1) / kSmoothBlock), (unsigned)std::max<int64_t>(1, nchunks));
}

// Zero-state response of frames [tb, T) into dy (frame 0 passes through when first); optional end state.
// This is the first half of the two-call protocol of a frame-sharded track.
template <typename KS>
int smooth_local_dev(snowtri_ctx *ctx, hipStream_t st, int64_t T, int64_t n, const double *dx, double *dy,
                     bool first, const KS &k, double *d_end) {
    const int tb = first ? 1 : 0;
    const int64_t m = T - tb, nchunks = (m + kSmoothChunk - 1) / kSmoothChunk;
    if (nchunks > 65535) return SNOWTRI_ERR_BAD_ARG;
    if (m <= 0) {  // a one-frame first shard: nothing is filtered
        HIP_TRY(hipMemcpyAsync(dy, dx, sizeof(double) * (size_t)T * n, hipMemcpyDeviceToDevice, st));
        if (d_end) HIP_TRY(hipMemsetAsync(d_end, 0, sizeof(double) * 2 * n, st));
        return SNOWTRI_OK;
    }
    int rc = ctx->work.ensure(sizeof(double) * 4 * (size_t)nchunks * n + 64);
    if (rc) return rc;
    double *E = (double *)ctx->work.p, *S = E + 2 * (size_t)nchunks * n;
    const dim3 g2 = smooth_grid(n, nchunks), g1 = smooth_grid(n, 1);
    hipLaunchKernelGGL((k_smooth_local<KS, NoHold>), g2, dim3(kSmoothBlock), 0, st, T, n, kSmoothChunk, tb, k, NoHold{}, dx,
                       (const double *)nullptr, dy, E);
    hipLaunchKernelGGL(k_smooth_carry<KS>, g1, dim3(kSmoothBlock), 0, st, n, nchunks, k, (const double *)nullptr,
                       (const double *)E, S);
    hipLaunchKernelGGL(k_smooth_fix<KS>, g2, dim3(kSmoothBlock), 0, st, T, n, kSmoothChunk, tb, k, (const double *)S, dy,
                       (double *)nullptr);
    if (d_end)
        hipLaunchKernelGGL(k_smooth_shard_end<KS>, g1, dim3(kSmoothBlock), 0, st, T, n, kSmoothChunk, tb, nchunks, k,
                           (const double *)S, (const double *)E, d_end);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

// dy += homogeneous response of the state `d_start` entering frame tb (second half of the sharded protocol).
template <typename KS>
int smooth_fix_dev(snowtri_ctx *ctx, hipStream_t st, int64_t T, int64_t n, double *dy, bool first,
                   const double *d_start, const KS &k) {
    const int tb = first ? 1 : 0;
    const int64_t m = T - tb, nchunks = (m + kSmoothChunk - 1) / kSmoothChunk;
    if (m <= 0) return SNOWTRI_OK;
    if (nchunks > 65535) return SNOWTRI_ERR_BAD_ARG;
    int rc = ctx->work.ensure(sizeof(double) * 2 * (size_t)nchunks * n + 64);
    if (rc) return rc;
    double *S = (double *)ctx->work.p;
    hipLaunchKernelGGL(k_smooth_carry<KS>, smooth_grid(n, 1), dim3(kSmoothBlock), 0, st, n, nchunks, k, d_start,
                       (const double *)nullptr, S);
    hipLaunchKernelGGL(k_smooth_fix<KS>, smooth_grid(n, nchunks), dim3(kSmoothBlock), 0, st, T, n, kSmoothChunk, tb, k,
                       (const double *)S, dy, (double *)nullptr);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

// A whole (unsharded) track in ONE pass over HBM: the chained scan with decoupled look-back of snowtri_smooth.hpp
// (k_smooth_scan) -- x read once, y written once.  d_seed_row: [n] inputs of frame 0 as the filters see them (x[0] itself for
// N1; the held x_eff[0] for N2).  skip4: every fourth lane (the score of a joint record) is copied instead of filtered.
// (Rounds 1-5 ran three passes -- chunk-end states, the carry over the chunks, the recurrence again: 24 bytes moved per 16
// algorithmic; the sharded protocol below still does, its entering state is not known before the exchange.)
// tb / d_start / d_end: the frame-shard forms (k_smooth_scan): tb = 0 filters every frame, d_start [2n] is the state entering the
// first filtered frame, d_end [2n] receives the state behind the last frame, dy == nullptr stores no track (the "reduce" pass).
template <typename KS, typename HS>
int smooth_whole_dev(snowtri_ctx *ctx, hipStream_t st, int64_t T, int64_t n, const double *dx, double *dy,
                     const double *d_seed_row, const KS &k, const HS &hs, bool skip4 = false, int tb = 1, const double *d_start = nullptr,
                     double *d_end = nullptr) {
    const int64_t m = T - tb, nsuper = (m + kScanSuper - 1) / kScanSuper, ncols = (n + 63) / 64;
    if (nsuper > 65535) return SNOWTRI_ERR_BAD_ARG;   // (the documented limit of the track length: 1 + 256 * 65535 frames)
    if (m <= 0) {   // a one-frame track / first shard: nothing is filtered; the state behind it is the entering one
        if (dy) HIP_TRY(hipMemcpyAsync(dy, dx, sizeof(double) * (size_t)T * n, hipMemcpyDeviceToDevice, st));
        if (d_end) {
            if (d_start)
                HIP_TRY(hipMemcpyAsync(d_end, d_start, sizeof(double) * 2 * n, hipMemcpyDeviceToDevice, st));
            else
                HIP_TRY(hipMemsetAsync(d_end, 0, sizeof(double) * 2 * n, st));
        }
        return SNOWTRI_OK;
    }
    if (nsuper * ncols >= ((int64_t)1 << 31) || n > ((int64_t)1 << 22)) return SNOWTRI_ERR_BAD_ARG;   // (lanes: the 32-bit buffer offsets of k_smooth_scan)
    const size_t nwg = (size_t)nsuper * ncols;
    const size_t off_flags = 256, off_agg = (off_flags + 4 * nwg + 255) & ~(size_t)255, off_incl = off_agg + sizeof(double) * 128 * nwg;
    int rc = ctx->work.ensure(off_incl + sizeof(double) * 128 * nwg);
    if (rc) return rc;
    char *base = (char *)ctx->work.p;
    HIP_TRY(hipMemsetAsync(base, 0, off_agg, st));   // the ticket and the flags
    unsigned int *ticket = (unsigned int *)base, *flags = (unsigned int *)(base + off_flags);
    double *agg = (double *)(base + off_agg), *incl = (double *)(base + off_incl);
    if (skip4)
        hipLaunchKernelGGL((k_smooth_scan<KS, HS, true>), dim3((unsigned)nwg), dim3(kScanThreads), 0, st, T, n, k, hs, dx, d_seed_row, dy, ticket, flags, agg, incl,
                           tb, d_start, d_end);
    else
        hipLaunchKernelGGL((k_smooth_scan<KS, HS, false>), dim3((unsigned)nwg), dim3(kScanThreads), 0, st, T, n, k, hs, dx, d_seed_row, dy, ticket, flags, agg, incl,
                           tb, d_start, d_end);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

bool smooth_args_ok(snowtri_ctx *ctx, int64_t T, int64_t n, double f, double dt, int memspace) {
    return ctx && T >= 0 && n >= 0 && f > 0.0 && dt > 0.0 && (memspace == SNOWTRI_HOST || memspace == SNOWTRI_DEVICE);
}
int smooth_track_impl(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, double f, double z, double r, double dt, double *y, int memspace,
                      void *stream, bool skip4);

}  // namespace

extern "C" {

int snowtri_smooth_coeffs(double f, double z, double r, double dt, double out[6]) {
    if (!out || !(f > 0.0) || !(dt > 0.0)) return SNOWTRI_ERR_BAD_ARG;
    const SmoothCoef k = smooth_coef(f, z, r, dt);
    out[0] = k.a00; out[1] = k.a01; out[2] = k.a10; out[3] = k.a11; out[4] = k.cx; out[5] = k.cxd;
    return SNOWTRI_OK;
}

int snowtri_smooth_shard_local(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, double f,
                               double z, double r, double dt, double *y, double *end_state, int memspace,
                               void *stream) {
    if (!smooth_args_ok(ctx, T, n, f, dt, memspace)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!x || !y) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const UniformCoef k{smooth_coef(f, z, r, dt)};
    const size_t bytes = sizeof(double) * (size_t)T * n;
    const double *dx = x;
    double *dy = y, *dend = end_state;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->in.ensure(bytes);
        if (rc) return rc;
        rc = ctx->out.ensure(bytes + sizeof(double) * 2 * n);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, x, bytes, hipMemcpyHostToDevice, st));
        dx = (const double *)ctx->in.p;
        dy = (double *)ctx->out.p;
        dend = end_state ? dy + (size_t)T * n : nullptr;
    }
    int rc = smooth_local_dev(ctx, st, T, n, dx, dy, first != 0, k, dend);
    if (rc) return rc;
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(y, dy, bytes, hipMemcpyDeviceToHost, st));
        if (end_state) HIP_TRY(hipMemcpyAsync(end_state, dend, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

int snowtri_smooth_shard_fix(snowtri_ctx *ctx, int64_t T, int64_t n, int first, const double *start_state, double f,
                             double z, double r, double dt, double *y, int memspace, void *stream) {
    if (!smooth_args_ok(ctx, T, n, f, dt, memspace)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!start_state || !y) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const UniformCoef k{smooth_coef(f, z, r, dt)};
    const size_t bytes = sizeof(double) * (size_t)T * n;
    double *dy = y;
    const double *dstart = start_state;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->out.ensure(bytes);
        if (rc) return rc;
        rc = ctx->misc.ensure(sizeof(double) * 2 * n);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->out.p, y, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->misc.p, start_state, sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
        dy = (double *)ctx->out.p;
        dstart = (const double *)ctx->misc.p;
    }
    int rc = smooth_fix_dev(ctx, st, T, n, dy, first != 0, dstart, k);
    if (rc) return rc;
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(y, dy, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

// The two-pass form of the sharded protocol (round 6): _reduce = the shard's zero-state end state from ONE read of x, _scan = the
// shard filtered from its true entering state in one more pass (k_smooth_scan both times); device pointers.
int snowtri_smooth_shard_reduce(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, double f, double z, double r,
                                double dt, double *end_state, void *stream) {
    if (!smooth_args_ok(ctx, T, n, f, dt, SNOWTRI_DEVICE)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!x || !end_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    return smooth_whole_dev(ctx, (hipStream_t)stream, T, n, x, (double *)nullptr, (const double *)nullptr, UniformCoef{smooth_coef(f, z, r, dt)}, NoHold{}, false,
                            first ? 1 : 0, (const double *)nullptr, end_state);
}

int snowtri_smooth_shard_scan(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, const double *start_state, double f,
                              double z, double r, double dt, double *y, void *stream) {
    if (!smooth_args_ok(ctx, T, n, f, dt, SNOWTRI_DEVICE)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!x || !y || !start_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    return smooth_whole_dev(ctx, (hipStream_t)stream, T, n, x, y, (const double *)nullptr, UniformCoef{smooth_coef(f, z, r, dt)}, NoHold{}, false,
                            first ? 1 : 0, start_state, (double *)nullptr);
}

int snowtri_smooth_shard_combine(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t n, const double *gathered, double f,
                                 double z, double r, double dt, double *start_state, int memspace, void *stream) {
    if (!smooth_args_ok(ctx, 0, n, f, dt, memspace) || world < 1 || rank < 0 || rank >= world) return SNOWTRI_ERR_BAD_ARG;
    if (n == 0) return SNOWTRI_OK;
    if (!gathered || !start_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const SmoothCoef k = smooth_coef(f, z, r, dt);
    const size_t gbytes = sizeof(double) * (size_t)world * (4 * (size_t)n + 1), sbytes = sizeof(double) * 2 * (size_t)n;
    const double *dg = gathered;
    double *ds = start_state;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->in.ensure(gbytes);
        if (rc) return rc;
        rc = ctx->misc.ensure(sbytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, gathered, gbytes, hipMemcpyHostToDevice, st));
        dg = (const double *)ctx->in.p;
        ds = (double *)ctx->misc.p;
    }
    hipLaunchKernelGGL(k_smooth_combine<UniformCoef>, dim3((unsigned)((n + kSmoothBlock - 1) / kSmoothBlock)), dim3(kSmoothBlock), 0, st, (int)world,
                       (int)rank, n, dg, UniformCoef{k}, ds);
    HIP_TRY(hipGetLastError());
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(start_state, ds, sbytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

int snowtri_smooth_joint_track(snowtri_ctx *ctx, int64_t T, int64_t m, const double *xyzs, double f, double z, double r,
                               double dt, double *out, int memspace, void *stream) {
    if (m < 0 || m > ((int64_t)1 << 40)) return SNOWTRI_ERR_BAD_ARG;
    return smooth_track_impl(ctx, T, 4 * m, xyzs, f, z, r, dt, out, memspace, stream, true);
}

int snowtri_smooth_track(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, double f, double z, double r,
                         double dt, double *y, int memspace, void *stream) {
    return smooth_track_impl(ctx, T, n, x, f, z, r, dt, y, memspace, stream, false);
}

}  // extern "C"

namespace {
int smooth_track_impl(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, double f, double z, double r,
                      double dt, double *y, int memspace, void *stream, bool skip4) {
    if (!smooth_args_ok(ctx, T, n, f, dt, memspace)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!x || !y) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const UniformCoef k{smooth_coef(f, z, r, dt)};
    const size_t bytes = sizeof(double) * (size_t)T * n;
    const double *dx = x;
    double *dy = y;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->in.ensure(bytes);
        if (rc) return rc;
        rc = ctx->out.ensure(bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, x, bytes, hipMemcpyHostToDevice, st));
        dx = (const double *)ctx->in.p;
        dy = (double *)ctx->out.p;
    }
    int rc = smooth_whole_dev(ctx, st, T, n, dx, dy, dx, k, NoHold{}, skip4);
    if (rc) return rc;
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(y, dy, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------- N4
int snowtri_ctx_set_distortion(snowtri_ctx *ctx, const double *D) {
    if (!ctx || !D || ctx->C <= 0) return SNOWTRI_ERR_BAD_ARG;
    std::vector<double> lens((size_t)ctx->C * kLensStride, 0.0);
    for (int c = 0; c < ctx->C; c++) {
        const double *K = &ctx->hK[9 * (size_t)c];
        // the pinhole part must be [[fx, s, cx], [0, fy, cy], [0, 0, 1]]
        if (K[3] != 0.0 || K[6] != 0.0 || K[7] != 0.0 || K[8] != 1.0 || K[0] == 0.0 || K[4] == 0.0) return SNOWTRI_ERR_BAD_ARG;
        double *L = &lens[(size_t)c * kLensStride];
        L[0] = K[0]; L[1] = K[1]; L[2] = K[2]; L[3] = K[4]; L[4] = K[5];
        L[5] = 1.0 / K[0]; L[6] = 1.0 / K[4];
        for (int i = 0; i < 5; i++) {
            if (!std::isfinite(D[5 * c + i])) return SNOWTRI_ERR_BAD_ARG;
            L[7 + i] = D[5 * c + i];
        }
    }
    ENTER_DEVICE(ctx->device);
    if (!ctx->dLens) HIP_TRY(hipMalloc(&ctx->dLens, sizeof(double) * lens.size()));
    HIP_TRY(hipMemcpy(ctx->dLens, lens.data(), sizeof(double) * lens.size(), hipMemcpyHostToDevice));
    return SNOWTRI_OK;
}

int snowtri_undistort_keypoints(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J, const void *kpts_in,
                                void *kpts_out, int dtype, int memspace, void *stream) {
    if (!ctx || F < 0 || Pmax <= 0 || J <= 0 || (dtype != SNOWTRI_F32 && dtype != SNOWTRI_F64) ||
        (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE) || ctx->C <= 0 || !ctx->dLens)
        return SNOWTRI_ERR_BAD_ARG;
    if (F == 0) return SNOWTRI_OK;
    if (!kpts_in || !kpts_out) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t per_cam = (int64_t)Pmax * J, n_obs = F * ctx->C * per_cam;
    if (per_cam > INT32_MAX) return SNOWTRI_ERR_BAD_ARG;
    const size_t bytes = (size_t)n_obs * 3 * (dtype == SNOWTRI_F32 ? 4 : 8);
    const void *din = kpts_in;
    void *dout = kpts_out;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->in.ensure(bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, kpts_in, bytes, hipMemcpyHostToDevice, st));
        din = dout = ctx->in.p;
    }
    const dim3 grid((unsigned)((n_obs + 255) / 256));
    if (dtype == SNOWTRI_F32)
        hipLaunchKernelGGL(k_undistort<float>, grid, dim3(256), 0, st, n_obs, (int)ctx->C, (int)per_cam,
                           (const double *)ctx->dLens, (const float *)din, (float *)dout);
    else
        hipLaunchKernelGGL(k_undistort<double>, grid, dim3(256), 0, st, n_obs, (int)ctx->C, (int)per_cam,
                           (const double *)ctx->dLens, (const double *)din, (double *)dout);
    HIP_TRY(hipGetLastError());
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(kpts_out, dout, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

// ---------------------------------------------------------------------------------------- N2
int snowtri_blender_points(snowtri_ctx *ctx, int64_t n, int32_t keypoint_num, const void *xyzs, int xyz_dtype,
                           double *out_points, uint8_t *out_valid, int memspace, void *stream) {
    if (!ctx || n < 0 || (xyz_dtype != SNOWTRI_F32 && xyz_dtype != SNOWTRI_F64) ||
        (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE))
        return SNOWTRI_ERR_BAD_ARG;
    if (keypoint_num < kBlenderMinJoints) return SNOWTRI_ERR_BAD_INDEX;  // person[129] would raise IndexError
    if (n == 0) return SNOWTRI_OK;
    if (!xyzs || !out_points || !out_valid) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = xyz_dtype == SNOWTRI_F32 ? 4 : 8;
    const size_t in_bytes = esz * 4 * (size_t)keypoint_num * n, pt_bytes = sizeof(double) * 4 * kBlenderPoints * n;
    const void *dx = xyzs;
    double *dp = out_points;
    uint8_t *dv = out_valid;
    if (memspace == SNOWTRI_HOST) {
        int rc = ctx->in.ensure(in_bytes);
        if (rc) return rc;
        rc = ctx->out.ensure(pt_bytes + (size_t)kBlenderPoints * n);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, xyzs, in_bytes, hipMemcpyHostToDevice, st));
        dx = ctx->in.p;
        dp = (double *)ctx->out.p;
        dv = (uint8_t *)ctx->out.p + pt_bytes;
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (xyz_dtype == SNOWTRI_F32)
        hipLaunchKernelGGL(k_blender_points<float>, grid, dim3(256), 0, st, n, (int)keypoint_num, (const float *)dx, dp, dv);
    else
        hipLaunchKernelGGL(k_blender_points<double>, grid, dim3(256), 0, st, n, (int)keypoint_num, (const double *)dx, dp, dv);
    HIP_TRY(hipGetLastError());
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(out_points, dp, pt_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_valid, dv, (size_t)kBlenderPoints * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

// per-bone coefficient table of the context: rebuilt and uploaded only when (fzr, dt) change -- the common caller passes the
// same smooth profile every time, and an upload has to drain the stream first
static int blender_table(snowtri_ctx *ctx, hipStream_t st, const double *fzr, double dt) {
    std::vector<double> key(fzr, fzr + 3 * kBlenderPoints);
    key.push_back(dt);
    if (!ctx->dBlenderTab) HIP_TRY(hipMalloc(&ctx->dBlenderTab, sizeof(SmoothCoef) * kBlenderPoints));
    if (key != ctx->blender_key) {
        SmoothCoef tab[kBlenderPoints];
        for (int i = 0; i < kBlenderPoints; i++) tab[i] = smooth_coef(fzr[3 * i], fzr[3 * i + 1], fzr[3 * i + 2], dt);
        HIP_TRY(hipStreamSynchronize(st));   // kernels of an earlier call may still read the old table
        HIP_TRY(hipMemcpy(ctx->dBlenderTab, tab, sizeof(tab), hipMemcpyHostToDevice));
        ctx->blender_key = key;
    }
    return SNOWTRI_OK;
}
static bool blender_shard_args_ok(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *fzr, double dt) {
    if (!ctx || T < 0 || n_persons < 0 || (fzr && !(dt > 0.0))) return false;
    if (fzr)
        for (int i = 0; i < kBlenderPoints; i++)
            if (!(fzr[3 * i] > 0.0)) return false;
    return true;
}

// ---- N2 on a FRAME-SHARDED track (device pointers only; include/snowtri.h describes the protocol)
int snowtri_blender_hold_shard_last(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *points, const uint8_t *valid,
                                    double *payload, void *stream) {
    if (!blender_shard_args_ok(ctx, T, n_persons, nullptr, 0.0) || !payload) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (n == 0) return SNOWTRI_OK;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (T == 0) {   // an empty shard: nothing found
        HIP_TRY(hipMemsetAsync(payload, 0, sizeof(double) * 2 * (size_t)n, st));
        return SNOWTRI_OK;
    }
    if (!points || !valid) return SNOWTRI_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_hold_block_last, smooth_grid(n, 1), dim3(256), 0, st, T, n, 4, points, valid, payload);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

int snowtri_blender_hold_shard_apply(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t T, int64_t n_persons, const double *points,
                                     const uint8_t *valid, const double *gathered, double *held, void *stream) {
    if (!blender_shard_args_ok(ctx, T, n_persons, nullptr, 0.0) || world < 1 || rank < 0 || rank >= world) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!points || !valid || !gathered || !held) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nchunks = (T - 1 + kSmoothChunk - 1) / kSmoothChunk;   // the filter's chunks: frames 1 .. T-1
    if (nchunks + 1 > 65535) return SNOWTRI_ERR_BAD_ARG;   // (k_hold_fill runs nchunks + 1 block rows: checked BEFORE anything is queued)
    const size_t cn = (size_t)std::max<int64_t>(1, nchunks) * n;
    const size_t off_H = 4096, off_start = off_H + sizeof(double) * cn, off_Hf = off_start + sizeof(double) * cn,
                 off_ent = (off_Hf + cn + 255) & ~(size_t)255;
    int rc = ctx->aux.ensure(off_ent + sizeof(double) * (size_t)n);
    if (rc) return rc;
    char *aux = (char *)ctx->aux.p;
    double *H = (double *)(aux + off_H), *start = (double *)(aux + off_start), *entering = (double *)(aux + off_ent);
    uint8_t *Hf = (uint8_t *)(aux + off_Hf);
    const dim3 g1 = smooth_grid(n, 1);
    hipLaunchKernelGGL(k_hold_entering, g1, dim3(256), 0, st, (int)world, (int)rank, n, gathered, entering);
    if (nchunks > 0)
        hipLaunchKernelGGL(k_hold_last, smooth_grid(n, nchunks), dim3(256), 0, st, T, n, kSmoothChunk, 4, points, valid, H, Hf);
    hipLaunchKernelGGL(k_hold_carry, g1, dim3(256), 0, st, n, nchunks, 4, points, valid, (const double *)H, (const uint8_t *)Hf, start,
                       (const double *)entering);
    hipLaunchKernelGGL(k_hold_fill, smooth_grid(n, nchunks + 1), dim3(256), 0, st, T, n, kSmoothChunk, 4, points, valid,
                       (const double *)start, (const double *)entering, held);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

int snowtri_blender_smooth_shard_local(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *fzr,
                                       double dt, double *y, double *end_state, void *stream) {
    if (!fzr || !blender_shard_args_ok(ctx, T, n_persons, fzr, dt)) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!held || !y) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    return smooth_local_dev(ctx, st, T, n, held, y, first != 0, k, end_state);
}

int snowtri_blender_smooth_shard_reduce(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *fzr,
                                        double dt, double *end_state, void *stream) {
    if (!fzr || !blender_shard_args_ok(ctx, T, n_persons, fzr, dt)) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!held || !end_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    return smooth_whole_dev(ctx, st, T, n, held, (double *)nullptr, (const double *)nullptr, k, NoHold{}, false, first ? 1 : 0, (const double *)nullptr, end_state);
}

int snowtri_blender_smooth_shard_scan(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *start_state,
                                      const double *fzr, double dt, double *y, void *stream) {
    if (!fzr || !blender_shard_args_ok(ctx, T, n_persons, fzr, dt)) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!held || !y || !start_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    return smooth_whole_dev(ctx, st, T, n, held, y, (const double *)nullptr, k, NoHold{}, false, first ? 1 : 0, start_state, (double *)nullptr);
}

int snowtri_blender_smooth_shard_combine(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t n_persons, const double *gathered,
                                         const double *fzr, double dt, double *start_state, void *stream) {
    if (!fzr || !blender_shard_args_ok(ctx, 0, n_persons, fzr, dt) || world < 1 || rank < 0 || rank >= world) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (n == 0) return SNOWTRI_OK;
    if (!gathered || !start_state) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    hipLaunchKernelGGL(k_smooth_combine<TableCoef>, dim3((unsigned)((n + kSmoothBlock - 1) / kSmoothBlock)), dim3(kSmoothBlock), 0, st, (int)world,
                       (int)rank, n, gathered, k, start_state);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

int snowtri_blender_smooth_shard_fix(snowtri_ctx *ctx, int64_t T, int64_t n_persons, int first, const double *start_state, const double *fzr,
                                     double dt, double *y, void *stream) {
    if (!fzr || !blender_shard_args_ok(ctx, T, n_persons, fzr, dt)) return SNOWTRI_ERR_BAD_ARG;
    const int64_t n = n_persons * kBlenderPoints * 4;
    if (T == 0 || n == 0) return SNOWTRI_OK;
    if (!start_state || !y) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    return smooth_fix_dev(ctx, st, T, n, y, first != 0, start_state, k);
}

int snowtri_blender_smooth(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *points,
                           const uint8_t *valid, const double *fzr, double dt, double *out, int memspace,
                           void *stream) {
    if (!ctx || T < 0 || n_persons < 0 || !fzr || !(dt > 0.0) ||
        (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE))
        return SNOWTRI_ERR_BAD_ARG;
    for (int i = 0; i < kBlenderPoints; i++)
        if (!(fzr[3 * i] > 0.0)) return SNOWTRI_ERR_BAD_ARG;
    if (T == 0 || n_persons == 0) return SNOWTRI_OK;
    if (!points || !valid || !out) return SNOWTRI_ERR_BAD_ARG;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = n_persons * kBlenderPoints * 4, nv = n_persons * kBlenderPoints;
    const int64_t nchunks = (T - 1 + kSmoothChunk - 1) / kSmoothChunk;  // the filter's chunks: frames 1..T-1
    if (nchunks > 65535) return SNOWTRI_ERR_BAD_ARG;
    const size_t bytes = sizeof(double) * (size_t)T * n, vbytes = (size_t)T * nv;
    // aux: [coef table | hold scratch (H, start, Hf)] (+ staged valid for host calls)
    const size_t cn = (size_t)std::max<int64_t>(1, nchunks) * n;
    const size_t off_H = 4096, off_start = off_H + sizeof(double) * cn, off_Hf = off_start + sizeof(double) * cn,
                 off_valid = (off_Hf + cn + 255) & ~(size_t)255;
    int rc = ctx->aux.ensure(off_valid + vbytes);
    if (rc) return rc;
    char *aux = (char *)ctx->aux.p;
    // per-bone coefficient table: rebuilt and uploaded only when (fzr, dt) change -- the common caller passes the
    // same smooth profile every time, and an upload has to drain the stream first
    rc = blender_table(ctx, st, fzr, dt);
    if (rc) return rc;
    const double *dx = points;
    const uint8_t *dv = valid;
    double *dy = out;
    if (memspace == SNOWTRI_HOST) {
        rc = ctx->in.ensure(bytes);
        if (rc) return rc;
        rc = ctx->out.ensure(bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in.p, points, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(aux + off_valid, valid, vbytes, hipMemcpyHostToDevice, st));
        dx = (const double *)ctx->in.p;
        dv = (const uint8_t *)(aux + off_valid);
        dy = (double *)ctx->out.p;
    }
    double *H = (double *)(aux + off_H), *start = (double *)(aux + off_start);
    uint8_t *Hf = (uint8_t *)(aux + off_Hf);
    const dim3 g1 = smooth_grid(n, 1);
    if (nchunks > 0)
        hipLaunchKernelGGL(k_hold_last, smooth_grid(n, nchunks), dim3(256), 0, st, T, n, kSmoothChunk, 4, dx, dv, H, Hf);
    hipLaunchKernelGGL(k_hold_carry, g1, dim3(256), 0, st, n, nchunks, 4, dx, dv, (const double *)H, (const uint8_t *)Hf,
                       start);
    HIP_TRY(hipGetLastError());
    const TableCoef k{(const SmoothCoef *)ctx->dBlenderTab, 4, kBlenderPoints};
    const HoldInput hs{dv, (const double *)start, 4};
    // frame 0 is returned as given, NaNs included (blender.py:176); the filters are seeded with the held row
    rc = smooth_whole_dev(ctx, st, T, n, dx, dy, (const double *)start, k, hs);
    if (rc) return rc;
    if (memspace == SNOWTRI_HOST) {
        HIP_TRY(hipMemcpyAsync(out, dy, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SNOWTRI_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------- fused A1..A4
namespace {

constexpr size_t kMaxScratchBytes = (size_t)8 << 30;

template <typename T>
const char *type_name() {
    return sizeof(T) == 4 ? "float" : "double";
}

// Frames per tile for the fast kernel: maximise (lane utilisation of the item loop: T*J items in
// 256-wide passes) x (balance of the tiles over the CUs -- the kernel is fp64-VALU-bound, so a CU
// with one more tile than its neighbours sets the launch time), under the LDS budget.  Prefer
// >= 2 workgroups per CU (latency hiding) unless that costs more than 5 %.
int choose_tile_frames(int64_t F, int J, int kn, int NP, int num_cus, int forced, int wg_per_cu) {
    const size_t lds_cap = wg_per_cu >= 3 ? 52 * 1024 : 64 * 1024;   // (three workgroups share a CU's 160 KB; 53 KB admits two: kRecomputeLdsBytes)
    if (forced >= 1 && forced <= 64 && fused_single_lds_bytes(forced, kn, NP) <= lds_cap) return forced;
    int best = 1;
    double best_score = -1.0;
    for (int T = 1; T <= 64; T++) {
        if (T > 1 && fused_single_lds_bytes(T, kn, NP) > lds_cap) break;
        const int64_t items = (int64_t)T * J;
        const double eff_pass = (double)items / (double)(((items + kBlock - 1) / kBlock) * kBlock);
        const int64_t ntiles = (F + T - 1) / T;
        const int64_t per_cu = (ntiles + num_cus - 1) / num_cus;
        const double eff_bal = (double)ntiles / (double)(per_cu * num_cus);
        double score = eff_pass * eff_bal;
        score *= (double)T / ((double)T + 1.0);  // fixed per-tile work (barriers, centre check, epilogue) ~ one frame
        if (per_cu < 2) score *= 0.80;  // one workgroup per CU = one wave per SIMD: no latency hiding (measured: 51 us vs 42 us)
        if (score > best_score + 1e-9) {
            best_score = score;
            best = T;
        }
    }
    return best;
}

template <int C, int METHOD, typename TIn, typename TOut>
int launch_fused_single(snowtri_ctx *ctx, hipStream_t st, int64_t F, int J, const TIn *d_kpts,
                        const int32_t *d_np, const Params &prm, int Pout, TOut *d_xyzs, TOut *d_ps,
                        int32_t *d_cnt, uint32_t *d_fl) {
    constexpr int NP = C * (C - 1) / 2;
    constexpr int kWgPerCu = FusedShape<C, METHOD, TIn>::kWaves;   // workgroups of four waves = waves per SIMD
    const int resident = ctx->num_cus * 2 * kWgPerCu;  // as many again queued even out the tail (measured at 2 per CU)
    const int T = choose_tile_frames(F, J, prm.kn, NP, ctx->num_cus, 0, kWgPerCu);
    const int64_t ntiles = (F + T - 1) / T;
    const int grid = (int)std::min<int64_t>(ntiles, resident);
    const size_t per_block = general_scratch_bytes(NP, J);
    int rc = ctx->cur->work.ensure(per_block * (size_t)grid);
    if (rc) return rc;
    const size_t lds = fused_single_lds_bytes(T, prm.kn, NP);
    auto kern = k_fused_single<C, METHOD, TIn, TOut>;
    if (lds > 48 * 1024)
        {
            if (ctx->raise_lds((const void *)kern, (int)lds)) {
                g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return SNOWTRI_ERR_HIP;
            }
        }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, F, J, T, ctx->rig(), d_kpts, d_np, prm, Pout,
                       d_xyzs, d_ps, d_cnt, d_fl, (char *)ctx->cur->work.p, per_block);
    HIP_TRY(hipGetLastError());
    static const std::string name = std::string("k_fused_single<") + std::to_string(C) + "," + std::to_string(METHOD) + "," +
                                    type_name<TIn>() + "," + type_name<TOut>() + ">";
    ctx->last_kernels = name.c_str();
    return SNOWTRI_OK;
}

// Production shape of the fast path (snowtri_lean.hpp): float32 outputs, keypoint_num == J == 133, one slot.
// Work is cut into wave tiles of <= kLeanTw frames of equal size (+-1 frame).
//   small batch (<= one full tile per resident wave): one tile per wave over 2 workgroups per CU;
//   larger batch: ~kLeanTilesPerWave tiles per wave and as many workgroups as that takes -- far more than fit on
//   the chip at once, so the hardware dispatcher evens out the tail (measured +3.5 % on a 2 000 000-frame launch
//   against exactly-resident persistent workgroups).
constexpr int kLeanJ = 133;
constexpr int kFusedSingleMaxCams = 4;   // k_fused_single's pairwise item (everything in registers): up to six pairs
constexpr int kLeanTilesPerWave = 4;

template <int C, typename TIn, typename TOut = float>
int launch_fused_lean(snowtri_ctx *ctx, hipStream_t st, int64_t F, const TIn *d_kpts, const int32_t *d_np,
                      const Params &prm, TOut *d_xyzs, TOut *d_ps, int32_t *d_cnt, uint32_t *d_fl) {
    constexpr int kScoreBytes = (int)sizeof(TOut);
    constexpr int NP = C * (C - 1) / 2;
    const int wg_small = ctx->lean_wg_per_cu, tpw = ctx->lean_tiles_per_wave > 0 ? ctx->lean_tiles_per_wave : kLeanTilesPerWave;
    const size_t per_block = general_scratch_bytes(NP, kLeanJ);
    auto kern = k_fused_lean<C, TIn, kLeanJ, TOut>;
    const int64_t W_small = (int64_t)ctx->num_cus * wg_small * kLeanWaves;
    const int64_t seg_max = (int64_t)kLeanTw * kLeanWaves * 512 * ((int64_t)ctx->num_cus * 6);  // <= 512 tiles per wave
    // small launch (at most kCoopMaxFrames frames per resident workgroup): workgroup tiles, passes dealt to the waves,
    // cooperative epilogue (k_fused_lean_coop)
    if (ctx->lean_coop != 0 && F <= (int64_t)ctx->num_cus * wg_small * kCoopMaxFrames) {
        auto kc = k_fused_lean_coop<C, TIn, kLeanJ, TOut>;
        const int grid = (int)std::min<int64_t>(F, (int64_t)ctx->num_cus * wg_small);
        const int base = (int)(F / grid);
        const int64_t rem = F % grid;
        const int nf_max = base + (rem ? 1 : 0);
        const size_t lds = lean_coop_lds_bytes(C, kLeanJ, nf_max, kScoreBytes);
        int rc = ctx->cur->work.ensure(per_block * (size_t)grid);
        if (rc) return rc;
        if (lds > 48 * 1024 && ctx->raise_lds((const void *)kc, (int)lds)) return SNOWTRI_ERR_HIP;
        if (ctx->debug) fprintf(stderr, "k_fused_lean_coop: F %lld grid %d frames per tile %d (+1 for %lld) lds %zu\n", (long long)F, grid, base, (long long)rem, lds);
        if (ctx->timing && ctx->timing_attach) {
            // the ring's event pair rides on the dispatch itself: begin and end of THIS kernel as its completion signal records
            // them (what rocprofv3's kernel trace reads), without the two barrier packets of a bracketing pair in the interval
            const int64_t slot = ctx->ev_count % kTimingRing;
            hipExtLaunchKernelGGL(kc, dim3(grid), dim3(kBlock), lds, st, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1], 0,
                                  F, base, rem, nf_max, ctx->rig(), d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl, (char *)ctx->cur->work.p, per_block);
            ctx->ev_attached = true;
        } else {
            hipLaunchKernelGGL(kc, dim3(grid), dim3(kBlock), lds, st, F, base, rem, nf_max, ctx->rig(), d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl,
                               (char *)ctx->cur->work.p, per_block);
        }
        HIP_TRY(hipGetLastError());
        static const std::string cname = std::string("k_fused_lean_coop<") + std::to_string(C) + "," + type_name<TIn>() + "," +
                                         std::to_string(kLeanJ) + (kScoreBytes == 8 ? ",double>" : ">");
        ctx->last_kernels = cname.c_str();
        return SNOWTRI_OK;
    }
    if (ctx->timing && ctx->timing_attach && F > seg_max) {   // several kernels: bracketed (fused_dispatch records the end)
        HIP_TRY(hipEventRecord(ctx->ev[0], st));
        HIP_TRY(hipEventRecord(ctx->ev_ring[2 * (ctx->ev_count % kTimingRing)], st));
    }
    for (int64_t s0 = 0; s0 < F; s0 += seg_max) {
        const int64_t Fs = std::min<int64_t>(seg_max, F - s0);
        int64_t W;  // waves of the launch
        if (Fs <= W_small * kLeanTw) {
            W = std::min<int64_t>(W_small, Fs);
        } else {
            const int64_t tiles0 = (Fs + kLeanTw - 1) / kLeanTw;
            W = std::max<int64_t>(W_small, (tiles0 + tpw - 1) / tpw);
        }
        // every workgroup owns a slab for the frames it has to re-do: keep that scratch under 256 MB
        const size_t scratch_mb = (size_t)ctx->lean_scratch_mb;
        const int64_t grid_cap = std::max<int64_t>((int64_t)ctx->num_cus * 6, (int64_t)((scratch_mb << 20) / per_block));
        const int grid = (int)std::min<int64_t>(grid_cap, (W + kLeanWaves - 1) / kLeanWaves);
        W = (int64_t)grid * kLeanWaves;
        const int64_t tiles_per_wave = (Fs + W * kLeanTw - 1) / (W * kLeanTw);
        const int64_t ntiles = std::min<int64_t>(Fs, W * tiles_per_wave);
        const int base = (int)(Fs / ntiles);
        const int64_t rem = Fs % ntiles;
        const int slow_words = (int)((tiles_per_wave * kLeanWaves * (1 << kLeanSlowShift) + 31) / 32);
        const size_t lds = lean_lds_bytes(C, kLeanJ, slow_words, kScoreBytes);
        int rc = ctx->cur->work.ensure(per_block * (size_t)grid);
        if (rc) return rc;
        if (lds > 48 * 1024)
            {
            if (ctx->raise_lds((const void *)kern, (int)lds)) {
                g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return SNOWTRI_ERR_HIP;
            }
        }
        if (ctx->debug) {
            int occ = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBlock, lds);
            fprintf(stderr, "k_fused_lean: F %lld grid %d tiles %lld (%d frames +1 for %lld) lds %zu occupancy/CU %d\n",
                    (long long)Fs, grid, (long long)ntiles, base, (long long)rem, lds, occ);
        }
        if (ctx->timing && ctx->timing_attach && Fs == F) {   // one segment = one kernel: its own begin / end (see k_fused_lean_coop above)
            const int64_t slot = ctx->ev_count % kTimingRing;
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1], 0,
                                  Fs, ntiles, base, rem, slow_words, ctx->rig(), d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl,
                                  (char *)ctx->cur->work.p, per_block);
            ctx->ev_attached = true;
        } else {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, Fs, ntiles, base, rem, slow_words, ctx->rig(),
                               d_kpts + s0 * (int64_t)(C * kLeanJ * 3), d_np ? d_np + s0 * C : nullptr, prm,
                               d_xyzs + s0 * (int64_t)(kLeanJ * 4), d_ps ? d_ps + s0 : nullptr, d_cnt + s0,
                               d_fl ? d_fl + s0 : nullptr, (char *)ctx->cur->work.p, per_block);
        }
        HIP_TRY(hipGetLastError());
    }
    static const std::string name = std::string("k_fused_lean<") + std::to_string(C) + "," + type_name<TIn>() + "," +
                                    std::to_string(kLeanJ) + (kScoreBytes == 8 ? ",double>" : ">");
    ctx->last_kernels = name.c_str();
    return SNOWTRI_OK;
}

template <typename TIn, typename TOut>
int launch_frame_general(snowtri_ctx *ctx, hipStream_t st, int64_t F, int Pmax, int J, const TIn *d_kpts,
                         const int32_t *d_np, const Params &prm, int Pout, TOut *d_xyzs, TOut *d_ps,
                         int32_t *d_cnt, uint32_t *d_fl) {
    const int64_t Kc = snowtri_num_candidate_slots(ctx->C, Pmax);
    if (Kc > kCondenseMaxN) return SNOWTRI_ERR_BAD_ARG;
    // a single camera has no pair (Kc == 0): the kernel then writes count = 0 and zero-filled slots, like the
    // reference's empty candidate list; keep the slab size non-zero for the division below
    const size_t per_block = std::max<size_t>(256, general_scratch_bytes(Kc, J));
    int64_t grid = std::min<int64_t>(F, (int64_t)ctx->num_cus * 2);
    grid = std::max<int64_t>(1, std::min<int64_t>(grid, (int64_t)(kMaxScratchBytes / per_block)));
    int rc = ctx->cur->work.ensure(per_block * (size_t)grid);
    if (rc) return rc;
    const size_t lds = condense_lds_bytes((int)Kc);
    auto kern = k_frame_general<TIn, TOut>;
    if (lds > 48 * 1024)
        {
            if (ctx->raise_lds((const void *)kern, (int)lds)) {
                g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return SNOWTRI_ERR_HIP;
            }
        }
    hipLaunchKernelGGL(kern, dim3((int)grid), dim3(kBlock), lds, st, F, Pmax, J, (int)Kc, ctx->rig(), d_kpts, d_np,
                       prm, Pout, d_xyzs, d_ps, d_cnt, d_fl, (char *)ctx->cur->work.p, per_block);
    HIP_TRY(hipGetLastError());
    static const std::string name = std::string("k_frame_general<") + type_name<TIn>() + "," + type_name<TOut>() + ">";
    ctx->last_kernels = name.c_str();
    return SNOWTRI_OK;
}

// The streaming kernels for the descriptors a k_frame_recompute launch over Fs frames left behind (snowtri_cluster.hpp):
// `cnt[0]` complete-graph clusters in desc[0, cap), `cnt[1]` clusters of any other shape in desc[cap, 2 cap) with their
// member words in `words`.
// method = SNOWTRI_DLT, one detection per camera, keypoint_num == J == 133, one slot: k_dlt_coop (snowtri_dlt_lean.hpp).  Workgroup
// tiles of equal size (+-1 frame): as many as the resident workgroups for a small batch, tiles of kCoopMaxFrames frames beyond.
template <int C, typename TIn, typename TOut>
int launch_dlt_coop(snowtri_ctx *ctx, hipStream_t st, int64_t F, const TIn *d_kpts, const int32_t *d_np, const Params &prm, TOut *d_xyzs,
                    TOut *d_ps, int32_t *d_cnt, uint32_t *d_fl) {
    const int64_t resident = (int64_t)ctx->num_cus * kDltCoopWaves;   // (2 / 3 / 4 / 6 tiles per CU: 27.2 / 26.7 / 26.5 / 27.8 us per 10 000 frames of 4 x 1)
    const int64_t tiles = std::max<int64_t>((F + kCoopMaxFrames - 1) / kCoopMaxFrames, std::min<int64_t>(F, resident));
    const int base = (int)(F / tiles);
    const int64_t rem = F % tiles;
    const int nf_max = base + (rem ? 1 : 0);
    const size_t lds = dlt_coop_lds_bytes(C, kLeanJ, nf_max, (int)sizeof(TOut));
    auto kern = k_dlt_coop<C, TIn, kLeanJ, TOut>;
    if (lds > 48 * 1024 && ctx->raise_lds((const void *)kern, (int)lds)) return SNOWTRI_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kBlock), lds, st, F, base, rem, nf_max, ctx->rig(), d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl);
    HIP_TRY(hipGetLastError());
    static const std::string name = std::string("k_dlt_coop<") + std::to_string(C) + "," + type_name<TIn>() + ",133," + type_name<TOut>() + ">";
    ctx->last_kernels = name.c_str();
    return SNOWTRI_OK;
}

template <int C, typename TIn, typename TOut>
int launch_cluster_fuse(snowtri_ctx *ctx, hipStream_t st, int64_t Fs, int Pmax, int J, const TIn *d_kpts, const Params &prm,
                        int Pout, TOut *d_xyzs, uint32_t *d_fl, const ClusterDesc *desc, const uint32_t *words,
                        const unsigned long long *cnt, uint32_t cap) {
    const int kn = prm.kn;
    const int64_t passes_max = (Fs * Pout * (int64_t)kn + 63) / 64;
    // exactly the waves that are resident at once (kClusterWaves per SIMD at this kernel's registers), each striding over
    // the passes: with the member lists in a kernel of their own every pass costs the same, and short-lived workgroups only
    // added their start-up (8 x 4: 334 -> 312 us).  (k_cluster_members on a second stream beside this kernel: measured, no gain.)
    const int64_t W = std::min<int64_t>(passes_max, (int64_t)ctx->num_cus * 4 * kClusterWaves);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((W + kBlock / 64 - 1) / (kBlock / 64), (int64_t)ctx->num_cus * 64));
    const unsigned long long kmagic = (((unsigned long long)1 << 40) + (unsigned long long)kn - 1) / (unsigned long long)kn;
    hipLaunchKernelGGL((k_cluster_fuse<C, TIn, TOut>), dim3(grid), dim3(kBlock), cluster_lds_bytes(C), st, desc, cnt, cap, ctx->rig(), d_kpts, prm,
                       Pmax, J, kn, kmagic, Pout, d_xyzs, d_fl);
    HIP_TRY(hipGetLastError());
    const int gridm = (int)std::max<int64_t>(1, std::min<int64_t>((passes_max + 3) / 4, (int64_t)ctx->num_cus * 16));
    hipLaunchKernelGGL((k_cluster_members<TIn, TOut>), dim3(gridm), dim3(kBlock), cluster_members_lds_bytes(C, ctx->npairs), st, desc, words, cnt,
                       cap, ctx->rig(), d_kpts, prm, Pmax, J, kn, kmagic, Pout, d_xyzs, d_fl);
    HIP_TRY(hipGetLastError());
    return SNOWTRI_OK;
}

// Launch shape of k_candidate_sums for a rig: threads per workgroup, LDS per workgroup, workgroups per CU.
struct SumsLaunch {
    int threads, lds, per_cu, Jc;
};
SumsLaunch sums_launch_shape(const snowtri_ctx *ctx, int Pmax, int J) {
    const int C = ctx->C, gs = p1_group_size(Pmax);
    const int64_t nitems = (int64_t)ctx->npairs * (Pmax / (gs >= 2 ? kSumsGA : 1)) * (Pmax / gs);   // (k_candidate_sums: GA x GS tiles)
    SumsLaunch L;
    // one pass over the items should keep every wave busy: 256 threads for the small rigs, the whole CU for the large.
    L.threads = ctx->sums_threads > 0 ? ctx->sums_threads : (nitems <= 256 ? 256 : (nitems <= 768 ? 512 : 1024));
    L.threads = L.threads <= 256 ? 256 : (L.threads <= 512 ? 512 : 1024);   // the instantiated shapes
    L.lds = ctx->sums_lds_kb > 0 ? ctx->sums_lds_kb * 1024 : (L.threads <= 256 ? 52 * 1024 : (L.threads <= 512 ? 80 * 1024 : 160 * 1024));
    L.lds = std::min(L.lds, 160 * 1024);
    L.per_cu = std::max(1, std::min((160 * 1024) / L.lds, 2048 / L.threads));
    const int pf = L.threads == 256 ? SumsShape<256>::kPrefetch : (L.threads == 512 ? SumsShape<512>::kPrefetch : SumsShape<1024>::kPrefetch);
    L.Jc = sums_chunk_joints(C, Pmax, J, ctx->npairs, L.threads, pf, L.lds);
    return L;
}

// Multi-person path without HBM candidate spill: k_frame_recompute (snowtri_general.hpp), or -- float32 outputs,
// pairwise method, <= 16 cameras -- the streaming association of snowtri_assoc.hpp with k_frame_recompute for the frames
// it leaves behind.
template <int METHOD, typename TIn, typename TOut>
int launch_frame_recompute(snowtri_ctx *ctx, hipStream_t st_call, int64_t F, int Pmax, int J, const TIn *d_kpts,
                           const int32_t *d_np, const Params &prm, int Pout, TOut *d_xyzs, TOut *d_ps,
                           int32_t *d_cnt, uint32_t *d_fl) {
    const int64_t Kc = snowtri_num_candidate_slots(ctx->C, Pmax);
    const int C = ctx->C, R = C * Pmax;
    const size_t per_block = recompute_scratch_bytes(Kc, R, prm.kn);
    // three workgroups per CU share the 160 KB of LDS (snowtri_general.hpp: the arena behind the tables is reused
    // phase by phase)
    const size_t lds = recompute_launch_lds(C, ctx->npairs);
    auto kern = k_frame_recompute<METHOD, TIn, TOut>;
    snowtri_ctx::OccCache &oc = ctx->recompute_occ[METHOD * 4 + (sizeof(TIn) == 8 ? 2 : 0) + (sizeof(TOut) == 8 ? 1 : 0)];
    if (oc.per_cu < 0 || oc.lds != lds) {
        // exactly the workgroups that are resident at once (registers and the ~50 KB of LDS decide: 3 per CU);
        // they pull frames from an atomic counter until none are left
        if (lds > 48 * 1024)
            {
            if (ctx->raise_lds((const void *)kern, (int)lds)) {
                g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return SNOWTRI_ERR_HIP;
            }
        }
        int q = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, kBlock, lds));
        oc.lds = lds;
        oc.per_cu = q;
        if (ctx->debug) fprintf(stderr, "k_frame_recompute: R %d Kc %lld lds %zu occupancy/CU %d\n", R, (long long)Kc, lds, q);
    }
    int per_cu = oc.per_cu;
    // Hand-over of the output persons to the streaming fusion kernels (snowtri_cluster.hpp): pairwise method, 4-bit person
    // fields, non-negative scores (kthr >= 0).  A person's mean score (:150) is derived from the candidate means when
    // keypoint_num == J and the outputs are float32; otherwise k_person_scores takes it from the fused joints afterwards.  The
    // filter of :151-152 -- which the association must decide BEFORE it assigns the slots -- is vacuous for condense_score_tol
    // <= 0 (the reference's default); keypoint_num < J with an active filter costs a second launch of k_candidate_sums over
    // the first keypoint_num joints (kn / J of the first).
    // handover_mode 1: the streaming association (<= 16 cameras); 2: descriptors written by k_frame_recompute itself
    // (<= 8 cameras, float32 outputs, keypoint_num == J: register-resident rays in k_cluster_fuse), staged in its arena.
    // METHOD = 1 (DLT behind the reference's association, row N3): the streaming association too, then k_cluster_dlt on its
    // descriptors.  A person's mean score there is the mean of the DLT joint scores (mean confidences >= keypoint_score_threshold
    // >= 0), which the association cannot know: the route is taken when the filter of :151-152 cannot drop anybody
    // (condense_score_tol <= 0, the reference's default); otherwise the frames stay on k_frame_recompute<1>.
    const bool can_hand = C >= 2 && Pmax <= kClusterMaxPersons && prm.kn >= 1 && J <= 256 && prm.kthr >= 0.0 && Pout >= 1 &&
                          (size_t)Pout * 24 + 80 <= (size_t)kRayChunkBytes && !(prm.score_tol != prm.score_tol) &&
                          (METHOD == 0 || prm.score_tol <= 0.0);
    // ONE detection per camera with average_score_threshold <= 0 <= keypoint_score_threshold and no mean-score filter: every
    // listed candidate is kept whatever its mean (:80-81; scores that pass the keypoint gate are >= 0), so the candidate pass
    // has nothing to decide.  It is skipped (`sumless`): k_associate reads a constant positive sum for every slot, the
    // persons' mean scores come from the fused joints (k_person_scores), and a singular pair is flagged where its joint is
    // fused (cluster_joint_sequential / cluster_member_passes).  This is the route of single-person rigs of five and more
    // cameras on the shapes the lean kernels do not take (fused_dispatch).
    const bool sumless = METHOD == 0 && Pmax == 1 && prm.avg_thr <= 0.0 && prm.kthr >= 0.0 && prm.score_tol <= 0.0 && ctx->sumless_mode != 0;
    const bool sums_kn = !sumless && prm.kn != J && !(prm.score_tol <= 0.0);   // second candidate-sum launch over the first keypoint_num joints
    const bool post_scores = sumless || sizeof(TOut) == 8 || prm.kn != J || METHOD == 1;   // the persons' mean scores by k_person_scores
    SumsLaunch SL{};
    // (distance_threshold >= 0: k_candidate_sums reads its distance gate off a sign bit, p1_tile_sums)
    bool stream = can_hand && ctx->handover_mode == 1 && C <= 16 && Kc < ((int64_t)1 << 24) && Pout <= 1024 && (sumless || prm.dthr >= 0.0);
    if (stream) {
        SL = sums_launch_shape(ctx, Pmax, J);
        stream = SL.Jc >= 1;
    }
    const bool handover = METHOD == 0 && !stream && can_hand && sizeof(TOut) == 4 && prm.kn == J && ctx->handover_mode != 0 && C <= kClusterMaxCams;
    // the two descriptor lists hold Pout persons for every frame of a segment (<= 2 M entries each, 64 MB together), the
    // member list Kc words per frame (<= 64 M words, 256 MB); row indices are 32-bit; the candidate sums of the
    // streaming association 8 Kc bytes per frame (<= 1 GB).  Sized for the LARGE rigs: a frame of 16 x 8 holds a CU for
    // ~225 us of its candidate pass, a launch ends with up to one such frame per CU in its tail, and 12 500 frames in
    // four segments of 3 125 (12 frames per CU) paid that tail four times -- now two segments of 6 250.
    int64_t seg_frames = F;
    if (stream || handover)
        seg_frames = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(((int64_t)2 << 20) / Pout, ((int64_t)64 << 20) / Kc),
                                                          ((int64_t)1 << 31) / R));
    if (stream) seg_frames = std::max<int64_t>(1, std::min<int64_t>(seg_frames, ((int64_t)128 << 20) / Kc));
    const int64_t seg_cap = ctx->handover_seg_frames > 0 ? ctx->handover_seg_frames : seg_frames;
    int64_t seg = (stream || handover) ? std::min(seg_frames, seg_cap) : F;
    // ONE call in (at least) two segments on two stream sets: every kernel of the streaming route holds the whole chip
    // while it runs, and a third of the route's time is latency-bound (k_associate: VALU busy ~20 %, the member lists);
    // with the segments alternating between the caller's stream and an internal one, those kernels of one segment run
    // beside the VALU-bound ones of the other (8 x 4: +8 % against one stream; what round 3 measured with two CALLS in
    // flight on two contexts, now inside the call).  Outputs do not depend on the cut (tests/test_gpu_handover.py).  Not
    // in overlap mode (whole calls already alternate over the sets) and not for batches too small to fill the chip twice.
    StreamSet *const set_call = ctx->cur;
    // (rigs whose candidate pass holds a whole CU per workgroup -- 16 x 8: 1024 threads, 160 KB -- gain little, only the tails
    // of the kernels fill one another: 17.20 -> 17.01 ms per 12 500 frames, six interleaved runs.  Staggering the segments --
    // the second candidate pass waiting for the first so that it runs beside the first's latency-bound kernels -- LOSES:
    // 8 x 4 1.10 -> 1.24 ms; two half-size candidate passes side by side fill each other's tails, one alone does not.)
    bool split = stream && ctx->split_segments >= 2 && ctx->overlap <= 1 && set_call == &ctx->sets[0];
    if (split) {
        int64_t nseg = std::max<int64_t>((F + seg - 1) / seg, ctx->split_segments);
        nseg += nseg & 1;
        const int64_t even = (F + nseg - 1) / nseg;
        if (even >= (ctx->split_forced ? 1 : (int64_t)ctx->num_cus * 4))   // (the knob set explicitly: tests split small batches too)
            seg = even;
        else
            split = false;
    }
    if (split) {
        if (ctx->ensure_set(1, true, st_call)) {
            g_last_error = "creating the internal stream of the multi-person split failed";
            return SNOWTRI_ERR_HIP;
        }
        HIP_TRY(hipEventRecord(ctx->ev_fork, st_call));
        HIP_TRY(hipStreamWaitEvent(ctx->sets[1].stream, ctx->ev_fork, 0));
    }
    // Whatever way this function is left, the caller's stream continues BEHIND the internal one: a segment already queued
    // there writes the caller's outputs, and an error return in the middle of the loop must not leave it unordered with the
    // caller's stream.  The regular exit joins explicitly and disarms the guard.
    struct SplitJoin {
        snowtri_ctx *ctx;
        hipStream_t caller;
        StreamSet *set_call;
        bool armed;
        ~SplitJoin() {
            ctx->cur = set_call;
            if (armed && hipEventRecord(ctx->sets[1].done, ctx->sets[1].stream) == hipSuccess)
                (void)hipStreamWaitEvent(caller, ctx->sets[1].done, 0);
        }
    } split_join{ctx, st_call, set_call, split};
    const uint32_t *final_slow_list = nullptr;                 // what the streaming association leaves to k_frame_recompute
    const unsigned long long *final_slow_count = nullptr;
    {   // the kernels of this route, in launch order (rebuilt only when the route changes)
        const long long key = ((long long)C << 8) | (METHOD << 7) | ((int)sizeof(TIn) << 3) | ((int)sizeof(TOut) >> 2 << 2) |
                              (stream ? 2 : 0) | (handover ? 1 : 0) | ((long long)SL.threads << 16) | ((long long)(sumless ? 1 : 0) << 32) |
                              ((long long)Pmax << 34);
        if (key != ctx->names_key) {
            const std::string tin = type_name<TIn>(), tout = type_name<TOut>();
            const std::string rec = "k_frame_recompute<" + std::to_string(METHOD) + "," + tin + "," + tout + ">";
            const std::string fuse = C <= kClusterMaxCams ? "k_cluster_fuse<" + std::to_string(C) + "," + tin + "> + k_cluster_members<" + tin + ">"
                                                           : "k_cluster_fuse_wide<" + tin + "> + k_cluster_members<" + tin + ">";
            if (stream && METHOD == 1)
                ctx->names_buf = "k_candidate_sums<" + tin + "," + std::to_string(SL.threads) + "> + k_candidate_sums_exact<" + tin +
                                 "> + k_associate<" + tin + "> + k_cluster_dlt<" + std::to_string(C <= kClusterMaxCams ? C : 0) + "," + tin + "," + tout + "> + k_person_scores<" + tout + "> + " + rec + " (frames left behind)";
            else if (stream && sumless)
                ctx->names_buf = "k_singular_scan<" + tin + "> + k_associate<" + tin + "> + " + fuse + " + k_person_scores<" + tout + "> + " + rec + " (frames left behind)";
            else if (stream)
                ctx->names_buf = "k_candidate_sums<" + tin + "," + std::to_string(SL.threads) + "> + k_candidate_sums_exact<" + tin +
                                 "> + k_associate<" + tin + "> + " + fuse + " + " + rec + " (frames left behind)";
            else if (handover)
                ctx->names_buf = rec + " + " + fuse;
            else
                ctx->names_buf = rec;
            ctx->names_key = key;
        }
        ctx->last_kernels = ctx->names_buf.c_str();
    }
    const unsigned long long kn1 = (unsigned long long)std::max(1, prm.kn), kmagic = (((unsigned long long)1 << 40) + kn1 - 1) / kn1;   // item / keypoint_num
    int seg_index = 0;
    for (int64_t s0 = 0; s0 < F; s0 += seg, seg_index++) {
        const int64_t Fs = std::min<int64_t>(seg, F - s0);
        StreamSet &S = split && (seg_index & 1) ? ctx->sets[1] : *set_call;
        const hipStream_t st = split && (seg_index & 1) ? S.stream : st_call;
        ctx->cur = &S;
        ctx->last_set = &S;
        unsigned long long *next_frame = S.d_counters + 2, *hand_counters = S.d_counters + kHandCountersAt, *slow_count = S.d_counters + 6,
                           *exact_count = S.d_counters + 7, *slow_count2 = S.d_counters + 8, *sums_ticket = S.d_counters + 9;
        int64_t grid = std::min<int64_t>(Fs, (int64_t)ctx->num_cus * std::max(1, per_cu));
        grid = std::max<int64_t>(1, std::min<int64_t>(grid, (int64_t)(kMaxScratchBytes / per_block)));
        int rc = ctx->cur->work.ensure(per_block * (size_t)grid);
        if (rc) return rc;
        uint32_t cap = 0, word_cap = 0;
        ClusterDesc *desc = nullptr;
        uint32_t *words = nullptr;
        if (stream || handover) {
            ctx->last_handover = true;
            cap = (uint32_t)(Fs * Pout);
            word_cap = (uint32_t)(Fs * Kc);
            rc = ctx->cur->desc.ensure((size_t)2 * cap * sizeof(ClusterDesc) + (size_t)word_cap * 4);
            if (rc) return rc;
            desc = (ClusterDesc *)ctx->cur->desc.p;
            words = (uint32_t *)(desc + (size_t)2 * cap);
        }
        HIP_TRY(hipMemsetAsync(next_frame, 0, (kCounterWords - 2) * sizeof(unsigned long long), st));   // next_frame, slow / exact / slow (second pass) frames, the frame tickets of k_candidate_sums, the hand-over list counters
        const TIn *kp_seg = d_kpts + s0 * (int64_t)R * J * 3;
        const int32_t *np_seg = d_np ? d_np + s0 * C : nullptr;
        TOut *xyz_seg = d_xyzs + s0 * (int64_t)Pout * prm.kn * 4;
        TOut *ps_seg = d_ps ? d_ps + s0 * Pout : nullptr;
        uint32_t *fl_seg = d_fl ? d_fl + s0 : nullptr;
        {
            if (stream) {
                // ---- k_candidate_sums -> k_associate -> k_cluster_fuse (DLT: k_cluster_dlt), then k_frame_recompute on the frames left behind
                const size_t sum_bytes = ((size_t)Fs * Kc * 8 + 255) & ~(size_t)255;
                rc = ctx->cur->sums.ensure((sums_kn ? 2 : 1) * sum_bytes + (size_t)Fs * 12 + 256);   // + the frames left behind (two passes) + the frames to re-do exactly
                if (rc) return rc;
                double *csum = (double *)ctx->cur->sums.p;
                double *csum_kn = sums_kn ? (double *)((char *)ctx->cur->sums.p + sum_bytes) : nullptr;
                uint32_t *slow_list = (uint32_t *)((char *)ctx->cur->sums.p + (sums_kn ? 2 : 1) * sum_bytes);
                // first launch: centres in registers where the rig's true pairs fit 4 / 16 rounds of 64 (snowtri_assoc.hpp)
                const AssocShape ash = associate_shape(C, ctx->npairs, Pout, Kc);
                auto k2 = k_associate<TIn, TOut, 0>;   // the second launch (every slot in LDS)
                auto k2a = ash.rc == 4 ? k_associate<TIn, TOut, 4> : (ash.rc == 16 ? k_associate<TIn, TOut, 16> : k2);
                const size_t lds2 = ash.lds;
                if (lds2 > 48 * 1024 && ctx->raise_lds((const void *)k2a, (int)lds2)) return SNOWTRI_ERR_HIP;
                const int grid1 = (int)std::min<int64_t>(Fs, (int64_t)ctx->num_cus * SL.per_cu);
                if (ctx->debug)
                    fprintf(stderr, "k_candidate_sums: threads %d lds %d per_cu %d grid %d Jc %d | k_associate rounds in registers %d lds %zu\n",
                            SL.threads, SL.lds, SL.per_cu, grid1, SL.Jc, ash.rc, lds2);
                uint32_t *exact_list = slow_list + Fs;
#define SNOWTRI_SUMS(TT)                                                                                                             \
    {                                                                                                                                \
        auto k1 = k_candidate_sums<TIn, TT>;                                                                                         \
        if (SL.lds > 48 * 1024 && ctx->raise_lds((const void *)k1, SL.lds)) return SNOWTRI_ERR_HIP;                                 \
        hipLaunchKernelGGL(k1, dim3(grid1), dim3(TT), SL.lds, st, Fs, Pmax, J, J, (int)Kc, ctx->rig(), kp_seg, np_seg, prm, csum, fl_seg, \
                           exact_list, exact_count, sums_ticket, SL.lds);                                                            \
        if (sums_kn) {   /* the sums over the first keypoint_num joints, on a ticket counter of their own */                        \
            hipLaunchKernelGGL(k1, dim3(grid1), dim3(TT), SL.lds, st, Fs, Pmax, prm.kn, J, (int)Kc, ctx->rig(), kp_seg, np_seg, prm,  \
                               csum_kn, (uint32_t *)nullptr, (uint32_t *)nullptr, exact_count, sums_ticket + 1, SL.lds);            \
        }                                                                                                                            \
    }
                if (sumless) {
                    // every byte 0x3f: the double 4.7e-4 in every slot -- positive, finite, kept by :80-81 for any threshold <= 0
                    HIP_TRY(hipMemsetAsync(csum, 0x3f, (size_t)Fs * Kc * 8, st));
                    if (fl_seg) {
                        HIP_TRY(hipMemsetAsync(fl_seg, 0, (size_t)Fs * sizeof(uint32_t), st));   // (the candidate pass clears the flags otherwise)
                        // ... and finds every singular pair of every listed candidate and joint; here a scan asks that one question
                        // (the flag must not depend on the route: k_singular_scan, snowtri_assoc.hpp)
                        hipLaunchKernelGGL((k_singular_scan<TIn>), dim3(grid_for(Fs * (int64_t)J, kBlock, ctx->num_cus * 16)), dim3(kBlock), 0, st, Fs, J,
                                           ctx->rig(), kp_seg, np_seg, fl_seg);
                        HIP_TRY(hipGetLastError());
                    }
                } else {
                    if (SL.threads == 256) SNOWTRI_SUMS(256) else if (SL.threads == 512) SNOWTRI_SUMS(512) else SNOWTRI_SUMS(1024)
                    HIP_TRY(hipGetLastError());
                    // the frames it listed (exact_count of them, known on the device only; normally none)
                    hipLaunchKernelGGL((k_candidate_sums_exact<TIn>), dim3((int)std::min<int64_t>(Fs, ctx->num_cus)), dim3(kBlock), 0, st, Pmax, J,
                                       (int)Kc, ctx->rig(), kp_seg, np_seg, prm, csum, fl_seg, (const uint32_t *)exact_list,
                                       (const unsigned long long *)exact_count);
                    HIP_TRY(hipGetLastError());
                }
#undef SNOWTRI_SUMS
                // small rigs: 16 waves per CU, each striding over the frames; large rigs (a frame takes > 100 us and only a few
                // fit a CU's LDS): one frame per workgroup, the dispatcher evens out the tail
                const int grid2 = (int)std::min<int64_t>(Fs, lds2 > 10 * 1024 ? Fs : (int64_t)ctx->num_cus * ctx->assoc_wg_per_cu);
                hipLaunchKernelGGL(k2a, dim3(grid2), dim3(64), lds2, st, Fs, Pmax, J, (int)Kc, ctx->rig(), kp_seg, np_seg, prm, Pout, csum,
                                   xyz_seg, ps_seg, d_cnt + s0, fl_seg, desc, words, hand_counters, cap, word_cap, slow_list,
                                   slow_count, (int)lds2, 1, (const uint32_t *)nullptr, (const unsigned long long *)nullptr, post_scores ? 0 : 1,
                                   (const double *)csum_kn);
                HIP_TRY(hipGetLastError());
                // the frames whose kept candidates did not fit that LDS (slow_count of them, known on the device only; none on
                // the reference's workloads): again with room for every slot, one wave per CU; what this launch lists is
                // left to k_frame_recompute
                final_slow_list = slow_list;
                final_slow_count = slow_count;
                const size_t lds2_full = associate_lds_bytes_full(C, ctx->npairs, Pout, Kc);
                if (lds2_full > lds2) {
                    if (lds2_full > 48 * 1024 && ctx->raise_lds((const void *)k2, (int)lds2_full)) return SNOWTRI_ERR_HIP;
                    uint32_t *slow_list2 = exact_list + Fs;
                    const int per_cu2 = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds2_full));
                    hipLaunchKernelGGL(k2, dim3((int)std::min<int64_t>(Fs, (int64_t)ctx->num_cus * per_cu2)), dim3(64), lds2_full, st, Fs, Pmax, J,
                                       (int)Kc, ctx->rig(), kp_seg, np_seg, prm, Pout, csum, xyz_seg, ps_seg, d_cnt + s0, fl_seg,
                                       desc, words, hand_counters, cap, word_cap, slow_list2, slow_count2, (int)lds2_full, 1,
                                       (const uint32_t *)slow_list, (const unsigned long long *)slow_count, post_scores ? 0 : 1,
                                       (const double *)csum_kn);
                    HIP_TRY(hipGetLastError());
                    final_slow_list = slow_list2;
                    final_slow_count = slow_count2;
                }
            }
        }
        if (!stream) {
            hipLaunchKernelGGL(kern, dim3((int)grid), dim3(kBlock), lds, st, Fs, Pmax, J, (int)Kc, ctx->rig(), kp_seg, np_seg, prm, Pout,
                               xyz_seg, ps_seg, d_cnt + s0, fl_seg, (char *)ctx->cur->work.p, per_block, next_frame, (int)lds, desc, words,
                               hand_counters, cap, word_cap, (const uint32_t *)nullptr, (const unsigned long long *)nullptr);
            HIP_TRY(hipGetLastError());
        }
        if constexpr (METHOD == 1) {
            if (stream) {
                // both descriptor lists in one launch: an N-view DLT per (output person, joint)
                const int64_t passes_max = (Fs * Pout * (int64_t)prm.kn + 63) / 64;
                const int gridd = (int)std::max<int64_t>(1, std::min<int64_t>((passes_max + 3) / 4, (int64_t)ctx->num_cus * kClusterDltWaves));
                switch (C) {
#define SNOWTRI_CASE(CC)                                                                                                                \
    case CC:                                                                                                                            \
        hipLaunchKernelGGL((k_cluster_dlt<CC, TIn, TOut>), dim3(gridd), dim3(kBlock), cluster_dlt_lds_bytes(C), st, desc, words, hand_counters, cap, \
                           ctx->rig(), kp_seg, prm, Pmax, J, prm.kn, kmagic, Pout, xyz_seg);                                            \
        break;
#ifndef SNOWTRI_DEV_MIN
                    SNOWTRI_CASE(2)
                    SNOWTRI_CASE(3)
                    SNOWTRI_CASE(5)
                    SNOWTRI_CASE(6)
                    SNOWTRI_CASE(7)
#endif
                    SNOWTRI_CASE(4)
                    SNOWTRI_CASE(8)
                    default:   // more than 8 cameras
                        hipLaunchKernelGGL((k_cluster_dlt<0, TIn, TOut>), dim3(gridd), dim3(kBlock), cluster_dlt_lds_bytes(C), st, desc, words, hand_counters, cap,
                                           ctx->rig(), kp_seg, prm, Pmax, J, prm.kn, kmagic, Pout, xyz_seg);
#undef SNOWTRI_CASE
                }
                HIP_TRY(hipGetLastError());
            }
        }
        {
            if constexpr (METHOD == 0) if (stream || handover) {
                switch (C) {
#define SNOWTRI_CASE(CC)                                                                                                       \
    case CC:                                                                                                                   \
        rc = launch_cluster_fuse<CC, TIn, TOut>(ctx, st, Fs, Pmax, J, kp_seg, prm, Pout, xyz_seg, fl_seg, desc, words,       \
                                                hand_counters, cap);                                                           \
        break;
#ifndef SNOWTRI_DEV_MIN
                    SNOWTRI_CASE(2)
                    SNOWTRI_CASE(3)
                    SNOWTRI_CASE(5)
                    SNOWTRI_CASE(6)
                    SNOWTRI_CASE(7)
#endif
                    SNOWTRI_CASE(4)
                    SNOWTRI_CASE(8)
#undef SNOWTRI_CASE
                    default: {   // more than 8 cameras: complete graphs with their rays in LDS, then the member lists
                        const int64_t passes_max = (Fs * Pout * (int64_t)prm.kn + 63) / 64;
                        const int gridm = (int)std::max<int64_t>(1, std::min<int64_t>((passes_max + 3) / 4, (int64_t)ctx->num_cus * 16));
                        if (stream) {
                            auto kw = k_cluster_fuse_wide<TIn, TOut>;
                            const size_t ldsw = cluster_wide_lds_bytes(C);
                            if (ldsw > 48 * 1024 && ctx->raise_lds((const void *)kw, (int)ldsw)) return SNOWTRI_ERR_HIP;
                            const int64_t wpasses = (Fs * Pout * (int64_t)prm.kn + 15) / 16;   // 16 items per wave pass
                            const int ppw_wide = 8;   // (many short-lived workgroups here: 6 -> 886, 24 -> 906, 400 -> 1042 us per 4 000 frames of 16 x 8)
                            const int gridw = (int)std::max<int64_t>(1, std::min<int64_t>((wpasses + 4 * ppw_wide - 1) / (4 * ppw_wide),
                                                                                         (int64_t)ctx->num_cus * 64));
                            hipLaunchKernelGGL(kw, dim3(gridw), dim3(kBlock), ldsw, st, desc, hand_counters, cap, ctx->rig(), kp_seg, prm, Pmax, J,
                                               prm.kn, kmagic, Pout, xyz_seg, fl_seg);
                            HIP_TRY(hipGetLastError());
                        }
                        hipLaunchKernelGGL((k_cluster_members<TIn, TOut>), dim3(gridm), dim3(kBlock), cluster_members_lds_bytes(C, ctx->npairs), st,
                                           desc, words, hand_counters, cap, ctx->rig(), kp_seg, prm, Pmax, J, prm.kn, kmagic, Pout, xyz_seg, fl_seg);
                        rc = hipGetLastError() == hipSuccess ? SNOWTRI_OK : SNOWTRI_ERR_HIP;
                    }
                }
                if (rc) return rc;
            }
            if (stream && post_scores && ps_seg) {
                // the persons' mean scores from the fused joints (keypoint_num < J, float64 outputs)
                const int gridp = (int)std::max<int64_t>(1, std::min<int64_t>((Fs * Pout + 3) / 4, (int64_t)ctx->num_cus * 32));
                hipLaunchKernelGGL((k_person_scores<TOut>), dim3(gridp), dim3(kBlock), 0, st, Fs, Pout, prm.kn, (const TOut *)xyz_seg, ps_seg,
                                   (const int32_t *)(d_cnt + s0));
                HIP_TRY(hipGetLastError());
            }
            if (stream) {
                // the frames k_associate listed (slow_count of them, known on the device only): phase 3 inside the kernel
                const int gridr = (int)std::max<int64_t>(1, std::min<int64_t>(grid, ctx->num_cus));
                hipLaunchKernelGGL(kern, dim3(gridr), dim3(kBlock), lds, st, Fs, Pmax, J, (int)Kc, ctx->rig(), kp_seg, np_seg, prm, Pout,
                                   xyz_seg, ps_seg, d_cnt + s0, fl_seg, (char *)ctx->cur->work.p, per_block, next_frame, (int)lds,
                                   (ClusterDesc *)nullptr, (uint32_t *)nullptr, hand_counters, 0u, 0u,
                                   final_slow_list, final_slow_count);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    ctx->cur = set_call;
    if (split) {   // the caller's stream continues behind the internal one
        split_join.armed = false;
        HIP_TRY(hipEventRecord(ctx->sets[1].done, ctx->sets[1].stream));
        HIP_TRY(hipStreamWaitEvent(st_call, ctx->sets[1].done, 0));
    }
    ctx->last_stream = stream;
    return SNOWTRI_OK;
}

template <typename TIn, typename TOut>
int fused_dispatch(snowtri_ctx *ctx, hipStream_t st, int64_t F, int Pmax, int J, const void *kpts,
                   const int32_t *d_np, const Params &prm, int Pout, void *xyzs, void *ps, int32_t *d_cnt,
                   uint32_t *d_fl, int method) {
    const TIn *d_kpts = (const TIn *)kpts;
    TOut *d_xyzs = (TOut *)xyzs, *d_ps = (TOut *)ps;
    const int C = ctx->C;
    // kthr >= 0: scores that pass the keypoint gate are non-negative, so no candidate mean can fall
    // below avg_thr <= 0 and the fast kernel needs no negative-score check
    const bool fast = Pmax == 1 && C >= 3 && C <= 8 && prm.kn >= 1 && prm.avg_thr <= 0.0 && prm.kthr >= 0.0 &&
                      !((double)ctx->npairs < prm.num_tol) && ctx->general_mode == 0;
    const int64_t ring_slot = ctx->ev_count % kTimingRing;
    ctx->ev_attached = false;
    // (float64 outputs -- the reference's own output type -- on the lean kernels from five cameras on: the rolled item's
    // Newton-refined branch; up to four cameras they stay on k_fused_single<C,0,TIn,double>)
    const bool lean = method != SNOWTRI_DLT && fast && (std::is_same<TOut, float>::value || C >= 5) && J == kLeanJ && prm.kn == kLeanJ &&
                      Pout == 1 && ctx->lean_mode != 0;
    // Five cameras and more on any other shape (float64 outputs, keypoint_num < J, other skeletons, several slots): the
    // unrolled pairwise_item of k_fused_single does not fit the register file there (10-28 pairs: 41-1 075 spilled VGPRs, round-4
    // review).  Those calls take the streaming route WITHOUT its candidate pass (launch_frame_recompute: `sumless`): with one
    // detection per camera and average_score_threshold <= 0 <= keypoint_score_threshold every candidate is kept whatever its
    // mean, k_associate clusters the C(C,2) centre joints, k_cluster_fuse runs the complete-graph item (cluster_item, no
    // scratch) and k_person_scores the mean of :150.
    const bool single_small = fast && C <= kFusedSingleMaxCams;
    // attached timing of a fast-kernel call: NO event record around the dispatch (a record is a barrier packet: the launches of a
    // timing loop would no longer be back to back); launch_fused_lean attaches the ring's pair to the kernel, or brackets a
    // call of several segments itself
    const bool attach = ctx->timing && ctx->timing_attach && lean;
    if (ctx->timing && !attach) {
        HIP_TRY(hipEventRecord(ctx->ev[0], st));
        HIP_TRY(hipEventRecord(ctx->ev_ring[2 * ring_slot], st));
    }
    int rc;
    if (method == SNOWTRI_DLT && (Pmax > 1 || C > 8)) {
        // several detections per camera: the reference's association (phases 1-2), then DLT per cluster
        if (prm.kn > kRecomputeMaxKn || !recompute_shape_ok(C, Pmax, J, ctx->npairs, (int)sizeof(TIn))) {
            g_last_error = "SNOWTRI_DLT with several detections per camera (or more than 8 cameras) supports at most 16 cameras, "
                           "C * Pmax <= 1024 detections per frame and keypoint_num <= 256 (include/snowtri.h); the pairwise method "
                           "has no such limit";
            return SNOWTRI_ERR_BAD_ARG;
        }
        rc = launch_frame_recompute<1, TIn, TOut>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout, d_xyzs, d_ps, d_cnt, d_fl);
    } else if (method == SNOWTRI_DLT) {
        // one detection per camera: no association needed.  The Wholebody skeleton with every joint asked for and one slot: k_dlt_coop
        const bool dlt_lean = J == kLeanJ && prm.kn == kLeanJ && Pout == 1 && ctx->lean_mode != 0;
        switch (C) {
#define SNOWTRI_CASE(CC)                                                                                      \
    case CC:                                                                                                  \
        rc = dlt_lean ? launch_dlt_coop<CC, TIn, TOut>(ctx, st, F, d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl)  \
                      : launch_fused_single<CC, 1, TIn, TOut>(ctx, st, F, J, d_kpts, d_np, prm, Pout, d_xyzs, d_ps, d_cnt, d_fl); \
        break;
#ifndef SNOWTRI_DEV_MIN
            SNOWTRI_CASE(2)
            SNOWTRI_CASE(3)
            SNOWTRI_CASE(5)
            SNOWTRI_CASE(6)
            SNOWTRI_CASE(7)
            SNOWTRI_CASE(8)
#endif
            SNOWTRI_CASE(4)
#undef SNOWTRI_CASE
            default: rc = SNOWTRI_ERR_BAD_ARG;
        }
    } else if (lean) {
        if constexpr (std::is_same<TOut, float>::value) {
            switch (C) {
#define SNOWTRI_CASE(CC)                                                                                          \
    case CC:                                                                                                      \
        rc = launch_fused_lean<CC, TIn>(ctx, st, F, d_kpts, d_np, prm, (float *)xyzs, (float *)ps, d_cnt, d_fl); \
        break;
#ifndef SNOWTRI_DEV_MIN
                SNOWTRI_CASE(3)
                SNOWTRI_CASE(5)
                SNOWTRI_CASE(6)
                SNOWTRI_CASE(7)
                SNOWTRI_CASE(8)
#endif
                SNOWTRI_CASE(4)
#undef SNOWTRI_CASE
                default: rc = SNOWTRI_ERR_BAD_ARG;
            }
        } else {
            switch (C) {
#define SNOWTRI_CASE(CC)                                                                                          \
    case CC:                                                                                                      \
        rc = launch_fused_lean<CC, TIn, double>(ctx, st, F, d_kpts, d_np, prm, d_xyzs, d_ps, d_cnt, d_fl);       \
        break;
#ifndef SNOWTRI_DEV_MIN
                SNOWTRI_CASE(5)
                SNOWTRI_CASE(6)
                SNOWTRI_CASE(7)
                SNOWTRI_CASE(8)
#endif
#undef SNOWTRI_CASE
                default: rc = SNOWTRI_ERR_BAD_ARG;
            }
        }
    } else if (single_small) {
        switch (C) {
#define SNOWTRI_CASE(CC)                                                                                   \
    case CC:                                                                                               \
        rc = launch_fused_single<CC, 0, TIn, TOut>(ctx, st, F, J, d_kpts, d_np, prm, Pout, d_xyzs, d_ps, d_cnt, d_fl); \
        break;
#ifndef SNOWTRI_DEV_MIN
            SNOWTRI_CASE(3)
#endif
            SNOWTRI_CASE(4)
#undef SNOWTRI_CASE
            default: rc = SNOWTRI_ERR_BAD_ARG;
        }
    } else if (prm.kn <= kRecomputeMaxKn && recompute_shape_ok(C, Pmax, J, ctx->npairs, (int)sizeof(TIn)) &&
               ctx->general_mode != 1) {
        rc = launch_frame_recompute<0, TIn, TOut>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout, d_xyzs, d_ps, d_cnt, d_fl);
    } else {
        rc = launch_frame_general<TIn, TOut>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout, d_xyzs, d_ps, d_cnt, d_fl);
    }
    if (rc) return rc;
    if (ctx->timing && ctx->ev_attached) {
        ctx->ev_count++;
        ctx->ev_valid = false;   // (snowtri_last_kernel_ms has no pair of its own for this call)
    } else if (ctx->timing) {
        HIP_TRY(hipEventRecord(ctx->ev_ring[2 * ring_slot + 1], st));
        ctx->ev_count++;
        HIP_TRY(hipEventRecord(ctx->ev[1], st));
        HIP_TRY(hipEventRecord(ctx->ev[2], st));
        HIP_TRY(hipEventRecord(ctx->ev[3], st));
        ctx->ev_valid = true;
        ctx->ev_stream = st;
    }
    return SNOWTRI_OK;
}

}  // namespace

extern "C" int snowtri_triangulate_condense(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J,
                                            const void *kpts, int in_dtype, const int32_t *n_persons,
                                            const snowtri_params *params, int method, int32_t Pout_max,
                                            void *out_xyzs, void *out_pscore, int out_dtype,
                                            int32_t *out_count, uint32_t *out_flags, int memspace,
                                            void *stream) {
    return snowtri_triangulate_condense_ex(ctx, F, Pmax, J, kpts, in_dtype, n_persons, params, method, Pout_max, out_xyzs, out_pscore,
                                           out_dtype, out_count, out_flags, memspace, stream, 0u);
}

extern "C" int snowtri_triangulate_condense_ex(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J,
                                               const void *kpts, int in_dtype, const int32_t *n_persons,
                                               const snowtri_params *params, int method, int32_t Pout_max,
                                               void *out_xyzs, void *out_pscore, int out_dtype,
                                               int32_t *out_count, uint32_t *out_flags, int memspace,
                                               void *stream, uint32_t call_flags) {
    if (!ctx || ctx->C < 1 || F < 0 || Pmax < 1 || J < 1 || Pout_max < 1 || !params) return SNOWTRI_ERR_BAD_ARG;
    if (call_flags & ~(uint32_t)SNOWTRI_CALL_NO_ZERO_FILL) return SNOWTRI_ERR_BAD_ARG;   // (an unknown flag is refused, not ignored)
    if ((in_dtype != SNOWTRI_F32 && in_dtype != SNOWTRI_F64) || (out_dtype != SNOWTRI_F32 && out_dtype != SNOWTRI_F64))
        return SNOWTRI_ERR_BAD_ARG;
    if (memspace != SNOWTRI_HOST && memspace != SNOWTRI_DEVICE) return SNOWTRI_ERR_BAD_ARG;
    if (method != SNOWTRI_PAIRWISE && method != SNOWTRI_DLT) return SNOWTRI_ERR_BAD_ARG;
    if (method == SNOWTRI_DLT && ctx->C < 2) return SNOWTRI_ERR_BAD_ARG;
    if (F == 0) return SNOWTRI_OK;
    if (!kpts || !out_xyzs || !out_count) return SNOWTRI_ERR_BAD_ARG;
    // device buffers: keypoints aligned to their element, joint records [x, y, z, s] to 16 bytes (vector stores)
    if (memspace == SNOWTRI_DEVICE && (((uintptr_t)kpts & (dtype_size(in_dtype) - 1)) != 0 || ((uintptr_t)out_xyzs & 15) != 0))
        return SNOWTRI_ERR_BAD_ARG;
    Params prm;
    int rc = validate_params(params, J, &prm, true);
    if (rc) return rc;
    prm.no_zero_fill = (call_flags & SNOWTRI_CALL_NO_ZERO_FILL) ? 1 : 0;
    ENTER_DEVICE(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    ctx->cur = &ctx->sets[0];
    ctx->last_set = &ctx->sets[0];
    ctx->last_stream = false;
    StreamSet *oset = nullptr;
    if (memspace == SNOWTRI_DEVICE && ctx->overlap >= 2) {
        // overlap mode: this call runs on the next internal stream behind everything `stream` holds now; the caller's
        // stream sees its results after snowtri_ctx_join
        // (sets 1 .. n: set 0 -- scratch, lists, counters -- stays with the caller's stream, where the host calls, the per-frame
        // entry points and the calls made outside overlap mode use it; ADVICE r4: an overlapped call on set 0 shared them unordered)
        const int k = 1 + (int)(ctx->call_index++ % ctx->overlap);
        if (ctx->ensure_set(k)) {
            g_last_error = "creating an internal stream of the overlap mode failed";
            return SNOWTRI_ERR_HIP;
        }
        oset = &ctx->sets[k];
        // the internal stream runs behind whatever the caller's stream holds now; an idle caller stream (the usual loop of
        // calls on resident inputs) needs no event pair: one query instead of a record + a wait per call
        if (hipStreamQuery(st) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipEventRecord(ctx->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(oset->stream, ctx->ev_fork, 0));
        }
        ctx->cur = ctx->last_set = oset;
        st = oset->stream;
    }
    const int kn = prm.kn;
    const size_t isz = dtype_size(in_dtype), osz = dtype_size(out_dtype);
    const size_t in_bytes = (size_t)F * ctx->C * Pmax * J * 3 * isz;
    const size_t np_bytes = n_persons ? sizeof(int32_t) * F * ctx->C : 0;
    const size_t o4 = (size_t)F * Pout_max * kn * 4 * osz, ops = (size_t)F * Pout_max * osz;
    const void *d_kpts = kpts;
    const int32_t *d_np = n_persons;
    void *d_xyzs = out_xyzs, *d_ps = out_pscore;
    int32_t *d_cnt = out_count;
    uint32_t *d_fl = out_flags;
    const size_t np_off = (in_bytes + 15) & ~(size_t)15;
    const size_t ps_off = (o4 + 15) & ~(size_t)15, cnt_off = ps_off + ((ops + 15) & ~(size_t)15);
    const size_t out_total = cnt_off + 8 * (size_t)F;
    const bool pinned = memspace == SNOWTRI_HOST && np_off + np_bytes <= kPinnedMaxBytes && out_total <= kPinnedMaxBytes;
    if (memspace == SNOWTRI_HOST) {
        rc = ctx->in.ensure(np_off + np_bytes + 64);
        if (rc) return rc;
        rc = ctx->out.ensure(out_total + 256);
        if (rc) return rc;
        d_kpts = ctx->in.p;
        if (n_persons) d_np = (int32_t *)((char *)ctx->in.p + np_off);
        if (pinned) {   // one staged upload (page-locked: truly asynchronous)
            rc = ctx->pin_in.ensure(np_off + np_bytes);
            if (rc) return rc;
            rc = ctx->pin_out.ensure(out_total);
            if (rc) return rc;
            std::memcpy(ctx->pin_in.p, kpts, in_bytes);
            if (n_persons) std::memcpy((char *)ctx->pin_in.p + np_off, n_persons, np_bytes);
            HIP_TRY(hipMemcpyAsync(ctx->in.p, ctx->pin_in.p, np_off + np_bytes, hipMemcpyHostToDevice, st));
        } else {
            HIP_TRY(hipMemcpyAsync(ctx->in.p, kpts, in_bytes, hipMemcpyHostToDevice, st));
            if (n_persons) HIP_TRY(hipMemcpyAsync((void *)d_np, n_persons, np_bytes, hipMemcpyHostToDevice, st));
        }
        char *o = (char *)ctx->out.p;
        d_xyzs = o;
        d_ps = o + ps_off;
        d_cnt = (int32_t *)(o + cnt_off);
        d_fl = (uint32_t *)(d_cnt + F);
    } else if (!d_fl) {
        rc = ctx->cur->misc.ensure(sizeof(uint32_t) * F);
        if (rc) return rc;
        d_fl = (uint32_t *)ctx->cur->misc.p;
    }
    // no memsets: the kernels own every output word, including the per-frame flags
    ctx->last_slow_frames = -1;
    ctx->last_handover = false;
    if (in_dtype == SNOWTRI_F32 && out_dtype == SNOWTRI_F32)
        rc = fused_dispatch<float, float>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout_max, d_xyzs, d_ps, d_cnt, d_fl, method);
#ifdef SNOWTRI_DEV_MIN  // kernel-development builds (scripts/ab_build.sh): float32 I/O, 4 cameras only
    else
        rc = SNOWTRI_ERR_BAD_ARG;
#else
    else if (in_dtype == SNOWTRI_F32)
        rc = fused_dispatch<float, double>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout_max, d_xyzs, d_ps, d_cnt, d_fl, method);
    else if (out_dtype == SNOWTRI_F32)
        rc = fused_dispatch<double, float>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout_max, d_xyzs, d_ps, d_cnt, d_fl, method);
    else
        rc = fused_dispatch<double, double>(ctx, st, F, Pmax, J, d_kpts, d_np, prm, Pout_max, d_xyzs, d_ps, d_cnt, d_fl, method);
#endif
    ctx->cur = &ctx->sets[0];
    if (rc) return rc;
    if (oset) oset->pending = true;   // (its `done` event is recorded once, by snowtri_ctx_join)
    if (memspace == SNOWTRI_HOST) {
        std::vector<uint32_t> fl_host;
        uint32_t *fl = out_flags;
        if (pinned) {   // one staged download
            HIP_TRY(hipMemcpyAsync(ctx->pin_out.p, ctx->out.p, out_total, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const char *o = (const char *)ctx->pin_out.p;
            std::memcpy(out_xyzs, o, o4);
            if (out_pscore) std::memcpy(out_pscore, o + ps_off, ops);
            std::memcpy(out_count, o + cnt_off, sizeof(int32_t) * F);
            if (out_flags)
                std::memcpy(out_flags, o + cnt_off + sizeof(int32_t) * F, sizeof(uint32_t) * F);
            else
                fl = (uint32_t *)(o + cnt_off + sizeof(int32_t) * F);
        } else {
            HIP_TRY(hipMemcpyAsync(out_xyzs, d_xyzs, o4, hipMemcpyDeviceToHost, st));
            if (out_pscore) HIP_TRY(hipMemcpyAsync(out_pscore, d_ps, ops, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_count, d_cnt, sizeof(int32_t) * F, hipMemcpyDeviceToHost, st));
            if (!fl) {
                fl_host.resize(F);
                fl = fl_host.data();
            }
            HIP_TRY(hipMemcpyAsync(fl, d_fl, sizeof(uint32_t) * F, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        uint32_t any = 0;
        int64_t slow = 0;
        for (int64_t f = 0; f < F; f++) {
            any |= fl[f];
            slow += (fl[f] & SNOWTRI_FLAG_FASTPATH) ? 0 : 1;
        }
        ctx->last_slow_frames = slow;
        if (any & SNOWTRI_FLAG_SINGULAR) return SNOWTRI_ERR_SINGULAR;
        if (any & SNOWTRI_FLAG_OVERFLOW) return SNOWTRI_ERR_OVERFLOW;
    }
    return SNOWTRI_OK;
}

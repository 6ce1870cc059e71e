/*
 * snowtri.h -- C ABI of the MI355X-native multi-view triangulation core (libsnowtri.so).
 *
 * Drop-in boundary for ONE path of liaochikon/SnowMocap: 2D keypoints from C calibrated cameras
 * -> pairwise two-ray triangulation + scoring -> cross-view association / fusion ("condense").
 * The reference has no FFI of its own (it is pure Python); each entry point below names the
 * reference interface it replaces (file:line relative to the reference tree).  A maintainer binds
 * them with ctypes -- see INTEGRATION.md; snowmocap_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; caller allocates every input and output buffer;
 *   - the library owns only the opaque context (rig constants + device scratch);
 *   - every function returns a snowtri_status; nothing throws across the ABI;
 *   - `memspace` says whether the data pointers are host (SNOWTRI_HOST: staged through the
 *     context's device scratch, synchronous) or device (SNOWTRI_DEVICE: used in place,
 *     asynchronous on `stream`, a hipStream_t passed as void*; NULL = the null stream);
 *   - all arithmetic is IEEE fp64 on the GPU (the reference is NumPy float64); only the I/O
 *     element type is selectable (snowtri_dtype);
 *   - a context is not thread-safe; distinct contexts may be used concurrently.
 *
 * Layouts (row-major, innermost last)
 *   kpts       [F][C][Pmax][J][3]   (u, v, score) per detected keypoint; persons p >= n_persons[f][c] ignored
 *   n_persons  [F][C] int32         detections per camera in add_human_2D_points call order; NULL = Pmax everywhere
 *   candidates slot k = pair(mc<sc) * Pmax*Pmax + pm * Pmax + ps, pairs in the reference's loop order
 *              (triangulation.py:56-65), so increasing k over valid slots IS the reference's list order
 *   out_xyzs   [F][Pout_max][keypoint_num][4]  (x, y, z, keypoint score)
 */
#ifndef SNOWTRI_H
#define SNOWTRI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNOWTRI_VERSION 100 /* 0.1.0 */

typedef struct snowtri_ctx snowtri_ctx;

/* Thresholds of Human_Triangulation (triangulation.py:50) and Human_Triangulation_Condense
 * (triangulation.py:95-100); JSON keys of configs/snowmocap_default_config.json:10-17. */
typedef struct snowtri_params {
    double keypoint_score_threshold;
    double average_score_threshold;
    double distance_threshold;
    double condense_distance_tol;
    double condense_person_num_tol;
    double condense_score_tol;
    int32_t center_point_index;
    int32_t keypoint_num;
} snowtri_params;

typedef enum snowtri_status {
    SNOWTRI_OK = 0,
    SNOWTRI_ERR_BAD_ARG = 1,    /* null pointer, non-positive size, unsupported dtype/method        */
    SNOWTRI_ERR_BAD_INDEX = 2,  /* center_point_index / keypoint_num outside [0, J]  (IndexError)   */
    SNOWTRI_ERR_HIP = 3,        /* a HIP call failed: see snowtri_last_error()                      */
    SNOWTRI_ERR_SINGULAR = 4,   /* an exactly singular ray pair was met (np.linalg.LinAlgError)     */
    SNOWTRI_ERR_OVERFLOW = 5,   /* more output persons than Pout_max in some frame (count is true)  */
    SNOWTRI_ERR_NO_DEVICE = 6   /* no HIP device visible                                            */
} snowtri_status;

typedef enum snowtri_dtype { SNOWTRI_F32 = 0, SNOWTRI_F64 = 1 } snowtri_dtype;
typedef enum snowtri_memspace { SNOWTRI_HOST = 0, SNOWTRI_DEVICE = 1 } snowtri_memspace;
typedef enum snowtri_method {
    SNOWTRI_PAIRWISE = 0, /* the reference's algorithm: pairwise skew-ray midpoints, score-weighted */
    SNOWTRI_DLT = 1       /* N-view DLT (A^T A smallest eigenvector; NOT reference behaviour).  One detection
                           * per camera: no association.  Several: the reference's association (candidates +
                           * greedy clustering: the streaming kernels of the pairwise method), then one DLT per
                           * cluster over its distinct observations (k_cluster_dlt; with condense_score_tol > 0,
                           * which is decided on the DLT joint scores, every frame inside k_frame_recompute<1>).
                           * Shape limits of that multi-detection route (and of DLT with more than 8 cameras):
                           * at most 16 cameras, C * Pmax <= 1024 detections per frame, keypoint_num <= 256;
                           * beyond them snowtri_triangulate_condense returns SNOWTRI_ERR_BAD_ARG and
                           * snowtri_last_error() says why.  SNOWTRI_PAIRWISE has no such limit (larger rigs
                           * fall back to the kernel that spills its candidates to HBM). */
} snowtri_method;

/* per-frame flag bits written to out_flags */
#define SNOWTRI_FLAG_SINGULAR 1u /* exactly singular pair in this frame                  */
#define SNOWTRI_FLAG_OVERFLOW 2u /* more than Pout_max persons; extra persons not written */
#define SNOWTRI_FLAG_FASTPATH 4u /* frame was resolved by the single-cluster fast path    */

int snowtri_version(void);
const char *snowtri_status_string(int status);
/* Message of the last failing HIP call on this thread ("" if none). */
const char *snowtri_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only box; never fails). */
int snowtri_device_count(void);

/* What this binary is: "version=<n>;arch=gfx950;variants=<comma-separated build variants>".  The production library
 * reports an empty variant list; the test build libsnowtri_dbg.so reports SNOWTRI_DEBUG_BOUNDS,SNOWTRI_TEST_KNOBS; development
 * builds name their switches (snowmocap_amd/csrc/snowtri_math.hpp lists them).  Never fails; the string lives as long as the library. */
const char *snowtri_build_info(void);

/* Rig constants.  Replaces Camera.__init__/CameraGroup.__init__ state used by the path
 * (camera.py:17-44,142-157): K[C][9], R[C][9] (camera->world), t[C][3] (camera centre), fp64.
 * C == 0 gives a scratch-only context (enough for snowtri_condense / snowtri_skew_ray_batch). */
int snowtri_ctx_create(int32_t C, const double *K, const double *R, const double *t, int device,
                       snowtri_ctx **out);
int snowtri_ctx_destroy(snowtri_ctx *ctx);
int snowtri_ctx_num_cameras(const snowtri_ctx *ctx);
/* Host copy of the per-camera ray matrices M_c = R_c * inv(K_c), [C][9] fp64. */
int snowtri_ctx_ray_matrices(const snowtri_ctx *ctx, double *M_out);
/* Block until everything queued by this context has finished. */
int snowtri_ctx_synchronize(snowtri_ctx *ctx);
/* Test knobs.  The PRODUCTION library reads no environment and snowtri_ctx_overrides() returns "" for every context of it.
 * The TEST build (-DSNOWTRI_TEST_KNOBS: snowmocap_amd/libsnowtri_dbg.so, which also carries the device-side bounds checks;
 * snowtri_build_info() names both variants) reads these environment variables ONCE, when a context is created, and names the
 * ones that were set as "NAME=value,..." (tests assert on it):
 *   SNOWTRI_GENERAL_MODE=1|2        multi-person batches on the spill kernel / on k_frame_recompute
 *   SNOWTRI_LEAN_MODE=0             float32-output single-detection batches stay on k_fused_single (DLT: off k_dlt_coop)
 *   SNOWTRI_LEAN_COOP=0             small launches stay on k_fused_lean
 *   SNOWTRI_SUMLESS_MODE=0          single-detection batches on the streaming route keep its candidate pass
 *   SNOWTRI_HANDOVER_MODE=0|2       0: the whole multi-person path inside k_frame_recompute; 2: its descriptors to k_cluster_fuse
 *   SNOWTRI_HANDOVER_SEG_FRAMES=n   frames per segment of the streaming multi-person route
 *   SNOWTRI_SPLIT_SEGMENTS=1|n      1: one multi-person call stays on the caller's stream; n >= 2: at least n segments alternating
 *                                   between the caller's stream and an internal one, also for small batches (default: 2
 *                                   segments once a segment holds >= 4 frames per CU)
 *   SNOWTRI_SUMS_THREADS, SNOWTRI_SUMS_LDS_KB, SNOWTRI_LEAN_TILES_PER_WAVE   launch shapes (tests force the rare ones)
 *   SNOWTRI_DEBUG=1                 launch shapes on stderr
 * Results never depend on a knob (that is what the tests that set them check); only the route does. */
const char *snowtri_ctx_overrides(const snowtri_ctx *ctx);

/* OVERLAP MODE for SNOWTRI_DEVICE calls of snowtri_triangulate_condense (off = 1 by default).  With n_streams = 2..4 the
 * context issues consecutive calls round-robin on n internal streams, each behind whatever `stream` held at the time of the
 * call: the ramp-up of one launch overlaps the tail of the previous one (a 10 000-frame step of the single-person path:
 * 26 -> 20 us per call), without the caller creating streams or twin contexts.  The calls must be independent (distinct
 * output buffers; inputs ready on `stream` when the call is made).  Results become visible to the caller's stream at
 * snowtri_ctx_join(ctx, stream) -- `stream` then waits for every call issued since the last join -- or after
 * snowtri_ctx_synchronize.  SNOWTRI_HOST calls are synchronous and ignore the mode.
 * Ordering rule: an overlapped call owns the scratch set of ITS internal stream (one set per stream, none of them the
 * caller's); every other entry point -- host calls, snowtri_triangulate / snowtri_condense_resident, device calls made
 * with the mode off -- works on the caller's own set, ordered by the stream it is given.  So entry points may be mixed
 * freely with overlapped calls in flight; only the OUTPUTS of an overlapped call need the join before they are read. */
int snowtri_ctx_set_overlap(snowtri_ctx *ctx, int n_streams);
int snowtri_ctx_join(snowtri_ctx *ctx, void *stream);
/* THE SPLIT of one multi-person call (several detections per camera, the streaming route).  By default a call whose batch fills
 * the chip at least twice (a segment holds >= 4 frames per CU) runs as two segments that alternate between the caller's stream
 * and ONE internal stream (fork / join events inside the call, results ordered behind `stream` as always): the latency-bound
 * kernels of one segment run beside the VALU-bound ones of the other (8 cameras x 4 persons: +8 %).  Outputs do not depend on
 * the cut (tested bit for bit).  segments = 1: the whole call stays on the caller's stream -- for a caller that captures the call
 * into a hipGraph on one stream, or profiles its kernels one at a time (scripts/pmc_multi.sh); segments = 2 .. 64: at least that
 * many segments (rounded up to an even count), small batches included; 0: the default policy again.  Not used in overlap mode
 * (whole calls alternate there). */
int snowtri_ctx_set_split(snowtri_ctx *ctx, int segments);

/* Test hook (no reference counterpart): evaluates the fast reciprocal / reciprocal-square-root helpers
 * the throughput kernels use (v_rcp_f64 / v_rsq_f64 + Newton steps) on x[n]; host pointers. */
int snowtri_fastmath_probe(snowtri_ctx *ctx, int64_t n, const double *x, double *rcp_nr2_out,
                           double *rcp_nr1_out, double *rsq_nr1_out);

/* Test hook: the RAW v_rcp_f64 / v_rsq_f64 results (no Newton step) on x[n]; host pointers.  The float32-output
 * kernels use the raw 1/sqrt for 1/dist: tests pin its accuracy (<= 2^-23 relative over the normal range). */
int snowtri_fastmath_probe_raw(snowtri_ctx *ctx, int64_t n, const double *x, double *rcp_raw_out, double *rsq_raw_out);
/* Measurement aid: streams a KNOWN number of bytes with the access shapes of the fused kernels -- read_bytes from src
 * as 12-byte records per lane, write_bytes to dst as 16-byte records per lane (device pointers; either may be 0) --
 * so that HBM counters (rocprofv3 FETCH_SIZE / WRITE_SIZE) are calibrated on a kernel other than the one measured. */
int snowtri_calib_stream(snowtri_ctx *ctx, const void *src, int64_t read_bytes, void *dst, int64_t write_bytes,
                         void *stream);

/* A1  CameraGroup.add_human_2D_points (camera.py:234-253): uv[n][2] pixels of camera `cam`
 * -> rays[n][3] = R . inv(K) . [u, v, 1] (un-normalised, world frame).  Host pointers, fp64. */
int snowtri_rays_from_pixels(snowtri_ctx *ctx, int32_t cam, int64_t n, const double *uv, double *rays);

/* A2  Skew_Ray_Solver (triangulation.py:24-31), batched: hm, hs, tm, ts [n][3] -> dist[n], W[n][3].
 * Host pointers, fp64.  *n_singular (optional) counts exactly singular pairs (reference raises). */
int snowtri_skew_ray_batch(snowtri_ctx *ctx, int64_t n, const double *hm, const double *hs,
                           const double *tm, const double *ts, double *dist, double *W,
                           int64_t *n_singular);

/* A1+A3  Human_Triangulation (triangulation.py:50-93), candidates materialised.
 * Outputs are indexed by candidate SLOT (see Layouts), fp64:
 *   cand_xyz[F][Kc][J][3], cand_kscore[F][Kc][J], cand_pscore[F][Kc] (np.mean of kscore),
 *   cand_keep[F][Kc] uint8 = slot is a real pair AND NOT (pscore < average_score_threshold).
 * Kc = snowtri_num_candidate_slots(C, Pmax).  Returns SNOWTRI_ERR_SINGULAR (outputs still
 * written) if any valid pair was exactly singular. */
int snowtri_triangulate(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J, const void *kpts,
                        int in_dtype, const int32_t *n_persons, const snowtri_params *params,
                        double *cand_xyz, double *cand_kscore, double *cand_pscore, uint8_t *cand_keep,
                        int memspace, void *stream);
int64_t snowtri_num_candidate_slots(int32_t C, int32_t Pmax);

/* A4  Human_Triangulation_Condense (triangulation.py:95-162) on materialised candidates.
 * cand_xyz[F][N][J][3], cand_kscore[F][N][J] fp64; cand_keep[F][N] (NULL = every slot is a
 * candidate) selects, in slot order, the candidate list the reference would hold.
 * Outputs fp64: out_xyz[F][Pout_max][kn][3], out_kscore[F][Pout_max][kn], out_pscore[F][Pout_max],
 * out_count[F] (true number of persons), out_flags[F] (optional). */
int snowtri_condense(snowtri_ctx *ctx, int64_t F, int32_t N, int32_t J, const double *cand_xyz,
                     const double *cand_kscore, const uint8_t *cand_keep, const snowtri_params *params,
                     int32_t Pout_max, double *out_xyz, double *out_kscore, double *out_pscore,
                     int32_t *out_count, uint32_t *out_flags, int memspace, void *stream);

/* The candidates of the last SNOWTRI_HOST snowtri_triangulate call on a context stay on the device.
 * snowtri_candidates_token names them (0 = none; any later snowtri_triangulate call on the context replaces them);
 * snowtri_condense_resident runs A4 on them -- the kept slots in slot order, i.e. exactly the list Human_Triangulation
 * returned -- without uploading them again: the reference calls Human_Triangulation and Human_Triangulation_Condense
 * back to back (main.py:62-71), and the second upload was a quarter of the per-frame latency.  Host outputs as
 * snowtri_condense (out_flags may be NULL); SNOWTRI_ERR_BAD_ARG if `token` is not the resident one (the caller then uses
 * snowtri_condense on its own arrays). */
int64_t snowtri_candidates_token(const snowtri_ctx *ctx);
int snowtri_condense_resident(snowtri_ctx *ctx, int64_t token, const snowtri_params *params, int32_t Pout_max, double *out_xyz,
                              double *out_kscore, double *out_pscore, int32_t *out_count, uint32_t *out_flags);

/* A1..A4 fused over a batch of frames -- the hot path.  Replaces the per-frame sequence
 * add_human_2D_points x (C*P) -> Human_Triangulation -> Human_Triangulation_Condense of
 * main.py:50-71,106.  No candidate list ever reaches HBM on the fast path.
 *   kpts [F][C][Pmax][J][3] of in_dtype;  out_xyzs [F][Pout_max][kn][4] of out_dtype
 *   (SNOWTRI_DEVICE: kpts aligned to its element size, out_xyzs to 16 bytes -- else SNOWTRI_ERR_BAD_ARG)
 *   out_pscore [F][Pout_max] of out_dtype (may be NULL), out_count[F] int32, out_flags[F] (may be NULL)
 * Entries of persons >= out_count[f] are zero-filled.  Returns OK / ERR_SINGULAR / ERR_OVERFLOW only
 * for SNOWTRI_HOST calls (device calls are asynchronous: inspect out_flags).
 * Accuracy: all arithmetic is fp64; SNOWTRI_F64 outputs are within 1e-8 m of the reference's.  SNOWTRI_F32 outputs are the
 * fp64 results rounded to float32, i.e. exact to ONE UNIT IN THE LAST PLACE OF THE VALUE: 6e-8 relative -- below the 1e-4 m
 * of the project's parity bar only for coordinates under ~840 m.  Real joints are metres from the origin (error < 1e-6 m);
 * the ghost clusters near-parallel rays produce under a wide condense_distance_tol can lie hundreds of metres out, where
 * a float32 cannot hold 1e-4 m: ask for SNOWTRI_F64 outputs if such points matter.
 * out_pscore (triangulation.py:150, the mean of a person's keypoint scores): where it is taken from the STORED joint scores --
 * keypoint_num < J, SNOWTRI_F64 outputs, one detection per camera on five and more cameras (k_person_scores) -- a SNOWTRI_F32
 * value is the mean of the already rounded scores, up to 2 float32 ulp from the fp64 mean rounded once.
 * SNOWTRI_FLAG_SINGULAR: a pair is exactly singular when a c == b b in separately rounded products (a = hm.hm, b = hm.hs,
 * c = hs.hs: what the reference's LU of [[a, b], [b, c]] sees for equal rays, np.linalg.inv raises, triangulation.py:26). */
int snowtri_triangulate_condense(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J,
                                 const void *kpts, int in_dtype, const int32_t *n_persons,
                                 const snowtri_params *params, int method, int32_t Pout_max,
                                 void *out_xyzs, void *out_pscore, int out_dtype, int32_t *out_count,
                                 uint32_t *out_flags, int memspace, void *stream);

/* The same call with flags (0 = exactly snowtri_triangulate_condense; an unknown bit is SNOWTRI_ERR_BAD_ARG).
 * SNOWTRI_CALL_NO_ZERO_FILL: the slots of persons >= out_count[f] -- out_xyzs[f][p >= count][.][.] and out_pscore[f][p >= count] --
 * are UNSPECIFIED after the call: the library does not spend HBM writes on them (whatever the caller's buffer held may still be
 * there, or zeros where a kernel writes them anyway).  The reference returns LISTS of out_count[f] persons
 * (triangulation.py:154-160); the [F][Pout_max] padding is this ABI's artefact, and on a multi-person batch with a generous
 * Pout_max the zeros are most of what the call writes (BASELINE configs[2], 8 cameras x 4 persons resolving to ~5 persons per
 * frame, Pout_max 16: 234 MB of zeros beside 106 MB of results per 10 000 frames).  A caller that reads out_count[f] first --
 * snowmocap_amd/sharded.py's compact gather, BatchTriangulator(zero_fill=False) -- loses nothing.  SNOWTRI_HOST calls copy the
 * whole padded block back either way (their device staging is not cleared: the unused slots then hold earlier results). */
#define SNOWTRI_CALL_NO_ZERO_FILL 1u
int snowtri_triangulate_condense_ex(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J,
                                    const void *kpts, int in_dtype, const int32_t *n_persons,
                                    const snowtri_params *params, int method, int32_t Pout_max,
                                    void *out_xyzs, void *out_pscore, int out_dtype, int32_t *out_count,
                                    uint32_t *out_flags, int memspace, void *stream, uint32_t call_flags);

/* N1  Human_Triangulation_Smooth / SecondOrderDynamic (triangulation.py:4-22,164-186) over a whole track.
 * x[T][n] fp64, frame-major, n = persons * joints * 3 lanes (persons matched by index, as the reference does)
 * -> y[T][n].  Frame 0 passes through and seeds xp = y = x0, yd = 0; f, z, r, dt as in the reference
 * (main.py:72-77).  Evaluated as ONE pass over HBM -- a chained scan over 256-frame blocks with decoupled look-back: x is read
 * once, y written once (results equal the sequential recurrence to rounding: ~1e-13 m).  T <= 1 + 256 * 65535 frames, n <= 2^22 lanes. */
int snowtri_smooth_track(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, double f, double z, double r,
                         double dt, double *y, int memspace, void *stream);
/* The same on a track of JOINT RECORDS as snowtri_triangulate_condense writes them with SNOWTRI_F64 outputs: xyzs[T][m][4] =
 * (x, y, z, score), m = persons * keypoint_num -> out[T][m][4] with the three coordinates filtered and the score COPIED (the
 * reference filters the points only, triangulation.py:169-184: a caller of snowtri_smooth_track had to filter the score lane
 * for nothing and copy the scores over afterwards).  out may not alias xyzs. */
int snowtri_smooth_joint_track(snowtri_ctx *ctx, int64_t T, int64_t m, const double *xyzs, double f, double z, double r,
                               double dt, double *out, int memspace, void *stream);

/* N1 on a FRAME-SHARDED track (one contiguous frame block per rank).  The filter is linear in its state, so a
 * shard is processed in two calls around ONE small exchange of carries (3n doubles per rank):
 *   snowtri_smooth_shard_local  y = response of the shard from a ZERO entering state (first != 0: this shard
 *                               starts the track, its frame 0 passes through; otherwise every frame is
 *                               filtered and the first one takes xd = 0), end_state[2n] = that response's
 *                               final (y, yd) per lane;
 *   snowtri_smooth_shard_fix    y += response of the true entering state start_state[2n].
 * Between the two, snowtri_smooth_shard_combine turns what the ranks all-gathered -- gathered[world][4n + 1] doubles,
 * per shard in frame order: end_state[2n] | first input row [n] | last input row [n] | its length T_q -- into the
 * start_state[2n] of shard `rank`, on the device and on `stream` (no host round trip between the all-gather and the
 * fix; empty shards have T_q = 0).  CONTRACT on `first`: combine takes the first NON-EMPTY shard of `gathered` as the
 * start of the track (seed (x_first, 0), its frame 0 passes through), so `first != 0` must be given to _local and _fix
 * on exactly that shard -- which is rank 0 only when no leading shard is empty (true for contiguous blocks of
 * ceil(F / world) frames, where only trailing blocks can be empty; a caller with another layout passes first on the
 * rank that holds frame 0).  snowmocap_amd/sharded.py::combine_carries is the same arithmetic on the host
 * (the gloo tests); snowtri_smooth_coeffs returns {a00,a01,a10,a11,cx,cxd} of the update
 * s_t = A s_{t-1} + (0, cx x_t + cxd (x_t - x_{t-1})) it needs. */
int snowtri_smooth_coeffs(double f, double z, double r, double dt, double out[6]);
int snowtri_smooth_shard_local(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, double f,
                               double z, double r, double dt, double *y, double *end_state, int memspace,
                               void *stream);
int snowtri_smooth_shard_fix(snowtri_ctx *ctx, int64_t T, int64_t n, int first, const double *start_state, double f,
                             double z, double r, double dt, double *y, int memspace, void *stream);
int snowtri_smooth_shard_combine(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t n, const double *gathered, double f,
                                 double z, double r, double dt, double *start_state, int memspace, void *stream);
/* The same exchange in TWO passes over the shard instead of five (round 6; DEVICE pointers, asynchronous on `stream`):
 *   snowtri_smooth_shard_reduce  end_state[2n] of the shard's zero-state response -- what _local returns -- from ONE read of x,
 *                                without writing a track;
 *   (the all-gather and snowtri_smooth_shard_combine as above)
 *   snowtri_smooth_shard_scan    y = the shard filtered from its true entering state start_state[2n]: x read once, y written once.
 * 24 bytes moved per lane-frame against the 40 of _local + _fix; same `first` contract; results equal to rounding (~1e-13 m).
 * snowmocap_amd/sharded.py::smooth_exchange2 drives it. */
int snowtri_smooth_shard_reduce(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, double f, double z, double r,
                                double dt, double *end_state, void *stream);
int snowtri_smooth_shard_scan(snowtri_ctx *ctx, int64_t T, int64_t n, const double *x, int first, const double *start_state, double f,
                              double z, double r, double dt, double *y, void *stream);

/* N2  Blender IK control points (blender.py:98-143; names and order of configs/blender_armature_profile.json:
 * root_position, root_rotation, clavicle_r_ik, clavicle_l_ik, arm_r_ik, arm_r_pole, arm_l_ik, arm_l_pole,
 * leg_r_ik, leg_r_pole, leg_l_ik, leg_l_pole, hand_r_ik, hand_r_pole, hand_l_ik, hand_l_pole, foot_r_ik,
 * foot_r_pole, foot_l_ik, foot_l_pole, chest_ik, chest_pole, head_ik, head_pole).
 *   xyzs [n][keypoint_num][4] of xyz_dtype -- n skeletons as written by snowtri_triangulate_condense
 *        (x, y, z, score; the score is not read); keypoint_num >= 130 (COCO-WholeBody joints up to 129 are used),
 *        otherwise SNOWTRI_ERR_BAD_INDEX (the reference raises IndexError);
 *   out_points [n][24][4] fp64: 3-vectors padded with 0, root_rotation as the quaternion (w, x, y, z)
 *        (blender.py:28-29);  out_valid [n][24]: 0 where the point has a NaN component (blender.py:135-139).
 * A NaN root_rotation (coincident hips, or spine parallel to the pelvis axis) makes the reference raise inside
 * SciPy's SVD; here it is reported as out_valid = 0 and the Python mirror raises. */
#define SNOWTRI_BLENDER_POINTS 24
int snowtri_blender_points(snowtri_ctx *ctx, int64_t n, int32_t keypoint_num, const void *xyzs, int xyz_dtype,
                           double *out_points, uint8_t *out_valid, int memspace, void *stream);

/* N2  Human_Triangulation_Blender_Smooth (blender.py:145-178) over a whole track of control points:
 *   points [T][n_persons][24][4] fp64, valid [T][n_persons][24], fzr [24][3] = per-point (f, z, r) of
 *   configs/blender_smooth_profile.json, dt = delta_time  ->  out [T][n_persons][24][4].
 * Frame 0 is returned as given (NaNs included) and seeds every filter with the point, or with zeros when it is
 * invalid; a later invalid point feeds the filter its previous input.  Evaluated as chunked scans over frames. */
int snowtri_blender_smooth(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *points,
                           const uint8_t *valid, const double *fzr, double dt, double *out, int memspace,
                           void *stream);

/* N2 on a FRAME-SHARDED track (one contiguous frame block per rank; DEVICE pointers, asynchronous on `stream`).  An invalid
 * point feeds a filter its previous input (blender.py:157-160), i.e. the filters run on the HELD sequence
 * x_eff[t] = valid[t] ? x[t] : x_eff[t-1] (x_eff[-1] = 0: a filter whose first point is invalid is seeded with zeros,
 * :172-173) -- and on x_eff they are plain linear filters, so a shard takes TWO small exchanges:
 *   1. snowtri_blender_hold_shard_last   payload[2n] = last valid input of the shard per lane | found (1.0 / 0.0) per lane,
 *                                        n = n_persons * 24 * 4 (an empty shard: all zeros);  all-gather the payloads;
 *      snowtri_blender_hold_shard_apply  held[T][n] = x_eff of the shard, entering with the last valid input of the nearest
 *                                        shard before `rank` that has one (gathered[world][2n], frame order);
 *   2. the carry exchange of row N1 on `held` with the per-point coefficients fzr[24][3]:
 *      snowtri_blender_smooth_shard_local / _combine / _fix = snowtri_smooth_shard_local / _combine / _fix (same payload
 *      gathered[world][4n + 1]: end state | first row of held | last row of held | T; same contract on `first`).
 * The shard that starts the track returns its frame 0 as the filter saw it (x_eff[0]); the reference returns the RAW points
 * of frame 0 (NaNs included, blender.py:176): the caller copies that row over (snowmocap_amd/sharded.py does).
 * snowmocap_amd/sharded.py::blender_smooth_sharded drives the protocol; tests compare it with snowtri_blender_smooth. */
int snowtri_blender_hold_shard_last(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *points, const uint8_t *valid,
                                    double *payload, void *stream);
int snowtri_blender_hold_shard_apply(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t T, int64_t n_persons, const double *points,
                                     const uint8_t *valid, const double *gathered, double *held, void *stream);
int snowtri_blender_smooth_shard_local(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *fzr,
                                       double dt, double *y, double *end_state, void *stream);
int snowtri_blender_smooth_shard_combine(snowtri_ctx *ctx, int32_t world, int32_t rank, int64_t n_persons, const double *gathered,
                                         const double *fzr, double dt, double *start_state, void *stream);
int snowtri_blender_smooth_shard_fix(snowtri_ctx *ctx, int64_t T, int64_t n_persons, int first, const double *start_state, const double *fzr,
                                     double dt, double *y, void *stream);
/* ... and its two-pass form on `held` (snowtri_smooth_shard_reduce / _scan with the per-point coefficients). */
int snowtri_blender_smooth_shard_reduce(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *fzr,
                                        double dt, double *end_state, void *stream);
int snowtri_blender_smooth_shard_scan(snowtri_ctx *ctx, int64_t T, int64_t n_persons, const double *held, int first, const double *start_state,
                                      const double *fzr, double dt, double *y, void *stream);

/* N4  Keypoint-level lens undistortion, for detections made on RAW frames (the reference undistorts whole
 * images before detection: main.py:52 cv2.undistort(frame, K, D)).  OpenCV's 5-coefficient Brown-Conrady model,
 * D[C][5] = (k1, k2, p1, p2, k3) per camera (camera_group_floor.json:53-61; Camera.D, camera.py:24,44); K as
 * given to snowtri_ctx_create, which must be [[fx, s, cx], [0, fy, cy], [0, 0, 1]].
 * snowtri_undistort_keypoints maps every (u, v) of kpts [F][C][Pmax][J][3] from the raw image to the
 * undistorted image (exact inverse of the forward model, Newton, 5e-13 px on the shipped rig) and copies the
 * score; kpts_out may alias kpts_in; the result feeds snowtri_triangulate_condense unchanged. */
int snowtri_ctx_set_distortion(snowtri_ctx *ctx, const double *D);
int snowtri_undistort_keypoints(snowtri_ctx *ctx, int64_t F, int32_t Pmax, int32_t J, const void *kpts_in,
                                void *kpts_out, int dtype, int memspace, void *stream);

/* Measurement aid: HIP-event time (ms) of the kernels launched by the LAST
 * snowtri_triangulate_condense call on this context, measured on the stream they ran on
 * (blocks until they finish).  kernel_ms[0] = dominant fused kernel, [1] = everything else. */
int snowtri_last_kernel_ms(snowtri_ctx *ctx, float kernel_ms[2]);
/* With timing enabled every fused call also records a (begin, end) HIP-event pair in a ring (1024 deep), so a
 * caller can queue many launches back to back on one stream and read their individual durations afterwards:
 * writes up to `cap` durations (ms, oldest first) of the calls recorded since the last collect, returns how
 * many (blocks until they have finished; -1 on error). */
int snowtri_timing_collect(snowtri_ctx *ctx, float *kernel_ms, int32_t cap);
/* Toggle per-call event timing (off by default: it adds two event records per launch).  enabled = 1: the ring's pair
 * BRACKETS the call (an event record before its first and after its last kernel: the interval includes the command
 * processor's hand-over from the begin event to the dispatch and from the kernel's end to the end event, ~1 us).
 * enabled = 2: when the call is ONE kernel (the single-detection fast kernels) the pair is ATTACHED to that dispatch
 * (hipExtLaunchKernelGGL's start / stop events): the kernel's own begin and end, the duration a rocprofv3 kernel
 * trace reports, and no event record -- a barrier packet -- sits between consecutive launches, so a timing loop stays
 * back to back (snowtri_last_kernel_ms is not available for such a call); calls of several kernels are bracketed as with 1. */
int snowtri_set_timing(snowtri_ctx *ctx, int enabled);
/* Frames of the last SNOWTRI_HOST fused call that were resolved by the general routine instead of
 * the single-cluster fast path (-1 after a SNOWTRI_DEVICE call: count the frames whose out_flags
 * lack SNOWTRI_FLAG_FASTPATH instead). */
int64_t snowtri_last_slow_frames(snowtri_ctx *ctx);

/* Diagnostics: the kernels the LAST snowtri_triangulate_condense call on this context launched, in launch order,
 * as their template names joined by " + " (e.g. "k_fused_lean<4,float,133>"); "" before the first call.  The string
 * stays valid until the next fused call on this context.  bench.py names the kernel its roofline belongs to with it. */
const char *snowtri_last_kernel_names(const snowtri_ctx *ctx);

/* Device-side bounds checks (debug builds only: `make -C snowmocap_amd/csrc debug` compiles the kernels with
 * -DSNOWTRI_DEBUG_BOUNDS into snowmocap_amd/libsnowtri_dbg.so): every index a kernel derives -- tile and frame ranges,
 * LDS arena offsets, candidate slots, descriptor and member-list positions, person fields -- is checked where it is
 * used.  Returns the number of violated checks since the last call and clears it (0 = clean); *first (may be NULL)
 * receives (check code << 32 | source line) of the first one.  -1: this library was built without the checks
 * (the production build); -2: HIP error.  Synchronises the device. */
int64_t snowtri_debug_faults(snowtri_ctx *ctx, uint64_t *first);
/* Debug builds: launches one wave in which exactly three lanes violate a check with code 99, so that a test can see the
 * mechanism report (snowtri_debug_faults then returns 3).  -1 in the production build. */
int snowtri_debug_selftest(snowtri_ctx *ctx);

/* Test / diagnostics hook for the multi-person path: output persons of the last snowtri_triangulate_condense call
 * (its last segment) whose fusion (triangulation.py:136-152) was handed from the association kernel to the streaming
 * cluster kernels.  Returns the persons whose cluster is the complete graph over one detection per camera, and in
 * *n_other (may be NULL) those of any other shape; -1 / -1 if that call did not arm the hand-over (single-person
 * batches, DLT, more than 16 cameras or 16 persons per camera, a negative keypoint threshold).  Synchronises the device. */
int64_t snowtri_last_handover_persons(snowtri_ctx *ctx, int64_t *n_other);
/* Diagnostics of the streaming multi-person route, last call (its last segment): counts[0] = frames the association
 * kernel could not finish in its first launch (kept list larger than its LDS), counts[1] = frames with a candidate whose sum
 * was re-done with the exact arithmetic, counts[2] = frames left to k_frame_recompute.  All three are 0 on the reference's
 * workloads (tests assert it: a regression that sends every frame down a fall-back still passes parity, but not this);
 * -1 each if that call did not take the route.  Synchronises the device. */
int snowtri_last_stream_counts(snowtri_ctx *ctx, int64_t counts[3]);
/* The context's internal streams (the second stream of the multi-person split, the streams of the overlap mode) are
 * checked when they are created: the HIP runtime multiplexes a process's streams over a few hardware queues, and an
 * internal stream that shares the caller's queue runs one after the other with it.  A pair of one-thread kernels tells
 * (<= 300 us, once per stream; the first call that needs the stream synchronises it and the caller's); a stream that fails
 * is replaced, up to six candidates.  out[0] = probes run, out[1] = streams discarded, out[2] = verdict on the stream kept
 * last (1 side by side, 0 none of the candidates was, -1 no stream created yet or the probe could not run: a caller
 * stream under graph capture is never probed).  That synchronisation happens ONCE per context, in the call that creates
 * the internal stream; a caller stream the context meets later is probed only while it is idle (never waited for), and a
 * context runs at most 16 probes in its life.  No device work. */
int snowtri_ctx_stream_probes(const snowtri_ctx *ctx, int64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif /* SNOWTRI_H */

/* Drives every entry point of include/snowtri.h with invalid arguments (the C ABI must report,
 * never crash or read through a bad pointer).  Plain C99.
 *
 *   - without a GPU (the CPU test builds it against the AddressSanitizer build of the library, `make asan`):
 *     null contexts, impossible sizes, bad enum values, and context creation failing cleanly with
 *     SNOWTRI_ERR_NO_DEVICE;
 *   - with a GPU (tests/test_gpu_parity.py runs it against libsnowtri.so): the same plus, on a real context,
 *     wrong shapes / dtype codes / memory spaces / method, center_point_index and keypoint_num out of range,
 *     missing required pointers, a camera index out of range, too few Blender joints, a singular K.
 * Exit code = number of failed expectations; prints each failure.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "snowtri.h"

static int failures = 0;
#define EXPECT(call, want)                                                                    \
    do {                                                                                      \
        long long got_ = (long long)(call);                                                   \
        if (got_ != (long long)(want)) {                                                      \
            printf("FAIL %s:%d  %s  -> %lld, expected %lld\n", __FILE__, __LINE__, #call, got_, (long long)(want)); \
            failures++;                                                                       \
        }                                                                                     \
    } while (0)

static snowtri_params good_params(int J) {
    snowtri_params p;
    memset(&p, 0, sizeof p);
    p.keypoint_score_threshold = 3.0;
    p.average_score_threshold = 0.0;
    p.distance_threshold = 0.05;
    p.condense_distance_tol = 10.0;
    p.condense_person_num_tol = 0.0;
    p.condense_score_tol = 0.0;
    p.center_point_index = 0;
    p.keypoint_num = J;
    return p;
}

int main(void) {
    double buf[4096];
    float fbuf[4096] __attribute__((aligned(16)));
    int32_t ibuf[64];
    uint32_t ubuf[64];
    uint8_t bbuf[256];
    int64_t nsing = 0;
    double six[6];
    float two[2];
    snowtri_params prm = good_params(5);
    memset(buf, 0, sizeof buf);
    memset(fbuf, 0, sizeof fbuf);
    memset(ibuf, 0, sizeof ibuf);
    memset(ubuf, 0, sizeof ubuf);
    memset(bbuf, 0, sizeof bbuf);

    /* ---- entries that need no context ------------------------------------------------------------------ */
    EXPECT(snowtri_version() >= 100, 1);
    EXPECT(snowtri_status_string(SNOWTRI_OK) != NULL, 1);
    EXPECT(snowtri_status_string(12345) != NULL, 1);
    EXPECT(snowtri_status_string(-7) != NULL, 1);
    EXPECT(snowtri_last_error() != NULL, 1);
    EXPECT(snowtri_device_count() >= 0, 1);
    EXPECT(snowtri_num_candidate_slots(4, 1), 6);
    EXPECT(snowtri_num_candidate_slots(16, 8), 7680);
    EXPECT(snowtri_num_candidate_slots(0, 3), 0);
    EXPECT(snowtri_num_candidate_slots(-2, 3), -1);
    EXPECT(snowtri_num_candidate_slots(4, -1), -1);
    EXPECT(snowtri_smooth_coeffs(2.0, 0.75, 0.0, 1.0 / 30, six), SNOWTRI_OK);
    EXPECT(snowtri_smooth_coeffs(2.0, 0.75, 0.0, 1.0 / 30, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_coeffs(0.0, 0.75, 0.0, 1.0 / 30, six), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_coeffs(2.0, 0.75, 0.0, 0.0, six), SNOWTRI_ERR_BAD_ARG);

    /* ---- null context: every entry refuses before touching anything ------------------------------------- */
    EXPECT(snowtri_ctx_destroy(NULL), SNOWTRI_OK);
    EXPECT(snowtri_ctx_num_cameras(NULL), -1);
    EXPECT(snowtri_ctx_ray_matrices(NULL, buf), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_ctx_synchronize(NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_fastmath_probe(NULL, 4, buf, buf, buf, buf), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_ctx_set_split(NULL, 1), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_rays_from_pixels(NULL, 0, 4, buf, buf), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_skew_ray_batch(NULL, 4, buf, buf, buf, buf, buf, buf, &nsing), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_triangulate(NULL, 1, 1, 5, fbuf, SNOWTRI_F32, NULL, &prm, buf, buf, buf, bbuf, SNOWTRI_HOST, NULL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_condense(NULL, 1, 6, 5, buf, buf, NULL, &prm, 1, buf, buf, buf, ibuf, ubuf, SNOWTRI_HOST, NULL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_triangulate_condense(NULL, 1, 1, 5, fbuf, SNOWTRI_F32, NULL, &prm, SNOWTRI_PAIRWISE, 1, fbuf, fbuf,
                                        SNOWTRI_F32, ibuf, ubuf, SNOWTRI_HOST, NULL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_triangulate_condense_ex(NULL, 1, 1, 5, fbuf, SNOWTRI_F32, NULL, &prm, SNOWTRI_PAIRWISE, 1, fbuf, fbuf,
                                           SNOWTRI_F32, ibuf, ubuf, SNOWTRI_HOST, NULL, SNOWTRI_CALL_NO_ZERO_FILL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_track(NULL, 4, 3, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_joint_track(NULL, 4, 3, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_shard_local(NULL, 4, 3, buf, 1, 2.0, 0.75, 0.0, 1.0 / 30, buf, buf, SNOWTRI_HOST, NULL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_shard_fix(NULL, 4, 3, 1, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL),
           SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_smooth_shard_combine(NULL, 2, 0, 3, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_blender_points(NULL, 1, 133, buf, SNOWTRI_F64, buf, bbuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_blender_smooth(NULL, 2, 1, buf, bbuf, buf, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_ctx_set_distortion(NULL, buf), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_undistort_keypoints(NULL, 1, 1, 5, fbuf, fbuf, SNOWTRI_F32, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_last_kernel_ms(NULL, two), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_timing_collect(NULL, fbuf, 4), -1);
    EXPECT(snowtri_set_timing(NULL, 1), SNOWTRI_ERR_BAD_ARG);
    EXPECT(snowtri_last_slow_frames(NULL), -1);
    EXPECT(snowtri_last_handover_persons(NULL, NULL), -1);
    EXPECT(snowtri_last_kernel_names(NULL)[0], 0);
    EXPECT(snowtri_debug_faults(NULL, NULL) < 0, 1);
    EXPECT(snowtri_debug_selftest(NULL) != SNOWTRI_OK, 1);
    {   /* round 4's entries */
        int64_t counts[3];
        EXPECT(snowtri_build_info() != NULL && strstr(snowtri_build_info(), "arch=gfx950") != NULL, 1);
        EXPECT(snowtri_ctx_overrides(NULL)[0], 0);
        EXPECT(snowtri_ctx_set_overlap(NULL, 2), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_join(NULL, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_last_stream_counts(NULL, counts), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_stream_probes(NULL, counts), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_candidates_token(NULL), 0);
        EXPECT(snowtri_condense_resident(NULL, 1, &prm, 1, buf, buf, buf, ibuf, ubuf), SNOWTRI_ERR_BAD_ARG);
    }

    /* ---- context creation with bad arguments -------------------------------------------------------------- */
    {
        snowtri_ctx *ctx = NULL;
        const double K1[9] = {700, 0, 640, 0, 700, 360, 0, 0, 1}, R1[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t1[3] = {0, 0, 0};
        EXPECT(snowtri_ctx_create(1, K1, R1, t1, 0, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_create(-1, K1, R1, t1, 0, &ctx), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_create(1, NULL, R1, t1, 0, &ctx), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_create(1, K1, NULL, t1, 0, &ctx), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_create(1, K1, R1, NULL, 0, &ctx), SNOWTRI_ERR_BAD_ARG);
        if (snowtri_device_count() == 0) {
            EXPECT(snowtri_ctx_create(1, K1, R1, t1, 0, &ctx), SNOWTRI_ERR_NO_DEVICE);
            EXPECT(ctx == NULL, 1);
            EXPECT(snowtri_ctx_create(0, NULL, NULL, NULL, 0, &ctx), SNOWTRI_ERR_NO_DEVICE);
        } else {
            EXPECT(snowtri_ctx_create(1, K1, R1, t1, snowtri_device_count() + 3, &ctx) != SNOWTRI_OK, 1);
        }
    }

    /* ---- a real context (GPU box only): shapes, enums, indices ------------------------------------------------ */
    if (snowtri_device_count() > 0) {
        snowtri_ctx *ctx = NULL;
        double K[4 * 9], R[4 * 9], t[4 * 3];
        int c, J = 5;
        for (c = 0; c < 4; c++) {
            const double k[9] = {700, 0, 640, 0, 700, 360, 0, 0, 1}, r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(K + 9 * c, k, sizeof k);
            memcpy(R + 9 * c, r, sizeof r);
            t[3 * c] = 2.0 * c;
            t[3 * c + 1] = (c & 1) ? 2.0 : 0.0;
            t[3 * c + 2] = 0.0;
        }
        {   /* singular K: np.linalg.inv(K) raises in the reference (camera.py:242) */
            double Ks[9] = {700, 0, 640, 0, 0, 360, 0, 0, 1};
            EXPECT(snowtri_ctx_create(1, Ks, R, t, 0, &ctx), SNOWTRI_ERR_SINGULAR);
        }
        EXPECT(snowtri_ctx_create(4, K, R, t, 0, &ctx), SNOWTRI_OK);
        EXPECT(snowtri_ctx_num_cameras(ctx), 4);
        EXPECT(snowtri_ctx_ray_matrices(ctx, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_rays_from_pixels(ctx, 4, 1, buf, buf), SNOWTRI_ERR_BAD_INDEX);
        EXPECT(snowtri_rays_from_pixels(ctx, -1, 1, buf, buf), SNOWTRI_ERR_BAD_INDEX);
        EXPECT(snowtri_rays_from_pixels(ctx, 0, -1, buf, buf), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_rays_from_pixels(ctx, 0, 1, NULL, buf), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_skew_ray_batch(ctx, -1, buf, buf, buf, buf, buf, buf, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_skew_ray_batch(ctx, 1, NULL, buf, buf, buf, buf, buf, NULL), SNOWTRI_ERR_BAD_ARG);
        prm = good_params(J);
#define FUSED(F_, P_, J_, kp_, idt_, prm_, meth_, pout_, ox_, odt_, cnt_, ms_)                                          \
    snowtri_triangulate_condense(ctx, F_, P_, J_, kp_, idt_, NULL, prm_, meth_, pout_, ox_, fbuf, odt_, cnt_, ubuf, ms_, NULL)
        EXPECT(FUSED(-1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 0, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, 0, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, 7, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, -1, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, 9, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 0, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, 5), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, NULL, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, NULL, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, NULL, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, NULL, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(0, 1, J, NULL, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, NULL, SNOWTRI_F32, NULL, SNOWTRI_HOST), SNOWTRI_OK); /* empty batch */
        /* device buffers: joint records must be 16-byte aligned, keypoints element-aligned (checked before any launch) */
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, (char *)fbuf + 4, SNOWTRI_F32, ibuf, SNOWTRI_DEVICE), SNOWTRI_ERR_BAD_ARG);
        EXPECT(FUSED(1, 1, J, (char *)fbuf + 2, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_DEVICE), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_set_split(ctx, -1), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_set_split(ctx, 65), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_set_split(ctx, 1), SNOWTRI_OK);
        EXPECT(snowtri_ctx_set_split(ctx, 0), SNOWTRI_OK);
        /* call flags: an unknown bit is refused, not ignored */
        EXPECT(snowtri_triangulate_condense_ex(ctx, 1, 1, J, fbuf, SNOWTRI_F32, NULL, &prm, SNOWTRI_PAIRWISE, 1, fbuf, fbuf, SNOWTRI_F32, ibuf, ubuf,
                                               SNOWTRI_HOST, NULL, 2u), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_triangulate_condense_ex(ctx, 0, 1, J, NULL, SNOWTRI_F32, NULL, &prm, SNOWTRI_PAIRWISE, 1, NULL, NULL, SNOWTRI_F32, NULL, NULL,
                                               SNOWTRI_HOST, NULL, SNOWTRI_CALL_NO_ZERO_FILL), SNOWTRI_OK);
        prm.center_point_index = J;   /* reference: IndexError */
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_INDEX);
        prm = good_params(J);
        prm.keypoint_num = J + 1;
        EXPECT(FUSED(1, 1, J, fbuf, SNOWTRI_F32, &prm, SNOWTRI_PAIRWISE, 1, fbuf, SNOWTRI_F32, ibuf, SNOWTRI_HOST), SNOWTRI_ERR_BAD_INDEX);
        prm = good_params(J);
        EXPECT(snowtri_triangulate(ctx, 1, 1, J, fbuf, 9, NULL, &prm, buf, buf, buf, bbuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_triangulate(ctx, 1, 1, J, fbuf, SNOWTRI_F32, NULL, &prm, NULL, buf, buf, bbuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_condense(ctx, 1, -3, J, buf, buf, NULL, &prm, 1, buf, buf, buf, ibuf, ubuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_condense(ctx, 1, 6, J, NULL, buf, NULL, &prm, 1, buf, buf, buf, ibuf, ubuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_smooth_track(ctx, 4, 3, buf, -2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_smooth_track(ctx, 4, 3, NULL, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_smooth_track(ctx, -1, 3, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_smooth_joint_track(ctx, 4, -1, buf, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_smooth_joint_track(ctx, 4, 3, NULL, 2.0, 0.75, 0.0, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_blender_points(ctx, 1, 100, buf, SNOWTRI_F64, buf, bbuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_INDEX);
        EXPECT(snowtri_blender_points(ctx, 1, 133, buf, 5, buf, bbuf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_blender_smooth(ctx, 2, 1, buf, bbuf, NULL, 1.0 / 30, buf, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_ctx_set_distortion(ctx, NULL), SNOWTRI_ERR_BAD_ARG);
        EXPECT(snowtri_undistort_keypoints(ctx, 1, 1, J, fbuf, fbuf, SNOWTRI_F32, SNOWTRI_HOST, NULL), SNOWTRI_ERR_BAD_ARG); /* no D set */
        EXPECT(snowtri_last_kernel_ms(ctx, NULL), SNOWTRI_ERR_BAD_ARG);
        {   /* round 4's entries on a real context */
            int64_t counts[3] = {7, 7, 7};
            EXPECT(snowtri_ctx_set_overlap(ctx, 0), SNOWTRI_ERR_BAD_ARG);
            EXPECT(snowtri_ctx_set_overlap(ctx, 9), SNOWTRI_ERR_BAD_ARG);
            EXPECT(snowtri_ctx_set_overlap(ctx, 2), SNOWTRI_OK);
            EXPECT(snowtri_ctx_join(ctx, NULL), SNOWTRI_OK);             /* nothing in flight */
            EXPECT(snowtri_ctx_set_overlap(ctx, 1), SNOWTRI_OK);
            EXPECT(snowtri_last_stream_counts(ctx, NULL), SNOWTRI_ERR_BAD_ARG);
            EXPECT(snowtri_last_stream_counts(ctx, counts), SNOWTRI_OK);
            EXPECT(counts[0] == -1 && counts[1] == -1 && counts[2] == -1, 1);   /* no multi-person call yet */
            EXPECT(snowtri_ctx_stream_probes(ctx, NULL), SNOWTRI_ERR_BAD_ARG);
            EXPECT(snowtri_ctx_stream_probes(ctx, counts), SNOWTRI_OK);
            EXPECT(snowtri_condense_resident(ctx, 0, &prm, 1, buf, buf, buf, ibuf, ubuf), SNOWTRI_ERR_BAD_ARG);       /* no candidates resident */
            EXPECT(snowtri_condense_resident(ctx, 123456789, &prm, 1, buf, buf, buf, ibuf, ubuf), SNOWTRI_ERR_BAD_ARG); /* a stale token */
            EXPECT(snowtri_ctx_overrides(ctx) != NULL, 1);
        }
        EXPECT(snowtri_ctx_destroy(ctx), SNOWTRI_OK);
    }
    printf("abi_badargs: %d failure(s)\n", failures);
    return failures;
}

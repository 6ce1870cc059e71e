/* Plain-C consumer of include/snowtri.h: proves the boundary is a C ABI (no C++ / torch types).
 * Built by tests/test_abi_and_host.py (gcc -std=c99, links libsnowtri.so); run on the GPU box by
 * tests/test_gpu_parity.py.  Two cameras looking down +z from (0,0,0) and (2,0,0) plus a third at (0,2,0),
 * K = R = I, one point at (1,0,4): every pair intersects it (nearly) exactly; prints the fused joint. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "snowtri.h"

int main(void) {
    if (snowtri_version() != SNOWTRI_VERSION) return 10;
    if (snowtri_device_count() <= 0) {
        printf("no device: %s\n", snowtri_status_string(SNOWTRI_ERR_NO_DEVICE));
        return 0;
    }
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double K[27], R[27];
    const double t[9] = {0, 0, 0, 2, 0, 0, 0, 2, 0};
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 9; i++) K[9 * c + i] = R[9 * c + i] = I3[i];
    snowtri_ctx *ctx = NULL;
    int rc = snowtri_ctx_create(3, K, R, t, 0, &ctx);
    if (rc) { printf("ctx_create: %s %s\n", snowtri_status_string(rc), snowtri_last_error()); return 11; }
    /* 2 frames x 3 cameras x 1 person x 2 joints; joint 0 at (1,0,4) (+ a small offset so dist != 0), joint 1 at (0.5,0.5,2) */
    const double X[2][3] = {{1.0, 0.0, 4.0}, {0.5, 0.5, 2.0}};
    float kpts[2][3][1][2][3];
    for (int f = 0; f < 2; f++)
        for (int c = 0; c < 3; c++)
            for (int j = 0; j < 2; j++) {
                kpts[f][c][0][j][0] = (float)((X[j][0] - t[3 * c]) / X[j][2]) + 1e-4f * (float)(c + f);
                kpts[f][c][0][j][1] = (float)((X[j][1] - t[3 * c + 1]) / X[j][2]);
                kpts[f][c][0][j][2] = 5.0f;
            }
    snowtri_params p = {3.0, 0.0, 0.05, 10.0, 0.0, 0.0, 0, 2};
    float out[2][1][2][4], ps[2][1];
    int32_t count[2];
    uint32_t flags[2];
    rc = snowtri_triangulate_condense(ctx, 2, 1, 2, kpts, SNOWTRI_F32, NULL, &p, SNOWTRI_PAIRWISE, 1, out, ps, SNOWTRI_F32,
                                      count, flags, SNOWTRI_HOST, NULL);
    if (rc) { printf("triangulate_condense: %s\n", snowtri_status_string(rc)); return 12; }
    for (int f = 0; f < 2; f++)
        for (int j = 0; j < 2; j++) {
            printf("frame %d joint %d: %.5f %.5f %.5f score %.4g (count %d flags %u)\n", f, j, out[f][0][j][0], out[f][0][j][1],
                   out[f][0][j][2], out[f][0][j][3], count[f], flags[f]);
            if (fabs(out[f][0][j][0] - X[j][0]) > 5e-3 || fabs(out[f][0][j][1] - X[j][1]) > 5e-3 ||
                fabs(out[f][0][j][2] - X[j][2]) > 2e-2 || count[f] != 1)
                return 13;
        }
    /* rows after the hot path through the same ABI: lens undistortion with zero coefficients is the identity,
     * a one-lane track passes its first frame through the temporal filter and then moves towards the input */
    const double D[15] = {0};
    rc = snowtri_ctx_set_distortion(ctx, D);
    if (rc) { printf("set_distortion: %s\n", snowtri_status_string(rc)); return 14; }
    float und[2][3][1][2][3];
    rc = snowtri_undistort_keypoints(ctx, 2, 1, 2, kpts, und, SNOWTRI_F32, SNOWTRI_HOST, NULL);
    if (rc) { printf("undistort: %s\n", snowtri_status_string(rc)); return 15; }
    for (int i = 0; i < 2 * 3 * 2 * 3; i++)
        if (fabs(((float *)und)[i] - ((float *)kpts)[i]) > 1e-6) return 16;
    const double track[4] = {0.0, 1.0, 1.0, 1.0};
    double smooth[4];
    rc = snowtri_smooth_track(ctx, 4, 1, track, 2.5, 0.75, 0.0, 1.0 / 30, smooth, SNOWTRI_HOST, NULL);
    if (rc) { printf("smooth_track: %s\n", snowtri_status_string(rc)); return 17; }
    if (smooth[0] != 0.0 || !(smooth[3] > smooth[2] && smooth[2] > smooth[1] && smooth[3] < 1.0)) return 18;
    printf("smooth: %.4f %.4f %.4f %.4f\n", smooth[0], smooth[1], smooth[2], smooth[3]);
    snowtri_ctx_destroy(ctx);
    printf("c abi ok\n");
    return 0;
}

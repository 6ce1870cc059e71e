/*
 * snowtri_oracle.c -- CPU restatement of SnowMocap's multi-view triangulation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: it may be called from tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product path
 * (snowmocap_amd/ never imports, links or executes anything under oracle/).
 *
 * Parity status: PINNED.  Every function below is checked against outputs of the unmodified
 * reference (imported in the build container by tests/golden/make_golden.py) through the
 * committed fixtures under tests/golden/ -- see tests/test_oracle_golden.py.
 *
 * Plain C, IEEE fp64 throughout (the reference is NumPy float64: camera.py:41-44), compiled
 * with -ffp-contract=off so the operation order written here is the order executed.
 * Where the reference calls LAPACK (np.linalg.inv on a 2x2 / 3x3) the closed-form inverse is
 * used; the two agree to ~1e-13 relative (measured), far inside the parity budget.
 *
 * Reference citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double keypoint_score_threshold;   /* triangulation.py:50,73 */
    double average_score_threshold;    /* triangulation.py:50,80 */
    double distance_threshold;         /* triangulation.py:50,73 */
    double condense_distance_tol;      /* triangulation.py:96,125 */
    double condense_person_num_tol;    /* triangulation.py:97,133 */
    double condense_score_tol;         /* triangulation.py:98,151 */
    int32_t center_point_index;        /* triangulation.py:99,112,121 */
    int32_t keypoint_num;              /* triangulation.py:100,136-138 */
} orc_params;

enum { ORC_OK = 0, ORC_SINGULAR = 1, ORC_BAD_INDEX = 2, ORC_OVERFLOW = 3 };

/* ---- numpy's pairwise summation (np.sum / np.mean on a contiguous 1-D float64 array).
 * Restated from numpy's published algorithm (numpy/core/src/umath/loops_utils.h.src,
 * DOUBLE_pairwise_sum; numpy 2.2.6 installed here, 1.24.4 pinned by requirements.txt:2 use
 * the same scheme): <8 elements sequential, <=128 eight interleaved accumulators, larger
 * inputs split in two at a multiple of 8.  Verified bit-exact against np.sum for n = 1..299. */
static double np_pairwise_sum(const double *a, long n, long stride)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        long i;
        for (i = 0; i < 8; i++) r[i] = a[i * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[(i + k) * stride];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2, stride) + np_pairwise_sum(a + n2 * stride, n - n2, stride);
    }
}

/* ---- 3x3 inverse, closed form (camera.py:242 calls np.linalg.inv(K) per keypoint). */
static int inv3(const double *m, double *o)
{
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c01 = m[5] * m[6] - m[3] * m[8];
    double c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0) return ORC_SINGULAR;
    double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return ORC_OK;
}

/* ---- A1: CameraGroup.add_human_2D_points (camera.py:234-253).
 * For each keypoint: w = [u, v, 1]; f = inv(K) . w; f = R . f.  Rays are NOT normalised. */
int orc_rays_from_pixels(const double *K, const double *R, const double *uv /*[J][2]*/, int J,
                         double *rays /*[J][3]*/)
{
    double Ki[9];
    int rc = inv3(K, Ki);                                   /* camera.py:242 */
    if (rc) return rc;
    for (int j = 0; j < J; j++) {
        double w0 = uv[2 * j], w1 = uv[2 * j + 1], w2 = 1.0; /* camera.py:241 */
        double f0 = Ki[0] * w0 + Ki[1] * w1 + Ki[2] * w2;    /* camera.py:242 */
        double f1 = Ki[3] * w0 + Ki[4] * w1 + Ki[5] * w2;
        double f2 = Ki[6] * w0 + Ki[7] * w1 + Ki[8] * w2;
        rays[3 * j + 0] = R[0] * f0 + R[1] * f1 + R[2] * f2; /* camera.py:243 */
        rays[3 * j + 1] = R[3] * f0 + R[4] * f1 + R[5] * f2;
        rays[3 * j + 2] = R[6] * f0 + R[7] * f1 + R[8] * f2;
    }
    return ORC_OK;
}

/* ---- A2: Skew_Ray_Solver (triangulation.py:24-31).
 * H = [hm hs]; ne1 = inv(H^T H); ne2 = H^T (ts - tm); S = ne1 . ne2;
 * Wm = hm*S0 + tm; Ws = -hs*S1 + ts; returns ||Wm - Ws|| and (Wm + Ws)/2.
 * Exactly singular H^T H -> ORC_SINGULAR (the reference raises LinAlgError at :26). */
int orc_skew_ray_solver(const double *hm, const double *hs, const double *tm, const double *ts,
                        double *dist, double *W)
{
    double a = hm[0] * hm[0] + hm[1] * hm[1] + hm[2] * hm[2];   /* (H^T H)[0][0] */
    double b = hm[0] * hs[0] + hm[1] * hs[1] + hm[2] * hs[2];   /* (H^T H)[0][1] */
    double c = hs[0] * hs[0] + hs[1] * hs[1] + hs[2] * hs[2];   /* (H^T H)[1][1] */
    double det = a * c - b * b;
    if (det == 0.0) {
        *dist = NAN; W[0] = W[1] = W[2] = NAN;
        return ORC_SINGULAR;
    }
    double i00 = c / det, i01 = -b / det, i11 = a / det;        /* :26 */
    double d0 = ts[0] - tm[0], d1 = ts[1] - tm[1], d2 = ts[2] - tm[2];
    double e = hm[0] * d0 + hm[1] * d1 + hm[2] * d2;            /* :27 */
    double f = hs[0] * d0 + hs[1] * d1 + hs[2] * d2;
    double S0 = i00 * e + i01 * f;                              /* :28 */
    double S1 = i01 * e + i11 * f;
    double Wm[3], Ws[3], df[3];
    for (int k = 0; k < 3; k++) {
        Wm[k] = hm[k] * S0 + tm[k];                             /* :29 */
        Ws[k] = -hs[k] * S1 + ts[k];                            /* :30 */
        df[k] = Wm[k] - Ws[k];
        W[k] = (Wm[k] + Ws[k]) / 2.0;                           /* :31 */
    }
    *dist = sqrt(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
    return ORC_OK;
}

/* Batched A2 for the G5 unit fixture. */
int orc_skew_ray_solver_batch(long n, const double *hm, const double *hs, const double *tm,
                              const double *ts, double *dist, double *W, int32_t *singular)
{
    int any = 0;
    for (long i = 0; i < n; i++) {
        int rc = orc_skew_ray_solver(hm + 3 * i, hs + 3 * i, tm + 3 * i, ts + 3 * i, dist + i, W + 3 * i);
        if (singular) singular[i] = rc;
        any |= rc;
    }
    return any;
}

/* ---- A3: Human_Triangulation (triangulation.py:50-93) for ONE frame.
 *
 * rays  [C][Pmax][J][3], score [C][Pmax][J] (fp64 storage), n_persons[C], t[C][3].
 * score_is_f32: the caller handed float32 confidence arrays, so the reference's
 * (sm + ss) / 2 at :72 is evaluated in float32 (NumPy scalar arithmetic) before the
 * float64 division -- reproduced here.
 * Candidate enumeration order (:56-65): mc < sc camera pairs, then persons of mc, then of sc.
 * Per joint (:70-78): A2; score = ((sm+ss)/2)/(dist*1000); score = 0 when sm<kthr or ss<kthr or
 * dist>dthr; the 3D point is kept either way.  Candidate mean = np.mean(p_score) (:79), dropped
 * iff mean < average_score_threshold (:80-81).
 * Outputs hold the KEPT candidates in order: cand_xyz[n][J][3], cand_kscore[n][J],
 * cand_pscore[n]; *n_out = n.  cand_index (optional) gets the enumeration index of each.
 * Returns ORC_SINGULAR if any pair was exactly singular (reference raises), ORC_OVERFLOW if
 * more than max_cand would be kept. */
int orc_human_triangulation(int C, int Pmax, int J, const int32_t *n_persons,
                            const double *rays, const double *score, int score_is_f32,
                            const double *t, const orc_params *prm, int max_cand,
                            double *cand_xyz, double *cand_kscore, double *cand_pscore,
                            int32_t *cand_index, int32_t *n_out, int32_t *n_singular)
{
    int n = 0, enum_idx = 0, nsing = 0, rc_all = ORC_OK;
    double *p_xyz = (double *)malloc(sizeof(double) * (size_t)J * 4);
    double *p_score = p_xyz + (size_t)J * 3;
    for (int mc = 0; mc < C - 1; mc++) {
        const double *tm = t + 3 * mc;
        for (int sc = mc + 1; sc < C; sc++) {
            const double *ts = t + 3 * sc;
            for (int pm = 0; pm < n_persons[mc]; pm++) {
                for (int ps = 0; ps < n_persons[sc]; ps++, enum_idx++) {
                    const double *rm = rays + ((size_t)(mc * Pmax + pm) * J) * 3;
                    const double *rs = rays + ((size_t)(sc * Pmax + ps) * J) * 3;
                    const double *smv = score + (size_t)(mc * Pmax + pm) * J;
                    const double *ssv = score + (size_t)(sc * Pmax + ps) * J;
                    for (int j = 0; j < J; j++) {
                        double dist;
                        int rc = orc_skew_ray_solver(rm + 3 * j, rs + 3 * j, tm, ts, &dist, p_xyz + 3 * j);
                        if (rc) { nsing++; rc_all = ORC_SINGULAR; }
                        double sm = smv[j], ss = ssv[j], half;
                        if (score_is_f32) half = (double)(((float)sm + (float)ss) / 2.0f);
                        else half = (sm + ss) / 2.0;
                        double sc_j = half / (dist * 1000.0);                       /* :72 */
                        if (sm < prm->keypoint_score_threshold || ss < prm->keypoint_score_threshold ||
                            dist > prm->distance_threshold)
                            sc_j = 0.0;                                              /* :73-74 */
                        p_score[j] = sc_j;
                    }
                    double avg = np_pairwise_sum(p_score, J, 1) / (double)J;         /* :79 */
                    if (avg < prm->average_score_threshold) continue;                /* :80-81 */
                    if (n >= max_cand) { rc_all = ORC_OVERFLOW; goto done; }
                    memcpy(cand_xyz + (size_t)n * J * 3, p_xyz, sizeof(double) * (size_t)J * 3);
                    memcpy(cand_kscore + (size_t)n * J, p_score, sizeof(double) * (size_t)J);
                    cand_pscore[n] = avg;
                    if (cand_index) cand_index[n] = enum_idx;
                    n++;
                }
            }
        }
    }
done:
    free(p_xyz);
    *n_out = n;
    if (n_singular) *n_singular = nsing;
    return rc_all;
}

/* ---- A4: Human_Triangulation_Condense (triangulation.py:95-162) for ONE frame.
 *
 * cand_xyz[n][J][3], cand_kscore[n][J].  Greedy clustering in list order (:107-130): seeds
 * mc in range(n-1) (the last candidate never seeds; n == 1 gives no output), absorbed seeds are
 * skipped, every later un-absorbed candidate whose centre joint lies within
 * condense_distance_tol of the SEED's centre joint is absorbed (`dist > tol` -> skip, so NaN
 * distances absorb).  Clusters smaller than condense_person_num_tol are dropped (:132-134)
 * but their members stay absorbed.  Fusion per joint b < keypoint_num (:138-148):
 * sum = np.sum(scores); sum == 0 -> joint stays (0,0,0)/0; else weights s/sum,
 * xyz = sum_i w_i * W_i (row by row), score = sum / cluster_size.  Cluster mean over keypoint_num
 * (:150), dropped iff mean < condense_score_tol (:151-152).
 * Outputs: out_xyz[max_out][kn][3], out_kscore[max_out][kn], out_pscore[max_out]; *n_out is the
 * TRUE number of output persons (entries beyond max_out are not written, ORC_OVERFLOW returned).
 * member_count (optional, [max_out]) receives each output person's cluster size. */
int orc_condense(int n, int J, const double *cand_xyz, const double *cand_kscore,
                 const orc_params *prm, int max_out,
                 double *out_xyz, double *out_kscore, double *out_pscore,
                 int32_t *member_count, int32_t *n_out)
{
    int kn = prm->keypoint_num, ci = prm->center_point_index;
    *n_out = 0;
    if (n - 1 <= 0) return ORC_OK;                   /* range(person_num - 1) is empty */
    if (ci < 0) ci += J;                              /* Python negative indexing */
    if (ci < 0 || ci >= J || kn > J || kn < 0) return ORC_BAD_INDEX;   /* IndexError in the reference */
    unsigned char *absorbed = (unsigned char *)calloc((size_t)n, 1);
    int32_t *members = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    double *wts = (double *)malloc(sizeof(double) * (size_t)n);
    double *p_xyz = (double *)malloc(sizeof(double) * (size_t)(kn > 0 ? kn : 1) * 4);
    double *p_sc = p_xyz + (size_t)(kn > 0 ? kn : 1) * 3;
    int nout = 0, rc = ORC_OK;
    for (int mc = 0; mc < n - 1; mc++) {                                  /* :107 */
        if (absorbed[mc]) continue;                                       /* :108-109 */
        const double *mcen = cand_xyz + ((size_t)mc * J + ci) * 3;        /* :112 */
        int m = 0;
        members[m++] = mc;
        for (int sc = mc + 1; sc < n; sc++) {                             /* :116 */
            if (absorbed[sc]) continue;
            const double *scen = cand_xyz + ((size_t)sc * J + ci) * 3;    /* :121 */
            double d0 = mcen[0] - scen[0], d1 = mcen[1] - scen[1], d2 = mcen[2] - scen[2];
            double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);              /* :124 */
            if (dist > prm->condense_distance_tol) continue;              /* :125-126 */
            absorbed[sc] = 1;                                             /* :128 */
            members[m++] = sc;
        }
        if ((double)m < prm->condense_person_num_tol) continue;           /* :132-134 */
        for (int b = 0; b < kn; b++) {                                    /* :138 */
            p_xyz[3 * b] = p_xyz[3 * b + 1] = p_xyz[3 * b + 2] = 0.0;
            p_sc[b] = 0.0;
            for (int i = 0; i < m; i++) wts[i] = cand_kscore[(size_t)members[i] * J + b];
            double sum = np_pairwise_sum(wts, m, 1);                      /* :141 */
            if (sum == 0.0) continue;                                     /* :142-143 */
            double x = 0.0, y = 0.0, z = 0.0;
            for (int i = 0; i < m; i++) {
                double w = wts[i] / sum;                                  /* :144 */
                const double *W = cand_xyz + ((size_t)members[i] * J + b) * 3;
                if (i == 0) { x = W[0] * w; y = W[1] * w; z = W[2] * w; }  /* :145-147 axis-0 sum */
                else { x += W[0] * w; y += W[1] * w; z += W[2] * w; }
            }
            p_xyz[3 * b] = x; p_xyz[3 * b + 1] = y; p_xyz[3 * b + 2] = z;
            p_sc[b] = sum / (double)m;                                    /* :148 */
        }
        double avg = kn > 0 ? np_pairwise_sum(p_sc, kn, 1) / (double)kn : NAN;   /* :150 */
        if (avg < prm->condense_score_tol) continue;                      /* :151-152 */
        if (nout < max_out) {
            memcpy(out_xyz + (size_t)nout * kn * 3, p_xyz, sizeof(double) * (size_t)kn * 3);
            memcpy(out_kscore + (size_t)nout * kn, p_sc, sizeof(double) * (size_t)kn);
            out_pscore[nout] = avg;
            if (member_count) member_count[nout] = m;
        } else rc = ORC_OVERFLOW;
        nout++;
    }
    free(absorbed); free(members); free(wts); free(p_xyz);
    *n_out = nout;
    return rc;
}

/* ---- The per-frame caller sequence of main.py:50-71,106 over a batch:
 * add_human_2D_points x (C*P) -> Human_Triangulation -> Human_Triangulation_Condense.
 *
 * kpts [F][C][Pmax][J][3] = (u, v, score) as float (in_is_f32=1) or double; n_persons[F][C].
 * Outputs per frame: out_xyz[F][max_out][kn][3], out_kscore[F][max_out][kn],
 * out_pscore[F][max_out], out_count[F], status[F].  Frames run under OpenMP (they are
 * independent: SURVEY.md §8e); returns the number of threads used (>=1) or a negative errno. */
int orc_triangulate_condense_batch(long F, int C, int Pmax, int J,
                                   const double *K /*[C][9]*/, const double *R /*[C][9]*/,
                                   const double *t /*[C][3]*/,
                                   const void *kpts, int in_is_f32, const int32_t *n_persons,
                                   const orc_params *prm, int max_out,
                                   double *out_xyz, double *out_kscore, double *out_pscore,
                                   int32_t *out_count, int32_t *status, int nthreads)
{
    int used = 1;
    int kn = prm->keypoint_num;
    int maxc = 0;
    for (int a = 0; a < C; a++) for (int b = a + 1; b < C; b++) maxc += Pmax * Pmax;
    if (maxc < 1) maxc = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#endif
    #pragma omp parallel
    {
        size_t nray = (size_t)C * Pmax * J;
        double *rays = (double *)malloc(sizeof(double) * nray * 3);
        double *score = (double *)malloc(sizeof(double) * nray);
        double *uv = (double *)malloc(sizeof(double) * (size_t)J * 2);
        double *cxyz = (double *)malloc(sizeof(double) * (size_t)maxc * J * 3);
        double *cks = (double *)malloc(sizeof(double) * (size_t)maxc * J);
        double *cps = (double *)malloc(sizeof(double) * (size_t)maxc);
        #pragma omp for schedule(static)
        for (long f = 0; f < F; f++) {
            const int32_t *np_f = n_persons + (size_t)f * C;
            for (int c = 0; c < C; c++) {
                for (int p = 0; p < np_f[c]; p++) {
                    size_t base = (((size_t)f * C + c) * Pmax + p) * J;
                    for (int j = 0; j < J; j++) {
                        double u, v, s;
                        if (in_is_f32) {
                            const float *q = (const float *)kpts + (base + j) * 3;
                            u = q[0]; v = q[1]; s = q[2];
                        } else {
                            const double *q = (const double *)kpts + (base + j) * 3;
                            u = q[0]; v = q[1]; s = q[2];
                        }
                        uv[2 * j] = u; uv[2 * j + 1] = v;
                        score[(size_t)(c * Pmax + p) * J + j] = s;
                    }
                    orc_rays_from_pixels(K + 9 * c, R + 9 * c, uv, J,
                                         rays + ((size_t)(c * Pmax + p) * J) * 3);
                }
            }
            int32_t nc = 0, nsing = 0, no = 0;
            int rc = orc_human_triangulation(C, Pmax, J, np_f, rays, score, in_is_f32, t, prm, maxc,
                                             cxyz, cks, cps, NULL, &nc, &nsing);
            int rc2 = ORC_OK;
            if (rc != ORC_SINGULAR)
                rc2 = orc_condense(nc, J, cxyz, cks, prm, max_out,
                                   out_xyz + (size_t)f * max_out * kn * 3,
                                   out_kscore + (size_t)f * max_out * kn,
                                   out_pscore + (size_t)f * max_out, NULL, &no);
            out_count[f] = no;
            if (status) status[f] = rc ? rc : rc2;
        }
        free(rays); free(score); free(uv); free(cxyz); free(cks); free(cps);
    }
    return used;
}

/* ---- N1 (next row): SecondOrderDynamic.update (triangulation.py:15-22) applied to a track.
 * x[T][n] -> y[T][n]; frame 0 passes through and seeds xp=y=x0, yd=0 (:11-13; Smooth :180-181). */
void orc_second_order_track(long T, long n, const double *x, double f, double z, double r,
                            double dt, double *y_out)
{
    const double pi = 3.141592653589793;
    double k1 = z / (pi * f);                                    /* :7 */
    double k2 = 1.0 / ((2 * pi * f) * (2 * pi * f));             /* :8 */
    double k3 = r * z / (2 * pi * f);                            /* :9 */
    for (long i = 0; i < n; i++) {
        double xp = x[i], y = x[i], yd = 0.0;
        y_out[i] = x[i];
        for (long k = 1; k < T; k++) {
            double xk = x[k * n + i];
            double xd = (xk - xp) / dt;                          /* :17 */
            xp = xk;                                             /* :18 */
            y = y + dt * yd;                                     /* :20 */
            yd = yd + dt * (xk + k3 * xd - y - k1 * yd) / k2;    /* :21 */
            y_out[k * n + i] = y;
        }
    }
}

/* CPU ORACLE (test infrastructure, NOT product code) for TeraStitcher's MIP-NCC pairwise
 * tile registration.
 *
 * Plain-C restatement of the reference CPU algorithm in
 *   TeraStitcher/src/crossmips/libcrossmips.cpp   (norm_cross_corr_mips, :101-515)
 *   TeraStitcher/src/crossmips/compute_funcs.cu   (CPU branch; cited per function below)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the shipped library (libmi_ipp.so) never links or calls it.
 *
 * Pinned: tests/test_oracle_ncc.py checks every output (9 scalars, mutated wRangeThr,
 * MIPs, NCC maps) bit-for-bit against golden vectors produced by the compiled
 * reference (oracle/_ref, built from /root/reference by oracle/Makefile; generator
 * tests/golden/make_ncc_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  No FMA
 * contraction: float/double expression order below is part of the specification.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_TILE 32 /* TILE_SIDE, compute_funcs.h:66 */

typedef struct {
    int maxIter;        /* PDAlgoMIPNCC.cpp:82 */
    float maxThr;       /* :83 */
    float widthThr;     /* :93 */
    int wRangeThr_i, wRangeThr_j, wRangeThr_k; /* in-out: libcrossmips.cpp:275-277 */
    int minPoints, minDim_NCCsrc, minDim_NCCmap;
    float UNR_NCC;
    int INF_W, INV_COORD;
} orc_params;

typedef struct { /* NCC_descr_t, CrossMIPs.h:58-62 */
    int coord[3];
    float NCC_maxs[3];
    int NCC_widths[3];
} orc_descr;

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int pos(int a) { return a > 0 ? a : 0; } /* START_IND, my_defs.h:57 */

/* compute_3_MIPs, compute_funcs.cu:502-521.  A1/B are the overlap views; the same
 * strides apply to both (libcrossmips.cpp:299-314). MIPs start at 0 (:319-337). */
void orc_mips(const float *A1, const float *B, int dimi_v, int dimj_v, int dimk_v, int stridei, int stridek,
              float *xy1, float *xz1, float *yz1, float *xy2, float *xz2, float *yz2) {
    const float *p = A1, *q = B;
    memset(xy1, 0, sizeof(float) * dimi_v * dimj_v); memset(xy2, 0, sizeof(float) * dimi_v * dimj_v);
    memset(xz1, 0, sizeof(float) * dimi_v * dimk_v); memset(xz2, 0, sizeof(float) * dimi_v * dimk_v);
    memset(yz1, 0, sizeof(float) * dimj_v * dimk_v); memset(yz2, 0, sizeof(float) * dimj_v * dimk_v);
    for (int k = 0; k < dimk_v; k++, p += stridek, q += stridek)
        for (int i = 0; i < dimi_v; i++, p += stridei, q += stridei)
            for (int j = 0; j < dimj_v; j++, p++, q++) {
                float a = *p, b = *q;
                if (a > xy1[i * dimj_v + j]) xy1[i * dimj_v + j] = a;
                if (a > xz1[i * dimk_v + k]) xz1[i * dimk_v + k] = a;
                if (a > yz1[j * dimk_v + k]) yz1[j * dimk_v + k] = a;
                if (b > xy2[i * dimj_v + j]) xy2[i * dimj_v + j] = b;
                if (b > xz2[i * dimk_v + k]) xz2[i * dimk_v + k] = b;
                if (b > yz2[j * dimk_v + k]) yz2[j * dimk_v + k] = b;
            }
}

/* seq_cpu_compute_partial_sums, compute_funcs.cu:474-500: per 32x32 tile a FLOAT running sum,
 * rows then columns.  ps has (height/32) x (width/32) entries. */
void orc_tile_sums(const float *img, int height, int width, float *ps) {
    int nh = height - height % ORC_TILE, nw = width - width % ORC_TILE, pw = nw / ORC_TILE;
    for (int i = 0; i < nh; i += ORC_TILE)
        for (int j = 0; j < nw; j += ORC_TILE) {
            float s = 0.0f;
            for (int l = 0; l < ORC_TILE; l++)
                for (int k = 0; k < ORC_TILE; k++) s += img[(i + l) * width + (j + k)];
            ps[(i / ORC_TILE) * pw + j / ORC_TILE] = s;
        }
}

/* mean numerator of one MIP window, compute_funcs.cu:1186-1262 (tile path) */
static double window_sum_tiled(const float *mip, const float *ps, int dimv, int r0, int c0, int nr, int nc) {
    int pw = dimv / ORC_TILE;
    int su = r0 - r0 % ORC_TILE, sv = c0 - c0 % ORC_TILE;
    if (su != r0) su += ORC_TILE;
    if (sv != c0) sv += ORC_TILE;
    int eu = r0 + nr - (r0 + nr) % ORC_TILE, ev = c0 + nc - (c0 + nc) % ORC_TILE;
    double s = 0.0;
    for (int i = su; i < eu; i += ORC_TILE)
        for (int j = sv; j < ev; j += ORC_TILE) s += ps[(i / ORC_TILE) * pw + j / ORC_TILE];
    for (int i = r0; i < r0 + nr; i++) {
        int j = c0;
        while (j < c0 + nc) {
            if (j < sv || j >= ev || i < su || i >= eu) { s += mip[i * dimv + j]; j++; }
            else j = ev;
        }
    }
    return s;
}

/* compute_NCC, compute_funcs.cu:1163-1292 */
float orc_ncc(const float *m1, const float *m2, int dimu, int dimv, int u, int v, const float *ps1, const float *ps2) {
    int nr = dimu - abs(u), nc = dimv - abs(v);
    const float *im1 = m1 + pos(u * dimv) + pos(v);
    const float *im2 = m2 + pos(-u * dimv) + pos(-v);
    int stride = abs(v);
    double fm = 0.0, tm = 0.0;
    if (ps1 && ps2 && ORC_TILE <= dimu && ORC_TILE <= dimv) {
        fm = window_sum_tiled(m1, ps1, dimv, pos(u), pos(v), nr, nc);
        tm = window_sum_tiled(m2, ps2, dimv, pos(-u), pos(-v), nr, nc);
    } else {
        const float *p = im1, *q = im2;
        for (int i = 0; i < nr; i++, p += stride, q += stride)
            for (int j = 0; j < nc; j++, p++, q++) { fm += *p; tm += *q; }
    }
    fm /= (nr * nc);
    tm /= (nr * nc);
    double num = 0.0, f1 = 0.0, f2 = 0.0;
    const float *p = im1, *q = im2;
    for (int i = 0; i < nr; i++, p += stride, q += stride)
        for (int j = 0; j < nc; j++, p++, q++) {
            double fp = *p - fm, tp = *q - tm;
            num += *p * tp;
            f1 += fp * fp;
            f2 += tp * tp;
        }
    return (float)(num / sqrt(f1 * f2));
}

/* compute_NCC_map CPU branch, compute_funcs.cu:1026-1085 */
void orc_ncc_map(float *map, const float *m1, const float *m2, int dimu, int dimv, int du, int dv) {
    int ph = dimu / ORC_TILE, pw = dimv / ORC_TILE;
    float *ps1 = NULL, *ps2 = NULL;
    if (ph * pw > 0) {
        ps1 = (float *)malloc(sizeof(float) * ph * pw);
        ps2 = (float *)malloc(sizeof(float) * ph * pw);
        orc_tile_sums(m1, dimu, dimv, ps1);
        orc_tile_sums(m2, dimu, dimv, ps2);
    }
    for (int u = -du; u <= du; u++)
        for (int v = -dv; v <= dv; v++)
            map[(u + du) * (2 * dv + 1) + (v + dv)] = orc_ncc(m1, m2, dimu, dimv, u, v, ps1, ps2);
    free(ps1); free(ps2);
}

/* compute_MAX_ind, compute_funcs.cu:1294-1305: first strict maximum; a leading NaN stays */
int orc_argmax(const float *v, int len) {
    float best = v[0];
    int ind = 0;
    for (int i = 0; i < len; i++)
        if (v[i] > best) { best = v[i]; ind = i; }
    return ind;
}

/* compute_Neighborhood, compute_funcs.cu:1324-1592.  Returns -1 on the reference's exceptions. */
int orc_neighborhood(const orc_params *P, const float *NCC, int delayu, int delayv, int newu, int newv, int ind_max,
                     const float *m1, const float *m2, int dimu, int dimv, float *win, int *du, int *dv, int *failed) {
    int ph = dimu / ORC_TILE, pw = dimv / ORC_TILE;
    float *ps1 = NULL, *ps2 = NULL;
    if (ph * pw > 0) {
        ps1 = (float *)malloc(sizeof(float) * ph * pw);
        ps2 = (float *)malloc(sizeof(float) * ph * pw);
        orc_tile_sums(m1, dimu, dimv, ps1);
        orc_tile_sums(m2, dimu, dimv, ps2);
    }
    int H = 2 * newu + 1, W = 2 * newv + 1, Wm = 2 * delayv + 1;
    int initu = imin(imax(0, ind_max / Wm - newu), 2 * (delayu - newu));
    int initv = imin(imax(0, ind_max % Wm - newv), 2 * (delayv - newv));
    if (initu * Wm + initv < 0) { free(ps1); free(ps2); return -1; }
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) win[r * W + c] = NCC[(initu + r) * Wm + initv + c];
    *du = initu - delayu + newu;
    *dv = initv - delayv + newv;
    int pr = ind_max / Wm - initu, pc = ind_max % Wm - initv; /* peak position inside the window */
    ind_max = W * pr + pc;
    int ind_ref = W * newu + newv;
    float *tmp = (float *)malloc(sizeof(float) * H * W);
    int it = 0;
    while (it < P->maxIter && ind_max != ind_ref) {
        int deltau = ind_max / W - newu, deltav = ind_max % W - newv;
        /* the four overlapping-copy branches (:1411-1451) are one shift by (deltau, deltav) */
        memcpy(tmp, win, sizeof(float) * H * W);
        *du += deltau;
        *dv += deltav;
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int sr = r + deltau, sc = c + deltav;
                if (sr >= 0 && sr < H && sc >= 0 && sc < W) win[r * W + c] = tmp[sr * W + sc];
                else win[r * W + c] = orc_ncc(m1, m2, dimu, dimv, r - newu + *du, c - newv + *dv, ps1, ps2);
            }
        ind_max = orc_argmax(win, H * W);
        it++;
    }
    if (ind_ref != ind_max) {
        *du += ind_max / W - newu;
        *dv += ind_max % W - newv;
        *failed = 1;
    }
    free(tmp); free(ps1); free(ps2);
    return 0;
}

/* one direction of compute_NCC_width (compute_funcs.cu:160-282); step = 1 (horizontal) or row
 * length (vertical); range = wRangeThr of this direction; range2 = the bound used by the
 * "second chance" loops, which the reference takes from the HORIZONTAL range in both cases
 * (:252,267).  Where that bound exceeds the window (range2 > range) the reference reads outside
 * the array (undefined behaviour); oracle and product clamp the bound to the window there. */
static int peak_width(const orc_params *P, const float *M, int ind, int step, int range, int range2) {
    if (range < P->minDim_NCCmap) return P->INF_W;
    if (range2 > range) range2 = range;
    float thr = P->widthThr * M[ind];
    int found = 0, w = 1;
    while (w <= range && !found) { if (M[ind - w * step] <= thr) found = 1; else w++; }
    found = 0;
    while (w <= range && !found) { if (M[ind + w * step] <= thr) found = 1; else w++; }
    if (found) return w;
    float prec = M[ind - P->minPoints * step];
    int dist = P->minPoints + 1;
    while (dist <= range2 && !found) {
        if (M[ind - dist * step] >= prec) found = 1;
        else { prec = M[ind - dist * step]; dist++; }
    }
    if (dist < 2 * P->minPoints) w = P->INF_W;
    else w = (int)floorf((dist - 1) * (M[ind] - thr) / (M[ind] - prec));
    found = 0;
    prec = M[ind + P->minPoints * step];
    dist = P->minPoints + 1;
    while (dist <= range2 && !found) {
        if (M[ind + dist * step] >= prec) found = 1;
        else { prec = M[ind + dist * step]; dist++; }
    }
    if (dist < 2 * P->minPoints) w = P->INF_W;
    else w = imin(imax(w, (int)floorf((dist - 1) * (M[ind] - thr) / (M[ind] - prec))), P->INF_W - 1);
    return w;
}

void orc_widths(const orc_params *P, const float *M, int rowlen, int ind, int range1, int range2, int failed,
                int *w1, int *w2) {
    if (failed) { *w1 = *w2 = P->INF_W; return; }
    *w2 = peak_width(P, M, ind, 1, range2, range2);
    *w1 = peak_width(P, M, ind, rowlen, range1, range2);
}

/* compute_NCC_alignment, compute_funcs.cu:297-342 */
static void align_axis(const orc_params *P, orc_descr *R, int ax, int d1, float p1, int w1, int d2, float p2, int w2) {
    if (w1 == 1) w1 = P->INF_W;
    if (w2 == 1) w2 = P->INF_W;
    int ok1 = (p1 >= P->maxThr && w1 < P->INF_W), ok2 = (p2 >= P->maxThr && w2 < P->INF_W);
    if (ok1 && ok2) {
        if (abs(d1 - d2) < imin(w1, w2)) {
            R->coord[ax] = (int)floor((p1 * d1 + p2 * d2) / (p1 + p2) + 0.5);
            R->NCC_maxs[ax] = (p1 * p1 + p2 * p2) / (p1 + p2);
            R->NCC_widths[ax] = imax(w1, w2);
        } else if (p1 / w1 > p2 / w2) { R->coord[ax] = d1; R->NCC_maxs[ax] = p1; R->NCC_widths[ax] = w1; }
        else { R->coord[ax] = d2; R->NCC_maxs[ax] = p2; R->NCC_widths[ax] = w2; }
    } else if (ok1) { R->coord[ax] = d1; R->NCC_maxs[ax] = p1; R->NCC_widths[ax] = w1; }
    else if (ok2) { R->coord[ax] = d2; R->NCC_maxs[ax] = p2; R->NCC_widths[ax] = w2; }
    else { R->coord[ax] = P->INV_COORD; R->NCC_maxs[ax] = P->UNR_NCC; R->NCC_widths[ax] = P->INF_W; }
}

/* norm_cross_corr_mips, libcrossmips.cpp:101-515.  Returns 0, or a negative code where the
 * reference throws (-1 wRangeThr > delay :212-219, -2 bad side :316, -3 neighbourhood).
 * dbg (optional): [0..5] MIPs xy1,xz1,yz1,xy2,xz2,yz2 ; [6..8] full NCC maps xy,xz,yz ;
 * caller-allocated, any entry may be NULL.  delays_out (optional): clamped delayi,j,k. */
int orc_norm_cross_corr_mips(const float *A, const float *B, int dimk, int dimi, int dimj, int nk, int ni, int nj,
                             int delayk, int delayi, int delayj, int side, orc_params *P, orc_descr *R,
                             float **dbg, int *delays_out) {
    if (P->wRangeThr_i > delayi || P->wRangeThr_j > delayj || P->wRangeThr_k > delayk) return -1;
    delayi = imin(delayi, imax(0, dimi - ni - P->minDim_NCCsrc));
    delayj = imin(delayj, imax(0, dimj - nj - P->minDim_NCCsrc));
    delayk = imin(delayk, imax(0, dimk - nk - P->minDim_NCCsrc));
    P->wRangeThr_i = imin(P->wRangeThr_i, delayi);
    P->wRangeThr_j = imin(P->wRangeThr_j, delayj);
    P->wRangeThr_k = imin(P->wRangeThr_k, delayk);
    if (delays_out) { delays_out[0] = delayi; delays_out[1] = delayj; delays_out[2] = delayk; }

    int dimk_v = dimk, dimi_v, dimj_v, stridei, stridek;
    const float *A1;
    if (side == 0) { dimi_v = dimi - ni; dimj_v = dimj; stridei = 0; stridek = ni * dimj; A1 = A + stridek; }
    else if (side == 1) { dimi_v = dimi; dimj_v = dimj - nj; stridei = nj; stridek = 0; A1 = A + stridei; }
    else return -2;

    float *mip[6];
    int msz[3] = { dimi_v * dimj_v, dimi_v * dimk_v, dimj_v * dimk_v };
    for (int m = 0; m < 6; m++) mip[m] = (float *)malloc(sizeof(float) * (msz[m % 3] > 0 ? msz[m % 3] : 1));
    orc_mips(A1, B, dimi_v, dimj_v, dimk_v, stridei, stridek, mip[0], mip[1], mip[2], mip[3], mip[4], mip[5]);

    int mu[3] = { dimi_v, dimi_v, dimj_v }, mv[3] = { dimj_v, dimk_v, dimk_v };
    int du_[3] = { delayi, delayi, delayj }, dv_[3] = { delayj, delayk, delayk };
    int wu[3] = { P->wRangeThr_i, P->wRangeThr_i, P->wRangeThr_j }, wv[3] = { P->wRangeThr_j, P->wRangeThr_k, P->wRangeThr_k };
    float *map[3], *win[3];
    int d_u[3], d_v[3], failed[3] = { 0, 0, 0 }, rc = 0;
    for (int m = 0; m < 3; m++) {
        int len = (2 * du_[m] + 1) * (2 * dv_[m] + 1);
        map[m] = (float *)calloc(len, sizeof(float));
        win[m] = (float *)calloc((2 * wu[m] + 1) * (2 * wv[m] + 1), sizeof(float));
        orc_ncc_map(map[m], mip[m], mip[m + 3], mu[m], mv[m], du_[m], dv_[m]);
    }
    for (int m = 0; m < 3 && rc == 0; m++) {
        int len = (2 * du_[m] + 1) * (2 * dv_[m] + 1);
        int ind = orc_argmax(map[m], len);
        if (orc_neighborhood(P, map[m], du_[m], dv_[m], wu[m], wv[m], ind, mip[m], mip[m + 3], mu[m], mv[m],
                             win[m], &d_u[m], &d_v[m], &failed[m]) != 0) rc = -3;
    }
    if (rc == 0) {
        /* compute_Alignment, compute_funcs.cu:1597-1609 */
        int w1x, w1y, w2x, w1z, w2y, w2z;
        int cxy = wu[0] * (2 * wv[0] + 1) + wv[0], cxz = wu[1] * (2 * wv[1] + 1) + wv[1], cyz = wu[2] * (2 * wv[2] + 1) + wv[2];
        orc_widths(P, win[0], 2 * wv[0] + 1, cxy, wu[0], wv[0], failed[0], &w1x, &w1y);
        orc_widths(P, win[1], 2 * wv[1] + 1, cxz, wu[1], wv[1], failed[1], &w2x, &w1z);
        orc_widths(P, win[2], 2 * wv[2] + 1, cyz, wu[2], wv[2], failed[2], &w2y, &w2z);
        align_axis(P, R, 0, d_u[0], win[0][cxy], w1x, d_u[1], win[1][cxz], w2x);
        align_axis(P, R, 1, d_v[0], win[0][cxy], w1y, d_u[2], win[2][cyz], w2y);
        align_axis(P, R, 2, d_v[1], win[1][cxz], w1z, d_v[2], win[2][cyz], w2z);
        if (side == 0) R->coord[0] += ni; else R->coord[1] += nj;
    }
    if (dbg) {
        for (int m = 0; m < 6; m++) if (dbg[m]) memcpy(dbg[m], mip[m], sizeof(float) * msz[m % 3]);
        for (int m = 0; m < 3; m++) if (dbg[6 + m]) memcpy(dbg[6 + m], map[m], sizeof(float) * (2 * du_[m] + 1) * (2 * dv_[m] + 1));
    }
    for (int m = 0; m < 6; m++) free(mip[m]);
    for (int m = 0; m < 3; m++) { free(map[m]); free(win[m]); }
    return rc;
}

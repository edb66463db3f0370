// C-ABI shim around the UNMODIFIED reference crossmips sources (test infrastructure only).
// Compiled by oracle/Makefile together with /root/reference/TeraStitcher/src/crossmips/
// {libcrossmips.cpp,compute_funcs.cpp} where they lie (never copied into this repo) into
// oracle/_ref/libcrossmips_ref.so.  Used to pin oracle/ncc_oracle.c, to generate
// tests/golden/ncc_*.npz (tests/golden/make_ncc_golden.py) and, when present, as the
// "reference" CPU baseline of bench.py.
#include "CrossMIPs.h"
#include "compute_funcs.h"

extern "C" {

struct ref_params {  // field order == oracle/ncc_oracle.c: orc_params
    int maxIter; float maxThr; float widthThr;
    int wRangeThr_i, wRangeThr_j, wRangeThr_k;
    int minPoints, minDim_NCCsrc, minDim_NCCmap;
    float UNR_NCC; int INF_W, INV_COORD;
};

static void to_ref(const ref_params *p, NCC_parms_t *q) {
    q->enhance = false; q->maxIter = p->maxIter; q->maxThr = p->maxThr; q->widthThr = p->widthThr;
    q->wRangeThr_i = p->wRangeThr_i; q->wRangeThr_j = p->wRangeThr_j; q->wRangeThr_k = p->wRangeThr_k;
    q->minPoints = p->minPoints; q->minDim_NCCsrc = p->minDim_NCCsrc; q->minDim_NCCmap = p->minDim_NCCmap;
    q->UNR_NCC = p->UNR_NCC; q->INF_W = p->INF_W; q->INV_COORD = p->INV_COORD;
    q->n_transforms = 0; q->percents = 0; q->c = 0;
}

// returns 0, or -1 if the reference threw
int ref_norm_cross_corr_mips(float *A, float *B, int dimk, int dimi, int dimj, int nk, int ni, int nj,
                             int delayk, int delayi, int delayj, int side, ref_params *p,
                             int *coord, float *maxs, int *widths) {
    NCC_parms_t q; to_ref(p, &q);
    try {
        NCC_descr_t *d = norm_cross_corr_mips(A, B, dimk, dimi, dimj, nk, ni, nj, delayk, delayi, delayj, side, &q);
        for (int a = 0; a < 3; a++) { coord[a] = d->coord[a]; maxs[a] = d->NCC_maxs[a]; widths[a] = d->NCC_widths[a]; }
        delete d;
    } catch (...) { return -1; }
    p->wRangeThr_i = q.wRangeThr_i; p->wRangeThr_j = q.wRangeThr_j; p->wRangeThr_k = q.wRangeThr_k;
    return 0;
}

void ref_compute_3_MIPs(float *A1, float *B, float *xy1, float *xz1, float *yz1, float *xy2, float *xz2, float *yz2,
                        int dimi_v, int dimj_v, int dimk_v, int stridei, int stridek) {
    compute_3_MIPs(A1, B, xy1, xz1, yz1, xy2, xz2, yz2, dimi_v, dimj_v, dimk_v, stridei, stridek);
}

void ref_compute_NCC_map(float *map, float *m1, float *m2, int dimu, int dimv, int delayu, int delayv) {
    compute_NCC_map(map, m1, m2, dimu, dimv, delayu, delayv);
}

int ref_compute_MAX_ind(float *v, int len) { return compute_MAX_ind(v, len); }

}  // extern "C"

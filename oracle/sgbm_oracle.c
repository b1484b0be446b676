/*
 * sgbm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See sgbm_oracle.h for scope,
 * the upstream files restated and the "PARITY UNPINNED" statement.
 *
 * Every function cites the reference call site it stands behind (/root/reference/main.ipynb:N)
 * and the section of SURVEY.md Appendix A/B that specifies its arithmetic.
 *
 * All cost arithmetic is carried in `int`; the largest value that upstream would have held
 * in an int16 lane is tracked, and `headroom_ok` reports whether exact == int16 arithmetic.
 */
#include "sgbm_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <math.h>

#define MAX_COST 32767
#define DISP_SHIFT 4
#define DISP_SCALE 16

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---- parameter normalisation: SURVEY.md A.1 (behind main.ipynb:655-666) ---------------- */
typedef struct {
    int H, W;
    int minD, maxD, D;
    int minX1, maxX1, W1;
    int SW2, SH2;
    int P1, P2;
    int uniq, d12;
    int ftzero;
    int invalid_scaled;
    int npasses;
} geom_t;

static void make_geom(const oracle_sgbm_params *p, int H, int W, geom_t *g)
{
    int dim = p->blockSize > 0 ? p->blockSize : 5;
    g->H = H;
    g->W = W;
    g->minD = p->minDisparity;
    g->D = p->numDisparities;
    g->maxD = g->minD + g->D;
    g->minX1 = imax(g->maxD, 0);
    g->maxX1 = W + imin(g->minD, 0);
    g->W1 = g->maxX1 - g->minX1;
    g->SW2 = g->SH2 = dim / 2;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    g->P2 = imax(p->P2 > 0 ? p->P2 : 5, g->P1 + 1);
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g->ftzero = imax(p->preFilterCap, 15) | 1;
    g->invalid_scaled = (g->minD - 1) * DISP_SCALE;
    g->npasses = (p->mode == 1) ? 2 : 1;
}

void oracle_sgbm_geometry(const oracle_sgbm_params *p, int W, int *minX1, int *W1)
{
    geom_t g;
    make_geom(p, 1, W, &g);
    if (minX1) *minX1 = g.minX1;
    if (W1) *W1 = g.W1;
}

/* ---- per-row prefilter + half-pixel interval: SURVEY.md A.2, A.3 ------------------------ *
 * For one image row produce, for the two "channels" (x-gradient prefilter, raw intensity):
 *   val[c][x], lo[c][x] = min(val, (val+left)/2, (val+right)/2), hi[c][x] = max(...)
 * Border columns 0 and W-1 hold ftzero in BOTH channels (upstream quirk).               */
typedef struct {
    uint8_t *val[2], *lo[2], *hi[2];
} rowfeat_t;

static void row_features(const uint8_t *img, int y, int H, int W, int64_t stride, int ftzero,
                         rowfeat_t *f)
{
    const uint8_t *row = img + (int64_t)y * stride;
    const uint8_t *up = y > 0 ? row - stride : row;
    const uint8_t *dn = y < H - 1 ? row + stride : row;
    for (int c = 0; c < 2; c++) {
        f->val[c][0] = (uint8_t)ftzero;
        f->val[c][W - 1] = (uint8_t)ftzero;
    }
    for (int x = 1; x < W - 1; x++) {
        int g = (row[x + 1] - row[x - 1]) * 2 + (up[x + 1] - up[x - 1]) + (dn[x + 1] - dn[x - 1]);
        g = imin(imax(g, -ftzero), ftzero) + ftzero;
        f->val[0][x] = (uint8_t)g;
        f->val[1][x] = row[x];
    }
    for (int c = 0; c < 2; c++) {
        const uint8_t *v = f->val[c];
        for (int x = 0; x < W; x++) {
            int a = v[x];
            int l = x > 0 ? (a + v[x - 1]) / 2 : a;
            int r = x < W - 1 ? (a + v[x + 1]) / 2 : a;
            f->lo[c][x] = (uint8_t)imin(a, imin(l, r));
            f->hi[c][x] = (uint8_t)imax(a, imax(l, r));
        }
    }
}

/* Birchfield-Tomasi pixel cost of image row y for every valid column and disparity:
 * pix[(x-minX1)*D + (d-minD)]  (upstream calcPixelCostBT).                */
static void bt_row(const geom_t *g, const rowfeat_t *L, const rowfeat_t *R, int16_t *pix)
{
    const int D = g->D;
    for (int x = g->minX1; x < g->maxX1; x++) {
        int16_t *out = pix + (int64_t)(x - g->minX1) * D;
        for (int d = 0; d < D; d++) out[d] = 0;
        for (int c = 0; c < 2; c++) {
            const int sh = c == 0 ? 0 : 2;
            const int u = L->val[c][x], u0 = L->lo[c][x], u1 = L->hi[c][x];
            /* right column xr = x - (minD + d) descends as d ascends */
            const uint8_t *v = R->val[c], *v0 = R->lo[c], *v1 = R->hi[c];
            const int xr0 = x - g->minD;
            for (int d = 0; d < D; d++) {
                int xr = xr0 - d;
                int c0 = imax(0, imax(u - v1[xr], v0[xr] - u));
                int c1 = imax(0, imax(v[xr] - u1, u0 - v[xr]));
                out[d] = (int16_t)(out[d] + (imin(c0, c1) >> sh));
            }
        }
    }
}

/* horizontal box sum in the valid-column domain with clamped window: SURVEY.md A.4 */
static void hsum_row(const geom_t *g, const int16_t *pix, int16_t *hs)
{
    const int D = g->D, W1 = g->W1, SW2 = g->SW2;
    for (int d = 0; d < D; d++) {
        int s = pix[d] * (SW2 + 1);
        for (int i = 1; i <= SW2; i++) s += pix[(int64_t)imin(i, W1 - 1) * D + d];
        hs[d] = (int16_t)s;
    }
    for (int x = 1; x < W1; x++) {
        const int16_t *add = pix + (int64_t)imin(x + SW2, W1 - 1) * D;
        const int16_t *sub = pix + (int64_t)imax(x - SW2 - 1, 0) * D;
        const int16_t *prev = hs + (int64_t)(x - 1) * D;
        int16_t *cur = hs + (int64_t)x * D;
        for (int d = 0; d < D; d++) cur[d] = (int16_t)(prev[d] + add[d] - sub[d]);
    }
}

/* ---- 3x3 median, replicate border: SURVEY.md A.7 (inside .compute, main.ipynb:668) ------ */
static inline int16_t med9(int16_t *v)
{
    /* plain insertion sort of 9; order statistics need no particular network */
    for (int i = 1; i < 9; i++) {
        int16_t k = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > k) {
            v[j + 1] = v[j];
            j--;
        }
        v[j + 1] = k;
    }
    return v[4];
}

void oracle_median3x3_i16(const int16_t *src, int16_t *dst, int H, int W)
{
    for (int y = 0; y < H; y++) {
        const int16_t *r0 = src + (int64_t)imax(y - 1, 0) * W;
        const int16_t *r1 = src + (int64_t)y * W;
        const int16_t *r2 = src + (int64_t)imin(y + 1, H - 1) * W;
        for (int x = 0; x < W; x++) {
            int xl = imax(x - 1, 0), xr = imin(x + 1, W - 1);
            int16_t v[9] = {r0[xl], r0[x], r0[xr], r1[xl], r1[x], r1[xr], r2[xl], r2[x], r2[xr]};
            dst[(int64_t)y * W + x] = med9(v);
        }
    }
}

/* ---- speckle filter: SURVEY.md A.8 (inside .compute, main.ipynb:664-665,668) ------------ *
 * 4-connected components of pixels != newVal linked when |a-b| <= maxDiff; components of
 * at most maxSpeckleSize pixels are overwritten with newVal.  Row-major seeds, stack flood. */
void oracle_filter_speckles_i16(int16_t *img, int H, int W, int newVal, int maxSpeckleSize,
                                int maxDiff)
{
    int64_t n = (int64_t)H * W;
    int32_t *label = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    uint8_t *small = (uint8_t *)malloc((size_t)n + 1);
    int32_t cur = 0;
    for (int64_t i = 0; i < n; i++) {
        if (img[i] == newVal) continue;
        if (label[i]) {
            if (small[label[i]]) img[i] = (int16_t)newVal;
            continue;
        }
        cur++;
        label[i] = cur;
        int64_t sp = 0;
        int64_t p = i;
        int count = 0;
        for (;;) {
            count++;
            int py = (int)(p / W), px = (int)(p % W);
            int dp = img[p];
            if (py < H - 1 && !label[p + W] && img[p + W] != newVal && abs(dp - img[p + W]) <= maxDiff) {
                label[p + W] = cur;
                stack[sp++] = (int32_t)(p + W);
            }
            if (py > 0 && !label[p - W] && img[p - W] != newVal && abs(dp - img[p - W]) <= maxDiff) {
                label[p - W] = cur;
                stack[sp++] = (int32_t)(p - W);
            }
            if (px < W - 1 && !label[p + 1] && img[p + 1] != newVal && abs(dp - img[p + 1]) <= maxDiff) {
                label[p + 1] = cur;
                stack[sp++] = (int32_t)(p + 1);
            }
            if (px > 0 && !label[p - 1] && img[p - 1] != newVal && abs(dp - img[p - 1]) <= maxDiff) {
                label[p - 1] = cur;
                stack[sp++] = (int32_t)(p - 1);
            }
            if (sp == 0) break;
            p = stack[--sp];
        }
        if (count <= maxSpeckleSize) {
            small[cur] = 1;
            img[i] = (int16_t)newVal;
        } else {
            small[cur] = 0;
        }
    }
    free(label);
    free(stack);
    free(small);
}

/* ---- path aggregation state for one image row -------------------------------------------- *
 * L[dir][(x+1)*(D+2) + d+1] with sentinel slots d=-1 and d=D fixed at MAX_COST and border   *
 * columns x=-1, x=W1 fixed at 0; M[dir][x+1] = min_d L.  SURVEY.md A.5.                      */
typedef struct {
    int16_t *L[4];
    int16_t *M[4];
} lrow_t;

static void lrow_alloc(lrow_t *r, int W1, int D)
{
    for (int k = 0; k < 4; k++) {
        r->L[k] = (int16_t *)malloc((size_t)(W1 + 2) * (D + 2) * sizeof(int16_t));
        r->M[k] = (int16_t *)malloc((size_t)(W1 + 2) * sizeof(int16_t));
    }
}
static void lrow_free(lrow_t *r)
{
    for (int k = 0; k < 4; k++) {
        free(r->L[k]);
        free(r->M[k]);
    }
}
static void lrow_clear(lrow_t *r, int W1, int D)
{
    for (int k = 0; k < 4; k++) {
        for (int x = 0; x < W1 + 2; x++) {
            int16_t *v = r->L[k] + (int64_t)x * (D + 2);
            v[0] = MAX_COST;
            for (int d = 1; d <= D; d++) v[d] = 0;
            v[D + 1] = MAX_COST;
            r->M[k][x] = 0;
        }
    }
}

/* one step of the recurrence  L(p,d) = C(p,d) + min(Lq[d], Lq[d-1]+P1, Lq[d+1]+P1, mq+P2) - mq
 * acc[d] += L ; returns min_d L.  Lq points at d=0 of the predecessor (sentinels at -1, D). */
static inline int path_step(const int16_t *Cp, const int16_t *Lq, int mq, int P1, int P2, int D,
                            int16_t *Lout, int32_t *acc, int32_t *max_delta)
{
    const int delta = mq + P2;
    int mn = MAX_COST;
    if (delta > *max_delta) *max_delta = delta;
    for (int d = 0; d < D; d++) {
        int t = imin(imin((int)Lq[d - 1], (int)Lq[d + 1]) + P1, imin((int)Lq[d], delta));
        int L = Cp[d] + t - mq;
        Lout[d] = (int16_t)L;
        acc[d] += L;
        mn = imin(mn, L);
    }
    /* min_d L_r(p, .) + P2 is the next step's delta; recorded for every pixel (also the last of a
     * line, where no successor forms it) so that the record does not depend on the scan order */
    if (mn + P2 > *max_delta) *max_delta = mn + P2;
    return mn;
}

/* ---- the matcher: upstream computeDisparitySGBM (behind main.ipynb:668) ------------------ */
static int sgbm_core(const geom_t *g, const uint8_t *left, const uint8_t *right, int64_t stride,
                     int16_t *disp, oracle_sgbm_taps *taps)
{
    const int H = g->H, W = g->W, D = g->D, W1 = g->W1, SH2 = g->SH2;
    const int P1 = g->P1, P2 = g->P2;
    const int64_t rowsz = (int64_t)W1 * D;
    const int full = (g->npasses == 2);
    int32_t max_cp2 = 0, max_delta = 0;

    /* volumes: rolling single row unless two passes (or a tap) need every row */
    int16_t *Cfull = (taps && taps->C) ? taps->C : NULL;
    int16_t *Sfull = (taps && taps->S) ? taps->S : NULL;
    int own_C = 0, own_S = 0;
    if (full && !Cfull) {
        Cfull = (int16_t *)malloc((size_t)rowsz * H * sizeof(int16_t));
        own_C = 1;
    }
    if (full && !Sfull) {
        Sfull = (int16_t *)malloc((size_t)rowsz * H * sizeof(int16_t));
        own_S = 1;
    }
    int16_t *Crow = (int16_t *)calloc((size_t)rowsz, sizeof(int16_t));
    int16_t *Srow = (int16_t *)calloc((size_t)rowsz, sizeof(int16_t));
    int16_t *pix = (int16_t *)malloc((size_t)rowsz * sizeof(int16_t));
    const int nring = 2 * SH2 + 2;
    int16_t *hring = (int16_t *)malloc((size_t)rowsz * nring * sizeof(int16_t));
    int32_t *acc = (int32_t *)malloc((size_t)D * sizeof(int32_t));
    int16_t *disp2 = (int16_t *)malloc((size_t)W * sizeof(int16_t));
    int16_t *disp2cost = (int16_t *)malloc((size_t)W * sizeof(int16_t));
    int16_t *L0tmp = (int16_t *)malloc((size_t)(W1 + 2) * (D + 2) * sizeof(int16_t));
    int16_t *M0tmp = (int16_t *)malloc((size_t)(W1 + 2) * sizeof(int16_t));

    rowfeat_t fL, fR;
    uint8_t *featbuf = (uint8_t *)malloc((size_t)W * 12);
    for (int c = 0; c < 2; c++) {
        fL.val[c] = featbuf + (size_t)W * (c * 3 + 0);
        fL.lo[c] = featbuf + (size_t)W * (c * 3 + 1);
        fL.hi[c] = featbuf + (size_t)W * (c * 3 + 2);
        fR.val[c] = featbuf + (size_t)W * (6 + c * 3 + 0);
        fR.lo[c] = featbuf + (size_t)W * (6 + c * 3 + 1);
        fR.hi[c] = featbuf + (size_t)W * (6 + c * 3 + 2);
    }

    lrow_t rows[2];
    lrow_alloc(&rows[0], W1, D);
    lrow_alloc(&rows[1], W1, D);

#define HS(r) (hring + (int64_t)((r) % nring) * rowsz)

    for (int pass = 1; pass <= g->npasses; pass++) {
        const int y1 = pass == 1 ? 0 : H - 1, y2 = pass == 1 ? H : -1, dy = pass == 1 ? 1 : -1;
        const int x1 = pass == 1 ? 0 : W1 - 1, x2 = pass == 1 ? W1 : -1, dx = pass == 1 ? 1 : -1;
        int id = 0;
        lrow_clear(&rows[0], W1, D);
        lrow_clear(&rows[1], W1, D);

        for (int y = y1; y != y2; y += dy) {
            int16_t *C = Cfull ? Cfull + (int64_t)y * rowsz : Crow;
            int16_t *S = Sfull ? Sfull + (int64_t)y * rowsz : Srow;
            int16_t *drow = disp + (int64_t)y * W;

            if (pass == 1) {
                /* block cost for row y, vertical running sum over hsum rows */
                if (y == 0) {
                    memset(C, 0, (size_t)rowsz * sizeof(int16_t));
                    for (int k = 0; k <= SH2; k++) {
                        int r = imin(k, H - 1);
                        if (k < H) {
                            row_features(left, k, H, W, stride, g->ftzero, &fL);
                            row_features(right, k, H, W, stride, g->ftzero, &fR);
                            bt_row(g, &fL, &fR, pix);
                            hsum_row(g, pix, HS(r));
                        }
                        const int16_t *ha = HS(r);
                        const int scale = k == 0 ? SH2 + 1 : 1;
                        for (int64_t i = 0; i < rowsz; i++) C[i] = (int16_t)(C[i] + ha[i] * scale);
                    }
                } else {
                    int k = y + SH2, r = imin(k, H - 1);
                    if (k < H) {
                        row_features(left, k, H, W, stride, g->ftzero, &fL);
                        row_features(right, k, H, W, stride, g->ftzero, &fR);
                        bt_row(g, &fL, &fR, pix);
                        hsum_row(g, pix, HS(r));
                    }
                    const int16_t *ha = HS(r), *hb = HS(imax(y - SH2 - 1, 0));
                    const int16_t *Cprev = Cfull ? Cfull + (int64_t)(y - 1) * rowsz : Crow;
                    int mx = 0;
                    for (int64_t i = 0; i < rowsz; i++) {
                        int t = Cprev[i] + ha[i]; /* upstream holds this sum (+P2) in int16 */
                        mx = imax(mx, t);
                        C[i] = (int16_t)(t - hb[i]);
                    }
                    max_cp2 = imax(max_cp2, mx + P2);
                }
                {
                    int mx = 0;
                    for (int64_t i = 0; i < rowsz; i++) mx = imax(mx, C[i]);
                    max_cp2 = imax(max_cp2, mx + P2);
                }
                memset(S, 0, (size_t)rowsz * sizeof(int16_t));
            }

            /* four paths whose predecessors are already known: previous pixel of this row and
             * the three neighbours in the previously processed row           */
            lrow_t *cur = &rows[id], *prv = &rows[1 - id];
            for (int x = x1; x != x2; x += dx) {
                const int16_t *Cp = C + (int64_t)x * D;
                int16_t *Sp = S + (int64_t)x * D;
                for (int d = 0; d < D; d++) acc[d] = Sp[d];
                const int xq[4] = {x - dx, x - 1, x, x + 1};
                for (int k = 0; k < 4; k++) {
                    const lrow_t *src = k == 0 ? cur : prv;
                    const int16_t *Lq = src->L[k] + (int64_t)(xq[k] + 1) * (D + 2) + 1;
                    int mq = src->M[k][xq[k] + 1];
                    int16_t *Lo = cur->L[k] + (int64_t)(x + 1) * (D + 2) + 1;
                    cur->M[k][x + 1] = (int16_t)path_step(Cp, Lq, mq, P1, P2, D, Lo, acc, &max_delta);
                }
                for (int d = 0; d < D; d++) Sp[d] = (int16_t)imin(acc[d], MAX_COST);
            }

            if (pass == g->npasses) {
                for (int x = 0; x < W; x++) {
                    drow[x] = disp2[x] = (int16_t)g->invalid_scaled;
                    disp2cost[x] = MAX_COST;
                }
                /* sentinel / border state for the right-to-left path of single-pass mode */
                if (g->npasses == 1) {
                    int16_t *b = L0tmp + (int64_t)(W1 + 1) * (D + 2);
                    b[0] = MAX_COST;
                    for (int d = 1; d <= D; d++) b[d] = 0;
                    b[D + 1] = MAX_COST;
                    M0tmp[W1 + 1] = 0;
                }
                for (int x = W1 - 1; x >= 0; x--) {
                    int16_t *Sp = S + (int64_t)x * D;
                    int minS = MAX_COST, best = -1;
                    if (g->npasses == 1) {
                        /* fifth path (predecessor x+1 of the same row), then S is final */
                        const int16_t *Cp = C + (int64_t)x * D;
                        const int16_t *Lq = L0tmp + (int64_t)(x + 2) * (D + 2) + 1;
                        int16_t *Lo = L0tmp + (int64_t)(x + 1) * (D + 2) + 1;
                        Lo[-1] = MAX_COST;
                        Lo[D] = MAX_COST;
                        for (int d = 0; d < D; d++) acc[d] = Sp[d];
                        M0tmp[x + 1] = (int16_t)path_step(Cp, Lq, M0tmp[x + 2], P1, P2, D, Lo, acc, &max_delta);
                        for (int d = 0; d < D; d++) Sp[d] = (int16_t)imin(acc[d], MAX_COST);
                    }
                    for (int d = 0; d < D; d++) {
                        if (Sp[d] < minS) {
                            minS = Sp[d];
                            best = d;
                        }
                    }
                    /* uniqueness: SURVEY.md A.6 step 2 */
                    int d;
                    for (d = 0; d < D; d++)
                        if (Sp[d] * (100 - g->uniq) < minS * 100 && abs(best - d) > 1) break;
                    if (d < D) continue;
                    d = best;
                    /* right-view disparity, strictly-smaller-cost wins: A.6 step 3 */
                    /* (best = -1 when every S of the pixel is saturated -- nothing is < MAX_COST: upstream then reads
                     * disp2cost one element past the pixel's own column, for the right-most pixel one element past
                     * the buffer; the comparison "> 32767" is false whatever is read, so skipping the read keeps
                     * every value.  Found by the AddressSanitizer pass, tools/sanitize_oracle.sh.) */
                    int x2r = x + g->minX1 - d - g->minD;
                    if (d >= 0 && disp2cost[x2r] > minS) {
                        disp2cost[x2r] = (int16_t)minS;
                        disp2[x2r] = (int16_t)(d + g->minD);
                    }
                    /* parabola fit, C integer division: A.6 step 4 */
                    if (0 < d && d < D - 1) {
                        int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                        d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
                    } else {
                        d *= DISP_SCALE;
                    }
                    drow[x + g->minX1] = (int16_t)(d + g->minD * DISP_SCALE);
                }
                /* left-right consistency: both roundings must disagree to invalidate */
                for (int x = g->minX1; x < g->maxX1; x++) {
                    int d1 = drow[x];
                    if (d1 == g->invalid_scaled) continue;
                    int dlo = d1 >> DISP_SHIFT;
                    int dhi = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
                    int xlo = x - dlo, xhi = x - dhi;
                    if (0 <= xlo && xlo < W && disp2[xlo] >= g->minD && abs(disp2[xlo] - dlo) > g->d12 &&
                        0 <= xhi && xhi < W && disp2[xhi] >= g->minD && abs(disp2[xhi] - dhi) > g->d12)
                        drow[x] = (int16_t)g->invalid_scaled;
                }
            }
            id = 1 - id;
        }
    }
#undef HS

    if (taps) {
        taps->max_cost_plus_p2 = max_cp2;
        taps->max_delta = max_delta;
        taps->headroom_ok = (max_cp2 <= MAX_COST && max_delta <= MAX_COST) ? 1 : 0;
    }
    lrow_free(&rows[0]);
    lrow_free(&rows[1]);
    free(featbuf);
    free(L0tmp);
    free(M0tmp);
    free(disp2);
    free(disp2cost);
    free(acc);
    free(hring);
    free(pix);
    free(Crow);
    free(Srow);
    if (own_C) free(Cfull);
    if (own_S) free(Sfull);
    return 0;
}

/* upstream StereoSGBMImpl::compute : matcher -> medianBlur(3) -> filterSpeckles (main.ipynb:668) */
int oracle_sgbm_compute(const oracle_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                        int H, int W, int64_t stride, int16_t *disp, oracle_sgbm_taps *taps)
{
    geom_t g;
    if (!p || !left || !right || !disp || H <= 0 || W <= 0 || p->numDisparities <= 0) return -1;
    if (p->mode != 0 && p->mode != 1) return -2; /* 3WAY / HH4 are never selected by the reference */
    if (W < 2) return -3;
    make_geom(p, H, W, &g);
    const int64_t n = (int64_t)H * W;
    if (g.W1 <= 0) {
        for (int64_t i = 0; i < n; i++) disp[i] = (int16_t)g.invalid_scaled;
        if (taps) {
            taps->max_cost_plus_p2 = taps->max_delta = 0;
            taps->headroom_ok = 1;
        }
    } else {
        int rc = sgbm_core(&g, left, right, stride, disp, taps);
        if (rc) return rc;
    }
    if (taps && taps->disp_raw) memcpy(taps->disp_raw, disp, (size_t)n * sizeof(int16_t));
    int16_t *tmp = (int16_t *)malloc((size_t)n * sizeof(int16_t));
    memcpy(tmp, disp, (size_t)n * sizeof(int16_t));
    oracle_median3x3_i16(tmp, disp, H, W);
    free(tmp);
    if (taps && taps->disp_median) memcpy(taps->disp_median, disp, (size_t)n * sizeof(int16_t));
    if (p->speckleRange >= 0 && p->speckleWindowSize > 0) /* upstream's condition (stereosgbm.cpp, StereoSGBMImpl::compute) */
        oracle_filter_speckles_i16(disp, H, W, (p->minDisparity - 1) * DISP_SCALE, p->speckleWindowSize,
                                   DISP_SCALE * p->speckleRange);
    return 0;
}

/* main.ipynb:668-670 */
void oracle_disp_to_float(const int16_t *disp, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        float f = (float)disp[i] / 16.0f;
        float m = f > 0.0f ? 1.0f : 0.0f;
        out[i] = f * m;
    }
}

/* main.ipynb:697 ; SURVEY.md Appendix B.  Double arithmetic, sums in index order starting
 * from 0, divide implemented as multiply by the reciprocal, one rounding to float each.
 * Built with -ffp-contract=off so no fused multiply-add is formed.                         */
void oracle_reproject_f32(const float *disp, int H, int W, const double Q[16], int handle_missing,
                          float *xyz)
{
    double minDisparity = FLT_MAX;
    if (handle_missing) {
        double m = (double)disp[0];
        for (int64_t i = 1; i < (int64_t)H * W; i++)
            if ((double)disp[i] < m) m = (double)disp[i];
        minDisparity = m;
    }
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const double d = (double)disp[(int64_t)y * W + x];
            const double v[4] = {(double)x, (double)y, d, 1.0};
            double h[4];
            for (int i = 0; i < 4; i++) {
                double s = 0.0;
                for (int k = 0; k < 4; k++) s += Q[i * 4 + k] * v[k];
                h[i] = s;
            }
            float *o = xyz + ((int64_t)y * W + x) * 3;
            const double ia = 1.0 / h[3];
            for (int i = 0; i < 3; i++) {
                /* volatile: gcc -O3 otherwise folds the float round trip away when it
                 * vectorises this loop (observed: last-bit differences vs -O0/-O2) */
                volatile float f = (float)h[i];
                o[i] = (float)((double)f * ia);
            }
            if (fabs(d - minDisparity) <= FLT_EPSILON) o[2] = 10000.f;
        }
    }
}

/* main.ipynb:726-730 */
void oracle_valid_mask(const float *xyz, const float *disp, int64_t n, uint8_t *mask)
{
    for (int64_t i = 0; i < n; i++) {
        float X = xyz[i * 3];
        mask[i] = (uint8_t)(!isnan(X) && !isinf(X) && disp[i] > 0.0f);
    }
}

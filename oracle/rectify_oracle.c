/*
 * rectify_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rectify_oracle.h.
 *
 * Restates OpenCV 4.11 initUndistortRectifyMap (scalar line computer) and remap(INTER_LINEAR,
 * BORDER_CONSTANT 0) on 8-bit images.  Build with -ffp-contract=off (oracle/Makefile): the map
 * arithmetic is double precision in a fixed operation order.
 */
#include "rectify_oracle.h"

#include <limits.h>
#include <math.h>
#include <string.h>

/* cv::invert, n == 3 branch of modules/core/src/lapack.cpp (double) */
int oracle_invert3x3(const double m[9], double out[9])
{
#define S(r, c) m[(r) * 3 + (c)]
    double d = S(0, 0) * (S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) - S(0, 1) * (S(1, 0) * S(2, 2) - S(1, 2) * S(2, 0)) +
               S(0, 2) * (S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0));
    if (d == 0.) {
        memset(out, 0, 9 * sizeof(double));
        return 0;
    }
    d = 1. / d;
    out[0] = (S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) * d;
    out[1] = (S(0, 2) * S(2, 1) - S(0, 1) * S(2, 2)) * d;
    out[2] = (S(0, 1) * S(1, 2) - S(0, 2) * S(1, 1)) * d;
    out[3] = (S(1, 2) * S(2, 0) - S(1, 0) * S(2, 2)) * d;
    out[4] = (S(0, 0) * S(2, 2) - S(0, 2) * S(2, 0)) * d;
    out[5] = (S(0, 2) * S(1, 0) - S(0, 0) * S(1, 2)) * d;
    out[6] = (S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0)) * d;
    out[7] = (S(0, 1) * S(2, 0) - S(0, 0) * S(2, 1)) * d;
    out[8] = (S(0, 0) * S(1, 1) - S(0, 1) * S(1, 0)) * d;
#undef S
    return 1;
}

int oracle_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double *R,
                                      const double *P, int pcols, int W, int H, float *map1, float *map2)
{
    double k[12] = {0};
    if (dist) {
        if (ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12) return -1;
        for (int i = 0; i < ndist; i++) k[i] = dist[i];
    }
    if (P && pcols != 3 && pcols != 4) return -1;
    if (W <= 0 || H <= 0) return -1;
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];
    double Ar[9], Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ArR[9], ir[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Ar[r * 3 + c] = P ? P[r * pcols + c] : K[r * 3 + c];
    if (R) memcpy(Rm, R, sizeof(Rm));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int q = 0; q < 3; q++) s += Ar[r * 3 + q] * Rm[q * 3 + c];
            ArR[r * 3 + c] = s;
        }
    if (!oracle_invert3x3(ArR, ir)) return -1;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    for (int i = 0; i < H; i++) {
        float *m1f = map1 + (int64_t)i * W, *m2f = map2 + (int64_t)i * W;
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            /* matTilt is the identity without tilt coefficients: vecTilt = (xd, yd, 1), invProj = 1 */
            double u = fx * 1. * xd + u0;
            double v = fy * 1. * yd + v0;
            m1f[j] = (float)u;
            m2f[j] = (float)v;
        }
    }
    return 0;
}

void oracle_bilinear_tab_i16(int16_t tab[32 * 32 * 4])
{
    /* initInterTab2D(INTER_LINEAR, fixpt): products of the 1-D weights {1 - f/32, f/32} times 2^15.
     * Every product is a multiple of 2^-10, so the int16 value is exact -- except 1.0 * 2^15, which
     * saturate_cast<short> turns into 32767; the sum check then adds the missing 1 to the largest
     * of the entries it inspects (index 3 of this entry and the still-zero next entries): index 3. */
    for (int fy = 0; fy < 32; fy++)
        for (int fx = 0; fx < 32; fx++) {
            int16_t *t = tab + (fy * 32 + fx) * 4;
            const float wy[2] = {1.f - fy * (1.f / 32), fy * (1.f / 32)};
            const float wx[2] = {1.f - fx * (1.f / 32), fx * (1.f / 32)};
            int isum = 0;
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) {
                    const float v = wy[a] * wx[b] * 32768.f;
                    long r = lrintf(v);
                    if (r > SHRT_MAX) r = SHRT_MAX;
                    t[a * 2 + b] = (int16_t)r;
                    isum += (int)r;
                }
            if (isum != 32768) t[3] = (int16_t)(t[3] - (isum - 32768));
        }
}

/* cvRound(float) on x86: round half to even; out-of-range -> INT_MIN like cvtss2si */
static int cv_round_f(float v)
{
    if (!(v > -2147483648.f && v < 2147483648.f)) return INT_MIN;
    return (int)lrintf(v);
}
static int sat_short(int v) { return v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v); }

void oracle_remap_linear_u8(const uint8_t *src, int sH, int sW, int64_t sstride, int cn, const float *map1,
                            const float *map2, int dH, int dW, uint8_t *dst)
{
    static int16_t tab[32 * 32 * 4];
    static int have_tab = 0;
    if (!have_tab) {
        oracle_bilinear_tab_i16(tab);
        have_tab = 1;
    }
    for (int dy = 0; dy < dH; dy++)
        for (int dx = 0; dx < dW; dx++) {
            const int64_t o = (int64_t)dy * dW + dx;
            /* remap(): float maps -> XY (int16 pixel) + A (5+5 fraction bits) */
            const int fxq = cv_round_f(map1[o] * 32), fyq = cv_round_f(map2[o] * 32);
            const int sx = sat_short(fxq >> 5), sy = sat_short(fyq >> 5);
            const int16_t *w = tab + ((fyq & 31) * 32 + (fxq & 31)) * 4;
            uint8_t *D = dst + o * cn;
            if ((unsigned)sx < (unsigned)(sW - 1) && (unsigned)sy < (unsigned)(sH - 1)) {
                const uint8_t *S = src + sy * sstride + (int64_t)sx * cn;
                for (int c = 0; c < cn; c++) {
                    const int v = S[c] * w[0] + S[c + cn] * w[1] + S[sstride + c] * w[2] + S[sstride + c + cn] * w[3];
                    const int r = (v + (1 << 14)) >> 15;
                    D[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            } else if (sx >= sW || sx + 1 < 0 || sy >= sH || sy + 1 < 0) {
                for (int c = 0; c < cn; c++) D[c] = 0; /* BORDER_CONSTANT, borderValue 0 */
            } else {
                const int ok00 = (unsigned)sx < (unsigned)sW && (unsigned)sy < (unsigned)sH;
                const int ok01 = (unsigned)(sx + 1) < (unsigned)sW && (unsigned)sy < (unsigned)sH;
                const int ok10 = (unsigned)sx < (unsigned)sW && (unsigned)(sy + 1) < (unsigned)sH;
                const int ok11 = (unsigned)(sx + 1) < (unsigned)sW && (unsigned)(sy + 1) < (unsigned)sH;
                for (int c = 0; c < cn; c++) {
                    const int v0 = ok00 ? src[sy * sstride + (int64_t)sx * cn + c] : 0;
                    const int v1 = ok01 ? src[sy * sstride + (int64_t)(sx + 1) * cn + c] : 0;
                    const int v2 = ok10 ? src[(sy + 1) * sstride + (int64_t)sx * cn + c] : 0;
                    const int v3 = ok11 ? src[(sy + 1) * sstride + (int64_t)(sx + 1) * cn + c] : 0;
                    const int v = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
                    const int r = (v + (1 << 14)) >> 15;
                    D[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            }
        }
}

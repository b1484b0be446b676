// Rectification in front of the disparity path:
//   cv2.initUndistortRectifyMap(K, dist, R, P, size, CV_32FC1)   /root/reference/gui.py:160-161
//   cv2.remap(img, map1, map2, INTER_LINEAR)                     /root/reference/gui.py:163-164
// Arithmetic as in OpenCV 4.11 (undistort.dispatch.cpp scalar line computer; imgwarp.cpp remap
// with float maps -> 1/32-pixel fixed point, int16 bilinear table, (v + 2^14) >> 15).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgm {

struct RectifyArgs {
    double ir[9];  // (P[:, :3] * R)^-1
    double k[12];  // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double fx, fy, u0, v0;
};

// One thread per destination row.  Upstream walks a row with _x += ir[0] (and _y, _w likewise):
// every column's value is the rounded sum of the previous one, so the row is a sequential chain
// of f64 additions and a parallel "x0 + j*ir[0]" would differ in the last bits.  The maps are
// made once per calibration; H threads of W steps take well under a millisecond of f64 work.
// Built with -ffp-contract=off.
__global__ __launch_bounds__(64) void k_rectify_map(RectifyArgs a, int W, int H, float *__restrict__ map1,
                                                    float *__restrict__ map2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const double k1 = a.k[0], k2 = a.k[1], p1 = a.k[2], p2 = a.k[3], k3 = a.k[4], k4 = a.k[5], k5 = a.k[6], k6 = a.k[7];
    const double s1 = a.k[8], s2 = a.k[9], s3 = a.k[10], s4 = a.k[11];
    float *m1f = map1 + (int64_t)i * W, *m2f = map2 + (int64_t)i * W;
    double _x = i * a.ir[1] + a.ir[2], _y = i * a.ir[4] + a.ir[5], _w = i * a.ir[7] + a.ir[8];
    for (int j = 0; j < W; j++, _x += a.ir[0], _y += a.ir[3], _w += a.ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
        const double u = a.fx * 1. * xd + a.u0;
        const double v = a.fy * 1. * yd + a.v0;
        m1f[j] = (float)u;
        m2f[j] = (float)v;
    }
}

// cvRound(float) (round half to even); out of int range -> INT_MIN as cvtss2si does
__device__ __forceinline__ int cv_round_f(float v)
{
    if (!(v > -2147483648.f && v < 2147483648.f)) return (int)0x80000000;
    return __float2int_rn(v);
}

// int16 bilinear weights of fraction (fx, fy) in 1/32 pixel, scale 2^15; (0,0) -> {32767,0,0,1}
__device__ __forceinline__ void bilinear_w(int fx, int fy, int &w0, int &w1, int &w2, int &w3)
{
    w0 = (32 - fy) * (32 - fx) * 32;
    w1 = (32 - fy) * fx * 32;
    w2 = fy * (32 - fx) * 32;
    w3 = fy * fx * 32;
    if ((fx | fy) == 0) {
        w0 = 32767;
        w3 = 1;
    }
}

// One thread per destination pixel, CN interleaved 8-bit channels, BORDER_CONSTANT 0.
template <int CN>
__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t *__restrict__ src, int sH, int sW, int64_t sstride,
                                                      const float *__restrict__ map1, const float *__restrict__ map2,
                                                      int dH, int dW, uint8_t *__restrict__ dst, int64_t dstride)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dW) return;
    const int64_t o = (int64_t)dy * dW + dx;
    const int fxq = cv_round_f(map1[o] * 32.f), fyq = cv_round_f(map2[o] * 32.f);
    const int sx = min(max(fxq >> 5, -32768), 32767), sy = min(max(fyq >> 5, -32768), 32767);
    int w0, w1, w2, w3;
    bilinear_w(fxq & 31, fyq & 31, w0, w1, w2, w3);
    uint8_t *D = dst + dy * dstride + (int64_t)dx * CN;
    // taps outside the image read as 0 (this covers upstream's three cases: all four inside,
    // all four outside -> borderValue, mixed -> per-tap borderValue)
    const bool x0 = (unsigned)sx < (unsigned)sW, x1 = (unsigned)(sx + 1) < (unsigned)sW;
    const bool y0 = (unsigned)sy < (unsigned)sH, y1 = (unsigned)(sy + 1) < (unsigned)sH;
    const uint8_t *r0 = src + (int64_t)(y0 ? sy : 0) * sstride, *r1 = src + (int64_t)(y1 ? sy + 1 : 0) * sstride;
    const int64_t c0 = (int64_t)(x0 ? sx : 0) * CN, c1 = (int64_t)(x1 ? sx + 1 : 0) * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int v00 = (x0 && y0) ? r0[c0 + c] : 0, v01 = (x1 && y0) ? r0[c1 + c] : 0;
        const int v10 = (x0 && y1) ? r1[c0 + c] : 0, v11 = (x1 && y1) ? r1[c1 + c] : 0;
        const int r = (v00 * w0 + v01 * w1 + v10 * w2 + v11 * w3 + (1 << 14)) >> 15;
        D[c] = (uint8_t)min(max(r, 0), 255);
    }
}

}  // namespace sgm

/*
 * rectify_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, scalar) of the rectification step in front of the reference's
 * disparity path:
 *
 *     cv2.initUndistortRectifyMap(K, None, R, P, image_size, cv2.CV_32F)   /root/reference/gui.py:160-161
 *     cv2.remap(img, map1, map2, interpolation=cv2.INTER_LINEAR)           /root/reference/gui.py:163-164
 *                                                                          /root/reference/main.ipynb cell 7
 *
 * The arithmetic lives in opencv-python==4.11.0.86 (/root/reference/environment.yml:89-90), not
 * vendored in /root/reference: modules/calib3d/src/undistort.dispatch.cpp
 * (initUndistortRectifyMap, scalar line computer), modules/core/src/lapack.cpp (3x3 inverse),
 * modules/imgproc/src/imgwarp.cpp (remap: float maps -> 1/32-pixel fixed point with cvRound,
 * initInterTab2D's int16 bilinear table, remapBilinear with FixedPtCast<int, uchar, 15>).
 * This file restates the published algorithm of that version.
 *
 * PARITY UNPINNED, as for sgbm_oracle.h: cv2 is absent here and the reference holds no fixtures
 * for this step.  Two things are known to be build-dependent upstream and are fixed here to the
 * scalar C++ path: the map computer accumulates _x += ir[0] column by column (the AVX2 line
 * computer steps four columns at a time), and the remap runs in 1/32-pixel fixed point (an IPP /
 * OpenCL build may not).  The integer part (remap given maps) has no such freedom.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef RECTIFY_ORACLE_H
#define RECTIFY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* inverse of a 3x3 double matrix the way cv::invert(DECOMP_LU) does it for n = 3 (closed form).
 * Returns 0 and leaves out[] zero when the determinant is 0. */
int oracle_invert3x3(const double m[9], double out[9]);

/* cv2.initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, (W, H), CV_32FC1).
 *   K     3x3 row-major
 *   dist  NULL or ndist in {4, 5, 8, 12} coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]);
 *         the tilted-sensor terms (14 coefficients) are not restated
 *   R     3x3 row-major or NULL (identity)
 *   P     new camera matrix: 3x3 (pcols = 3) or 3x4 (pcols = 4, fourth column ignored) or NULL (= K)
 *   map1, map2   [H][W] float32: source x and y for every destination pixel
 * Returns 0, or -1 on an unsupported argument / singular P*R. */
int oracle_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double *R,
                                      const double *P, int pcols, int W, int H, float *map1, float *map2);

/* cv2.remap(src, map1, map2, INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for 8-bit
 * images with cn interleaved channels (1..4).  src is [sH][sW][cn] with row stride sstride bytes;
 * dst is [dH][dW][cn] dense; the maps are [dH][dW] float32. */
void oracle_remap_linear_u8(const uint8_t *src, int sH, int sW, int64_t sstride, int cn, const float *map1,
                            const float *map2, int dH, int dW, uint8_t *dst);

/* the int16 bilinear weight table of imgwarp.cpp: tab[(fy*32 + fx)*4 + {0,1,2,3}] for the taps
 * (x,y), (x+1,y), (x,y+1), (x+1,y+1); entry (0,0) is {32767, 0, 0, 1} (saturate_cast quirk). */
void oracle_bilinear_tab_i16(int16_t tab[32 * 32 * 4]);

#ifdef __cplusplus
}
#endif
#endif

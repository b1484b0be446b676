/*
 * sgbm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, scalar, single thread) of the arithmetic behind the
 * reference's "Run Disparity" path:
 *
 *     cv2.StereoSGBM_create(...).compute(imgL, imgR)      /root/reference/main.ipynb:655-668
 *     .astype(np.float32)/16 ; mask ; multiply            /root/reference/main.ipynb:668-670
 *     cv2.reprojectImageTo3D(disparity_map, Q)            /root/reference/main.ipynb:697
 *     validity mask of visualize_point_cloud              /root/reference/main.ipynb:726-730
 *
 * The arithmetic itself lives in a third-party dependency that is NOT vendored in
 * /root/reference: opencv-python==4.11.0.86 (/root/reference/environment.yml:89-90),
 * files modules/calib3d/src/stereosgbm.cpp (StereoSGBMImpl::compute, computeDisparitySGBM,
 * calcPixelCostBT, filterSpecklesImpl), modules/imgproc/src/median_blur.simd.hpp (3x3, 16S),
 * modules/calib3d/src/calibration.cpp (reprojectImageTo3D), modules/core/.../matx.hpp.
 * This file restates the published algorithm of that version.
 *
 * PARITY UNPINNED: cv2 is not importable in this pipeline (plain absence, no index) and the
 * reference holds no tests, golden vectors or numeric fixtures for this path.
 * The restatement is anchored by implementation-independent known-answer tests
 * (tests/test_oracle_known_answers.py) and by a second, brute-force formula-level
 * restatement in numpy (tests/bruteforce_sgbm.py).  Claims made against it read
 * "bit-exact vs. a restatement of OpenCV 4.11 MODE_SGBM / MODE_HH inside the int16
 * no-overflow regime", never "bit-exact vs. the cv2 wheel".
 *
 * One place where no parity is definable: upstream's first element of the horizontal running sum
 * reads pixDiff[x + d] for x <= SW2*D without clamping x to width1 - 1.  When width1 <= SW2 (fewer
 * matchable columns than the window radius) that read runs past the row upstream allocated
 * (undefined); this restatement clamps the column to width1 - 1 (sgbm_oracle.c, hsum_row), which is
 * what the definition of A.4 says.  Every frame with width1 > SW2 is unaffected.
 * Also: filterSpeckles runs only if speckleRange >= 0 && speckleWindowSize > 0, as upstream.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef SGBM_ORACLE_H
#define SGBM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* keyword arguments of cv2.StereoSGBM_create as used at main.ipynb:655-666 (+ mode) */
typedef struct {
    int32_t minDisparity;
    int32_t numDisparities;
    int32_t blockSize;
    int32_t P1;
    int32_t P2;
    int32_t disp12MaxDiff;
    int32_t preFilterCap;
    int32_t uniquenessRatio;
    int32_t speckleWindowSize;
    int32_t speckleRange;
    int32_t mode; /* 0 = MODE_SGBM (5 paths, one pass), 1 = MODE_HH (8 paths, two passes) */
} oracle_sgbm_params;

/* Optional stage taps; every pointer may be NULL. Volumes are [H][W1][D] int16, d fastest. */
typedef struct {
    int16_t *C;           /* block cost WITHOUT the +P2 bias upstream adds */
    int16_t *S;           /* aggregated cost after every path has been added */
    int16_t *disp_raw;    /* [H][W] after WTA/uniqueness/subpixel/LR check, before median */
    int16_t *disp_median; /* [H][W] after the 3x3 median, before the speckle filter */
    /* outputs: largest value any int16-typed intermediate of upstream would have held */
    int32_t max_cost_plus_p2;   /* max C_true + P2 (and the running-sum intermediate)   */
    int32_t max_delta;          /* max P2 + min_d L_r(p, d) over all pixels and directions */
    int32_t headroom_ok;        /* 1 iff both stayed <= 32767             */
} oracle_sgbm_taps;

/* geometry helper: W1 = number of valid columns, first valid image column */
void oracle_sgbm_geometry(const oracle_sgbm_params *p, int W, int *minX1, int *W1);

/* full .compute(): u8 H x W (row stride in bytes) x 2 -> int16 H x W (disp * 16). returns 0 / <0 */
int oracle_sgbm_compute(const oracle_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                        int H, int W, int64_t stride, int16_t *disp, oracle_sgbm_taps *taps);

/* stage functions (also used by compute) */
void oracle_median3x3_i16(const int16_t *src, int16_t *dst, int H, int W);
void oracle_filter_speckles_i16(int16_t *img, int H, int W, int newVal, int maxSpeckleSize,
                                int maxDiff);

/* main.ipynb:668-670 : f = i16/16 ; f * (f > 0) */
void oracle_disp_to_float(const int16_t *disp, float *out, int64_t n);

/* cv2.reprojectImageTo3D(float32 disparity, Q, handleMissingValues) -> float32 H x W x 3 */
void oracle_reproject_f32(const float *disp, int H, int W, const double Q[16],
                          int handle_missing, float *xyz);

/* main.ipynb:726-730 : ~isnan(X) & ~isinf(X) & (disp > 0) */
void oracle_valid_mask(const float *xyz, const float *disp, int64_t n, uint8_t *mask);

#ifdef __cplusplus
}
#endif
#endif

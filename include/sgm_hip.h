/*
 * sgm_hip.h -- C ABI of the MI355X (gfx950) dense stereo disparity engine.
 *
 * Drop-in boundary for the reference's "Run Disparity" path.  The reference has no FFI of its
 * own for this path: the boundary is the cv2 Python API at three call sites, and each entry
 * point below names the call it replaces (file:line in /root/reference):
 *
 *   sgm_create           <- cv2.StereoSGBM_create(minDisparity=..., ...)      main.ipynb:655-666
 *   sgm_compute          <- stereo.compute(imgL, imgR)  -> int16 HxW          main.ipynb:668
 *   sgm_disp_to_float    <- .astype(np.float32)/16 ; mask = >0 ; multiply      main.ipynb:668-670
 *   sgm_reproject        <- cv2.reprojectImageTo3D(disparity_map, Q)           main.ipynb:697
 *   sgm_valid_mask       <- ~isnan(X) & ~isinf(X) & (disparity_map > 0)        main.ipynb:726-730
 *   sgm_median3x3,
 *   sgm_filter_speckles  <- the two post-filters .compute() applies internally (upstream
 *                           medianBlur(disp,3) / filterSpeckles; cv2.medianBlur /
 *                           cv2.filterSpeckles are their public faces)        main.ipynb:664-665,668
 *   sgm_compact_points   <- valid_points = points_3D[mask]; valid_colors = colors[mask]
 *                           (ordered compaction + colour gather)                main.ipynb:726-737
 *   sgm_pipeline_device  <- cell c13: compute -> scale/mask -> reproject       main.ipynb:781,790
 *   sgm_init_undistort_rectify_map
 *                        <- cv2.initUndistortRectifyMap(K, None, R1, P1, size, cv2.CV_32F)
 *                                                                    gui.py:160-161, main.ipynb cell 7
 *   sgm_remap_linear_u8  <- cv2.remap(img, map1, map2, interpolation=cv2.INTER_LINEAR)
 *                                                                    gui.py:163-164, main.ipynb cell 7
 *   sgm_compute_batch,
 *   sgm_pipeline_batch_device
 *                        <- cell c13 over N independent pairs (BASELINE config 4: a batch per GPU;
 *                           the frame sharding unit).  With SGM_OPT_SCHEDULE = 2 ("throughput mode")
 *                           the pairs of a batch share one chained sweep launch per pass     main.ipynb:780-797
 *   sgm_get_headroom     <- (no counterpart) tells the caller whether the last compute stayed inside
 *                           the int16 regime in which OpenCV's own arithmetic is exact
 *   sgm_check, sgm_trim  <- (no counterpart) status of a stream-ordered engine; memory of the batch entries' groups
 *
 * Conventions: plain pointers and sizes, no C++ types, no exceptions across the boundary.
 * Every function returns 0 on success or a negative sgm_status; sgm_last_error() returns a
 * thread-local message.  All host buffers are caller-owned; device memory lives in the engine
 * and is reused between calls (regrown on shape change).  An engine is bound to one GPU and
 * one HIP stream and is not thread-safe: one engine per thread / GPU.  Host-pointer entry
 * points block until the result is in the caller's buffer (the notebook / Tk call shape);
 * *_device entry points take device pointers, enqueue on the given stream and return.
 * There is NO CPU fallback: without a usable GPU sgm_create fails with SGM_ERR_NO_DEVICE.
 */
#ifndef SGM_HIP_H
#define SGM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGM_ABI_VERSION 4   /* 2: sgm_get_headroom, SGM_OPT_PREPASS_ROWS (round 2); 3: sgm_pipeline_batch_device, SGM_OPT_CHAIN_WGS, schedule 2;
                             * 4: sgm_check, sgm_trim, sgm_compact_points_device_async, SGM_OPT_GROUP_MAX (round 4) */

typedef enum {
    SGM_OK = 0,
    SGM_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, unsupported parameter       */
    SGM_ERR_NO_DEVICE = -2,     /* no HIP device / device index out of range                     */
    SGM_ERR_HIP = -3,           /* a HIP runtime call failed; message has the hipError string    */
    SGM_ERR_UNSUPPORTED = -4,   /* mode 2/3 (3WAY/HH4), numDisparities not a multiple of 16, ... */
    SGM_ERR_NOMEM = -5
} sgm_status;

/* keyword arguments of cv2.StereoSGBM_create (main.ipynb:655-666); mode: 0 = MODE_SGBM
 * (5 paths, what the notebook runs), 1 = MODE_HH (8 paths).  Zero / negative values are
 * normalised exactly as OpenCV 4.11 does. */
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
    int32_t mode;
} sgm_params;

typedef struct sgm_engine sgm_engine; /* opaque */

/* stage taps of the last sgm_compute*: device buffers copied to host for the parity tests */
typedef enum {
    SGM_TAP_COST = 0,        /* int16 [H][W1][D]  block cost C (without upstream's +P2 bias)     */
    SGM_TAP_AGGR = 1,        /* int16 [H][W1][D]  aggregated cost S (needs sgm_set_option KEEP)  */
    SGM_TAP_DISP_RAW = 2,    /* int16 [H][W]      after WTA / uniqueness / sub-pixel / LR check  */
    SGM_TAP_DISP_MEDIAN = 3  /* int16 [H][W]      after the 3x3 median                           */
} sgm_tap;

typedef enum {
    SGM_OPT_KEEP_AGGR = 0,   /* 1: the last path kernel also stores S (debug; costs bandwidth)   */
    SGM_OPT_PROFILE = 1,     /* 1: bracket every stage with HIP events on the engine's stream    */
    SGM_OPT_SCHEDULE = 2,    /* 0: one kernel per path direction; 1 (default): fused 4-direction sweeps behind a boundary
                              * pre-pass (lowest latency of one pair); 2: chained sweeps, no pre-pass (throughput mode: fewer
                              * bytes per pair, one pair alone is slower -- meant for several pairs in flight) */
    SGM_OPT_SWEEP_ROWS = 3,  /* rows per band of the fused sweep; 0 = automatic                  */
    SGM_OPT_CHAIN_WGS = 6,   /* schedule 2: workgroups (bands in flight) per sweep launch; 0 = automatic       */
    SGM_OPT_PREPASS_ROWS = 5, /* rows per chunk (= launch) of the boundary pre-pass; 0 = automatic (about 135, a multiple of 8) */
    SGM_OPT_GROUP_MAX = 7,   /* schedule 2, batch entries: pairs per chained launch (= internal engines kept, about 9 GB each at
                              * 4K D=256); 0 = automatic: as many as device memory holds beside a reserve, up to 64; 1 = never chain */
    /* 4 = SGM_OPT_DEBUG: A/B switches for measurements -- not part of this interface (csrc/sgm_debug.h) */
    SGM_OPT_RESERVED_4 = 4
} sgm_option;

#define SGM_MAX_STAGES 32
typedef struct {
    int32_t n;                         /* number of stages recorded by the last compute        */
    const char *name[SGM_MAX_STAGES];  /* static strings                                       */
    float ms[SGM_MAX_STAGES];          /* HIP-event elapsed time of each stage                 */
    int32_t launches[SGM_MAX_STAGES];  /* kernel launches inside the stage                     */
} sgm_stage_times;

int sgm_abi_version(void);
int sgm_device_count(void);
const char *sgm_last_error(void);

/* stream: a hipStream_t passed as void*, or NULL to let the engine create its own. */
int sgm_create(const sgm_params *params, int device_id, void *stream, sgm_engine **out);
void sgm_destroy(sgm_engine *e);
int sgm_set_option(sgm_engine *e, int option, int value);

/* geometry: W1 = number of columns that can be matched, minX1 = first such column */
int sgm_geometry(const sgm_params *params, int W, int *minX1, int *W1);

/* ---- host-pointer entry points (blocking) ---- */
int sgm_compute(sgm_engine *e, const uint8_t *left, const uint8_t *right, int H, int W,
                int64_t stride_bytes, int16_t *disp_out /* H*W */);
/* N dense pairs (tight rows) from / to host memory (pageable is fine).  Default schedule: up to three pairs in flight on
 * internal peer engines, images and maps staged through page-locked buffers.  SGM_OPT_SCHEDULE = 2: chained groups as in
 * sgm_pipeline_batch_device with two groups in flight -- uploads of the next and downloads of the previous group run
 * beside the kernels of the current one */
int sgm_compute_batch(sgm_engine *e, int N, const uint8_t *lefts, const uint8_t *rights, int H,
                      int W, int16_t *disps_out /* N*H*W */, float *xyz_out /* N*H*W*3 or NULL */,
                      const double *Q16 /* needed iff xyz_out */);
int sgm_disp_to_float(sgm_engine *e, const int16_t *disp, int64_t n, float *out);
int sgm_reproject(sgm_engine *e, const float *disp, int H, int W, const double Q[16],
                  int handle_missing, float *xyz_out /* H*W*3 */);
int sgm_valid_mask(sgm_engine *e, const float *xyz, const float *disp, int64_t n, uint8_t *mask);
/* out_points: up to n*3 floats, out_colors (nullable, needs colors): up to n*3 bytes; *n_valid = rows written */
int sgm_compact_points(sgm_engine *e, const float *xyz, const float *disp, const uint8_t *colors_rgb,
                       int64_t n, float *out_points, uint8_t *out_colors, int64_t *n_valid);
int sgm_median3x3(sgm_engine *e, const int16_t *src, int H, int W, int16_t *dst);
int sgm_filter_speckles(sgm_engine *e, int16_t *img /* in place */, int H, int W, int newVal,
                        int maxSpeckleSize, int maxDiff);
int sgm_get_tap(sgm_engine *e, int tap, void *host_dst, int64_t bytes);
/* Regime record of the last compute (blocks on the engine's stream).  OpenCV 4.11's StereoSGBM
 * (behind stereo.compute, /root/reference/main.ipynb:668) keeps C + P2 and min_d L_r + P2 in int16
 * lanes with a mix of wrapping, truncating and saturating arithmetic; its results are exact integer
 * arithmetic -- and this engine's results are claimed bit-exact -- only while both stay <= 32767
 * (SURVEY.md A.9).  The kernels record both maxima on the device:
 *   *max_cost_plus_p2 = P2 + max of the block cost, including the running-sum intermediate
 *                       C(y-1) + hsum(y+r) upstream forms;   *max_delta = P2 + max over pixels and
 *   directions of min_d L_r(p, d);   *ok = 1 iff both fit.  The notebook's own setting
 *   (blockSize=11, P2=11616, main.ipynb:655-666) CAN leave the regime on high-contrast input. */
int sgm_get_headroom(sgm_engine *e, int *max_cost_plus_p2, int *max_delta, int *ok);

/* Rectification in front of the path.
 * K: 3x3 row-major.  dist: NULL or ndist in {4,5,8,12} coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6
 * [s1 s2 s3 s4]]]); 14 (tilt) -> SGM_ERR_UNSUPPORTED.  R: 3x3 or NULL (identity).  P: 3x3
 * (pcols 3) / 3x4 (pcols 4) or NULL (= K).  map1/map2: [H][W] float32 (the CV_32FC1 map pair). */
int sgm_init_undistort_rectify_map(sgm_engine *e, const double K[9], const double *dist, int ndist,
                                   const double *R, const double *P, int pcols, int W, int H,
                                   float *map1_out, float *map2_out);
/* INTER_LINEAR, BORDER_CONSTANT with borderValue 0; src [sH][sW][cn] uint8 (cn 1..4, row stride
 * sstride bytes), maps [dH][dW] float32, dst [dH][dW][cn] dense. */
int sgm_remap_linear_u8(sgm_engine *e, const uint8_t *src, int sH, int sW, int64_t sstride, int cn,
                        const float *map1, const float *map2, int dH, int dW, uint8_t *dst);

/* ---- device-pointer entry points (asynchronous on the engine's stream) ---- */
int sgm_compute_device(sgm_engine *e, const void *d_left, const void *d_right, int H, int W,
                       int64_t stride_bytes, void *d_disp_i16);
int sgm_disp_to_float_device(sgm_engine *e, const void *d_disp_i16, int64_t n, void *d_out_f32);
int sgm_reproject_device(sgm_engine *e, const void *d_disp_f32, int H, int W, const double Q[16],
                         int handle_missing, void *d_xyz_f32);
int sgm_valid_mask_device(sgm_engine *e, const void *d_xyz, const void *d_disp_f32, int64_t n,
                          void *d_mask_u8);
/* device buffers sized for the worst case (n points); blocks until the count is known */
int sgm_compact_points_device(sgm_engine *e, const void *d_xyz, const void *d_disp_f32,
                              const void *d_colors_rgb, int64_t n, void *d_out_points,
                              void *d_out_colors, int64_t *n_valid);
/* the same without a host round trip: the count is written to *d_n_valid_i64 (an int64 in DEVICE memory) in stream order and
 * nothing is synchronised -- the form the sharded pipeline uses (counts and points travel to rank 0 behind an event).
 * Rows of d_out_points past the count are left as they were. */
int sgm_compact_points_device_async(sgm_engine *e, const void *d_xyz, const void *d_disp_f32,
                                    const void *d_colors_rgb, int64_t n, void *d_out_points,
                                    void *d_out_colors, void *d_n_valid_i64);
int sgm_init_undistort_rectify_map_device(sgm_engine *e, const double K[9], const double *dist, int ndist,
                                          const double *R, const double *P, int pcols, int W, int H,
                                          void *d_map1_f32, void *d_map2_f32);
int sgm_remap_linear_u8_device(sgm_engine *e, const void *d_src, int sH, int sW, int64_t sstride, int cn,
                               const void *d_map1_f32, const void *d_map2_f32, int dH, int dW, void *d_dst,
                               int64_t dstride);
/* cell c13 in one call: disparity (int16) -> float disparity -> XYZ; any output may be NULL */
int sgm_pipeline_device(sgm_engine *e, const void *d_left, const void *d_right, int H, int W,
                        int64_t stride_bytes, const double Q[16], void *d_disp_i16,
                        void *d_disp_f32, void *d_xyz_f32);
/* The driver cell over N resident pairs (BASELINE config 4: a batch of pairs per GPU; main.ipynb:780-797 once per pair),
 * throughput mode: arrays of N device pointers (d_disp_f32 / d_xyz_f32 may be NULL).  With SGM_OPT_SCHEDULE = 2 the
 * pairs share ONE chained sweep launch per pass, in groups on internal engines (about 9 GB of device memory per 4K D=256
 * pair of a group): a group is as large as device memory allows beside a reserve (at most 64 pairs, or
 * SGM_OPT_GROUP_MAX), and a batch larger than that is cut into groups of equal size; configurations the chained sweeps do
 * not cover (D <= 32, a single band) and other schedules run pair after pair.  Results equal N calls of
 * sgm_pipeline_device; sgm_get_headroom afterwards covers every pair of the call.  Asynchronous: sgm_synchronize(e)
 * waits for all of it.  An error return leaves nothing in flight (every internal stream is drained first) and the engine
 * usable.  sgm_trim(e) gives the internal engines' memory back. */
int sgm_pipeline_batch_device(sgm_engine *e, int N, const void *const *d_left, const void *const *d_right, int H, int W,
                              int64_t stride_bytes, const double Q[16], void *const *d_disp_i16,
                              void *const *d_disp_f32, void *const *d_xyz_f32);
/* blocks until everything enqueued on the engine's stream is done; reports (once) a chained sweep that gave up, see sgm_check */
int sgm_synchronize(sgm_engine *e);
/* Status of the engine WITHOUT waiting for its stream, for callers that order the stream with events of their own
 * (stream-ordered pipelines): SGM_ERR_HIP if a workgroup of a chained sweep (schedule 2) gave up waiting for the band
 * above it since the last check -- a bounded wait that cannot end in a healthy launch (kernels_sweep.h: ChainWait); the
 * results computed since the last check are then invalid.  The condition is reported once and cleared: the engine stays
 * usable.  sgm_synchronize, sgm_compute, sgm_compute_batch, sgm_get_tap and sgm_trim report it as well. */
int sgm_check(sgm_engine *e);
/* destroys the internal engines a batch entry created (and their device memory) and the page-locked staging buffers;
 * the engine's own buffers stay.  Blocks until the engine's stream is idle. */
int sgm_trim(sgm_engine *e);

/* per-stage HIP-event timing of the last compute (requires SGM_OPT_PROFILE = 1) */
int sgm_get_stage_times(sgm_engine *e, sgm_stage_times *out);
/* algorithmic HBM bytes of one compute at this shape (SURVEY.md 8d model, stated in DESIGN.md) */
int64_t sgm_algorithmic_bytes(const sgm_params *params, int H, int W, int with_reproject);

#ifdef __cplusplus
}
#endif
#endif

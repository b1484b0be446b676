/* abi_smoke.c -- the drop-in boundary exercised from plain C: no Python, no torch, nothing but include/sgm_hip.h, dlopen
 * and the CPU oracle as the checker.  Built and run by tests/test_c_abi.py.
 *
 *   abi_smoke <libsgm_hip.so> symbols            every entry point the header declares resolves; geometry and parameter
 *                                                validation work without a GPU
 *   abi_smoke <libsgm_hip.so> parity <liboracle> one small pair through sgm_create / sgm_compute / sgm_disp_to_float /
 *                                                sgm_reproject (the calls of /root/reference/main.ipynb:655-670, 697) and a
 *                                                batch of three through sgm_compute_batch in throughput mode, compared with
 *                                                oracle_sgbm_compute / oracle_reproject_f32 bit for bit (needs an MI355X)
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sgm_hip.h"
#include "../../oracle/sgbm_oracle.h"

#define LOAD(h, name)                                              \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) {                                               \
        fprintf(stderr, "missing symbol %s\n", #name);             \
        return 2;                                                  \
    }

/* the synthetic pair of a counter-based generator (not the bench's: any textured pair with a shift will do here) */
static uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static void make_pair(uint8_t *l, uint8_t *r, int H, int W, int shift, uint32_t seed)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            /* smooth-ish texture: average of hashed cells at two scales */
            const uint32_t a = mix(seed + (uint32_t)(y / 4) * 7919u + (uint32_t)(x / 4)) & 0xff;
            const uint32_t b = mix(seed * 3u + (uint32_t)(y / 16) * 104729u + (uint32_t)(x / 16)) & 0xff;
            l[y * W + x] = (uint8_t)((a + 3 * b) / 4);
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int xs = x + shift + (y > H / 2 ? 3 : 0);
            r[y * W + x] = xs < W ? l[y * W + xs] : (uint8_t)(mix(seed + 77u + (uint32_t)(y * W + x)) & 0xff);
        }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror());
        return 2;
    }
    LOAD(h, sgm_abi_version) LOAD(h, sgm_device_count) LOAD(h, sgm_last_error) LOAD(h, sgm_create) LOAD(h, sgm_destroy)
    LOAD(h, sgm_set_option) LOAD(h, sgm_geometry) LOAD(h, sgm_compute) LOAD(h, sgm_compute_batch) LOAD(h, sgm_disp_to_float)
    LOAD(h, sgm_reproject) LOAD(h, sgm_valid_mask) LOAD(h, sgm_compact_points) LOAD(h, sgm_median3x3) LOAD(h, sgm_filter_speckles)
    LOAD(h, sgm_get_tap) LOAD(h, sgm_get_headroom) LOAD(h, sgm_init_undistort_rectify_map) LOAD(h, sgm_remap_linear_u8)
    LOAD(h, sgm_compute_device) LOAD(h, sgm_disp_to_float_device) LOAD(h, sgm_reproject_device) LOAD(h, sgm_valid_mask_device)
    LOAD(h, sgm_compact_points_device) LOAD(h, sgm_compact_points_device_async) LOAD(h, sgm_init_undistort_rectify_map_device)
    LOAD(h, sgm_remap_linear_u8_device) LOAD(h, sgm_pipeline_device) LOAD(h, sgm_pipeline_batch_device) LOAD(h, sgm_synchronize)
    LOAD(h, sgm_check) LOAD(h, sgm_trim) LOAD(h, sgm_get_stage_times) LOAD(h, sgm_algorithmic_bytes)
    if (p_sgm_abi_version() != SGM_ABI_VERSION) {
        fprintf(stderr, "ABI version %d, header says %d\n", p_sgm_abi_version(), SGM_ABI_VERSION);
        return 2;
    }
    /* the notebook's keyword arguments (main.ipynb:655-666) at a small size */
    sgm_params p = {0, 64, 7, 8 * 3 * 49, 32 * 3 * 49, 1, 63, 10, 100, 32, 1};
    int minX1 = -1, W1 = -1;
    if (p_sgm_geometry(&p, 480, &minX1, &W1) != SGM_OK || minX1 != 64 || W1 != 416) return 3;
    sgm_params bad = p;
    bad.numDisparities = 24;
    if (p_sgm_geometry(&bad, 480, NULL, NULL) != SGM_ERR_UNSUPPORTED || !strstr(p_sgm_last_error(), "divisible by 16")) return 3;
    if (strcmp(argv[2], "symbols") == 0) {
        printf("C ABI v%d: 34 entry points resolve, geometry and validation answer without a GPU\n", p_sgm_abi_version());
        return 0;
    }
    if (argc < 4) return 64;
    void *o = dlopen(argv[3], RTLD_NOW | RTLD_LOCAL);
    if (!o) {
        fprintf(stderr, "dlopen %s: %s\n", argv[3], dlerror());
        return 2;
    }
    LOAD(o, oracle_sgbm_compute) LOAD(o, oracle_disp_to_float) LOAD(o, oracle_reproject_f32)
    enum { H = 72, W = 480, N = 3 };
    uint8_t *L = malloc((size_t)N * H * W), *R = malloc((size_t)N * H * W);
    int16_t *got = malloc((size_t)N * H * W * 2), *want = malloc((size_t)N * H * W * 2);
    float *gf = malloc((size_t)H * W * 4), *wf = malloc((size_t)H * W * 4), *gx = malloc((size_t)H * W * 12), *wx = malloc((size_t)H * W * 12);
    for (int i = 0; i < N; i++) make_pair(L + (size_t)i * H * W, R + (size_t)i * H * W, H, W, 9 + 4 * i, 1234u + (uint32_t)i);
    oracle_sgbm_params op = {p.minDisparity, p.numDisparities, p.blockSize, p.P1, p.P2, p.disp12MaxDiff, p.preFilterCap,
                             p.uniquenessRatio, p.speckleWindowSize, p.speckleRange, p.mode};
    for (int i = 0; i < N; i++) {
        oracle_sgbm_taps t;
        memset(&t, 0, sizeof t);
        if (p_oracle_sgbm_compute(&op, L + (size_t)i * H * W, R + (size_t)i * H * W, H, W, W, want + (size_t)i * H * W, &t) != 0 || !t.headroom_ok) return 5;
    }
    {
        long nv = 0;
        for (int k = 0; k < H * W; k++) nv += want[k] >= 0;
        fprintf(stderr, "oracle: %ld of %d pixels valid in pair 0\n", nv, H * W);
    }
    if (p_sgm_device_count() < 1) {
        fprintf(stderr, "no GPU: %s\n", p_sgm_last_error());
        return 4;
    }
    sgm_engine *e = NULL;
    if (p_sgm_create(&p, 0, NULL, &e) != SGM_OK) {
        fprintf(stderr, "sgm_create: %s\n", p_sgm_last_error());
        return 4;
    }
    /* stereo.compute(imgL, imgR) */
    if (p_sgm_compute(e, L, R, H, W, W, got) != SGM_OK) {
        fprintf(stderr, "sgm_compute: %s\n", p_sgm_last_error());
        return 6;
    }
    long nbad = 0, nvalid = 0;
    for (int k = 0; k < H * W; k++) {
        nbad += got[k] != want[k];
        nvalid += got[k] >= 0;
    }
    /* .astype(float32) / 16 ; mask ; reprojectImageTo3D with the notebook's Q scaled to this width */
    const double s = W / 3840.0;
    const double Q[16] = {1, 0, 0, -1909.9754 * s, 0, 1, 0, -1057.74529 * s, 0, 0, 0, 2045.48384 * s, 0, 0, -1.0, 0};
    if (p_sgm_disp_to_float(e, got, (int64_t)H * W, gf) != SGM_OK || p_sgm_reproject(e, gf, H, W, Q, 0, gx) != SGM_OK) return 6;
    p_oracle_disp_to_float(want, wf, (int64_t)H * W);
    p_oracle_reproject_f32(wf, H, W, Q, 0, wx);
    long fbad = memcmp(gf, wf, (size_t)H * W * 4) != 0, xbad = 0;
    for (int k = 0; k < H * W * 3; k++) {
        if (isfinite(wx[k]) != isfinite(gx[k])) xbad++;
        else if (isfinite(wx[k]) && wx[k] != gx[k]) xbad++;
    }
    /* a batch of three in throughput mode through the host entry */
    if (p_sgm_set_option(e, SGM_OPT_SCHEDULE, 2) != SGM_OK || p_sgm_set_option(e, SGM_OPT_SWEEP_ROWS, 5) != SGM_OK) return 6;
    memset(got, 0x55, (size_t)N * H * W * 2);
    if (p_sgm_compute_batch(e, N, L, R, H, W, got, NULL, NULL) != SGM_OK) {
        fprintf(stderr, "sgm_compute_batch: %s\n", p_sgm_last_error());
        return 6;
    }
    long bbad = 0;
    for (size_t k = 0; k < (size_t)N * H * W; k++) bbad += got[k] != want[k];
    int hc = -1, hd = -1, hok = -1;
    if (p_sgm_get_headroom(e, &hc, &hd, &hok) != SGM_OK || p_sgm_check(e) != SGM_OK || p_sgm_trim(e) != SGM_OK) return 6;
    p_sgm_destroy(e);
    printf("C ABI parity: %ld of %d disparities differ (%ld valid), float map %s, %ld XYZ values differ, batch of %d: %ld differ, headroom ok=%d\n",
           nbad, H * W, nvalid, fbad ? "differs" : "identical", xbad, N, bbad, hok);
    return (nbad || fbad || xbad || bbad || hok != 1 || nvalid < H * W / 4) ? 1 : 0;
}

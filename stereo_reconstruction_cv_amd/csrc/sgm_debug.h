/* sgm_debug.h -- A/B switches of the engine (SGM_OPT_DEBUG, a bit mask).  NOT part of the drop-in boundary
 * (include/sgm_hip.h): measurement scaffolding for tools/ and tests/ only.  Results stay bit-exact for every bit
 * except SGM_DBG_SKIP_BOUNDARY_LOADS, which sgm_set_option refuses unless SGM_ALLOW_WRONG_RESULTS=1 is in the
 * environment.  Bits that selected kernels round 2 measured and rejected (4096: four-direction row kernel and
 * three-role grouped pre-pass for small D; 8192: that schedule for D <= 32 only; 16384: short prefetch blocks in
 * the upward pre-pass) went away with those kernels in round 3. */
#ifndef SGM_DEBUG_H
#define SGM_DEBUG_H

#define SGM_OPT_DEBUG 4 /* sgm_set_option(e, SGM_OPT_DEBUG, mask) */

enum {
    SGM_DBG_WTA_IN_LAST_PATH = 2,        /* winner-take-all fused into the last path kernel everywhere (pre-pass schedule) */
    SGM_DBG_NO_LANE_GROUPS = 4,          /* D <= 64 through the wave-per-pixel kernels */
    SGM_DBG_NARROW_VSUM = 8,             /* k_vsum_ring with 4 int16 per thread */
    SGM_DBG_PREPASS_3_LAUNCHES = 16,     /* boundary pre-pass as three launches of the single-direction kernel */
    SGM_DBG_NO_PREPASS_OVERLAP = 32,     /* MODE_HH: upward pre-pass on the main stream */
    SGM_DBG_SKIP_BOUNDARY_LOADS = 64,    /* the sweep's loader wave skips its HBM loads: timing only, results WRONG */
    SGM_DBG_FORK_PREPASS_EARLY = 128,    /* fork the upward pre-pass right after the cost stage */
    SGM_DBG_INT16_COST = 256,            /* int16 cost pipeline (k_hsum + k_vsum_ring) instead of the byte one */
    SGM_DBG_PREPASS_ONE_CHUNK = 512,     /* pre-pass in one chunk with the plain line-per-block layout */
    SGM_DBG_WTA_SEPARATE = 2048,         /* winner-take-all always as its own pass */
    SGM_DBG_SMALL_D_RECORD = 8192,       /* D <= 64, MODE_SGBM: per-row record + element-wise vertical kernel (k_prepass3_g + k_vert3_g) instead of k_lines3_g's volumes */
    SGM_DBG_IN_ROW_ON_MAIN_STREAM = 4096, /* D <= 64, MODE_SGBM: the left-to-right in-row path after the vertical kernel (S +=) instead of beside it */
    SGM_DBG_FIFTH_PATH_AFTER_SWEEP = 65536 /* MODE_SGBM, D <= 128: the fifth path after the sweep (S +=) instead of beside it */
};

#endif
